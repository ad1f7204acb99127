#!/bin/bash
R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS -d /tmp/pa -o a -- python $R/tools/attn_only.py > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob, collections
db = glob.glob('/tmp/pa/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
    if 'attn' in name: acc[name.split('(')[0]][cn].append(val)
for k, d in acc.items():
    print(k)
    wc = sum(d['SQ_WAVE_CYCLES'])/len(d['SQ_WAVE_CYCLES'])
    for cn, v in sorted(d.items()): print(f"   {cn:28s} {sum(v)/len(v):16.0f}  ({100*sum(v)/len(v)/wc:5.1f}% of wave cycles)")
PY
