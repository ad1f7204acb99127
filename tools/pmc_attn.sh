#!/bin/bash
# PMC passes (counters only, no tracing) over the attention kernels at the bench shape, one kernel family per run:
#   bash tools/pmc_attn.sh "1,1,1" "5,3,2"   -> gpurun_out/pmc_attn_<family>.txt
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out; cd /tmp
for FAM in "$@"; do
  TAG=$(echo $FAM | tr -d ,)
  rm -rf /tmp/pa1 /tmp/pa2
  ROUNDS=1 ITERS=3 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS -d /tmp/pa1 -o a -- python $R/tools/attn_lab.py $FAM > /dev/null 2>&1
  ROUNDS=1 ITERS=3 timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE -d /tmp/pa2 -o a -- python $R/tools/attn_lab.py $FAM > /dev/null 2>&1
  python - "$FAM" > $R/gpurun_out/pmc_attn_$TAG.txt <<'PY'
import sqlite3, glob, collections, sys
print("kernel family", sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/pa1", "/tmp/pa2"):
    for db in glob.glob(d + '/**/*.db', recursive=True):
        c = sqlite3.connect(db)
        try:
            rows = list(c.execute("select kernel_name, counter_name, value from counters_collection"))
        except Exception as e:
            print("  (no counters in", db, e, ")"); continue
        for name, cn, val in rows:
            if 'attn' in name: acc[name.split('(')[0]][cn].append(val)
for k, d in sorted(acc.items()):
    print(k)
    wc = sum(d.get('SQ_WAVE_CYCLES', [1])) / max(len(d.get('SQ_WAVE_CYCLES', [1])), 1)
    for cn, v in sorted(d.items()): print(f"   {cn:28s} {sum(v)/len(v):16.0f}  ({100*sum(v)/len(v)/wc:6.1f}% of SQ_WAVE_CYCLES)  n={len(v)}")
PY
  cat $R/gpurun_out/pmc_attn_$TAG.txt
done
