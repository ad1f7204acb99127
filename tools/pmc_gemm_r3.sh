#!/bin/bash
# PMC passes (counters only) over one launch set of tools/gemm_pmc_driver.py -> gpurun_out/pmc_gemm_r3.txt
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out; cd /tmp
rm -rf /tmp/pg1 /tmp/pg2
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS -d /tmp/pg1 -o a -- python $R/tools/gemm_pmc_driver.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE -d /tmp/pg2 -o a -- python $R/tools/gemm_pmc_driver.py > /dev/null 2>&1
python - > $R/gpurun_out/pmc_gemm_r3.txt <<'PY'
import sqlite3, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/pg1", "/tmp/pg2"):
    for db in glob.glob(d + '/**/*.db', recursive=True):
        c = sqlite3.connect(db)
        try:
            rows = list(c.execute("select kernel_name, counter_name, value from counters_collection"))
        except Exception as e:
            print("  (no counters in", db, e, ")"); continue
        for name, cn, val in rows:
            if 'gemm_w256' in name: acc[name.split('(')[0]][cn].append(val)
for k, d in sorted(acc.items()):
    print(k)
    wc = sum(d.get('SQ_WAVE_CYCLES', [1])) / max(len(d.get('SQ_WAVE_CYCLES', [1])), 1)
    for cn, v in sorted(d.items()): print(f"   {cn:28s} {sum(v)/len(v):16.0f}  ({100*sum(v)/len(v)/wc:6.1f}% of SQ_WAVE_CYCLES)  n={len(v)}")
PY
cat $R/gpurun_out/pmc_gemm_r3.txt
