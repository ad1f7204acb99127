#!/bin/bash
# counter passes (counters only, one pass per set) over tools/gemm_landing_pmc.py -> gpurun_out/r5/gemm_landing_pmc.txt
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r5; cd /tmp
rocprofv3 -L > /tmp/avail.txt 2>&1
grep -o -E "\b(TCC|TCP)_[A-Z0-9_]+(_sum)?\b" /tmp/avail.txt | sort -u > $R/gpurun_out/r5/avail_tcc_tcp.txt
wc -l $R/gpurun_out/r5/avail_tcc_tcp.txt
have() { grep -qx "$1" $R/gpurun_out/r5/avail_tcc_tcp.txt; }
pass() {  # name, counters...
  local tag=$1; shift; local ok=""
  for c in "$@"; do if have $c; then ok="$ok $c"; else echo "  (counter $c not available)"; fi; done
  [ -z "$ok" ] && return
  rm -rf /tmp/lp_$tag
  timeout 300 rocprofv3 --pmc $ok -d /tmp/lp_$tag -o a -- python $R/tools/gemm_landing_pmc.py > /tmp/lp_$tag.out 2>/tmp/lp_$tag.err || { echo "pass $tag failed"; tail -3 /tmp/lp_$tag.err; }
}
pass a TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass b TCC_TAG_STALL_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass c TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum
pass d TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum
python - <<'PY' | tee $R/gpurun_out/r5/gemm_landing_pmc.txt
import sqlite3, glob, collections, json, re
plan = None
for f in glob.glob('/tmp/lp_*.out'):
    for ln in open(f):
        if ln.startswith('PLAN '): plan = json.loads(ln[5:])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob('/tmp/lp_?')):
    for db in glob.glob(d + '/**/*.db', recursive=True):
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        order = 'dispatch_id' if 'dispatch_id' in cols else 'rowid'
        rows = list(c.execute(f"select {order}, kernel_name, counter_name, value from counters_collection order by {order}"))
        per = collections.OrderedDict()
        for did, name, cn, val in rows:
            if 'gemm_bf16' not in name: continue
            per.setdefault(did, {})[cn] = per.setdefault(did, {}).get(cn, 0.0) + val
            per[did]['_k'] = name.split('(')[0]
        if plan and len(per) == len(plan):
            for (did, rec), label in zip(per.items(), plan):
                for cn, v in rec.items():
                    if cn != '_k': acc[label + ' | ' + re.sub(r'^void ', '', rec['_k'])][cn].append(v)
        else:
            print('plan / dispatch mismatch in', db, len(per), len(plan) if plan else None, cols)
print(f"{'role | variant | kernel':92s} " + "counter = mean over 3 launches")
for k, d in acc.items():
    cells = []
    h, m = d.get('TCC_HIT_sum'), d.get('TCC_MISS_sum')
    if h and m: cells.append(f"L2 hit rate {sum(h)/(sum(h)+sum(m)):.4f}")
    for cn, v in sorted(d.items()): cells.append(f"{cn} {sum(v)/len(v):.4g}")
    print(f"{k:92s} " + " ; ".join(cells))
PY
