"""Dump per-kernel averages of every counter in a rocprofv3 --pmc rocpd database (dev tool)."""
import re
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "gemm|attn|vq_nn"
acc = defaultdict(lambda: defaultdict(list))
for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
    if re.search(pat, name):
        short = re.sub(r"\(.*", "", re.sub(r"^void ", "", name))
        acc[short][cn].append(val)
for k, d in sorted(acc.items()):
    print(k)
    for cn, v in sorted(d.items()):
        print(f"    {cn:32s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")
