"""rocprofv3 (ROCm 7.2) writes rocpd SQLite databases; this turns them into the small text artefacts kept under
profiles/:  kernel-trace --stats summary (per-kernel calls / total / average / share) and, for --pmc passes, the
per-launch HBM traffic of the GEMM instantiations (profiles/pmc_gemm.json).
Usage: python tools/rocpd_summary.py stats <results.db> <out.csv>
       python tools/rocpd_summary.py pmc <fetch.db> <write.db> <out.json> [<config> <batch>]
The pmc mode keys EVERY kernel by its symbol (template arguments kept, parameter list dropped) and records the config / per-GPU batch of the profiled
command, so that bench.py attaches `traffic` only to a run of the same workload (profiles/pmc_step.json)."""
import csv
import json
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    return name if len(name) < 100 else name[:97] + "..."


def stats(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for n, calls, tot, avg, pct in rows:
            w.writerow([short(n), calls, f"{tot:.1f}", f"{avg:.2f}", f"{pct:.2f}"])
    for n, calls, tot, avg, pct in rows[:14]:
        print(f"{pct:6.2f}%  {calls:6d} x {avg:10.2f} us   {short(n)[:90]}")


def kernel_key(name: str):
    """'void gemm_w256_kernel<true, true, 6>(Args...)' -> 'gemm_w256_kernel<true, true, 6>';  'attn_fwd_kernel(...)' -> 'attn_fwd_kernel'"""
    name = re.sub(r"^void ", "", name)
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def pmc(fetch_db, write_db, out, config=None, batch=None):
    res = {}
    for db, cname in ((fetch_db, "FETCH_SIZE"), (write_db, "WRITE_SIZE")):
        c = sqlite3.connect(db)
        for name, val in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (cname,)):
            k = kernel_key(name)
            if k and not k.startswith(("at::", "void at::")) and "elementwise_kernel" not in k:
                res.setdefault(k, {}).setdefault(cname, []).append(val)
    outd = {}
    for k, d in res.items():
        f = sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [])), 1)
        w = sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [])), 1)
        outd[k] = {"FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w, "hbm_bytes_per_launch": (2 * f + w) * 1024,
                   "launches": len(d.get("FETCH_SIZE", [])),
                   "note": "per-launch average over one bench.py step; reads doubled (gfx950 FETCH_SIZE reports half of a wide "
                           "coalesced stream, MI355X_MICROARCH.md §HBM); WRITE_SIZE uncalibrated"}
    doc = {"config": config, "batch": int(batch) if batch is not None else None, "kernels": outd} if config is not None else outd
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in sorted(outd.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch x {v['launches']:5d}   {k}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], *(sys.argv[5:7] if len(sys.argv) >= 7 else ()))
