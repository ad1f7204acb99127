"""Steady-state per-step kernel table from TWO rocprofv3 --kernel-trace --stats summaries of the same command at different --steps (tools/rocpd_summary.py stats):
the difference removes start-up work (warm-up passes, HIP-graph capture, parameter broadcasts / clones).
Usage: python tools/kstats_steady.py <few_steps.csv> <many_steps.csv> <steps_difference> [<out.csv>]"""
import csv
import sys


def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationUs"])) for r in csv.DictReader(open(p))}


a, b, n = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3])
rows = []
for k, (cb, tb) in b.items():
    ca, ta = a.get(k, (0, 0.0))
    if cb - ca > 0:
        rows.append((k, (cb - ca) / n, (tb - ta) / n))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print(f"steady state: {tot / 1e3:.3f} ms of kernel time per step, {sum(r[1] for r in rows):.0f} launches per step")
for k, c, t in rows[:28]:
    print(f"{100 * t / tot:6.2f}%  {c:8.1f} x {t / max(c, 1e-9):9.2f} us = {t / 1e3:7.3f} ms   {k[:100]}")
if len(sys.argv) > 4:
    with open(sys.argv[4], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "CallsPerStep", "UsPerStep", "AverageUs", "Percentage"])
        for k, c, t in rows:
            w.writerow([k, f"{c:.1f}", f"{t:.1f}", f"{t / max(c, 1e-9):.2f}", f"{100 * t / tot:.2f}"])
