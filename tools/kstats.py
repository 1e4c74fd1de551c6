"""dev tool: per-step summary of a rocprofv3 kernel_stats.csv.  usage: python tools/kstats.py CSV [steps [rows]]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
top = int(sys.argv[3]) if len(sys.argv) > 3 else 16
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("kernel time per step: %.3f ms" % (tot / steps / 1e6))
for r in rows[:top]:
    print(f'{r["Name"][:72]:72s} {int(r["Calls"]) / steps:5.1f} x {float(r["AverageNs"]) / 1e3:8.1f} us = {int(r["TotalDurationNs"]) / steps / 1e3:8.1f} us/step')
