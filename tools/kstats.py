"""dev tool: per-step kernel breakdown from a rocprofv3 kernel trace of tools/time_train.py (first / second half = BL6 / REF6)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
by = collections.defaultdict(list)
for r in rows:
    by[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for half in (0, 1):
    tot = []
    for k, v in by.items():
        h = v[:len(v) // 2] if half == 0 else v[len(v) // 2:]
        tot.append((sum(h) / steps / 1e3, len(h) / steps, k))
    tot.sort(reverse=True)
    print("== half", half, "sum us/step", round(sum(t for t, _, _ in tot), 1))
    for t, n, k in tot[:12]:
        print(f"{t:9.1f} us/step {n:5.1f} calls/step {t/max(n,1e-9):9.1f} us/call  {k[:80]}")
