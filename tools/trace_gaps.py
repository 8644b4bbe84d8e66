"""Idle time between consecutive kernels of the timed steps in a rocprofv3 kernel trace (csv): python tools/trace_gaps.py <dir> <steps> <warmup>"""
import csv, glob, os, sys
src, steps, warmup = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
trace = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(trace))]
rows.sort()
marks = [i for i, r in enumerate(rows) if "patchify_kernel" in r[2]]           # one per step (first kernel of the ViT)
first = marks[-steps]
seg = rows[first:]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0]
gaps = [(seg[i + 1][0] - seg[i][1], seg[i][2][:60], seg[i + 1][2][:60]) for i in range(len(seg) - 1)]
idle = sum(max(g[0], 0) for g in gaps)
print(f"{steps} timed steps: span {span/1e6/steps:.2f} ms/step, kernel-busy {busy/1e6/steps:.2f} ms/step, idle between kernels {idle/1e6/steps:.2f} ms/step over {len(seg)//steps} launches/step")
big = sorted(gaps, key=lambda g: -g[0])[:12]
for g in big:
    print(f"  gap {g[0]/1e3:8.1f} us after {g[1]}  before {g[2]}")
import collections
by = collections.defaultdict(lambda: [0, 0])
for g in gaps:
    by[g[2]][0] += 1; by[g[2]][1] += max(g[0], 0)
print("idle time by the kernel that follows the gap:")
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t/1e6/steps:6.3f} ms/step over {n/steps:6.1f} gaps/step (avg {t/n/1e3:5.1f} us)  {k}")
