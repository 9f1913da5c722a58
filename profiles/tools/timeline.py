"""One steady-state substep of a rocprofv3 --kernel-trace CSV as a timeline: per launch the queue, start relative to the substep's
first kernel, duration and the idle time of ITS queue before it; then the substep's span, busy time per queue and the idle total.
   python profiles/tools/timeline.py <trace dir> [substep index from the start, default 25: inside bench.py's timed region] [first-kernel substring, default closure_lds]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
idx = int(sys.argv[2]) if len(sys.argv) > 2 else 25
first = sys.argv[3] if len(sys.argv) > 3 else "closure_lds"
kern = []
for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(fn) as f:
        for r in csv.DictReader(f):
            kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
kern.sort()


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:60]


# substep starts: a `first` kernel whose predecessor in time is not a `first` kernel
starts = [i for i, k in enumerate(kern) if first in k[2] and (i == 0 or first not in kern[i - 1][2])]
# (the edge + interior closure launches are two `first` kernels with a pack kernel in between: merge starts closer than 3 launches)
merged = []
for s in starts:
    if merged and s - merged[-1] <= 3:
        continue
    merged.append(s)
starts = merged
if len(starts) < idx + 2:
    print("too few substeps in the trace:", len(starts))
    sys.exit(1)
a, b = starts[idx], starts[idx + 1]
sub = kern[a:b]
t0 = sub[0][0]
last_end = {}
busy = {}
print(f"{'queue':>6} {'start_us':>9} {'dur_us':>8} {'idle_before_us':>14}  kernel")
for s, e, n, q in sub:
    idle = (s - last_end[q]) / 1e3 if q in last_end else 0.
    print(f"{q:>6} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {idle:14.1f}  {short(n)}")
    last_end[q] = e
    busy[q] = busy.get(q, 0) + (e - s)
span = (kern[b][0] - t0) / 1e3
# union of busy intervals over all queues
iv = sorted((s, e) for s, e, _, _ in sub)
u, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        u += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
u += ce - cs
print(f"substep span {span:.1f} us, {len(sub)} launches; busy per queue: " + ", ".join(f"{q}: {v / 1e3:.1f}" for q, v in busy.items()))
print(f"time with no kernel running on any queue: {span - u / 1e3:.1f} us")
