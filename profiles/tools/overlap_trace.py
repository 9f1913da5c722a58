"""Reads a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV set and reports how much of every ghost-row pack / unpack kernel
and exchange copy ran while a compute kernel of the main stream was running (time overlap), per compute kernel name."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
kern = []
for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(fn) as f:
        for r in csv.DictReader(f):
            kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
copies = []
for fn in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    with open(fn) as f:
        for r in csv.DictReader(f):
            copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", ""), ""))
kern.sort()
mq = [k[3] for k in kern if "mom_lds_kernel" in k[2]]
main_q = mq[0] if mq else max(set(k[3] for k in kern), key=lambda q: sum(1 for k in kern if k[3] == q))      # the compute stream's queue
# what runs on the communication stream: pack / unpack kernels and the copies that stand in for the exchange
comm_q = set(k[3] for k in kern if k[3] != main_q and "halo_pack_kernel" in k[2])
halo = [k for k in kern if k[3] in comm_q and ("halo_pack_kernel" in k[2] or "halo_unpack_kernel" in k[2] or "copyBuffer" in k[2])]
if len(sys.argv) > 2 and sys.argv[2] == "nccl":      # RCCL's own kernels instead (a real communicator: pipe_trace.sh)
    halo = [k for k in kern if k[3] != main_q and ("nccl" in k[2].lower() or "rccl" in k[2].lower())]
    print("RCCL kernels:", sorted(set(short_ for short_ in (k[2][:50] for k in halo))), "on queues", sorted(set(k[3] for k in halo)))
comp = [k for k in kern if k[3] == main_q and "halo_" not in k[2] and "__amd_rocclr" not in k[2]]
print(f"{len(kern)} kernel records, {len(halo)} pack / unpack launches, {len(copies)} copies")
queues = sorted(set(k[3] for k in kern))
print("queues:", queues, "| pack / unpack kernels on:", sorted(set(k[3] for k in halo)), "| compute on:", sorted(set(k[3] for k in comp)))


def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("<")[0].split("(")[0]


tot, ov, by = 0, 0, defaultdict(int)
for s, e, n, q in halo:
    tot += e - s
    for cs, ce, cn, cq in comp:
        if ce <= s:
            continue
        if cs >= e:
            break
        o = min(e, ce) - max(s, cs)
        if o > 0:
            ov += o
            by[short(cn)] += o
print(f"communication-stream kernel time (pack, exchange copy, unpack) {tot / 1e3:.1f} us, of which {ov / 1e3:.1f} us ({100. * ov / max(tot, 1):.0f} %) while a compute kernel of another queue was running")
for n, o in sorted(by.items(), key=lambda t: -t[1])[:8]:
    print(f"   beside {n}: {o / 1e3:.1f} us")
