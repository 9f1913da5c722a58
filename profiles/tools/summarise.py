"""Condense rocprofv3 output (profiles/tools/collect.sh) into per-kernel tables.

kernel_stats.csv : name, calls, avg/min/max duration (us) from the --stats pass
pmc_summary.json : per kernel, per counter: mean value per launch; FETCH_SIZE/WRITE_SIZE converted to bytes
                   as MI355X_MICROARCH.md prescribes (FETCH_SIZE is tallied in 32-B units on gfx950 while the
                   stock metric assumes 64 B -> x2 correction; WRITE_SIZE in 64-B units as reported... see below).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\s*\[clone.*", "", name)
    m = re.match(r"(?:void\s+)?([A-Za-z_0-9:]+(?:<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:90]


# ---- kernel stats
rows = []
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append(r)
if rows:
    with open(os.path.join(out, "kernel_stats.csv"), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent"])
        for r in sorted(rows, key=lambda r: -float(r.get("TotalDurationNs", 0))):
            w.writerow([short(r["Name"]), r["Calls"], "%.1f" % (float(r["TotalDurationNs"]) / 1e3),
                        "%.2f" % (float(r["AverageNs"]) / 1e3), "%.2f" % (float(r["MinNs"]) / 1e3),
                        "%.2f" % (float(r["MaxNs"]) / 1e3), r.get("Percentage", "")])

# ---- counters
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = short(r["Kernel_Name"])
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
summary = {}
for k, cs in acc.items():
    ent = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
    ent["launches_seen"] = max(v[1] for v in cs.values())
    # rocprofv3 reports FETCH_SIZE/WRITE_SIZE in kilobytes assuming 64-B requests; on gfx950 the fetch
    # tally is in 32-B units for the dominant request size -> the guide's x2 correction.  (Same treatment as
    # profiles/r01/pmc_traffic_*.json of the earlier passes.)
    if "FETCH_SIZE" in ent:
        ent["hbm_read_bytes"] = ent["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in ent:
        ent["hbm_write_bytes"] = ent["WRITE_SIZE"] * 1024
    summary[k] = ent
with open(os.path.join(out, "pmc_summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1, sort_keys=True)
print("kernels:", len(summary), "stats rows:", len(rows))

# ---- bench.py's traffic file (profiles/pmc_traffic.json): kernels keyed by workload and by bench.py's profile names.
# WORKLOAD_KEY = "<nx>x<nyl>x<nz>/<sgs>/nsv<n>" (bench.py's key), CELLS = cells per GPU; both set by collect.sh
# momentum: 88 B on RK stages 2 and 3, 64 B on stage 1 (um aliases u0 and is not read): 80 B averaged over whole RK3 steps
ALGO = {"closure": 40, "mom_truetruetruetrue": 80, "div_rhs": 32, "thomas": 24, "project_integrate": 72, "scalar": 48, "scalar_kappa_faces": 48, "scalar_lds_cd2": 48}
MAP = [("closure_lds_kernel", "closure"), ("mom_lds_kernel", "mom_truetruetruetrue"), ("div_rhs_kernel", "div_rhs"),
       ("thomas_lds_kernel", "thomas"), ("thomas_kernel", "thomas"), ("integrate_kernel", "project_integrate"),
       ("scalar_kernel", "scalar"), ("scalar_kappa_faces_kernel", "scalar_kappa_faces"), ("scalar_lds_kernel", "scalar_lds_cd2")]
key = os.environ.get("WORKLOAD_KEY", "256x256x256/vreman/nsv0")
cells = int(os.environ.get("CELLS", str(256 ** 3)))
nsv = int(key.rsplit("nsv", 1)[1])
traffic = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python bench.py "
                   "--no-cpu --no-pmc --no-dropin --steps 12 --warmup 3 [workload flags]` via profiles/tools/collect.sh; raw values in KiB; "
                   "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE doubled per MI355X_MICROARCH.md: "
                   "gfx950 tallies 128-B read requests as 64 B)",
           "workloads": {key: {}}}
for k, ent in summary.items():
    for pat, name in MAP:
        if k.startswith(pat) and "FETCH_SIZE" in ent and "WRITE_SIZE" in ent and name not in traffic["workloads"][key]:
            hb = ent["hbm_read_bytes"] + ent["hbm_write_bytes"]
            ab = (ALGO[name] + (24 * nsv if name == "project_integrate" else 0)) * cells
            traffic["workloads"][key][name] = {"FETCH_SIZE_KiB": ent["FETCH_SIZE"], "WRITE_SIZE_KiB": ent["WRITE_SIZE"],
                                               "hbm_bytes_per_launch": int(hb), "algorithmic_bytes_per_launch": ab,
                                               "ratio": round(hb / ab, 3), "source": os.environ.get("TRAFFIC_SOURCE", "")}
with open(os.path.join(out, "pmc_traffic.json"), "w") as fh:
    json.dump(traffic, fh, indent=1)
