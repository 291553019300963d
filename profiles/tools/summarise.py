"""Summarise the rocprofv3 outputs of profile_chain.sh into <out>/chain_kernel_stats.csv and chain_hbm_pmc.json.

rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) gfx950's
FETCH_SIZE is exactly half of the bytes of wide coalesced reads (128-B requests tallied at 64 B) and has to be
doubled; WRITE_SIZE is exact.  `bytes_per_dispatch_corrected` applies that.
"""
import collections
import csv
import glob
import json
import re
import sys

out = sys.argv[1]


def kname(full):
    m = re.search(r"(\w+_kernel(<[^>(]*>)?)", full)
    return m.group(1) if m else re.sub(r"\(.*", "", full)[:48]


stats = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(out + "/chain_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "percent"])
        for r in rows:
            w.writerow([kname(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]])

pmc = {}
for cname, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    acc = collections.defaultdict(list)
    for path in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == cname:
                acc[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    pmc[cname] = {
        k: {
            "dispatches": len(v),
            "mean_KiB_per_dispatch": round(sum(v) / len(v), 1),
            "bytes_per_dispatch_corrected": round((2 if cname == "FETCH_SIZE" else 1) * 1024 * sum(v) / len(v)),
        }
        for k, v in acc.items()
    }
pmc["n_frames_per_launch"] = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
json.dump(pmc, open(out + "/chain_hbm_pmc.json", "w"), indent=1)
print(json.dumps({c: {k: v["bytes_per_dispatch_corrected"] for k, v in d.items() if "wmx" in k or "_kernel" in k} for c, d in pmc.items() if isinstance(d, dict)}))
