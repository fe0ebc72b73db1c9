"""Condense rocprofv3 outputs under gpurun_out/ into the tracked profiles/ directory.
usage: python tools/summarize_prof.py <round-tag> <stats-dir> [<fetch-dir> <write-dir>]
FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by exactly 2x
(MI355X_MICROARCH.md §HBM) -> both the raw and the x2-corrected read figure are recorded."""
import csv, glob, json, os, sys
from collections import defaultdict

tag, stats = sys.argv[1], sys.argv[2]
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(out_dir, exist_ok=True)


def short(name):
    n = name.replace("void ", "").replace("iba::", "")
    return n.split("(")[0]


rows = list(csv.DictReader(open(glob.glob(os.path.join(stats, "**", "*_kernel_stats.csv"), recursive=True)[0])))
lines = ["# rocprofv3 --kernel-trace --stats summary (%s)" % tag, "", "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
for r in rows:
    lines.append("| %s | %s | %.1f | %.1f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
traffic = {}
if len(sys.argv) > 4:
    for cname, d in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
        acc = defaultdict(list)
        for r in csv.DictReader(open(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0])):
            if r["Counter_Name"] == cname:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            traffic.setdefault(k, {})[cname] = (sum(v) / len(v), len(v))
    lines += ["", "## HBM traffic per launch (separate --pmc passes, KiB -> bytes)", "", "| kernel | launches | FETCH_SIZE raw MB | FETCH x2 (gfx950 correction) MB | WRITE_SIZE MB |", "|---|---|---|---|---|"]
    for k, v in traffic.items():
        f = v.get("FETCH_SIZE", (0, 0)); w = v.get("WRITE_SIZE", (0, 0))
        lines.append("| %s | %d | %.2f | %.2f | %.2f |" % (k, f[1], f[0] * 1024 / 1e6, 2 * f[0] * 1024 / 1e6, w[0] * 1024 / 1e6))
    key = [k for k in traffic if k.startswith("iba_frame_kernel<3")] or [k for k in traffic if k.startswith("iba_frame_kernel<0")]
    if key:
        v = traffic[key[0]]
        json.dump({"round": tag, "kernel": key[0], "fetch_bytes_raw": v["FETCH_SIZE"][0] * 1024, "write_bytes": v["WRITE_SIZE"][0] * 1024,
                   "hbm_bytes_per_launch": 2 * v["FETCH_SIZE"][0] * 1024 + v["WRITE_SIZE"][0] * 1024,
                   "note": "read side doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 128-B requests at 64 B)"}, open(os.path.join(out_dir, "traffic_latest.json"), "w"), indent=1)
open(os.path.join(out_dir, "%s_rocprof_summary.md" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
