"""Per-phase deltas of the counters collected by tools/pmc_phases.sh (fused frame kernel, mean per launch)."""
import csv, collections, sys
tag = sys.argv[1]
order = ["0", "1", "7", "2", "3", "4", "5", "6", "full"]
names = {"0": "entry", "1": "init", "7": "cull+1a", "2": "1b", "3": "ties", "4": "list+plane+3d2d", "5": "NN", "6": "refit", "full": "finalize"}
rows = {}
for k in order:
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    try:
        for r in csv.DictReader(open(f"gpurun_out/{tag}/{k}/pmc_counter_collection.csv")):
            if "frame_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    except FileNotFoundError:
        continue
    rows[k] = {c: acc[c] / n[c] for c in acc}
cs = sorted(next(iter(rows.values())).keys())
print("%-18s" % "phase", " ".join("%22s" % c for c in cs))
prev = {c: 0.0 for c in cs}
for k in order:
    if k not in rows: continue
    print("%-18s" % names[k], " ".join("%22.3e" % (rows[k].get(c, 0) - prev[c]) for c in cs))
    prev = {c: rows[k].get(c, 0) for c in cs}
print("%-18s" % "TOTAL", " ".join("%22.3e" % prev[c] for c in cs))
