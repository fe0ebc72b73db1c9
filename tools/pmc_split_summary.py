"""Per-phase counter deltas collected by tools/pmc_split.sh (mean per launch of iba_assoc_kernel / iba_nn_kernel)."""
import csv, collections, glob, sys
tag = sys.argv[1]
def load(k, pat):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for fn in glob.glob(f"gpurun_out/{tag}/{k}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    return {c: acc[c] / n[c] for c in acc}
for kern, pat, order, names in (
        ("iba_assoc_kernel", "iba_assoc_kernel", ["a1", "a2", "a3", "a4", "a5", "a6", "a7", "full"],
         {"a1": "init", "a2": "cull", "a3": "1a stream", "a4": "1b exact", "a5": "ties", "a6": "list", "a7": "planes", "full": "3d-2d + sums"}),
        ("iba_nn_kernel", "iba_nn_kernel", ["n1", "n2", "n4", "n5", "full"],
         {"n1": "nodes+counts", "n2": "init slots", "n4": "refill only", "n5": "+ steps", "full": "+ finish, sums"})):
    rows = {k: load(k, pat) for k in order}
    rows = {k: v for k, v in rows.items() if v}
    if not rows:
        continue
    cs = sorted(next(iter(rows.values())).keys())
    print(kern)
    print("%-16s" % "phase", " ".join("%20s" % c for c in cs))
    prev = {c: 0.0 for c in cs}
    for k in order:
        if k not in rows:
            continue
        print("%-16s" % names[k], " ".join("%20.3e" % (rows[k].get(c, 0) - prev[c]) for c in cs))
        prev = {c: rows[k].get(c, 0) for c in cs}
    print("%-16s" % "TOTAL", " ".join("%20.3e" % prev[c] for c in cs))
