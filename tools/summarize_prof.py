"""Condense the rocprofv3 outputs of tools/prof_run.sh (gpurun_out/<tag>/) into the tracked profiles/ directory:
profiles/<tag>_rocprof_summary.md (kernel stats + counters per launch) and profiles/pmc_latest.json (what bench.py quotes as
roofline.traffic / issue_frac, stamped with the git head and a hash of the kernel sources so that it is dropped when stale).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by exactly 2x
(MI355X_MICROARCH.md, HBM) -> both the raw and the x2-corrected read figure are recorded, the corrected one is used.
usage: python tools/summarize_prof.py <tag>"""
import csv, glob, hashlib, json, os, subprocess, sys, time
from collections import defaultdict

tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", tag)
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)
N_SIMD = 1024


def short(name):
    return name.replace("void ", "").replace("iba::", "").split("(")[0]


def counters(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for fn in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


rows = list(csv.DictReader(open(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0])))
lines = ["# rocprofv3 --kernel-trace --stats summary (%s): python3 bench.py --steps 15 --warmup 2 --no-cpu-baseline --no-extras" % tag, "",
         "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
avg_us = {}
for r in rows:
    avg_us[short(r["Name"])] = float(r["AverageNs"]) / 1e3
    lines.append("| %s | %s | %.1f | %.1f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
fetch, write, insts, cyc, mfma, l2 = counters("fetch"), counters("write"), counters("insts"), counters("cycles"), counters("mfma"), counters("l2")
kernels = [k for k in avg_us if k.startswith(("iba_assoc", "iba_pairs", "iba_nn_kernel", "iba_anchor", "iba_factor", "iba_reduce2", "iba_he_kernel", "iba_fetch"))]
lines += ["", "## counters per launch (separate --pmc passes on the same command; KiB -> bytes, FETCH_SIZE x2 on gfx950)", "",
          "| kernel | HBM read MB (x2) | HBM write MB | SQ_INSTS_VALU | SQ_INSTS_SALU | SQ_INSTS_LDS | SQ_INSTS_VMEM | VALU-active share of SIMD cycles | SQ_WAIT_ANY / SQ_WAVE_CYCLES | MFMA F64 insts | L2 requests (TCC_REQ) | L2 hit share | L2 requests / s |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
tot_hbm = 0.0; act = 0.0; gui = 0.0
step_hbm = 0.0; step_act = 0.0; step_gui = 0.0; per_kernel = {}
for k in kernels:
    f = 2 * fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024; w = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
    i, c, m = insts.get(k, {}), cyc.get(k, {}), mfma.get(k, {})
    share = 4 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (N_SIMD * c["GRBM_GUI_ACTIVE"] / 8) if c.get("GRBM_GUI_ACTIVE") else float("nan")
    wait = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else float("nan")
    q = l2.get(k, {})
    req, hit = q.get("TCC_REQ_sum", 0.0), q.get("TCC_HIT_sum", 0.0)
    req_rate = req / (avg_us[k] * 1e-6) if avg_us.get(k) else float("nan")
    lines.append("| %s | %.2f | %.2f | %.3g | %.3g | %.3g | %.3g | %.2f | %.2f | %.3g | %.3g | %.2f | %.3g |" % (k, f / 1e6, w / 1e6, i.get("SQ_INSTS_VALU", 0), i.get("SQ_INSTS_SALU", 0), i.get("SQ_INSTS_LDS", 0),
                                                                                        i.get("SQ_INSTS_VMEM", 0), share, wait, m.get("SQ_INSTS_VALU_MFMA_F64", 0), req, hit / req if req else float("nan"), req_rate))
    if k.startswith(("iba_assoc", "iba_pairs", "iba_nn_kernel")):
        tot_hbm += f + w; act += 4 * c.get("SQ_ACTIVE_INST_VALU", 0.0); gui += c.get("GRBM_GUI_ACTIVE", 0.0) / 8
    if not k.startswith("iba_anchor"):   # every kernel of a step (the anchor lists are built once, outside the timed steps)
        step_hbm += f + w; step_act += 4 * c.get("SQ_ACTIVE_INST_VALU", 0.0); step_gui += c.get("GRBM_GUI_ACTIVE", 0.0) / 8
    per_kernel[k] = {"avg_us": avg_us.get(k), "hbm_read_bytes": f, "hbm_write_bytes": w, "valu_active_share": share, "wait_any_share": wait, "insts_valu": i.get("SQ_INSTS_VALU", 0), "l2_requests": req, "l2_hit_share": hit / req if req else None, "l2_requests_per_s": req_rate}
bench_line = None
try:
    bench_line = json.loads([l for l in open(os.path.join(src, "bench_under_rocprof.json")) if l.startswith("{")][-1])
except Exception:
    pass
hsh = hashlib.sha256()
d = os.path.join(ROOT, "spatial-temporal-lidar-camera-calibration_amd", "csrc")
for fn in sorted(os.listdir(d)):
    if fn.endswith((".hip", ".hpp", ".cpp")):
        hsh.update(open(os.path.join(d, fn), "rb").read())
head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "spatial-temporal-lidar-camera-calibration_amd/csrc"], cwd=ROOT, capture_output=True, text=True).stdout.strip())
cfg = (bench_line or {}).get("config", {})
pmc = {"round": tag, "git_head": head + ("+uncommitted kernel changes" if dirty else ""), "source_stamp": hsh.hexdigest()[:16], "taken_on": time.strftime("%Y-%m-%d"),
       "frames": cfg.get("frames_this_rank"), "pts": cfg.get("points_per_frame"), "batch": cfg.get("candidates_per_step"),
       "kernels": "iba_pairs_kernel + iba_assoc2_kernel + iba_nn_kernel", "hbm_bytes_per_launch": tot_hbm, "valu_issue_frac": act / (N_SIMD * gui) if gui else None,
       "hbm_bytes_per_step": step_hbm, "valu_issue_frac_step": step_act / (N_SIMD * step_gui) if step_gui else None, "per_kernel": per_kernel,
       "counters": "FETCH_SIZE x2 + WRITE_SIZE (KiB); 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); TCC_REQ_sum / TCC_HIT_sum (64-byte L2 requests)",
       "note": "one rocprofv3 --pmc pass per counter list on `python3 bench.py --steps 15 --warmup 2 --no-cpu-baseline --no-extras`; mean per launch"}
json.dump(pmc, open(os.path.join(out_dir, "pmc_latest.json"), "w"), indent=1)
lines += ["", "association (pairs + assoc2) + search kernels per launch: HBM %.1f MB, VALU-active share of the SIMD cycles %.2f  (source stamp %s, head %s)" % (tot_hbm / 1e6, pmc["valu_issue_frac"] or float("nan"), pmc["source_stamp"], pmc["git_head"])]
if bench_line:
    lines += ["", "bench line under rocprofv3 (kernel trace): value %.0f evals/s, ms_per_step %.3f, roofline %s" % (bench_line["value"], bench_line["ms_per_step"], json.dumps({k: bench_line["roofline"][k] for k in ("achieved", "frac", "launch_ms", "kernel_ms")}))]
open(os.path.join(out_dir, "%s_rocprof_summary.md" % tag), "w").write("\n".join(lines) + "\n")
for fn in ("bench_under_rocprof.json", "bench_full.json"):
    p = os.path.join(src, fn)
    if os.path.exists(p):
        open(os.path.join(out_dir, "%s_%s" % (tag, fn)), "w").write(open(p).read())
print("\n".join(lines))
