"""Register / LDS budget of every kernel as the compiler reports it (-Rpass-analysis=kernel-resource-usage through `make asm`),
written to profiles/<tag>_resource_usage.md. usage: python tools/resource_usage.py <tag>"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
csrc = os.path.join(ROOT, "spatial-temporal-lidar-camera-calibration_amd", "csrc")
out = subprocess.run(["make", "-B", "-C", csrc, "asm"], capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.split("\n"):
    m = re.search(r"remark: +(.*?) \[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": name.split("(")[0].replace("void ", "").replace("iba::", "")}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.rsplit(":", 1)
        cur[k.strip()] = v.strip()
lines = ["# kernel resource usage (%s): hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage" % tag, "",
         "| kernel | VGPRs | VGPR spills | SGPRs | SGPR spills | scratch B/lane | LDS B/block (static) | occupancy waves/SIMD |", "|---|---|---|---|---|---|---|---|"]
for r in rows:
    lines.append("| %s | %s | %s | %s | %s | %s | %s | %s |" % (r["name"], r.get("VGPRs", ""), r.get("VGPRs Spill", ""), r.get("TotalSGPRs", ""), r.get("SGPRs Spill", ""),
                                                               r.get("ScratchSize [bytes/lane]", ""), r.get("LDS Size [bytes/block]", ""), r.get("Occupancy [waves/SIMD]", "")))
lines += ["", "iba_assoc_kernel and iba_nn_kernel allocate their LDS dynamically (73-80 KB and ~13 KB per block at the bench shape); the occupancy the",
          "compiler prints for them is the register bound (the association kernel runs two blocks = 4 waves/SIMD per CU)."]
open(os.path.join(ROOT, "profiles", "%s_resource_usage.md" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
