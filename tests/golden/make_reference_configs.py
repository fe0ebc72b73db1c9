"""VALUES held by the reference's own configuration files -> tests/golden/reference_configs.json (build container only:
/root/reference does not exist on the GPU box). Parsed here with PyYAML (yaml.safe_load after dropping OpenCV's "%YAML:1.0"
directive line) — an independent reader — so that the tests can assert, without /root/reference:
  * csrc/iba_config.cpp reads the six config/calib/NN/iba_calib_global.yml exactly (every io / orb / runtime entry);
  * iba_default_params + the yml overrides, iba_default_mads_options (lb / ub / init_frame / min_mesh / max_bbeval / he_threshold /
    valid_rate) equal what the reference runs with;
  * csrc/iba_io.cpp's OpenCV-YAML reader gets the intrinsics of config/orb_ori/*.yaml right (the `%YAML:1.0` FileStorage dialect
    of KeyFrames/NNNNNN.yml).
Only parsed VALUES are stored (a JSON tree of numbers / strings / lists) together with the yml TEXT's sha256, not the files."""
import hashlib
import json
import os
import sys

import yaml

REF = "/root/reference/config"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(path):
    txt = open(path).read()
    import re
    # (OpenCV's FileStorage dialect also accepts "key:value" without the space YAML wants: config/orb_ori/KITTI00-02.yaml:49)
    body = "\n".join(re.sub(r"^([A-Za-z_][\w.]*):(\S)", r"\1: \2", l) for l in txt.split("\n") if not l.startswith("%YAML"))
    return yaml.safe_load(body), hashlib.sha256(txt.encode()).hexdigest()


out = {"calib": {}, "orb": {}, "source": "gitouni/Spatial-Temporal-LiDAR-camera-Calibration config/ (values only)"}
for seq in sorted(os.listdir(os.path.join(REF, "calib"))):
    p = os.path.join(REF, "calib", seq, "iba_calib_global.yml")
    if os.path.exists(p):
        tree, h = load(p)
        out["calib"][seq] = {"file": "config/calib/%s/iba_calib_global.yml" % seq, "sha256": h, "values": tree}
for name in sorted(os.listdir(os.path.join(REF, "orb_ori"))):
    if name.endswith(".yaml"):
        tree, h = load(os.path.join(REF, "orb_ori", name))
        out["orb"][name] = {"file": "config/orb_ori/%s" % name, "sha256": h, "values": tree}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "reference_configs.json"), "w"), indent=1, sort_keys=True)
print({k: len(v) for k, v in out.items() if isinstance(v, dict)})
