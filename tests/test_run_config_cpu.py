"""Reference-held DATA as a pin (VERDICT r3 missing #1, #4): the reference's own configuration files — the six
config/calib/NN/iba_calib_global.yml and config/orb_ori/*.yaml — against what this library runs with.
tests/golden/reference_configs.json holds their VALUES (parsed by PyYAML in the build container, make_reference_configs.py).
Here: csrc/iba_config.cpp (the yaml-cpp subset main() reads) and csrc/iba_io.cpp's cv::FileStorage reader parse
  (a) the real files where /root/reference exists (the build container's CPU tier), byte for byte as the reference ships them, and
  (b) files re-emitted from the JSON in the reference's style (inline comments, flow sequences without spaces, "1.0E-6", OpenCV's
      "%YAML:1.0" head, "key:value" without a space) where it does not (the GPU box),
and every entry must equal the JSON; the library's built-in defaults (iba_default_params + yml overrides as abi.py hard-codes them,
iba_default_mads_options) must equal sequence 00's file."""
import ctypes as C
import hashlib
import importlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd")
abi = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.abi")
fmt = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.formats")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_configs.json")))
REF = os.environ.get("IBA_REFERENCE_DIR", "/root/reference")   # (the GPU box has none: the re-emitted files are parsed there)


def _emit_scalar(v):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, float):
        r = repr(v)
        return r.replace("e-0", "E-").replace("e-", "E-") if "e" in r else r
    return str(v)


def _emit_run_config(tree, path):
    with open(path, "w") as f:
        f.write("%YAML:1.0\n---\n")
        for sec in ("io", "orb", "runtime"):
            f.write("%s:\n" % sec)
            for i, (k, v) in enumerate(tree[sec].items()):
                if isinstance(v, list):
                    txt = "[" + ",".join(_emit_scalar(x) for x in v) + "]"
                else:
                    txt = _emit_scalar(v)
                f.write("  %s: %s%s\n" % (k, txt, "  # a comment" if i % 3 == 0 else ("" if i % 3 == 1 else " ")))
            f.write("\n")


def _run_config_file(seq, tmp_path):
    entry = GOLD["calib"][seq]
    real = os.path.join(REF, entry["file"])
    if os.path.exists(real):
        assert hashlib.sha256(open(real, "rb").read()).hexdigest() == entry["sha256"], "the fixture is stale: rerun tests/golden/make_reference_configs.py"
        return real
    p = str(tmp_path / ("iba_calib_global_%s.yml" % seq))
    _emit_run_config(entry["values"], p)
    return p


@pytest.mark.parametrize("seq", sorted(GOLD["calib"]))
def test_run_config_reader_against_the_reference_files(seq, tmp_path):
    tree = GOLD["calib"][seq]["values"]
    cfg = fmt.RunConfig(_run_config_file(seq, tmp_path))
    # every entry of the three maps, as text -> value
    for sec in ("io", "orb", "runtime"):
        for k, v in tree[sec].items():
            got = cfg.get("%s.%s" % (sec, k))
            assert got is not None, (sec, k)
            if isinstance(v, bool):
                assert got.lower() == ("true" if v else "false")
            elif isinstance(v, (int, float)):
                assert float(got) == float(v), (k, got, v)
            elif isinstance(v, list):
                assert [float(t) for t in got.strip("[]").split(",")] == [float(x) for x in v]
            else:
                assert got == str(v)
    r, io = tree["runtime"], tree["io"]
    p = cfg.params(local_stage=False)
    assert (p.max_pixel_dist, p.corr_3d_2d_threshold, p.corr_3d_3d_threshold, p.norm_max_pts, p.norm_min_pts, p.norm_radius, p.norm_reg_threshold, p.min_diff_dist, bool(p.use_plane)) == \
           (r["max_pixel_dist"], r["corr_3d_2d_threshold"], r["corr_3d_3d_threshold"], r["norm_max_pts"], r["norm_min_pts"], r["norm_radius"], r["norm_reg_threshold"], r["min_diff_dist"], r["use_plane"])
    assert list(p.err_weight) == r["err_weight"]
    base = io["BaseDir"] if io["BaseDir"].endswith("/") else io["BaseDir"] + "/"
    d = cfg.paths(local_stage=False)
    assert d["frame_id_file"] == base + io["VOIdFile"] and d["lidar_pose_file"] == base + io["LOFile"]
    assert d["pointcloud_dir"].rstrip("/") == io["PointCloudDir"].rstrip("/") and d["keyframe_dir"].rstrip("/") == tree["orb"]["KeyFrameDir"].rstrip("/") and d["map_file"] == tree["orb"]["MapFile"]
    assert (d["pointcloud_skip"], d["only_positive_x"]) == (1, 0)                       # iba_global reads the flags and ignores them (iba_global.cpp:494)
    dl = cfg.paths(local_stage=True)
    assert (dl["pointcloud_skip"], dl["only_positive_x"]) == (io["PointCloudskip"], int(io["PointCloudOnlyPositiveX"]))
    assert (d["num_best_covis"], d["min_covis_weight"]) == (r["num_best_covis"], r["min_covis_weight"])
    x0 = np.array([1.2, -1.2, 1.2, 0.05, -0.08, -0.27, 0.3])
    m = cfg.mads(x0)
    assert np.array_equal(np.array(m.lb[:]), x0 + np.array(r["lb"])) and np.array_equal(np.array(m.ub[:]), x0 + np.array(r["ub"]))
    assert list(m.init_frame) == r["init_frame"] and m.min_mesh == r["min_mesh"] and m.max_bb_eval == r["max_bbeval"]
    assert (m.he_threshold, m.valid_rate, m.seed) == (r["he_threshold"], r["valid_rate"], r["seed"]) and (m.vns_max_idle > 0) == r["use_vns"]
    assert cfg.path("init_sim3") == base + io["init_sim3"]
    cfg.close()


def test_built_in_defaults_are_sequence_00s_file():
    """abi.reference_yaml_params (what every test and bench.py runs with) and iba_default_mads_options re-type values of
    config/calib/00/iba_calib_global.yml by hand: pinned here to the file's values."""
    r = GOLD["calib"]["00"]["values"]["runtime"]
    p = abi.reference_yaml_params()
    assert (p.max_pixel_dist, p.corr_3d_2d_threshold, p.corr_3d_3d_threshold, p.norm_max_pts, p.norm_min_pts, p.norm_radius, p.norm_reg_threshold, p.min_diff_dist, bool(p.use_plane)) == \
           (r["max_pixel_dist"], r["corr_3d_2d_threshold"], r["corr_3d_3d_threshold"], r["norm_max_pts"], r["norm_min_pts"], r["norm_radius"], r["norm_reg_threshold"], r["min_diff_dist"], r["use_plane"])
    assert list(p.err_weight) == r["err_weight"]
    x0 = np.array([0.3, -0.2, 0.1, 0.0, 0.1, -0.2, 1.0])
    o = pkg.mads_options(x0)
    assert np.allclose(np.array(o.lb[:]) - x0, r["lb"], rtol=0, atol=1e-15) and np.allclose(np.array(o.ub[:]) - x0, r["ub"], rtol=0, atol=1e-15)
    assert list(o.init_frame) == r["init_frame"] and o.min_mesh == r["min_mesh"] and o.max_bb_eval == r["max_bbeval"]
    assert (o.he_threshold, o.valid_rate, o.seed) == (r["he_threshold"], r["valid_rate"], r["seed"])
    # SURVEY 8(d)'s synthetic camera is KITTI00-02.yaml's
    k = GOLD["orb"]["KITTI00-02.yaml"]["values"]
    synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth")
    prob, _ = synth.make_scene(n_frames=2, pts_per_frame=300, n_keypoints=50, seed=0)
    assert list(prob.arrays["intrinsics"][:4]) == [k["Camera.fx"], k["Camera.fy"], k["Camera.cx"], k["Camera.cy"]]
    assert k["ORBextractor.nFeatures"] == 2000


def test_missing_and_malformed_entries_fail_loudly(tmp_path):
    tree = json.loads(json.dumps(GOLD["calib"]["00"]["values"]))
    del tree["runtime"]["he_threshold"]
    p = str(tmp_path / "a.yml")
    _emit_run_config(tree, p)
    cfg = fmt.RunConfig(p)
    cfg.params(local_stage=False)                                                   # not needed for the parameters
    with pytest.raises(pkg.IbaError) as ei:
        cfg.mads(np.zeros(7))
    assert "he_threshold" in str(ei.value)
    cfg.close()
    open(p, "w").write("io:\n  BaseDir: x\n  nested:\n    deeper: 1\n")
    with pytest.raises(pkg.IbaError):
        fmt.RunConfig(p)
    open(p, "w").write("runtime:\n  lb: [1, 2\n")
    with pytest.raises(pkg.IbaError):
        fmt.RunConfig(p)
    with pytest.raises(pkg.IbaError):
        fmt.RunConfig(str(tmp_path / "missing.yml"))


def test_local_stage_reads_the_reference_s_own_key_spelling(tmp_path):
    """iba_local.cpp:372 reads io["PointCloudSkip"] (capital S); iba_global.cpp:456 io["PointCloudskip"]. A config file
    written for the reference's iba_local (the keys IBALocalParams needs, iba_local.cpp:358-377, with ITS spelling) must load for the
    local stage; the global spelling is accepted as a fallback; neither present is an error that names the stage's own key."""
    tree = json.loads(json.dumps(GOLD["calib"]["00"]["values"]))
    del tree["io"]["PointCloudskip"]
    tree["io"]["PointCloudSkip"] = 3
    tree["io"]["PointCloudOnlyPositiveX"] = True
    tree["runtime"].update({"neigh_radius": 0.55, "neigh_max_pts": 28, "robust_kernel_delta": 2.5})
    p = str(tmp_path / "iba_calib_local.yml")
    _emit_run_config(tree, p)
    cfg = fmt.RunConfig(p)
    dl = cfg.paths(local_stage=True)
    assert (dl["pointcloud_skip"], dl["only_positive_x"]) == (3, 1)
    dg = cfg.paths(local_stage=False)                      # the global stage ignores both flags whichever spelling the file has
    assert (dg["pointcloud_skip"], dg["only_positive_x"]) == (1, 0)
    lp = cfg.params(local_stage=True)
    assert (lp.neigh_radius, lp.neigh_max_pts, lp.robust_kernel_delta) == (0.55, 28, 2.5)
    cfg.close()
    del tree["io"]["PointCloudSkip"]
    _emit_run_config(tree, p)
    cfg = fmt.RunConfig(p)
    with pytest.raises(pkg.IbaError) as ei:
        cfg.paths(local_stage=True)
    assert "PointCloudSkip" in str(ei.value)
    with pytest.raises(pkg.IbaError) as ei:
        cfg.paths(local_stage=False)
    assert "PointCloudskip" in str(ei.value)
    cfg.close()


@pytest.mark.parametrize("name", sorted(GOLD["orb"]))
def test_cv_yaml_reader_on_the_orb_settings_files(name, tmp_path):
    """The cv::FileStorage reader behind iba_dataset_load (KeyFrames/NNNNNN.yml are written in this dialect) on the ORB-SLAM2
    settings files the reference ships: every numeric entry equals PyYAML's reading."""
    entry = GOLD["orb"][name]
    real = os.path.join(REF, entry["file"])
    if os.path.exists(real):
        assert hashlib.sha256(open(real, "rb").read()).hexdigest() == entry["sha256"]
        path = real
    else:
        path = str(tmp_path / name)
        with open(path, "w") as f:
            f.write("%YAML:1.0\n\n# Camera Parameters\n")
            for i, (k, v) in enumerate(entry["values"].items()):
                f.write("%s:%s%s\n" % (k, "" if i % 5 == 4 else " ", _emit_scalar(v)))
    n = 0
    for k, v in entry["values"].items():
        if isinstance(v, (int, float)) and not isinstance(v, bool):
            got = fmt.read_cv_yaml_numbers(path, k)
            assert list(got) == [float(v)], (k, got, v)
            n += 1
    assert n >= 10
    with pytest.raises(pkg.IbaError):
        fmt.read_cv_yaml_numbers(path, "No.Such.Key")
