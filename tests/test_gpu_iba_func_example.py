"""examples/iba_func.cpp — the reference's batch evaluator (iba_func.cpp:23-38, 454-471) written against the C-ABI only: a
dataset directory in the reference's formats + a text file of candidate 7-vectors in, one line "f1 f2 C valid_rate" per
candidate out, in the reference's stream format. The lines must be byte-identical to the ORACLE's numbers printed the same way."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd")
fmt = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.formats")
synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth")
abi = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.abi")
from oracle import binding as ob  # noqa: E402
from oracle import formats as ofmt  # noqa: E402

pytestmark = pytest.mark.gpu


def test_iba_func_lines_match_the_oracle(tmp_path):
    exe = os.path.join(ROOT, "examples", "iba_func")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    prob, meta = synth.make_scene(n_frames=6, pts_per_frame=3000, n_keypoints=600, seed=23, new_mappoints=120, scan_kp=150)
    paths = ofmt.write_dataset(str(tmp_path), prob, meta, keypoint_layout="nested")
    packed, _, _ = fmt.load_dataset(**paths, num_best_covis=3)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(1), n=69)])   # 70 candidates: two launches
    lst = tmp_path / "sim3_list.txt"
    with open(lst, "w") as f:
        for x in xs:
            f.write(" ".join(repr(float(v)) for v in x) + "\n")   # trailing newline: the reference would append a junk record here
    out = tmp_path / "res.txt"
    precision = 8
    subprocess.check_call([exe, paths["frame_id_file"], paths["lidar_pose_file"], paths["pointcloud_dir"], paths["keyframe_dir"], paths["map_file"],
                           str(lst), str(out), str(precision)], stdout=subprocess.DEVNULL)
    got = open(out).read()
    assert not got.endswith("\n")
    p = abi.reference_yaml_params()
    ref = ob.Oracle(packed).eval_cost(p, xs, nthreads=8)
    want = "\n".join("%.*g %.*g %.*g %.*g" % (precision, r.f1, precision, r.f2, precision, r.C, precision, r.valid_cnt_3d_2d / r.cnt_3d_2d) for r in ref)
    assert got.split("\n") == want.split("\n")


def test_iba_func_takes_the_references_own_config_file(tmp_path):
    """The reference's call is `iba_func <config.yml>` (iba_func.cpp:356-406): the same run from a yaml-cpp style config file — the
    io / orb / runtime maps of config/calib/00/iba_calib_global.yml plus iba_func's res_file / precision — gives the same bytes as
    the run with explicit arguments (VERDICT r3 missing #4: no hand transcription of the YAML)."""
    exe = os.path.join(ROOT, "examples", "iba_func")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    prob, meta = synth.make_scene(n_frames=5, pts_per_frame=2500, n_keypoints=500, seed=29, new_mappoints=100, scan_kp=120)
    paths = ofmt.write_dataset(str(tmp_path), prob, meta)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(2), n=9)])
    lst = tmp_path / "sim3_list.txt"
    with open(lst, "w") as f:
        for x in xs:
            f.write(" ".join(repr(float(v)) for v in x) + "\n")
    out1, out2 = tmp_path / "res_args.txt", tmp_path / "res_cfg.txt"
    subprocess.check_call([exe, paths["frame_id_file"], paths["lidar_pose_file"], paths["pointcloud_dir"], paths["keyframe_dir"], paths["map_file"], str(lst), str(out1), "9"], stdout=subprocess.DEVNULL)
    base = str(tmp_path)
    rel = lambda p: os.path.relpath(p, base)
    cfg = tmp_path / "iba_func.yml"
    cfg.write_text("""%%YAML:1.0
---
io:
  BaseDir: %s
  VOFile: slam_res/Twc.txt
  LOFile: %s
  VOIdFile: %s
  init_sim3: %s
  res_file: %s
  precision: 9
  PointCloudDir: %s
  PointCloudskip: 1
  PointCloudOnlyPositiveX: true 
orb:
  Vocabulary: ../data/Vocabulary/ORBvoc.txt
  Config: ../config/orb_ori/KITTI00-02.yaml
  KeyFrameDir: %s
  MapFile: %s

runtime:
  max_pixel_dist: 1.5
  num_best_covis: 3 # set negative to use min_covis_weight
  min_covis_weight: 100
  kdtree2d_max_leaf_size: 10
  kdtree3d_max_leaf_size: 30
  corr_3d_2d_threshold: 40
  corr_3d_3d_threshold: 10
  norm_max_pts: 30
  norm_min_pts: 5
  norm_radius: 0.6
  norm_reg_threshold: 0.02  # 0.02
  min_diff_dist: 0.2 # 0.2
  err_weight: [1.0, 1.0]
  verborse: true
  use_plane: true  # set to false for the ablation study""" % (base, rel(paths["lidar_pose_file"]), rel(paths["frame_id_file"]), str(lst), str(out2), paths["pointcloud_dir"], paths["keyframe_dir"], paths["map_file"]))
    subprocess.check_call([exe, "--config", str(cfg)], stdout=subprocess.DEVNULL)
    assert open(out1).read() == open(out2).read() and len(open(out2).read().split("\n")) == 10
