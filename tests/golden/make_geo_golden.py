"""Generates tests/golden/geo_corr_nanoflann_v150.npz: GeoCalib.h:18-33 computeCorrespondence run by the REFERENCE's own nanoflann (compiled where it
lies by oracle/Makefile into oracle/_ref: KDTreeSingleIndexAdaptor, L2_Simple, max_leaf 15, knnSearch(1), sq_dist <= maxDistance) on seeded clouds.
Inputs and the reference's outputs only; nothing of the reference is copied. Run in the build container:  python tests/golden/make_geo_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    """name -> (src [n, 3], tgt [m, 3] float32-valued doubles: a scan as loaded, max_distance)"""
    rng = np.random.default_rng(20261003)
    out = {}
    # a scan-like target (float32 as a KITTI .bin holds it), sources = its points moved by a small rigid motion + noise: most sources have a
    # neighbour within 0.05 (squared, as the reference compares), some do not
    tgt = (rng.normal(size=(6000, 3)) * [20, 8, 1.5]).astype(np.float32).astype(np.float64)
    th = 0.002
    Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    src = tgt[rng.choice(len(tgt), 1500, replace=False)] @ Rz.T + [0.03, -0.02, 0.01] + rng.normal(0, 0.05, (1500, 3))
    src = np.vstack([src, rng.uniform(-60, 60, (100, 3))])            # far queries: no pair
    out["scan"] = (src, tgt, 0.05)
    out["scan_wide"] = (src, tgt, 1.0)                                  # the same with a gate of 1 (metre squared): every near query kept
    out["scan_exact"] = (tgt[:200].copy(), tgt, 0.0)                    # a source ON a target point: sq_dist = 0 <= 0 keeps it (<=, not <)
    small = (rng.normal(size=(9, 3)) * 3).astype(np.float32).astype(np.float64)
    out["tiny"] = (rng.normal(size=(40, 3)) * 3, small, 4.0)            # fewer target points than one leaf
    out["empty_src"] = (np.zeros((0, 3)), tgt[:50], 0.05)
    return out


if __name__ == "__main__":
    from oracle import binding as ob
    assert ob.ref_lib() is not None, "oracle/_ref not built (needs /root/reference)"
    z = {}
    for name, (src, tgt, md) in cases().items():
        s, t = ob.geo_correspondences("ref", src, tgt, md)
        z[name + "_src"] = src; z[name + "_tgt"] = tgt; z[name + "_max_distance"] = np.float64(md)
        z[name + "_pairs_src"] = s; z[name + "_pairs_tgt"] = t
        print(name, len(src), len(tgt), md, "->", len(s), "pairs")
    np.savez_compressed(os.path.join(HERE, "geo_corr_nanoflann_v150.npz"), **z)
