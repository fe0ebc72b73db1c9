"""host/iba_reference_adaptors.hpp — the glue a maintainer of the reference compiles in THEIR tree — through a compiler at last:
`g++ -fsyntax-only` against declaration stubs of the Eigen / OpenCV / ORB_SLAM2 members it touches (tests/adaptor_syntax/).
Catches typos, wrong member names, overload ambiguities at the reference's call sites (its BAError parameter is NAMED iba_params
and shadows the C struct: found by this check). NOT parity evidence: nothing is linked or run."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adaptor_header_and_reference_shaped_call_sites_type_check():
    d = os.path.join(ROOT, "tests", "adaptor_syntax")
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(d, "stubs"), "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "spatial-temporal-lidar-camera-calibration_amd", "host"), os.path.join(d, "call_sites.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
