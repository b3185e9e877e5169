"""The facade's THREAD CONTRACT on CPU, under sanitizers.

tests/thread_contract.cpp drives the product's own ThreadSLAM / ThreadMapping / ThreadLocalize sources (csrc/host) against
tests/stub_tsd_hip.c -- a recording stand-in for the device ABI with canned results (NOT the oracle, and nothing the product
links) -- and checks what the reference promises: the first scan initialises synchronously on the caller's thread
(/root/reference/src/ThreadLocalize.cpp:248-276, ThreadMapping.cpp:23-41), the newest scan wins (ThreadLocalize.cpp:319-332), the
mapper is LIFO (ThreadMapping.cpp:43-76), an announced scan is used only if it is the one that comes, shutdown ends and joins
both loops with work still queued (ThreadSLAM.cpp:19-33).  Built twice: -fsanitize=thread (data races, lock order) and
-fsanitize=address,undefined (lifetime of the queued sensor copies, leaks at shutdown)."""
import math
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "ohm_tsd_slam_amd", "csrc", "host")
CASES = ["first_scan_is_synchronous", "newest_scan_wins", "mapper_is_lifo", "threaded_unfused_scan_goes_through_the_mapper",
         "announce_next_accept_and_drop", "threaded_fused_stages_the_queued_scan", "shutdown_with_work_queued", "tf_map_to_odom"]


def _build(tmp, tag, flags):
    exe = os.path.join(tmp, "thread_contract_" + tag)
    stub = os.path.join(tmp, f"stub_{tag}.o")
    common = ["-O1", "-g", "-fno-omit-frame-pointer"] + flags
    subprocess.run(["gcc", "-c", *common, "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "stub_tsd_hip.c"), "-o", stub],
                   check=True, capture_output=True, text=True)
    srcs = [os.path.join(ROOT, "tests", "thread_contract.cpp")] + [os.path.join(HOST, f) for f in
                                                                   ("obvision/obvious.cpp", "ThreadSLAM.cpp", "ThreadMapping.cpp", "ThreadLocalize.cpp")]
    r = subprocess.run(["g++", "-std=c++17", *common, "-pthread", "-DOHM_TSD_SLAM_NO_ROS", "-I" + HOST, "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "tests"), *srcs, stub, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.parametrize("tag,flags,env", [
    ("tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=0 exitcode=66 second_deadlock_stack=1"}),
    ("asan", ["-fsanitize=address,undefined"], {"ASAN_OPTIONS": "detect_leaks=1 exitcode=67", "UBSAN_OPTIONS": "print_stacktrace=1 halt_on_error=1"}),
])
def test_thread_contract_under_sanitizer(tmp_path, tag, flags, env):
    if not (shutil.which("g++") and shutil.which("gcc")):
        pytest.skip("no host compiler")
    exe = _build(str(tmp_path), tag, flags)
    r = subprocess.run([exe], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert "Sanitizer" not in out, out[-4000:]               # no report of either sanitizer
    assert r.returncode == 0, out[-4000:]
    for c in CASES:
        assert f"ok {c}" in r.stdout, out[-2000:]
    assert "thread_contract: all cases ok" in r.stdout
    _check_tf(r.stdout)


def _mat(v):
    """4 x 4 homogeneous matrix of a geometry_msgs/Transform (tx ty tz qx qy qz qw), the textbook quaternion formula"""
    tx, ty, tz, x, y, z, w = v
    n = x * x + y * y + z * z + w * w
    x, y, z, w = (c / math.sqrt(n) for c in (x, y, z, w))
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = (tx, ty, tz)
    return M


def _check_tf(stdout):
    """map -> odom as the facade published it == laser pose * T(laser <- base_footprint) * T(base_footprint <- odom)
    (/root/reference/src/ThreadLocalize.cpp:617-661), the two look-ups being the inverses of the tree's odom -> base_footprint ->
    laser edges the test fed the buffer with"""
    v = {}
    for line in stdout.splitlines():
        if line.startswith("tf_result "):
            f = line.split()
            v[f[1]] = [float(x) for x in f[2:]]
    assert set(v) == {"laser_pose", "odom_base", "base_laser", "map_odom"}, v.keys()
    expect = _mat(v["laser_pose"]) @ np.linalg.inv(_mat(v["base_laser"])) @ np.linalg.inv(_mat(v["odom_base"]))
    got = _mat(v["map_odom"])
    assert np.max(np.abs(expect - got)) <= 1e-12, (expect, got)
    assert abs(sum(c * c for c in v["map_odom"][3:]) - 1.0) <= 1e-12          # a unit quaternion leaves
    # not the laser pose itself: the correction is in
    assert np.max(np.abs(got - _mat(v["laser_pose"]))) > 0.1
