"""The facade's THREAD CONTRACT on CPU, under sanitizers.

tests/thread_contract.cpp drives the product's own ThreadSLAM / ThreadMapping / ThreadLocalize sources (csrc/host) against
tests/stub_tsd_hip.c -- a recording stand-in for the device ABI with canned results (NOT the oracle, and nothing the product
links) -- and checks what the reference promises: the first scan initialises synchronously on the caller's thread
(/root/reference/src/ThreadLocalize.cpp:248-276, ThreadMapping.cpp:23-41), the newest scan wins (ThreadLocalize.cpp:319-332), the
mapper is LIFO (ThreadMapping.cpp:43-76), an announced scan is used only if it is the one that comes, shutdown ends and joins
both loops with work still queued (ThreadSLAM.cpp:19-33).  Built twice: -fsanitize=thread (data races, lock order) and
-fsanitize=address,undefined (lifetime of the queued sensor copies, leaks at shutdown)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "ohm_tsd_slam_amd", "csrc", "host")
CASES = ["first_scan_is_synchronous", "newest_scan_wins", "mapper_is_lifo", "threaded_unfused_scan_goes_through_the_mapper",
         "announce_next_accept_and_drop", "threaded_fused_stages_the_queued_scan", "shutdown_with_work_queued"]


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
