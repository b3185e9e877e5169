"""A short run of the randomised parity sweep (tools/fuzz_parity.py: random grids, scenes, scan geometries, poses and spoiled readings;
pushes cell-for-cell, ray casts, registrations and occupancy maps against the oracle).  The long runs are profiles/r4_fuzz_parity.txt."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_randomised_parity_sweep_short():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "40", "777", "hard"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "all 40 cases ok" in p.stdout


@pytest.mark.gpu
def test_randomised_facade_closed_loop_short():
    """tools/fuzz_slam.py: the C++ facade's fused scan path against the oracle's SLAM loop on random trajectories with spoiled readings
    and look-ahead announcements that are kept, replaced by a decoy or absent; poses 1e-9 per scan (the oracle gets the float32-rounded
    scan angles a LaserScan carries), gates and pair counts exact, the final grid 1e-9."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_slam.py"), "30", "4242", "mode3"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "all 30 cases ok" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("tool,args", [("fuzz_slam.py", ["20", "5151"]), ("fuzz_async.py", ["15", "6161"])])
def test_scan_through_pinned_memory_when_the_bar_is_not_mapped(tool, args):
    """The fall-back of the scan's way to the device (TSD_SCAN_PINNED=1: pinned host buffer read by the registration over the host link,
    device copy on the side stream) instead of the host's stores into device memory through the PCIe BAR -- the same closed loops,
    look-ahead kept / replaced / absent, asynchronous mapping with a lagging push stream."""
    env = dict(os.environ, TSD_SCAN_PINNED="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "all %s cases ok" % args[0] in p.stdout


@pytest.mark.gpu
def test_scan_through_the_bar_with_the_device_side_cross_check():
    """TSD_SCAN_BAR_VERIFY=1 (debug): every scan the host stores into device memory through the PCIe BAR is also written to the pinned
    buffer, and ahead of its registration a kernel compares the two copies as the DEVICE sees them.  The closed loops of the sweep
    run clean with it (no scan ever differs on this platform); with the pinned copy spoiled on purpose (=2) the very first fused scan
    is refused with an error instead of producing a pose."""
    env = dict(os.environ, TSD_SCAN_BAR_VERIFY="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_slam.py"), "12", "7171"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "all 12 cases ok" in p.stdout
    code = (
        "import numpy as np\n"
        "from ohm_tsd_slam_amd import capi, synth\n"
        "from oracle import pyoracle as O\n"
        "from tests.slam_driver import HipSlamFused, slam_kwargs\n"
        "gc = synth.GridConfig(9, 0.05); geo = synth.ScanGeometry.full_circle_360(); world = synth.World('room', gc)\n"
        "scans = synth.scans_for(world, geo, synth.trajectory(world, 3))\n"
        "sh = HipSlamFused(O, **slam_kwargs(gc, geo))\n"
        "sh.process_scan(scans[0])\n"
        "print('PATH', sh.grid.lib.tsd_debug_sensor_scan_path(sh.sensor.h))\n"
        "try:\n"
        "    sh.process_scan(scans[1])\n"
        "    print('NO ERROR')\n"
        "except capi.TsdError as e:\n"
        "    print('REFUSED:', e)\n")
    for mode, want in (("2", "REFUSED"), ("1", "NO ERROR")):
        q = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, TSD_SCAN_BAR_VERIFY=mode), capture_output=True, text=True, timeout=300)
        assert q.returncode == 0, (mode, q.stdout[-1500:], q.stderr[-1500:])
        if "PATH 0" in q.stdout:
            pytest.skip("this box does not map the device's memory into the host's address space: scans go through pinned memory")
        assert "PATH 2" in q.stdout and want in q.stdout, (mode, q.stdout[-1500:], q.stderr[-1500:])
        if mode == "2":
            assert "PCIe BAR" in q.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 3])
def test_in_launch_handoffs_under_load_change_nothing(mode):
    """tools/handoff_stress.py: the fused closed loop with the push's halo pass and (mode 3) the pre-registration's arg-max as kernels
    of their own on an idle chip, against the same loop with both passes riding inside the launches behind them (the default) while a
    second context pushes a 16384^2 grid and extracts its map in a loop on another thread: every pose and the grid's digest (halo
    cells included) must be the same, bit for bit."""
    tool = os.path.join(ROOT, "tools", "handoff_stress.py")
    a = subprocess.run([sys.executable, tool, "120", str(mode)], cwd=ROOT, env=dict(os.environ, TSD_HALO_KERNEL="1", TSD_PDF_ARGMAX_KERNEL="1"),
                       capture_output=True, text=True, timeout=600)
    b = subprocess.run([sys.executable, tool, "120", str(mode), "--load"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert a.returncode == 0 and b.returncode == 0, a.stderr[-1500:] + b.stderr[-1500:]
    la = [l for l in a.stdout.splitlines() if l.startswith("poses ")]
    lb = [l for l in b.stdout.splitlines() if l.startswith("poses ")]
    assert len(la) == 1 and la == lb, (a.stdout[-600:], b.stdout[-600:])
    assert "load yes" in b.stdout and "load no" in a.stdout


@pytest.mark.gpu
def test_randomised_async_mapping_short():
    """tools/fuzz_async.py: asynchronous mapping through the staged scan with random staging (kept / replaced / absent) and a push stream
    held back by up to 3 ms per push, against the one-push-behind order on the oracle's primitives."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_async.py"), "30", "31337"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "all 30 cases ok" in p.stdout


@pytest.mark.gpu
def test_randomised_batched_robots_short():
    """tools/fuzz_batch.py: 2-6 robots on one grid through tsd_batch_* -- every round split at random into one or two slots begun
    together -- against the same order on the oracle's primitives."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_batch.py"), "16", "2024", "mode3"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "all 16 cases ok" in p.stdout


@pytest.mark.gpu
def test_randomised_icp_point_sets_short():
    """tools/fuzz_icp.py: tsd_icp on synthetic point sets -- lattices (exact multi-way ties), circles, polylines, clouds, duplicates,
    outliers, non-finite points, 3..2048 model points in random order."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_icp.py"), "600", "1"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "all 600 cases ok" in p.stdout
