#!/usr/bin/env python3
"""Randomised parity sweep of the HIP path against the oracle (test infrastructure: uses oracle/), beyond the fixed cases of tests/:
random grid sizes / cell sizes / scenes / scan geometries, random sensor poses (not a trajectory), scans with zero / NaN / over-range
readings sprinkled in, then per case: pushes (stats + every cell), ray casts (hit masks exact, coordinates 1e-9), registrations
(pairs / iterations / state exact, T 1e-9), occupancy maps (byte-exact).  Stops at the first mismatch and prints the seed.
"hard": poses anywhere in the grid, at its edges and outside of it, any heading; registrations from up to 0.5 m / 0.2 rad away (few
pairs, dropped points, not-matchable results) and with the point-to-line estimator.
usage (GPU box): python3 tools/fuzz_parity.py [cases] [first_seed] [hard|easy] [big]"""
import math, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ohm_tsd_slam_amd import capi, synth
from oracle import pyoracle as O
from tests import helpers as H

O.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hard = len(sys.argv) > 3 and sys.argv[3] == "hard"
t_start = time.time()
tot = dict(pushes=0, raycasts=0, icps=0, occs=0, files=0)


def spoil(rng, r32):
    """zero / NaN / over-range / tiny readings at random beams (Sensor.cpp:246-272 treats each differently)"""
    r = r32.copy()
    n = len(r)
    for val in (0.0, np.nan, 45.0, 0.0005):
        k = rng.integers(0, max(2, n // 40))
        r[rng.integers(0, n, k)] = val
    if n > 40 and rng.random() < 0.3:            # a dropped sector
        a = rng.integers(0, n - 20); r[a:a + rng.integers(3, 20)] = 0.0
    return r


for case in range(n_cases):
    seed = seed0 + case
    rng = np.random.default_rng(seed)
    map_log2 = int(rng.choice([8, 9, 9, 10, 10]))
    cs = float(rng.choice([0.03, 0.05, 0.05, 0.07, 0.1]))
    scene = str(rng.choice(["room", "pillars"]))
    geo = synth.ScanGeometry.full_circle_360() if rng.random() < 0.4 else synth.ScanGeometry.utm30lx()
    if hard and rng.random() < 0.35:
        # any scanner: 5..2048 beams over 20..360 degrees starting anywhere -- fields of view that reach past +-pi (the reference's
        # backProject never names the beams beyond the cut: atan2 lives in (-pi, pi]), odd beam counts, coarse and fine resolutions
        nb = int(rng.choice([5, 17, 64, 181, 361, 541, 1000, 1440, 2048]))
        fov = math.radians(float(rng.uniform(20.0, 360.0)))
        geo = synth.ScanGeometry(nb, float(rng.uniform(-math.pi, math.pi - 0.1)), fov / max(nb - 1, 1))
    if len(sys.argv) > 4 and sys.argv[4] == "big":
        # 2^12 cells at 0.01 m: 16 384 tiles, all of them inside the 30 m range -- the classification kernel's 1 024-thread form (windows
        # of more than 12 288 tiles) and several tiles per workgroup on the full device
        map_log2, cs = 12, 0.01
    gc = synth.GridConfig(map_log2, cs)
    world = synth.World(scene, gc)
    W = gc.cells * cs
    # a device that shows only a few compute units (TSD_DEBUG_N_CUS, read when the context is created): the update kernel's workgroups
    # then take SEVERAL tiles each off the ticket queue -- on the full device the small grids of this sweep give every workgroup one tile
    if hard and rng.random() < 0.4:
        os.environ["TSD_DEBUG_N_CUS"] = str(int(rng.choice([1, 2, 4, 16])))
    else:
        os.environ.pop("TSD_DEBUG_N_CUS", None)
    og = O.Grid(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    dg = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
    content = np.full(gc.cells * gc.cells, -1, dtype=np.int8)
    tag = f"seed {seed}: 2^{map_log2} cells @ {cs} m, {scene}, {geo.beams} beams" + (f", {os.environ['TSD_DEBUG_N_CUS']} compute units" if "TSD_DEBUG_N_CUS" in os.environ else "")
    try:
        # a cluster of poses around a random point of the free space near the start (pushes must overlap for the registration to work)
        x0 = world.start[0] + rng.uniform(-0.15, 0.15) * min(W, 20.0)
        y0 = world.start[1] + rng.uniform(-0.15, 0.15) * min(W, 20.0)
        if hard and rng.random() < 0.7:
            x0 = rng.uniform(-0.1, 1.1) * W; y0 = rng.uniform(-0.1, 1.1) * W
        yaw0 = rng.uniform(-math.pi, math.pi)
        n_push = int(rng.integers(3, 9))
        for k in range(n_push):
            x = x0 + rng.uniform(-0.4, 0.4); y = y0 + rng.uniform(-0.4, 0.4); yaw = yaw0 + rng.uniform(-0.3, 0.3)
            exact_pose = False
            if hard and rng.random() < 0.3:
                exact_pose = True
                # exact poses: the sensor on a cell centre / corner / tile corner, the heading a multiple of the angular resolution or of
                # pi / 2 -- cell centres then project EXACTLY onto beam boundaries and onto the +-pi cut of atan2
                q = float(rng.choice([cs, 0.5 * cs, 32 * cs]))
                x = round(x / q) * q + (0.5 * cs if rng.random() < 0.5 else 0.0); y = round(y / q) * q + (0.5 * cs if rng.random() < 0.5 else 0.0)
                # (not HALF a resolution step: the cells along the eight principal directions would then project onto a beam boundary to
                # within 1e-13, where round((atan2 - phiMin) / res) is decided by the last bit of libm's atan2 -- glibc's on the oracle's
                # side, the device library's here: 32 cells of 400 000 differ in such a push)
                qa = float(rng.choice([geo.angle_increment, math.pi / 2]))
                yaw = round(yaw / qa) * qa
            pose = synth.pose_matrix(x, y, yaw)
            r32 = world.scan(x, y, yaw, geo)
            if rng.random() < 0.6:
                r32 = spoil(rng, r32)
            data, mask = O.ingest_f32(r32, H.MAX_RANGE, geo.angle_increment)
            so = og.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
            sd = dg.push(pose, data, mask, geo.angle_increment, geo.angle_min, H.MAX_RANGE, H.MIN_RANGE, H.LOW_REFL)
            if exact_pose:
                # The cell whose centre is the sensor position to within rounding has no direction: PoseInv * centre is ~1e-15 of noise
                # (the oracle's LU inverse and the library's differ in the last bit), and atan2 of noise names an arbitrary beam.  Any
                # OTHER difference fails; that cell (up to four copies: it can be a halo cell too) is taken over from the oracle.
                od, gd = og.dump(), dg.download_tiles()
                ot_, gt_, ow_, gw_ = od[2].reshape(-1, 33, 33), gd[2].reshape(-1, 33, 33), od[3].reshape(-1, 33, 33), gd[3].reshape(-1, 33, 33)
                assert np.array_equal(od[0], gd[0]) and np.array_equal(od[1], gd[1]), f"push {k} (exact pose): tile state differs"
                bad = np.argwhere(~((ot_ == gt_) | (np.isnan(ot_) & np.isnan(gt_))) | (ow_ != gw_))
                PXt = gc.cells // 32
                for p_, iy_, ix_ in bad:
                    ccx = ((p_ % PXt) * 32 + ix_ + 0.5) * cs; ccy = ((p_ // PXt) * 32 + iy_ + 0.5) * cs
                    assert math.hypot(ccx - x, ccy - y) < 1e-9, f"push {k} (exact pose): cell ({ix_}, {iy_}) of tile {p_} differs, {math.hypot(ccx - x, ccy - y)} m from the sensor"
                if len(bad):
                    dg.upload_tiles(*od); tot["sensor_cell"] = tot.get("sensor_cell", 0) + 1
                else:
                    assert so == sd, f"push {k}: stats differ\n oracle {so}\n hip    {sd}"
            else:
                assert so == sd, f"push {k}: stats differ\n oracle {so}\n hip    {sd}"
                H.assert_grids_equal(og.dump(), dg.download_tiles(), 0.0)
            tot["pushes"] += 1
            if hard and rng.random() < 0.25:
                # TsdGrid::freeFootprint (TsdGrid.cpp:609-638) between two pushes: rectangles inside, across the edge of and outside the
                # grid (the reference refuses those: both sides must agree on that too); the next push refreshes the halos it dirtied
                fc = [x + rng.uniform(-1.0, 1.0), y + rng.uniform(-1.0, 1.0)]
                fw, fh_ = float(rng.uniform(0.05, 2.5)), float(rng.uniform(0.05, 2.5))
                r_o, r_h = og.free_footprint(fc, fw, fh_), dg.free_footprint(fc, fw, fh_)
                assert bool(r_o) == bool(r_h), f"freeFootprint accepted by one side only ({r_o} / {r_h})"
                od_, gd_ = og.dump(), dg.download_tiles()
                assert np.array_equal(od_[0], gd_[0]) and np.array_equal(od_[1], gd_[1]), "freeFootprint: tile state differs"
                sel_ = od_[0].astype(bool)
                # (interior cells: the halos a footprint dirties are refreshed by the NEXT push on both sides, TsdGrid.cpp:372-427)
                a_, b_ = od_[2].reshape(-1, 33, 33)[sel_][:, :32, :32], gd_[2].reshape(-1, 33, 33)[sel_][:, :32, :32]
                assert np.array_equal(np.isnan(a_), np.isnan(b_)) and np.array_equal(a_[~np.isnan(a_)], b_[~np.isnan(b_)]), "freeFootprint: cells differ"
                tot["footprints"] = tot.get("footprints", 0) + 1
            if rng.random() < 0.15:
                wimg, himg = (gc.cells, gc.cells) if rng.random() < 0.3 else (int(rng.integers(8, 700)), int(rng.integers(8, 500)))
                io_, ih_ = og.color_image(wimg, himg), dg.color_image(wimg, himg)
                assert np.array_equal(io_, ih_), f"colour image {wimg} x {himg} differs"
                tot["images"] = tot.get("images", 0) + 1
            if rng.random() < 0.3:
                inflate = bool(rng.random() < 0.5)
                oo, no = og.occupancy(content, inflate, 2)
                od, nd = dg.occupancy(inflate, 2)
                assert no == nd, f"occupancy: sign changes {no} / {nd}"
                assert np.array_equal(oo.reshape(gc.cells, gc.cells), od), "occupancy maps differ"
                tot["occs"] += 1
        if map_log2 <= 9 and rng.random() < 0.3:
            # TsdGrid::storeGrid / TsdGrid(file) (TsdGrid.cpp:548-607, :25-110): the text files byte for byte, the reloaded grids cell for cell
            with tempfile.TemporaryDirectory() as td:
                fo, fh = os.path.join(td, "oracle.grid"), os.path.join(td, "hip.grid")
                assert og.store_text(fo)
                dg.store_text(fh)
                assert open(fo, "rb").read() == open(fh, "rb").read(), "text grid files differ"
                og2 = O.Grid.load_text(fo, gc.cell_size)
                dg2 = capi.TsdGridDevice(gc.map_size_log2, gc.cell_size, gc.max_trunc)
                dg2.load_text(fh)
                H.assert_grids_equal(og2.dump(), dg2.download_tiles(), 0.0)
                dg2.close()
            tot["files"] += 1
        for k in range(3):
            x = x0 + rng.uniform(-0.5, 0.5); y = y0 + rng.uniform(-0.5, 0.5); yaw = yaw0 + rng.uniform(-0.4, 0.4)
            if hard and rng.random() < 0.3:
                # exact poses for the ray cast too: rays along the axes and diagonals, samples ON cell and tile boundaries, the repeated
                # `position += ray` of the reference landing on exact ties
                q = float(rng.choice([cs, 0.5 * cs, 32 * cs]))
                x = round(x / q) * q + (0.5 * cs if rng.random() < 0.5 else 0.0); y = round(y / q) * q + (0.5 * cs if rng.random() < 0.5 else 0.0)
                qa = float(rng.choice([geo.angle_increment, 0.5 * geo.angle_increment, math.pi / 2, math.pi / 4]))
                yaw = round(yaw / qa) * qa
            pose = synth.pose_matrix(x, y, yaw)
            rl, rw = H.world_rays(O, geo, pose, gc.cell_size)
            co, no_, mo, cnt_o = og.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
            cd, nd_, md, cnt_d = dg.raycast(pose, rw, H.MIN_RANGE, H.MAX_RANGE)
            assert np.array_equal(mo, md), f"ray cast {k}: hit masks differ at beams {np.nonzero(mo != md)[0][:10]}"
            sel = np.repeat(mo.astype(bool), 2)
            if sel.any():
                assert np.max(np.abs(co[sel] - cd[sel])) <= 1e-9 and np.max(np.abs(no_[sel] - nd_[sel])) <= 1e-9, "ray cast coordinates"
            tot["raycasts"] += 1
            # registration of a scan taken from a displaced pose against this ray cast's model
            dxy = rng.uniform(-0.08, 0.08, 2); dyaw = rng.uniform(-0.03, 0.03)
            if hard and rng.random() < 0.5:
                dxy = rng.uniform(-0.5, 0.5, 2); dyaw = rng.uniform(-0.2, 0.2)
            r32 = world.scan(x + dxy[0], y + dxy[1], yaw + dyaw, geo)
            if rng.random() < 0.5:
                r32 = spoil(rng, r32)
            data, mask = O.ingest_f32(r32, H.MAX_RANGE, geo.angle_increment)
            scn, ms, _ = O.scene_from_scan(rl, data, mask)
            M = co.reshape(-1, 2)[mo.astype(bool)]
            S = scn.reshape(-1, 2)[ms.astype(bool)]
            if len(M) < 3 or len(S) < 3:
                continue
            iters = int(rng.choice([30, 30, 25, 11, 5]))
            bounds = (0.0, og.max_x, 0.0, og.max_x)
            ro = O.icp(M, S, pose, iters, 0.4, 0.02, bounds, nn_mode=0)
            rd = dg.icp(M, S, pose, dg.icp_params(iters, 0.4, 0.02))
            ptl = False
            if hard and rng.random() < 0.3 and max(len(M), len(S)) <= 1500:      # (beyond: the estimator's normals do not fit one CU's LDS beside 2 048 points -- refused with an error)
                ptl = True
                # PointToLine2DEstimator on the same pairs machinery (the ray cast's normals)
                N = no_.reshape(-1, 2)[mo.astype(bool)]
                ro = O.icp(M, S, pose, iters, 0.4, 0.02, bounds, nn_mode=0, model_normals_xy=N)
                rd = dg.icp(M, S, pose, dg.icp_params(iters, 0.4, 0.02, estimator=1), model_normals_xy=N)
            assert (ro["pairs"], ro["iterations"], ro["state"]) == (rd.pairs, rd.iterations, rd.state), \
                f"registration {k}: oracle {(ro['pairs'], ro['iterations'], ro['state'])} hip {(rd.pairs, rd.iterations, rd.state)}"
            d, a = H.pose_delta(ro["T"], rd.T)
            tol_T = 1e-6 if ptl else 1e-9      # (point-to-line: the device library's cos / sin of the LU solution against libm's, every step)
            assert d <= tol_T and a <= tol_T, (f"registration {k}: |dT| {d} m {a} rad; estimator {'point-to-line' if ptl else 'closed form'}, pairs {rd.pairs} of "
                                             f"{len(S)} scene / {len(M)} model points, iterations {rd.iterations}, state {rd.state}, rms {rd.rms} / {ro['rms']}, offset {dxy} {dyaw}")
            tot["icps"] += 1
    except AssertionError as e:
        print("MISMATCH", tag, "--", e)
        sys.exit(1)
    finally:
        dg.close() if hasattr(dg, "close") else None
    if case % 10 == 9:
        print(f"{case + 1} cases ok ({tag}); {tot}; {time.time() - t_start:.0f} s", flush=True)
print(f"all {n_cases} cases ok from seed {seed0}: {tot}; {time.time() - t_start:.0f} s")
