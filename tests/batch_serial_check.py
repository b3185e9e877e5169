"""Body of tests/test_gpu_batch.py::test_batched_path_never_delivers_stale_registrations, run as `python -m tests.batch_serial_check`
in a CHILD process whose environment decides how HIP maps streams onto hardware queues (GPU_MAX_HW_QUEUES=1: every stream on
one in-order queue) and how the library hands work over between the streams of a batch.

Two batch slots, pushes enqueued ahead of the results -- the dispatcher's pattern -- at cfg 1 against the oracle.  Every round
ends in exactly one of two ways: the results equal the oracle's (counts / gates exact, pose within 1e-4), or a call returned a
non-zero code (capi.TsdError).  Results that differ while every call returned TSD_OK are exit code 9.  After the first error the
two sides no longer see the same scans (the reference's contract: "failures are logged and the scan is skipped"), so the
comparison stops there; the script then checks that the slot still WORKS (later rounds return results or errors, never hang)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    from ohm_tsd_slam_amd import capi
    from oracle import pyoracle as O
    from tests import helpers as H
    from tests.test_gpu_batch import _setup, _compare
    O.build()
    n_robots, n_scans = 4, 8
    gc, geo, kw, og, dg, robots, scans, sensors, params, gates = _setup(O, "cfg1", n_robots, n_scans)
    slots = [capi.TsdBatch(dg, 2), capi.TsdBatch(dg, 2)]
    groups = [[0, 1], [2, 3]]
    bounds = (dg.min_x, dg.max_x, dg.min_y, dg.max_y)
    errors, compared, first_error_round = 0, 0, None
    t0 = time.time()
    for k in range(1, n_scans):
        ing = [rb.ingest(sc[k]) for rb, sc in zip(robots, scans)]
        in_sync = errors == 0
        if in_sync:
            ros = [rb.localise(og, d_, m_, bounds) for rb, (d_, m_, _) in zip(robots, ing)]
            for rb in robots:
                rb.apply_push(og)
        begun = []
        for slot, grp in zip(slots, groups):
            try:
                slot.begin([sensors[i] for i in grp], [ing[i][0] for i in grp], [ing[i][1] for i in grp], [ing[i][2] for i in grp], params, gates)
                begun.append((slot, grp))
            except capi.TsdError as e:
                errors += 1
                first_error_round = first_error_round or k
                print(f"round {k}: begin -> error: {e}")
        for slot, _ in begun:
            try:
                slot.push()
            except capi.TsdError as e:
                errors += 1
                first_error_round = first_error_round or k
                print(f"round {k}: push -> error: {e}")
        for slot, grp in begun:
            try:
                res = slot.results()
            except capi.TsdError as e:
                errors += 1
                first_error_round = first_error_round or k
                print(f"round {k}: results -> error: {e}")
                continue
            if in_sync and errors == 0:
                for i, sr in zip(grp, res):
                    try:
                        _compare(k, i, ros[i], sr)
                    except AssertionError as e:
                        print(f"WRONG RESULTS WITH rc 0: {e}")
                        sys.exit(9)
                    compared += 1
    if errors == 0:
        H.assert_grids_equal(og.dump(), dg.download_tiles(), 1e-5)
    for slot in slots:
        slot.close()
    for s in sensors:
        s.close()
    print(f"batch_serial_check ok: {compared} results equal to the oracle's, {errors} calls returned an error"
          + (f" (first in round {first_error_round})" if errors else "") + f", {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main()
