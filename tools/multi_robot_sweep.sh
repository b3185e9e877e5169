#!/bin/bash
# bench.py --robots R for R = 1, 2, 4, 8 (one shared 4096^2 grid), batched dispatcher (default) and the split scan on one
# stream per robot (TSD_NO_BATCH=1); writes gpurun_out/multi_robot_one_grid.json (copy to profiles/<tag>_multi_robot_one_grid.json)
# usage (GPU box): tools/multi_robot_sweep.sh [steps]
STEPS=${1:-200}
mkdir -p gpurun_out
OUT=gpurun_out/multi_robot_rows.jsonl
: > $OUT
for r in 1 2 4 8; do
  python bench.py --robots $r --steps $STEPS --warmup 10 --no-cpu-baseline | grep '^{' | sed 's/^{/{"path": "batched dispatcher (tsd_batch_*), two slots", /' >> $OUT
done
for r in 2 4 8; do
  TSD_NO_BATCH=1 python bench.py --robots $r --steps $STEPS --warmup 10 --no-cpu-baseline | grep '^{' | sed 's/^{/{"path": "split scan, one stream per robot (TSD_NO_BATCH=1)", /' >> $OUT
done
python - <<'PY'
import json
rows = []
for l in open("gpurun_out/multi_robot_rows.jsonl"):
    j = json.loads(l)
    rows.append({"path": j["path"], "robots": j["config"]["robots"], "scans_per_s": j["value"], "ms_per_round": j["ms_per_step"],
                 "scans_per_batch": j["config"].get("scans_per_batch"), "tracking_error_m": j["tracking_error_m"], "stages_ms": j["stages_ms"]})
base = next(r["scans_per_s"] for r in rows if r["robots"] == 1)
for r in rows:
    r["speedup_vs_one_robot"] = r["scans_per_s"] / base
json.dump({"what": "bench.py --robots R: R robots on ONE 4096^2 grid in one process (the reference's multi-robot mode), scans replayed by "
                   "one native publisher thread per robot (tsd_node_play)", "rows": rows}, open("gpurun_out/multi_robot_one_grid.json", "w"), indent=1)
for r in rows:
    print(r["path"][:20], r["robots"], round(r["scans_per_s"]), round(r["speedup_vs_one_robot"], 2), r["scans_per_batch"])
PY
