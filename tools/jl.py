"""one-line digest of bench.py's JSON line (stdin)"""
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        j = json.loads(l)
        c = j["config"]
        print(c.get("robots"), "robots", round(j["value"]), j["unit"], "err", round(j.get("tracking_error_m") or 0, 3), "scans/batch", c.get("scans_per_batch"),
              {k: round(1e3 * v, 1) for k, v in (j.get("stages_ms") or {}).items() if v is not None})
    elif "Error" in l or "error" in l:
        print(l.strip())
