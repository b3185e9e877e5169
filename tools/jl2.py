import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        j = json.loads(l); sp = j.get("ms_icp_iterate_spread") or {}
        print(round(j["value"]), "scans/s icp mean %.1f min %.1f max %.1f std %.1f us" % (1e3*j["ms_icp_iterate"], 1e3*sp.get("min",0), 1e3*sp.get("max",0), 1e3*sp.get("std",0)))
