#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_bench.sh into the small files committed under profiles/:
  profiles/<tag>_kernel_stats.csv   the --kernel-trace --stats table (per kernel: calls, total, avg, min, max)
  profiles/<tag>_pmc.json           per kernel: average duration and HBM counters per launch, FETCH_SIZE /
                                    WRITE_SIZE in KB as reported plus bytes with the gfx950 corrections
usage: tools/profile_summarise.py <tag>      (reads gpurun_out/prof_<tag>_{stats,fetch,write})"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "gpurun_out", "profiles_new") if os.environ.get("GRAFT_REPO_ROOT") else os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    return g[0] if g else None


stats = one(f"prof_{tag}_stats/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
bench = os.path.join(src, f"prof_{tag}_stats.json")
if os.path.exists(bench):
    shutil.copy(bench, os.path.join(dst, f"{tag}_bench_under_rocprof.json"))


def pmc(kind):
    cc = one(f"prof_{tag}_{kind}/**/*counter_collection.csv")
    if not cc:
        return {}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc)):
        name = r["Kernel_Name"].split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[name]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": len(d["dur_ns"])} for k, d in acc.items()}


fetch, write = pmc("fetch"), pmc("write")
out = {"tag": tag, "units": {"FETCH_SIZE": "KB per launch (as reported)", "WRITE_SIZE": "KB per launch (as reported)"},
       "corrections": "MI355X_MICROARCH.md HBM section: FETCH_SIZE under-reports wide (16 B/lane) coalesced reads by 2x on "
                      "gfx950; this path reads 8 B/lane (fp64 cells), calibrated with tsd_calibrate kernel k_calib_rmw "
                      "(see calib entry: known bytes vs reported) when present", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    e = {}
    if k in fetch:
        e["avg_us_fetch_pass"] = fetch[k]["dur_ns"] / 1e3
        e["FETCH_SIZE_KB"] = fetch[k].get("FETCH_SIZE")
        e["launches"] = fetch[k]["launches"]
    if k in write:
        e["avg_us_write_pass"] = write[k]["dur_ns"] / 1e3
        e["WRITE_SIZE_KB"] = write[k].get("WRITE_SIZE")
    out["kernels"][k] = e
cal = out["kernels"].get("tsd::k_calib_rmw")
if cal:
    known_kb = 16.0 * (8 << 20) / 1024.0      # bench.py CALIB_DOUBLES: 16 B read + 16 B written per element
    cal["known_read_KB"] = cal["known_write_KB"] = known_kb
    if cal.get("FETCH_SIZE_KB"):
        cal["fetch_factor_known_over_reported"] = known_kb / cal["FETCH_SIZE_KB"]
    if cal.get("WRITE_SIZE_KB"):
        cal["write_factor_known_over_reported"] = known_kb / cal["WRITE_SIZE_KB"]
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
want = ("tsd::k_push_update", "tsd::k_calib_rmw", "tsd::k_occ_cells", "tsd::k_occ_mark", "tsd::k_push_classify", "tsd::k_push_halo")
print(json.dumps({k: v for k, v in out["kernels"].items() if k in want}, indent=1)[:3000])
