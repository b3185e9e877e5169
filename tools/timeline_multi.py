"""GPU timeline of the batched multi-robot bench from a rocprofv3 kernel trace: per kernel class the busy time, and the idle gaps
of the grid-exclusive chain (ray casts + pushes).  usage: python tools/timeline_multi.py <kernel_trace.csv> [skip_first_ms]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
t_end = rows[-1][1]
# the last 60 % of the run: steady state of the timed region
t0 = rows[0][0] + int(0.4 * (t_end - rows[0][0]))
rows = [r for r in rows if r[0] >= t0]
span = rows[-1][1] - rows[0][0]


def cls(n):
    for k in ("k_icp_batch", "k_icp", "k_raycast_batch", "k_raycast", "k_push_tables", "k_push_classify", "k_push_update", "k_push_halo"):
        if k in n:
            return k
    return n[:30]


busy = collections.defaultdict(int); cnt = collections.Counter(); queues = collections.defaultdict(set)
for s, e, n, q in rows:
    c = cls(n); busy[c] += e - s; cnt[c] += 1; queues[c].add(q)
print(f"span {span / 1e3:.0f} us")
for c in sorted(busy, key=lambda c: -busy[c]):
    print(f"  {c:22s} n {cnt[c]:5d}  mean {busy[c] / cnt[c] / 1e3:7.1f} us  busy {100.0 * busy[c] / span:5.1f} % of span   queues {sorted(queues[c])}")
# grid-exclusive chain: ray casts and push kernels; union of their intervals vs span
ex = sorted((s, e) for s, e, n, q in rows if cls(n) in ("k_raycast_batch", "k_raycast", "k_push_classify", "k_push_update", "k_push_halo"))
u = 0; cur_s, cur_e = ex[0]
gaps = []
for s, e in ex[1:]:
    if s > cur_e:
        u += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
u += cur_e - cur_s
print(f"grid-exclusive kernels cover {100.0 * u / span:.1f} % of the span; {len(gaps)} gaps, mean {sum(gaps) / max(len(gaps), 1) / 1e3:.1f} us; "
      f"gaps > 10 us: {sum(1 for g in gaps if g > 10000)} totalling {sum(g for g in gaps if g > 10000) / 1e3:.0f} us")
icp = sorted((s, e) for s, e, n, q in rows if "k_icp" in n)
ov = 0
for (s1, e1), (s2, e2) in zip(icp, icp[1:]):
    ov += max(0, min(e1, e2) - s2)
print(f"registration launches {len(icp)}, mean {sum(e - s for s, e in icp) / len(icp) / 1e3:.1f} us, consecutive overlap {100.0 * ov / span:.1f} % of span")
# a few rounds verbatim
base = rows[len(rows) // 2][0]
print("excerpt (us from an arbitrary origin):")
for s, e, n, q in rows[len(rows) // 2: len(rows) // 2 + 40]:
    print(f"  {(s - base) / 1e3:8.1f} .. {(e - base) / 1e3:8.1f}  q{q}  {cls(n)}")
