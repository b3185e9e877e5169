#!/usr/bin/env python3
"""One line per variant of a tools/icp_ab.sh log: tools/ab_summary.py <log>"""
import re, sys, collections
d = collections.OrderedDict(); cur = None
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith('['): cur = l; d.setdefault(cur, []); continue
    m = re.match(r'input (\d+): (\d+) distinct.*?([\d.]+) us, pairs \[(\d+)\] oracle (\d+)', l)
    if m: d[cur].append(('in' + m.group(1), float(m.group(3)), m.group(2) == '1' and m.group(4) == m.group(5)))
    m = re.match(r'slam: (\d+) scans/s \| icp ([\d.]+) us', l)
    if m: d[cur].append(('slam', float(m.group(2)), int(m.group(1))))
    if 'failed to build' in l: print(l)
for k, v in d.items():
    print(k, ' '.join(f"{a}:{b}{'' if c is True else ('!' if c is False else '/' + str(c))}" for a, b, c in v))
