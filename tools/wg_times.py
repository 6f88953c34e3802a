#!/usr/bin/env python3
"""Summary of the lines tools/wg_times.sh prints (the last launch in the file): start and end of the sampled workgroups
of the tile kernel by kind (whole tile, band of N rows), and how many of the sampled workgroups were running in every
twentieth of the launch.  usage: python3 tools/wg_times.py lines.txt"""
import re
import sys

# every launch prints a header line first
launches = []
for l in open(sys.argv[1]):
    if l.startswith("launch "):
        launches.append([])
    m = re.match(r"wg (\d+) rows (\d+) pairs (\d+) t (\d+) (\d+)", l)
    if m and launches:
        launches[-1].append(tuple(int(v) for v in m.groups()))
wg = launches[-1]
t0 = min(r[3] for r in wg); t1 = max(r[4] for r in wg)
us = lambda t: (t - t0) / 100.0
print(f"{len(launches)} launches; last one: {len(wg)} sampled workgroups, {us(t1):.1f} us from the first start to the last end")
for kind in sorted({r[1] for r in wg}, reverse=True):
    g = [r for r in wg if r[1] == kind]
    st = sorted(us(r[3]) for r in g); en = sorted(us(r[4]) for r in g); du = sorted((r[4] - r[3]) / 100.0 for r in g)
    q = lambda a, f: a[min(len(a) - 1, int(f * len(a)))]
    print(f"rows {kind:2d}: {len(g):4d} workgroups  start {st[0]:7.1f} / {q(st, .5):7.1f} / {st[-1]:7.1f}   end {en[0]:7.1f} / {q(en, .5):7.1f} / {en[-1]:7.1f}   duration {du[0]:7.1f} / {q(du, .5):7.1f} / {du[-1]:7.1f} us (min / median / max)")
n = 20
for i in range(n):
    a, b = t0 + (t1 - t0) * i / n, t0 + (t1 - t0) * (i + 1) / n
    mid = (a + b) / 2
    print(f"  at {us(mid):7.1f} us: {sum(1 for r in wg if r[3] <= mid < r[4]):4d} of the sampled workgroups running")
