#!/usr/bin/env python3
"""Where a recorded frame's GPU time goes: from a rocprofv3 kernel trace of tools/bench_capture.py, the average duration
of every kernel of the render stream per frame and the idle time between consecutive kernels of that stream, split by
whether the compressor's kernel was running.  usage: python3 tools/trace_gaps.py <rocprof output dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
lz = [(r['s'], r['e']) for r in rows if 'lz4f_compress' in r['Kernel_Name']]
render = sorted((r for r in rows if 'lz4f' not in r['Kernel_Name']), key=lambda r: r['s'])
print('queues:', collections.Counter(r.get('Queue_Id', '?') for r in rows))


def during_lz(t):
    return any(a <= t <= b for a, b in lz)


for label, want in (('beside the compressor', True), ('alone', False)):
    dur = collections.defaultdict(list); gap = collections.defaultdict(list)
    frames = 0; span0 = None; spans = []
    prev = None
    for r in render:
        if during_lz(r['s']) != want:
            prev = None; continue
        name = r['Kernel_Name'][:28]
        dur[name].append(r['e'] - r['s'])
        if prev is not None:
            gap[prev['Kernel_Name'][:20] + ' -> ' + name[:20]].append(max(0, r['s'] - prev['e']))
        if name.startswith('rh_tile'):
            frames += 1
        prev = r
    if not frames:
        continue
    print(f'--- {label}: {frames} frames')
    tot = 0
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f'  {k:30s} n/frame {len(v)/frames:5.2f}  avg {sum(v)/len(v)/1e3:8.1f} us  per frame {sum(v)/frames/1e3:8.1f} us')
        tot += sum(v) / frames
    gt = 0
    for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print(f'  gap {k:44s} avg {sum(v)/len(v)/1e3:8.1f} us  per frame {sum(v)/frames/1e3:8.1f} us')
    gt = sum(sum(v) for v in gap.values()) / frames
    print(f'  busy per frame {tot/1e3:.1f} us, idle between kernels per frame {gt/1e3:.1f} us')
