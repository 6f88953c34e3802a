"""Prints, from a rocprofv3 kernel trace of bench.py, the kernels of a few consecutive frames with their start and end
times (ms, relative to the first of them) and the queue each ran on.  usage: trace_frames.py <trace dir> [first tile launch] [frames]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 20
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
tiles = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('rh_tile')]
a, b = tiles[first], tiles[first + frames]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[max(0, a - 6):b + 1]:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - t0) / 1e6
    print(f"{s:9.3f} {e:9.3f} {e - s:7.3f}  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:40]}")
