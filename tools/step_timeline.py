#!/usr/bin/env python3
"""Launch-by-launch timeline of the LAST stage-B batch from a rocprofv3 --kernel-trace csv (tools/perf_register.py under the profiler):
start offset, duration, gap to the previous launch, grid size, kernel.  usage: step_timeline.py <dir> [first_kernel_substring]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
key = sys.argv[2] if len(sys.argv) > 2 else 'ibl_radius_count'
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = [i for i, r in enumerate(rows) if key in r['Kernel_Name']][-1]
step = rows[max(0, last - 8):]
t0 = int(step[0]['Start_Timestamp'])
prev_end = t0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    grid = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) * int(r.get('Grid_Size_Y', 1) or 1)
    wg = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1)
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  wgs {grid // max(wg, 1):7d}  {r['Kernel_Name'].split('(')[0][:70]}")
    prev_end = max(prev_end, e)
