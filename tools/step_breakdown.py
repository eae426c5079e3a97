#!/usr/bin/env python3
"""Per-kernel time of the LAST bench step from a rocprofv3 --kernel-trace csv (the step starts at its crop resampling)."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = [i for i, r in enumerate(rows) if 'ibl_resample_h' in r['Kernel_Name']][-1]
step = rows[last:]
agg = collections.OrderedDict()
for r in step:
    n = r['Kernel_Name'].split('(')[0][:64]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += d
tot = sum(v[1] for v in agg.values())
span = (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6
print("last step: kernels sum %.1f ms, span %.1f ms" % (tot, span))
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"{n:64s} {c:5d} {d:8.2f} ms")
