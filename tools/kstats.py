"""Print the top rows of a rocprofv3 kernel_stats.csv found under a directory (GPU-box helper for the perf_* tools)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in list(csv.DictReader(open(f)))[:n]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), f'{float(r["AverageNs"]) / 1e3:10.1f} us', r["Percentage"].rjust(7))
