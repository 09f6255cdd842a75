"""<dir>/**/*kernel_stats.csv (rocprofv3 --kernel-trace --stats) -> the per-kernel table as text, longest first."""
import csv, glob, os, sys

root = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("%-100s %8s %12s %12s %7s" % ("kernel", "calls", "avg_us", "total_us", "pct"))
for r in rows:
    print("%-100s %8s %12.1f %12.1f %7s" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3,
                                           r["Percentage"]))
