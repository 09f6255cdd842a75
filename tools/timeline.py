"""rocprofv3 kernel trace csv -> the launch sequence of the LAST query (between two init/reset kernels): start offset,
duration and the idle gap before each kernel, in microseconds.   python tools/timeline.py <dir> [first-kernel-substring]"""
import csv, glob, os, sys

root = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "init_table_kernel"
per_query = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # markers per query
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
# the last complete query: from the second-to-last marker to the last one
if len(starts) < 1 + per_query:
    lo, hi = 0, len(rows)
else:
    lo, hi = starts[-1 - per_query], starts[-1]
    # several markers in a row belong to one query: back up to the first of the run
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
print("%10s %10s %8s  %s" % ("start_us", "dur_us", "gap_us", "kernel"))
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.1f %10.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:110]))
    prev_end = max(prev_end, e)
print("query span: %.1f us" % ((prev_end - t0) / 1e3))
