"""<dir>/{fetch,write}/**/*counter_collection.csv -> average FETCH_SIZE / WRITE_SIZE per kernel and dispatch (JSON on stdout).

rocprofv3 reports both counters in KB.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE counts a wide coalesced read at
half its bytes on gfx950: HBM read bytes ~= 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact."""
import csv, glob, json, os, sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
out = {}
if len(sys.argv) > 2:
    out["queries"] = int(sys.argv[2])  # executions of the query in each PMC pass (steps + warm-up): dispatches / queries = launches per query
for name, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    acc = {}
    for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != name:
                continue
            k = r["Kernel_Name"].split("(")[0]
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    out[name] = {k: {"dispatches": n, "avg_KB": s / n} for k, (n, s) in sorted(acc.items())}
try:  # the sources the profiled library was built from (bench.py drops `traffic` when they have changed since)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from query_amd import build as _b
    out["source_hash"] = _b.source_hash()
except Exception:
    pass
json.dump(out, sys.stdout, indent=1)
