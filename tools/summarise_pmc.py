"""gpurun_out/prof/{fetch,write}/bench_counter_collection.csv -> average counter value per kernel (JSON)."""
import csv, glob, json, sys
out = {}
for name, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    acc = {}
    for f in glob.glob("gpurun_out/prof/%s/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != name:
                continue
            k = r["Kernel_Name"].split("(")[0]
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    out[name] = {k: {"dispatches": n, "avg_KB": s / n} for k, (n, s) in acc.items()}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if "scan" in kk or "merge" in kk} for k, v in out.items()}, indent=1))
