"""<dir>/**/*counter_collection.csv (one rocprofv3 --pmc pass of SQ counters) -> per kernel the average of every counter per
dispatch, and the share of the waves' cycles that were parked (s_waitcnt / barrier), issue-stalled, or issuing."""
import csv, glob, os, sys

acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        a = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k, cs in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", [1, 0])[1]):
    avg = {c: s / n for c, (n, s) in cs.items()}
    wc = avg.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-62s disp %3d  wave_cycles %.3e  parked %.2f  issue_stall %.2f  active %.2f  | VALU %.3e LDS %.3e  lds_stall %.2f  bank_conflict %.3e" % (
        k, cs["SQ_WAVE_CYCLES"][0] if "SQ_WAVE_CYCLES" in cs else 0, wc, avg.get("SQ_WAIT_ANY", 0) / wc, avg.get("SQ_WAIT_INST_ANY", 0) / wc,
        avg.get("SQ_ACTIVE_INST_ANY", 0) / wc, avg.get("SQ_INSTS_VALU", 0), avg.get("SQ_INSTS_LDS", 0), avg.get("SQ_WAIT_INST_LDS", 0) / wc,
        avg.get("SQ_LDS_BANK_CONFLICT", 0)))
