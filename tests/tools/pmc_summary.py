"""Summarise rocprofv3 --pmc counter_collection.csv files: per-counter mean over the LAST n
dispatches of the named kernel (the timed steps, after spin-up)."""
import csv, glob, sys, collections
root, kern, last = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for name, vals in by.items():
        vals.sort()
        v = [x for _, x in vals[-last:]]
        print(f"{name:28s} {sum(v)/len(v):.6g}")
