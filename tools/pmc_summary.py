"""Parse rocprofv3 --pmc CSV (counter_collection) -> mean counter value per kernel name fragment."""
import csv, glob, json, sys, collections
root, counter = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if row.get("Counter_Name") == counter:
            acc[row["Kernel_Name"][:90]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"{counter} mean={sum(v)/len(v):14.1f} n={len(v):3d} {k}")
