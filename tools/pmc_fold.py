"""Average every counter per kernel from rocprofv3 --pmc CSVs: python tools/pmc_fold.py <dir> [name filter]"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        a = acc[row["Kernel_Name"]][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in acc.items():
    if flt and flt not in k: continue
    print(k[:100]); print("   " + "  ".join("%s=%.4g" % (c, v[0] / v[1]) for c, v in sorted(cs.items())))
