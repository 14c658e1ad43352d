"""Sums rocprofv3 --pmc counter_collection CSVs per kernel and counter."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for k in acc:
    n = max(1, len(cnt[k]) // max(1, len(glob.glob(root + "/p*/"))))
    print("==", k, "dispatches/pass ~", n)
    for c, v in sorted(acc[k].items()):
        print("   %-28s total %.4g   per dispatch %.4g" % (c, v, v / n))
