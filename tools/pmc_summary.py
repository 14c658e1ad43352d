"""Sums rocprofv3 --pmc counter_collection CSVs per kernel and counter; per-dispatch = total / dispatches that carry the counter."""
import csv, glob, sys, collections, json
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
ids = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("wpk::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ids[k][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
out = {}
for k in acc:
    print("==", k)
    out[k] = {}
    for c, v in sorted(acc[k].items()):
        n = max(1, len(ids[k][c]))
        out[k][c] = v / n
        print("   %-28s total %.4g   dispatches %d   per dispatch %.4g" % (c, v, n, v / n))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
