import csv, glob, collections, sys
root = sys.argv[1]
for d in sorted(glob.glob(root + "/*")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no counter file", glob.glob(d + "/**/*", recursive=True)[:6]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:70]
        if any(s in k for s in sys.argv[2:]):
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        print(d.split("/")[-1], k)
        for c, x in v.items():
            print("   %-28s %.5g per launch" % (c, x / max(1, cnt[(k, c)])))
