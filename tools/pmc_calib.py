"""Read the rocprofv3 --pmc passes of tools/pmc_calib.hip:  python tools/pmc_calib.py <dir with f/ and w/>  -> JSON
reported / true byte ratios per access pattern (FETCH_SIZE and WRITE_SIZE are in KiB)."""
import collections, csv, glob, json, sys
root = sys.argv[1]
MiB = 1 << 20
true_read = {"calib_read16": 1024 * MiB, "calib_dma16": 1024 * MiB, "calib_touch_line": 1024 * MiB, "calib_touch_half": 1024 * MiB,
             "calib_rmw16": 1024 * MiB}
true_write = {"calib_write16": 1024 * MiB, "calib_write16_nt": 1024 * MiB, "calib_write4": 1024 * MiB, "calib_rmw16": 1024 * MiB}
vals = collections.defaultdict(list)
for sub, ctr in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (root, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                name = r["Kernel_Name"].split("(")[0].split(" ")[-1]
                vals[(ctr, name)].append(float(r["Counter_Value"]) * 1024)
out = {}
for (ctr, name), v in sorted(vals.items()):
    big = [x for x in v if True]
    first = v[0]
    ent = {"launches": len(v), "first_launch_reported_MiB": round(first / MiB, 1)}
    tr = (true_read if ctr == "FETCH_SIZE" else true_write).get(name)
    if tr:
        ent["true_MiB_first_launch"] = tr / MiB
        ent["reported_over_true"] = round(first / tr, 4)
    if len(v) > 1:
        rest = v[1:]
        ent["pair_launches_reported_MiB_each"] = round(sum(rest) / len(rest) / MiB, 2)
    out["%s %s" % (ctr, name)] = ent
print(json.dumps(out, indent=1))
