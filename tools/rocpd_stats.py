"""Per-kernel statistics from a rocprofv3 rocpd database (the default output format of rocprofv3 on this image when
--output-format csv is not given):  python tools/rocpd_stats.py <results.db> [steps]  -> CSV on stdout
(name, calls, total_us, avg_us, pct, us_per_step)."""
import re
import sqlite3
import sys

db = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
con = sqlite3.connect(db)
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
# (round 6) a dispatch longer than 20 x its kernel's median is an artefact of the profiled process (first touch of a fresh allocation,
# a page migration: one 28-ms launch of a 50-us kernel moved a family's per-step figure by 5 %): dropped from the statistics, counted
# in the last column
per = {}
for n, dur in cur.execute("select s.%s, d.end - d.start from %s d join %s s on d.kernel_id = s.id" % (name_col, kd, ks)):
    per.setdefault(n, []).append(dur)
rows = []
for n, ds in per.items():
    ds.sort()
    med = ds[len(ds) // 2]
    keep = [x for x in ds if x <= 20 * med] or ds
    rows.append((n, len(keep), sum(keep), keep[0], keep[-1], len(ds) - len(keep)))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print("name,calls,total_us,avg_us,min_us,max_us,pct,us_per_step,outliers_dropped")
for n, c, t, mn, mx, drop in rows:
    n = re.sub(r"\s+", " ", n)
    print('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f,%s,%d' % (n[:160], c, t / 1e3, t / c / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot,
                                                   "%.1f" % (t / 1e3 / steps) if steps else "", drop))
print('"TOTAL",%d,%.1f,,,,100,%s,%d' % (sum(r[1] for r in rows), tot / 1e3, "%.1f" % (tot / 1e3 / steps) if steps else "", sum(r[5] for r in rows)))
