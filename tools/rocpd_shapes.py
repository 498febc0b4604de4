"""Kernel time by (kernel, grid size) from a rocprofv3 rocpd database - the NT GEMM shapes of a layer differ by their tile
count:  python tools/rocpd_shapes.py <results.db> [name substring ...]"""
import re, sqlite3, sys
db = sys.argv[1]; pats = sys.argv[2:] or ["gemm_nt_pp", "gemm_nt_ld"]
con = sqlite3.connect(db); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
dcols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
gx = "grid_size_x" if "grid_size_x" in dcols else "grid_x"
wx = "workgroup_size_x" if "workgroup_size_x" in dcols else "workgroup_x"
rows = cur.execute("select s.%s, d.%s, d.%s, count(*), sum(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by 1, 2, 3 order by 5 desc"
                   % (name_col, gx, wx, kd, ks)).fetchall()
tot = 0.0
for n, g, w, c, t in rows:
    if not any(p in n for p in pats): continue
    m = re.search(r"(gemm_nt_\w+?)_kernelI\w*?Li(\d+)E", n)
    short = "%s<epi %s>" % (m.group(1), m.group(2)) if m else n[:60]
    print("%-28s blocks %5d  calls %5d  avg %7.2f us  total %9.1f us" % (short, g // max(w, 1), c, t / c / 1e3, t / 1e3))
    tot += t / 1e3
print("sum %.1f us" % tot)
