"""One step's kernel sequence (start offset, duration, gap to the previous kernel's end, short name, grid) from a rocprofv3
rocpd database:  python tools/rocpd_timeline.py <results.db> <launches per step | 0 = find the period> [step index]"""
import re, sqlite3, sys
db = sys.argv[1]; per = int(sys.argv[2]); idx = int(sys.argv[3]) if len(sys.argv) > 3 else -2
con = sqlite3.connect(db); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
dcols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
gx = "grid_size_x" if "grid_size_x" in dcols else "grid_x"
wx = "workgroup_size_x" if "workgroup_size_x" in dcols else "workgroup_x"
rows = cur.execute("select d.start, d.end, s.%s, d.%s, d.%s from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, gx, wx, kd, ks)).fetchall()
if per <= 0:   # find the step's period in the tail of the trace (the last launches are whole steps)
    names = [r[2] for r in rows]
    tail = names[-3000:] if len(names) > 3000 else names
    per = next((p for p in range(20, len(tail) // 3) if all(tail[-1 - i] == tail[-1 - i - p] for i in range(2 * p))), 141)
    rows = rows[len(rows) % per:]
n = len(rows) // per
if idx < 0: idx += n
seg = rows[idx * per:(idx + 1) * per]
t0 = seg[0][0]; prev_end = t0; busy = 0
for st, en, nm, g, w in seg:
    m = re.search(r"N_1\d+(\w+?)I", nm) or re.search(r"(\w+?)[<(]", nm)
    short = re.sub(r"^\d+", "", m.group(1)) if m else nm[:40]
    e = re.search(r"kernelI\w*?Li(\d+)E", nm)
    print("%8.1f us  +%7.2f  gap %6.2f  %-28s %s blocks %d" % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, short[:28], ("epi %s" % e.group(1)) if e else "", g // max(w, 1)))
    busy += en - st; prev_end = max(prev_end, en)
print("step span %.1f us, kernel time %.1f us, launches %d (of %d steps in the trace)" % ((prev_end - t0) / 1e3, busy / 1e3, len(seg), n))
