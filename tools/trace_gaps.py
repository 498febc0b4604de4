"""Idle time between kernels from a rocprofv3 kernel trace (csv): usage trace_gaps.py <kernel_trace.csv> [skip_fraction]
Prints device busy/idle over the steady-state part and the largest gap contributors by (previous kernel -> next kernel)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48]) for r in rows))
n = len(ev)
lo = int(n * float(sys.argv[2]) if len(sys.argv) > 2 else n * 0.3); hi = int(n * 0.8)
ev = ev[lo:hi]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = 0; cur_end = ev[0][0]; gaps = collections.Counter(); gapn = collections.Counter(); prev = None
for s, e, k in ev:
    if s > cur_end:
        g = s - cur_end
        gaps[(prev, k)] += g; gapn[(prev, k)] += 1
        busy += 0
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e; prev = k
print("span %.3f ms, busy %.3f ms (%.1f%%), idle %.3f ms over %d kernels" % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), (t1 - t0 - busy) / 1e6, len(ev)))
for (a, b), g in gaps.most_common(14):
    print("  %8.1f us total  %5.1f us avg x%4d   %s -> %s" % (g / 1e3, g / 1e3 / gapn[(a, b)], gapn[(a, b)], a, b))
