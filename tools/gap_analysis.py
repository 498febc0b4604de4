"""Gaps between consecutive kernels of one training step, from a rocprofv3 --kernel-trace CSV: per boundary
(previous kernel -> next kernel) the idle time between the end of one and the start of the next, summed per pair of names."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", n); n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.split(r"[<(I]", n)[0][:40] if not n.startswith("void") else n[5:60]
# one step = from one cast_weights launch to the next (first kernel of the training forward)
starts = [i for i, r in enumerate(rows) if "cast_weights" in r["Kernel_Name"]]
# keep boundaries where a new step begins (gap between cast_weights launches > 1 ms)
steps = [s for k, s in enumerate(starts) if k == 0 or int(rows[s]["Start_Timestamp"]) - int(rows[starts[k - 1]]["Start_Timestamp"]) > 2e6]
if len(steps) < 4: sys.exit("too few steps found: %d" % len(steps))
a, b = steps[-3], steps[-2]
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
gaps = collections.Counter(); cnt = collections.Counter(); tot_gap = 0; neg = 0
for p, n in zip(seg, seg[1:] + [rows[b]]):
    g = int(n["Start_Timestamp"]) - int(p["End_Timestamp"])
    if g < 0: neg += g; continue
    key = short(p["Kernel_Name"]) + " -> " + short(n["Kernel_Name"])
    gaps[key] += g; cnt[key] += 1; tot_gap += g
print("step: %d launches, wall %.3f ms, kernel time %.3f ms, idle between kernels %.3f ms (overlap %.3f ms)" % (len(seg), (t1 - t0) / 1e6, busy / 1e6, tot_gap / 1e6, -neg / 1e6))
hist = collections.Counter()
for p, n in zip(seg, seg[1:] + [rows[b]]):
    g = int(n["Start_Timestamp"]) - int(p["End_Timestamp"])
    hist[min(max(g, 0) // 1000, 20)] += 1
print("gap histogram (us: count):", sorted(hist.items()))
for k, v in gaps.most_common(25):
    print("%8.1f us in %3d gaps (%.1f each)  %s" % (v / 1e3, cnt[k], v / 1e3 / cnt[k], k))
agg = collections.Counter(); n = collections.Counter()
for r in seg:
    k = short(r["Kernel_Name"]); agg[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n[k] += 1
print("kernel time of the step by kernel:")
for k, v in agg.most_common(30):
    print("%8.1f us  %3d launches  %s" % (v / 1e3, n[k], k))
