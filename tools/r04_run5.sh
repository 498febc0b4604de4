cd /root/repo
O=/root/repo/gpurun_out/r04_e; mkdir -p $O
uptime > $O/host.txt
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_graph.json 2> $O/bench_graph.err
timeout 600 python bench.py --no-cpu-baseline --no-secondary --step-mode eager > $O/bench_eager.json 2> $O/bench_eager.err
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-step --no-per-shape --no-secondary"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o c2a -- $B --steps 35 --warmup 5 > $O/bench_profiled_run.json 2> $O/prof.err
cd /root/repo
python tools/rocpd_stats.py $(find $O/prof -name "*.db" | head -1) > $O/c2a_kernel_stats.csv 2> $O/c2a.err
rm -rf $O/prof
cat $O/host.txt; tail -3 $O/bench_graph.err; python - <<'PY'
import json
for f in ("bench_graph","bench_eager","bench_profiled_run"):
    try:
        d=json.loads(open("/root/repo/gpurun_out/r04_e/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["step_mode"][:30], d.get("eager"), d["roofline"]["frac"], d["roofline"].get("events_from","")[:40], d.get("graph_replay"))
    except Exception as e: print(f, "ERR", e)
PY
head -5 $O/c2a_kernel_stats.csv | cut -c1-150; tail -1 $O/c2a_kernel_stats.csv
