cd /root/repo
O=/root/repo/gpurun_out/r04_c; mkdir -p $O
bash tools/r04_ab.sh r04_c libtimhip_base.so libtimhip.so 3
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/gpu_suite.txt
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-step --no-graph --no-per-shape --no-secondary"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o c2a -- $B --steps 35 --warmup 5 > $O/bench_profiled_run.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o c4 -- python /root/repo/tools/prof_secondary.py C4 16 30 --det-train > /dev/null 2>&1
cd /root/repo
python tools/rocpd_stats.py $(find $O/prof -name "*.db" | head -1) > $O/c2a_kernel_stats.csv 2> $O/c2a.err
python tools/rocpd_stats.py $(find $O/prof_c4 -name "*.db" | head -1) > $O/c4_kernel_stats.csv 2> $O/c4.err
rm -rf $O/prof $O/prof_c4
cat $O/gpu_suite.txt
