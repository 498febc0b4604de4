mkdir -p gpurun_out/r05_ks3
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-step --no-per-shape --no-secondary --no-roofline"
for ks in 0 1; do
  TIMHIP_ATTN_KS=$ks timeout 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r05_ks3/prof$ks -o c2a -- $B --steps 40 --warmup 5 > /dev/null 2>&1
  DB=$(find /root/repo/gpurun_out/r05_ks3/prof$ks -name "*.db" | head -1)
  python /root/repo/tools/rocpd_stats.py $DB > /root/repo/gpurun_out/r05_ks3/kernel_stats_ks$ks.csv 2>/dev/null
  rm -rf /root/repo/gpurun_out/r05_ks3/prof$ks
done
cd /root/repo
grep -h "attn\|TOTAL" gpurun_out/r05_ks3/kernel_stats_ks0.csv gpurun_out/r05_ks3/kernel_stats_ks1.csv
bash tools/r05_ab.sh r05_ks3_ab 3 TIMHIP_ATTN_KS=0 TIMHIP_ATTN_KS=1 > /dev/null 2>&1; cat gpurun_out/r05_ks3_ab/ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_parity.py tests/test_gpu_train_step.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -3
