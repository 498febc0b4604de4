cd /root/repo
O=/root/repo/gpurun_out/r04_b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/gpu_suite.txt
TIM_AMD_LIB=/root/repo/tim_amd/libtimhip_tuning.so timeout 900 python -m pytest tests -m "gpu and tuning" -x -q 2>&1 | tail -15 > $O/gpu_tuning_suite.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o c4 -- python /root/repo/tools/prof_secondary.py C4 16 30 --det-train > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c1 -o c1 -- python /root/repo/tools/prof_secondary.py C1 64 30 > /dev/null 2>&1
cd /root/repo
python tools/rocpd_stats.py $(find $O/prof_c4 -name "*.db" | head -1) > $O/c4_kernel_stats.csv 2> $O/c4.err
python tools/rocpd_stats.py $(find $O/prof_c1 -name "*.db" | head -1) > $O/c1_kernel_stats.csv 2> $O/c1.err
rm -rf $O/prof_c4 $O/prof_c1
cat $O/gpu_suite.txt $O/gpu_tuning_suite.txt
