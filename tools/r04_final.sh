# final verification of a round's tree on one box: smoke, product suite, tuning suite
cd /root/repo
O=/root/repo/gpurun_out/r04_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/gpu_suite.txt
TIM_AMD_LIB=/root/repo/tim_amd/libtimhip_tuning.so timeout 1200 python -m pytest tests -m "gpu and tuning" -q 2>&1 | tail -4 > $O/gpu_tuning_suite.txt
tail -2 $O/smoke.txt; cat $O/gpu_suite.txt $O/gpu_tuning_suite.txt
