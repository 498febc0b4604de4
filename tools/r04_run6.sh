cd /root/repo
O=/root/repo/gpurun_out/r04_f; mkdir -p $O
uptime > $O/host.txt
bash tools/r04_ab.sh r04_f libtimhip_base.so libtimhip.so 3
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/gpu_suite.txt
cat $O/host.txt $O/gpu_suite.txt
