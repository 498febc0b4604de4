cd /root/repo
O=/root/repo/gpurun_out/r04_d; mkdir -p $O
nproc > $O/host.txt; lscpu | grep -E "Model name|MHz|^CPU\(s\)" >> $O/host.txt; uptime >> $O/host.txt
python tools/host_bound.py > $O/host_bound.txt 2>&1
bash tools/r04_ab.sh r04_d libtimhip_base.so libtimhip.so 3
hipcc --offload-arch=gfx950 -O3 -o $O/coop_probe tools/coop_probe.hip && $O/coop_probe > $O/coop_probe.txt 2>&1; rm -f $O/coop_probe
cat $O/host.txt $O/coop_probe.txt; head -70 $O/host_bound.txt
