cd /root/repo
O=/root/repo/gpurun_out/r04_calib; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o $O/pmc_calib tools/pmc_calib.hip
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o p --output-format csv -- $O/pmc_calib > $O/run_f.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o p --output-format csv -- $O/pmc_calib > $O/run_w.txt 2>&1
cd /root/repo
python tools/pmc_calib.py $O > $O/pmc_calibration.json 2> $O/calib.err
cat $O/pmc_calibration.json; tail -3 $O/calib.err
rm -rf $O/f $O/w $O/pmc_calib
