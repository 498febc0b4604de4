#!/bin/bash
# kernel stats + one step's timeline of the C2a step (eager and replayed) and of the C4 training step:  bash tools/r05_profile.sh <tag>
TAG=${1:-r05_0}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-step --no-per-shape --no-secondary --no-roofline"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof -o c2a -- $B --steps 40 --warmup 5 > $OUT/bench_profiled_run.json 2> /dev/null
cd /root/repo
DB=$(find $OUT/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_stats.csv 2> $OUT/kernel_stats.err
python tools/rocpd_timeline.py $DB 0 -2 > $OUT/timeline_c2a.txt 2>&1
rm -rf $OUT/prof
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_c4 -o c4 -- python /root/repo/tools/prof_secondary.py C4 16 30 --det-train > /dev/null 2>&1
cd /root/repo
DB=$(find $OUT/prof_c4 -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/c4_train_kernel_stats.csv 2> $OUT/c4.err
python tools/rocpd_timeline.py $DB 0 -2 > $OUT/timeline_c4.txt 2>&1
rm -rf $OUT/prof_c4
tail -3 $OUT/timeline_c2a.txt $OUT/timeline_c4.txt
