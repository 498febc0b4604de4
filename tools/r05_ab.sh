#!/bin/bash
# interleaved A/B of environment switches on ONE box:  bash tools/r05_ab.sh <tag> <rounds> "ENV_A" "ENV_B" ...
TAG=$1; ROUNDS=$2; shift 2
OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /root/repo
for r in $(seq 1 $ROUNDS); do
  i=0
  for arm in "$@"; do
    env $arm python bench.py --no-cpu-baseline --no-secondary --no-per-shape --no-extra-step --no-roofline --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('arm %d [%s] round %s: %.3f ms replay, eager %.3f (host %.3f)' % ($i, '$arm', '$r', d['ms_per_step'], d['eager']['ms_per_step'], d['eager']['host_issue_ms_per_step']))" | tee -a $OUT/ab.txt
    i=$((i+1))
  done
done
