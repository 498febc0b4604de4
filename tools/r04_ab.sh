# A/B of two library builds on one box, interleaved:  bash tools/r04_ab.sh <tag> <libA> <libB> [runs]
TAG=$1; A=$2; B=$3; N=${4:-3}
O=/root/repo/gpurun_out/$TAG; mkdir -p $O
cd /root/repo
Q="python bench.py --no-cpu-baseline --no-extra-step --no-per-shape --no-secondary --steps 40 --warmup 10"
for i in $(seq 1 $N); do
  for L in $A $B; do
    TIM_AMD_LIB=/root/repo/tim_amd/$L timeout 300 $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['ms_per_step'], d['roofline']['frac'], 'eager', d.get('eager',{}).get('ms_per_step'), d.get('eager',{}).get('host_issue_ms_per_step'))" >> $O/ab.txt
  done
done
cat $O/ab.txt
