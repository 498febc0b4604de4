# A/B of environment settings on one box, interleaved:  bash tools/r04_env_ab.sh <tag> <runs> "ENV1=a" "ENV1=b" ...
TAG=$1; N=$2; shift 2
O=/root/repo/gpurun_out/$TAG; mkdir -p $O
cd /root/repo
Q="python bench.py --no-cpu-baseline --no-extra-step --no-per-shape --no-secondary --steps 40 --warmup 10"
for i in $(seq 1 $N); do
  for E in "$@"; do
    env $E timeout 300 $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$E', d['ms_per_step'], d['roofline']['frac'], 'eager', d.get('eager',{}).get('ms_per_step'))" >> $O/ab.txt
  done
done
cat $O/ab.txt
