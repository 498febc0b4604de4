#!/bin/bash
# round 6: final check + evidence of the tree on one box: smoke, the whole GPU suite, then the evidence set (tools/collect_evidence.sh)
TAG=${1:-r06_g}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt; tail -2 $OUT/smoke.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
bash tools/collect_evidence.sh $TAG > $OUT/collect.log 2>&1
tail -5 $OUT/collect.log
TAGX=$TAG python - <<'PY'
import json, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/"+__import__("os").environ.get("TAGX","r06_g")+"/bench_default.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["repeat_ms"], d["value"], d["roofline"]["frac"], d["whole_step"], d["eager"], d["forward_only"])
PY
