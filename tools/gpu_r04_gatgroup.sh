#!/bin/bash
# Round 4: clustered GAT aggregation: clusters walked together through all their slices (GTS_OPT_GAT_CLUSTER_GROUP = option 16).
set -o pipefail
OUT=gpurun_out/${1:-r04m}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
GTS_OPTIONS="16=16" timeout -k 10 300 python -m pytest tests/test_gpu_gat_cluster.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest (group 16) rc=$?"; tail -2 $OUT/pytest.log
grep -q " passed" $OUT/pytest.log || exit 1
for g in 0 8 16 32 64 128 0 16; do
  GTS_OPTIONS="16=$g" timeout -k 10 300 python bench.py --config c3 --steps 5 --warmup 2 --blocks 3 --no-cpu-baseline > $OUT/c3_$g.json 2> $OUT/c3_$g.err
  python - $OUT/c3_$g.json $g <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("group", sys.argv[2], d["value"], d["ms_per_step"], [(h["kernel"], h["avg_launch_us"], h["frac"]) for h in d["roofline_hbm"]])
PY
done
