#!/bin/bash
# Round 4: clustered GAT aggregation with its units dealt off a counter per XCD: parity, L2 replay (FETCH_SIZE), A/B against static dealing, C3.
set -o pipefail
OUT=gpurun_out/${1:-r04d}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_gat_cluster.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $OUT/pytest.log
grep -q " passed" $OUT/pytest.log || exit 1
timeout -k 10 300 bash tools/diag/run_gat_l2_replay.sh ${1:-r04d}_replay | tee $OUT/replay.log
for opt in "17=1" "17=0" "17=1" "17=0"; do
  echo "== GTS_OPTIONS=$opt"
  GTS_OPTIONS="$opt" timeout -k 10 200 python tools/diag/gat_passes_ab.py --group 16 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.log
done
for opt in "17=1" "17=0" "17=1" "17=0"; do
  GTS_OPTIONS="$opt" timeout -k 10 300 python bench.py --config c3 --steps 10 --warmup 3 --blocks 5 --no-cpu-baseline > $OUT/c3.json 2> $OUT/c3.err || { tail -5 $OUT/c3.err; exit 1; }
  python - $OUT/c3.json "$opt" <<'PY' | tee -a $OUT/c3.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c3", sys.argv[2], d["value"], d["ms_per_step"], [(h["kernel"], h["avg_launch_us"], h["frac"]) for h in d["roofline_hbm"]])
PY
done
