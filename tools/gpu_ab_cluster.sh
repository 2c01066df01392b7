#!/bin/bash
# In-step A/B of the clustered K1 / K2: bench.py at several --graphs-per-gpu under different environments.
# Usage: bash tools/gpu_ab_cluster.sh <tag> "<graphs per gpu ...>" "<ENV=.. ENV=..>" ...
set -o pipefail
TAG=${1:-rXX}; shift
BS=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for b in $BS; do
  steps=20; [ $b -ge 16 ] && steps=5
  i=0
  for envs in "$@"; do
    i=$((i+1))
    env $envs timeout -k 10 300 python bench.py --graphs-per-gpu $b --steps $steps --warmup 3 --blocks 5 --no-cpu-baseline > $OUT/ab_b${b}_$i.json 2> $OUT/ab_b${b}_$i.err || { tail -20 $OUT/ab_b${b}_$i.err; exit 1; }
    python - $OUT/ab_b${b}_$i.json "b=$b [$envs]" <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["unit"], d["ms_per_step"], "ms/step |",
      " ".join(f'{h["kernel"]} {h["avg_launch_us"]}us frac {h["frac"]}' for h in d["roofline_hbm"]))
PY
  done
done
