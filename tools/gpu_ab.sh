#!/bin/bash
# A/B of bench.py under different environments.  Usage: bash tools/gpu_ab.sh <tag> "<ENV=.. ENV=..>" "<ENV..>" ...
# (first argument after the tag that starts with "--" is passed to bench.py as extra flags)
set -o pipefail
TAG=${1:-rXX}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
FLAGS="${AB_FLAGS:---steps 20 --warmup 5 --no-cpu-baseline}"
i=0
for envs in "$@"; do
  i=$((i+1))
  echo "== [$i] $envs" | tee -a $OUT/ab.log
  env $envs timeout -k 10 300 python bench.py $FLAGS > $OUT/ab_$i.json 2> $OUT/ab_$i.err || { tail -20 $OUT/ab_$i.err; exit 1; }
  python - $OUT/ab_$i.json <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], "ms/step; blocks", d["blocks"]["min"], d["blocks"]["max"],
      "| K11", r["achieved"], "TF", {k: (v["avg_launch_us"], v["tflops"]) for k, v in r["by_kind"].items()},
      "|", [(h["kernel"], h["avg_launch_us"]) for h in d["roofline_hbm"]])
PY
done
