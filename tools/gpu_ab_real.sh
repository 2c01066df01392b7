#!/bin/bash
# A/B of bench.py --config real (the reference's own workload) under different environments.
# Usage (through gpurun): bash tools/gpu_ab_real.sh <tag> "<ENV=..>" "<ENV=..>" ...
set -o pipefail
TAG=${1:-rXX}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for envs in "$@"; do
  echo "== $envs" | tee -a $OUT/ab.log
  env $envs timeout -k 10 200 python bench.py --config real --steps 40 --warmup 8 --blocks 10 --no-cpu-baseline > $OUT/x.json 2> $OUT/x.err || { tail -5 $OUT/x.err; exit 1; }
  python - $OUT/x.json <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], "resident", d["config"]["resident_batches"]["value"], d["config"]["resident_batches"]["ms_per_step"], "K11", r["achieved"], {k: v["avg_launch_us"] for k, v in r["by_kind"].items()})
PY
done
