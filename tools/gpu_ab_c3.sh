#!/bin/bash
# A/B of bench.py --config c3 under different environments.  Usage (through gpurun): bash tools/gpu_ab_c3.sh <tag> "<ENV=..>" ...
set -o pipefail
TAG=${1:-rXX}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for envs in "$@"; do
  echo "== $envs" | tee -a $OUT/ab.log
  env $envs timeout -k 10 200 python bench.py --config c3 --no-cpu-baseline --blocks 6 > $OUT/x.json 2> $OUT/x.err || { tail -5 $OUT/x.err; exit 1; }
  python - $OUT/x.json <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
hb = {e["kernel"]: (e.get("avg_launch_us"), e.get("frac")) for e in d.get("roofline_hbm", [])}
print(d["value"], d["ms_per_step"], "K11", d["roofline"]["achieved"], hb, flush=True)
PY
done
