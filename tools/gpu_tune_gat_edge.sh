#!/bin/bash
# Cluster limits of the GAT edge pass (third part of GTS_GAT_CLUSTER_LIMITS), C3 step and the bracketed call times.
set -o pipefail
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() {
  local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --config c3 --no-cpu-baseline --blocks 4 > $OUT/c3_$label.json 2> $OUT/c3_$label.err || { tail -5 $OUT/c3_$label.err; return 1; }
  python - $OUT/c3_$label.json "$label" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
hb = {e["kernel"]: e.get("avg_launch_us") for e in d.get("roofline_hbm", [])}
print(sys.argv[2], d["value"], hb, flush=True)
PY
}
G="32,64,256;32,64,256"
run e24_50 "GTS_GAT_CLUSTER_LIMITS=$G;24,50,192" && run e32_64_1wg "GTS_GAT_CLUSTER_LIMITS=$G;32,64,256" GTS_OPTIONS=15=16 && \
run e20_44 "GTS_GAT_CLUSTER_LIMITS=$G;20,44,160" && run e16_36_x3 "GTS_GAT_CLUSTER_LIMITS=$G;16,36,128" GTS_OPTIONS=11=3,15=10 && \
run e24_50_w16 "GTS_GAT_CLUSTER_LIMITS=$G;24,50,192" GTS_OPTIONS=15=16 && run e24_50_w8 "GTS_GAT_CLUSTER_LIMITS=$G;24,50,192" GTS_OPTIONS=15=8
