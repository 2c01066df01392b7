#!/bin/bash
# C3 (GAT 4 x 4 x 256) with the clustered GAT kernels against the plain ones; cluster limits, waves per workgroup and
# workgroups per CU of the clustered form.  Usage (through gpurun): bash tools/gpu_ab_gat_cluster.sh <tag>
set -o pipefail
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() {   # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --config c3 --no-cpu-baseline --blocks 6 > $OUT/c3_$label.json 2> $OUT/c3_$label.err || { tail -5 $OUT/c3_$label.err; return 1; }
  python - $OUT/c3_$label.json "$label" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
hb = {e["kernel"]: (e.get("avg_launch_us"), e.get("frac")) for e in d.get("roofline_hbm", [])}
print(sys.argv[2], d["value"], d["ms_per_step"], hb, flush=True)
PY
}
run plain GTS_CLUSTER_GAT=0 && run default GTS_CLUSTER_GAT=1
