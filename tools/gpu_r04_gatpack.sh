#!/bin/bash
# Round 4: GATConv's fc / res_fc weights in fragment order: parity, then A/B of the C3 step.
set -o pipefail
OUT=gpurun_out/${1:-r04i}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_full_size.py tests/test_gpu_gat_cluster.py -m gpu -x -q -k "gat or GAT or c3" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $OUT/pytest.log
grep -q " passed" $OUT/pytest.log || exit 1
for v in 1 0 1 0; do
  GTS_PACK_GAT=$v timeout -k 10 300 python bench.py --config c3 --steps 10 --warmup 3 --blocks 5 --no-cpu-baseline > $OUT/c3_$v.json 2> $OUT/c3_$v.err
  python - $OUT/c3_$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("GTS_PACK_GAT=" + sys.argv[2], d["value"], d["ms_per_step"], d["blocks"]["min"], d["blocks"]["max"], "K11", r["frac"], {k: (v["avg_launch_us"], v["tflops"]) for k, v in r["by_kind"].items()})
PY
done
