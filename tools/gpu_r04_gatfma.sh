#!/bin/bash
# Round 4: GAT kernels with fused multiply-adds, the mask-free common path and the scalar unit walk: parity tests, what-if timings and
# phase stamps, A/B of the three passes, C3 bench.
set -o pipefail
OUT=gpurun_out/${1:-r04u}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_gat_cluster.py tests/test_gpu_kernels.py ${FULL:+tests/test_gpu_full_size.py} -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest.log
grep -q " passed" $OUT/pytest.log || exit 1
GTS_DIAG_REBUILD=1 timeout -k 10 300 python tools/diag/gat_whatif.py 4 2>&1 | grep -v 'warning\|amdgpu.ids\|\^\|__global__\|In file' | tee $OUT/whatif.log | grep -v 'per_cu 1'
for args in "" "--waves 8"; do
  echo "== $args"
  timeout -k 10 200 python tools/diag/gat_passes_ab.py --group 16 $args 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.log
done
timeout -k 10 300 python bench.py --config c3 --steps 10 --warmup 3 --blocks 5 --no-cpu-baseline > $OUT/c3.json 2> $OUT/c3.err || { tail -5 $OUT/c3.err; exit 1; }
python - $OUT/c3.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c3", d["value"], d["ms_per_step"], [(h["kernel"], h["avg_launch_us"], h["frac"]) for h in d["roofline_hbm"]])
PY
