#!/bin/bash
# Round 4: activation fragments fetched coalesced + ds_bpermute (AP): parity, then A/B of the steps.
set -o pipefail
OUT=gpurun_out/${1:-r04h}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_stack.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $OUT/pytest.log
grep -q " passed" $OUT/pytest.log || exit 1
GTS_OPTIONS="7=65" timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_stack.py -m gpu -x -q -k "fragment_order or one_call or chained" > $OUT/pytest65.log 2>&1
echo "pytest (AP at every height) rc=$?"; tail -2 $OUT/pytest65.log
bash tools/gpu_ab_real.sh ${1:-r04h} "GTS_OPTIONS=7=1" "GTS_OPTIONS=7=33" "GTS_OPTIONS=7=1"
AB_FLAGS="--steps 20 --warmup 5 --no-cpu-baseline --blocks 10" bash tools/gpu_ab.sh ${1:-r04h}c2 "GTS_OPTIONS=7=1" "GTS_OPTIONS=7=65" "GTS_OPTIONS=7=1" "GTS_OPTIONS=7=65"
