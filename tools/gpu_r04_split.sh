#!/bin/bash
# Round 4: two-pass epilogue / wave-priority experiments on the panel GEMMs (GTS_OPT_GEMM_SCHED = option 7).
set -o pipefail
OUT=gpurun_out/${1:-r04c}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
GTS_OPTIONS="7=49" timeout -k 10 600 python -m pytest tests/test_gpu_stack.py tests/test_gpu_full_size.py -m gpu -x -q > $OUT/pytest_split3.log 2>&1
echo "pytest (split 3) rc=$?"; tail -3 $OUT/pytest_split3.log
bash tools/gpu_ab.sh ${1:-r04c} "GTS_OPTIONS=7=1" "GTS_OPTIONS=7=17" "GTS_OPTIONS=7=33" "GTS_OPTIONS=7=49" "GTS_OPTIONS=7=129" "GTS_OPTIONS=7=257" "GTS_OPTIONS=7=177" "GTS_OPTIONS=7=1"
