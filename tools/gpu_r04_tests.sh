#!/bin/bash
# Round 4: the full GPU suite in one process, log under gpurun_out/<tag>/.
set -o pipefail
OUT=gpurun_out/${1:-r04t}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/pytest_gpu.log
