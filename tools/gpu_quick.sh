#!/bin/bash
# Short GPU-box session: the -m gpu tests (or a subset), then the headline bench line.
# Usage (through gpurun): bash tools/gpu_quick.sh <tag> [pytest args...]
set -o pipefail
TAG=${1:-rXX}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=15 "$@" > $OUT/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err || { tail -20 $OUT/bench_c2.err; exit 1; }
cat $OUT/bench_c2.json
