#!/bin/bash
# tools/tune_cluster.py under several cluster limits.  Usage: bash tools/gpu_tune_cluster.sh <tag> "<graphs per batch ...>" "<limits>" ...
set -o pipefail
TAG=${1:-rXX}; shift
BS=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lim in "$@"; do
  echo "== $lim" | tee -a $OUT/tune_cluster.log
  GTS_CLUSTER_LIMITS="$lim" timeout -k 10 300 python tools/tune_cluster.py $BS 2>&1 | grep -v amdgpu.ids | tee -a $OUT/tune_cluster.log || exit 1
done
