#!/bin/bash
# Round 4: the clustered GAT passes after the round's changes (memory-bound, dealt units): waves per workgroup, smaller clusters with three workgroups per CU.
set -o pipefail
OUT=gpurun_out/${1:-r04gs}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/diag/gat_passes_ab.py --group 16 $ARGS 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep.log; }
ARGS="" run X=1
ARGS="--waves 16" run X=1
ARGS="--waves 8" run X=1
ARGS="--waves 8 --per-cu 3" run "GTS_GAT_CLUSTER_LIMITS=20,44,160;20,44,160;16,34,128"
ARGS="--waves 12" run "GTS_GAT_CLUSTER_LIMITS=24,52,192;24,52,192;24,50,192"
ARGS="" run "GTS_GAT_CLUSTER_LIMITS=32,64,256;32,64,256;32,60,256"
ARGS="" run X=1
