#!/bin/bash
# Round 4: clustered GAT aggregation with the gathers 1 - 3 units ahead: bit-exactness, what-if timings, A/B of the three passes.
set -o pipefail
OUT=gpurun_out/${1:-r04s}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for opt in "17=1" "17=2" "17=3,11=1,15=16" "17=3,11=1,15=12"; do
  GTS_OPTIONS="$opt" timeout -k 10 300 python -m pytest tests/test_gpu_gat_cluster.py -m gpu -x -q > $OUT/pytest_$opt.log 2>&1
  echo "pytest ($opt) rc=$?"; tail -1 $OUT/pytest_$opt.log
  grep -q " passed" $OUT/pytest_$opt.log || exit 1
done
timeout -k 10 300 python tools/diag/gat_whatif.py 4 > $OUT/whatif.log 2>&1 || { tail -20 $OUT/whatif.log; exit 1; }
cat $OUT/whatif.log
for args in "--depth 1" "--depth 2" "--depth 2 --per-cu 1 --waves 16" "--depth 3 --per-cu 1 --waves 16" "--depth 3 --per-cu 1 --waves 12" "--depth 1"; do
  echo "== $args"
  timeout -k 10 200 python tools/diag/gat_passes_ab.py --only dense --group 16 $args 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.log
done
