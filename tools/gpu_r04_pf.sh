#!/bin/bash
# Round 4: 144-row panels with the prefetch wave: parity, isolated timing by prefetch distance.
set -o pipefail
OUT=gpurun_out/${1:-r04e}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_gemm.py -m gpu -q -k "144_row" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "differs|passed|failed" $OUT/pytest.log | cut -c1-200 | head
for sched in 17 1 33 49 17 33; do
  GTS_OPTIONS="7=$sched" timeout -k 10 200 python tools/diag/panel_rows_sweep.py 34992 > $OUT/sweep_$sched.log 2>&1; echo "sched $sched: $(grep '144 rows' $OUT/sweep_$sched.log)"
done
