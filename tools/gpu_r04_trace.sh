#!/bin/bash
# Round 4: per-launch kernel trace of one training step (C2 and the reference's real shape).
set -o pipefail
OUT=gpurun_out/${1:-r04tr}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in c2 real; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$cfg -- python bench.py --config $cfg --steps 3 --warmup 2 --blocks 1 --no-cpu-baseline > $OUT/trace_$cfg.log 2> $OUT/trace_$cfg.err || { tail -5 $OUT/trace_$cfg.err; exit 1; }
  python tools/step_trace.py $OUT/trace_$cfg $OUT/step_$cfg.md $([ $cfg = c2 ] && echo 3 || echo 0) > /dev/null
  rm -rf $OUT/trace_$cfg
done
tail -3 $OUT/step_c2.md; tail -3 $OUT/step_real.md
