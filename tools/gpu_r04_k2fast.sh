#!/bin/bash
# Round 4: K2's mask-free body for row pairs of equal degree: parity tests, then the K2 tuner and bench lines on the build before
# (tools/diag/build/libgts_hip_prev.so) and after, alternating in one session.
set -o pipefail
OUT=gpurun_out/${1:-r04y}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_kernels.py tests/test_gpu_stack.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $OUT/pytest.log
grep -q " passed" $OUT/pytest.log || exit 1
PREV=$GRAFT_REPO_ROOT/tools/diag/build/libgts_hip_prev.so
for lib in prev new prev new; do
  echo "== $lib"
  if [ $lib = prev ]; then export GTS_LIB_PATH=$PREV; else unset GTS_LIB_PATH; fi
  GTS_TUNE_QUICK=1 timeout -k 10 200 python tools/tune_k2_small.py 2>&1 | grep -v amdgpu.ids | grep '32,60,512' | tee -a $OUT/tune_$lib.log
done
for lib in prev new prev new; do
  if [ $lib = prev ]; then export GTS_LIB_PATH=$PREV; else unset GTS_LIB_PATH; fi
  for cfg in "" "--config real"; do
    timeout -k 10 300 python bench.py $cfg --steps 20 --warmup 5 --blocks 10 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
    python - $OUT/b.json "$lib $cfg" <<'PY' | tee -a $OUT/bench.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["ms_per_step"], d["blocks"]["min"], d["blocks"]["max"], [(h["kernel"], h["avg_launch_us"], h["frac"]) for h in d["roofline_hbm"]], d["config"].get("resident_batches", {}).get("value"))
PY
  done
done
