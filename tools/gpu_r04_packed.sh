#!/bin/bash
# Round 4: weights in fragment order: parity, isolated timing, the steps.
set -o pipefail
OUT=gpurun_out/${1:-r04f}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_stack.py tests/test_gpu_full_size.py tests/test_gpu_model.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest.log
grep -q " passed" $OUT/pytest.log || exit 1
line() { tail -1 $1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], d['blocks']['min'], d['blocks']['max'], d['config'].get('resident_batches'), 'K11', r['frac'], {k:(v['avg_launch_us'], v['tflops']) for k,v in r['by_kind'].items()}, [(h['kernel'], h['avg_launch_us'], h['frac']) for h in d['roofline_hbm']])"; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err; line $OUT/bench_c2.json
timeout -k 10 300 python bench.py --config real --steps 40 --warmup 8 --no-cpu-baseline > $OUT/bench_real.json 2> $OUT/bench_real.err; line $OUT/bench_real.json
timeout -k 10 300 python bench.py --config c4 --steps 10 --warmup 3 --blocks 5 --no-cpu-baseline > $OUT/bench_b8.json 2> $OUT/bench_b8.err; line $OUT/bench_b8.json
