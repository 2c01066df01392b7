#!/bin/bash
# Round 4: host collate on the GPU box (real-shape bench, epoch throughput at 48 steps per epoch).
set -o pipefail
OUT=gpurun_out/${1:-r04b}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
line() { tail -1 $1 | cut -c1-300; }
timeout -k 10 300 python bench.py --config real --steps 40 --warmup 8 --no-cpu-baseline > $OUT/bench_real.json 2> $OUT/bench_real.err; line $OUT/bench_real.json
grep "block seconds" $OUT/bench_real.err
GTS_PREFETCH_TRACE=1 timeout -k 10 300 python tools/measure_epoch_throughput.py --config real --epochs 2 > $OUT/epoch_real.jsonl 2> $OUT/epoch_real.err
cat $OUT/epoch_real.jsonl; grep "gts prefetch" $OUT/epoch_real.err | tail -4
GTS_SWITCH_INTERVAL=0.0002 timeout -k 10 300 python tools/measure_epoch_throughput.py --config real --epochs 2 > $OUT/epoch_real_si.jsonl 2> $OUT/epoch_real_si.err
cat $OUT/epoch_real_si.jsonl
timeout -k 10 300 python tools/measure_epoch_throughput.py --config c2 --epochs 2 > $OUT/epoch_c2.jsonl 2> $OUT/epoch_c2.err
cat $OUT/epoch_c2.jsonl
echo done
