#!/bin/bash
# One GPU-box session: parity tests, headline bench, extra configs, rocprof stats, PMC passes.
# Usage (through gpurun): bash tools/gpu_validate.sh <tag> [notest]
set -o pipefail
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$2" != "notest" ]; then
  timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -4 $OUT/pytest_gpu.log
  if grep -q "Memory access fault" $OUT/pytest_gpu.log; then echo "GPU FAULT"; exit 1; fi
fi
line() { tail -1 $1 | cut -c1-260; }
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; line $OUT/bench_c2.json
timeout -k 10 300 python bench.py --config c5 --steps 20 --warmup 3 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; line $OUT/bench_c5.json
timeout -k 10 300 python bench.py --config c3 --steps 10 --warmup 3 --blocks 5 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err; line $OUT/bench_c3.json
# C4's per-GPU workload (8 graphs), the streaming regime (32 graphs: 491 MB per activation tensor), the reference's real workload
timeout -k 10 300 python bench.py --config c4 --steps 10 --warmup 3 --blocks 5 --no-cpu-baseline > $OUT/bench_c2_b8.json 2> $OUT/bench_c2_b8.err; line $OUT/bench_c2_b8.json
timeout -k 10 400 python bench.py --graphs-per-gpu 32 --steps 5 --warmup 2 --blocks 3 --no-cpu-baseline > $OUT/bench_c2_b32.json 2> $OUT/bench_c2_b32.err; line $OUT/bench_c2_b32.json
timeout -k 10 300 python bench.py --config real --steps 40 --warmup 8 > $OUT/bench_real.json 2> $OUT/bench_real.err; line $OUT/bench_real.json
timeout -k 10 300 python tools/measure_epoch_throughput.py --config c2 --epochs 2 > $OUT/epoch_throughput.jsonl 2> $OUT/epoch.err
timeout -k 10 300 python tools/measure_epoch_throughput.py --config real --epochs 2 >> $OUT/epoch_throughput.jsonl 2>> $OUT/epoch.err
cat $OUT/epoch_throughput.jsonl
# kernel-only durations of the same commands
prof() { timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1 -- python bench.py $2 --no-cpu-baseline > $OUT/$1.log 2> $OUT/$1.err; }
prof prof "--steps 20 --warmup 5 --blocks 2"
prof prof_c3 "--config c3 --steps 5 --warmup 2 --blocks 1"
prof prof_b8 "--config c4 --steps 5 --warmup 2 --blocks 1"
prof prof_b32 "--graphs-per-gpu 32 --steps 3 --warmup 1 --blocks 1"
prof prof_real "--config real --steps 20 --warmup 5 --blocks 1"
# HBM traffic of K1 / K2: separate FETCH_SIZE and WRITE_SIZE passes at 4, 8 and 32 graphs per GPU
pmc() { timeout -k 10 300 rocprofv3 --pmc $1 --output-format csv -d $OUT/$2 -- python bench.py $3 --no-cpu-baseline > $OUT/$2.log 2>&1; }
pmc FETCH_SIZE pmc_fetch "--steps 3 --warmup 1 --blocks 1"
pmc WRITE_SIZE pmc_write "--steps 3 --warmup 1 --blocks 1"
python tools/parse_pmc.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json | tail -12
pmc FETCH_SIZE pmc_fetch_b8 "--config c4 --steps 2 --warmup 1 --blocks 1"
pmc WRITE_SIZE pmc_write_b8 "--config c4 --steps 2 --warmup 1 --blocks 1"
python tools/parse_pmc.py $OUT/pmc_fetch_b8 $OUT/pmc_write_b8 $OUT/pmc_traffic_b8.json | tail -12
pmc FETCH_SIZE pmc_fetch_b32 "--graphs-per-gpu 32 --steps 2 --warmup 1 --blocks 1"
pmc WRITE_SIZE pmc_write_b32 "--graphs-per-gpu 32 --steps 2 --warmup 1 --blocks 1"
python tools/parse_pmc.py $OUT/pmc_fetch_b32 $OUT/pmc_write_b32 $OUT/pmc_traffic_b32.json | tail -12
pmc FETCH_SIZE pmc_fetch_c3 "--config c3 --steps 2 --warmup 1 --blocks 1"
pmc WRITE_SIZE pmc_write_c3 "--config c3 --steps 2 --warmup 1 --blocks 1"
python tools/parse_pmc.py $OUT/pmc_fetch_c3 $OUT/pmc_write_c3 $OUT/pmc_traffic_c3.json | tail -16
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_aux -- python tools/measure_aux_kernels.py > $OUT/aux_kernels.jsonl 2> $OUT/aux_kernels.err
rm -f $OUT/prof*/*/*kernel_trace.csv $OUT/pmc*/*/*kernel_trace.csv   # large; the stats files are what we keep
echo done
