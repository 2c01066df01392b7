#!/bin/bash
# One GPU-box session: parity tests, headline bench, extra configs, rocprof stats, PMC passes.
# Usage (through gpurun): bash tools/gpu_validate.sh <tag> [notest]
set -o pipefail
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$2" != "notest" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -4 $OUT/pytest_gpu.log
  if grep -q "Memory access fault" $OUT/pytest_gpu.log; then echo "GPU FAULT"; exit 1; fi
fi
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
tail -1 $OUT/bench_c2.json | cut -c1-330
timeout -k 10 300 python bench.py --config c5 --steps 20 --warmup 3 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
tail -1 $OUT/bench_c5.json | cut -c1-330
timeout -k 10 300 python bench.py --config c3 --steps 10 --warmup 3 --blocks 5 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err
tail -1 $OUT/bench_c3.json | cut -c1-330
# the C4 per-GPU workload (8 graphs) and the streaming regime (32 graphs: 491 MB per activation tensor)
timeout -k 10 300 python bench.py --graphs-per-gpu 8 --steps 10 --warmup 3 --blocks 5 --no-cpu-baseline > $OUT/bench_c2_b8.json 2> $OUT/bench_c2_b8.err
tail -1 $OUT/bench_c2_b8.json | cut -c1-200
timeout -k 10 400 python bench.py --graphs-per-gpu 32 --steps 5 --warmup 2 --blocks 3 --no-cpu-baseline > $OUT/bench_c2_b32.json 2> $OUT/bench_c2_b32.err
tail -1 $OUT/bench_c2_b32.json | cut -c1-200
# kernel-only durations of the same headline command
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py --steps 20 --warmup 5 --blocks 2 --no-cpu-baseline > $OUT/bench_prof.log 2> $OUT/bench_prof.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -- python bench.py --config c3 --steps 5 --warmup 2 --blocks 1 --no-cpu-baseline > $OUT/bench_prof_c3.log 2> $OUT/bench_prof_c3.err
# HBM traffic of K1 / K2: separate FETCH_SIZE and WRITE_SIZE passes, C2 and the streaming regime
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
python tools/parse_pmc.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json | tail -12
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_b32 -- python bench.py --graphs-per-gpu 32 --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline > $OUT/pmc_fetch_b32.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_b32 -- python bench.py --graphs-per-gpu 32 --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline > $OUT/pmc_write_b32.log 2>&1
python tools/parse_pmc.py $OUT/pmc_fetch_b32 $OUT/pmc_write_b32 $OUT/pmc_traffic_b32.json | tail -12
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b32 -- python bench.py --graphs-per-gpu 32 --steps 3 --warmup 1 --blocks 1 --no-cpu-baseline > $OUT/bench_prof_b32.log 2> $OUT/bench_prof_b32.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_aux -- python tools/measure_aux_kernels.py > $OUT/aux_kernels.jsonl 2> $OUT/aux_kernels.err
rm -f $OUT/prof*/*/*kernel_trace.csv   # large; the stats file is what we keep
echo done
