#!/bin/bash
# Round 4: clustered K1 / K2 with their units dealt off a counter per XCD: parity, bench lines at 4 / 8 / 32 graphs per GPU against static
# dealing (GTS_OPTIONS=18=1), FETCH_SIZE / WRITE_SIZE at 32 graphs.
set -o pipefail
OUT=gpurun_out/${1:-r04k12}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_stack.py tests/test_gpu_kernels.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $OUT/pytest.log
grep -q " passed" $OUT/pytest.log || exit 1
for opt in "18=0" "18=1" "18=0" "18=1"; do
  for cfg in "--steps 20 --warmup 5 --blocks 10" "--config c4 --steps 10 --warmup 3 --blocks 5" "--graphs-per-gpu 32 --steps 5 --warmup 2 --blocks 3"; do
    GTS_OPTIONS="$opt" timeout -k 10 300 python bench.py $cfg --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
    python - $OUT/b.json "$opt $cfg" <<'PY' | tee -a $OUT/bench.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "|", d["value"], d["ms_per_step"], [(h["kernel"], h["avg_launch_us"], h["frac"]) for h in d["roofline_hbm"]])
PY
  done
done
pmc() { GTS_OPTIONS="$4" timeout -k 10 300 rocprofv3 --pmc $1 --output-format csv -d $OUT/$2 -- python bench.py $3 --no-cpu-baseline > $OUT/$2.log 2>&1; }
for opt in "18=1" "18=0"; do
  pmc FETCH_SIZE pmc_fetch_b32_$opt "--graphs-per-gpu 32 --steps 2 --warmup 1 --blocks 1" $opt
  pmc WRITE_SIZE pmc_write_b32_$opt "--graphs-per-gpu 32 --steps 2 --warmup 1 --blocks 1" $opt
  python tools/parse_pmc.py $OUT/pmc_fetch_b32_$opt $OUT/pmc_write_b32_$opt $OUT/pmc_traffic_b32_$opt.json | grep bytes_per_launch
  pmc FETCH_SIZE pmc_fetch_b8_$opt "--config c4 --steps 2 --warmup 1 --blocks 1" $opt
  pmc WRITE_SIZE pmc_write_b8_$opt "--config c4 --steps 2 --warmup 1 --blocks 1" $opt
  python tools/parse_pmc.py $OUT/pmc_fetch_b8_$opt $OUT/pmc_write_b8_$opt $OUT/pmc_traffic_b8_$opt.json | grep bytes_per_launch
done
