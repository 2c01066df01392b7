#!/bin/bash
# K1 / K2 at a given --graphs-per-gpu (default 8 = the C4 per-GPU shape): rocprofv3 kernel-only means and the
# FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as MI355X_MICROARCH.md §HBM prescribes), plus the L2 hit /
# miss counters of the same launches.
# Usage (through gpurun): bash tools/gpu_b8_evidence.sh <tag> [graphs per gpu]
set -o pipefail
TAG=${1:-rXX}
B=${2:-8}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
FLAGS="--graphs-per-gpu $B --no-cpu-baseline"
timeout -k 10 300 python bench.py $FLAGS --steps 10 --warmup 3 --blocks 5 > $OUT/bench_c2_b$B.json 2> $OUT/bench_c2_b$B.err || { tail -20 $OUT/bench_c2_b$B.err; exit 1; }
tail -1 $OUT/bench_c2_b$B.json | cut -c1-200
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b$B -- python bench.py $FLAGS --steps 5 --warmup 2 --blocks 1 > $OUT/prof_b$B.log 2> $OUT/prof_b$B.err || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_b$B -- python bench.py $FLAGS --steps 2 --warmup 1 --blocks 1 > $OUT/pmc_fetch_b$B.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_b$B -- python bench.py $FLAGS --steps 2 --warmup 1 --blocks 1 > $OUT/pmc_write_b$B.log 2>&1 || exit 1
python tools/parse_pmc.py $OUT/pmc_fetch_b$B $OUT/pmc_write_b$B $OUT/pmc_traffic_b$B.json | tail -12
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2_b$B -- python bench.py $FLAGS --steps 2 --warmup 1 --blocks 1 > $OUT/pmc_l2_b$B.log 2>&1 || exit 1
python - $OUT/pmc_l2_b$B <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmm_max" in r["Kernel_Name"] or "spmm_cluster" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: sum(x) / len(x) for c, x in v.items()})
PY
rm -f $OUT/prof*/*/*kernel_trace.csv $OUT/pmc*/*/*kernel_trace.csv
python - $OUT/prof_b$B <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmm" in r["Name"]:
            print(r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / int(r["Calls"]) / 1e3, "us")
PY
echo done
