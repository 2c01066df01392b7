#!/bin/bash
# Round 4: clustered K1 / K2, static against dealt units after the register fix, one session, kernel-only times from rocprofv3.
set -o pipefail
OUT=gpurun_out/${1:-r04k14}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_stack.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $OUT/pytest.log
grep -q " passed" $OUT/pytest.log || exit 1
for opt in "18=0" "18=1" "18=0" "18=1"; do
  for cfg in "--steps 20 --warmup 5 --blocks 10" "--config c4 --steps 10 --warmup 3 --blocks 5" "--config real --steps 40 --warmup 8 --blocks 10"; do
    GTS_OPTIONS="$opt" timeout -k 10 300 python bench.py $cfg --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
    python - $OUT/b.json "$opt $cfg" <<'PY' | tee -a $OUT/bench.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "|", d["value"], d["ms_per_step"], [(h["kernel"], h["avg_launch_us"], h["frac"]) for h in d["roofline_hbm"]])
PY
  done
done
for opt in "18=1" "18=0"; do
  for name in c2 b8; do
    if [ $name = c2 ]; then cfg="--steps 10 --warmup 3 --blocks 2"; else cfg="--config c4 --steps 5 --warmup 2 --blocks 1"; fi
    GTS_OPTIONS="$opt" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${name}_$opt -- python bench.py $cfg --no-cpu-baseline > $OUT/prof.log 2>&1
    python - $OUT/prof_${name}_$opt "$opt $name" <<'PY' | tee -a $OUT/kernels.log
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmm_cluster_stream" in r["Name"] or "spmm_max_fwd_kernel<4, 64" in r["Name"]:
            print(sys.argv[2], r["Name"][31:95], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2))
PY
  done
done
rm -f $OUT/prof*/*/*kernel_trace.csv
