#!/bin/bash
# L2 miss merging probe (tools/diag/l2_merge_probe.hip, built here by hipcc into tools/diag/build).  Usage: bash tools/diag/run_l2_merge_probe.sh <tag>
set -o pipefail
OUT=gpurun_out/${1:-r04q}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 0 1 2; do
  timeout -k 10 120 tools/diag/build/l2_merge_probe $m | tee -a $OUT/probe.log
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$m -- tools/diag/build/l2_merge_probe $m > $OUT/fetch_$m.log 2>&1
  timeout -k 10 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/tcc_$m -- tools/diag/build/l2_merge_probe $m > $OUT/tcc_$m.log 2>&1
done
python - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for m in (0, 1, 2):
    for d in (f"fetch_{m}", f"tcc_{m}"):
        acc = collections.defaultdict(list)
        for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "probe" in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(d, {c: [round(v, 1) for v in vals] for c, vals in sorted(acc.items())})
PY
