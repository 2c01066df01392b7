#!/bin/bash
# Round 4: what the clustered GAT aggregation fetches through the L2 (FETCH_SIZE per kernel) against the walk (clusters walked together
# through their slices; 100000 = the XCD's whole span, slice by slice), and the times of the three calls.
set -o pipefail
OUT=gpurun_out/${1:-r04r}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # label, args
  local label=$1; shift
  timeout -k 10 200 python tools/diag/gat_passes_ab.py "$@" > $OUT/t_$label.log 2>&1 || { tail -5 $OUT/t_$label.log; return 1; }
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f_$label -- python tools/diag/gat_passes_ab.py --reps 3 "$@" > $OUT/f_$label.log 2>&1
  python - $OUT $label <<'PY'
import csv, glob, collections, sys
out, label = sys.argv[1:3]
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/f_{label}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gat_cluster_stream" in k:
            acc[k[k.find("gat_"):][:31]].append(float(row["Counter_Value"]))
line = open(f"{out}/t_{label}.log").read().strip().splitlines()[-1]
print(label, {k: round(2 * 1024 * sum(v) / len(v) / 1e6, 1) for k, v in sorted(acc.items())}, "MB fetched |", line, flush=True)
PY
}
run g16 --group 16 && run g64 --group 64 && run g128 --group 128 && run span --group 100000 && run g32 --group 32 && run g16b --group 16
