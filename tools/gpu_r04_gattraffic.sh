#!/bin/bash
# Round 4: what the clustered GAT aggregation fetches through the L2 (FETCH_SIZE per kernel) against workgroups per CU, cluster size and walk.
set -o pipefail
OUT=gpurun_out/${1:-r04r}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # label, limits, args
  local label=$1 lim=$2; shift 2
  GTS_GAT_CLUSTER_LIMITS="$lim" timeout -k 10 200 python tools/diag/gat_passes_ab.py --only dense "$@" > $OUT/t_$label.log 2>&1 || { tail -5 $OUT/t_$label.log; return 1; }
  GTS_GAT_CLUSTER_LIMITS="$lim" timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f_$label -- python tools/diag/gat_passes_ab.py --only dense --reps 3 "$@" > $OUT/f_$label.log 2>&1
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
D="32,64,256;32,64,256;24,50,192"
run base "$D" --group 16 && run g0 "$D" --group 0 && run g0_cu1 "$D" --group 0 --per-cu 1 && run g16_cu1 "$D" --group 16 --per-cu 1 && \
run g0_small "16,36,128;16,36,128;12,28,96" --group 0 && run g0_small_cu1 "16,36,128;16,36,128;12,28,96" --group 0 --per-cu 1 && \
run g0_w8 "$D" --group 0 --waves 8 && run g4 "$D" --group 4 && run g1 "$D" --group 1
