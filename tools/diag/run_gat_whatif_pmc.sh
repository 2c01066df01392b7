#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the what-if forms of the clustered GAT forward kernel (tools/diag/gat_whatif.py): does the kernel fetch less
# through the L2 when it stores nothing?  Usage: bash tools/diag/run_gat_whatif_pmc.sh <tag>
set -o pipefail
OUT=gpurun_out/${1:-r04z}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$(echo $c | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/$d -- python tools/diag/gat_whatif.py 4 > $OUT/$d.log 2>&1
done
python - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for d in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum_TCC_MISS_sum"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "gat_cluster_stream" in k:
                acc[k[k.find("gat_cluster_stream_kernel"):][:44]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(d, k, {c: (len(vals), round(sum(vals) / len(vals), 1)) for c, vals in sorted(v.items())})
PY
