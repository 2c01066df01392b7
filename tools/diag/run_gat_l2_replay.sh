#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r04l2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python tools/diag/gat_l2_replay.py > $OUT/replay.log 2>&1 || { tail $OUT/replay.log; exit 1; }
python - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(f"{out}/fetch/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gat_cluster_stream_kernel<0, 6, 1," in row["Kernel_Name"]:
            rows.append((int(row["Dispatch_Id"]), int(row["Grid_Size"]) // int(row["Workgroup_Size"]), float(row["Counter_Value"])))
rows.sort()
cases = [l.strip()[5:] for l in open(f"{out}/replay.log") if l.startswith("CASE ")]
rows = rows[-2 * len(cases):]          # the script's own launches (two per case), after ops._gat_fwd's
for i, case in enumerate(cases):
    v = (rows[2 * i][2] + rows[2 * i + 1][2]) / 2
    print(f"{case}: fetched {2 * 1024 * v / 1e6:.1f} MB = {2 * 1024 * v / (60000 * 4096):.3f} x the table")
PY
