#!/bin/bash
# L1 / L2 / SQ counters of the forward pair GEMM (K = 512): 144-row panels at 34 992 rows, cold and "A-hot", prefetch depth 1 / 2,
# next to 240-row panels at 60 000 rows.  Usage: bash tools/diag/run_panel144_pmc.sh <tag>
TAG=${1:-r04pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
python tools/diag/panel_case.py 1441 34992 0 3 > /dev/null 2>&1     # builds the probe library once
i=0
for case in "1441 34992 0" "1441 34992 1" "1442 34992 0" "10 60000 0" "10 60000 1"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum --output-format csv -d gpurun_out/$TAG/a$i -- python tools/diag/panel_case.py $case > gpurun_out/$TAG/a$i.log 2>&1
  timeout -k 10 120 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$TAG/b$i -- python tools/diag/panel_case.py $case > gpurun_out/$TAG/b$i.log 2>&1
  timeout -k 10 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/$TAG/c$i -- python tools/diag/panel_case.py $case > gpurun_out/$TAG/c$i.log 2>&1
  timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/$TAG/d$i -- python tools/diag/panel_case.py $case > gpurun_out/$TAG/d$i.log 2>&1
  python - $TAG $i "$case" <<'PY'
import csv, glob, collections, sys
tag, i, case = sys.argv[1], sys.argv[2], sys.argv[3]
print("== case", case)
for d in "abcd":
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/{tag}/{d}{i}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "panel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for c, vals in sorted(acc.items()):
        print(f"   {c:34s} n={len(vals):3d} mean={sum(vals)/len(vals):.5g}")
PY
  rm -rf gpurun_out/$TAG/?$i
done
