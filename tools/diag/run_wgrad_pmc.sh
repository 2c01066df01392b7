#!/bin/bash
# SQ counters of the 19-problem weight-gradient launch per tile variant (tools/tune_wgrad.py under rocprofv3 --pmc):
# how much of the matrix pipe is busy, and how many vector instructions sit beside the MFMAs.
# Usage: bash tools/diag/run_wgrad_pmc.sh <tag> <variants, e.g. 4,6>
TAG=${1:-r03x}; V=${2:-4,6}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$TAG/pmc_sq -- python tools/tune_wgrad.py $V > gpurun_out/$TAG/pmc_sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d gpurun_out/$TAG/pmc_sq2 -- python tools/tune_wgrad.py $V > gpurun_out/$TAG/pmc_sq2.log 2>&1
python - $TAG <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for d in ("pmc_sq", "pmc_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{tag}/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "wgrad_stream" in row["Kernel_Name"] or "gemm_kernel<256, 256, 4, 4, false, false, true" in row["Kernel_Name"]:
                acc[row["Kernel_Name"][:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(k)
        for c, vals in sorted(v.items()):
            print(f"   {c:32s} n={len(vals):4d} mean={sum(vals)/len(vals):.4g}")
PY
