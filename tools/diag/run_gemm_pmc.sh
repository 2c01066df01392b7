#!/bin/bash
# SQ counters of the forward GEMM variants (tools/tune_gemm.py <variants> nowgrad under rocprofv3 --pmc).
# Usage: bash tools/diag/run_gemm_pmc.sh <tag> <variants>
TAG=${1:-r02x}; V=${2:-8,10}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/$TAG/pmc_sq -- python tools/tune_gemm.py $V nowgrad > gpurun_out/$TAG/pmc_sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d gpurun_out/$TAG/pmc_sq2 -- python tools/tune_gemm.py $V nowgrad > gpurun_out/$TAG/pmc_sq2.log 2>&1
python - $TAG <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for d in ("pmc_sq", "pmc_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{tag}/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "gemm" in row["Kernel_Name"]:
                acc[row["Kernel_Name"][:100]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(k)
        for c, vals in sorted(v.items()):
            print(f"   {c:32s} n={len(vals):4d} mean={sum(vals)/len(vals):.4g}")
PY
