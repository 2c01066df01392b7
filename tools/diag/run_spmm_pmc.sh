#!/bin/bash
# Counters of K1 / K2 (tools/diag/spmm_loop.py under rocprofv3 --pmc).  Usage: bash tools/diag/run_spmm_pmc.sh <tag>
TAG=${1:-r02x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
run() { timeout -k 10 200 rocprofv3 --pmc $2 --output-format csv -d gpurun_out/$TAG/$1 -- python tools/diag/spmm_loop.py > gpurun_out/$TAG/$1.log 2>&1; }
run pmc_a "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"
run pmc_b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
run pmc_c "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"
run pmc_d "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
python - $TAG <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for d in ("pmc_a", "pmc_b", "pmc_c", "pmc_d"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{tag}/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "spmm_max" in row["Kernel_Name"]:
                acc[row["Kernel_Name"][31:62]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(d, k)
        for c, vals in sorted(v.items()):
            print(f"   {c:36s} n={len(vals):3d} mean={sum(vals)/len(vals):.4g}")
PY
