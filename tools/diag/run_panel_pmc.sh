#!/bin/bash
# L1 / L2 counters of the panel kernel.  Usage: bash tools/diag/run_panel_pmc.sh <tag> [sched]
TAG=${1:-r02x}; S=${2:-1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
timeout -k 10 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum --output-format csv -d gpurun_out/$TAG/pmc_a -- python tools/diag/panel_whatif.py $S > gpurun_out/$TAG/pmc_a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$TAG/pmc_b -- python tools/diag/panel_whatif.py $S > gpurun_out/$TAG/pmc_b.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/$TAG/pmc_c -- python tools/diag/panel_whatif.py $S > gpurun_out/$TAG/pmc_c.log 2>&1
python - $TAG <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for d in ("pmc_a", "pmc_b", "pmc_c"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{tag}/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "panel" in row["Kernel_Name"]:
                acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(d, k)
        for c, vals in sorted(v.items()):
            # launches alternate single (3+30), pair (33), igrad pair (33): print the three groups
            n = len(vals) // 3
            print(f"   {c:32s} n={len(vals):4d} single={sum(vals[:n])/n:.4g} pair={sum(vals[n:2*n])/n:.4g} igrad={sum(vals[2*n:])/max(1,len(vals)-2*n):.4g}")
PY
