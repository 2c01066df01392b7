cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02e
timeout -k 10 400 python tools/diag/gemm_stamps.py > gpurun_out/r02e/stamps.log 2>&1; tail -40 gpurun_out/r02e/stamps.log
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/r02e/pmc_sq -- python tools/tune_gemm.py 8,9 nowgrad > gpurun_out/r02e/pmc_sq.log 2>&1; tail -3 gpurun_out/r02e/pmc_sq.log
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r02e/pmc_sq2 -- python tools/tune_gemm.py 8,9 nowgrad > gpurun_out/r02e/pmc_sq2.log 2>&1; tail -3 gpurun_out/r02e/pmc_sq2.log
python - <<'PY'
import csv, glob, collections
for d in ("pmc_sq", "pmc_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/r02e/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "gemm_kernel" in row["Kernel_Name"]:
                acc[row["Kernel_Name"][:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(k)
        for c, vals in sorted(v.items()):
            print(f"   {c:32s} n={len(vals):4d} mean={sum(vals)/len(vals):.4g}")
PY
