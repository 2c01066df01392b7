"""Diagnostic: per-WAVE timeline of the direct-to-fragment forward GEMM (variant 10) at the C2 layer shape:
when each of a workgroup's twelve waves finishes its reduction and its stores, by age rank on its SIMD."""
import ctypes
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "gnn-tumor-seg_amd"))
import torch  # noqa: E402

so = "/tmp/libgts_probe.so"
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off",
                           f"-I{REPO}/include", f"-I{REPO}/gnn-tumor-seg_amd/csrc", "-o", so,
                           os.path.join(REPO, "tools/diag/gemm_probe.hip"),
                           os.path.join(REPO, "gnn-tumor-seg_amd/csrc/gts_project.hip"),
                           os.path.join(REPO, "gnn-tumor-seg_amd/csrc/gts_gat.hip"),
                           os.path.join(REPO, "gnn-tumor-seg_amd/csrc/gts_gat_reduce.hip")])
lib = ctypes.CDLL(so)
p, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
lib.gts_probe_linear_fwd.argtypes = [p, p, p, p, p, p, i64, i64, i64, i64, i32, i32, i32, p]
lib.gts_probe_set_buffer.argtypes = [p]
M, F = 60000, 256
x = torch.randn(M, F, device="cuda"); y = torch.randn(M, F, device="cuda")
w = torch.randn(F, F, device="cuda") * 0.05; w2 = torch.randn(F, F, device="cuda") * 0.05
b = torch.randn(F, device="cuda"); out = torch.empty(M, F, device="cuda")
st = torch.cuda.current_stream().cuda_stream
n_wg, waves = 250, 12
for dual in (False, True):
    for hot in (0, 1):
        args = (x.data_ptr(), w.data_ptr(), y.data_ptr() if dual else None, w2.data_ptr() if dual else None,
                b.data_ptr(), out.data_ptr(), M, F, F, F if dual else 0, 1, 110, hot, st)
        lib.gts_probe_set_buffer(None)
        for _ in range(100):
            lib.gts_probe_linear_fwd(*args)
        stamps = torch.zeros(n_wg * waves * 8, dtype=torch.int64, device="cuda")
        lib.gts_probe_set_buffer(stamps.data_ptr())
        lib.gts_probe_linear_fwd(*args)
        torch.cuda.synchronize()
        raw = stamps.cpu().numpy().reshape(n_wg, waves, 4, 2)
        t = raw[..., 0].astype(np.float64) * 0.01
        where = raw[:, 0, 0, 1]
        xcc, se, cu = (where >> 32) & 15, (where >> 13) & 7, (where >> 8) & 15
        t0 = t[:, :, 0].min()
        start, main_end, end = t[:, :, 0] - t0, t[:, :, 2] - t0, t[:, :, 3] - t0
        print(f"{'pair' if dual else 'single'} A-hot={hot}: kernel span {end.max():.1f} us; wave start median {np.median(start):.1f}")
        # rank the three waves of each SIMD (same wave & 3) by the time their reduction ends
        for simd in range(1):
            ids = [wv for wv in range(waves) if wv % 4 == simd]
            order = np.sort(main_end[:, ids], axis=1)
            stores = np.sort((end - main_end)[:, ids], axis=1)
            print("   reduction ends (1st / 2nd / 3rd wave of a SIMD), median over workgroups: "
                  + " / ".join(f"{np.median(order[:, r]):.1f}" for r in range(3))
                  + f" us;  store phase median {np.median(end - main_end):.1f} us, p90 {np.percentile(end - main_end, 90):.1f}")
        print(f"   last reduction end per workgroup: median {np.median(main_end.max(1)):.1f}, max {main_end.max():.1f};"
              f" last store end per workgroup: median {np.median(end.max(1)):.1f}, max {end.max():.1f}", flush=True)
        wg_end, wg_start = end.max(1), start.min(1)
        print("   workgroup end percentiles 10/50/90/99/100: " + " ".join(f"{np.percentile(wg_end, q):.1f}" for q in (10, 50, 90, 99, 100))
              + "; start percentiles 50/90/100: " + " ".join(f"{np.percentile(wg_start, q):.1f}" for q in (50, 90, 100))
              + f"; corr(start, end) {np.corrcoef(wg_start, wg_end)[0, 1]:.2f}")
        print("   per XCD (n, median end, max end): " + " ".join(
            f"{x}:({int((xcc == x).sum())},{np.median(wg_end[xcc == x]):.1f},{wg_end[xcc == x].max():.1f})" for x in sorted(set(xcc.tolist()))))
        cuid = xcc * 1000 + se * 16 + cu
        print(f"   distinct (xcc, se, cu) ids {len(set(cuid.tolist()))} for {n_wg} workgroups", flush=True)
