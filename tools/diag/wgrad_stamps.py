"""Diagnostic: phase timeline and in-kernel clock of the 19-problem weight-gradient launch of C2
(13 splits x 19 output tiles = 247 workgroups, 144 reduction tiles each)."""
import ctypes
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "gnn-tumor-seg_amd"))
import torch  # noqa: E402

so = "/tmp/libgts_probe.so"
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off",
                       f"-I{REPO}/include", f"-I{REPO}/gnn-tumor-seg_amd/csrc", "-o", so,
                       os.path.join(REPO, "tools/diag/gemm_probe.hip"),
                       os.path.join(REPO, "gnn-tumor-seg_amd/csrc/gts_project.hip"),
                           os.path.join(REPO, "gnn-tumor-seg_amd/csrc/gts_gat.hip"),
                           os.path.join(REPO, "gnn-tumor-seg_amd/csrc/gts_gat_reduce.hip")])
lib = ctypes.CDLL(so)
p, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
lib.gts_probe_wgrad.argtypes = [p, p, i32, p, i64, i64, i64, i32, i32, p]
lib.gts_probe_set_buffer.argtypes = [p]
M, F, Q, SPLITS = 60000, 256, 19, 13
gs = [torch.randn(M, F, device="cuda") for _ in range(Q)]
acts = [torch.randn(M, F, device="cuda") for _ in range(Q)]
ws = torch.empty(Q * SPLITS * F * F, device="cuda")
arr = ctypes.c_void_p * Q
st = torch.cuda.current_stream().cuda_stream
n_wg = Q * SPLITS
for variant in (4, 7, 71, 72):
    args = (arr(*[t.data_ptr() for t in gs]), arr(*[t.data_ptr() for t in acts]), Q, ws.data_ptr(), M, F, F, variant, SPLITS, st)
    lib.gts_probe_set_buffer(None)
    for _ in range(20):
        lib.gts_probe_wgrad(*args)
    stamps = torch.zeros(8 * n_wg, dtype=torch.int64, device="cuda")
    lib.gts_probe_set_buffer(stamps.data_ptr())
    lib.gts_probe_wgrad(*args)
    torch.cuda.synchronize()
    raw = stamps.cpu().numpy().reshape(n_wg, 4, 2).astype(np.float64)
    t = raw[:, :, 0] * 0.01
    ghz = (raw[:, 2, 1] - raw[:, 1, 1]) / np.maximum(raw[:, 2, 0] - raw[:, 1, 0], 1) * 0.1
    t0 = t[:, 0].min()
    main, epi, end = t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t0
    cycles = (raw[:, 2, 1] - raw[:, 1, 1])
    ideal = 145 * 16 * 4 * 64          # tiles x steps x MFMAs per wave and step x 64 cycles, 4 waves per SIMD -> per SIMD
    print(f"variant {variant}: kernel span {end.max():.1f} us; main loop median {np.median(main):.1f} us "
          f"(p10 {np.percentile(main, 10):.1f}, p90 {np.percentile(main, 90):.1f}), epilogue {np.median(epi):.1f} us; "
          f"in-kernel clock median {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}); "
          f"main loop {np.median(cycles):.0f} shader cycles vs {ideal * 4} MFMA cycles per SIMD = "
          f"{ideal * 4 / np.median(cycles):.3f} of the matrix pipe", flush=True)
