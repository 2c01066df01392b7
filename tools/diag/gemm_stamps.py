"""Diagnostic: per-workgroup phase timeline of the K11 forward GEMM at the C2 layer shape.
Builds tools/diag/gemm_probe.hip (the library's kernels with a stamping probe; never shipped) and prints when workgroups start,
how long prologue / main loop / epilogue take and when the last one ends.
  python tools/diag/gemm_stamps.py      (on the GPU box)"""
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
lib.gts_probe_linear_fwd.argtypes = [p, p, p, p, p, p, i64, i64, i64, i64, i32, i32, i32, p]
lib.gts_probe_set_buffer.argtypes = [p]
M, F = 60000, 256
x = torch.randn(M, F, device="cuda"); y = torch.randn(M, F, device="cuda")
w = torch.randn(F, F, device="cuda") * 0.05; w2 = torch.randn(F, F, device="cuda") * 0.05
b = torch.randn(F, device="cuda"); out = torch.empty(M, F, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for variant, bm, bn in ((9, 240, 256), (11, 240, 256), (12, 240, 256)):
    n_blocks = ((M + bm - 1) // bm) * ((F + bn - 1) // bn)
    stamps = torch.zeros(8 * n_blocks, dtype=torch.int64, device="cuda")
    for dual in (False, True):
        args = (x.data_ptr(), w.data_ptr(), y.data_ptr() if dual else None, w2.data_ptr() if dual else None,
                b.data_ptr(), out.data_ptr(), M, F, F, F if dual else 0, 1, variant, 0, st)
        lib.gts_probe_set_buffer(None)
        for _ in range(200):   # bring the chip to its loaded clock before the stamped launch
            lib.gts_probe_linear_fwd(*args)
        stamps.zero_()
        lib.gts_probe_set_buffer(stamps.data_ptr())
        lib.gts_probe_linear_fwd(*args)
        torch.cuda.synchronize()
        raw = stamps.cpu().numpy().reshape(n_blocks, 4, 2).astype(np.float64)
        t = raw[:, :, 0] * 0.01                      # 100 MHz -> us
        ghz = (raw[:, 2, 1] - raw[:, 1, 1]) / np.maximum(raw[:, 2, 0] - raw[:, 1, 0], 1) * 0.1   # main loop only
        t0 = t[:, 0].min()
        start, pro, main, epi = t[:, 0] - t0, t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
        end = t[:, 3] - t0
        q = lambda a: f"p10 {np.percentile(a, 10):6.1f} med {np.median(a):6.1f} p90 {np.percentile(a, 90):6.1f} max {a.max():6.1f}"  # noqa: E731
        first_round = start < 2.0
        print(f"variant {variant} ({bm}x{bn}) {'pair' if dual else 'single'}: {n_blocks} workgroups, kernel span {end.max():6.1f} us")
        print(f"   start   {q(start)}   ({first_round.sum()} start within 2 us)")
        print(f"   prologue {q(pro)}\n   mainloop {q(main)}\n   epilogue {q(epi)}")
        print(f"   end     {q(end)}")
        print(f"   in-kernel clock over the main loop: median {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz, 10):.3f}, "
              f"p90 {np.percentile(ghz, 90):.3f})", flush=True)

# A operand cache-hot (row stride 0): is the main loop waiting on A from HBM?
lib.gts_probe_set_buffer(None)
for variant in (9, 10, 11, 12):
    for hot in (0, 1):
        for dual in (False, True):
            args = (x.data_ptr(), w.data_ptr(), y.data_ptr() if dual else None, w2.data_ptr() if dual else None,
                    b.data_ptr(), out.data_ptr(), M, F, F, F if dual else 0, 1, variant, hot, st)
            for _ in range(3):
                lib.gts_probe_linear_fwd(*args)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                lib.gts_probe_linear_fwd(*args)
            e.record()
            torch.cuda.synchronize()
            print(f"variant {variant} A-hot={hot} {'pair' if dual else 'single'}: {s.elapsed_time(e) * 50:.1f} us", flush=True)
