"""Diagnostic: is the 144-row panel GEMM (the reference's batch shape, 35 000 rows) waiting for its activations?
K = 256 / K = 512 at M = 34 992 with the fragments 1 / 2 / 3 reduction groups ahead, activations from HBM / the Infinity
Cache (hot = 0) or with the activation rows (hot & 1) / the weight rows (hot & 2) of a 16-row fragment all the SAME row
(row stride 0): a fragment load then touches one 64-byte piece instead of sixteen rows — what a coalesced (packed) operand
layout would cost the vector L1."""
import ctypes
import os
import subprocess
import sys

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
lib.gts_probe_set_buffer(None)
F = 256
st = torch.cuda.current_stream().cuda_stream
for M in (34992, 60000):
    x = torch.randn(M, F, device="cuda"); y = torch.randn(M, F, device="cuda")
    w = torch.randn(F, F, device="cuda") * 0.05; w2 = torch.randn(F, F, device="cuda") * 0.05
    b = torch.randn(F, device="cuda"); out = torch.empty(M, F, device="cuda")
    for variant in ((1441, 1442, 1443) if M < 40000 else (10,)):
        for hot in (0, 1, 2, 3):
            row = []
            for dual in (False, True):
                args = (x.data_ptr(), w.data_ptr(), y.data_ptr() if dual else None, w2.data_ptr() if dual else None,
                        b.data_ptr(), out.data_ptr(), M, F, F, F if dual else 0, 1, variant, hot, st)
                for _ in range(5):
                    assert lib.gts_probe_linear_fwd(*args) == 0
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(40):
                    lib.gts_probe_linear_fwd(*args)
                e.record()
                torch.cuda.synchronize()
                row.append(s.elapsed_time(e) * 25)
            print(f"M={M} variant {variant} hot={hot}: K256 {row[0]:6.1f} us | K512 {row[1]:6.1f} us | per 256 of K {row[1] - row[0]:5.1f} us", flush=True)
