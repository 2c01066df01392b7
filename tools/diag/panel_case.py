"""Diagnostic: ONE panel-GEMM case in a loop, for counter collection (rocprofv3 --pmc ... -- python tools/diag/panel_case.py ...).
  panel_case.py <variant: 1441 | 1442 | 10> <M> <a_hot 0|1> [reps]      (K = 256 + 256 forward pair through tools/diag/gemm_probe.hip)"""
import ctypes
import os
import subprocess
import sys

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
lib.gts_probe_set_buffer(None)
variant, M, hot = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
F = 256
x = torch.randn(M, F, device="cuda"); y = torch.randn(M, F, device="cuda")
w = torch.randn(F, F, device="cuda") * 0.05; w2 = torch.randn(F, F, device="cuda") * 0.05
b = torch.randn(F, device="cuda"); out = torch.empty(M, F, device="cuda")
st = torch.cuda.current_stream().cuda_stream
args = (x.data_ptr(), w.data_ptr(), y.data_ptr(), w2.data_ptr(), b.data_ptr(), out.data_ptr(), M, F, F, F, 1, variant, hot, st)
for _ in range(reps):
    assert lib.gts_probe_linear_fwd(*args) == 0
torch.cuda.synchronize()
print("done", variant, M, hot)
