"""What the epilogue arithmetic of the panel kernel costs: plain product vs bias + ReLU, same shape (M = 60 000, 256 -> 256)."""
import sys
sys.path.insert(0, "gnn-tumor-seg_amd")
import torch
from gts import _lib
lib = _lib.load()
M, F = 60000, 256
x = torch.randn(M, F, device="cuda"); w = torch.randn(F, F, device="cuda") * 0.05; b = torch.randn(F, device="cuda")
y = torch.randn(M, F, device="cuda"); w2 = torch.randn(F, F, device="cuda") * 0.05
out = torch.empty(M, F, device="cuda")
P = lambda t: t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=40):
    for _ in range(5): assert fn() == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
for rep in range(2):
    for name, fn in (("single plain      ", lambda: lib.gts_linear_fwd_f32(P(x), P(w), None, None, None, P(out), M, F, F, 0, 0, None, None, st)),
                     ("single bias + relu", lambda: lib.gts_linear_fwd_f32(P(x), P(w), None, None, P(b), P(out), M, F, F, 0, 1, None, None, st)),
                     ("pair plain        ", lambda: lib.gts_linear_fwd_f32(P(x), P(w), P(y), P(w2), None, P(out), M, F, F, F, 0, None, None, st)),
                     ("pair bias + relu  ", lambda: lib.gts_linear_fwd_f32(P(x), P(w), P(y), P(w2), P(b), P(out), M, F, F, F, 1, None, None, st))):
        print(name, f"{timeit(fn):7.1f} us", flush=True)
