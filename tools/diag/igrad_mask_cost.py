"""Diagnostic: what does the fused ReLU' mask cost in the pair input-gradient GEMM epilogue
(fp32 mask = the previous layer's output vs the 1-bit-per-element mask a forward launch wrote)?"""
import sys

sys.path.insert(0, "gnn-tumor-seg_amd")
import torch  # noqa: E402

from gts import _lib  # noqa: E402

lib = _lib.load()
M, F = 60000, 256
x = torch.randn(M, F, device="cuda"); y = torch.randn(M, F, device="cuda")
w = torch.randn(F, F, device="cuda") * 0.05; w2 = torch.randn(F, F, device="cuda") * 0.05
out = torch.empty(M, F, device="cuda"); mask = torch.randn(M, F, device="cuda")
bits = torch.zeros(M, F // 32, dtype=torch.int32, device="cuda")
P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


for rep in range(3):
    a = timeit(lambda: lib.gts_linear_bwd_input_f32(P(x), P(w), P(y), P(w2), None, None, P(out), M, F, F, F, st))
    b = timeit(lambda: lib.gts_linear_bwd_input_f32(P(x), P(w), P(y), P(w2), P(mask), None, P(out), M, F, F, F, st))
    d = timeit(lambda: lib.gts_linear_bwd_input_f32(P(x), P(w), P(y), P(w2), None, P(bits), P(out), M, F, F, F, st))
    c = timeit(lambda: lib.gts_linear_bwd_input_f32(P(x), P(w), None, None, None, None, P(out), M, F, F, 0, st))
    f0 = timeit(lambda: lib.gts_linear_fwd_f32(P(x), P(w), P(y), P(w2), None, P(out), M, F, F, F, 1, None, st))
    f1 = timeit(lambda: lib.gts_linear_fwd_f32(P(x), P(w), P(y), P(w2), None, P(out), M, F, F, F, 1, P(bits), st))
    print(f"pair igrad: no mask {a:6.1f} us | fp32 mask {b:6.1f} us | bit mask {d:6.1f} us | single {c:6.1f} us"
          f" || pair forward: {f0:6.1f} us | writing the bit mask {f1:6.1f} us", flush=True)
