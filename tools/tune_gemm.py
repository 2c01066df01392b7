"""Time the K11 tile variants at the C2 layer shapes (M = 60 000, 256-wide) on one MI355X.
Raw C-ABI calls on preallocated buffers, so the GPU (not Python) sets the pace."""
import ctypes
import sys

sys.path.insert(0, "gnn-tumor-seg_amd")
import torch  # noqa: E402

from gts import _lib  # noqa: E402

lib = _lib.load()
M, F = 60000, 256
dev = "cuda"
x = torch.randn(M, F, device=dev)
y = torch.randn(M, F, device=dev)
w = torch.randn(F, F, device=dev) * 0.05
w2 = torch.randn(F, F, device=dev) * 0.05
b = torch.randn(F, device=dev)
out = torch.empty(M, F, device=dev)
gw = [torch.empty(F, F, device=dev) for _ in range(3)]
gb = [torch.empty(F, device=dev) for _ in range(3)]
ws = torch.empty(64 << 20, device=dev)  # 256 MB scratch
P = lambda t: t.data_ptr()  # noqa: E731
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, flops, reps=30):
    for _ in range(3):
        assert fn() == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    return us, flops / us / 1e6


g1 = 2.0 * M * F * F
VARIANTS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 10, 1, 8, 10]


def check(v):
    """Both GEMM kinds of variant v against torch (fp32 library GEMM): max error relative to the scale."""
    lib.gts_linear_fwd_f32(P(x), P(w), P(y), P(w2), P(b), P(out), M, F, F, F, 1, None, None, st)
    want = torch.relu(x @ w.t() + y @ w2.t() + b)
    e1 = float((out - want).abs().max() / want.abs().max())
    lib.gts_linear_bwd_input_f32(P(x), P(w), P(y), P(w2), P(x), P(out), M, F, F, F, st)
    want = (x @ w + y @ w2) * (x > 0)
    e2 = float((out - want).abs().max() / want.abs().max())
    assert e1 < 1e-5 and e2 < 1e-5, (v, e1, e2)
    return max(e1, e2)


for v in VARIANTS:
    lib.gts_set_option(1, v)
    lib.gts_set_option(3, v)
    err = check(v)
    r = [timeit(lambda: lib.gts_linear_fwd_f32(P(x), P(w), None, None, P(b), P(out), M, F, F, 0, 1, None, None, st), g1),
         timeit(lambda: lib.gts_linear_fwd_f32(P(x), P(w), P(y), P(w2), P(b), P(out), M, F, F, F, 1, None, None, st), 2 * g1),
         timeit(lambda: lib.gts_linear_bwd_input_f32(P(x), P(w), None, None, None, P(out), M, F, F, 0, st), g1),
         timeit(lambda: lib.gts_linear_bwd_input_f32(P(x), P(w), P(y), P(w2), P(x), P(out), M, F, F, F, st), 2 * g1)]
    print(f"fwd/igrad variant {v}: " + " | ".join(f"{us:7.1f} us {tf:6.1f} TF" for us, tf in r)
          + f" | max rel err {err:.1e}", flush=True)
lib.gts_set_option(1, -1)
lib.gts_set_option(3, 1)
if len(sys.argv) > 2 and sys.argv[2] == "nowgrad":
    sys.exit(0)
arr1, arr3 = ctypes.c_void_p * 1, ctypes.c_void_p * 3
for v in (1, 2, 4, 1, 2, 4):
    lib.gts_set_option(2, v)
    r = [timeit(lambda: lib.gts_linear_bwd_weight_f32(arr1(P(x)), arr1(P(y)), arr1(P(gw[0])), arr1(P(gb[0])), 1,
                                                      P(ws), ws.numel() * 4, M, F, F, st), g1),
         timeit(lambda: lib.gts_linear_bwd_weight_f32(arr3(P(x), P(x), P(y)), arr3(P(y), P(x), P(x)),
                                                      arr3(*[P(t) for t in gw]), arr3(P(gb[0]), None, P(gb[2])), 3,
                                                      P(ws), ws.numel() * 4, M, F, F, st), 3 * g1)]
    print(f"wgrad variant {v}: " + " | ".join(f"{us:7.1f} us {tf:6.1f} TF" for us, tf in r), flush=True)
lib.gts_set_option(2, -1)
t = timeit(lambda: torch.mm(x, w.t(), out=out) is None, g1)
print(f"hipBLASLt mm (reference point): {t[0]:7.1f} us {t[1]:6.1f} TF")
