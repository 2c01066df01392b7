"""Launch times of the panel kernel (variant 10) at the C2 layer shape under GTS_OPT_GEMM_SCHED settings
(argv[1] = comma-separated values; 1 = default, 3 = non-temporal stores, 5 = generic epilogue): forward single /
pair / pair with mask bits written, and the transposed input gradient without mask, with the float mask and with
the bit mask.  The round-2 what-if experiments (operand loads pinned to one reduction group = always L1 hits; no
epilogue; no mask read) patched the kernel temporarily; their results are in profiles/r02_panel_whatif.log."""
import sys

sys.path.insert(0, "gnn-tumor-seg_amd")
import torch  # noqa: E402

from gts import _lib  # noqa: E402

lib = _lib.load()
M, F = 60000, 256
x = torch.randn(M, F, device="cuda")
y = torch.randn(M, F, device="cuda")
w = torch.randn(F, F, device="cuda") * 0.05
w2 = torch.randn(F, F, device="cuda") * 0.05
b = torch.randn(F, device="cuda")
out = torch.empty(M, F, device="cuda")
bits = torch.empty(lib.gts_relu_bits_bytes(M, F) // 8, dtype=torch.int64, device="cuda")
P = lambda t: t.data_ptr()  # noqa: E731
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=30):
    for _ in range(3):
        assert fn() == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


lib.gts_set_option(1, 10)
lib.gts_set_option(3, 10)
for sched in [int(v) for v in sys.argv[1].split(",")]:
    lib.gts_set_option(7, sched)
    r = [timeit(lambda: lib.gts_linear_fwd_f32(P(x), P(w), None, None, P(b), P(out), M, F, F, 0, 1, None, None, st)),
         timeit(lambda: lib.gts_linear_fwd_f32(P(x), P(w), P(y), P(w2), P(b), P(out), M, F, F, F, 1, None, None, st)),
         timeit(lambda: lib.gts_linear_fwd_f32(P(x), P(w), P(y), P(w2), P(b), P(out), M, F, F, F, 1, P(bits), None, st)),
         timeit(lambda: lib.gts_linear_bwd_input_t_f32(P(x), P(w), P(y), P(w2), None, None, P(out), M, F, F, F, None, st)),
         timeit(lambda: lib.gts_linear_bwd_input_t_f32(P(x), P(w), P(y), P(w2), P(x), None, P(out), M, F, F, F, None, st)),
         timeit(lambda: lib.gts_linear_bwd_input_t_f32(P(x), P(w), P(y), P(w2), P(x), P(bits), P(out), M, F, F, F, None, st))]
    print(f"sched {sched:3d}: single {r[0]:7.1f} us | pair {r[1]:7.1f} | pair+bits out {r[2]:7.1f} | igrad pair: no mask {r[3]:7.1f}"
          f" float mask {r[4]:7.1f} bit mask {r[5]:7.1f}", flush=True)
