"""Diagnostic: the panel GEMMs (K11 variant 10) by panel height and row count — single K = 256, pair K = 512 and the
chained forward launch (pair, then K = 256 on the rows just stored) — against the LDS-tile variants.
  python tools/diag/panel_rows_sweep.py [M ...]        (default 34992 60000)"""
import sys

sys.path.insert(0, "gnn-tumor-seg_amd")
import torch  # noqa: E402

from gts import _lib  # noqa: E402

lib = _lib.load()
F = 256
P = lambda t: t.data_ptr()  # noqa: E731
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=40):
    for _ in range(5):
        assert fn() == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


for M in [int(v) for v in sys.argv[1:]] or [34992, 60000]:
    x = torch.randn(M, F, device="cuda"); y = torch.randn(M, F, device="cuda")
    w = torch.randn(F, F, device="cuda") * 0.05; w2 = torch.randn(F, F, device="cuda") * 0.05; w3 = torch.randn(F, F, device="cuda") * 0.05
    b = torch.randn(F, device="cuda"); out = torch.empty(M, F, device="cuda"); out2 = torch.empty(M, F, device="cuda")
    bits = torch.empty(lib.gts_relu_bits_bytes(M, F) // 8 + 8, dtype=torch.int64, device="cuda")
    g1 = 2.0 * M * F * F
    for rows in (144, 192, 240):
        lib.gts_set_option(13, rows)
        lib.gts_set_option(1, 10)
        t1 = timeit(lambda: lib.gts_linear_fwd_f32(P(x), P(w), None, None, P(b), P(out), M, F, F, 0, 1, None, None, st))
        t2 = timeit(lambda: lib.gts_linear_fwd_f32(P(x), P(w), P(y), P(w2), P(b), P(out), M, F, F, F, 1, None, None, st))
        t3 = timeit(lambda: lib.gts_linear_fwd_chain_f32(P(x), P(w), P(y), P(w2), P(b), P(out), P(w3), P(b), P(out2), M, F, F, F, 1, F,
                                                         1, P(bits), None, st))
        wgs = (M + rows - 1) // rows
        print(f"M={M} panels of {rows} rows ({wgs} workgroups): K256 {t1:6.1f} us {g1 / t1 / 1e6:6.1f} TF | K512 {t2:6.1f} us "
              f"{2 * g1 / t2 / 1e6:6.1f} TF | chained {t3:6.1f} us {3 * g1 / t3 / 1e6:6.1f} TF | per 256 of K: "
              f"{t2 - t1:5.1f} us, fixed {2 * t1 - t2:5.1f} us", flush=True)
    lib.gts_set_option(13, 0)
    for v in (8, 3, 1):
        lib.gts_set_option(1, v)
        t1 = timeit(lambda: lib.gts_linear_fwd_f32(P(x), P(w), None, None, P(b), P(out), M, F, F, 0, 1, None, None, st))
        t2 = timeit(lambda: lib.gts_linear_fwd_f32(P(x), P(w), P(y), P(w2), P(b), P(out), M, F, F, F, 1, None, None, st))
        print(f"M={M} LDS tile variant {v}: K256 {t1:6.1f} us {g1 / t1 / 1e6:6.1f} TF | K512 {t2:6.1f} us {2 * g1 / t2 / 1e6:6.1f} TF", flush=True)
    lib.gts_set_option(1, -1)
