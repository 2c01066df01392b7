import sys
sys.path.insert(0, "gnn-tumor-seg_amd")
import torch
from gts import _lib
lib = _lib.load()
M, F = 60000, 256
x = torch.randn(M, F, device="cuda"); y = torch.randn(M, F, device="cuda")
w = torch.randn(F, F, device="cuda") * 0.05; w2 = torch.randn(F, F, device="cuda") * 0.05
b = torch.randn(F, device="cuda"); out = torch.empty(M, F, device="cuda")
P = lambda t: t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
for rep in range(2):
    for v in (8, 10):
        for sched in [int(a) for a in sys.argv[1].split(",")]:
            lib.gts_set_option(1, v); lib.gts_set_option(7, sched)
            t1 = timeit(lambda: lib.gts_linear_fwd_f32(P(x), P(w), None, None, P(b), P(out), M, F, F, 0, 1, None, None, st))
            t2 = timeit(lambda: lib.gts_linear_fwd_f32(P(x), P(w), P(y), P(w2), P(b), P(out), M, F, F, F, 1, None, None, st))
            print(f"variant {v} sched {sched}: single {t1:6.1f} us  pair {t2:6.1f} us", flush=True)
