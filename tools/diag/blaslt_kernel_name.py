"""Diagnostic: which hipBLASLt kernel torch.mm picks for the C2 layer GEMM (60000x256 @ 256x256),
and how long it takes (run under rocprofv3 --kernel-trace --stats to see the kernel name, whose
suffix encodes macro tile, MFMA instruction and scheduling)."""
import torch

M, F = 60000, 256
x = torch.randn(M, F, device="cuda")
w = torch.randn(F, F, device="cuda") * 0.05
y = torch.randn(M, F, device="cuda")
w2 = torch.randn(F, 2 * F, device="cuda") * 0.05
out = torch.empty(M, F, device="cuda")
xy = torch.cat([x, y], dim=1)
for _ in range(5):
    torch.mm(x, w.t(), out=out)
    torch.mm(xy, w2.t(), out=out)
    torch.mm(x, w, out=out)
torch.cuda.synchronize()
for name, fn, flops in (("NT K=256", lambda: torch.mm(x, w.t(), out=out), 2.0 * M * F * F),
                        ("NT K=512", lambda: torch.mm(xy, w2.t(), out=out), 4.0 * M * F * F),
                        ("NN K=256", lambda: torch.mm(x, w, out=out), 2.0 * M * F * F)):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 30
    print(f"{name}: {us:7.1f} us  {flops / us / 1e6:6.1f} TF", flush=True)
