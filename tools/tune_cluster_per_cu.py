"""Clustered K1 / K2 at 8 and 32 lattice graphs per batch: persistent workgroups per CU 2 (round 3) against 3.   python tools/tune_cluster_per_cu.py"""
import sys

sys.path.insert(0, "gnn-tumor-seg_amd")
import torch  # noqa: E402

import gts  # noqa: E402
from gts import _lib, ops, synth  # noqa: E402

lib = _lib.load()


def timeit(fn, n_sets, reps=5):
    for i in range(n_sets):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        for i in range(n_sets):
            fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (reps * n_sets)


for b in (8, 32):
    g = gts.batch([synth.lattice_graph() for _ in range(b)]).to("cuda")
    n = g.n
    n_sets = max(2, int(700e6 // (n * 256 * 4 * 3)) + 1)
    xs = [torch.randn(n, 256, device="cuda").relu_() for _ in range(n_sets)]
    gouts = [torch.randn(n, 256, device="cuda") for _ in range(n_sets)]
    args = [ops.spmm_max_fwd(g, x, relu_input=True)[1] for x in xs]
    for per_cu in (2, 3, 2, 3):
        lib.gts_set_option(11, per_cu)
        tf = timeit(lambda i: ops.spmm_max_fwd(g, xs[i], relu_input=True), n_sets)
        tb = timeit(lambda i: ops.spmm_max_bwd(g, gouts[i], args[i]), n_sets)
        print(f"B={b} workgroups per CU {per_cu}: K1 {tf:6.1f} us  K2 {tb:6.1f} us", flush=True)
    lib.gts_set_option(11, 0)
    del xs, gouts, args
    torch.cuda.empty_cache()
