"""Time K1/K2 at the C2 shape (4 lattice graphs, F=256) for rows-per-wave / streaming knobs."""
import sys

sys.path.insert(0, "gnn-tumor-seg_amd")
import torch  # noqa: E402

import gts  # noqa: E402
from gts import _lib, ops, synth  # noqa: E402

lib = _lib.load()
g = gts.batch([synth.lattice_graph() for _ in range(4)]).to("cuda")
x = torch.randn(g.n, 256, device="cuda").relu_()
gout = torch.randn(g.n, 256, device="cuda")
out, arg = ops.spmm_max_fwd(g, x)


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


e_b, n_b = g.number_of_edges(), g.n
fwd_bytes = 4 * 256 * e_b + 4 * 256 * n_b + 256 * n_b + 4 * (e_b + n_b + 1)
bwd_bytes = (4 * 256 + 256) * e_b + 2 * 4 * 256 * n_b + 12 * e_b
for nt in (-1, 0, 1, 2, 3):
    for seq in (0, 1, 2):
        lib.gts_set_option(4, seq)
        lib.gts_set_option(5, nt)
        tf = timeit(lambda: ops.spmm_max_fwd(g, x))
        tb = timeit(lambda: ops.spmm_max_bwd(g, gout, arg, relu_src=x))
        print(f"nt={nt} rows/wave={seq:2d}: fwd {tf:6.1f} us ({fwd_bytes / tf / 1e6:6.2f} TB/s alg)  "
              f"bwd {tb:6.1f} us ({bwd_bytes / tb / 1e6:6.2f} TB/s alg)", flush=True)
lib.gts_set_option(4, 0)
lib.gts_set_option(5, -1)
