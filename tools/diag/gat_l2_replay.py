"""What an XCD's L2 keeps of the clustered GAT forward's gathers when the units pass through it strictly in order (tools/diag): the
gathers-only form of the streaming kernel (what-if 2) with ONE workgroup per XCD (grid 8) and with 2 / 8 / 64 per XCD, walking the whole
span slice by slice (group 100000) or 16 clusters through all their slices.  Run under rocprofv3 --pmc FETCH_SIZE; the offline model
(tools/diag/l2_halo_sim.py: 4 MiB LRU) says 1.08 x the table for the whole-span walk, 1.66 x for groups of 16.
Usage: rocprofv3 --pmc FETCH_SIZE ... -- python tools/diag/gat_l2_replay.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gnn-tumor-seg_amd"))
import torch  # noqa: E402

import gts  # noqa: E402
from gts import ops, synth  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools/diag/build/libgat_whatif.so"))
P, I32, I64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
lib.gts_whatif_gat_fwd.argtypes = [P, I64, I32, I32, I32, P, P, I32, P, P, I64, I64, I32, P]
lib.gts_whatif_knobs.argtypes = [I32, I32, I32, I32]
lib.gts_whatif_grid.argtypes = [I32]
g = gts.batch([synth.lattice_graph() for _ in range(4)]).to("cuda")
n, h, d = g.n, 4, 256
ft = torch.randn(n, h, d, device="cuda")
out = torch.empty(n, h, d, device="cuda")
al, ar = torch.randn(h, d, device="cuda") * 0.1, torch.randn(h, d, device="cuda") * 0.1
bias = torch.randn(h * d, device="cuda") * 0.1
el, er = ops.gat_scores(ft, al, ar)
ops._gat_fwd(g, ft, el, er, 0.2, bias, None, 1)
ws = ops._gat_ws(ft.device, 0)
ds = g.dev_schedule("gat_in")
hs = ds.host
st = torch.cuda.current_stream().cuda_stream
lib.gts_whatif_nt.argtypes = [I32]
lib.gts_whatif_dealing.argtypes = [I32]
# (group, grid, what-if, nt, dealing): static round-robin dealing (1) — gathers only, one workgroup per XCD and the launch's 512; the
# whole kernel; everything but the stores; ordinary instead of non-temporal stores — then the units dealt off a counter per XCD (0)
cases = [(group, grid, 2, 1, 1) for group in (100000, 16) for grid in (8, 512)]
cases += [(100000, 512, 0, 1, 1), (100000, 512, 3, 1, 1), (100000, 512, 0, 0, 1), (100000, 8, 0, 1, 1), (16, 512, 0, 1, 1)]
cases += [(100000, 512, 0, 1, 0), (100000, 512, 2, 1, 0), (100000, 512, 0, 0, 0), (100000, 8, 0, 1, 0), (100000, 512, 0, 1, 0)]
for group, grid, whatif, nt, dealing in cases:
    lib.gts_whatif_knobs(1, 2, 12, group)
    lib.gts_whatif_grid(grid)
    lib.gts_whatif_nt(nt)
    lib.gts_whatif_dealing(dealing)
    for _ in range(2):
        assert lib.gts_whatif_gat_fwd(ds.packed.data_ptr(), hs.n_clusters, hs.limits[0], hs.limits[1], hs.loc_words, ft.data_ptr(),
                                      bias.data_ptr(), 1, out.data_ptr(), ws.data_ptr(), n, h, whatif, st) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    print(f"CASE group {group} grid {grid} whatif {whatif} nt {nt} dealing {dealing}", flush=True)
