"""Where the clustered GAT forward aggregation spends its time (tools/diag): the streaming kernel alone with the neighbour gathers,
the reduction or the stores switched off, for gather depths 1 - 3 and 1 / 2 workgroups per CU, on the C3 hidden-layer shape
(B lattice graphs, 4 heads x 256).  Tables rotated so that inputs come from HBM.  Usage: python tools/diag/gat_whatif.py [graphs]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gnn-tumor-seg_amd"))
import torch  # noqa: E402

import gts  # noqa: E402
from gts import _lib, ops, synth  # noqa: E402

so = os.path.join(ROOT, "tools/diag/build/libgat_whatif.so")
if not os.path.exists(so) or os.environ.get("GTS_DIAG_REBUILD"):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off",
                           f"-I{ROOT}/include", f"-I{ROOT}/gnn-tumor-seg_amd/csrc", os.path.join(ROOT, "tools/diag/gat_whatif.hip"),
                           "-o", so])
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--build-only":
    sys.exit(0)
lib = ctypes.CDLL(so)
P, I32, I64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
lib.gts_whatif_gat_fwd.argtypes = [P, I64, I32, I32, I32, P, P, I32, P, P, I64, I64, I32, P]
lib.gts_whatif_knobs.argtypes = [I32, I32, I32, I32]


def timeit(fn, n_sets, reps=6):
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


b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = gts.batch([synth.lattice_graph() for _ in range(b)]).to("cuda")
n, h, d = g.n, 4, 256
n_sets = 3
fts = [torch.randn(n, h, d, device="cuda") for _ in range(n_sets)]
outs = [torch.empty(n, h, d, device="cuda") for _ in range(n_sets)]
al, ar = torch.randn(h, d, device="cuda") * 0.1, torch.randn(h, d, device="cuda") * 0.1
bias = torch.randn(h * d, device="cuda") * 0.1
el, er = ops.gat_scores(fts[0], al, ar)
ops._gat_fwd(g, fts[0], el, er, 0.2, bias, None, 1)          # leaves the weight blocks in the workspace
ws = ops._gat_ws(fts[0].device, 0)
ds = g.dev_schedule("gat_in")
hs = ds.host
st = torch.cuda.current_stream().cuda_stream
print(f"B={b}: {hs.n_clusters} clusters, limits {hs.limits}, staged rows per row {hs.staged_rows / n:.2f}", flush=True)
for depth, per_cu, waves in ((1, 2, 12), (2, 2, 12), (1, 1, 16), (2, 1, 16), (3, 1, 16), (2, 1, 12), (3, 1, 12)):
    lib.gts_whatif_knobs(depth, per_cu, waves, 16)
    line = []
    for whatif, label in ((0, "everything"), (1, "no gathers"), (2, "gathers only"), (3, "no stores")):
        for act in ((1, 0) if whatif == 0 else (1,)):
            def run(i):
                code = lib.gts_whatif_gat_fwd(ds.packed.data_ptr(), hs.n_clusters, hs.limits[0], hs.limits[1], hs.loc_words,
                                              fts[i].data_ptr(), bias.data_ptr(), act, outs[i].data_ptr(), ws.data_ptr(), n, h, whatif, st)
                assert code == 0, code
            line.append(f"{label}{'' if act else ' (no ELU)'} {timeit(run, n_sets):6.1f}")
    print(f"depth {depth} per_cu {per_cu} waves {waves:2d}: " + " | ".join(line) + " us", flush=True)

# ---- where an iteration of the persistent kernel spends its time (wave 0 of every workgroup, shader clock, no stores)
import numpy as np  # noqa: E402

for depth, per_cu, waves in ((1, 2, 12), (1, 2, 8), (2, 1, 12)):
    lib.gts_whatif_knobs(depth, per_cu, waves, 16)
    dbg = torch.zeros(8 * 4096, dtype=torch.int64, device="cuda")
    for _ in range(3):
        assert lib.gts_whatif_gat_fwd(ds.packed.data_ptr(), hs.n_clusters, hs.limits[0], hs.limits[1], hs.loc_words, fts[0].data_ptr(),
                                      bias.data_ptr(), 1, dbg.data_ptr(), ws.data_ptr(), n, h, 9, st) == 0
    torch.cuda.synchronize()
    dd = dbg.cpu().numpy().reshape(-1, 8)
    dd = dd[dd[:, 4] > 0]
    per_it = dd[:, :4] / dd[:, 4:5]
    print(f"depth {depth} per_cu {per_cu} waves {waves}: {len(dd)} workgroups, {dd[:, 4].mean():.1f} units each; clock ticks per unit "
          f"{per_it.sum(1).mean():.0f} = wait for gathers {per_it[:, 0].mean():.0f} | barrier {per_it[:, 1].mean():.0f} | issue "
          f"{per_it[:, 2].mean():.0f} | reduce (no stores) {per_it[:, 3].mean():.0f}   (s_memtime: 100 MHz ticks)", flush=True)
