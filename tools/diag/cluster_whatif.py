"""Where the persistent clustered K1 / K2 kernel spends its time: the same launch with the row gathers switched
off (records + reduction + stores only) and with the reduction switched off (records + gathers only), on B lattice
graphs, tensors rotated so that inputs come from HBM.  Usage: python tools/diag/cluster_whatif.py [graphs ...]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gnn-tumor-seg_amd"))
import torch  # noqa: E402

import gts  # noqa: E402
from gts import ops, synth  # noqa: E402

so = "/tmp/libcluster_whatif.so"
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off",
                       f"-I{ROOT}/include", f"-I{ROOT}/gnn-tumor-seg_amd/csrc", os.path.join(ROOT, "tools/diag/cluster_whatif.hip"),
                       "-o", so])
lib = ctypes.CDLL(so)
P, I32, I64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
lib.gts_whatif_cluster.argtypes = [P, I64, I32, I32, I32, P, P, P, P, I32, I32, I64, P]


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


for b in [int(a) for a in sys.argv[1:]] or [8]:
    g = gts.batch([synth.lattice_graph() for _ in range(b)]).to("cuda")
    n = g.n
    n_sets = max(2, int(700e6 // (n * 256 * 4 * 3)) + 1)
    xs = [torch.randn(n, 256, device="cuda").relu_() for _ in range(n_sets)]
    gouts = [torch.randn(n, 256, device="cuda") for _ in range(n_sets)]
    args = [ops.spmm_max_fwd(g, x, relu_input=True)[1] for x in xs]
    outs = [torch.empty_like(x) for x in xs]
    st = torch.cuda.current_stream().cuda_stream
    for bwd, which in ((0, "in"), (1, "out")):
        ds = g.dev_schedule(which)
        h = ds.host
        cases = [(0, "everything"), (1, "no row gathers"), (2, "no reduction / stores")]
        if not bwd:
            cases += [(3, "no stores"), (4, "no reduction")]
        for whatif, label in cases:
            def run(i):
                table = gouts[i] if bwd else xs[i]
                code = lib.gts_whatif_cluster(ds.packed.data_ptr(), h.n_clusters, h.limits[0], h.limits[1], h.loc_words,
                                              table.data_ptr(), args[i].data_ptr(), outs[i].data_ptr(),
                                              None if bwd else args[i].data_ptr(), bwd, whatif, n, st)
                assert code == 0, code
            print(f"B={b} {'K2' if bwd else 'K1'} {label:24s}: {timeit(run, n_sets):7.1f} us", flush=True)

# ---- where an iteration of the persistent K1 kernel spends its time (wave 0 of every workgroup, shader clock)
import numpy as np  # noqa: E402

for b in (4, 8):
    g = gts.batch([synth.lattice_graph() for _ in range(b)]).to("cuda")
    n = g.n
    x = torch.randn(n, 256, device="cuda").relu_()
    out = torch.empty_like(x)
    ds = g.dev_schedule("in")
    h = ds.host
    dbg = torch.zeros(8 * 4096, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert lib.gts_whatif_cluster(ds.packed.data_ptr(), h.n_clusters, h.limits[0], h.limits[1], h.loc_words, x.data_ptr(),
                                      None, out.data_ptr(), dbg.data_ptr(), 0, 9, n, st) == 0
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(-1, 8)
    d = d[d[:, 4] > 0]
    per_it = d[:, :4] / d[:, 4:5]
    tot = per_it.sum(1).mean()
    print(f"B={b}: {len(d)} workgroups, {d[:, 4].mean():.1f} units each; cycles per unit {tot:.0f} = wait for gathers "
          f"{per_it[:, 0].mean():.0f} | barrier {per_it[:, 1].mean():.0f} | issue {per_it[:, 2].mean():.0f} | reduce + stores "
          f"{per_it[:, 3].mean():.0f}", flush=True)
