"""K2 (clustered max-pool backward, F = 256) at the small shapes of the path — the reference's batches (6 graphs of 5 832 nodes) and
C2 (4 x 15 000) — by cluster limits, persistent workgroups per CU and waves per workgroup; tensors rotate through a footprint beyond the
Infinity Cache, as between the GEMMs of a step.   python tools/tune_k2_small.py"""
import os
import sys

sys.path.insert(0, "gnn-tumor-seg_amd")
import torch  # noqa: E402

import gts  # noqa: E402
from gts import _lib, ops, schedule, synth  # noqa: E402

lib = _lib.load()


def timeit(fn, n_sets, reps=8):
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


for label, dims, b in (("real 6 x 5832", (18, 18, 18), 6), ("C2 4 x 15000", (25, 25, 24), 4)):
    for limits in ("32,76,512;32,60,512", "32,76,512;16,36,256", "32,76,512;24,48,384", "32,76,512;48,84,512"):
        os.environ["GTS_CLUSTER_LIMITS"] = limits
        g = gts.batch([synth.lattice_graph(dims) for _ in range(b)]).to("cuda")
        n, e_b = g.n, g.number_of_edges()
        n_sets = max(2, int(700e6 // (n * 256 * 4 * 3)) + 1)
        gouts = [torch.randn(n, 256, device="cuda") for _ in range(n_sets)]
        xs = [torch.randn(n, 256, device="cuda").relu_() for _ in range(2)]
        args = [ops.spmm_max_fwd(g, xs[i % 2], relu_input=True)[1] for i in range(n_sets)]
        need_b = 4 * 256 * n * 2 + 256 * n + 4 * (2 * e_b + n + 1)
        s_out = g.cluster_schedule("out")
        if s_out is None:
            print(f"{label} limits {limits}: no schedule")
            continue
        row = []
        for per_cu in (0, 1, 3):
            for waves in (0, 8, 16):
                lib.gts_set_option(11, per_cu)
                lib.gts_set_option(12, waves)
                try:
                    tb = timeit(lambda i: ops.spmm_max_bwd(g, gouts[i], args[i]), n_sets)
                    row.append(f"cu{per_cu or 2}/w{waves or 12} {tb:5.1f}")
                except Exception as exc:      # noqa: BLE001
                    row.append(f"cu{per_cu or 2}/w{waves or 12} --")
        lib.gts_set_option(11, 0)
        lib.gts_set_option(12, 0)
        print(f"{label} out-limits {limits.split(';')[1]:>11s} clusters {s_out.n_clusters:5d} staged/row {s_out.staged_rows / n:.2f}: "
              + " | ".join(row) + f"   (us; {need_b / 1e6:.1f} MB compulsory)", flush=True)
        del gouts, args, xs, g
        torch.cuda.empty_cache()
