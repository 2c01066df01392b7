"""The three clustered GAT aggregation calls (forward, edge pass, source pass) at C3's hidden-layer shape, 20 times each between HIP
events, for the knobs of the streaming kernel: clusters walked together (--group), workgroups per CU, waves.  Outputs of
every run are compared bit for bit with the first.  One variant per process suits a rocprofv3 --pmc pass around it.
(Round 4 also ran this on PITCHED tables — rows 4 KiB + 512 B apart, to spread a slice column over the L2's sets: no change in time or
in FETCH_SIZE, the L2's set index is hashed; profiles/r04/README.md.)
Usage: python tools/diag/gat_passes_ab.py [--graphs 4] [--group 0] [--per-cu 1] [--waves 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gnn-tumor-seg_amd"))
import gts   # noqa: E402
from gts import _lib, ops, synth   # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        out = fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return out, a.elapsed_time(b) * 1000.0 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=4)
    ap.add_argument("--only", default="dense", help="kept for the scripts of profiles/r04: only 'dense' exists")
    ap.add_argument("--group", type=int, default=None)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--per-cu", type=int, default=0, help="persistent workgroups per CU (option 11; 0 = default 2)")
    ap.add_argument("--waves", type=int, default=0, help="waves per workgroup (option 15; 0 = default 12)")
    args = ap.parse_args()
    lib = _lib.load()
    lib.gts_set_option(11, args.per_cu)
    lib.gts_set_option(15, args.waves)
    dev = torch.device("cuda:0")
    g = gts.batch([synth.lattice_graph() for _ in range(args.graphs)]).to(dev)
    n, h, d = g.n, 4, 256
    gen = torch.Generator(device="cpu").manual_seed(5)
    ft = torch.randn((n, h, d), generator=gen).to(dev)
    gout = torch.randn((n, h, d), generator=gen).to(dev)
    al, ar = torch.randn((h, d), generator=gen).to(dev) * 0.1, torch.randn((h, d), generator=gen).to(dev) * 0.1
    bias = torch.randn((h * d,), generator=gen).to(dev) * 0.1
    el, er = ops.gat_scores(ft, al, ar)
    tables = {"dense": (ft, gout)}
    ref = None
    compulsory = {"fwd": 2 * n * h * d * 4, "edge": 2 * n * h * d * 4, "src": 2 * n * h * d * 4}
    for variant in ("dense",):
        for group in ([args.group] if args.group is not None else [16, 0, 64]):
            lib.gts_set_option(16, group)
            t_ft, t_g = tables[variant]
            (out, attn), us_f = timed(lambda: ops._gat_fwd(g, t_ft, el, er, 0.2, bias, None, 1), args.reps)
            ops.KERNEL_TIMERS.clear()
            stamps = {}

            def bracket(name):
                def wrap(launch):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    rc = launch()
                    b.record()
                    stamps.setdefault(name, []).append((a, b))
                    return rc
                return wrap
            ops.KERNEL_TIMERS["gat_bwd_edge"] = bracket("edge")
            ops.KERNEL_TIMERS["gat_bwd_src"] = bracket("src")
            for _ in range(3):
                res = ops._gat_bwd(g, t_ft, el, er, attn, t_g, 0.2, al, ar)
            stamps.clear()
            for _ in range(args.reps):
                res = ops._gat_bwd(g, t_ft, el, er, attn, t_g, 0.2, al, ar)
            torch.cuda.synchronize()
            ops.KERNEL_TIMERS.clear()
            us = {k: sum(a.elapsed_time(b) for a, b in v) * 1000.0 / len(v) for k, v in stamps.items()}
            us["fwd"] = us_f
            got = (out, attn) + tuple(res)
            if ref is None:
                ref = got
            same = all(torch.equal(x, y) for x, y in zip(ref, got))
            print(f"group {group:3d} per_cu {args.per_cu} waves {args.waves}: " +
                  "  ".join(f"{k} {us[k]:7.1f} us = {compulsory[k] / us[k] / 8e6:.3f}" for k in ("fwd", "edge", "src")) +
                  f"  same bits as the first run: {same}", flush=True)
            assert same


if __name__ == "__main__":
    main()
