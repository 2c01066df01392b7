"""K1 / K2 over the cluster row schedule (csrc/gts_spmm_cluster.hip: LDS-staged neighbour tiles) against the
plain kernels and the CPU oracle, on a real MI355X.  Bar: BIT-EXACT — values, winners and gradients — on every
graph family, because the schedule only changes which workgroup produces a row: DGL's copy_u / max
(/root/reference/model/networks.py:25,28,30) as restated in oracle/torch_ref.py is reduced in CSR slot order."""
import numpy as np
import pytest
import torch

import gts
from gts import ops, schedule, synth
from oracle import graph_ref, torch_ref
from tests.helpers import slots_to_sources

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _lib(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    old = schedule.MIN_ROWS_FORWARD, schedule.MAX_MEAN_DEGREE_FORWARD, schedule.MAX_DEGREE_BACKWARD
    # the size and density rules are speed choices: here every graph with a schedule takes the clustered kernels
    schedule.MIN_ROWS_FORWARD, schedule.MAX_MEAN_DEGREE_FORWARD, schedule.MAX_DEGREE_BACKWARD = 0, 1e9, 10 ** 9
    yield hip_lib
    schedule.MIN_ROWS_FORWARD, schedule.MAX_MEAN_DEGREE_FORWARD, schedule.MAX_DEGREE_BACKWARD = old


def _plain(fn):
    """Run fn with the clustered path switched off (the plain per-edge kernels)."""
    old = schedule.ENABLED
    schedule.ENABLED = False
    try:
        return fn()
    finally:
        schedule.ENABLED = old


def _graphs():
    lat = synth.lattice_graph((9, 8, 7))
    geo = synth.geometric_graph(n=1500, k=8, seed=3, self_loops=True)
    keep = geo.dst % 97 != 5                                   # some rows without in-edges
    holes = gts.Graph(geo.src[keep], geo.dst[keep], geo.n)
    wide = synth.geometric_graph(n=900, k=20, seed=4)          # degree 20 .. 40: clusters limited by their edges
    return {"lattice": lat, "lattice_self_loops": synth.lattice_graph((5, 6, 7), self_loops=True),
            "geometric_holes": holes, "geometric_wide": wide,
            "batch": gts.batch([lat, holes, synth.lattice_graph((4, 4, 30))]),
            # enough clusters that every persistent workgroup walks through several units (its pipeline in steady state)
            "geometric_large": synth.geometric_graph(n=60000, k=7, seed=6)}


@pytest.mark.parametrize("name", ["lattice", "lattice_self_loops", "geometric_holes", "geometric_wide", "batch", "geometric_large"])
@pytest.mark.parametrize("relu_input", [False, True])
def test_clustered_k1_k2_equal_plain_kernels_and_oracle_bit_for_bit(name, relu_input):
    g = _graphs()[name]
    assert g.cluster_schedule("in") is not None and g.cluster_schedule("out") is not None
    gd = g.to(DEV)
    gen = torch.Generator().manual_seed(len(name))
    x = torch.randn(g.n, 256, generator=gen)
    x[::7] = torch.randint(-2, 3, (x[::7].shape[0], 256), generator=gen).float()     # ties
    if relu_input:
        x = torch.relu(x)
    x[3, 5] = float("inf")
    gout = torch.randn(g.n, 256, generator=gen)
    xd, gd_out = x.to(DEV), gout.to(DEV)

    out, arg = ops.spmm_max_fwd(gd, xd, relu_input=relu_input)
    out_p, arg_p = _plain(lambda: ops.spmm_max_fwd(gd, xd, relu_input=relu_input))
    assert torch.equal(out, out_p) and torch.equal(arg, arg_p)
    out_only, none = ops.spmm_max_fwd(gd, xd, want_arg=False, relu_input=relu_input)
    assert none is None and torch.equal(out_only, out)

    gx = ops.spmm_max_bwd(gd, gd_out, arg)
    gx_p = _plain(lambda: ops.spmm_max_bwd(gd, gd_out, arg))
    assert torch.equal(gx, gx_p)

    tg = torch_ref.TGraph(graph_ref.RefGraph(g.src, g.dst, g.n))
    out_ref, arg_ref = torch_ref.spmm_max_with_arg(tg, x)
    assert torch.equal(out.cpu(), out_ref)
    if not relu_input:
        assert np.array_equal(slots_to_sources(g, arg), arg_ref.numpy())
        xr = x.clone().requires_grad_(True)
        torch_ref.spmm_max(tg, xr).backward(gout)
        # the oracle scatters in edge order, the kernels gather in out-CSR order: compare on integer gradients
        gi = torch.randint(-3, 4, gout.shape, generator=gen).float()
        xr.grad = None
        torch_ref.spmm_max(tg, xr).backward(gi)
        assert torch.equal(ops.spmm_max_bwd(gd, gi.to(DEV), arg).cpu(), xr.grad)


def test_clustered_path_is_the_one_that_runs_at_the_c2_shape():
    """4 x 15k-node lattice graphs, F = 256: the schedule exists, the clustered entry points are taken, results equal
    the plain kernels bit for bit at full size (N_b = 60 000)."""
    g = gts.batch([synth.lattice_graph() for _ in range(4)]).to(DEV)
    assert g.dev_schedule("in") is not None and g.dev_schedule("out") is not None
    x = torch.relu(torch.randn(g.n, 256, device=DEV))
    gout = torch.randn(g.n, 256, device=DEV)
    out, arg = ops.spmm_max_fwd(g, x, relu_input=True)
    out_p, arg_p = _plain(lambda: ops.spmm_max_fwd(g, x, relu_input=True))
    assert torch.equal(out, out_p) and torch.equal(arg, arg_p)
    assert torch.equal(ops.spmm_max_bwd(g, gout, arg), _plain(lambda: ops.spmm_max_bwd(g, gout, arg)))


def test_generator_b_forced_through_the_schedule_is_bit_exact_too():
    """Generator B (ring + uniform random pairs) shares almost no neighbours between rows, so the policy never schedules
    it; forced through the clustered kernels anyway (every edge its own staged row) it must give the same bits."""
    old = schedule.WORTHWHILE
    schedule.WORTHWHILE = 10.0
    try:
        g = synth.random_graph(n=15000, n_pairs=30000, seed=1000)
        s_in, s_out = g.cluster_schedule("in"), g.cluster_schedule("out")
        assert s_in is not None and s_out is not None and s_in.staged_rows > 0.8 * s_in.n_edges
        gd = g.to(DEV)
        x = torch.relu(torch.randn(g.n, 256, device=DEV))
        gout = torch.randn(g.n, 256, device=DEV)
        out, arg = ops.spmm_max_fwd(gd, x, relu_input=True)
        out_p, arg_p = _plain(lambda: ops.spmm_max_fwd(gd, x, relu_input=True))
        assert torch.equal(out, out_p) and torch.equal(arg, arg_p)
        assert torch.equal(ops.spmm_max_bwd(gd, gout, arg), _plain(lambda: ops.spmm_max_bwd(gd, gout, arg)))
        tg = torch_ref.TGraph(graph_ref.RefGraph(g.src, g.dst, g.n))
        out_ref, arg_ref = torch_ref.spmm_max_with_arg(tg, x.cpu())
        assert torch.equal(out.cpu(), out_ref)
    finally:
        schedule.WORTHWHILE = old


def test_other_widths_and_unworthy_graphs_take_the_plain_kernels():
    g = synth.random_graph(n=3000, n_pairs=6000, seed=1).to(DEV)
    assert g.dev_schedule("in") is None
    lat = synth.lattice_graph((6, 6, 6)).to(DEV)
    x = torch.randn(lat.n, 64, device=DEV)
    out, arg = ops.spmm_max_fwd(lat, x)                         # F = 64: plain kernel, no schedule upload
    assert not lat.dev().schedules


def test_cluster_entry_points_reject_bad_arguments():
    lib = gts._lib.load()
    g = synth.lattice_graph((6, 6, 6)).to(DEV)
    ds = g.dev_schedule("in")
    h = ds.host
    x = torch.randn(g.n, 256, device=DEV)
    out = torch.empty_like(x)
    P = lambda t: t.data_ptr()      # noqa: E731
    rec, lim = P(ds.packed), (h.limits[0], h.limits[1], h.loc_words)
    assert lib.gts_spmm_max_fwd_cluster_f32(rec, h.n_clusters, *lim, P(x), P(out), None, 0, 0, g.n, 128, None, None) == -2    # F != 256
    assert lib.gts_spmm_max_fwd_cluster_f32(rec, h.n_clusters, *lim, P(x), P(out), None, 1, 0, g.n, 256, None, None) == -1    # arg missing
    assert lib.gts_spmm_max_fwd_cluster_f32(rec, h.n_clusters, *lim, P(x), P(out), None, 4, 0, g.n, 256, None, None) == -1
    assert lib.gts_spmm_max_fwd_cluster_f32(rec, h.n_clusters, *lim, None, P(out), None, 0, 0, g.n, 256, None, None) == -1
    assert lib.gts_spmm_max_fwd_cluster_f32(rec, h.n_clusters, h.limits[0], 300, h.loc_words, P(x), P(out), None, 0, 0,
                                            g.n, 256, None, None) == -2                                          # byte-indexed neighbours
    assert lib.gts_spmm_max_fwd_cluster_f32(rec, h.n_clusters, *lim, P(x), P(out), None, 0, 0, g.n, 256, None, None) == 0   # no counters: static dealing
    want = out.clone()
    counters = torch.zeros(256, dtype=torch.int32, device=DEV)
    assert lib.gts_set_option(18, 1) == 0                       # units dealt off the counters
    try:
        for _ in range(3):        # the launch leaves its counters zero: the same buffer serves the next one
            out.fill_(float("nan"))
            assert lib.gts_spmm_max_fwd_cluster_f32(rec, h.n_clusters, *lim, P(x), P(out), None, 0, 0, g.n, 256, P(counters), None) == 0
            assert torch.equal(out, want)
            assert int(counters.abs().sum()) == 0
    finally:
        lib.gts_set_option(18, 0)
    torch.cuda.synchronize()


@pytest.mark.parametrize("options", [{10: 2}, {12: 4}, {12: 6, 11: 1}, {11: 3}, {12: 16, 11: 3, 10: 2}, {8: 0}, {18: 1}, {18: 2}, {18: 1, 10: 2}])
def test_every_launch_geometry_gives_the_same_bits(options):
    """Waves per workgroup, persistent workgroups per CU (2 by default, 3 for short backward launches), gathers one or two
    units ahead, streaming stores: tuning choices of the persistent form — results must not depend on them."""
    lib = gts._lib.load()
    g = gts.batch([synth.lattice_graph((25, 25, 24)), synth.geometric_graph(n=20000, k=8, seed=5)]).to(DEV)
    x = torch.relu(torch.randn(g.n, 256, device=DEV))
    gout = torch.randn(g.n, 256, device=DEV)
    out, arg = ops.spmm_max_fwd(g, x, relu_input=True)
    gx = ops.spmm_max_bwd(g, gout, arg)
    try:
        for k, v in options.items():
            assert lib.gts_set_option(k, v) == 0
        out2, arg2 = ops.spmm_max_fwd(g, x, relu_input=True)
        gx2 = ops.spmm_max_bwd(g, gout, arg)
    finally:
        for k in options:
            lib.gts_set_option(k, -1 if k == 8 else 0)
    assert torch.equal(out, out2) and torch.equal(arg, arg2) and torch.equal(gx, gx2)
