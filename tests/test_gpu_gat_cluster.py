"""GATConv's aggregation over the cluster row schedule (csrc/gts_gat_cluster.hip: LDS-staged neighbour slices) against
the plain kernels (csrc/gts_gat.hip), on a real MI355X.  Bar: BIT-EXACT — attention weights, outputs, gradients —
because the schedule only changes which workgroup produces a row: the edge softmax keeps the plain kernel's
association and the weighted sum runs in CSR slot order (DGL GATConv as called at
/root/reference/model/networks.py:46,52,56; the plain kernels are checked against oracle/torch_ref.py in
tests/test_gpu_kernels.py and tests/test_gpu_model.py)."""
import pytest
import torch

import gts
from gts import _lib, ops, schedule, synth
from gts.ops import current_stream, ptr

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _lib_and_rules(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    old = schedule.MIN_ROWS_GAT, schedule.MAX_DEGREE_GAT, schedule.WORTHWHILE
    # size and density rules are speed choices: here every graph a schedule can be built for takes the clustered kernels
    schedule.MIN_ROWS_GAT, schedule.MAX_DEGREE_GAT, schedule.WORTHWHILE = 0, 10 ** 9, 10.0
    yield hip_lib
    schedule.MIN_ROWS_GAT, schedule.MAX_DEGREE_GAT, schedule.WORTHWHILE = old


def _plain(fn):
    old = schedule.ENABLED_GAT
    schedule.ENABLED_GAT = False
    try:
        return fn()
    finally:
        schedule.ENABLED_GAT = old


def _graphs():
    lat = synth.lattice_graph((9, 8, 7))
    geo = synth.geometric_graph(n=1500, k=8, seed=3, self_loops=True)
    return {"lattice": lat, "lattice_self_loops": synth.lattice_graph((5, 6, 7), self_loops=True), "geometric": geo,
            "generator_b": synth.random_graph(n=3000, n_pairs=6000, seed=11),          # rows of two and three 8-edge chunks
            "batch": gts.batch([lat, geo, synth.lattice_graph((4, 4, 30))]),
            # enough clusters that every persistent workgroup walks through several units (its pipeline in steady state)
            "lattice_large": gts.batch([synth.lattice_graph() for _ in range(2)])}


def _inputs(g, heads, seed):
    gen = torch.Generator().manual_seed(seed)
    ft = torch.randn(g.n, heads, 256, generator=gen)
    el, er = torch.randn(g.n, heads, generator=gen) * 2, torch.randn(g.n, heads, generator=gen) * 2
    return ft.to(DEV), el.to(DEV), er.to(DEV)


@pytest.mark.parametrize("max_degree", [3, 8, 13, 30, 64])
def test_attention_weights_alone_equal_the_fused_forward_kernels(max_degree):
    """gts_gat_attn_f32 with 8 / 16 / 32 / 64 lanes per row against attn as written by gts_gat_fwd_f32 (64 lanes)."""
    n = 500
    gen = torch.Generator().manual_seed(max_degree)
    deg = torch.randint(1, max_degree + 1, (n,), generator=gen)
    deg[0] = max_degree
    dst = torch.repeat_interleave(torch.arange(n), deg)
    src = torch.randint(0, n, (int(deg.sum()),), generator=gen)
    g = gts.Graph(src.numpy().astype("int32"), dst.numpy().astype("int32"), n)
    assert g.max_in_degree == max_degree
    gd = g.to(DEV)
    ft, el, er = _inputs(g, 4, 1)
    _, want = _plain(lambda: ops._gat_fwd(gd, ft, el, er, 0.2))
    got = torch.full_like(want, float("nan"))
    d = gd.dev()
    _lib.check(_lib.load().gts_gat_attn_f32(ptr(d.indptr), ptr(d.indices), ptr(el), ptr(er), 0.2, ptr(got), n, 4, max_degree,
                                           current_stream()), "gts_gat_attn_f32")
    assert torch.equal(got, want)


@pytest.mark.parametrize("name", ["lattice", "lattice_self_loops", "geometric", "generator_b", "batch", "lattice_large"])
@pytest.mark.parametrize("heads,bias,act", [(4, True, 1), (1, False, 0), (3, True, 0)])
def test_clustered_gat_forward_and_source_pass_equal_the_plain_kernels_bit_for_bit(name, heads, bias, act):
    g = _graphs()[name]
    assert g.cluster_schedule("gat_in") is not None and g.cluster_schedule("gat_out") is not None
    # the edge pass of the backward takes its clustered form on rows of one 8-edge chunk (both lattices, their batches)
    assert (g.cluster_schedule("gat_edge_in") is not None) or g.max_in_degree > 8 or name == "batch"
    gd = g.to(DEV)
    ft, el, er = _inputs(g, heads, len(name))
    b = torch.randn(heads * 256, generator=torch.Generator().manual_seed(5)).to(DEV) if bias else None
    out, attn = ops._gat_fwd(gd, ft, el, er, 0.2, bias=b, activation=act)
    out_p, attn_p = _plain(lambda: ops._gat_fwd(gd, ft, el, er, 0.2, bias=b, activation=act))
    assert torch.equal(attn, attn_p)
    assert torch.equal(out, out_p)

    gout = torch.randn(g.n, heads, 256, generator=torch.Generator().manual_seed(6)).to(DEV)
    al = torch.randn(heads, 256, generator=torch.Generator().manual_seed(7)).to(DEV)
    ar = torch.randn(heads, 256, generator=torch.Generator().manual_seed(8)).to(DEV)
    for vecs in ((al, ar), (None, None)):
        got = ops._gat_bwd(gd, ft, el, er, attn, gout, 0.2, *vecs)
        want = _plain(lambda: ops._gat_bwd(gd, ft, el, er, attn, gout, 0.2, *vecs))
        for a, w, what in zip(got, want, ("gft", "gel", "ger")):
            assert torch.equal(a, w), what


@pytest.mark.parametrize("name", ["lattice_large", "batch", "generator_b"])
def test_units_dealt_off_the_counters_and_round_robin_give_the_same_bits(name):
    """The streaming kernel's workgroups take their units off one counter per XCD (GTS_OPT_GAT_CLUSTER_DEALING 0, the default) or
    round-robin (1): which workgroup computes a row changes nothing in the row — repeated launches included (the counters are
    zeroed by every call)."""
    lib = _lib.load()
    g = _graphs()[name]
    gd = g.to(DEV)
    heads = 4
    ft, el, er = _inputs(g, heads, 3)
    b = torch.randn(heads * 256, generator=torch.Generator().manual_seed(5)).to(DEV)
    gout = torch.randn(g.n, heads, 256, generator=torch.Generator().manual_seed(6)).to(DEV)
    al = torch.randn(heads, 256, generator=torch.Generator().manual_seed(7)).to(DEV)
    ar = torch.randn(heads, 256, generator=torch.Generator().manual_seed(8)).to(DEV)
    assert lib.gts_get_option(17) == 0
    results = []
    try:
        for dealing in (0, 1, 0, 0):
            assert lib.gts_set_option(17, dealing) == 0
            out, attn = ops._gat_fwd(gd, ft, el, er, 0.2, bias=b, activation=1)
            results.append((out, attn) + tuple(ops._gat_bwd(gd, ft, el, er, attn, gout, 0.2, al, ar)))
    finally:
        lib.gts_set_option(17, 0)
    assert lib.gts_set_option(17, 2) != 0
    for other in results[1:]:
        for a, w in zip(other, results[0]):
            assert torch.equal(a, w)


@pytest.mark.parametrize("name", ["lattice", "lattice_self_loops", "batch"])
def test_clustered_gatconv_layer_with_bias_and_elu_matches_the_oracle(name):
    """One GATConv layer (bias + ELU, 4 heads x 256: the C3 hidden layer of /root/reference/model/networks.py:52) on a
    forced schedule — clustered forward, edge pass and source pass — against oracle/torch_ref.RefGATConv directly
    (forward rtol 1e-4, gradients rtol 1e-3, as the plain-kernel layer test in tests/test_gpu_model.py)."""
    import torch.nn.functional as F

    from gts import nn as gnn
    from oracle import graph_ref, torch_ref
    from tests.helpers import copy_state

    g = _graphs()[name]
    for which in ("gat_in", "gat_out"):
        assert g.cluster_schedule(which) is not None
    edge_clustered = g.cluster_schedule("gat_edge_in") is not None and g.max_in_degree <= 8
    assert edge_clustered or name == "batch"
    tg = torch_ref.TGraph(graph_ref.RefGraph(g.src, g.dst, g.n))
    torch.manual_seed(len(name))
    fin, heads, dim = 64, 4, 256
    ref = torch_ref.RefGATConv(fin, dim, heads, residual=False, activation=F.elu)
    mine = gnn.GATConv(fin, dim, heads, 0, 0, 0.2, False, F.elu)
    with torch.no_grad():
        ref.bias.normal_(0, 0.5)
    copy_state(mine, ref)
    mine.to(DEV)
    x = torch.randn(g.n, fin) * 0.5
    gout = torch.randn(g.n, heads, dim)
    xr = x.clone().requires_grad_(True)
    yr = ref(tg, xr)
    yr.backward(gout)
    ran = []
    lib = _lib.load()
    real = {k: getattr(lib, k) for k in ("gts_gat_fwd_cluster_f32", "gts_gat_bwd_edge_cluster_f32", "gts_gat_bwd_src_cluster_f32")}
    try:
        for k, fn in real.items():
            setattr(lib, k, (lambda k_, fn_: (lambda *a: (ran.append(k_), fn_(*a))[1]))(k, fn))
        xd = x.to(DEV).requires_grad_(True)
        yd = mine(g.to(DEV), xd)
        yd.backward(gout.to(DEV))
    finally:
        for k, fn in real.items():
            setattr(lib, k, fn)
    assert "gts_gat_fwd_cluster_f32" in ran and "gts_gat_bwd_src_cluster_f32" in ran
    assert ("gts_gat_bwd_edge_cluster_f32" in ran) == edge_clustered

    def close(a, b, rtol, atol):
        a, b = a.detach().cpu(), b.detach()
        assert torch.allclose(a, b, rtol=rtol, atol=atol), float((a - b).abs().max())
    close(yd, yr, 1e-4, 1e-5 * max(1.0, float(yr.abs().max())))
    close(xd.grad, xr.grad, 1e-3, 1e-5 * max(1.0, float(xr.grad.abs().max())))
    for (pname, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        close(p.grad, q.grad, 1e-3, 2e-5 * max(1.0, float(q.grad.abs().max())))


def test_default_rules_pick_the_clustered_kernels_for_the_c3_batch_and_not_for_dense_or_small_graphs(monkeypatch):
    monkeypatch.setattr(schedule, "MIN_ROWS_GAT", 20000)
    monkeypatch.setattr(schedule, "MAX_DEGREE_GAT", 8)
    monkeypatch.setattr(schedule, "WORTHWHILE", 0.6)
    big = gts.batch([synth.lattice_graph() for _ in range(2)]).to(DEV)
    assert ops._gat_cluster_schedule(big, "gat_in", big.n, 4, 256) is not None
    assert ops._gat_cluster_schedule(big, "gat_out", big.n, 4, 256) is not None
    assert ops._gat_cluster_schedule(big, "gat_edge_in", big.n, 4, 256) is not None
    assert ops._gat_cluster_schedule(big, "gat_in", big.n, 4, 64) is None                      # D != 256
    small = synth.lattice_graph((9, 8, 7)).to(DEV)
    assert ops._gat_cluster_schedule(small, "gat_in", small.n, 4, 256) is None
    dense = synth.geometric_graph(n=30000, k=12, seed=2).to(DEV)                                # rows of two chunks
    assert ops._gat_cluster_schedule(dense, "gat_in", dense.n, 4, 256) is None
