"""Parity of every HIP kernel (through the C ABI) with the CPU oracle, on a real MI355X.

Bars: bit-exact for max aggregation (+ its argmax), for the sum/mean/gcn reducers (the
kernels accumulate in the oracle's slot order and divide like it) and for the voxel
projection; rtol/atol 1e-5 for the GAT attention path (device expf is not bit-identical
to the host's)."""
import os

import numpy as np
import pytest
import torch

import gts
from gts import ops, synth
from oracle import graph_ref, torch_ref
from tests.helpers import HAND_EDGES, HAND_X, random_coo, ref_and_gts, slots_to_sources

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _lib(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return hip_lib


def _hand():
    src, dst = map(np.array, zip(*HAND_EDGES))
    return ref_and_gts(src, dst, 6)


def _feat(n, f, seed, integer=False):
    gen = torch.Generator().manual_seed(seed)
    if integer:   # small integers: many ties, exact sums
        return torch.randint(-3, 4, (n, f), generator=gen).float()
    return torch.randn(n, f, generator=gen)


# ------------------------------------------------------------------ K1 / K2
def test_max_hand_graph_bit_exact():
    tg, g = _hand()
    x = torch.from_numpy(HAND_X)
    out_ref, arg_ref = torch_ref.spmm_max_with_arg(tg, x)
    out, arg = ops.spmm_max_fwd(g.to(DEV), x.to(DEV))
    assert torch.equal(out.cpu(), out_ref)
    assert np.array_equal(slots_to_sources(g, arg), arg_ref.numpy())
    assert arg.dtype == torch.uint8


@pytest.mark.parametrize("f", [1, 3, 4, 8, 20, 64, 100, 256, 260, 512])
@pytest.mark.parametrize("integer", [False, True])
def test_max_forward_random_bit_exact(f, integer):
    n = 300
    src, dst = random_coo(n, 2000, seed=f)
    dst[dst == 7] = 8                     # node 7: zero in-degree
    tg, g = ref_and_gts(src, dst, n)
    x = _feat(n, f, seed=f + 1, integer=integer)
    out_ref, arg_ref = torch_ref.spmm_max_with_arg(tg, x)
    out, arg = ops.spmm_max_fwd(g.to(DEV), x.to(DEV))
    assert torch.equal(out.cpu(), out_ref)
    assert np.array_equal(slots_to_sources(g, arg), arg_ref.numpy())
    assert torch.all(out[7] == 0)
    out_only, none = ops.spmm_max_fwd(g.to(DEV), x.to(DEV), want_arg=False)
    assert none is None and torch.equal(out_only, out)


def test_max_inf_and_high_degree_int32_arg():
    n = 40
    hub_src = np.arange(n).repeat(8)              # node 0 gets 320 in-edges -> int32 slots
    hub_dst = np.zeros(n * 8, dtype=np.int64)
    src, dst = random_coo(n, 100, seed=4)
    src, dst = np.concatenate([hub_src, src]), np.concatenate([hub_dst, dst])
    tg, g = ref_and_gts(src, dst, n)
    assert g.arg_bytes == 4
    x = _feat(n, 12, seed=5)
    x[3, 2] = float("inf"); x[4, 5] = float("-inf")
    out_ref, arg_ref = torch_ref.spmm_max_with_arg(tg, x)
    out, arg = ops.spmm_max_fwd(g.to(DEV), x.to(DEV))
    assert arg.dtype == torch.int32
    assert torch.equal(out.cpu(), out_ref)
    assert np.array_equal(slots_to_sources(g, arg), arg_ref.numpy())
    assert out[0, 2] == 0                         # +inf maximum is replaced by 0 (R-max)


@pytest.mark.parametrize("f", [3, 4, 20, 256, 512])
def test_max_backward_matches_oracle(f):
    n = 200
    src, dst = random_coo(n, 1500, seed=100 + f)
    tg, g = ref_and_gts(src, dst, n)
    gd = g.to(DEV)
    x = _feat(n, f, seed=f, integer=True)         # integer data: ties everywhere
    gout = _feat(n, f, seed=f + 9, integer=True)  # integer grads: sums are exact in any order
    xr = x.clone().requires_grad_(True)
    torch_ref.spmm_max(tg, xr).backward(gout)
    xd = x.to(DEV).requires_grad_(True)
    ops.spmm_max(gd, xd).backward(gout.to(DEV))
    assert torch.equal(xd.grad.cpu(), xr.grad)
    # fused ReLU mask: gradient zeroed where the (ReLU-ed) source is <= 0
    out, arg = ops.spmm_max_fwd(gd, x.to(DEV))
    masked = ops.spmm_max_bwd(gd, gout.to(DEV), arg, relu_src=x.to(DEV))
    assert torch.equal(masked.cpu(), xr.grad * (x > 0))


# ------------------------------------------------------------------ K3 / K4
@pytest.mark.parametrize("mode", ["sum", "mean", "gcn"])
@pytest.mark.parametrize("f", [1, 4, 8, 20, 256, 300])
def test_sum_family_forward_backward_bit_exact(mode, f):
    n = 257
    src, dst = random_coo(n, 1800, seed=f)
    dst[dst == 5] = 6
    tg, g = ref_and_gts(src, dst, n)
    ref_fn = {"sum": torch_ref.spmm_sum, "mean": torch_ref.spmm_mean, "gcn": torch_ref.spmm_gcn}[mode]
    x = _feat(n, f, seed=f + 3)
    gout = _feat(n, f, seed=f + 4)
    xr = x.clone().requires_grad_(True)
    yr = ref_fn(tg, xr)
    yr.backward(gout)
    xd = x.to(DEV).requires_grad_(True)
    yd = ops.spmm_reduce(g.to(DEV), xd, mode)
    yd.backward(gout.to(DEV))
    assert torch.equal(yd.detach().cpu(), yr.detach()), "forward differs"
    assert torch.equal(xd.grad.cpu(), xr.grad), "backward differs"


@pytest.mark.parametrize("f", [4, 64, 256])
def test_relu_input_winner_record_equals_relu_mask_in_the_backward(f):
    """K1 with relu_input drops the winners whose maximum is not positive; the backward on that
    record equals the backward on the plain record followed by the ReLU' mask of the source
    (bit for bit), and the forward values are untouched."""
    src, dst = random_coo(900, 5000, seed=f)
    g = gts.Graph(src, dst, 900).to(DEV)
    p = torch.relu(torch.randn(900, f, device=DEV))             # about half the entries are exactly 0
    p[17] = 0                                                   # a whole row without any positive value
    gout = torch.randn(900, f, device=DEV)
    out_plain, arg_plain = ops.spmm_max_fwd(g, p)
    out_relu, arg_relu = ops.spmm_max_fwd(g, p, relu_input=True)
    assert torch.equal(out_plain, out_relu)
    none = 0xFF if arg_plain.dtype == torch.uint8 else -1
    changed = arg_plain != arg_relu
    assert changed.any() and torch.all(arg_relu[changed] == none) and torch.all(out_plain[changed] == 0)
    want = ops.spmm_max_bwd(g, gout, arg_plain, relu_src=p)
    assert torch.equal(ops.spmm_max_bwd(g, gout, arg_relu), want)


def test_sum_order_sensitivity_case():
    src = np.array([0, 1, 2]); dst = np.array([3, 3, 3])
    _, g = ref_and_gts(src, dst, 4)
    x = torch.tensor([[1e8], [1.0], [-1e8], [0.0]])
    assert ops.spmm_reduce(g.to(DEV), x.to(DEV), "sum")[3, 0].item() == 0.0


# ------------------------------------------------------------------ K5-K8
@pytest.mark.parametrize("heads,dim", [(1, 4), (2, 3), (4, 16), (3, 64), (4, 256), (2, 320)])
def test_gat_aggregate_matches_oracle(heads, dim):
    n = 150
    src, dst = random_coo(n, 700, seed=heads * 1000 + dim, min_in_degree=1)
    tg, g = ref_and_gts(src, dst, n)
    gen = torch.Generator().manual_seed(dim)
    ft = torch.randn(n, heads, dim, generator=gen)
    el = torch.randn(n, heads, generator=gen)
    er = torch.randn(n, heads, generator=gen)
    gout = torch.randn(n, heads, dim, generator=gen)
    r = [t.clone().requires_grad_(True) for t in (ft, el, er)]
    out_ref, a_ref = torch_ref.gat_aggregate(tg, r[0], r[1], r[2], 0.2)
    out_ref.backward(gout)
    d = [t.to(DEV).requires_grad_(True) for t in (ft, el, er)]
    out = ops.gat_aggregate(g.to(DEV), d[0], d[1], d[2], 0.2)
    out.backward(gout.to(DEV))
    tol = dict(rtol=1e-5, atol=1e-5)
    assert torch.allclose(out.detach().cpu(), out_ref.detach(), **tol)
    for got, want, name in zip(d, r, ("ft", "el", "er")):
        assert torch.allclose(got.grad.cpu(), want.grad, rtol=1e-4, atol=2e-5), name


@pytest.mark.parametrize("heads,dim", [(4, 256), (2, 64), (1, 512)])
def test_gat_aggregate_hubs_and_isolated_sources(heads, dim):
    """Rows on both sides of the 64-edge limit of the one-lane-per-edge path (dim >= 256: one wave per (node, head)):
    a node with 100 in-edges, one with 80 out-edges, 9 .. 64 edges (several 8-edge chunks), and a node nobody
    points at / that points at nobody (the backward's empty out-row)."""
    n = 220
    src, dst = random_coo(n, 900, seed=heads + dim, min_in_degree=1)
    keep = (src != 7) & (dst != 7)                                 # node 7: no in-edge from others, no out-edge
    src, dst = src[keep], dst[keep]
    extra_src = np.concatenate([np.arange(20, 120), np.full(80, 3), np.arange(130, 170), [7]])
    extra_dst = np.concatenate([np.full(100, 5), np.arange(125, 205), np.full(40, 9), [7]])      # 7 -> 7 self-loop only
    src, dst = np.concatenate([src, extra_src]), np.concatenate([dst, extra_dst])
    tg, g = ref_and_gts(src, dst, n)
    assert g.max_in_degree >= 100 and g.min_in_degree >= 1
    gen = torch.Generator().manual_seed(dim)
    ft, el, er = torch.randn(n, heads, dim, generator=gen), torch.randn(n, heads, generator=gen), torch.randn(n, heads, generator=gen)
    gout = torch.randn(n, heads, dim, generator=gen)
    r = [t.clone().requires_grad_(True) for t in (ft, el, er)]
    out_ref, _ = torch_ref.gat_aggregate(tg, r[0], r[1], r[2], 0.2)
    out_ref.backward(gout)
    d = [t.to(DEV).requires_grad_(True) for t in (ft, el, er)]
    out = ops.gat_aggregate(g.to(DEV), d[0], d[1], d[2], 0.2)
    out.backward(gout.to(DEV))
    assert torch.allclose(out.detach().cpu(), out_ref.detach(), rtol=1e-5, atol=1e-5)
    for got, want, name in zip(d, r, ("ft", "el", "er")):
        assert torch.allclose(got.grad.cpu(), want.grad, rtol=1e-4, atol=5e-5), name


@pytest.mark.parametrize("dim", [256, 64, 3])
def test_gat_elu_is_torchs_expm1_to_a_few_ulp_at_every_magnitude(dim):
    """GATConv's activation (model/networks.py:52: F.elu, which torch computes with expm1).  The kernels' own form (csrc/gts_rows.h
    elu_expm1: a series near zero, v_exp_f32 below -0.35) on a graph of self-loops only — one in-edge of weight exactly 1, so
    out = elu(ft) — against torch at magnitudes 1e-30 .. 100 and around the split: relative error below 4 ulp (5e-7)."""
    n, heads = 4096, 2
    idx = np.arange(n)
    g = gts.Graph(idx.astype("int32"), idx.astype("int32"), n)
    mags = torch.logspace(-30, 2, n * heads * dim, dtype=torch.float64)
    ft = -mags.to(torch.float32).reshape(n, heads, dim)
    ft[:, :, 0] = torch.linspace(-0.36, -0.34, n).reshape(n, 1)                  # around the split
    ft[0, 0, :3] = torch.tensor([float("-inf"), 0.0, 5.0])
    el = torch.zeros(n, heads)
    out, attn = ops._gat_fwd(g.to(DEV), ft.to(DEV), el.to(DEV), el.to(DEV), 0.2, activation=1)
    assert torch.equal(attn.cpu(), torch.ones(n, heads))
    want = torch.nn.functional.elu(ft.double())
    got = out.cpu().double()
    assert got[0, 0, 0].item() == -1.0 and got[0, 0, 1].item() == 0.0 and got[0, 0, 2].item() == 5.0
    finite = torch.isfinite(want)
    rel = ((got - want).abs() / want.abs().clamp_min(1e-300))[finite & (want != 0)]
    assert rel.max().item() < 5e-7, rel.max().item()
    nan_in = ft.clone()
    nan_in[1, 1, 1] = float("nan")
    out2, _ = ops._gat_fwd(g.to(DEV), nan_in.to(DEV), el.to(DEV), el.to(DEV), 0.2, activation=1)
    assert torch.isnan(out2[1, 1, 1]).item() and torch.isnan(out2).sum().item() == 1


# ------------------------------------------------------------------ K12
def test_projection_matches_reference_fixture(golden_dir):
    fx = np.load(os.path.join(golden_dir, "ref_project.npz"))
    svs = torch.from_numpy(fx["svs"]).to(DEV)
    lab = ops.project_rows(svs, torch.from_numpy(fx["labels"]).to(DEV), torch.zeros(1, dtype=torch.int64, device=DEV))
    assert lab.dtype == torch.int64 and np.array_equal(lab.cpu().numpy(), fx["out_labels"])
    logit = ops.project_rows(svs, torch.from_numpy(fx["logits"]).to(DEV),
                             torch.tensor(fx["background"][0], dtype=torch.float32, device=DEV))
    assert np.array_equal(logit.cpu().numpy().astype(np.float64), fx["out_logits"])
    # the drop-in wrappers (numpy in -> numpy out, reference dtypes)
    from data_processing import graph_io
    out = graph_io.project_nodes_to_img(fx["svs"], fx["labels"])
    assert out.dtype == fx["out_labels"].dtype and np.array_equal(out, fx["out_labels"])
    out = graph_io.project_node_logits_to_img(fx["svs"], fx["logits"], fx["background"].tolist())
    assert out.dtype == np.float64 and np.array_equal(out, fx["out_logits"])


@pytest.mark.parametrize("shape", [(1,), (7,), (513,), (33, 17, 9), (64, 64, 40)])
def test_projection_ragged_sizes_bit_exact(shape):
    rng = np.random.default_rng(sum(shape))
    n_nodes = 1000
    svs = rng.integers(-1, n_nodes, size=shape).astype(np.int16)
    labels = rng.integers(0, 4, size=n_nodes).astype(np.int64)
    logits = rng.standard_normal((n_nodes, 4)).astype(np.float32)
    sv = torch.from_numpy(svs).to(DEV)
    got = ops.project_rows(sv, torch.from_numpy(labels).to(DEV), torch.zeros(1, dtype=torch.int64, device=DEV))
    assert np.array_equal(got.cpu().numpy(), graph_ref.project_nodes_to_img_ref(svs, labels))
    got = ops.project_rows(sv, torch.from_numpy(logits).to(DEV), torch.tensor([1.0, -1, -1, -1], device=DEV))
    assert np.array_equal(got.cpu().numpy().astype(np.float64), graph_ref.project_logits_to_img_ref(svs, logits))
    relabel = torch.tensor([0, 2, 1, 4], dtype=torch.int16, device=DEV)
    pred = ops.project_argmax(sv, torch.from_numpy(logits).to(DEV), relabel).cpu().numpy()
    node_pred = torch.max(torch.from_numpy(logits), dim=1)[1].numpy()
    want = graph_ref.swap_labels_to_brats_ref(graph_ref.project_nodes_to_img_ref(svs, node_pred))
    assert pred.dtype == np.int16 and np.array_equal(pred, want)


def test_projection_unaligned_view_and_empty():
    base = torch.randint(-1, 50, (1001,), dtype=torch.int16, device=DEV)
    table = torch.arange(50, dtype=torch.int64, device=DEV) * 3
    bg = torch.tensor([-7], dtype=torch.int64, device=DEV)
    view = base[1:].contiguous()              # fresh allocation; still exercises odd length
    got = ops.project_rows(view, table, bg).cpu().numpy()
    want = np.append(table.cpu().numpy(), -7)[view.cpu().numpy()]
    assert np.array_equal(got, want)
    assert ops.project_rows(base[:0], table, bg).numel() == 0


# ------------------------------------------------------------------ full-size properties (C2 / C5)
@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 1000, 2049, 100003])
def test_label_confusion_bit_exact(n):
    rng = np.random.default_rng(n)
    pred = rng.integers(-3, 8, n).astype(np.int16)
    true = rng.choice(np.array([0, 1, 2, 3, 4, -1, 32767, -32768], dtype=np.int16), n)
    got = ops.label_confusion(torch.from_numpy(pred).to(DEV), torch.from_numpy(true).to(DEV))
    assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), graph_ref.label_confusion_ref(pred, true))


def test_label_confusion_unaligned_views_and_errors():
    rng = np.random.default_rng(5)
    pred = torch.from_numpy(rng.integers(0, 4, 5003).astype(np.int16)).to(DEV)
    true = torch.from_numpy(rng.integers(0, 4, 5003).astype(np.int16)).to(DEV)
    for a, b in ((1, 0), (0, 3), (5, 5)):        # 2-byte-aligned starts: the scalar path
        p, t = pred[a:a + 4990], true[b:b + 4990]
        want = graph_ref.label_confusion_ref(p.cpu().numpy(), t.cpu().numpy())
        assert np.array_equal(ops.label_confusion(p, t).cpu().numpy(), want)
    with pytest.raises(gts.GtsError):
        ops.label_confusion(pred, true[:-1])
    with pytest.raises(gts.GtsError):
        ops.label_confusion(pred.int(), true.int())


def test_label_confusion_reference_fixture_and_full_size(golden_dir):
    from model import evaluation

    fx = np.load(os.path.join(golden_dir, "ref_evaluation.npz"))
    for i in range(3):
        pred, true = fx[f"pred{i}"].astype(np.int16), fx[f"true{i}"].astype(np.int16)
        table = ops.label_confusion(torch.from_numpy(pred).to(DEV), torch.from_numpy(true).to(DEV)).cpu().numpy()
        assert np.array_equal(np.array(evaluation.dices_from_confusion(table), dtype=np.float64), fx[f"brats{i}"][:3])
        assert np.array_equal(evaluation.label_counts_from_confusion(table)[:4], fx[f"counts{i}"])
    # C5-sized volume: 240^3 voxels, predictions projected from node arg-max
    svs = torch.from_numpy(synth.supervoxel_volume((240, 240, 240), cube=10, shell=20)).to(DEV)
    rng = np.random.default_rng(1)
    logits = torch.from_numpy(rng.standard_normal((15000, 4)).astype(np.float32)).to(DEV)
    truth = torch.from_numpy(rng.choice(4, size=240 ** 3, p=[0.85, 0.07, 0.05, 0.03]).astype(np.int16)).to(DEV)
    vox = ops.project_argmax(svs, logits)
    table = ops.label_confusion(vox, truth.view(240, 240, 240)).cpu().numpy()
    assert table.sum() == 240 ** 3
    assert np.array_equal(table, graph_ref.label_confusion_ref(vox.cpu().numpy(), truth.cpu().numpy()))


def test_full_size_max_properties_c2():
    """N_b = 60 000, F = 256 (4 lattice graphs): checked against torch's own GPU reducers."""
    g = gts.batch([synth.lattice_graph() for _ in range(4)]).to(DEV)
    assert g.n == 60000 and g.number_of_edges() == 4 * 86350
    x = torch.randn(g.n, 256, device=DEV)
    out, arg = ops.spmm_max_fwd(g, x)
    d = g.dev()
    dst = torch.repeat_interleave(torch.arange(g.n, device=DEV), (d.indptr[1:] - d.indptr[:-1]).long())
    want = torch.full_like(x, float("-inf")).scatter_reduce(
        0, dst[:, None].expand(-1, 256), x[d.indices.long()], "amax", include_self=True)
    assert torch.equal(out, want)                           # bit-exact maximum
    pos = d.indptr[:-1].long()[:, None] + arg.long()         # argmax consistency: x[arg] == out
    assert torch.equal(x[d.indices.long()[pos], torch.arange(256, device=DEV)[None, :]], out)
    # idempotence of the reducer on a constant field and linearity of the sum reducer
    ones = torch.ones(g.n, 256, device=DEV)
    assert torch.equal(ops.spmm_max_fwd(g, ones, want_arg=False)[0], ones)
    deg = (d.indptr[1:] - d.indptr[:-1]).float()[:, None]
    assert torch.equal(ops.spmm_sum_raw(g, ones), deg.expand(-1, 256))
    y = torch.randn(g.n, 256, device=DEV)
    lhs = ops.spmm_sum_raw(g, x + y)
    rhs = ops.spmm_sum_raw(g, x) + ops.spmm_sum_raw(g, y)
    assert torch.allclose(lhs, rhs, rtol=1e-5, atol=1e-5)
    # backward conserves mass: every gout element lands on exactly one source
    gout = torch.rand(g.n, 256, device=DEV)
    gx = ops.spmm_max_bwd(g, gout, arg)
    assert torch.allclose(gx.sum(0), gout.sum(0), rtol=1e-4)


def test_full_size_projection_c5():
    vol = synth.supervoxel_volume((240, 240, 240), cube=10, shell=20)
    logits = torch.randn(15000, 4, device=DEV)
    sv = torch.from_numpy(vol).to(DEV)
    out = ops.project_rows(sv, logits, torch.tensor([1.0, -1, -1, -1], device=DEV))
    assert out.shape == (240, 240, 240, 4)
    table = torch.cat([logits, torch.tensor([[1.0, -1, -1, -1]], device=DEV)])
    assert torch.equal(out, table[sv.long()])
    assert torch.equal(out[0, 0, 0], torch.tensor([1.0, -1, -1, -1], device=DEV))


# ------------------------------------------------------------------ weighted cross-entropy
@pytest.mark.parametrize("n,c,weighted", [(1, 4, True), (1000, 4, True), (60000, 4, True), (777, 4, False),
                                          (513, 7, True), (64, 1, True), (300, 32, True)])
def test_weighted_cross_entropy_matches_torch(n, c, weighted):
    import torch.nn.functional as F

    gen = torch.Generator().manual_seed(n + c)
    x = torch.randn(n, c, generator=gen) * 3
    y = torch.randint(0, c, (n,), generator=gen)
    w = (torch.rand(c, generator=gen) + 0.1) if weighted else None
    x64 = x.double().requires_grad_(True)
    for reduction in ("mean", "sum"):
        want = F.cross_entropy(x64, y, weight=None if w is None else w.double(), reduction=reduction)
        (gw,) = torch.autograd.grad(want, x64)
        xd = x.to(DEV).requires_grad_(True)
        got = ops.weighted_cross_entropy(xd, y.to(DEV), None if w is None else w.to(DEV), reduction=reduction)
        (gg,) = torch.autograd.grad(got, xd)
        assert abs(float(got) - float(want)) <= 2e-6 * max(1.0, abs(float(want)))
        assert torch.allclose(gg.cpu().double(), gw, rtol=1e-5, atol=1e-6 * max(1.0, float(gw.abs().max())))
    a = ops.weighted_cross_entropy(x.to(DEV), y.to(DEV), None if w is None else w.to(DEV))
    b = ops.weighted_cross_entropy(x.to(DEV), y.to(DEV), None if w is None else w.to(DEV))
    assert torch.equal(a, b)                      # deterministic reduction


def test_device_csr_is_owned_by_the_device_views_not_by_the_cached_host_graph():
    """A dataset caches host graphs; their device CSRs must go away with the last device view."""
    import gc
    import weakref

    host = gts.Graph(*random_coo(200, 900, seed=5), 200)
    view = host.to(DEV)
    csr = view.dev()
    assert host.to(DEV).dev() is csr and view.dev() is csr      # one upload, shared while in use
    alive = weakref.ref(csr)
    del view, csr
    gc.collect()
    assert alive() is None and host.device.type == "cpu"
    assert host.to(DEV).dev().indptr.numel() == 201             # and it is rebuilt on demand


@pytest.mark.parametrize("n,heads,dim,k", [(60000, 4, 256, 1024), (60000, 4, 64, 4), (49999, 2, 128, 132), (300, 4, 256, 64),
                                          (60000, 3, 48, 64)])
def test_gat_projection_with_scores_in_the_gemm_epilogue(n, heads, dim, k):
    """gts_gat_fc_scores_f32: ft bitwise the plain GEMM's; el / er within fp32 rounding of the separate pass
    (gat_scores) and of fp64 — fused epilogue at the tall shapes with dim % 64 == 0, GEMM + gat_scores otherwise."""
    from gts import dense

    gen = torch.Generator().manual_seed(n + dim)
    h = torch.randn(n, k, generator=gen).to(DEV)
    w = (torch.randn(heads * dim, k, generator=gen) * 0.1).to(DEV)
    al, ar = torch.randn(heads, dim, generator=gen).to(DEV), torch.randn(heads, dim, generator=gen).to(DEV)
    ft, el, er = ops.gat_fc_scores(h, w, al, ar, heads, dim)
    want_ft = dense.linear_fwd(h, w).view(n, heads, dim)
    assert torch.equal(ft, want_ft)
    el2, er2 = ops.gat_scores(want_ft, al, ar)
    ref_l = (want_ft.double() * al.double()).sum(-1)
    ref_r = (want_ft.double() * ar.double()).sum(-1)
    bound = (want_ft.double().abs() * al.double().abs()).sum(-1)
    assert torch.all((el.double() - ref_l).abs() <= 2e-6 * bound + 1e-30)
    assert torch.all((er.double() - ref_r).abs() <= 2e-6 * (want_ft.double().abs() * ar.double().abs()).sum(-1) + 1e-30)
    assert torch.all((el2.double() - ref_l).abs() <= 2e-6 * bound + 1e-30)        # the separate pass meets the same bar
