"""Oracle parity and size-independent properties at the REAL shapes of BASELINE.json's configs:

  C2  7xSAGE-pool-256 on 4 lattice graphs (N_b = 60 000): full forward + loss + every gradient
      against the fp32 and fp64 oracle; the 19-problem deferred weight-gradient launch at
      M = 60 000 (13 splits of 144 reduction tiles, double-buffered 256x256 tile) against fp64.
  C3  GAT layer_sizes=[256]*4, heads=[4]*4 (5 GATConv: 4->4x256, 3x(1024->4x256), 1024->1x4,
      /root/reference/model/networks.py:40-66) against the oracle on a 1.5k-node graph, and the
      K5-K8 kernels at N_b = 60 000, H = 4, D = 256 through properties.
  C4  the per-GPU shape of the 8-GPU run (8 graphs, N_b = 120 000): K1 / K2 / sum properties,
      K11 against fp64, and block-diagonal independence (batch of 8 == two batches of 4).

Tolerances as in tests/test_gpu_model.py (fp32): logits rtol 1e-4 / atol 1e-4 x scale; gradients
judged against the fp64 oracle (the GPU must be as close to it as the CPU fp32 oracle x10, floor
1e-3 of the gradient scale)."""
from collections import namedtuple

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import gts
from gts import dense, ops, synth
from model.networks import init_graph_net
from oracle import graph_ref, torch_ref
from tests.helpers import copy_state, random_coo, ref_and_gts

pytestmark = pytest.mark.gpu
DEV = "cuda"
HP = namedtuple("HP", "in_feats out_classes layer_sizes gat_heads gat_residuals")
CLASS_W = [0.1, 1.0, 2.0, 2.0]


@pytest.fixture(scope="module", autouse=True)
def _lib(hip_lib):
    assert torch.cuda.is_available()
    return hip_lib


def _net_triple(model_type, hp, seed):
    """(fp32 oracle, fp64 oracle, product on the GPU) with identical weights."""
    torch.manual_seed(seed)
    ref = torch_ref.ref_init_graph_net(model_type, hp)
    ref64 = torch_ref.ref_init_graph_net(model_type, hp).double()
    ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
    mine = init_graph_net(model_type, hp)
    assert list(mine.state_dict()) == list(ref.state_dict())
    copy_state(mine, ref)
    return ref, ref64, mine.to(DEV)


def _check_against_oracles(ref, ref64, mine, tg, g, x, y):
    w = torch.tensor(CLASS_W)
    lr = ref(tg, x)
    loss_r = F.cross_entropy(lr, y, weight=w)
    loss_r.backward()
    l64 = ref64(tg, x.double())
    F.cross_entropy(l64, y, weight=w.double()).backward()
    lm = mine(g.to(DEV), x.to(DEV))
    loss_m = F.cross_entropy(lm, y.to(DEV), weight=w.to(DEV))
    loss_m.backward()
    scale = max(1.0, float(lr.abs().max()))
    diff = (lm.detach().cpu().double() - lr.detach().double()).abs()
    assert torch.all(diff <= 1e-4 * lr.detach().double().abs() + 1e-4 * scale), f"logits differ by {diff.max():.3e}"
    err_gpu = float((lm.detach().cpu().double() - l64).abs().max())
    err_cpu = float((lr.detach().double() - l64).abs().max())
    print(f"logits: |gpu-fp64|={err_gpu:.2e} |cpu32-fp64|={err_cpu:.2e} scale {scale:.2e}")
    assert err_gpu < 10 * max(err_cpu, 1e-6 * scale)
    assert abs(float(loss_m) - float(loss_r)) < 1e-4 * max(1.0, abs(float(loss_r)))
    for (name, p), (_, q), (_, q64) in zip(mine.named_parameters(), ref.named_parameters(),
                                            ref64.named_parameters()):
        s = max(float(q64.grad.abs().max()), 1e-6)
        e_gpu = float((p.grad.cpu().double() - q64.grad).abs().max())
        e_cpu = float((q.grad.double() - q64.grad).abs().max())
        assert e_gpu < max(10 * e_cpu, 1e-3 * s), f"{name}: gpu {e_gpu:.3e} cpu {e_cpu:.3e} scale {s:.3e}"
        assert e_gpu < 1e-2 * s, f"{name}: {e_gpu:.3e} vs scale {s:.3e}"


def _lattice_batch(n_graphs):
    parts = [synth.make_sample(i, kind="lattice", in_feats=4) for i in range(n_graphs)]
    g = gts.batch([p[1] for p in parts])
    x = torch.from_numpy(np.concatenate([p[2] for p in parts]))
    y = torch.from_numpy(np.concatenate([p[3] for p in parts]))
    return g, x, y


# ---------------------------------------------------------------------------------- C3
@pytest.mark.timeout(900)
def test_c3_gat_4x4x256_network_matches_oracle():
    """The network BASELINE config 3 names: 5 GATConv, widths 4 -> 1024 -> 1024 -> 1024 -> 1024 -> 4."""
    hp = HP(4, 4, [256] * 4, [4] * 4, [False] * 4)
    n = 1500
    src, dst = random_coo(n, 8000, seed=33, min_in_degree=1)
    tg, g = ref_and_gts(src, dst, n)
    ref, ref64, mine = _net_triple("GAT", hp, seed=2)
    assert [tuple(l.fc.weight.shape) for l in mine.layers] == [(1024, 4), (1024, 1024), (1024, 1024),
                                                               (1024, 1024), (4, 1024)]
    x = torch.from_numpy(synth.node_features(n, 4, 9))
    y = torch.from_numpy(synth.node_labels(n, 9))
    _check_against_oracles(ref, ref64, mine, tg, g, x, y)


@pytest.mark.timeout(1500)
def test_c3_network_on_the_clustered_gat_kernels_matches_oracle():
    """The path `bench.py --config c3` actually runs: two 15 000-node lattice graphs (30 000 rows, in-degree <= 6) are
    sent through the CLUSTERED GATConv kernels by the default rules (gts/schedule.py: D = 256, >= 20 000 rows, rows of one
    8-edge chunk) — forward, source pass and edge pass of all four hidden layers — and the logits, the loss and every
    gradient are laid beside the fp32 and fp64 oracle, not only beside the plain kernels."""
    from gts import schedule

    assert (schedule.ENABLED_GAT, schedule.MIN_ROWS_GAT, schedule.MAX_DEGREE_GAT) == (True, 20000, 8), "default rules"
    hp = HP(4, 4, [256] * 4, [4] * 4, [False] * 4)
    g, x, y = _lattice_batch(2)
    assert g.n == 30000 and g.max_in_degree <= 6
    gd = g.to(DEV)
    for which in ("gat_in", "gat_edge_in", "gat_out"):
        assert ops._gat_cluster_schedule(gd, which, g.n, 4, 256) is not None, which
    calls = []
    lib = gts._lib.load()
    real = {name: getattr(lib, name) for name in ("gts_gat_fwd_cluster_f32", "gts_gat_bwd_edge_cluster_f32",
                                                  "gts_gat_bwd_src_cluster_f32")}

    class Spy:            # counts the clustered launches without changing them
        def __init__(self, name):
            self.name = name

        def __call__(self, *args):
            calls.append(self.name)
            return real[self.name](*args)
    tg = torch_ref.TGraph(graph_ref.RefGraph(g.src, g.dst, g.n))
    ref, ref64, mine = _net_triple("GAT", hp, seed=4)
    try:
        for name in real:
            setattr(lib, name, Spy(name))
        _check_against_oracles(ref, ref64, mine, tg, g, x, y)
    finally:
        for name, fn in real.items():
            setattr(lib, name, fn)
    # four hidden layers (H * D = 1024) each way; the one-head classifier layer keeps the plain kernels
    assert calls.count("gts_gat_fwd_cluster_f32") == 4
    assert calls.count("gts_gat_bwd_edge_cluster_f32") == 4 and calls.count("gts_gat_bwd_src_cluster_f32") == 4


@pytest.mark.timeout(900)
def test_gat_kernels_full_size_properties_c3():
    """K5-K8 at N_b = 60 000, E_b = 345 400, H = 4, D = 256 (one C3 hidden layer)."""
    g, _, _ = _lattice_batch(4)
    g = g.to(DEV)
    n, e, h, d = g.n, g.number_of_edges(), 4, 256
    assert (n, e) == (60000, 4 * 86350)
    gen = torch.Generator(device=DEV).manual_seed(0)
    ft = torch.randn(n, h, d, device=DEV, generator=gen)
    el = torch.randn(n, h, device=DEV, generator=gen)
    er = torch.randn(n, h, device=DEV, generator=gen)
    out, attn = ops._gat_fwd(g, ft, el, er, 0.2)
    dcsr = g.dev()
    deg = (dcsr.indptr[1:] - dcsr.indptr[:-1]).long()
    dst_of_slot = torch.repeat_interleave(torch.arange(n, device=DEV), deg)
    # (1) attention over the in-edges of every destination sums to 1, per head; all in (0, 1]
    sums = torch.zeros(n, h, device=DEV).index_add_(0, dst_of_slot, attn)
    assert torch.allclose(sums, torch.ones_like(sums), rtol=0, atol=1e-5)
    assert float(attn.min()) > 0.0 and float(attn.max()) <= 1.0
    # (2) the attention equals torch's own softmax of the same scores on a sample of rows, and the
    #     output equals the attention-weighted sum it defines
    scores = F.leaky_relu(el[dcsr.indices.long()] + er[dst_of_slot], 0.2)
    rows = torch.randint(0, n, (2000,), generator=torch.Generator().manual_seed(1)).tolist()
    ptr = dcsr.indptr.cpu().tolist()
    for v in rows[:300]:
        a, b = ptr[v], ptr[v + 1]
        want = torch.softmax(scores[a:b], dim=0)
        assert torch.allclose(attn[a:b], want, rtol=1e-5, atol=1e-6)
    want_out = torch.zeros_like(ft).index_add_(0, dst_of_slot, attn[:, :, None] * ft[dcsr.indices.long()])
    assert torch.allclose(out, want_out, rtol=1e-5, atol=1e-5)
    # (3) uniform scores => plain neighbourhood mean (K3 with division), to rounding
    zero = torch.zeros(n, h, device=DEV)
    out_u, attn_u = ops._gat_fwd(g, ft, zero, zero, 0.2)
    mean = ops.spmm_sum_raw(g, ft.view(n, h * d), div_out=dcsr.deg_clamped).view(n, h, d)
    assert torch.allclose(out_u, mean, rtol=1e-5, atol=1e-5)
    assert torch.allclose(attn_u, (1.0 / deg.float())[dst_of_slot][:, None].expand(-1, h), rtol=1e-6, atol=0)
    # (4) backward: the feature gradient conserves mass (sum_u gft[u] = sum_v gout[v], because every
    #     destination's attention sums to 1) and softmax gradients sum to 0 per destination
    gout = torch.rand(n, h, d, device=DEV, generator=gen)
    gft, gel, ger = ops._gat_bwd(g, ft, el, er, attn, gout, 0.2)
    assert torch.allclose(gft.sum(0), gout.sum(0), rtol=1e-4, atol=1e-2)
    gft_u, gel_u, ger_u = ops._gat_bwd(g, ft, zero, zero, attn_u, gout, 0.2)
    gscale = float(gout.abs().max()) * d
    assert float(ger_u.abs().max()) < 1e-5 * gscale        # uniform leaky' => sum_e dsoftmax_e = 0
    assert abs(float(gel_u.sum())) < 1e-4 * gscale * n ** 0.5
    want_gft_u = ops.spmm_sum_raw(g, gout.view(n, h * d), transposed=True, div_in=dcsr.deg_clamped)
    assert torch.allclose(gft_u.view(n, h * d), want_gft_u, rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------- C2
@pytest.mark.timeout(1200)
def test_c2_full_size_forward_loss_and_gradients_match_oracle():
    """The headline workload itself: [256]*7 (8 SAGEConv), 4 x 15 000-node lattice graphs, in 4."""
    hp = HP(4, 4, [256] * 7, None, None)
    g, x, y = _lattice_batch(4)
    assert g.n == 60000
    tg = torch_ref.TGraph(graph_ref.batch_ref([graph_ref.RefGraph(p.src, p.dst, p.n)
                                               for p in [synth.lattice_graph()] * 4]))
    ref, ref64, mine = _net_triple("GSpool", hp, seed=0)
    _check_against_oracles(ref, ref64, mine, tg, g, x, y)


@pytest.mark.timeout(600)
def test_c2_backward_is_the_same_with_bit_masks_float_masks_and_generic_epilogues(hip_lib):
    """The switches of the 256-wide layer launches change where the ReLU masks come from (bits recorded by the
    forward GEMM vs the floats of the saved activation) and which instantiation runs (compile-time vs run-time
    epilogue), when a wave asks for its operand fragments (one load per eight MFMAs vs nine in a burst) and which
    kernel walks the weight-gradient tiles (wgrad_stream_kernel vs the LDS tile), never a sum: loss and every gradient
    are bitwise the same."""
    from gts import nn as gnn

    hp = HP(4, 4, [256] * 7, None, None)
    g, x, y = _lattice_batch(4)
    torch.manual_seed(0)
    mine = init_graph_net("GSpool", hp).to(DEV)
    gd, xd, yd, w = g.to(DEV), x.to(DEV), y.to(DEV), torch.tensor(CLASS_W, device=DEV)

    def run():
        mine.zero_grad(set_to_none=True)
        loss = F.cross_entropy(mine(gd, xd), yd, weight=w)
        loss.backward()
        return float(loss), [p.grad.clone() for p in mine.parameters()]

    base = run()
    assert all(torch.isfinite(t).all() for t in base[1])
    try:
        gnn.RELU_MASK_BITS = False
        floats = run()
        gnn.RELU_MASK_BITS = True
        assert hip_lib.gts_set_option(7, 5) == 0          # GTS_OPT_GEMM_SCHED bit 4: generic epilogues
        generic = run()
        assert hip_lib.gts_set_option(7, 9) == 0          # bit 8: the fragment loads of a reduction group in one burst (rounds 1 - 2)
        burst = run()
        assert hip_lib.gts_set_option(7, 1) == 0
        assert hip_lib.gts_set_option(2, 4) == 0          # GTS_OPT_WGRAD_TILE 4: the LDS tile wgrad_stream_kernel (6, automatic) replaced
        lds_tile = run()
    finally:
        gnn.RELU_MASK_BITS = True
        hip_lib.gts_set_option(7, 1)
        hip_lib.gts_set_option(2, -1)
    for other in (floats, generic, burst, lds_tile):
        assert other[0] == base[0]
        for a, b in zip(base[1], other[1]):
            assert torch.equal(a, b)


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("m,count", [(60000, 19), (120000, 19)])
def test_deferred_weight_gradient_launch_at_full_size(hip_lib, m, count):
    """What _SagePoolStack.backward issues at C2 / at C4's per-GPU shape: 19 problems of
    [M, 256]^T [M, 256] in ONE split-reduction launch, against fp64, and bitwise repeatable."""
    assert hip_lib.gts_linear_bwd_weight_workspace(m, 256, 256, count) > 0
    gs = [torch.randn(m, 256, generator=torch.Generator().manual_seed(300 + q)) for q in range(count)]
    acts = [torch.randn(m, 256, generator=torch.Generator().manual_seed(400 + q)) for q in range(count)]
    dev = [(g_.to(DEV), a.to(DEV), q % 3 != 2) for q, (g_, a) in enumerate(zip(gs, acts))]
    out = dense.linear_bwd_weight_multi(dev)
    again = dense.linear_bwd_weight_multi(dev)
    for q, ((gw, gb), g_, a) in enumerate(zip(out, gs, acts)):
        want = g_.double().t() @ a.double()
        bound = g_.double().abs().t() @ a.double().abs()
        err = (gw.cpu().double() - want).abs()
        assert torch.all(err <= 2e-6 * bound), f"problem {q}: max err {err.max():.3e}"
        assert (gb is not None) == (q % 3 != 2)
        if gb is not None:
            errb = (gb.cpu().double() - g_.double().sum(0)).abs()
            assert torch.all(errb <= 2e-6 * g_.double().abs().sum(0))
        assert torch.equal(gw, again[q][0])


# ---------------------------------------------------------------------------------- C4 (per-GPU shape)
@pytest.mark.timeout(900)
def test_c4_per_gpu_shape_aggregation_properties():
    """8 graphs per GPU: N_b = 120 000, E_b = 690 800, F = 256 — K1, K2 and the sum reducer
    against torch's own GPU reducers."""
    g, _, _ = _lattice_batch(8)
    g = g.to(DEV)
    assert g.n == 120000 and g.number_of_edges() == 8 * 86350
    x = torch.randn(g.n, 256, device=DEV)
    out, arg = ops.spmm_max_fwd(g, x)
    d = g.dev()
    deg = (d.indptr[1:] - d.indptr[:-1]).long()
    dst = torch.repeat_interleave(torch.arange(g.n, device=DEV), deg)
    want = torch.full_like(x, float("-inf")).scatter_reduce(
        0, dst[:, None].expand(-1, 256), x[d.indices.long()], "amax", include_self=True)
    assert torch.equal(out, want)
    pos = d.indptr[:-1].long()[:, None] + arg.long()
    assert torch.equal(x[d.indices.long()[pos], torch.arange(256, device=DEV)[None, :]], out)
    gout = torch.rand(g.n, 256, device=DEV)
    gx = ops.spmm_max_bwd(g, gout, arg)
    # exact statement of the backward: gx[u, f] = sum of gout[v, f] over the v whose winner is u
    winner = d.indices.long()[pos]                                   # [N, 256] source ids
    want_gx = torch.zeros_like(gx).scatter_add_(0, winner, gout)
    assert torch.allclose(gx, want_gx, rtol=1e-5, atol=1e-5)
    assert torch.allclose(gx.sum(0), gout.sum(0), rtol=1e-4)
    ones = torch.ones(g.n, 256, device=DEV)
    assert torch.equal(ops.spmm_sum_raw(g, ones), deg.float()[:, None].expand(-1, 256))


@pytest.mark.timeout(900)
def test_c4_per_gpu_shape_gemms_against_fp64():
    m = 120000
    a0, a1 = (torch.randn(m, 256, generator=torch.Generator().manual_seed(s)) for s in (1, 2))
    w0, w1 = (torch.randn(256, 256, generator=torch.Generator().manual_seed(s)) for s in (3, 4))
    b = torch.randn(256, generator=torch.Generator().manual_seed(5))
    want = a0.double() @ w0.double().t() + a1.double() @ w1.double().t() + b.double()
    bound = a0.double().abs() @ w0.double().abs().t() + a1.double().abs() @ w1.double().abs().t() + b.double().abs()
    got = dense.linear_fwd(a0.to(DEV), w0.to(DEV), a1.to(DEV), w1.to(DEV), bias=b.to(DEV), relu=True)
    assert torch.all((got.cpu().double() - want.clamp(min=0)).abs() <= 2e-6 * bound)
    mask = torch.randn(m, 256, generator=torch.Generator().manual_seed(6))
    gin = dense.linear_bwd_input(a0.to(DEV), w0.to(DEV), a1.to(DEV), w1.to(DEV), relu_mask=mask.to(DEV))
    want = (a0.double() @ w0.double() + a1.double() @ w1.double()) * (mask > 0)
    bound = a0.double().abs() @ w0.double().abs() + a1.double().abs() @ w1.double().abs()
    assert torch.all((gin.cpu().double() - want).abs() <= 2e-6 * bound)


@pytest.mark.timeout(900)
def test_c4_batch_of_8_equals_two_batches_of_4():
    """Block-diagonal independence at the C4 per-GPU shape: logits AND parameter gradients of the
    8-graph batch are those of its two 4-graph halves (each of which is the C2 shape the oracle
    test above covers); sums of the two halves' loss numerators / denominators give the batch loss."""
    hp = HP(4, 4, [256] * 7, None, None)
    torch.manual_seed(0)
    net = init_graph_net("GSpool", hp).to(DEV)
    parts = [synth.make_sample(i, kind="lattice", in_feats=4) for i in range(8)]
    w = torch.tensor(CLASS_W, device=DEV)

    def run(items):
        g = gts.batch([p[1] for p in items]).to(DEV)
        x = torch.from_numpy(np.concatenate([p[2] for p in items])).to(DEV)
        y = torch.from_numpy(np.concatenate([p[3] for p in items])).to(DEV)
        net.zero_grad()
        logits = net(g, x)
        num, stats = ops.weighted_cross_entropy_stats(logits, y, w)
        num.backward()
        return logits.detach(), float(stats[0]), float(stats[1]), [p.grad.clone() for p in net.parameters()]

    whole, num, den, grads = run(parts)
    lo, num0, den0, g0 = run(parts[:4])
    hi, num1, den1, g1 = run(parts[4:])
    assert torch.equal(whole, torch.cat([lo, hi]))                   # rows never mix across graphs
    assert abs(num - (num0 + num1)) < 1e-5 * abs(num) and abs(den - (den0 + den1)) < 1e-5 * den
    for a, b0, b1 in zip(grads, g0, g1):
        s = max(float(a.abs().max()), 1e-6)
        assert float((a - (b0 + b1)).abs().max()) < 1e-4 * s
