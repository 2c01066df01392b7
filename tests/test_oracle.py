"""The oracle's own known-answer and self-consistency tests (CPU)."""
import numpy as np
import pytest
import torch

from oracle import graph_ref, torch_ref
from tests.helpers import HAND_X, hand_graph, random_coo


def test_in_csr_keeps_coo_order():
    g = hand_graph()
    assert g.indptr.tolist() == [0, 2, 3, 6, 6, 8, 9]
    assert g.indices.tolist() == [1, 2, 0, 0, 2, 3, 5, 1, 4]
    assert g.t_indptr.tolist() == [0, 2, 4, 6, 7, 8, 9]
    assert g.t_indices.tolist() == [1, 2, 0, 4, 0, 2, 2, 5, 4]


def test_max_known_answer():
    g = torch_ref.TGraph(hand_graph())
    out, arg = torch_ref.spmm_max_with_arg(g, torch.from_numpy(HAND_X))
    assert out.tolist() == [[3, 5], [1, 5], [7, 7], [0, 0], [3, 9], [2, 2]]
    assert arg.tolist() == [[1, 1], [0, 0], [3, 3], [-1, -1], [1, 5], [4, 4]]   # tie -> first (node 1)


def test_max_backward_goes_to_first_maximum_only():
    g = torch_ref.TGraph(hand_graph())
    x = torch.from_numpy(HAND_X).clone().requires_grad_(True)
    torch_ref.spmm_max(g, x).sum().backward()
    expect = np.zeros((6, 2), dtype=np.float32)
    for v_arg in [[1, 1], [0, 0], [3, 3], [1, 5], [4, 4]]:
        for f, u in enumerate(v_arg):
            expect[u, f] += 1
    assert np.array_equal(x.grad.numpy(), expect)


def test_max_inf_is_replaced_by_zero_and_blocks_gradient():
    g = torch_ref.TGraph(hand_graph())
    x = torch.from_numpy(HAND_X).clone()
    x[0, 0] = float("inf")            # node 1's only source
    x.requires_grad_(True)
    out, arg = torch_ref.spmm_max_with_arg(g, x)
    assert out[1, 0] == 0 and arg[1, 0] == -1
    out.sum().backward()
    assert x.grad[0, 0] == 0


def test_mean_and_gcn_known_answers():
    g = torch_ref.TGraph(hand_graph())
    x = torch.from_numpy(HAND_X)
    mean = torch_ref.spmm_mean(g, x)
    assert torch.allclose(mean, torch.tensor([[3, 2], [1, 5], [11 / 3, 11 / 3], [0, 0], [2.5, 7], [2, 2]]))
    gcn = torch_ref.spmm_gcn(g, x)
    assert torch.allclose(gcn[3], torch.tensor([7.0, 7.0]))
    assert torch.allclose(gcn[0], torch.tensor([7 / 3, 3.0]))


def test_sum_is_sequential_in_slot_order():
    """fp32 sums must be accumulated left to right in in-edge order (what the kernel does)."""
    src = np.array([0, 1, 2]); dst = np.array([3, 3, 3])
    g = torch_ref.TGraph(graph_ref.RefGraph(src, dst, 4))
    x = torch.tensor([[1e8], [1.0], [-1e8], [0.0]], dtype=torch.float32)
    # (1e8 + 1) - 1e8 = 0 in fp32 when added in order; a different order would give 1
    assert torch_ref.spmm_sum(g, x)[3, 0] == 0.0


@pytest.mark.parametrize("fn", [torch_ref.spmm_max, torch_ref.spmm_mean, torch_ref.spmm_gcn, torch_ref.spmm_sum])
def test_reducer_gradcheck_fp64(fn):
    src, dst = random_coo(9, 30, seed=3)
    g = torch_ref.TGraph(graph_ref.RefGraph(src, dst, 9))
    x = torch.randn(9, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(0)).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda t: fn(g, t), (x,), eps=1e-6, atol=1e-5)


def test_edge_softmax_rows_sum_to_one_and_gat_uniform_equals_mean():
    src, dst = random_coo(12, 40, seed=5, min_in_degree=1)
    g = torch_ref.TGraph(graph_ref.RefGraph(src, dst, 12))
    gen = torch.Generator().manual_seed(1)
    e = torch.randn(g.number_of_edges(), 3, generator=gen)
    a = torch_ref.edge_softmax(g, e)
    sums = torch.zeros(12, 3).index_add(0, g.dst_of_slot, a)
    assert torch.allclose(sums, torch.ones(12, 3), atol=1e-6)
    ft = torch.randn(12, 3, 5, generator=gen)
    zeros = torch.zeros(12, 3)
    out, _ = torch_ref.gat_aggregate(g, ft, zeros, zeros, 0.2)   # equal scores -> plain mean
    assert torch.allclose(out, torch_ref.spmm_mean(g, ft.reshape(12, 15)).reshape(12, 3, 5), atol=1e-6)


def test_gat_layer_gradcheck_fp64():
    src, dst = random_coo(7, 20, seed=9, min_in_degree=1)
    g = torch_ref.TGraph(graph_ref.RefGraph(src, dst, 7))
    torch.manual_seed(0)
    layer = torch_ref.RefGATConv(3, 2, 2, residual=True, activation=torch.nn.functional.elu).double()
    x = torch.randn(7, 3, dtype=torch.float64).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda t: layer(g, t), (x,), eps=1e-6, atol=1e-5)


def test_gat_rejects_zero_in_degree():
    g = torch_ref.TGraph(hand_graph())
    layer = torch_ref.RefGATConv(2, 2, 1)
    with pytest.raises(RuntimeError):
        layer(g, torch.from_numpy(HAND_X))


def test_state_dict_layout_and_depth():
    from collections import namedtuple

    HP = namedtuple("HP", "in_feats out_classes layer_sizes gat_heads gat_residuals")
    net = torch_ref.ref_init_graph_net("GSpool", HP(4, 4, [256] * 7, None, None))
    assert len(net.layers) == 8
    assert sum(p.numel() for p in net.parameters()) == 1_252_888
    keys = list(net.state_dict())
    assert keys[:5] == ["layers.0.bias", "layers.0.fc_pool.weight", "layers.0.fc_pool.bias",
                        "layers.0.fc_self.weight", "layers.0.fc_neigh.weight"]
    gat = torch_ref.ref_init_graph_net("GAT", HP(4, 4, [8, 8], [2, 3], [False, True]))
    assert gat.layers[1].fc.weight.shape == (24, 16) and gat.layers[2].fc.weight.shape == (4, 24)
    with pytest.raises(Exception, match="Unknown model type"):
        torch_ref.ref_init_graph_net("GSlstm", HP(4, 4, [8], None, None))


def test_lin_before_mp_branch_is_algebraically_the_same():
    """C1's 8->4 output layer applies fc_neigh before the mean; must equal after (fp64)."""
    src, dst = random_coo(10, 30, seed=2)
    g = torch_ref.TGraph(graph_ref.RefGraph(src, dst, 10))
    torch.manual_seed(0)
    layer = torch_ref.RefSAGEConv(8, 4, "mean").double()
    x = torch.randn(10, 8, dtype=torch.float64)
    after = layer.fc_self(x) + layer.fc_neigh(torch_ref.spmm_mean(g, x)) + layer.bias
    assert torch.allclose(layer(g, x), after, atol=1e-12)


# ------------------------------------------------------------------ independent dense formulation
# Guard against restatement slips in oracle/torch_ref.py (it does NOT lift "parity unpinned": DGL is
# still absent): every layer is recomputed in fp64 from a dense adjacency matrix — A[v, u] = number
# of edges u -> v, `A @ x`, a masked softmax over a dense [N, N, H] score tensor — which shares no
# code with the CSR / slot-order formulation of the oracle.
def _dense_case(seed, ties=False):
    n = 30
    src, dst = random_coo(n, 90, seed=seed)
    dst[dst == 7] = 8                                   # node 7: zero in-degree
    src, dst = np.concatenate([src, [3, 11, 11]]), np.concatenate([dst, [3, 11, 12]])   # self-loops (+1 edge)
    ref = graph_ref.RefGraph(src, dst, n)
    a = torch.zeros(n, n, dtype=torch.float64)
    a.index_put_((torch.from_numpy(dst), torch.from_numpy(src)), torch.ones(len(src), dtype=torch.float64),
                 accumulate=True)
    gen = torch.Generator().manual_seed(seed)
    x = torch.randint(-2, 3, (n, 6), generator=gen).double() if ties else torch.randn(n, 6, generator=gen).double()
    return torch_ref.TGraph(ref), a, x


def _dense_sage(layer, a, h):
    deg = a.sum(1, keepdim=True)
    lin_first = layer._in > layer._out
    if layer._aggre_type == "pool":
        p = torch.relu(h @ layer.fc_pool.weight.t() + layer.fc_pool.bias)
        masked = torch.where(a[:, :, None] > 0, p[None, :, :], torch.full((), -float("inf"), dtype=h.dtype))
        m = masked.amax(dim=1)
        m = torch.where(torch.isinf(m), torch.zeros_like(m), m)          # no in-edge -> 0
        return h @ layer.fc_self.weight.t() + m @ layer.fc_neigh.weight.t() + layer.bias
    s = h @ layer.fc_neigh.weight.t() if lin_first else h
    if layer._aggre_type == "mean":
        neigh = (a @ s) / deg.clamp(min=1)
    else:
        neigh = (a @ s + s) / (deg + 1)
    if not lin_first:
        neigh = neigh @ layer.fc_neigh.weight.t()
    return (neigh if layer._aggre_type == "gcn" else h @ layer.fc_self.weight.t() + neigh) + layer.bias


@pytest.mark.parametrize("aggr", ["pool", "mean", "gcn"])
@pytest.mark.parametrize("fout", [4, 9])                # 6 -> 4 takes the lin_before_mp branch
def test_sage_layers_equal_a_dense_adjacency_formulation(aggr, fout):
    for ties in (False, True):
        g, a, x = _dense_case(seed=fout, ties=ties)
        torch.manual_seed(fout)
        layer = torch_ref.RefSAGEConv(6, fout, aggr).double()
        with torch.no_grad():
            layer.bias.normal_()
        xr = x.clone().requires_grad_(True)
        out = layer(g, xr)
        xd = x.clone().requires_grad_(True)
        want = _dense_sage(layer, a, xd)
        assert torch.allclose(out, want, rtol=1e-12, atol=1e-12)
        # node 7 has no in-edge: the neighbour term is 0 (pool / mean) or the node itself (gcn)
        lone = x[7] @ (layer.fc_neigh if aggr == "gcn" else layer.fc_self).weight.t() + layer.bias
        assert torch.allclose(out[7].detach(), lone.detach(), rtol=1e-12, atol=1e-12)
        if ties and aggr == "pool":
            continue      # torch.amax splits a tie's gradient evenly; the first-maximum rule has its own tests
        gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).double()
        grads = torch.autograd.grad(out, [xr, *layer.parameters()], gout)
        wants = torch.autograd.grad(want, [xd, *layer.parameters()], gout)
        for got, exp in zip(grads, wants):
            assert torch.allclose(got, exp, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("heads,dim,residual,fin", [(3, 5, False, 6), (2, 3, True, 6), (1, 4, True, 4)])
def test_gat_layer_equals_a_dense_masked_softmax_formulation(heads, dim, residual, fin):
    n = 30
    src, dst = random_coo(n, 80, seed=heads, min_in_degree=1)
    src, dst = np.concatenate([src, [5, 5]]), np.concatenate([dst, [5, 6]])     # a self-loop
    g = torch_ref.TGraph(graph_ref.RefGraph(src, dst, n))
    a = torch.zeros(n, n, dtype=torch.float64)
    a.index_put_((torch.from_numpy(dst), torch.from_numpy(src)), torch.ones(len(src), dtype=torch.float64),
                 accumulate=True)
    torch.manual_seed(dim)
    layer = torch_ref.RefGATConv(fin, dim, heads, residual=residual, activation=torch.nn.functional.elu).double()
    with torch.no_grad():
        layer.bias.normal_()
    x = torch.randn(n, fin, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    out = layer(g, xr)

    xd = x.clone().requires_grad_(True)
    ft = (xd @ layer.fc.weight.t()).view(n, heads, dim)
    el = (ft * layer.attn_l).sum(-1)
    er = (ft * layer.attn_r).sum(-1)
    score = torch.nn.functional.leaky_relu(el[None, :, :] + er[:, None, :], 0.2)          # [v, u, h]
    score = torch.where(a[:, :, None] > 0, score, torch.full((), -float("inf"), dtype=torch.float64))
    w = a[:, :, None] * torch.exp(score - score.amax(dim=1, keepdim=True).detach())       # multiplicity counts
    w = w / w.sum(dim=1, keepdim=True)
    want = torch.einsum("vuh,uhd->vhd", w, ft)
    if residual:
        res = xd if fin == heads * dim else xd @ layer.res_fc.weight.t()
        want = want + res.view(n, heads, dim)
    want = torch.nn.functional.elu(want + layer.bias.view(1, heads, dim))
    assert torch.allclose(out, want, rtol=1e-12, atol=1e-12)
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(2)).double()
    grads = torch.autograd.grad(out, [xr, *layer.parameters()], gout)
    wants = torch.autograd.grad(want, [xd, *layer.parameters()], gout)
    for got, exp in zip(grads, wants):
        assert torch.allclose(got, exp, rtol=1e-9, atol=1e-12)
