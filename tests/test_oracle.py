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
