"""The fused SAGEConv-pool stack behind one C-ABI call each way (gts_sage_pool_stack_fwd_f32 / _bwd_f32) against the
same stack issued launch by launch from Python (gts/nn.py::_SagePoolStack): the library enqueues the same kernels in
the same order, so logits, loss and every gradient must be equal BIT FOR BIT — and a training run that lets the stack
write its gradients straight into the optimizer's flat buffer must produce the same parameters as one that hands
autograd per-parameter tensors.  Reference call sites: /root/reference/model/networks.py:25-36 (the layer loop),
/root/reference/model/gnn_model.py:41-46 (forward, loss, backward, step)."""
import io
from collections import namedtuple
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

import gts
from gts import nn as gnn
from gts import ops, schedule, synth
from model.networks import init_graph_net

pytestmark = pytest.mark.gpu
DEV = "cuda"
HP = namedtuple("HP", "in_feats out_classes layer_sizes gat_heads gat_residuals")
W = torch.tensor([0.1, 1.0, 2.0, 2.0])


@pytest.fixture(scope="module", autouse=True)
def _lib(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    old = schedule.MAX_MEAN_DEGREE_FORWARD, schedule.MAX_DEGREE_BACKWARD
    schedule.MAX_MEAN_DEGREE_FORWARD, schedule.MAX_DEGREE_BACKWARD = 1e9, 10 ** 9     # every graph with a schedule runs clustered
    yield hip_lib
    schedule.MAX_MEAN_DEGREE_FORWARD, schedule.MAX_DEGREE_BACKWARD = old


def _run(net, g, x, y, one_call, train=True):
    old = gnn.STACK_IN_ONE_CALL
    gnn.STACK_IN_ONE_CALL = one_call
    try:
        net.zero_grad(set_to_none=True)
        x = x.clone().requires_grad_(True)
        if not train:
            with torch.no_grad():
                return net(g, x), None, None, None
        logits = net(g, x)
        loss = ops.weighted_cross_entropy(logits, y, W.to(DEV))
        loss.backward()
        return logits.detach(), loss.detach(), x.grad, [p.grad.clone() for p in net.parameters()]
    finally:
        gnn.STACK_IN_ONE_CALL = old


CASES = {
    # name: (in_feats, layer_sizes, graph maker, MIN_ROWS_FORWARD)
    "reference_shape": (20, [256] * 4, lambda: gts.batch([synth.lattice_graph((9, 9, 9)) for _ in range(3)]), 0),
    "c2_shape_small": (4, [256] * 7, lambda: gts.batch([synth.lattice_graph((12, 11, 10)), synth.geometric_graph(2500, 7, 3)]), 0),
    "plain_kernels": (4, [256] * 3, lambda: synth.random_graph(n=3000, n_pairs=6000, seed=2), 0),      # no schedule at all
    "forward_plain_backward_clustered": (20, [256, 256], lambda: synth.lattice_graph((10, 10, 10)), 10 ** 9),
    "narrow_layers": (8, [64, 128, 32], lambda: synth.geometric_graph(1800, 8, 4), 0),                 # nothing 256 wide
    "one_layer": (20, [], lambda: synth.lattice_graph((8, 8, 8)), 0),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_one_call_stack_equals_launch_by_launch_bit_for_bit(name):
    in_feats, sizes, make, min_rows = CASES[name]
    old_min = schedule.MIN_ROWS_FORWARD
    schedule.MIN_ROWS_FORWARD = min_rows
    try:
        g = make().to(DEV)
        torch.manual_seed(len(name))
        net = init_graph_net("GSpool", HP(in_feats, 4, sizes, None, None)).to(DEV)
        x = torch.randn(g.n, in_feats, device=DEV)
        y = torch.randint(0, 4, (g.n,), device=DEV)
        a = _run(net, g, x, y, one_call=False)
        b = _run(net, g, x, y, one_call=True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        for (k, _), ga, gb in zip(net.named_parameters(), a[3], b[3]):
            assert torch.equal(ga, gb), k
        ea, eb = _run(net, g, x, y, False, train=False)[0], _run(net, g, x, y, True, train=False)[0]
        assert torch.equal(ea, eb)
        net.eval()
        with torch.no_grad():
            assert torch.equal(net(g, x), ea) or torch.allclose(net(g, x), ea)
    finally:
        schedule.MIN_ROWS_FORWARD = old_min


def test_gradients_written_into_the_optimizer_flat_buffer_train_the_same_network():
    """GNN.train_step with the grad sink (the stack writes into one flat buffer, FlatAdamW consumes it) against the
    same steps with per-parameter gradient tensors concatenated by the optimizer."""
    from model.gnn_model import GNN
    from utils.hyperparam_helpers import FullParamSet

    hp = FullParamSet(1, 20, 4, 1e-3, 0.98, 1e-4, [0.1, 1, 2, 2], [256, 256], 0, None, None)
    g = gts.batch([synth.lattice_graph((9, 8, 7)), synth.lattice_graph((6, 6, 6))]).to(DEV)
    x = torch.randn(g.n, 20, device=DEV)
    y = torch.randint(0, 4, (g.n,), device=DEV)
    states, losses = [], []
    for use_sink in (True, False):
        torch.manual_seed(5)
        with redirect_stdout(io.StringIO()):
            model = GNN("GSpool", hp, None)
        model.net.train()
        if not use_sink:
            model.grad_sink = gnn.GradSink([torch.nn.Parameter(torch.zeros(1, device=DEV))])   # covers nothing: never filled
        out = [float(model.train_step(g, x, y)) for _ in range(4)]
        assert model.grad_sink.filled == use_sink
        losses.append(out)
        states.append({k: v.clone() for k, v in model.net.state_dict().items()})
    assert losses[0] == losses[1]
    for k in states[0]:
        assert torch.equal(states[0][k], states[1][k]), k


def test_a_sink_wider_than_the_stack_is_not_used_and_no_gradient_is_lost():
    """A network with parameters OUTSIDE the fused stack (here a learnable logit scale) under an active GradSink that covers
    all of them: the stack must hand autograd ordinary per-parameter gradients (the sink stays unfilled), so that the
    optimizer / FlatGradSync fallback of `p.grad` sees every gradient — none may silently become zero."""
    g = synth.lattice_graph((9, 8, 7)).to(DEV)
    torch.manual_seed(3)
    net = init_graph_net("GSpool", HP(20, 4, [256], None, None)).to(DEV)
    scale = torch.nn.Parameter(torch.full((4,), 1.5, device=DEV))
    x = torch.randn(g.n, 20, device=DEV)
    y = torch.randint(0, 4, (g.n,), device=DEV)

    def grads(sink):
        net.zero_grad(set_to_none=True)
        scale.grad = None
        loss = ops.weighted_cross_entropy(net(g, x) * scale, y, W.to(DEV))
        if sink is None:
            loss.backward()
        else:
            sink.new_buffer()
            with gnn.grad_sink(sink):
                loss.backward()
        return [p.grad.clone() if p.grad is not None else None for p in (*net.parameters(), scale)]

    want = grads(None)
    sink = gnn.GradSink([*net.parameters(), scale])
    got = grads(sink)
    assert not sink.filled
    assert all(a is not None and torch.equal(a, b) for a, b in zip(got, want))
    exact = gnn.GradSink(list(net.parameters()))          # a sink that IS the stack's parameters is filled as before
    got = grads(exact)
    assert exact.filled and all(a is None for a in got[:-1]) and torch.equal(got[-1], want[-1])
    flat = torch.cat([w.reshape(-1) for w in want[:-1]])
    assert torch.equal(exact.flat[:flat.numel()], flat)


def test_stack_entry_points_reject_bad_arguments():
    import ctypes

    lib = gts._lib.load()
    widths = (ctypes.c_int64 * 3)(20, 256, 4)
    offsets = (ctypes.c_int64 * 10)()
    assert lib.gts_sage_pool_stack_fwd_arena(1000, widths, 2, 1, 1, 7, offsets) > 0
    assert offsets[1] > 0 and offsets[5] > 0 and offsets[7] == -1          # winners kept; no bits behind the last layer
    assert lib.gts_sage_pool_stack_fwd_arena(1000, widths, 2, 0, 1, 7, offsets) > 0 and offsets[1] == -1
    bad = (ctypes.c_int64 * 3)(20, 255, 4)
    assert lib.gts_sage_pool_stack_fwd_arena(1000, bad, 2, 1, 1, 7, None) == -1      # widths must be multiples of 4
    assert lib.gts_sage_pool_stack_fwd_arena(1000, widths, 0, 1, 1, 7, None) == -1
    assert lib.gts_sage_pool_stack_bwd_scratch(1000, widths, 2, 7) > 0
    x = torch.zeros(1000, 20, device=DEV)
    table = (ctypes.c_void_p * 10)(*([x.data_ptr()] * 10))
    arena = torch.empty(64, dtype=torch.uint8, device=DEV)
    ip = torch.zeros(1001, dtype=torch.int32, device=DEV)
    code = lib.gts_sage_pool_stack_fwd_f32(ip.data_ptr(), ip.data_ptr(), None, 0, 0, 0, 0, x.data_ptr(), table, 1000, widths, 2,
                                           1, 1, 7, arena.data_ptr(), 64, None)
    assert code == -2                                                      # arena too small
    code = lib.gts_sage_pool_stack_fwd_f32(ip.data_ptr(), ip.data_ptr(), None, 0, 0, 0, 0, None, table, 1000, widths, 2,
                                           1, 1, 7, arena.data_ptr(), 64, None)
    assert code == -1
