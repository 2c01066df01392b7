"""Layer / network / harness parity with the oracle on a real MI355X.

Tolerances (fp32, stated per SURVEY.md §8c): a single layer rtol=1e-5/atol=1e-5; the full
8-layer network logits rtol=1e-4/atol=1e-4; gradients rtol=1e-3/atol=1e-5 relative to the
gradient scale.  The fp64 oracle is run beside the fp32 one to show which side is closer."""
from collections import namedtuple

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import gts
from gts import nn as gnn
from gts import synth
from gts.optim import FlatAdamW
from model.networks import init_graph_net
from oracle import graph_ref, torch_ref
from tests.helpers import copy_state, random_coo, ref_and_gts

pytestmark = pytest.mark.gpu
DEV = "cuda"
HP = namedtuple("HP", "in_feats out_classes layer_sizes gat_heads gat_residuals")


@pytest.fixture(scope="module", autouse=True)
def _lib(hip_lib):
    assert torch.cuda.is_available()
    return hip_lib


def _close(a, b, rtol, atol):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"max abs diff {(a - b).abs().max():.3e}"


@pytest.mark.parametrize("aggr", ["pool", "mean", "gcn"])
@pytest.mark.parametrize("fin,fout", [(4, 8), (8, 4), (20, 256), (256, 256), (256, 4)])
def test_sage_layer_forward_backward(aggr, fin, fout):
    n = 400
    src, dst = random_coo(n, 2400, seed=fin * 7 + fout)
    dst[dst == 0] = 1
    tg, g = ref_and_gts(src, dst, n)
    torch.manual_seed(fin + fout)
    ref = torch_ref.RefSAGEConv(fin, fout, aggr, activation=F.relu)
    with torch.no_grad():
        ref.bias.normal_()
    mine = gnn.SAGEConv(fin, fout, aggr, activation=F.relu)
    copy_state(mine, ref)
    mine.to(DEV)
    x = torch.randn(n, fin)
    gout = torch.randn(n, fout)
    xr = x.clone().requires_grad_(True)
    yr = ref(tg, xr)
    yr.backward(gout)
    xd = x.to(DEV).requires_grad_(True)
    yd = mine(g.to(DEV), xd)
    yd.backward(gout.to(DEV))
    scale = max(1.0, float(yr.detach().abs().max()))
    _close(yd, yr, 1e-5, 1e-5 * scale)
    gscale = max(1.0, float(xr.grad.abs().max()))
    _close(xd.grad, xr.grad, 1e-4, 1e-5 * gscale)
    for (name, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        s = max(1.0, float(q.grad.abs().max()))
        _close(p.grad, q.grad, 1e-4, 2e-5 * s)


@pytest.mark.parametrize("residual,fin,heads,dim", [(False, 4, 4, 8), (True, 32, 4, 8), (True, 12, 2, 16),
                                                     (False, 1024, 1, 4)])
def test_gat_layer_forward_backward(residual, fin, heads, dim):
    n = 300
    src, dst = random_coo(n, 1500, seed=fin, min_in_degree=1)
    tg, g = ref_and_gts(src, dst, n)
    torch.manual_seed(fin)
    ref = torch_ref.RefGATConv(fin, dim, heads, residual=residual, activation=F.elu)
    mine = gnn.GATConv(fin, dim, heads, 0, 0, 0.2, residual, F.elu)
    copy_state(mine, ref)
    mine.to(DEV)
    x = torch.randn(n, fin) * 0.5
    gout = torch.randn(n, heads, dim)
    xr = x.clone().requires_grad_(True)
    yr = ref(tg, xr)
    yr.backward(gout)
    xd = x.to(DEV).requires_grad_(True)
    yd = mine(g.to(DEV), xd)
    yd.backward(gout.to(DEV))
    _close(yd, yr, 1e-4, 1e-5 * max(1.0, float(yr.abs().max())))
    _close(xd.grad, xr.grad, 1e-3, 1e-5 * max(1.0, float(xr.grad.abs().max())))
    for (name, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        _close(p.grad, q.grad, 1e-3, 2e-5 * max(1.0, float(q.grad.abs().max())))


def test_gat_zero_in_degree_raises():
    src, dst = np.array([0, 1]), np.array([1, 0])
    g = gts.Graph(src, dst, 3).to(DEV)
    layer = gnn.GATConv(4, 4, 2).to(DEV)
    with pytest.raises(gnn.GraphError):
        layer(g, torch.zeros(3, 4, device=DEV))


def _net_pair(model_type, hp, seed):
    torch.manual_seed(seed)
    ref = torch_ref.ref_init_graph_net(model_type, hp)
    mine = init_graph_net(model_type, hp)
    assert list(mine.state_dict()) == list(ref.state_dict())          # checkpoint key layout
    copy_state(mine, ref)
    return ref, mine.to(DEV)


def test_c1_plumbing_config_matches_oracle():
    """BASELINE config 1: 2-layer SAGE-mean (4->8->4) on one random 1k-node / 6k-edge graph."""
    g = synth.random_graph(n=1000, n_pairs=2000, seed=1000)            # ring 1000 + 2000 = 3000 pairs
    assert g.number_of_edges() == 6000
    tg = torch_ref.TGraph(graph_ref.RefGraph(g.src, g.dst, g.n))
    ref, mine = _net_pair("GSmean", HP(4, 4, [8], None, None), seed=0)
    x = torch.from_numpy(synth.node_features(1000, 4, 1000))
    want = ref(tg, x)
    got = mine(g.to(DEV), x.to(DEV))
    _close(got, want, 1e-5, 1e-5)


@pytest.mark.parametrize("model_type,hp", [
    ("GSpool", HP(4, 4, [256] * 7, None, None)),
    ("GSpool", HP(20, 4, [256] * 4, None, None)),            # the shipped-weights architecture
    ("GSgcn", HP(20, 4, [64] * 3, None, None)),
    ("GSmean", HP(20, 4, [128] * 3, None, None)),
    ("GAT", HP(4, 4, [64] * 3, [4, 4, 4], [False, True, False])),
])
def test_full_network_logits_loss_and_grads(model_type, hp):
    n = 1500
    src, dst = random_coo(n, 8000, seed=len(hp.layer_sizes), min_in_degree=1)
    tg, g = ref_and_gts(src, dst, n)
    ref, mine = _net_pair(model_type, hp, seed=1)
    ref64 = torch_ref.ref_init_graph_net(model_type, hp).double()
    ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
    x = torch.from_numpy(synth.node_features(n, hp.in_feats, 5))
    y = torch.from_numpy(synth.node_labels(n, 5))
    w = torch.tensor([0.1, 1, 2, 2])
    lr = ref(tg, x)
    loss_r = F.cross_entropy(lr, y, weight=w)
    loss_r.backward()
    l64 = ref64(tg, x.double())
    F.cross_entropy(l64, y, weight=w.double()).backward()
    lm = mine(g.to(DEV), x.to(DEV))
    loss_m = F.cross_entropy(lm, y.to(DEV), weight=w.to(DEV))
    loss_m.backward()
    scale = max(1.0, float(lr.abs().max()))
    _close(lm, lr, 1e-4, 1e-4 * scale)
    err_gpu = (lm.detach().cpu().double() - l64).abs().max()
    err_cpu = (lr.detach().double() - l64).abs().max()
    print(f"{model_type}: |gpu-fp64|={err_gpu:.2e} |cpu32-fp64|={err_cpu:.2e}")
    assert err_gpu < 10 * max(float(err_cpu), 1e-6 * scale)
    assert abs(float(loss_m) - float(loss_r)) < 1e-4 * max(1.0, abs(float(loss_r)))
    # Gradients: ReLU / max-pool decisions are discrete, so through 8 layers an fp32 run may
    # take a different branch than another fp32 run on near-ties.  The yardstick is the fp64
    # oracle: the GPU must be as close to it as the CPU fp32 oracle is (x10 slack, floor 1e-3 of the scale), and within
    # 1e-2 of the gradient scale in absolute terms.
    for (name, p), (_, q), (_, q64) in zip(mine.named_parameters(), ref.named_parameters(),
                                            ref64.named_parameters()):
        s = max(float(q64.grad.abs().max()), 1e-6)
        e_gpu = float((p.grad.cpu().double() - q64.grad).abs().max())
        e_cpu = float((q.grad.double() - q64.grad).abs().max())
        assert e_gpu < max(10 * e_cpu, 1e-3 * s), f"{name}: gpu {e_gpu:.3e} cpu {e_cpu:.3e} scale {s:.3e}"
        assert e_gpu < 1e-2 * s, f"{name}: {e_gpu:.3e} vs scale {s:.3e}"


def test_batched_forward_equals_separate_forwards():
    parts = [gts.Graph(*random_coo(n, 5 * n, seed=n), n) for n in (50, 120, 7)]
    _, mine = _net_pair("GSpool", HP(4, 4, [32, 32], None, None), seed=3)
    feats = [torch.randn(p.n, 4) for p in parts]
    with torch.no_grad():
        whole = mine(gts.batch(parts).to(DEV), torch.cat(feats).to(DEV))
        sep = torch.cat([mine(p.to(DEV), f.to(DEV)) for p, f in zip(parts, feats)])
    assert torch.equal(whole, sep) or torch.allclose(whole, sep, rtol=1e-6, atol=1e-6)


def test_legacy_checkpoint_bias_folding_and_eval_mode():
    hp = HP(20, 4, [16], None, None)
    net = init_graph_net("GSpool", hp)
    sd = net.state_dict()
    legacy = {}
    for k, v in sd.items():
        if k.endswith(".bias") and "fc_pool" not in k:
            prefix = k[:-len("bias")]
            legacy[prefix + "fc_self.bias"] = torch.full_like(v, 0.25)
            legacy[prefix + "fc_neigh.bias"] = torch.full_like(v, 0.5)
        else:
            legacy[k] = v
    net2 = init_graph_net("GSpool", hp)
    net2.load_state_dict(legacy)
    assert torch.allclose(net2.layers[0].bias, torch.full((16,), 0.75))
    net2.to(DEV).eval()
    g = gts.Graph(*random_coo(30, 100, seed=1), 30).to(DEV)
    with torch.no_grad():
        out = net2(g, torch.randn(30, 20, device=DEV))
    assert out.shape == (30, 4) and torch.isfinite(out).all()


def test_checkpoint_with_only_fc_self_bias_loads_as_the_shared_bias():
    """DGL >= 1.0 keeps the layer bias in `fc_self.bias` (fc_neigh has none, there is no separate `bias`): such a
    checkpoint must load with that vector as the layer's bias — the same real-valued function (nn.py folds
    whichever of fc_self.bias / fc_neigh.bias are present)."""
    hp = HP(20, 4, [16], None, None)
    ref = init_graph_net("GSpool", hp)
    want = {k: torch.randn_like(v) for k, v in ref.state_dict().items()}
    ref.load_state_dict(want)
    newer = {(k[:-len("bias")] + "fc_self.bias" if k.endswith(".bias") and "fc_pool" not in k else k): v
             for k, v in want.items()}
    assert not any(k.endswith("layers.0.bias") or "fc_neigh.bias" in k for k in newer)
    net = init_graph_net("GSpool", hp)
    net.load_state_dict(newer)
    for a, b in zip(net.state_dict().items(), ref.state_dict().items()):
        assert a[0] == b[0] and torch.equal(a[1], b[1])
    g = gts.Graph(*random_coo(30, 100, seed=1), 30).to(DEV)
    x = torch.randn(30, 20, device=DEV)
    with torch.no_grad():
        assert torch.equal(net.to(DEV).eval()(g, x), ref.to(DEV).eval()(g, x))


class _MemDataset(torch.utils.data.Dataset):
    """In-memory stand-in for ImageGraphDataset (same item layout)."""

    def __init__(self, n_samples, n=400, in_feats=20):
        self.items = []
        for i in range(n_samples):
            g = synth.random_graph(n=n, n_pairs=2 * n, seed=1000 + i)
            self.items.append((f"s{i}", g, synth.node_features(n, in_feats, 1000 + i).astype(np.float64),
                               synth.node_labels(n, 1000 + i)))
        self.read_label = True

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def test_gnn_harness_trains_and_checkpoints(tmp_path):
    from model.gnn_model import GNN
    from utils.hyperparam_helpers import FullParamSet
    from utils.training_helpers import train_on_fold

    hp = FullParamSet(4, 20, 4, 1e-3, 0.98, 1e-4, [0.1, 1, 2, 2], [32, 32], 0, None, None)
    torch.manual_seed(0)
    model = GNN("GSpool", hp, _MemDataset(12))
    assert model.device.type == "cuda" and len(model.train_loader) == 2
    first = model.run_epoch()
    for _ in range(5):
        last = model.run_epoch()
    assert np.isfinite(first) and last < first
    train_on_fold(model, str(tmp_path) + "/", 2, "run", 1)
    sd = torch.load(tmp_path / "run_f1.pt", weights_only=True)
    assert list(sd) == list(model.net.state_dict())
    # one harness step == one oracle step from the same weights (world_size 1 path)
    ref = torch_ref.ref_init_graph_net("GSpool", hp)
    copy_state(ref, model.net.cpu())
    model.net.to(model.device)
    item = _MemDataset(1).items[0]
    tg = torch_ref.TGraph(graph_ref.RefGraph(item[1].src, item[1].dst, item[1].n))
    feats, labels = torch.FloatTensor(item[2]), torch.LongTensor(item[3])
    opt = torch_ref.make_optimizer(ref, lr=hp.lr, w_decay=hp.w_decay)
    loss_ref = torch_ref.train_step(ref, tg, feats, labels, torch.tensor(hp.class_weights), opt)
    model.optimizer = FlatAdamW(model.net.parameters(), lr=hp.lr, weight_decay=hp.w_decay)
    model.net.train()
    loss = model.train_step(item[1].to(model.device), feats.to(model.device), labels.to(model.device))
    assert abs(float(loss) - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
    for (k, a), (_, b) in zip(model.net.state_dict().items(), ref.state_dict().items()):
        assert torch.allclose(a.cpu(), b, rtol=1e-4, atol=1e-6), k


def test_flat_adamw_matches_torch_adamw_over_several_steps():
    """gts_adamw_f32 over the flat buffer against torch.optim.AdamW (the reference's optimizer,
    model/gnn_model.py:28) on a CPU twin: 25 steps with a decaying learning rate, odd tensor sizes;
    parameters stay views of the flat buffer and load_state_dict keeps working."""
    torch.manual_seed(3)
    shapes = [(7, 5), (5,), (33, 9), (1,), (256, 4), (4,)]
    cpu = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    gpu = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in cpu]
    ref = torch.optim.AdamW(cpu, lr=3e-3, weight_decay=1e-2)
    mine = FlatAdamW(gpu, lr=3e-3, weight_decay=1e-2)
    sched_ref = torch.optim.lr_scheduler.ExponentialLR(ref, 0.9)
    sched = torch.optim.lr_scheduler.ExponentialLR(mine, 0.9)
    for step in range(25):
        grads = [torch.randn(s) * (1.0 + step) for s in shapes]
        for p, q, g in zip(cpu, gpu, grads):
            p.grad, q.grad = g.clone(), g.to(DEV)
        ref.step()
        mine.step()
        if step % 5 == 4:
            sched_ref.step()
            sched.step()
        mine.zero_grad()
        assert all(q.grad is None for q in gpu)
    assert mine.param_groups[0]["lr"] == ref.param_groups[0]["lr"]
    off = 0
    for p, q in zip(cpu, gpu):
        assert torch.allclose(q.detach().cpu(), p.detach(), rtol=2e-5, atol=1e-6)
        assert q.data_ptr() == mine.flat_param.data_ptr() + 4 * off
        off += q.numel()
    flat = torch.cat([torch.randn(s).reshape(-1) for s in shapes]).to(DEV)
    before = mine.flat_param.clone()
    mine.step(flat_grad=flat)                       # the data-parallel entry: gradients already flat
    assert not torch.equal(before, mine.flat_param)
    gpu[0].grad = None
    with pytest.raises(gts.GtsError):
        mine.step()
    with pytest.raises(gts.GtsError):
        mine.step(flat_grad=flat[:-1])
    gpu[2].data = gpu[2].data.clone()               # re-allocated parameter: refuse instead of updating a stale buffer
    with pytest.raises(gts.GtsError):
        mine.step(flat_grad=flat)


def test_rccl_world_size_one_flat_gradient_path():
    """The data-parallel step on one rank over the real RCCL backend ('nccl' on ROCm): same
    gradients and loss as the plain single-GPU step."""
    import os
    import socket

    import torch.distributed as dist

    from gts import dist as gdist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        hp = HP(4, 4, [64, 64], None, None)
        _, mine = _net_pair("GSpool", hp, seed=7)
        _, twin = _net_pair("GSpool", hp, seed=7)
        src, dst = random_coo(500, 3000, seed=1)
        g = gts.Graph(src, dst, 500).to(DEV)
        x = torch.randn(500, 4, device=DEV)
        y = torch.randint(0, 4, (500,), device=DEV)
        w = torch.tensor([0.1, 1, 2, 2], device=DEV)
        sync = gdist.FlatGradSync(mine.parameters())
        sync.zero_grad()
        sync.weighted_ce_backward(mine(g, x), y, w)
        loss = sync.all_reduce_and_normalise()
        ref_loss = F.cross_entropy(twin(g, x), y, weight=w)
        ref_loss.backward()
        assert abs(float(loss) - float(ref_loss)) < 1e-5 * abs(float(ref_loss))
        for p, q in zip(mine.parameters(), twin.parameters()):
            assert torch.allclose(p.grad, q.grad, rtol=1e-4, atol=1e-6)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("hp", [HP(4, 4, [256] * 3, None, None), HP(20, 4, [64, 32], None, None)])
def test_fused_pool_stack_equals_layer_by_layer_path(hp):
    """The fused stack moves the ReLU backward into a GEMM epilogue (logits and input gradients are
    bitwise identical) and batches the weight gradients of all layers into one launch."""
    src, dst = random_coo(700, 4000, seed=3)
    g = gts.Graph(src, dst, 700).to(DEV)
    _, fused = _net_pair("GSpool", hp, seed=11)
    _, plain = _net_pair("GSpool", hp, seed=11)
    plain.fuse_layers = False
    x = torch.randn(700, hp.in_feats, device=DEV)
    y = torch.randint(0, 4, (700,), device=DEV)
    outs = []
    for net in (fused, plain):
        xi = x.clone().requires_grad_(True)
        logits = net(g, xi)
        F.cross_entropy(logits, y).backward()
        outs.append((logits.detach(), xi.grad, [p.grad for p in net.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    # the stack sums each weight gradient's node reduction in fewer, longer splits (all layers in
    # one launch) than the per-layer path: same terms, different association
    for a, b in zip(outs[0][2], outs[1][2]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * max(1.0, float(b.abs().max())))
    with torch.no_grad():
        assert torch.equal(fused(g, x), outs[0][0])          # inference path (no argmax written)
    plain.layers[0].feat_drop.p = 0.5                          # active dropout -> stack declines
    from gts.nn import sage_pool_stack
    plain.train()
    assert sage_pool_stack(g, x, list(plain.layers)) is None


@pytest.mark.parametrize("residual,fin,heads,dim,act", [(False, 4, 4, 8, F.elu), (True, 32, 4, 8, F.elu),
                                                         (True, 12, 2, 16, F.elu), (False, 64, 1, 4, None),
                                                         (True, 1024, 4, 256, F.elu)])
def test_fused_gat_layer_equals_op_by_op_path(residual, fin, heads, dim, act, monkeypatch):
    """Same kernels for attention/aggregation; the fused node only moves scores, bias, residual,
    ELU and the node-axis reductions into them.  fp32 rounding differs slightly (different
    summation trees): rtol 1e-5 forward, 1e-4 gradients."""
    n = 400
    src, dst = random_coo(n, 2500, seed=fin, min_in_degree=1)
    g = gts.Graph(src, dst, n).to(DEV)
    torch.manual_seed(fin)
    layer = gnn.GATConv(fin, dim, heads, 0, 0, 0.2, residual, act).to(DEV)
    with torch.no_grad():
        layer.bias.normal_()
    x = torch.randn(n, fin, device=DEV) * 0.5
    gout = torch.randn(n, heads, dim, device=DEV)
    res = []
    for fused in (True, False):
        monkeypatch.setattr(gnn, "FUSE_GAT_LAYER", fused)
        layer.zero_grad()
        xi = x.clone().requires_grad_(True)
        y = layer(g, xi)
        y.backward(gout)
        res.append((y.detach(), xi.grad.clone(), {k: p.grad.clone() for k, p in layer.named_parameters()}))
    assert torch.allclose(res[0][0], res[1][0], rtol=1e-5, atol=1e-5)
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-4, atol=1e-5 * max(1.0, float(res[1][1].abs().max())))
    for k in res[0][2]:
        a, b = res[0][2][k], res[1][2][k]
        assert torch.allclose(a, b, rtol=1e-4, atol=2e-5 * max(1.0, float(b.abs().max()))), k


@pytest.mark.parametrize("aggr", ["pool", "mean", "gcn"])
def test_sage_layer_with_widths_that_are_not_multiples_of_four(aggr):
    """5 -> 7 features: the GEMM kernels need 16-byte rows, so the host pads (dense._pad4_*)."""
    n = 120
    src, dst = random_coo(n, 700, seed=5)
    tg, g = ref_and_gts(src, dst, n)
    torch.manual_seed(5)
    ref = torch_ref.RefSAGEConv(5, 7, aggr, activation=F.relu)
    mine = gnn.SAGEConv(5, 7, aggr, activation=F.relu)
    copy_state(mine, ref)
    mine.to(DEV)
    x = torch.randn(n, 5)
    gout = torch.randn(n, 7)
    xr = x.clone().requires_grad_(True)
    ref(tg, xr).backward(gout)
    xd = x.to(DEV).requires_grad_(True)
    yd = mine(g.to(DEV), xd)
    yd.backward(gout.to(DEV))
    _close(yd, ref(tg, x), 1e-5, 1e-5)
    _close(xd.grad, xr.grad, 1e-4, 1e-5)
    for (name, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        _close(p.grad, q.grad, 1e-4, 2e-5 * max(1.0, float(q.grad.abs().max())))


def test_gat_layer_with_odd_head_width_uses_the_op_by_op_path():
    n = 90
    src, dst = random_coo(n, 500, seed=6, min_in_degree=1)
    tg, g = ref_and_gts(src, dst, n)
    torch.manual_seed(6)
    ref = torch_ref.RefGATConv(10, 6, 3, residual=True, activation=F.elu)     # D = 6: not a multiple of 4
    mine = gnn.GATConv(10, 6, 3, 0, 0, 0.2, True, F.elu)
    copy_state(mine, ref)
    mine.to(DEV)
    x = torch.randn(n, 10) * 0.5
    gout = torch.randn(n, 3, 6)
    xr = x.clone().requires_grad_(True)
    ref(tg, xr).backward(gout)
    xd = x.to(DEV).requires_grad_(True)
    yd = mine(g.to(DEV), xd)
    yd.backward(gout.to(DEV))
    _close(yd, ref(tg, x), 1e-4, 1e-5)
    _close(xd.grad, xr.grad, 1e-3, 1e-5)


def test_shape_mismatches_raise_on_the_host_before_any_launch():
    """The kernels index by the sizes they are handed, so every wrapper checks operand shapes
    first (a mismatch must be a Python error, never a wrong result or a GPU fault)."""
    from gts import dense, ops

    src, dst = random_coo(50, 300, seed=2)
    g = gts.Graph(src, dst, 50).to(DEV)
    x = torch.randn(50, 8, device=DEV)
    with pytest.raises(gts.GtsError):
        ops.spmm_max_fwd(g, x[:49])
    out, arg = ops.spmm_max_fwd(g, x)
    with pytest.raises(gts.GtsError):
        ops.spmm_max_bwd(g, out, arg[:, :4])
    with pytest.raises(gts.GtsError):
        ops.spmm_max_bwd(g, out, arg, relu_src=x[:, :4])
    with pytest.raises(gts.GtsError):
        ops.spmm_sum_raw(g, x, div_out=torch.ones(49, device=DEV))
    ft, el = torch.randn(50, 2, 4, device=DEV), torch.randn(50, 2, device=DEV)
    with pytest.raises(gts.GtsError):
        ops._gat_fwd(g, ft, el[:, :1], el, 0.2)
    with pytest.raises(gts.GtsError):
        ops._gat_fwd(g, ft, el, el, 0.2, bias=torch.zeros(7, device=DEV))
    with pytest.raises(gts.GtsError):
        ops._gat_fwd(g, ft, el, el, 0.2, residual=torch.zeros(50, 7, device=DEV))
    with pytest.raises(gts.GtsError):
        ops.gat_scores(ft, torch.zeros(3, 4, device=DEV), torch.zeros(2, 4, device=DEV))
    a, w = torch.randn(50, 8, device=DEV), torch.randn(16, 12, device=DEV)
    with pytest.raises(gts.GtsError):
        dense.linear_fwd(a, w)
    with pytest.raises(gts.GtsError):
        dense.linear_fwd(a, w[:, :8], bias=torch.zeros(15, device=DEV))
    with pytest.raises(gts.GtsError):
        dense.linear_fwd(a, w[:, :8], a[:49], w[:, :8])
    with pytest.raises(gts.GtsError):
        dense.linear_bwd_input(a, w)                        # g [50,8] @ w [16,12]
    with pytest.raises(gts.GtsError):
        dense.linear_bwd_input(a, w[:8], relu_mask=torch.ones(50, 11, device=DEV))
    with pytest.raises(gts.GtsError):
        dense.linear_bwd_weight_multi([(a, a, False), (a[:49], a[:49], False)])
    with pytest.raises(gts.GtsError):
        ops.weighted_cross_entropy(a, torch.zeros(50, dtype=torch.int64, device=DEV), torch.ones(7, device=DEV))


def test_cross_entropy_labels_outside_the_classes():
    """ignore_index (-100) rows contribute nothing, as in torch; any other out-of-range label
    gives a NaN loss (torch: device assert) and never an out-of-bounds read."""
    from gts import ops

    logits = torch.randn(40, 4, device=DEV, requires_grad=True)
    labels = torch.randint(0, 4, (40,), device=DEV)
    labels[::7] = -100
    w = torch.tensor([0.1, 1.0, 2.0, 2.0], device=DEV)
    got = ops.weighted_cross_entropy(logits, labels, w)
    want = F.cross_entropy(logits.detach().cpu(), labels.cpu(), weight=w.cpu())
    assert abs(float(got) - float(want)) < 1e-5
    got.backward()
    assert not logits.grad[::7].any()
    labels[3] = 4
    assert torch.isnan(ops.weighted_cross_entropy(logits.detach(), labels, w))
    labels[3] = 2 ** 40
    assert torch.isnan(ops.weighted_cross_entropy(logits.detach(), labels, w))


def test_reference_hardcoded_gat_defaults_are_inconsistent_and_fail_loudly():
    """utils/hyperparam_helpers.py:39-41 pairs 4 layer sizes with heads [4,4,3,3,4,4]: the output
    layer is built for 256*4 inputs but receives 256*3 (model/networks.py:52-56), so the reference
    fails in the last layer's matmul.  Same model here, same failure — as a shape error."""
    from model.networks import init_graph_net
    from utils.hyperparam_helpers import populate_hardcoded_hyperparameters

    net = init_graph_net("GAT", populate_hardcoded_hyperparameters("GAT")).to(DEV)
    assert [layer.fc.weight.shape[1] for layer in net.layers] == [20, 1024, 1024, 768, 1024]
    src, dst = random_coo(60, 400, seed=3)
    g = gts.Graph(np.concatenate([src, np.arange(60)]), np.concatenate([dst, np.arange(60)]), 60).to(DEV)
    with pytest.raises(gts.GtsError, match="shapes do not match"):
        net(g, torch.randn(60, 20, device=DEV))


def test_prefetched_epochs_equal_plain_epochs_and_errors_surface():
    """run_epoch with the batch-prefetch thread (collate + upload one step ahead on a copy
    stream) gives bit-identical epoch losses and weights to iterating the loader in line; an
    exception inside the dataset reaches the caller."""
    from model.gnn_model import GNN
    from utils.hyperparam_helpers import FullParamSet

    hp = FullParamSet(3, 20, 4, 1e-3, 0.98, 1e-4, [0.1, 1, 2, 2], [64, 64], 0, None, None)
    results = []
    # host_collate: batches assembled by gts_collate_batch (one C call, one upload) vs the Python collate + per-array uploads
    for prefetch, host_collate in ((True, True), (False, False), (False, True), (True, False)):
        torch.manual_seed(5)
        model = GNN("GSpool", hp, _MemDataset(14), batch_size=3, prefetch=prefetch, host_collate=host_collate)
        losses = [model.run_epoch() for _ in range(3)]
        results.append((losses, [p.detach().clone() for p in model.net.parameters()]))
    for other in results[1:]:
        assert results[0][0] == other[0]
        assert all(torch.equal(a, b) for a, b in zip(results[0][1], other[1]))

    class Broken(_MemDataset):
        def __getitem__(self, i):
            if i == 5:
                raise KeyError("sample 5 is unreadable")
            return super().__getitem__(i)

    torch.manual_seed(5)
    model = GNN("GSpool", hp, Broken(9), batch_size=3)
    with pytest.raises(KeyError, match="unreadable"):
        model.run_epoch()
    # abandoning an epoch half way (consumer stops early) must not leave the producer stuck
    model = GNN("GSpool", hp, _MemDataset(12), batch_size=2)
    batches = model._device_batches()
    next(batches)
    batches.close()
    assert np.isfinite(model.run_epoch())


def test_host_collated_batch_on_the_device_equals_the_python_collate():
    """gts.collate.HostCollator: one C call + one upload; every device array the kernels read (features, labels, both CSRs,
    degree vectors, the 'in' / 'out' schedule records) equals what minibatch_graphs + Graph.dev() + dev_schedule() upload,
    and a 256-wide pool layer run over either graph gives the same bits (the clustered K2 reads the collated records)."""
    from data_processing.data_loader import minibatch_graphs
    from gts import collate
    from gts.graph import PinnedRing

    dims = [(12, 12, 12), (13, 12, 11), (12, 11, 10)]
    samples = [(f"s{i}", synth.lattice_graph(d), synth.node_features(int(np.prod(d)), 20, i).astype(np.float64),
                synth.node_labels(int(np.prod(d)), i)) for i, d in enumerate(dims)]
    ring = PinnedRing(slabs=2, nbytes=1 << 16)          # too small on purpose: reserve() replaces the slab
    dev = torch.device("cuda", torch.cuda.current_device())
    col = collate.HostCollator(dev, ring, lambda n_rows: ("out", "in"))
    for _ in range(3):                                   # slabs are reused: the bytes must be right every time round
        ids, g, feats, labels = col(samples)
        ring.next_batch()
        rids, rg, rfeats, rlabels = minibatch_graphs(samples)
        rg = rg.to(DEV)
        assert ids == rids and torch.equal(feats.cpu(), rfeats) and torch.equal(labels.cpu(), rlabels)
        d, rd = g.dev(), rg.dev()
        for name in ("indptr", "indices", "t_indptr", "t_indices", "t_slot", "t_pos", "deg_clamped", "deg_plus1"):
            assert torch.equal(getattr(d, name), getattr(rd, name)), name
        for which in ("in", "out"):
            a, b = g.dev_schedule(which), rg.dev_schedule(which)
            assert a is not None and b is not None
            assert torch.equal(a.packed, b.packed) and a.host.n_clusters == b.host.n_clusters
            assert (a.host.limits, a.host.loc_words, a.host.tagged) == (b.host.limits, b.host.loc_words, b.host.tagged)
    torch.manual_seed(0)
    layer = gnn.SAGEConv(256, 256, "pool", activation=F.relu).to(DEV)
    x = torch.randn(g.n, 256, device=DEV)
    outs = []
    for graph in (g, rg):
        xi = x.clone().requires_grad_(True)
        y = layer(graph, xi)
        y.square().sum().backward()
        outs.append((y.detach(), xi.grad))
        layer.zero_grad()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_pinned_ring_grows_a_slab_a_batch_has_outgrown_and_keeps_the_bytes():
    """gts.graph.PinnedRing: uploads that do not fit the current page-locked slab go up as plain copies (same bytes on the
    device) and make next_batch() grow the slabs, after which a batch of that size is staged entirely through them."""
    from gts.graph import PinnedRing

    ring = PinnedRing(slabs=2, nbytes=1 << 16)
    rng = np.random.default_rng(3)
    parts = [torch.from_numpy(rng.integers(0, 1 << 30, size=n, dtype=np.int32)) for n in (9000, 6000, 5000)]   # 80 000 bytes > 64 KiB
    for _ in range(5):
        up = [ring.upload(t, torch.device(DEV)) for t in parts]
        assert all(torch.equal(d.cpu(), h) for d, h in zip(up, parts))
        ring.next_batch()
    assert all(slab.numel() >= 80000 for slab in ring.slabs)
    assert ring.at == 0
    up = [ring.upload(t, torch.device(DEV)) for t in parts]
    assert ring.at >= 80000 and ring.at <= ring.slabs[ring.cur].numel()      # the whole batch sits in the slab now
    assert all(torch.equal(d.cpu(), h) for d, h in zip(up, parts))


def test_row_count_invariant_batched_forward_is_bitwise_the_per_sample_forward():
    """What GNN.evaluate relies on: under dense.row_count_invariant() a block-diagonal batch tall enough
    to select the 240-row panels otherwise (8 x 7 000 nodes) gives every sample the logits of its own
    one-sample forward, bit for bit; without the pin the batch rounds differently (16x16x4 MFMA)."""
    from gts import dense

    parts = [synth.random_graph(n=7000, n_pairs=14000, seed=50 + i) for i in range(8)]
    feats = [torch.from_numpy(synth.node_features(7000, 20, 50 + i)) for i in range(8)]
    _, mine = _net_pair("GSpool", HP(20, 4, [256] * 4, None, None), seed=4)
    mine.eval()
    with torch.no_grad():
        sep = torch.cat([mine(p.to(DEV), f.to(DEV)) for p, f in zip(parts, feats)])
        g, x = gts.batch(parts).to(DEV), torch.cat(feats).to(DEV)
        with dense.row_count_invariant():
            pinned = mine(g, x)
        free = mine(g, x)
    assert torch.equal(pinned, sep)
    assert not torch.equal(free, sep) and torch.allclose(free, sep, rtol=1e-4, atol=1e-4)
    with torch.no_grad():                          # the pin is released on exit
        assert torch.equal(mine(g, x), free)


@pytest.mark.parametrize("aggr", ["mean", "gcn"])
@pytest.mark.parametrize("fin,fout", [(6, 3), (3, 6), (12, 12)])
def test_sage_sum_layers_with_widths_that_are_not_multiples_of_four(aggr, fin, fout):
    """Odd widths take the op-by-op route (fout % 4 != 0) or the fused node with host-side padding (fin only)."""
    n = 200
    src, dst = random_coo(n, 900, seed=fin + fout)
    tg, g = ref_and_gts(src, dst, n)
    torch.manual_seed(1)
    ref = torch_ref.RefSAGEConv(fin, fout, aggr, activation=F.relu)
    mine = gnn.SAGEConv(fin, fout, aggr, activation=F.relu)
    copy_state(mine, ref)
    mine.to(DEV)
    x, gout = torch.randn(n, fin), torch.randn(n, fout)
    xr = x.clone().requires_grad_(True)
    ref(tg, xr).backward(gout)
    xd = x.to(DEV).requires_grad_(True)
    yd = mine(g.to(DEV), xd)
    yd.backward(gout.to(DEV))
    _close(yd, ref(tg, x), 1e-5, 1e-5)
    _close(xd.grad, xr.grad, 1e-4, 1e-5)
    for (_, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        _close(p.grad, q.grad, 1e-4, 2e-5 * max(1.0, float(q.grad.abs().max())))


@pytest.mark.parametrize("n,hp", [(6000, HP(4, 4, [64] * 3, [4, 4, 4], [False, True, False])),
                                  (60000, HP(4, 4, [64] * 2, [4, 4], [False, False]))])
def test_gat_act_backward_in_the_next_layers_gemm_equals_the_pass_of_its_own(n, hp, monkeypatch):
    """ActLink (GAT.forward): ELU' of a hidden layer and its bias gradient inside the next layer's input-gradient GEMM
    (60 000 rows: the panel kernel's epilogue; 6 000 rows: GEMM + in-place pass) vs the layer-by-layer backward: every
    gradient but the hidden biases bit for bit, those as a different fixed-order sum of the same numbers."""
    src, dst = random_coo(n, 5 * n, seed=3, min_in_degree=1)
    g = gts.Graph(src, dst, n).to(DEV)
    _, mine = _net_pair("GAT", hp, seed=2)
    x = torch.from_numpy(synth.node_features(n, hp.in_feats, 5)).to(DEV)
    y = torch.from_numpy(synth.node_labels(n, 5)).to(DEV)
    grads = {}
    for fold in (True, False):
        monkeypatch.setattr(gnn, "FOLD_GAT_ACT_BWD", fold)
        mine.zero_grad(set_to_none=True)
        F.cross_entropy(mine(g, x), y).backward()
        grads[fold] = {k: p.grad.clone() for k, p in mine.named_parameters()}
    for name, want in grads[False].items():
        got = grads[True][name]
        if name.endswith("bias") and not name.startswith(f"layers.{len(hp.layer_sizes)}."):
            scale = float(want.abs().max()) + 1e-30
            assert float((got - want).abs().max()) <= 1e-4 * scale + 1e-9, name
        else:
            assert torch.equal(got, want), name


def test_gat_act_link_stands_down_when_a_dropout_sits_between_the_layers():
    """With feat_drop > 0 in training mode a layer's input is no longer the producer's ELU output, so the consumer must not
    apply ELU' through it: the linked and the unlinked backward must agree (same dropout masks via the same seed)."""
    from model.networks import GAT
    n = 6000
    src, dst = random_coo(n, 5 * n, seed=4, min_in_degree=1)
    g = gts.Graph(src, dst, n).to(DEV)
    torch.manual_seed(0)
    net = GAT(4, [64, 64], 4, [4, 4], [False, False], feat_drop=0.3).to(DEV).train()
    x = torch.from_numpy(synth.node_features(n, 4, 5)).to(DEV)
    y = torch.from_numpy(synth.node_labels(n, 5)).to(DEV)
    grads = []
    for fold in (True, False):
        gnn.FOLD_GAT_ACT_BWD = fold
        try:
            net.zero_grad(set_to_none=True)
            torch.manual_seed(7)
            F.cross_entropy(net(g, x), y).backward()
        finally:
            gnn.FOLD_GAT_ACT_BWD = True
        grads.append([p.grad.clone() for p in net.parameters()])
    for a, b in zip(*grads):
        assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("case", ["unfused_producer", "relu_activation", "no_activation"])
def test_gat_act_link_is_armed_by_the_producer_only(case):
    """ActLink is two-sided: the consumer folds ELU' into its input-gradient GEMM only when the producer ran as the
    fused node WITH ELU (and will therefore skip its own activation backward).  Producers the reference's constructor
    allows but the fold does not cover — a 50-wide hidden layer (op-by-op path: 50 % 4 != 0, and 4 heads give the fused
    output layer a 200-wide input at n >= 4096), `activation=F.relu`, `activation=None` (model/networks.py:40-58
    exposes both arguments) — must give the same gradients with the fold switched on and off, and match autograd of
    the op-by-op network."""
    from model.networks import GAT
    n = 6000
    src, dst = random_coo(n, 5 * n, seed=6, min_in_degree=1)
    g = gts.Graph(src, dst, n).to(DEV)
    torch.manual_seed(1)
    kwargs = {"unfused_producer": dict(layer_sizes=[50], heads=[4], residuals=[False]),
              "relu_activation": dict(layer_sizes=[64], heads=[4], residuals=[False], activation=F.relu),
              "no_activation": dict(layer_sizes=[64], heads=[4], residuals=[False], activation=None)}[case]
    sizes, heads, res = kwargs.pop("layer_sizes"), kwargs.pop("heads"), kwargs.pop("residuals")
    net = GAT(4, sizes, 4, heads, res, **kwargs).to(DEV).train()
    x = torch.from_numpy(synth.node_features(n, 4, 5)).to(DEV)
    y = torch.from_numpy(synth.node_labels(n, 5)).to(DEV)
    grads = {}
    for mode in ("fold", "nofold", "unfused"):
        saved = gnn.FOLD_GAT_ACT_BWD, gnn.FUSE_GAT_LAYER
        gnn.FOLD_GAT_ACT_BWD, gnn.FUSE_GAT_LAYER = mode == "fold", mode != "unfused"
        try:
            net.zero_grad(set_to_none=True)
            F.cross_entropy(net(g, x), y).backward()
        finally:
            gnn.FOLD_GAT_ACT_BWD, gnn.FUSE_GAT_LAYER = saved
        grads[mode] = {k: p.grad.clone() for k, p in net.named_parameters()}
    for name, want in grads["nofold"].items():
        assert torch.isfinite(grads["fold"][name]).all()
        assert torch.equal(grads["fold"][name], want), name                  # the link stood down: same launches, same bits
        ref = grads["unfused"][name]
        scale = float(ref.abs().max()) + 1e-30
        assert float((want - ref).abs().max()) <= 2e-4 * scale + 1e-8, name    # and they are the right gradients
