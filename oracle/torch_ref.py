"""Oracle (test infrastructure): CPU restatement of the GNN layer stack in plain PyTorch.

Never imported by the product (``gnn-tumor-seg_amd/``).  Works in fp32 or fp64
(the dtype of the inputs / ``module.double()``).

What it follows
  model/networks.py:20-36   GraphSage  (stack of SAGEConv, ReLU on all but last)
  model/networks.py:39-66   GAT        (stack of GATConv, flatten(1) / mean(1))
  model/networks.py:68-81   init_graph_net
  model/gnn_model.py:28-30,34-48  optimizer / loss / one training step
and, for the arithmetic inside ``SAGEConv`` / ``GATConv`` (third-party DGL,
unpinned "DGL>=0.4" README.md:20, absent from /root/reference and from this
container), DGL's published algorithm (dgl.nn.pytorch.conv.SAGEConv / GATConv,
dgl.ops.edge_softmax, the max/mean/sum reducers of update_all), restated here.
PARITY UNPINNED for that part: the reference holds no fixture for it.

DGL rules encoded (each is also what the HIP path implements):
  R-max   out[v,f] = max_{u in N_in(v)} x[u,f]; the FIRST maximum in in-edge order
          (sources in COO/CSR order) owns the gradient (strict '<' update from -inf).
          +-inf results are replaced by 0 (dgl.ops.gspmm: replace_inf_with_zero),
          which also yields 0 for zero in-degree rows and blocks their gradient.
  R-mean  sum in in-edge order, then a true division by clamp(in_degree, 1).
  R-gcn   (sum + x[v]) / (in_degree + 1).
  R-lin   SAGEConv mean/gcn apply fc_neigh BEFORE aggregation when in_feats > out_feats.
  R-bias  ONE bias vector added after fc_self(h) + h_neigh.  Where DGL keeps it differs by version: a separate
          `bias` parameter (0.8 - 0.9), `fc_self.bias` (>= 1.0), `fc_self.bias` + `fc_neigh.bias` (<= 0.6: their
          sum); the function is the same, the product's loader folds whichever keys a checkpoint has.
  R-gat   e = leaky_relu(el[src] + er[dst]); a = softmax over the in-edges of each dst, per
          head (max-subtracted); out = sum_k a_k * ft[src_k]; + res_fc(h) ; + bias ; activation.
          Zero in-degree nodes raise (allow_zero_in_degree=False).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ---------------------------------------------------------------- graph view
class TGraph:
    """Torch (CPU, int64) view of an oracle.graph_ref.RefGraph."""

    def __init__(self, ref):
        self.n = ref.n
        self.indptr = torch.from_numpy(np.ascontiguousarray(ref.indptr)).long()
        self.indices = torch.from_numpy(np.ascontiguousarray(ref.indices)).long()
        self.t_indptr = torch.from_numpy(np.ascontiguousarray(ref.t_indptr)).long()
        self.t_indices = torch.from_numpy(np.ascontiguousarray(ref.t_indices)).long()
        self.in_eid = torch.from_numpy(np.ascontiguousarray(ref.in_eid)).long()
        self.out_eid = torch.from_numpy(np.ascontiguousarray(ref.out_eid)).long()
        self.deg = (self.indptr[1:] - self.indptr[:-1])
        self.t_deg = (self.t_indptr[1:] - self.t_indptr[:-1])
        # destination of every in-edge slot (CSR order)
        self.dst_of_slot = torch.repeat_interleave(torch.arange(self.n), self.deg)
        # position of out-edge k (out-CSR order) inside the in-CSR: inverse of in_eid at out_eid
        inv = torch.empty_like(self.in_eid)
        inv[self.in_eid] = torch.arange(self.in_eid.numel())
        self.out_to_in_slot = inv[self.out_eid]

    def to(self, *_a, **_k):
        return self

    def number_of_edges(self):
        return int(self.indices.numel())


def _slot_iter(indptr, deg):
    """Yield (rows, edge_positions) for slot k = 0,1,... : sequential in-edge order."""
    maxdeg = int(deg.max()) if deg.numel() else 0
    for k in range(maxdeg):
        rows = torch.nonzero(deg > k, as_tuple=False).squeeze(1)
        yield rows, indptr[rows] + k


def seq_spmm_sum(indptr, indices, deg, x):
    """acc[v] = sum_k x[indices[indptr[v]+k]] accumulated sequentially in slot order."""
    out = torch.zeros((deg.numel(),) + tuple(x.shape[1:]), dtype=x.dtype)
    for rows, e in _slot_iter(indptr, deg):
        out[rows] = out[rows] + x[indices[e]]
    return out


class _SpMMMax(torch.autograd.Function):
    """R-max.  Returns (out, argsrc) with argsrc = -1 where nothing was selected."""

    @staticmethod
    def forward(ctx, g, x):
        n, f = g.n, x.shape[1]
        best = torch.full((n, f), -math.inf, dtype=x.dtype)
        arg = torch.full((n, f), -1, dtype=torch.long)
        for rows, e in _slot_iter(g.indptr, g.deg):
            u = g.indices[e]
            val = x[u]
            cur = best[rows]
            upd = cur < val
            best[rows] = torch.where(upd, val, cur)
            arg[rows] = torch.where(upd, u[:, None].expand_as(val), arg[rows])
        dead = torch.isinf(best)
        out = torch.where(dead, torch.zeros_like(best), best)
        arg = torch.where(dead, torch.full_like(arg, -1), arg)
        ctx.save_for_backward(arg)
        ctx.n_src = x.shape[0]
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, gout, _garg):
        (arg,) = ctx.saved_tensors
        f = arg.shape[1]
        gx = torch.zeros((ctx.n_src, f), dtype=gout.dtype)
        valid = arg >= 0
        cols = torch.arange(f)[None, :].expand_as(arg)
        gx.index_put_((arg[valid], cols[valid]), gout[valid], accumulate=True)
        return None, gx


class _SpMMSum(torch.autograd.Function):
    """Plain in-edge sum (slot order); backward = out-edge sum (slot order of the out-CSR)."""

    @staticmethod
    def forward(ctx, g, x):
        ctx.g = g
        return seq_spmm_sum(g.indptr, g.indices, g.deg, x)

    @staticmethod
    def backward(ctx, gout):
        g = ctx.g
        return None, seq_spmm_sum(g.t_indptr, g.t_indices, g.t_deg, gout)


def spmm_max(g, x):
    return _SpMMMax.apply(g, x)[0]


def spmm_max_with_arg(g, x):
    return _SpMMMax.apply(g, x)


def spmm_sum(g, x):
    return _SpMMSum.apply(g, x)


def spmm_mean(g, x):
    """R-mean."""
    deg = g.deg.clamp(min=1).to(x.dtype)
    return spmm_sum(g, x) / deg.view(-1, *([1] * (x.dim() - 1)))


def spmm_gcn(g, x):
    """R-gcn."""
    deg = g.deg.to(x.dtype)
    return (spmm_sum(g, x) + x) / (deg.view(-1, *([1] * (x.dim() - 1))) + 1)


def edge_softmax(g, e):
    """e: [E,H] in in-CSR slot order -> a: [E,H]; softmax over the in-edges of each dst."""
    dst = g.dst_of_slot
    h = e.shape[1]
    emax = torch.full((g.n, h), -math.inf, dtype=e.dtype)
    emax = emax.scatter_reduce(0, dst[:, None].expand(-1, h), e.detach(), "amax", include_self=True)
    ex = torch.exp(e - emax[dst])
    den = torch.zeros((g.n, h), dtype=e.dtype).index_add(0, dst, ex)
    return ex / den[dst]


def gat_aggregate(g, ft, el, er, negative_slope):
    """R-gat core: ft [N,H,D], el/er [N,H] -> (out [N,H,D], a [E,H])."""
    src, dst = g.indices, g.dst_of_slot
    e = F.leaky_relu(el[src] + er[dst], negative_slope)
    a = edge_softmax(g, e)
    msg = ft[src] * a[:, :, None]
    out = torch.zeros_like(ft).index_add(0, dst, msg)
    return out, a


# ---------------------------------------------------------------- layers
class RefSAGEConv(nn.Module):
    def __init__(self, in_feats, out_feats, aggregator_type, feat_drop=0.0, activation=None):
        super().__init__()
        if aggregator_type not in ("mean", "gcn", "pool"):
            raise KeyError(f"Invalid aggregator_type {aggregator_type}")
        self._in, self._out, self._aggre_type = in_feats, out_feats, aggregator_type
        self.feat_drop = nn.Dropout(feat_drop)
        self.activation = activation
        if aggregator_type == "pool":
            self.fc_pool = nn.Linear(in_feats, in_feats)
        if aggregator_type != "gcn":
            self.fc_self = nn.Linear(in_feats, out_feats, bias=False)
        self.fc_neigh = nn.Linear(in_feats, out_feats, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_feats))
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        if self._aggre_type == "pool":
            nn.init.xavier_uniform_(self.fc_pool.weight, gain=gain)
        if self._aggre_type != "gcn":
            nn.init.xavier_uniform_(self.fc_self.weight, gain=gain)
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)

    def forward(self, g, feat):
        h = self.feat_drop(feat)
        lin_before_mp = self._in > self._out
        if self._aggre_type == "mean":
            src = self.fc_neigh(h) if lin_before_mp else h
            neigh = spmm_mean(g, src)
            if not lin_before_mp:
                neigh = self.fc_neigh(neigh)
        elif self._aggre_type == "gcn":
            src = self.fc_neigh(h) if lin_before_mp else h
            neigh = spmm_gcn(g, src)
            if not lin_before_mp:
                neigh = self.fc_neigh(neigh)
        else:
            neigh = self.fc_neigh(spmm_max(g, F.relu(self.fc_pool(h))))
        rst = neigh if self._aggre_type == "gcn" else self.fc_self(h) + neigh
        rst = rst + self.bias
        if self.activation is not None:
            rst = self.activation(rst)
        return rst


class RefGATConv(nn.Module):
    def __init__(self, in_feats, out_feats, num_heads, feat_drop=0.0, attn_drop=0.0,
                 negative_slope=0.2, residual=False, activation=None):
        super().__init__()
        self._in, self._out, self._heads = in_feats, out_feats, num_heads
        self.fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)
        self.attn_l = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.attn_r = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.feat_drop = nn.Dropout(feat_drop)
        self.attn_drop = nn.Dropout(attn_drop)
        self.negative_slope = negative_slope
        self.bias = nn.Parameter(torch.zeros(num_heads * out_feats))
        if residual:
            if in_feats != out_feats * num_heads:
                self.res_fc = nn.Linear(in_feats, num_heads * out_feats, bias=False)
            else:
                self.res_fc = nn.Identity()
        else:
            self.register_buffer("res_fc", None)
        self.activation = activation
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_normal_(self.fc.weight, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        nn.init.xavier_normal_(self.attn_r, gain=gain)
        nn.init.constant_(self.bias, 0)
        if isinstance(self.res_fc, nn.Linear):
            nn.init.xavier_normal_(self.res_fc.weight, gain=gain)

    def forward(self, g, feat):
        if bool((g.deg == 0).any()):
            raise RuntimeError("There are 0-in-degree nodes in the graph, output for those nodes "
                               "will be invalid.")
        n = feat.shape[0]
        h = self.feat_drop(feat)
        ft = self.fc(h).view(n, self._heads, self._out)
        el = (ft * self.attn_l).sum(-1)
        er = (ft * self.attn_r).sum(-1)
        rst, a = gat_aggregate(g, ft, el, er, self.negative_slope)
        if self.res_fc is not None:
            rst = rst + self.res_fc(h).view(n, -1, self._out)
        rst = rst + self.bias.view(1, self._heads, self._out)
        if self.activation is not None:
            rst = self.activation(rst)
        return rst


class RefGraphSage(nn.Module):
    """model/networks.py:20-36."""

    def __init__(self, in_feats, layer_sizes, n_classes, aggregator_type, dropout):
        super().__init__()
        dims = [in_feats] + list(layer_sizes)
        self.layers = nn.ModuleList(
            RefSAGEConv(dims[i], dims[i + 1], aggregator_type, feat_drop=dropout, activation=F.relu)
            for i in range(len(layer_sizes)))
        self.layers.append(RefSAGEConv(dims[-1], n_classes, aggregator_type, feat_drop=0, activation=None))

    def forward(self, graph, features):
        h = features
        for layer in self.layers:
            h = layer(graph, h)
        return h


class RefGAT(nn.Module):
    """model/networks.py:39-66."""

    def __init__(self, in_feats, layer_sizes, n_classes, heads, residuals,
                 activation=F.elu, feat_drop=0, attn_drop=0, negative_slope=0.2):
        super().__init__()
        self.layers = nn.ModuleList()
        self.layers.append(RefGATConv(in_feats, layer_sizes[0], heads[0], feat_drop, attn_drop,
                                      negative_slope, False, activation))
        for i in range(1, len(layer_sizes)):
            self.layers.append(RefGATConv(layer_sizes[i - 1] * heads[i - 1], layer_sizes[i], heads[i],
                                          feat_drop, attn_drop, negative_slope, residuals[i], activation))
        self.layers.append(RefGATConv(layer_sizes[-1] * heads[-1], n_classes, 1, feat_drop, attn_drop,
                                      negative_slope, False, None))

    def forward(self, g, inputs):
        h = inputs
        for layer in self.layers[:-1]:
            h = layer(g, h).flatten(1)
        return self.layers[-1](g, h).mean(1)


def ref_init_graph_net(model_type, hp):
    """model/networks.py:68-81."""
    dropout = hp.feature_dropout if "feature_dropout" in hp._fields else 0
    aggr = {"GSpool": "pool", "GSgcn": "gcn", "GSmean": "mean"}
    if model_type in aggr:
        return RefGraphSage(hp.in_feats, hp.layer_sizes, hp.out_classes, aggr[model_type], dropout)
    if model_type == "GAT":
        return RefGAT(hp.in_feats, hp.layer_sizes, hp.out_classes, hp.gat_heads, hp.gat_residuals)
    raise Exception(f"Unknown model type: {model_type}")


# ---------------------------------------------------------------- training step
def make_optimizer(net, lr=1e-4, w_decay=1e-4):
    """model/gnn_model.py:28."""
    return torch.optim.AdamW(net.parameters(), lr=lr, weight_decay=w_decay)


def train_step(net, g, feats, labels, class_weights, optimizer):
    """model/gnn_model.py:41-46 — forward, weighted CE, zero_grad, backward, step."""
    logits = net(g, feats)
    loss = F.cross_entropy(logits, labels, weight=class_weights)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return float(loss.detach())
