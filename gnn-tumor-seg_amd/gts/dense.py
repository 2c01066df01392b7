"""Dense layer GEMMs of the GNN path (K11): out = act(a0 @ w0^T [+ a1 @ w1^T] + bias).

Single seam for the dense work inside SAGEConv / GATConv so that the fused layer ops in
gts.nn never touch a GEMM API directly.  fp32 throughout (the reference trains in fp32).
"""
import torch


def linear_fwd(a0, w0, a1=None, w1=None, bias=None, relu=False):
    """a0 [M,K0], w0 [N,K0] (torch Linear layout), optional second operand pair."""
    if bias is not None:
        out = torch.addmm(bias, a0, w0.t())
    else:
        out = torch.mm(a0, w0.t())
    if a1 is not None:
        out.addmm_(a1, w1.t())
    if relu:
        out.relu_()
    return out


def linear_bwd_input(g, w, out=None):
    """g [M,N], w [N,K] -> g @ w [M,K]; accumulates into `out` when given."""
    if out is None:
        return torch.mm(g, w)
    return out.addmm_(g, w)


def linear_bwd_weight(g, a):
    """g [M,N], a [M,K] -> g^T @ a [N,K]."""
    return torch.mm(g.t(), a)


def bias_grad(g):
    return g.sum(dim=0)


def relu_bwd(g, out):
    """g * (out > 0) where `out` is the ReLU output (new tensor; g is left untouched)."""
    return g * (out > 0)
