"""Python wrappers + autograd glue over the C-ABI HIP kernels (include/gts_hip.h).

Every function here enqueues on torch's current HIP stream and returns torch tensors that
own the memory; nothing falls back to PyTorch or the CPU when the library is missing.
"""
import torch

from . import _lib
from ._lib import check, current_stream, ptr, require_device

_ARG_DTYPE = {0: torch.uint8, 1: torch.uint8, 4: torch.int32}

# bench.py installs callables here (kernel name -> wrapper) to bracket each launch of a kernel
# with HIP events; empty in normal use.  Names: "spmm_max_fwd_f256", "spmm_max_bwd_f256", "gat_fwd", "gat_bwd_edge",
# "gat_bwd_src" (the GAT kernels are bracketed for the HIDDEN-layer shape only: H * D >= 256; the 1-head classifier
# launches move a hundredth of the bytes and would dilute the means), 
# "project_rows".
KERNEL_TIMERS = {}
# True while bench.py runs its instrumented block: the fused layer stack then issues its launches one ctypes call at a
# time (the route the timers bracket) instead of through the one-call entry points.
INSTRUMENTED = False


def _timed(name, launch):
    timer = KERNEL_TIMERS.get(name)
    return timer(launch) if timer is not None else launch()


def _expect(tensor, shape, name):
    """Operand shapes are checked on the host: the kernels index by the sizes they are given."""
    if tensor is not None and tuple(tensor.shape) != tuple(shape):
        raise _lib.GtsError(f"{name}: expected shape {tuple(shape)}, got {tuple(tensor.shape)}")


def _f32(*tensors):
    for t in tensors:
        if t is not None and t.dtype != torch.float32:
            raise _lib.GtsError(f"gts kernels are fp32: got {t.dtype}")


_cluster_counters = {}


def cluster_counters(device):
    """The unit counters of the clustered K1 / K2 kernels (include/gts_hip.h: GTS_CLUSTER_COUNTER_WORDS zero words that every launch
    leaves zero): one buffer per (device, stream), zeroed once."""
    key = (device, current_stream())
    buf = _cluster_counters.get(key)
    if buf is None:
        buf = torch.zeros(256, dtype=torch.int32, device=device)
        _cluster_counters[key] = buf
    return buf


# ---------------------------------------------------------------- raw kernel calls
def spmm_max_fwd(g, x, want_arg=True, relu_input=False):
    """K1.  x [N,F] -> (out [N,F], arg [N,F] slot ids or None).  relu_input: x is a ReLU output;
    maxima that are not positive get no winner, so spmm_max_bwd on this arg yields the gradient
    w.r.t. the pre-activation directly (no relu_src needed)."""
    x = x.contiguous()
    _f32(x)
    require_device(x)
    d = g.dev()
    if x.dim() != 2 or x.shape[0] != g.n:
        raise _lib.GtsError(f"features must be [graph nodes = {g.n}, F], got {tuple(x.shape)}")
    n, f = g.n, x.shape[1]
    out = torch.empty((n, f), dtype=torch.float32, device=x.device)
    ab = g.arg_bytes if want_arg else 0
    arg = torch.empty((n, f), dtype=_ARG_DTYPE[ab], device=x.device) if want_arg else None
    lib = _lib.load()
    ds = _cluster_schedule(g, "in", n, f, ab)
    if ds is not None:      # LDS-staged neighbour tiles over the graph's cluster row schedule: same result, bit for bit
        h = ds.host

        def launch():
            return lib.gts_spmm_max_fwd_cluster_f32(ptr(ds.packed), h.n_clusters, h.limits[0], h.limits[1], h.loc_words,
                                                    ptr(x), ptr(out), ptr(arg), ab, 1 if relu_input else 0, n, f,
                                                    ptr(cluster_counters(x.device)), current_stream())
    else:
        def launch():
            return lib.gts_spmm_max_fwd_f32(ptr(d.indptr), ptr(d.indices), ptr(x), ptr(out), ptr(arg),
                                            ab, 1 if relu_input else 0, n, f, current_stream())

    code = _timed("spmm_max_fwd_f256", launch) if (f == 256 and want_arg) else launch()
    check(code, "gts_spmm_max_fwd_f32")
    return out, arg


def _cluster_schedule(g, which, n, f, arg_bytes):
    """Device schedule for the clustered K1 / K2 kernels, or None when they do not apply (F != 256, 4-byte
    winners, tables of 4 GiB and more, no worthwhile schedule, GTS_CLUSTER_SPMM=0)."""
    from . import schedule

    if not schedule.ENABLED or f != 256 or arg_bytes not in (0, 1) or n * 1024 >= 2 ** 32 or n == 0:
        return None
    if which == "in" and n < schedule.MIN_ROWS_FORWARD:
        return None
    return g.dev_schedule(which)


def spmm_max_bwd(g, gout, arg, relu_src=None):
    """K2.  gout [N,F], arg from spmm_max_fwd -> gx [N,F]; optional fused ReLU mask."""
    gout = gout.contiguous()
    _f32(gout, relu_src)
    require_device(gout, arg, relu_src)
    d = g.dev()
    if gout.dim() != 2 or gout.shape[0] != g.n:
        raise _lib.GtsError(f"gradient must be [graph nodes = {g.n}, F], got {tuple(gout.shape)}")
    n, f = g.n, gout.shape[1]
    _expect(arg, (n, f), "arg")
    _expect(relu_src, (n, f), "relu_src")
    if arg.dtype != _ARG_DTYPE[g.arg_bytes]:
        raise _lib.GtsError(f"arg dtype {arg.dtype} does not belong to this graph ({_ARG_DTYPE[g.arg_bytes]})")
    gx = torch.empty((n, f), dtype=torch.float32, device=gout.device)
    lib = _lib.load()
    ds = _cluster_schedule(g, "out", n, f, arg.element_size()) if relu_src is None else None
    if ds is not None:
        h = ds.host

        def launch():
            return lib.gts_spmm_max_bwd_cluster_f32(ptr(ds.packed), h.n_clusters, h.limits[0], h.limits[1], h.loc_words,
                                                    ptr(gout), ptr(arg), 1, ptr(gx), n, f, ptr(cluster_counters(gout.device)),
                                                    current_stream())
    else:
        def launch():
            return lib.gts_spmm_max_bwd_f32(ptr(d.t_indptr), ptr(d.t_indices), ptr(d.t_slot), ptr(gout),
                                            ptr(arg), arg.element_size(), ptr(relu_src), ptr(gx), n, f,
                                            current_stream())

    check(_timed("spmm_max_bwd_f256", launch) if f == 256 else launch(), "gts_spmm_max_bwd_f32")
    return gx


def spmm_sum_raw(g, x, transposed=False, div_in=None, div_out=None, add_self=False, accum=None):
    """K3/K4.  Generic sum reducer over the in-CSR (or the out-CSR when transposed); `accum` [N,F] is added
    to the result (a gradient that reached the rows by another path)."""
    x = x.contiguous()
    _f32(x, div_in, div_out, accum)
    require_device(x, div_in, div_out, accum)
    d = g.dev()
    indptr, indices = (d.t_indptr, d.t_indices) if transposed else (d.indptr, d.indices)
    n = g.n
    if x.dim() != 2 or x.shape[0] != n:
        raise _lib.GtsError(f"features must be [graph nodes = {n}, F], got {tuple(x.shape)}")
    f = x.shape[1]
    _expect(div_in, (n,), "div_in")
    _expect(div_out, (n,), "div_out")
    _expect(accum, (n, f), "accum")
    out = torch.empty_like(x)
    check(_lib.load().gts_spmm_sum_f32(ptr(indptr), ptr(indices), ptr(x), ptr(out), ptr(div_in), ptr(div_out),
                                       ptr(accum), 1 if add_self else 0, n, f, current_stream()), "gts_spmm_sum_f32")
    return out


# ---------------------------------------------------------------- autograd: reducers
class _SpMMMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, x):
        out, arg = spmm_max_fwd(g, x, want_arg=True)
        ctx.g = g
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, gout):
        (arg,) = ctx.saved_tensors
        return None, spmm_max_bwd(ctx.g, gout, arg)


class _SpMMReduce(torch.autograd.Function):
    """mode: 'sum' | 'mean' | 'gcn' (DGL reducers reached from SAGEConv)."""

    @staticmethod
    def forward(ctx, g, x, mode):
        ctx.g, ctx.mode = g, mode
        d = g.dev()
        if mode == "sum":
            return spmm_sum_raw(g, x)
        if mode == "mean":
            return spmm_sum_raw(g, x, div_out=d.deg_clamped)
        if mode == "gcn":
            return spmm_sum_raw(g, x, div_out=d.deg_plus1, add_self=True)
        raise ValueError(mode)

    @staticmethod
    def backward(ctx, gout):
        g, mode = ctx.g, ctx.mode
        d = g.dev()
        if mode == "sum":
            gx = spmm_sum_raw(g, gout, transposed=True)
        elif mode == "mean":
            gx = spmm_sum_raw(g, gout, transposed=True, div_in=d.deg_clamped)
        else:
            gx = spmm_sum_raw(g, gout, transposed=True, div_in=d.deg_plus1, add_self=True)
        return None, gx, None


def spmm_max(g, x):
    return _SpMMMax.apply(g, x)


def spmm_reduce(g, x, mode):
    return _SpMMReduce.apply(g, x, mode)


# ---------------------------------------------------------------- autograd: GAT
def _gat_fwd(g, ft, el, er, slope, bias=None, residual=None, activation=0):
    d = g.dev()
    if ft.dim() != 3 or ft.shape[0] != g.n:
        raise _lib.GtsError(f"ft must be [graph nodes = {g.n}, H, D], got {tuple(ft.shape)}")
    n, h, dim = ft.shape
    _expect(el, (n, h), "el")
    _expect(er, (n, h), "er")
    _expect(bias, (h * dim,), "bias")
    if residual is not None and (residual.shape[0] != n or residual.numel() != n * h * dim):
        raise _lib.GtsError(f"residual must hold [{n}, {h}*{dim}] values, got {tuple(residual.shape)}")
    out = torch.empty_like(ft)
    attn = torch.empty((g.number_of_edges(), h), dtype=torch.float32, device=ft.device)
    lib = _lib.load()
    ds = _gat_cluster_schedule(g, "gat_in", n, h, dim) if residual is None else None
    if ds is not None:      # LDS-staged neighbour slices over the graph's cluster row schedule: same result, bit for bit
        hs = ds.host
        nbytes = lib.gts_gat_cluster_workspace(hs.n_clusters, hs.limits[0], hs.loc_words, h, 0)
        ws = _gat_ws(ft.device, nbytes)
        launch = lambda: lib.gts_gat_fwd_cluster_f32(      # noqa: E731
            ptr(d.indptr), ptr(d.indices), ptr(ds.packed), hs.n_clusters, hs.limits[0], hs.limits[1], hs.loc_words,
            1 if hs.tagged else 0, ptr(ft), ptr(el), ptr(er), float(slope), ptr(bias), activation, ptr(out), ptr(attn),
            ptr(ws), nbytes, n, h, dim, g.max_in_degree, current_stream())
    else:
        launch = lambda: lib.gts_gat_fwd_f32(      # noqa: E731
            ptr(d.indptr), ptr(d.indices), ptr(ft), ptr(el), ptr(er), float(slope), ptr(bias), ptr(residual),
            activation, ptr(out), ptr(attn), n, h, dim, current_stream())
    check(_timed("gat_fwd", launch) if h * dim >= 256 else launch(), "gts_gat_fwd_f32")
    return out, attn


_gat_workspaces = {}


def _gat_ws(device, nbytes):
    """Scratch of the clustered GAT kernels (their weight blocks), grown on demand and reused: one per (device, stream)."""
    key = (device, current_stream())
    buf = _gat_workspaces.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _gat_workspaces[key] = buf
    return buf


def _gat_cluster_schedule(g, which, n, heads, dim):
    """Device schedule for the clustered GATConv kernels, or None when they do not apply (D != 256, tables of 4 GiB and
    more, small graphs, rows of more than one 8-edge chunk, no worthwhile schedule, GTS_CLUSTER_GAT=0)."""
    from . import schedule

    if not schedule.ENABLED_GAT or dim != 256 or n < schedule.MIN_ROWS_GAT or n * heads * 1024 >= 2 ** 32 or heads > 64:
        return None
    if which.endswith("in") and g.max_in_degree > 64:
        return None
    return g.dev_schedule(which)


def _gat_bwd(g, ft, el, er, attn, gout, slope, attn_l=None, attn_r=None):
    """(gft, gel, ger); with attn_l/attn_r the score-dot-product gradient is folded into gft."""
    d = g.dev()
    if ft.dim() != 3 or ft.shape[0] != g.n:
        raise _lib.GtsError(f"ft must be [graph nodes = {g.n}, H, D], got {tuple(ft.shape)}")
    n, h, dim = ft.shape
    _expect(el, (n, h), "el")
    _expect(er, (n, h), "er")
    _expect(attn, (g.number_of_edges(), h), "attn")
    _expect(gout, (n, h, dim), "gout")
    _expect(attn_l, (h, dim), "attn_l")
    _expect(attn_r, (h, dim), "attn_r")
    lib = _lib.load()
    ge = torch.empty_like(attn)
    ger = torch.empty_like(er)
    wide = h * dim >= 256
    es = _gat_cluster_schedule(g, "gat_edge_in", n, h, dim) if g.max_in_degree <= 8 else None
    if es is not None:
        hs = es.host
        nbytes = 2 * lib.gts_gat_cluster_workspace(hs.n_clusters, hs.limits[0], hs.loc_words, h, 0)
        ws_e = _gat_ws(ft.device, nbytes)
        edge = lambda: lib.gts_gat_bwd_edge_cluster_f32(      # noqa: E731
            ptr(d.indptr), ptr(d.indices), ptr(es.packed), hs.n_clusters, hs.limits[0], hs.limits[1], hs.loc_words,
            1 if hs.tagged else 0, ptr(ft), ptr(el), ptr(er), ptr(attn), ptr(gout), float(slope), ptr(ge), ptr(ger),
            ptr(ws_e), nbytes, n, h, dim, g.max_in_degree, current_stream())
    else:
        edge = lambda: lib.gts_gat_bwd_edge_f32(ptr(d.indptr), ptr(d.indices), ptr(ft), ptr(el), ptr(er), ptr(attn),   # noqa: E731
                                                ptr(gout), float(slope), ptr(ge), ptr(ger), n, h, dim, current_stream())
    check(_timed("gat_bwd_edge", edge) if wide else edge(), "gts_gat_bwd_edge_f32")
    gft = torch.empty_like(ft)
    gel = torch.empty_like(el)
    ds = _gat_cluster_schedule(g, "gat_out", n, h, dim)
    if ds is not None:
        hs = ds.host
        nbytes = lib.gts_gat_cluster_workspace(hs.n_clusters, hs.limits[0], hs.loc_words, h, 1)
        ws = _gat_ws(ft.device, nbytes)
        attn_lr = torch.stack([attn_l.reshape(-1), attn_r.reshape(-1)]) if attn_l is not None else None
        src = lambda: lib.gts_gat_bwd_src_cluster_f32(      # noqa: E731
            ptr(d.t_indptr), ptr(d.t_pos), ptr(ds.packed), hs.n_clusters, hs.limits[0], hs.limits[1], hs.loc_words,
            1 if hs.tagged else 0, ptr(attn), ptr(ge), ptr(gout), ptr(attn_lr), ptr(ger) if attn_l is not None else None,
            ptr(gft), ptr(gel), ptr(ws), nbytes, n, h, dim, g.max_out_degree, current_stream())
    else:
        src = lambda: lib.gts_gat_bwd_src_f32(ptr(d.t_indptr), ptr(d.t_indices), ptr(d.t_pos), ptr(attn), ptr(ge),     # noqa: E731
                                              ptr(gout), ptr(attn_l), ptr(attn_r),
                                              ptr(ger) if attn_l is not None else None, ptr(gft), ptr(gel), n, h, dim,
                                              current_stream())
    check(_timed("gat_bwd_src", src) if wide else src(), "gts_gat_bwd_src_f32")
    return gft, gel, ger


class _GATAggregate(torch.autograd.Function):
    """K5-K8.  (ft [N,H,D], el [N,H], er [N,H]) -> out [N,H,D]."""

    @staticmethod
    def forward(ctx, g, ft, el, er, negative_slope):
        ft, el, er = ft.contiguous(), el.contiguous(), er.contiguous()
        _f32(ft, el, er)
        require_device(ft, el, er)
        out, attn = _gat_fwd(g, ft, el, er, negative_slope)
        ctx.g, ctx.slope = g, float(negative_slope)
        ctx.save_for_backward(ft, el, er, attn)
        return out

    @staticmethod
    def backward(ctx, gout):
        ft, el, er, attn = ctx.saved_tensors
        gft, gel, ger = _gat_bwd(ctx.g, ft, el, er, attn, gout.contiguous(), ctx.slope)
        return None, gft, gel, ger, None


def gat_aggregate(g, ft, el, er, negative_slope):
    return _GATAggregate.apply(g, ft, el, er, negative_slope)


def gat_scores(ft, attn_l, attn_r):
    """(el, er) [N,H]: <ft[n,h,:], attn_l[h,:]>, <ft[n,h,:], attn_r[h,:]> in one pass over ft."""
    ft, attn_l, attn_r = ft.contiguous(), attn_l.contiguous(), attn_r.contiguous()
    _f32(ft, attn_l, attn_r)
    require_device(ft, attn_l, attn_r)
    if ft.dim() != 3:
        raise _lib.GtsError(f"ft must be [N, H, D], got {tuple(ft.shape)}")
    n, h, dim = ft.shape
    attn_l, attn_r = attn_l.reshape(-1, dim), attn_r.reshape(-1, dim)
    _expect(attn_l, (h, dim), "attn_l")
    _expect(attn_r, (h, dim), "attn_r")
    el = torch.empty((n, h), dtype=torch.float32, device=ft.device)
    er = torch.empty_like(el)
    check(_lib.load().gts_gat_scores_f32(ptr(ft), ptr(attn_l), ptr(attn_r), ptr(el), ptr(er), n, h, dim,
                                         current_stream()), "gts_gat_scores_f32")
    return el, er


def gat_fc_scores(h, w_fc, attn_l, attn_r, heads, dim, packed=None):
    """(ft [N,H,D], el [N,H], er [N,H]) = (h @ w_fc^T viewed [N,H,D], <ft, attn_l>, <ft, attn_r>): GATConv's projection
    with the attention scores computed in the GEMM's epilogue when the operands are tall (else GEMM + gat_scores).
    packed: dense.pack_weights copy of w_fc (fragment order; read by the panel kernels, same values)."""
    h, w_fc = h.contiguous(), w_fc.contiguous()
    attn_l, attn_r = attn_l.reshape(heads, dim).contiguous(), attn_r.reshape(heads, dim).contiguous()
    _f32(h, w_fc, attn_l, attn_r)
    require_device(h, w_fc, attn_l, attn_r)
    n, k = h.shape
    if w_fc.dim() != 2 or w_fc.shape[1] != k:
        raise _lib.GtsError(f"shapes do not match: inner dims of h @ fc.weight^T ({k} vs {tuple(w_fc.shape)[-1]})")
    if w_fc.shape[0] != heads * dim:
        raise _lib.GtsError(f"shapes do not match: fc.weight rows ({w_fc.shape[0]}) vs heads * out_feats ({heads * dim})")
    if k % 4:
        raise _lib.GtsError("gat_fc_scores needs an input width that is a multiple of 4")
    lib = _lib.load()
    ft = torch.empty((n, heads, dim), dtype=torch.float32, device=h.device)
    el = torch.empty((n, heads), dtype=torch.float32, device=h.device)
    er = torch.empty_like(el)
    nbytes = lib.gts_gat_fc_scores_workspace(n, heads, dim)
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=h.device)
    from . import dense
    if packed is not None:
        dense._packed_array((packed,), (w_fc,))      # size check only
    dense._timed("fwd", 2.0 * n * heads * dim * k, lambda: check(lib.gts_gat_fc_scores_f32(
        ptr(h), ptr(w_fc), ptr(attn_l), ptr(attn_r), ptr(ft), ptr(el), ptr(er), ptr(ws), ws.numel() * 4, n, heads, dim, k,
        ptr(packed), current_stream()), "gts_gat_fc_scores_f32"))
    return ft, el, er


def _reduce_ws(n, cols, device):
    nbytes = _lib.load().gts_gat_reduce_workspace(n, cols)
    return torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=device)


def gat_act_bwd(gout, out, activation, want_bias_grad):
    """(g_pre, g_bias): g_pre = gout * act'(out) (activation 1 = ELU, 2 = ReLU, through the output; 0 = none),
    g_bias = g_pre.sum(0)."""
    gout = gout.contiguous()
    n, cols = gout.shape[0], gout[0].numel()
    if activation:
        _expect(out, gout.shape, "out")
    g_pre = torch.empty_like(gout) if activation else gout
    g_bias = torch.empty(cols, dtype=torch.float32, device=gout.device) if want_bias_grad else None
    ws = _reduce_ws(n, cols, gout.device) if want_bias_grad else None
    check(_lib.load().gts_gat_act_bwd_f32(ptr(gout), ptr(out), activation, ptr(g_pre) if activation else None,
                                          ptr(g_bias), ptr(ws), ws.numel() * 4 if ws is not None else 0,
                                          n, cols, current_stream()), "gts_gat_act_bwd_f32")
    return g_pre, g_bias


def gat_param_grad(ft, gel, ger):
    """(g_attn_l, g_attn_r) [H,D] = sum_n gel[n,h] ft[n,h,:], sum_n ger[n,h] ft[n,h,:]."""
    n, h, dim = ft.shape
    _expect(gel, (n, h), "gel")
    _expect(ger, (n, h), "ger")
    gl = torch.empty((h, dim), dtype=torch.float32, device=ft.device)
    gr = torch.empty_like(gl)
    ws = _reduce_ws(n, h * dim, ft.device)
    check(_lib.load().gts_gat_param_grad_f32(ptr(ft), ptr(gel), ptr(ger), ptr(gl), ptr(gr), ptr(ws),
                                             ws.numel() * 4, n, h, dim, current_stream()),
          "gts_gat_param_grad_f32")
    return gl, gr


# ---------------------------------------------------------------- node -> voxel projection
def project_rows(svs, table, bg_row):
    """K12.  svs int16 [...], table [N, ...] with 4/8/16-byte rows, bg_row one such row.
    Returns table_plus_bg[svs] with shape svs.shape + table.shape[1:]."""
    if svs.dtype != torch.int16:
        raise _lib.GtsError("supervoxel partitioning must be int16 (mri2graph/graphgen.py:77)")
    svs = svs.contiguous()          # NIfTI volumes arrive in Fortran order: C-order copy, same shape
    table = table.contiguous()
    bg_row = bg_row.to(table.dtype).contiguous()
    require_device(svs, table, bg_row)
    row_bytes = table.element_size() * (table[0].numel() if table.shape[0] else bg_row.numel())
    if bg_row.numel() * bg_row.element_size() != row_bytes:
        raise _lib.GtsError("background row does not match table rows")
    out = torch.empty(tuple(svs.shape) + tuple(table.shape[1:]), dtype=table.dtype, device=svs.device)
    if svs.numel() == 0:
        return out
    lib = _lib.load()
    check(_timed("project_rows", lambda: lib.gts_project_rows_i16(
        ptr(svs), ptr(table), ptr(bg_row), ptr(out), svs.numel(), table.shape[0], row_bytes,
        current_stream())), "gts_project_rows_i16")
    return out


def project_argmax(svs, logits, relabel=None):
    """K12 fused: argmax over classes of the voxel's node, 0 for background; int16 out."""
    _f32(logits)
    if svs.dtype != torch.int16:
        raise _lib.GtsError("supervoxel partitioning must be int16")
    svs, logits = svs.contiguous(), logits.contiguous()
    require_device(svs, logits, relabel)
    if relabel is not None and (relabel.dtype != torch.int16 or relabel.numel() < logits.shape[1]):
        raise _lib.GtsError("relabel must be int16 with one entry per class")
    out = torch.empty(svs.shape, dtype=torch.int16, device=svs.device)
    if svs.numel() == 0:
        return out
    check(_lib.load().gts_project_argmax_i16(ptr(svs), ptr(logits), ptr(relabel), ptr(out), svs.numel(),
                                             logits.shape[0], logits.shape[1], current_stream()),
          "gts_project_argmax_i16")
    return out


def project_argmax_occupancy(svs, logits):
    """K12 arg-max projection of a 3-D partitioning plus the tumour-plane flags of the result:
    (int16 labels [X,Y,Z], (any over (y,z) [X], any over (x,z) [Y], any over (x,y) [Z]) as uint8)."""
    _f32(logits)
    if svs.dtype != torch.int16 or svs.dim() != 3:
        raise _lib.GtsError("supervoxel partitioning must be a 3-D int16 volume")
    svs, logits = svs.contiguous(), logits.contiguous()
    require_device(svs, logits)
    dx, dy, dz = svs.shape
    out = torch.empty(svs.shape, dtype=torch.int16, device=svs.device)
    occ = torch.zeros(dx + dy + dz, dtype=torch.uint8, device=svs.device)
    if svs.numel():
        check(_lib.load().gts_project_argmax_occupancy_i16(ptr(svs), ptr(logits), ptr(out), ptr(occ), dx, dy, dz,
                                                           logits.shape[0], logits.shape[1], current_stream()),
              "gts_project_argmax_occupancy_i16")
    return out, (occ[:dx], occ[dx:dx + dy], occ[dx + dy:])


class CropBox:
    """Outer product of three ascending plane-index vectors (what np.ix_ of boolean plane masks
    selects) inside a [X, Y, Z] volume; validated on the host, kept as int32 on the device."""

    def __init__(self, xs, ys, zs, volume_shape, device):
        import numpy as np

        self.volume_shape = tuple(int(d) for d in volume_shape)
        self.host = []
        for idx, extent in zip((xs, ys, zs), self.volume_shape):
            idx = np.asarray(idx).reshape(-1).astype(np.int64)
            if idx.size and (idx.min() < 0 or idx.max() >= extent or np.any(np.diff(idx) <= 0)):
                raise _lib.GtsError("crop indices must be strictly ascending and inside the volume")
            self.host.append(idx)
        self.shape = tuple(len(i) for i in self.host)
        self.dev = [torch.from_numpy(i.astype(np.int32)).to(device) for i in self.host]

    def as_ix(self):
        import numpy as np

        return np.ix_(*self.host)


def crop_concat(img, svs, table, bg_row, box):
    """K16.  [1, Ci + Ct, cx, cy, cz] fp32 = cat([img, table_plus_bg[svs]], -1)[box] moved to
    channels-first, without materialising the voxel-logit volume.  img [X,Y,Z,Ci] fp32 (or None),
    svs [X,Y,Z] int16, table [N,Ct] fp32, bg_row [Ct]."""
    _f32(img, table, bg_row)
    if svs.dtype != torch.int16 or tuple(svs.shape) != box.volume_shape:
        raise _lib.GtsError("partitioning must be int16 with the crop box's volume shape")
    ci = 0
    if img is not None:
        if img.dim() != 4 or tuple(img.shape[:3]) != box.volume_shape:
            raise _lib.GtsError("image must be [X, Y, Z, C] over the same volume")
        img, ci = img.contiguous(), img.shape[3]
    svs, table, bg_row = svs.contiguous(), table.contiguous(), bg_row.contiguous()
    require_device(svs, table, bg_row, img, *box.dev)
    ct = table.shape[1]
    if bg_row.numel() != ct:
        raise _lib.GtsError("background row does not match table rows")
    cx, cy, cz = box.shape
    out = torch.empty((1, ci + ct, cx, cy, cz), dtype=torch.float32, device=svs.device)
    if out.numel():
        check(_lib.load().gts_crop_concat_f32(ptr(img), ptr(svs), ptr(table), ptr(bg_row), ptr(box.dev[0]),
                                              ptr(box.dev[1]), ptr(box.dev[2]), ptr(out), cx, cy, cz,
                                              box.volume_shape[1], box.volume_shape[2], table.shape[0], ci, ct,
                                              current_stream()), "gts_crop_concat_f32")
    return out


def argmax_scatter(scores, box, relabel=None):
    """K17.  int16 [X,Y,Z] volume, zero outside the box, arg-max over channels of
    scores [C, cx, cy, cz] (or [1, C, ...]) inside."""
    _f32(scores)
    if scores.dim() == 5 and scores.shape[0] == 1:
        scores = scores[0]
    if scores.dim() != 4 or tuple(scores.shape[1:]) != box.shape:
        raise _lib.GtsError("scores must be [C, cx, cy, cz] over the crop box")
    scores = scores.contiguous()
    require_device(scores, relabel, *box.dev)
    if relabel is not None and (relabel.dtype != torch.int16 or relabel.numel() < scores.shape[0]):
        raise _lib.GtsError("relabel must be int16 with one entry per class")
    out = torch.zeros(box.volume_shape, dtype=torch.int16, device=scores.device)
    cx, cy, cz = box.shape
    if scores.numel():
        check(_lib.load().gts_argmax_scatter_i16(ptr(scores), ptr(relabel), ptr(box.dev[0]), ptr(box.dev[1]),
                                                 ptr(box.dev[2]), ptr(out), cx, cy, cz, box.volume_shape[1],
                                                 box.volume_shape[2], scores.shape[0], current_stream()),
              "gts_argmax_scatter_i16")
    return out


def label_confusion(pred, truth):
    """K15.  int64 [5, 5] table on the device: entry [cp, ct] counts the positions where the
    predicted label has class cp and the true label class ct (class = label for 0..3, 4 for
    anything else).  pred / truth: int16 tensors of equal size."""
    if pred.dtype != torch.int16 or truth.dtype != torch.int16:
        raise _lib.GtsError("label_confusion takes int16 labels")
    if pred.numel() != truth.numel():
        raise _lib.GtsError(f"label_confusion: {pred.numel()} predictions vs {truth.numel()} labels")
    pred, truth = pred.contiguous(), truth.contiguous()
    require_device(pred, truth)
    counts = torch.zeros((5, 5), dtype=torch.int64, device=pred.device)
    if pred.numel():
        lib = _lib.load()
        scratch = torch.empty(lib.gts_label_confusion_workspace(pred.numel()), dtype=torch.uint8,
                              device=pred.device)
        check(lib.gts_label_confusion_i16(ptr(pred), ptr(truth), ptr(counts), ptr(scratch), scratch.numel(),
                                          pred.numel(), current_stream()), "gts_label_confusion_i16")
    return counts


# ---------------------------------------------------------------- class-weighted cross-entropy
class _WeightedCE(torch.autograd.Function):
    """(logits [N,C], labels int64 [N], class_w [C] or None) -> stats = [num, den, num/den]."""

    @staticmethod
    def forward(ctx, logits, labels, class_w):
        logits, labels = logits.contiguous(), labels.contiguous()
        _f32(logits, class_w)
        require_device(logits, labels, class_w)
        if logits.dim() != 2 or labels.dtype != torch.int64 or labels.shape != logits.shape[:1]:
            raise _lib.GtsError("weighted CE takes logits [N, C] and int64 labels [N]")
        n, c = logits.shape
        _expect(class_w, (c,), "class_w")
        lib = _lib.load()
        ws = torch.empty(max(1, lib.gts_weighted_ce_workspace(n) // 4), dtype=torch.float32, device=logits.device)
        grad = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        stats = torch.empty(3, dtype=torch.float32, device=logits.device)
        check(lib.gts_weighted_ce_f32(ptr(logits), ptr(labels), ptr(class_w), ptr(grad), ptr(ws), ws.numel() * 4,
                                      ptr(stats), n, c, current_stream()), "gts_weighted_ce_f32")
        ctx.save_for_backward(grad, stats)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)                 # an unused output arrives as None, not as a zero tensor
        return stats[2].clone(), stats[0].clone(), stats

    @staticmethod
    def backward(ctx, g_mean, g_sum, _g_stats):
        grad_unscaled, stats = ctx.saved_tensors
        # d(num)/dx = gu ; d(num/den)/dx = gu / den
        if g_mean is None and g_sum is None:
            return None, None, None
        if g_sum is None:
            scale = g_mean / stats[1]
        elif g_mean is None:
            scale = g_sum
        else:
            scale = g_sum + g_mean / stats[1]
        return grad_unscaled * scale, None, None


def weighted_cross_entropy(logits, labels, class_w=None, reduction="mean"):
    """F.cross_entropy(logits, labels, weight=class_w, reduction=...) as one fused HIP pass.
    'mean' is the weighted mean (sum_i w[y_i] nll_i / sum_i w[y_i]), 'sum' the numerator."""
    mean, total, _ = _WeightedCE.apply(logits, labels, class_w)
    if reduction == "mean":
        return mean
    if reduction == "sum":
        return total
    raise ValueError(reduction)


def weighted_ce_numerator_grad(logits, labels, class_w=None):
    """(d numerator / d logits [N, C], [num, den, num/den]) from one launch, outside autograd: what the
    data-parallel step back-propagates with `logits.backward(grad)` — no clone / fill / scale kernels around
    the loss (the normalisation by the GLOBAL denominator happens after the all-reduce)."""
    logits_c, labels = logits.detach().contiguous(), labels.contiguous()
    _f32(logits_c, class_w)
    require_device(logits_c, labels, class_w)
    if logits_c.dim() != 2 or labels.dtype != torch.int64 or labels.shape != logits_c.shape[:1]:
        raise _lib.GtsError("weighted CE takes logits [N, C] and int64 labels [N]")
    n, c = logits_c.shape
    _expect(class_w, (c,), "class_w")
    lib = _lib.load()
    ws = torch.empty(max(1, lib.gts_weighted_ce_workspace(n) // 4), dtype=torch.float32, device=logits_c.device)
    grad = torch.empty_like(logits_c)
    stats = torch.empty(3, dtype=torch.float32, device=logits_c.device)
    check(lib.gts_weighted_ce_f32(ptr(logits_c), ptr(labels), ptr(class_w), ptr(grad), ptr(ws), ws.numel() * 4,
                                  ptr(stats), n, c, current_stream()), "gts_weighted_ce_f32")
    return grad, stats


def weighted_cross_entropy_stats(logits, labels, class_w=None):
    """(numerator, [num, den, num/den]) — the numerator carries the gradient (used by the
    data-parallel step, which normalises by the global denominator after the all-reduce)."""
    _, total, stats = _WeightedCE.apply(logits, labels, class_w)
    return total, stats
