"""SAGEConv / GATConv for MI355X: the layer modules model/networks.py builds its stacks from.

Counterparts of dgl.nn.pytorch.conv.SAGEConv / dgl.nn.pytorch.GATConv as constructed at
/root/reference/model/networks.py:25,28,30 and :46,52,56 — same constructor arguments (in
the positional order the reference uses), same parameter names and shapes in
`state_dict()`, same initialisation scheme; the arithmetic runs in the HIP kernels of
libgts_hip.so (neighbour reducers, attention) plus dense fp32 GEMMs.

The pool layer is ONE autograd node (`_SagePoolLayer`): forward and backward are written
out by hand so that the ReLU of fc_pool is folded into the max-pool backward kernel, the
argmax is kept as one byte per element, and nothing but (h, m, arg, out) is retained.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dense, ops


class GraphError(RuntimeError):
    """Counterpart of dgl.DGLError for the checks the layers make."""


def _is_relu(fn):
    return fn in (F.relu, torch.relu, torch.nn.functional.relu)


class _SagePoolLayer(torch.autograd.Function):
    """out = act(h Ws^T + maxpool_g(relu(h Wp^T + bp)) Wn^T + b)."""

    @staticmethod
    def forward(ctx, g, h, w_pool, b_pool, w_self, w_neigh, bias, relu_out, need_bwd):
        h = h.contiguous()
        p = dense.linear_fwd(h, w_pool, bias=b_pool, relu=True)
        # relu_input: K1 keeps no winner where the maximum is not positive, which IS ReLU'(p) for
        # the backward; p itself is not retained
        m, arg = ops.spmm_max_fwd(g, p, want_arg=need_bwd, relu_input=True)
        out = dense.linear_fwd(h, w_self, m, w_neigh, bias=bias, relu=relu_out)
        if need_bwd:
            ctx.g, ctx.relu_out = g, relu_out
            ctx.save_for_backward(h, m, arg, out if relu_out else None, w_pool, w_self, w_neigh)
        return out

    @staticmethod
    def backward(ctx, gout):
        h, m, arg, out, w_pool, w_self, w_neigh = ctx.saved_tensors
        if not ctx.relu_out:
            g = gout.contiguous()
        elif out.shape[1] % 4 == 0 and out.shape[0] > 0:
            g, _ = ops.gat_act_bwd(gout.contiguous(), out, 2, want_bias_grad=False)   # gout where out > 0, one HIP pass
        else:
            g = dense.relu_bwd(gout, out)
        need = ctx.needs_input_grad
        gm = dense.linear_bwd_input(g, w_neigh)
        gp = ops.spmm_max_bwd(ctx.g, gm, arg)                  # ReLU'(p) is in the winner record
        g_ws, g_wn, g_wp, g_bias, g_bp = _pool_layer_weight_grads(g, gp, h, m, w_pool, w_self)
        gh = dense.linear_bwd_input(g, w_self, gp, w_pool) if need[1] else None   # one K=2N pass
        return None, gh, g_wp, g_bp, g_ws, g_wn, g_bias, None, None


class _SageSumLayer(torch.autograd.Function):
    """SAGEConv 'mean' / 'gcn' as ONE autograd node (DGL semantics as in the oracle's R-mean / R-gcn / R-lin):

        neigh = reduce_g(src)                                   K3/K4 (mean: / max(deg,1); gcn: (+self) / (deg+1))
        out   = act( [fc_self(h) +] fc_neigh-term + bias )      one GEMM launch, bias and ReLU in its epilogue

    where, when in_feats > out_feats, fc_neigh is applied BEFORE the aggregation (`lin_before_mp`) and the
    aggregated [N, out] term enters the GEMM through an identity weight (x * 1 + 0 ... : exact).  The backward
    runs the ReLU mask and the bias gradient in one pass (gts_gat_act_bwd_f32), the two weight gradients in one
    launch, and adds the fc_self path inside the reducer (`accum`) — no elementwise torch kernels."""

    @staticmethod
    def forward(ctx, g, h, w_self, w_neigh, bias, mode, relu_out, need_bwd):
        h = h.contiguous()
        d = g.dev()
        fin, fout = w_neigh.shape[1], w_neigh.shape[0]
        lin_first = fin > fout
        div = d.deg_clamped if mode == "mean" else d.deg_plus1
        src = dense.linear_fwd(h, w_neigh) if lin_first else h
        neigh = ops.spmm_sum_raw(g, src, div_out=div, add_self=(mode == "gcn"))
        last_w = _identity(fout, h.device) if lin_first else w_neigh
        if w_self is not None:
            out = dense.linear_fwd(h, w_self, neigh, last_w, bias=bias, relu=relu_out)
        else:
            out = dense.linear_fwd(neigh, last_w, bias=bias, relu=relu_out)
        if need_bwd:
            ctx.g, ctx.mode, ctx.relu_out, ctx.lin_first = g, mode, relu_out, lin_first
            ctx.has_bias = bias is not None
            ctx.save_for_backward(h, neigh if not lin_first else None, out if relu_out else None, w_self, w_neigh)
        return out

    @staticmethod
    def backward(ctx, gout):
        h, neigh, out, w_self, w_neigh = ctx.saved_tensors
        g, mode = ctx.g, ctx.mode
        d = g.dev()
        div = d.deg_clamped if mode == "mean" else d.deg_plus1
        need = ctx.needs_input_grad
        g_pre, g_bias = ops.gat_act_bwd(gout.contiguous(), out, 2 if ctx.relu_out else 0, want_bias_grad=ctx.has_bias)

        def reduce_bwd(t, accum=None):   # autograd of the reducer: the same sum over the out-CSR
            return ops.spmm_sum_raw(g, t, transposed=True, div_in=div, add_self=(mode == "gcn"), accum=accum)

        g_ws = None
        if ctx.lin_first:
            # out = fc_self(h) + reduce(fc_neigh(h)) + b: the gradient of the [N, out] neighbour term is
            # reduce^T(g_pre); both weight matrices then multiply h
            gt = reduce_bwd(g_pre)
            if w_self is not None:
                (g_ws, _), (g_wn, _) = dense.linear_bwd_weight_multi([(g_pre, h, False), (gt, h, False)])
                gh = dense.linear_bwd_input(g_pre, w_self, gt, w_neigh) if need[1] else None
            else:
                g_wn, _ = dense.linear_bwd_weight(gt, h)
                gh = dense.linear_bwd_input(gt, w_neigh) if need[1] else None
        else:
            if w_self is not None:
                (g_ws, _), (g_wn, _) = dense.linear_bwd_weight_multi([(g_pre, h, False), (g_pre, neigh, False)])
            else:
                g_wn, _ = dense.linear_bwd_weight(g_pre, neigh)
            gh = None
            if need[1]:
                gneigh = dense.linear_bwd_input(g_pre, w_neigh)
                gself = dense.linear_bwd_input(g_pre, w_self) if w_self is not None else None
                gh = reduce_bwd(gneigh, accum=gself)
        return None, gh, g_ws, g_wn, g_bias, None, None, None


_identities = {}


def _identity(n, device):
    key = (n, device)
    if key not in _identities:
        _identities[key] = torch.eye(n, dtype=torch.float32, device=device)
    return _identities[key]


# GTS_OVERLAP_WGRAD=1 runs the weight gradients of the fused stack on a second, low-priority stream
# (see _SagePoolStack.backward).  Off by default: with the current kernels the side stream takes
# CUs from the input-gradient chain it was meant to fill in behind (757 graphs/s with it, 777-789
# without, profiles/r01_tune_gemm.log).
OVERLAP_WEIGHT_GRADS = os.environ.get("GTS_OVERLAP_WGRAD", "0") != "0"
# GTS_TRANSPOSED_IGRAD=0 keeps the strided-weight input-gradient kernel in the fused stack (A/B runs).
TRANSPOSED_IGRAD = os.environ.get("GTS_TRANSPOSED_IGRAD", "1") != "0"
_side_streams = {}


def _side_stream(device):
    s = _side_streams.get(device)
    if s is None:
        try:
            priority = torch.cuda.Stream.priority_range()[0]   # least urgent
        except Exception:                                       # noqa: BLE001 - older runtimes
            priority = 0
        s = torch.cuda.Stream(device=device, priority=priority)
        _side_streams[device] = s
    return s


# GTS_CHAIN_GEMMS=0 keeps one launch per GEMM in the fused stack (A/B runs).
CHAIN_LAYER_GEMMS = os.environ.get("GTS_CHAIN_GEMMS", "1") != "0"
# GTS_RELU_BITS=0: the backward of the fused stack reads its ReLU masks from the saved activations (floats)
# instead of the bit masks the forward GEMMs record (A/B runs).
RELU_MASK_BITS = os.environ.get("GTS_RELU_BITS", "1") != "0"


def _chainable(a0, w0, a1, w2):
    """Shapes the chained launch takes: every width a multiple of 4, the intermediate at most 256 wide."""
    n, k0, k1, n2 = w0.shape[0], a0.shape[1], a1.shape[1], w2.shape[0]
    return n % 4 == 0 and k0 % 4 == 0 and k1 % 4 == 0 and n <= 256 and n2 <= 256 and w2.shape[1] == n


def _pool_layer_weight_grads(g, gp, h, m, w_pool, w_self):
    """(g_ws, g_wn, g_wp, g_bias, g_bp) of one pool layer; one launch when Fin == Fout."""
    if w_pool.shape == w_self.shape:
        (g_ws, g_bias), (g_wn, _), (g_wp, g_bp) = dense.linear_bwd_weight_multi(
            [(g, h, True), (g, m, False), (gp, h, True)])
    else:
        (g_ws, g_bias), (g_wn, _) = dense.linear_bwd_weight_multi([(g, h, True), (g, m, False)])
        g_wp, g_bp = dense.linear_bwd_weight(gp, h, want_bias_grad=True)
    return g_ws, g_wn, g_wp, g_bias, g_bp


def _pack_by_shape(weights, transposed):
    """{id(w): fragment-order copy} of `weights`, one gts_pack_weights_f32 launch per weight shape."""
    by_shape, out = {}, {}
    for w in weights:
        by_shape.setdefault(tuple(w.shape), []).append(w)
    for ws in by_shape.values():
        for w, wp in zip(ws, dense.pack_weights(ws, transposed=transposed)):
            out[id(w)] = wp
    return out


class _SagePoolStack(torch.autograd.Function):
    """A whole stack of SAGEConv('pool') layers (ReLU on all but the last) as ONE autograd node.

    Same arithmetic as chaining `_SagePoolLayer`, but the ReLU backward of layer L-1 rides in the
    epilogue of layer L's input-gradient GEMM (`relu_mask` = that layer's input, which IS layer
    L-1's ReLU output), so no elementwise pass over [N, F] remains in the backward, and each
    activation is stored once (layer L's input is layer L-1's output)."""

    @staticmethod
    def forward(ctx, g, x, need_bwd, *params):
        n_layers = len(params) // 5
        h = x.contiguous()
        saved = []
        hbits = None                      # h > 0 as bits (h is a ReLU output from layer 1 on): the backward's mask
        p = None                          # relu(fc_pool(h)) of the layer about to run, when already computed
        # the weights the panel GEMMs may read (outputs wider than 128 columns), in fragment order: one launch per shape,
        # as gts_sage_pool_stack_fwd_f32 does (csrc/gts_stack.hip)
        fragment = _pack_by_shape([w for i in range(n_layers) for w in (params[5 * i], params[5 * i + 2], params[5 * i + 3])
                                   if w.shape[0] > 128 and w.shape[1] % 4 == 0], transposed=False)
        for i in range(n_layers):
            w_pool, b_pool, w_self, w_neigh, bias = params[5 * i:5 * i + 5]
            last = i == n_layers - 1
            if p is None:
                p = dense.linear_fwd(h, w_pool, bias=b_pool, relu=True, packed=(fragment.get(id(w_pool)), None))
            m, arg = ops.spmm_max_fwd(g, p, want_arg=need_bwd, relu_input=True)   # p is not kept
            p = None
            nxt = params[5 * (i + 1):5 * (i + 1) + 2] if not last else None
            obits = dense.relu_bits_empty(h.shape[0], w_self.shape[0], h.device) \
                if RELU_MASK_BITS and need_bwd and not last and dense.relu_bits_pay(h.shape[0], w_self.shape[0]) else None
            if CHAIN_LAYER_GEMMS and nxt is not None and _chainable(h, w_self, m, nxt[0]):
                # fc_self + fc_neigh of this layer and fc_pool of the next one in one launch
                out, p = dense.linear_fwd_chain(h, w_self, m, w_neigh, bias, True, nxt[0], nxt[1], True, relu_bits=obits,
                                                packed=(fragment.get(id(w_self)), fragment.get(id(w_neigh)), fragment.get(id(nxt[0]))))
            else:
                out = dense.linear_fwd(h, w_self, m, w_neigh, bias=bias, relu=not last, relu_bits=obits,
                                       packed=(fragment.get(id(w_self)), fragment.get(id(w_neigh))))
            saved += [h, m, arg, hbits]
            h, hbits = out, obits
        if need_bwd:
            ctx.g, ctx.n_layers = g, n_layers
            ctx.save_for_backward(*saved, *params)
        return h

    @staticmethod
    def backward(ctx, gout):
        n = ctx.n_layers
        tensors = ctx.saved_tensors
        acts, params = tensors[:4 * n], tensors[4 * n:]
        grads = [None] * (5 * n)
        g = gout.contiguous()            # gradient w.r.t. the pre-activation output of layer i
        gx = None
        # Nothing in the backward chain reads the weight gradients, so they are not computed layer
        # by layer: every (gradient, activation) pair is kept and all problems of one shape go into
        # ONE split-reduction launch at the end (the 256-wide layers of C2: 19 problems).  Few long
        # reductions per workgroup instead of many short ones, one slab reduction instead of seven.
        # (GTS_OVERLAP_WGRAD=1 restores the older scheme: per-layer launches on a second,
        # low-priority stream.)
        main = torch.cuda.current_stream()
        side = _side_stream(g.device) if OVERLAP_WEIGHT_GRADS else None
        keep_alive = []
        deferred = {}                    # (N, K) -> [(g [M,N], a [M,K], want_bias, grads slot, bias slot)]

        # Input gradients run in the forward GEMM's form on transposed weights (one batched
        # transpose per weight shape and backward pass; wide outputs only — the 4-wide first /
        # last layers keep the strided-operand kernel).
        turned, turned_fragment = {}, {}      # W^T row-major / in fragment order (the panel kernels), one launch per shape
        if TRANSPOSED_IGRAD:
            by_shape = {}
            for w in (params[5 * i + j] for i in range(n) for j in (0, 2, 3)):
                if w.shape[1] >= 128 and w.shape[0] % 4 == 0 and w.shape[1] % 4 == 0:
                    by_shape.setdefault(tuple(w.shape), []).append(w)
            for ws in by_shape.values():
                plain, packed = dense.pack_weights(ws, transposed=True, want_plain=True)
                for w, wt, wp in zip(ws, plain, packed):
                    turned[id(w)], turned_fragment[id(w)] = wt, wp

        def igrad(g0, w0, g1=None, w1=None, relu_mask=None, relu_bits=None):
            if id(w0) in turned and (w1 is None or id(w1) in turned) and g0.is_contiguous():
                return dense.linear_bwd_input_t(g0, turned[id(w0)], g1, turned[id(w1)] if w1 is not None else None,
                                                relu_mask=relu_mask, relu_bits=relu_bits,
                                                packed=(turned_fragment[id(w0)], turned_fragment[id(w1)] if w1 is not None else None))
            return dense.linear_bwd_input(g0, w0, g1, w1, relu_mask=relu_mask)

        def defer(grad_out, act, slot, bias_slot):
            key = (grad_out.shape[1], act.shape[1])
            deferred.setdefault(key, []).append((grad_out, act, bias_slot is not None, slot, bias_slot))

        gm = None                         # g @ W_neigh of the layer about to run, when already computed
        for i in reversed(range(n)):
            h, m, arg, hbits = acts[4 * i:4 * i + 4]
            w_pool, _b_pool, w_self, w_neigh, _bias = params[5 * i:5 * i + 5]
            if gm is None:
                gm = igrad(g, w_neigh)
            gp = ops.spmm_max_bwd(ctx.g, gm, arg)       # ReLU'(p) is already in the winner record
            gm = None
            if side is None:
                defer(gp, h, 5 * i, 5 * i + 1)          # fc_pool.weight, fc_pool.bias
                defer(g, h, 5 * i + 2, 5 * i + 4)       # fc_self.weight, bias
                defer(g, m, 5 * i + 3, None)            # fc_neigh.weight
            else:
                ready = torch.cuda.Event()
                ready.record(main)                      # g and gp are complete on the main stream
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    g_ws, g_wn, g_wp, g_bias, g_bp = _pool_layer_weight_grads(g, gp, h, m, w_pool, w_self)
                keep_alive.append((g, gp))              # still being read by the side stream
                grads[5 * i:5 * i + 5] = [g_wp, g_bp, g_ws, g_wn, g_bias]
            if i > 0:      # h is layer i-1's ReLU output: its backward is the mask h > 0
                below = params[5 * (i - 1) + 3]          # W_neigh of the layer below
                if CHAIN_LAYER_GEMMS and all(id(w) in turned for w in (w_self, w_pool, below)) \
                        and _chainable(g, w_self.t(), gp, below.t()):
                    # this layer's input gradient and the next one's g @ W_neigh in one launch
                    g, gm = dense.linear_bwd_input_chain_t(g, turned[id(w_self)], gp, turned[id(w_pool)], h,
                                                           turned[id(below)], relu_bits=hbits,
                                                           packed=(turned_fragment[id(w_self)], turned_fragment[id(w_pool)],
                                                                   turned_fragment[id(below)]))
                else:
                    g = igrad(g, w_self, gp, w_pool, relu_mask=h, relu_bits=hbits)
            elif ctx.needs_input_grad[1]:
                gx = igrad(g, w_self, gp, w_pool)
        for problems in deferred.values():
            results = dense.linear_bwd_weight_multi([(go, act, want) for go, act, want, _, _ in problems])
            for (gw, gb), (_, _, _, slot, bias_slot) in zip(results, problems):
                grads[slot] = gw
                if bias_slot is not None:
                    grads[bias_slot] = gb
        if side is not None:
            main.wait_stream(side)
            for t in grads:
                t.record_stream(main)                   # allocated on `side`, consumed on `main`
            for pair in keep_alive:
                for t in pair:
                    t.record_stream(side)
        return (None, gx, None, *grads)


# GTS_STACK_C=0 keeps the launch-by-launch Python loop of the fused stack (A/B runs; bench.py's instrumented block
# uses it too, because HIP events are bracketed around single launches).
STACK_IN_ONE_CALL = os.environ.get("GTS_STACK_C", "1") != "0"


class GradSink:
    """One flat fp32 buffer laid out like an optimizer's flat parameters (`params` in order, then `extra` spare
    floats).  While a sink is active (`with grad_sink(sink)`), the fused pool stack writes its weight gradients
    straight into `sink.flat` and hands autograd no per-parameter tensors — nothing is concatenated afterwards
    (gts.optim.FlatAdamW.step(flat_grad=...), gts.dist.FlatGradSync).  `filled` says whether that happened."""

    def __init__(self, params, extra=0):
        self.params = list(params)
        self.offsets, at = {}, 0
        for p in self.params:
            self.offsets[id(p)] = at
            at += p.numel()
        self.n, self.extra = at, int(extra)
        self.flat, self.filled = None, False

    def new_buffer(self):
        self.flat = torch.empty(self.n + self.extra, dtype=torch.float32, device=self.params[0].device)
        self.filled = False
        return self.flat


_active_sink = None


class grad_sink:
    def __init__(self, sink):
        self.sink = sink

    def __enter__(self):
        global _active_sink
        self.previous, _active_sink = _active_sink, self.sink
        return self.sink

    def __exit__(self, *exc):
        global _active_sink
        _active_sink = self.previous
        return False


_stack_tables = {}   # (parameter storage addresses) -> (pointer table, widths) as ctypes arrays, built once per network


def _stack_table(params, in_feats):
    import ctypes

    # addresses AND shapes: the caching allocator may hand a freed network's addresses to one with other widths
    ptrs = [p.data_ptr() for p in params]
    key = (in_feats, *ptrs, *[tuple(p.shape) for p in params])
    hit = _stack_tables.get(key)
    if hit is None:
        n_layers = len(params) // 5
        widths = [in_feats] + [params[5 * i + 2].shape[0] for i in range(n_layers)]
        hit = ((ctypes.c_void_p * (5 * n_layers))(*ptrs), (ctypes.c_int64 * (n_layers + 1))(*widths), widths)
        if len(_stack_tables) > 64:
            _stack_tables.clear()
        _stack_tables[key] = hit
    return hit


def _stack_flags():
    return (1 if CHAIN_LAYER_GEMMS else 0) | (2 if RELU_MASK_BITS else 0) | (4 if TRANSPOSED_IGRAD else 0)


class _SagePoolStackCall(torch.autograd.Function):
    """`_SagePoolStack` behind one C-ABI call each way (gts_sage_pool_stack_fwd_f32 / _bwd_f32: the same launches in
    the same order, enqueued by the library; bit-identical, tests/test_gpu_stack.py).  Activations and winners live
    in ONE arena tensor; gradients go into one flat buffer (the active GradSink's, or one of their own)."""

    @staticmethod
    def forward(ctx, g, x, need_bwd, *params):
        import ctypes

        from . import _lib
        from ._lib import check, current_stream, ptr

        lib = _lib.load()
        x = x.contiguous()
        n, n_layers = x.shape[0], len(params) // 5
        table, c_widths, widths = _stack_table(params, x.shape[1])
        flags, ab = _stack_flags(), g.arg_bytes
        d = g.dev()
        offsets = (ctypes.c_int64 * (4 * n_layers + 2))()
        total = lib.gts_sage_pool_stack_fwd_arena(n, c_widths, n_layers, 1 if need_bwd else 0, ab, flags, offsets)
        if total < 0:
            raise _lib.GtsError("gts_sage_pool_stack_fwd_arena rejected the stack's shape")
        arena = torch.empty(max(int(total), 16), dtype=torch.uint8, device=x.device)
        ds = ops._cluster_schedule(g, "in", n, 256, ab if need_bwd else 0) if 256 in widths[:-1] else None
        h = ds.host if ds is not None else None
        check(lib.gts_sage_pool_stack_fwd_f32(
            ptr(d.indptr), ptr(d.indices), ptr(ds.packed) if ds else None, h.n_clusters if h else 0,
            h.limits[0] if h else 0, h.limits[1] if h else 0, h.loc_words if h else 0, ptr(x), table, n, c_widths,
            n_layers, 1 if need_bwd else 0, ab, flags, arena.data_ptr(), int(total), current_stream()),
            "gts_sage_pool_stack_fwd_f32")
        at = int(offsets[4 * (n_layers - 1) + 2])
        out = arena[at:at + 4 * n * widths[-1]].view(torch.float32).view(n, widths[-1])
        if need_bwd:
            ctx.g, ctx.widths, ctx.flags, ctx.ab = g, widths, flags, ab
            ctx.param_ids = [id(p) for p in params]
            ctx.save_for_backward(x, arena, *params)
        return out

    @staticmethod
    def backward(ctx, gout):
        import ctypes

        from . import _lib
        from ._lib import check, current_stream, ptr

        lib = _lib.load()
        x, arena, *params = ctx.saved_tensors
        g, widths, flags, ab = ctx.g, ctx.widths, ctx.flags, ctx.ab
        n, n_layers = x.shape[0], len(params) // 5
        gout = gout.contiguous()
        d = g.dev()
        sink = _active_sink
        # the sink stands for the optimizer's WHOLE flat gradient: it is used only when this stack holds exactly its
        # parameters.  A network with parameters outside the stack gets ordinary per-parameter gradients (views of one
        # buffer) from this node, and autograd fills in the rest — nothing is dropped silently.
        if sink is not None and (len(ctx.param_ids) != len(sink.params) or
                                 not all(i in sink.offsets for i in ctx.param_ids)):
            sink = None
        sizes = [p.numel() for p in params]
        if sink is not None:
            flat, starts = sink.flat, [sink.offsets[i] for i in ctx.param_ids]
        else:
            flat = torch.empty(sum(sizes), dtype=torch.float32, device=x.device)
            starts = [0] * len(sizes)
            for q in range(1, len(sizes)):
                starts[q] = starts[q - 1] + sizes[q - 1]
        base = flat.data_ptr()
        grads = (ctypes.c_void_p * (5 * n_layers))(*[base + 4 * s for s in starts])
        table, c_widths, _ = _stack_table(params, x.shape[1])
        need = lib.gts_sage_pool_stack_bwd_scratch(n, c_widths, n_layers, flags)
        scratch = torch.empty(max(int(need), 16), dtype=torch.uint8, device=x.device)
        gx = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        ds = ops._cluster_schedule(g, "out", n, 256, ab) if 256 in widths[:-1] else None
        h = ds.host if ds is not None else None
        check(lib.gts_sage_pool_stack_bwd_f32(
            ptr(d.t_indptr), ptr(d.t_indices), ptr(d.t_slot), ptr(ds.packed) if ds else None, h.n_clusters if h else 0,
            h.limits[0] if h else 0, h.limits[1] if h else 0, h.loc_words if h else 0, ptr(gout), ptr(x), table, n,
            c_widths, n_layers, ab, flags, arena.data_ptr(), grads, ptr(gx), scratch.data_ptr(), int(need),
            current_stream()), "gts_sage_pool_stack_bwd_f32")
        if sink is not None:
            sink.filled = True
            return (None, gx, None, *([None] * len(params)))
        return (None, gx, None, *[flat[s:s + k].view_as(p) for s, k, p in zip(starts, sizes, params)])


def sage_pool_stack(graph, features, layers):
    """Run `layers` (SAGEConv pool modules: ReLU on all but the last, no active dropout, bias on)
    as one fused autograd node.  Returns None when the stack does not have that shape, so the
    caller can fall back to the layer-by-layer path."""
    for i, layer in enumerate(layers):
        last = i == len(layers) - 1
        if not isinstance(layer, SAGEConv) or layer._aggre_type != "pool" or layer.bias is None \
                or layer.norm is not None or (layer.feat_drop.p > 0 and layer.training):
            return None
        if (layer.activation is not None) if last else (not _is_relu(layer.activation)):
            return None
    params = []
    for layer in layers:
        params += [layer.fc_pool.weight, layer.fc_pool.bias, layer.fc_self.weight,
                   layer.fc_neigh.weight, layer.bias]
    need_bwd = torch.is_grad_enabled() and (features.requires_grad or any(p.requires_grad for p in params))
    one_call = STACK_IN_ONE_CALL and not ops.INSTRUMENTED and not OVERLAP_WEIGHT_GRADS and features.shape[1] % 4 == 0 \
        and all(p.shape[-1] % 4 == 0 and p.shape[0] % 4 == 0 and p.is_contiguous() for p in params) \
        and features.dtype == torch.float32 and features.is_cuda
    return (_SagePoolStackCall if one_call else _SagePoolStack).apply(graph, features, need_bwd, *params)


class SAGEConv(nn.Module):
    """GraphSAGE layer, aggregator_type in {'mean', 'gcn', 'pool'} ('lstm' is not built:
    no reference script reaches it, model/networks.py:68-81)."""

    def __init__(self, in_feats, out_feats, aggregator_type, feat_drop=0.0, bias=True, norm=None,
                 activation=None):
        super().__init__()
        if aggregator_type not in ("mean", "gcn", "pool"):
            raise KeyError(f"Invalid aggregator_type. Must be one of mean, gcn, pool. "
                           f"But got {aggregator_type!r} instead.")
        self._in_src_feats = self._in_dst_feats = int(in_feats)
        self._out_feats = int(out_feats)
        self._aggre_type = aggregator_type
        self.norm = norm
        self.feat_drop = nn.Dropout(feat_drop)
        self.activation = activation
        if aggregator_type == "pool":
            self.fc_pool = nn.Linear(in_feats, in_feats)
        if aggregator_type != "gcn":
            self.fc_self = nn.Linear(in_feats, out_feats, bias=False)
        self.fc_neigh = nn.Linear(in_feats, out_feats, bias=False)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_feats))
        else:
            self.register_buffer("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        if self._aggre_type == "pool":
            nn.init.xavier_uniform_(self.fc_pool.weight, gain=gain)
        if self._aggre_type != "gcn":
            nn.init.xavier_uniform_(self.fc_self.weight, gain=gain)
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # Older DGL checkpoints carry fc_self.bias / fc_neigh.bias instead of one `bias`.
        legacy = [prefix + "fc_self.bias", prefix + "fc_neigh.bias"]
        if prefix + "bias" not in state_dict and any(k in state_dict for k in legacy):
            folded = sum(state_dict.pop(k) for k in legacy if k in state_dict)
            state_dict[prefix + "bias"] = folded
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, graph, feat):
        h = self.feat_drop(feat)
        fused_relu = _is_relu(self.activation)
        if self._aggre_type == "pool":
            bias = self.bias if self.bias is not None else torch.zeros(
                self._out_feats, dtype=h.dtype, device=h.device)
            need_bwd = torch.is_grad_enabled() and (
                h.requires_grad or any(p.requires_grad for p in self.parameters()))
            rst = _SagePoolLayer.apply(graph, h, self.fc_pool.weight, self.fc_pool.bias,
                                       self.fc_self.weight, self.fc_neigh.weight, bias,
                                       fused_relu, need_bwd)
        elif self._out_feats % 4 == 0 and h.shape[0] > 0:
            need_bwd = torch.is_grad_enabled() and (
                h.requires_grad or any(p.requires_grad for p in self.parameters()))
            w_self = self.fc_self.weight if self._aggre_type != "gcn" else None
            rst = _SageSumLayer.apply(graph, h, w_self, self.fc_neigh.weight, self.bias, self._aggre_type,
                                      fused_relu, need_bwd)
        else:   # odd output widths (the fused node's kernels work on 16-byte column groups): op by op
            lin_before_mp = self._in_src_feats > self._out_feats
            src = dense.linear(h, self.fc_neigh.weight) if lin_before_mp else h
            neigh = ops.spmm_reduce(graph, src, self._aggre_type)
            if not lin_before_mp:
                neigh = dense.linear(neigh, self.fc_neigh.weight)
            rst = neigh if self._aggre_type == "gcn" else dense.linear(h, self.fc_self.weight) + neigh
            if self.bias is not None:
                rst = rst + self.bias
            if fused_relu:
                rst = torch.relu(rst)
        if self.activation is not None and not fused_relu:
            rst = self.activation(rst)
        if self.norm is not None:
            rst = self.norm(rst)
        return rst


# One autograd node per GAT layer (scores, attention, aggregation, residual, bias and ELU fused into
# the K5-K8 kernels).  GTS_FUSE_GAT=0 keeps the op-by-op path (same kernels for the aggregation,
# torch for the small elementwise pieces).
FUSE_GAT_LAYER = os.environ.get("GTS_FUSE_GAT", "1") != "0"
FUSE_GAT_SCORES = os.environ.get("GTS_FUSE_GAT_SCORES", "1") != "0"   # el / er in the fc GEMM's epilogue (A/B switch)


FOLD_GAT_ACT_BWD = os.environ.get("GTS_FOLD_GAT_ACT", "1") != "0"   # A/B switch of ActLink


PACK_GAT_WEIGHTS = os.environ.get("GTS_PACK_GAT", "1") != "0"   # A/B switch: fragment-order copies of fc / res_fc weights


def _turn(ws):
    """(W^T row-major, W^T in fragment order or Nones) of same-shape weights, one launch (dense.pack_weights)."""
    if PACK_GAT_WEIGHTS:
        plain, packed = dense.pack_weights(ws, transposed=True, want_plain=True)
        return plain, tuple(packed) if len(packed) > 1 else (packed[0], None)
    return dense.transpose_batch(ws), (None, None)


class ActLink:
    """Joins a GATConv layer to the ONE layer that consumes its activated output (GAT.forward, reference
    model/networks.py:61-63: `h = self.layers[l](g, h).flatten(1)` feeds only the next layer).  In the backward pass the
    consumer's input-gradient GEMM then multiplies by ELU'(h) — its input h IS the producer's output — in its epilogue and
    sums the producer's bias gradient there (`dense.linear_bwd_input_t_act`), and the producer skips its own pass over
    [N, H*D] (`ops.gat_act_bwd`: three tensors of 245 MB at the C3 shapes).  Same g_pre bit for bit; the bias gradient is
    summed in a different fixed order.  The gradient autograd hands from consumer to producer is then d loss / d
    (pre-activation): do not link a layer whose output anything else reads.

    The link is two-sided: the PRODUCER arms it (`armed` = the activation code it will skip in its backward: 1 = ELU) only
    when it runs as the fused node with that activation and a backward; the consumer folds only through an armed link and
    with the producer's code.  A producer on the op-by-op path, with another activation callable or with none leaves the
    link unarmed, and both layers run their own backward passes."""
    __slots__ = ("folded", "g_bias", "armed")

    def __init__(self):
        self.folded, self.g_bias, self.armed = False, None, 0


class _GATLayer(torch.autograd.Function):
    """rst = act( softmax-attention aggregate of ft + res_fc(h) + bias ),  ft = h W^T viewed [N,H,D],
    el/er = <ft, attn_l/r>  — DGL GATConv (reference model/networks.py:46-58) as one node."""

    @staticmethod
    def forward(ctx, g, h, w_fc, attn_l, attn_r, bias, w_res, identity_res, slope, act_code, heads, dim,
                need_bwd, below=None, above=None):
        h = h.contiguous()
        n = h.shape[0]
        al, ar = attn_l.reshape(heads, dim), attn_r.reshape(heads, dim)
        if FUSE_GAT_SCORES and h.shape[1] % 4 == 0:
            # (fc.weight in fragment order buys the 1024-wide forward nothing — 593 against 591 us with the copy's launch,
            # profiles/r04 — so only the backward, whose transposes come with the copy for free, passes one)
            ft, el, er = ops.gat_fc_scores(h, w_fc, al, ar, heads, dim)     # el / er in the GEMM epilogue when tall
        else:
            ft = dense.linear_fwd(h, w_fc).view(n, heads, dim)
            el, er = ops.gat_scores(ft, al, ar)
        if w_res is not None:
            res = dense.linear_fwd(h, w_res)
        elif identity_res:
            res = h
        else:
            res = None
        out, attn = ops._gat_fwd(g, ft, el, er, slope, bias, res, act_code)
        if need_bwd:
            ctx.g, ctx.slope, ctx.act, ctx.identity_res = g, slope, act_code, identity_res
            ctx.save_for_backward(h, ft, el, er, attn, out if act_code else None, w_fc, al, ar, w_res)
            ctx.attn_shape = attn_l.shape
            ctx.below, ctx.above = below, above if act_code == 1 else None
            if ctx.above is not None:
                ctx.above.armed = act_code      # this node will skip its activation backward if the consumer folds it
        return out

    @staticmethod
    def backward(ctx, gout):
        h, ft, el, er, attn, out, w_fc, al, ar, w_res = ctx.saved_tensors
        n, heads, dim = ft.shape
        need = ctx.needs_input_grad
        if ctx.above is not None and ctx.above.folded:   # the layer above has applied ELU' and summed the bias gradient
            g_pre, g_bias = gout.reshape(n, heads * dim), ctx.above.g_bias if need[5] else None
            ctx.above.folded, ctx.above.g_bias, ctx.above.armed = False, None, 0
        else:
            g_pre, g_bias = ops.gat_act_bwd(gout.reshape(n, heads * dim),
                                            out.view(n, heads * dim) if out is not None else None, ctx.act,
                                            want_bias_grad=need[5])
        gft, gel, ger = ops._gat_bwd(ctx.g, ft, el, er, attn, g_pre.view(n, heads, dim), ctx.slope, al, ar)
        g_al, g_ar = ops.gat_param_grad(ft, gel, ger)
        gft2 = gft.view(n, heads * dim)
        g_wres = None
        # wide layers: the input gradient runs in the forward GEMM's form on transposed weights
        turn = TRANSPOSED_IGRAD and need[1] and w_fc.shape[1] >= 128 and w_fc.shape[0] % 4 == 0 \
            and w_fc.shape[1] % 4 == 0 and n >= 4096
        fold = turn and ctx.below is not None and ctx.below.armed == 1 and not ctx.identity_res and FOLD_GAT_ACT_BWD
        act_below = ctx.below.armed if fold else 0
        if w_res is not None:
            (g_wfc, _), (g_wres, _) = dense.linear_bwd_weight_multi([(gft2, h, False), (g_pre, h, False)])
            if fold:
                wt, wtp = _turn([w_fc, w_res])
                gh, ctx.below.g_bias = dense.linear_bwd_input_t_act(gft2, wt[0], h, act_below, g_pre, wt[1], packed=wtp)
                ctx.below.folded = True
            elif turn:
                wt, wtp = _turn([w_fc, w_res])
                gh = dense.linear_bwd_input_t(gft2, wt[0], g_pre, wt[1], packed=wtp)
            else:
                gh = dense.linear_bwd_input(gft2, w_fc, g_pre, w_res) if need[1] else None
        else:
            g_wfc, _ = dense.linear_bwd_weight(gft2, h)
            if fold:
                wt, wtp = _turn([w_fc])
                gh, ctx.below.g_bias = dense.linear_bwd_input_t_act(gft2, wt[0], h, act_below, packed=(wtp[0], None))
                ctx.below.folded = True
            elif turn:
                wt, wtp = _turn([w_fc])
                gh = dense.linear_bwd_input_t(gft2, wt[0], packed=(wtp[0], None))
            else:
                gh = dense.linear_bwd_input(gft2, w_fc) if need[1] else None
            if gh is not None and ctx.identity_res:
                gh = gh + g_pre
        return (None, gh, g_wfc, g_al.view(ctx.attn_shape), g_ar.view(ctx.attn_shape), g_bias, g_wres,
                None, None, None, None, None, None, None, None)


class GATConv(nn.Module):
    """Graph attention layer (positional order as called at model/networks.py:46-58)."""

    def __init__(self, in_feats, out_feats, num_heads, feat_drop=0.0, attn_drop=0.0,
                 negative_slope=0.2, residual=False, activation=None,
                 allow_zero_in_degree=False, bias=True):
        super().__init__()
        self._in_feats, self._out_feats, self._num_heads = int(in_feats), int(out_feats), int(num_heads)
        self._allow_zero_in_degree = allow_zero_in_degree
        self.fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)
        self.attn_l = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.attn_r = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.feat_drop = nn.Dropout(feat_drop)
        self.attn_drop = nn.Dropout(attn_drop)
        self.negative_slope = negative_slope
        if bias:
            self.bias = nn.Parameter(torch.empty(num_heads * out_feats))
        else:
            self.register_buffer("bias", None)
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
        if self.bias is not None:
            nn.init.constant_(self.bias, 0)
        if isinstance(self.res_fc, nn.Linear):
            nn.init.xavier_normal_(self.res_fc.weight, gain=gain)

    def forward(self, graph, feat, below=None, above=None):
        """`below` / `above`: optional ActLink objects shared with the GATConv that produced `feat` / the one that alone
        consumes this layer's output (set by model.networks.GAT.forward; see ActLink)."""
        if not self._allow_zero_in_degree and graph.min_in_degree == 0:
            raise GraphError(
                "There are 0-in-degree nodes in the graph, output for those nodes will be invalid. "
                "Adding self-loop on the input graph resolves the issue; setting "
                "allow_zero_in_degree=True suppresses the check.")
        if self.attn_drop.p > 0 and self.training:
            raise NotImplementedError("attention dropout is not built (the reference never sets it: "
                                      "model/networks.py:77-78 passes no dropout to GAT)")
        n = feat.shape[0]
        h = self.feat_drop(feat)
        if above is not None:
            above.armed = 0                      # armed below only by the fused node that will skip its ELU backward
        act_code = 0 if self.activation is None else 1 if self.activation in (F.elu, torch.nn.functional.elu) else -1
        if FUSE_GAT_LAYER and act_code >= 0 and self._out_feats % 4 == 0 and feat.dim() == 2:
            w_res = self.res_fc.weight if isinstance(self.res_fc, nn.Linear) else None
            identity_res = isinstance(self.res_fc, nn.Identity)
            need_bwd = torch.is_grad_enabled() and (
                h.requires_grad or any(p.requires_grad for p in self.parameters()))
            if self.feat_drop.p > 0 and self.training:   # a real dropout sits between the producer's output and this input
                below = None
            return _GATLayer.apply(graph, h, self.fc.weight, self.attn_l, self.attn_r, self.bias, w_res,
                                   identity_res, self.negative_slope, act_code, self._num_heads,
                                   self._out_feats, need_bwd, below, above)
        ft = dense.linear(h, self.fc.weight).view(n, self._num_heads, self._out_feats)
        el = (ft * self.attn_l).sum(dim=-1)
        er = (ft * self.attn_r).sum(dim=-1)
        rst = ops.gat_aggregate(graph, ft, el, er, self.negative_slope)
        if self.res_fc is not None:
            res = dense.linear(h, self.res_fc.weight) if isinstance(self.res_fc, nn.Linear) else h
            rst = rst + res.view(n, -1, self._out_feats)
        if self.bias is not None:
            rst = rst + self.bias.view(1, self._num_heads, self._out_feats)
        if self.activation is not None:
            rst = self.activation(rst)
        return rst
