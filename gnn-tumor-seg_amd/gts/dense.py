"""Dense layer GEMMs of the GNN path (K11) on the MI355X matrix cores.

Single seam for the dense work inside SAGEConv / GATConv: every function calls the
hand-written fp32 MFMA kernels of libgts_hip.so (gts_linear_*), never a BLAS library.
fp32 throughout (the reference trains in fp32; gfx950 has no TF32-like shortcut).

Shape rule of the kernels: every dimension that indexes a contiguous axis must be a multiple
of 4 (16-byte loads).  Feature widths of the reference (4, 20, 256, class count 4) satisfy
it; anything else is zero-padded here, on the host side of the ABI.
"""
import torch
import torch.nn.functional as F

from . import _lib
from ._lib import check, current_stream, ptr, require_device

# bench.py installs a callable here, timer(kind, flops, launch) with kind in {"fwd", "igrad",
# "wgrad"}, to bracket every K11 launch with HIP events; None in normal use.
GEMM_TIMER = None


class row_count_invariant:
    """Context manager: forward GEMMs inside pick their tile among the 32x32x2 variants only, which all
    walk the reduction in one order — a row's result is then independent of how many rows the call has
    (the 240-row panels chosen for very tall operands run on the 16x16x4 MFMA and round differently).
    Used where a batched forward must reproduce per-sample forwards bit for bit (GNN.evaluate)."""

    _depth = 0
    _saved = -1      # the forward-tile option in force outside the outermost block (tools / GTS_OPTIONS may have set one)
    _lock = __import__("threading").Lock()

    def __enter__(self):
        with row_count_invariant._lock:
            if row_count_invariant._depth == 0:
                lib = _lib.load()
                row_count_invariant._saved = int(lib.gts_get_option(1))
                check(lib.gts_set_option(1, -2), "gts_set_option")
            row_count_invariant._depth += 1
        return self

    def __exit__(self, *exc):
        with row_count_invariant._lock:
            row_count_invariant._depth -= 1
            if row_count_invariant._depth == 0:
                check(_lib.load().gts_set_option(1, row_count_invariant._saved), "gts_set_option")
        return False


def _timed(kind, flops, launch):
    return GEMM_TIMER(kind, flops, launch) if GEMM_TIMER is not None else launch()

_workspaces = {}


def _workspace(device, nbytes):
    """Caller-owned scratch for the split-reduction weight gradient (grown on demand, reused:
    kernels on one stream run in order, so one buffer per (device, stream) is enough)."""
    key = (device, current_stream())
    buf = _workspaces.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _workspaces[key] = buf
    return buf


def _pad4_cols(t):
    """Zero-pad the last dim of a 2-D tensor to a multiple of 4 (no copy when aligned)."""
    pad = (-t.shape[1]) % 4
    return t.contiguous() if pad == 0 else F.pad(t, (0, pad)).contiguous()


def _pad4_rows(t):
    pad = (-t.shape[0]) % 4
    return t.contiguous() if pad == 0 else F.pad(t, (0, 0, 0, pad)).contiguous()


def _mat(t, name):
    if t.dim() != 2:
        raise _lib.GtsError(f"{name} must be a matrix, got shape {tuple(t.shape)}")
    return t


def _same(a, b, what):
    if a != b:
        raise _lib.GtsError(f"shapes do not match: {what} ({a} vs {b})")


def _chk(*tensors):
    for t in tensors:
        if t is not None and t.dtype != torch.float32:
            raise _lib.GtsError(f"gts GEMMs are fp32: got {t.dtype}")
    return require_device(*tensors)


def _dense(*tensors):
    """The kernels read every operand as a dense row-major matrix with ld = shape[1]: make it so (no copy when it
    already is; a .t() view, a column slice or an expanded gradient is materialised here, never mis-read)."""
    return tuple(t if t is None or t.is_contiguous() else t.contiguous() for t in tensors)


def relu_bits_empty(m, n, device):
    """Buffer for the sign bits of an [m, n] activation (n % 64 == 0): what `relu_bits=` of the forward calls
    fills and `relu_bits=` of the transposed input-gradient calls reads (layout: include/gts_hip.h)."""
    nbytes = _lib.load().gts_relu_bits_bytes(m, n)
    if nbytes == 0 and m > 0:
        raise _lib.GtsError(f"relu bits need a width that is a multiple of 64, got {n}")
    return torch.empty(nbytes // 8, dtype=torch.int64, device=device)


def relu_bits_pay(m, n):
    """True when the GEMM launches of an [m, n] activation keep the mask bits inside their epilogues (tall operands);
    otherwise the bits would cost a pass of their own and the float mask is the better choice."""
    return _lib.load().gts_relu_bits_pay(m, n) == 1


def _chk_bits(bits, m, n, what):
    if bits is None:
        return
    if bits.dtype != torch.int64 or not bits.is_contiguous() or bits.numel() * 8 != _lib.load().gts_relu_bits_bytes(m, n):
        raise _lib.GtsError(f"shapes do not match: {what} must come from relu_bits_empty({m}, {n})")


def unpack_relu_bits(bits, m, n):
    """[m, n] bool tensor from the bit layout (tests / debugging; a few torch ops, not a hot path)."""
    words = bits.view(n // 64, (m + 3) // 4, 4)                            # [column block, row group, e]
    lanes = torch.arange(64, device=bits.device, dtype=torch.int64)
    on = ((words.unsqueeze(-1) >> lanes) & 1).bool()                       # [..., e, lane = 16 * (row % 4) + j]
    on = on.view(n // 64, -1, 4, 4, 16).permute(1, 3, 0, 4, 2)             # [group, row % 4, block, j, e]
    return on.reshape(-1, n)[:m]


def _packed_array(packed, weights):
    """ctypes pointer table for the `packed` argument of the GEMM entry points: one entry per weight operand (None
    where there is no fragment-order copy), or None when there is none at all.  A copy must come from pack_weights of
    the operand it stands beside (size checked here; the bytes are the caller's business)."""
    import ctypes

    if packed is None or all(q is None for q in packed):
        return None
    if len(packed) != len(weights):
        raise _lib.GtsError(f"packed= takes one entry per weight operand ({len(weights)}), got {len(packed)}")
    for q, w in zip(packed, weights):
        if q is None:
            continue
        if w is None:
            raise _lib.GtsError("a fragment-order copy was passed for a weight operand that is absent")
        want = _lib.load().gts_packed_weight_floats(w.shape[0], w.shape[1])
        if q.dtype != torch.float32 or not q.is_contiguous() or q.numel() != want or q.device != w.device:
            raise _lib.GtsError(f"shapes do not match: fragment-order copy of a {tuple(w.shape)} operand holds {want} floats")
    return (ctypes.c_void_p * len(packed))(*[ptr(q) for q in packed])


def pack_weights(mats, transposed=False, want_plain=False):
    """Fragment-order copies of same-shape weight matrices for the panel GEMMs (gts_pack_weights_f32; layout in
    include/gts_hip.h): [w [R, C], ...] -> [packed(w), ...], or with transposed=True [packed(w^T), ...] — and, with
    want_plain, also [w^T row-major, ...] from the same launch (what transpose_batch returns).  One launch per 32."""
    import ctypes

    if not mats:
        return ([], []) if want_plain else []
    rows, cols = mats[0].shape
    mats = [m.contiguous() for m in mats]
    dev = _chk(*mats)
    for m in mats:
        _same(tuple(m.shape), (rows, cols), "matrices of one pack batch")
    lib = _lib.load()
    n, k = (cols, rows) if transposed else (rows, cols)
    each = lib.gts_packed_weight_floats(n, k)
    out = torch.empty((len(mats), each), dtype=torch.float32, device=dev)
    plain = torch.empty((len(mats), cols, rows), dtype=torch.float32, device=dev) if (want_plain and transposed) else None
    arr = ctypes.c_void_p * len(mats)
    check(lib.gts_pack_weights_f32(arr(*[ptr(m) for m in mats]), arr(*[ptr(out[q]) for q in range(len(mats))]),
                                   arr(*[ptr(plain[q]) for q in range(len(mats))]) if plain is not None else None,
                                   len(mats), rows, cols, 1 if transposed else 0, current_stream()), "gts_pack_weights_f32")
    packed = list(out.unbind(0))
    return (list(plain.unbind(0)), packed) if plain is not None else packed


def linear_fwd(a0, w0, a1=None, w1=None, bias=None, relu=False, relu_bits=None, packed=None):
    """out = act(a0 @ w0^T [+ a1 @ w1^T] + bias).  a [M,K], w [N,K] (torch Linear layout).
    relu_bits (from relu_bits_empty): also filled with out > 0, one bit per element.
    packed: (pack_weights copy of w0, of w1) or None entries — read by the panel kernels instead of w (same values)."""
    _same(_mat(a0, "a0").shape[1], _mat(w0, "w0").shape[1], "inner dims of a0 @ w0^T")
    if (a1 is None) != (w1 is None):
        raise _lib.GtsError("a1 and w1 go together")
    if a1 is not None:
        _same(_mat(a1, "a1").shape[1], _mat(w1, "w1").shape[1], "inner dims of a1 @ w1^T")
        _same(a1.shape[0], a0.shape[0], "rows of a0 / a1")
        _same(w1.shape[0], w0.shape[0], "rows of w0 / w1")
    if bias is not None:
        _same(tuple(bias.shape), (w0.shape[0],), "bias vs output columns")
    a0, w0 = _pad4_cols(a0), _pad4_cols(w0)
    if a1 is not None:
        a1, w1 = _pad4_cols(a1), _pad4_cols(w1)
    dev = _chk(a0, w0, a1, w1, bias)
    m, n = a0.shape[0], w0.shape[0]
    _chk_bits(relu_bits, m, n, "relu_bits")
    out = torch.empty((m, n), dtype=torch.float32, device=dev)
    k0, k1 = a0.shape[1], a1.shape[1] if a1 is not None else 0
    bias = bias.contiguous() if bias is not None else None
    table = _packed_array(packed, (w0, w1))

    def launch():
        check(_lib.load().gts_linear_fwd_f32(ptr(a0), ptr(w0), ptr(a1), ptr(w1), ptr(bias), ptr(out),
                                             m, n, k0, k1, 1 if relu else 0, ptr(relu_bits), table, current_stream()),
              "gts_linear_fwd_f32")

    _timed("fwd", 2.0 * m * n * (k0 + k1), launch)
    return out


def linear_fwd_chain(a0, w0, a1, w1, bias, relu, w2, bias2, relu2, relu_bits=None, packed=None):
    """(out, out2) with out = act(a0 @ w0^T [+ a1 @ w1^T] + bias) and out2 = act2(out @ w2^T + bias2): the two
    GEMMs of consecutive layers in ONE launch when the operands are tall and at most 256 wide (the workgroup
    that has produced a row panel of `out` multiplies it on), otherwise two launches — same values either way.
    All widths must be multiples of 4 (no padding here: callers fall back to two linear_fwd calls).
    relu_bits: filled with out > 0 (as in linear_fwd).  packed: (copy of w0, of w1, of w2) from pack_weights, or None."""
    _same(_mat(a0, "a0").shape[1], _mat(w0, "w0").shape[1], "inner dims of a0 @ w0^T")
    if (a1 is None) != (w1 is None):
        raise _lib.GtsError("a1 and w1 go together")
    if a1 is not None:
        _same(_mat(a1, "a1").shape[1], _mat(w1, "w1").shape[1], "inner dims of a1 @ w1^T")
        _same(a1.shape[0], a0.shape[0], "rows of a0 / a1")
        _same(w1.shape[0], w0.shape[0], "rows of w0 / w1")
    m, n, n2 = a0.shape[0], w0.shape[0], w2.shape[0]
    _same(_mat(w2, "w2").shape[1], n, "inner dims of out @ w2^T")
    if bias is not None:
        _same(tuple(bias.shape), (n,), "bias vs output columns")
    if bias2 is not None:
        _same(tuple(bias2.shape), (n2,), "bias2 vs output columns")
    k0, k1 = a0.shape[1], a1.shape[1] if a1 is not None else 0
    if n % 4 or k0 % 4 or k1 % 4:
        raise _lib.GtsError("linear_fwd_chain needs widths that are multiples of 4")
    a0, w0, a1, w1, bias, w2, bias2 = _dense(a0, w0, a1, w1, bias, w2, bias2)
    dev = _chk(a0, w0, a1, w1, bias, w2, bias2)
    _chk_bits(relu_bits, m, n, "relu_bits")
    out = torch.empty((m, n), dtype=torch.float32, device=dev)
    out2 = torch.empty((m, n2), dtype=torch.float32, device=dev)
    table = _packed_array(packed, (w0, w1, w2))
    _timed("fwd", 2.0 * m * n * (k0 + k1) + 2.0 * m * n2 * n, lambda: check(
        _lib.load().gts_linear_fwd_chain_f32(ptr(a0), ptr(w0), ptr(a1), ptr(w1), ptr(bias), ptr(out), ptr(w2),
                                             ptr(bias2), ptr(out2), m, n, k0, k1, 1 if relu else 0, n2,
                                             1 if relu2 else 0, ptr(relu_bits), table, current_stream()),
        "gts_linear_fwd_chain_f32"))
    return out, out2


def linear_bwd_input_chain_t(g0, w0t, g1, w1t, relu_mask, w2t, relu_bits=None, packed=None):
    """(gin, gin2) with gin = (g0 @ w0 [+ g1 @ w1]) (zeroed where relu_mask <= 0) and gin2 = gin @ w2, from
    TRANSPOSED weights (w0t [K,N0], w1t [K,N1], w2t [K2,K]); one launch under the conditions of linear_fwd_chain.
    relu_bits: relu_mask > 0 as bits (from the forward call that made relu_mask) — read instead of its floats."""
    _same(_mat(g0, "g0").shape[1], _mat(w0t, "w0t").shape[1], "inner dims of g0 @ w0t^T")
    if (g1 is None) != (w1t is None):
        raise _lib.GtsError("g1 and w1t go together")
    if g1 is not None:
        _same(_mat(g1, "g1").shape[1], _mat(w1t, "w1t").shape[1], "inner dims of g1 @ w1t^T")
        _same(g1.shape[0], g0.shape[0], "rows of g0 / g1")
        _same(w1t.shape[0], w0t.shape[0], "rows of w0t / w1t")
    m, k, k2 = g0.shape[0], w0t.shape[0], w2t.shape[0]
    _same(_mat(w2t, "w2t").shape[1], k, "inner dims of gin @ w2t^T")
    if relu_mask is not None:
        _same(tuple(relu_mask.shape), (m, k), "relu_mask vs result")
    n0, n1 = g0.shape[1], g1.shape[1] if g1 is not None else 0
    if k % 4 or n0 % 4 or n1 % 4:
        raise _lib.GtsError("linear_bwd_input_chain_t needs widths that are multiples of 4")
    g0, w0t, g1, w1t, relu_mask, w2t = _dense(g0, w0t, g1, w1t, relu_mask, w2t)
    dev = _chk(g0, w0t, g1, w1t, relu_mask, w2t)
    _chk_bits(relu_bits if relu_mask is not None else None, m, k, "relu_bits")
    relu_bits = relu_bits if relu_mask is not None else None
    gin = torch.empty((m, k), dtype=torch.float32, device=dev)
    gin2 = torch.empty((m, k2), dtype=torch.float32, device=dev)
    table = _packed_array(packed, (w0t, w1t, w2t))
    _timed("igrad", 2.0 * m * k * (n0 + n1) + 2.0 * m * k2 * k, lambda: check(
        _lib.load().gts_linear_bwd_input_chain_t_f32(ptr(g0), ptr(w0t), ptr(g1), ptr(w1t), ptr(relu_mask),
                                                     ptr(relu_bits), ptr(gin), ptr(w2t), ptr(gin2), m, k, n0, n1, k2,
                                                     table, current_stream()),
        "gts_linear_bwd_input_chain_t_f32"))
    return gin, gin2


def linear_bwd_input(g0, w0, g1=None, w1=None, relu_mask=None):
    """g0 @ w0 [+ g1 @ w1].  g [M,N], w [N,K] -> [M,K]; zeroed where relu_mask [M,K] <= 0."""
    _same(_mat(g0, "g0").shape[1], _mat(w0, "w0").shape[0], "inner dims of g0 @ w0")
    if (g1 is None) != (w1 is None):
        raise _lib.GtsError("g1 and w1 go together")
    if g1 is not None:
        _same(_mat(g1, "g1").shape[1], _mat(w1, "w1").shape[0], "inner dims of g1 @ w1")
        _same(g1.shape[0], g0.shape[0], "rows of g0 / g1")
        _same(w1.shape[1], w0.shape[1], "columns of w0 / w1")
    if relu_mask is not None:
        _same(tuple(relu_mask.shape), (g0.shape[0], w0.shape[1]), "relu_mask vs result")
    k = w0.shape[1]
    kp = k + (-k) % 4
    g0, w0 = _pad4_cols(g0), _pad4_cols(_pad4_rows(w0))
    if g1 is not None:
        g1, w1 = _pad4_cols(g1), _pad4_cols(_pad4_rows(w1))
    if relu_mask is not None:
        relu_mask = _pad4_cols(relu_mask)
    dev = _chk(g0, w0, g1, w1, relu_mask)
    m = g0.shape[0]
    gin = torch.empty((m, kp), dtype=torch.float32, device=dev)
    n0, n1 = g0.shape[1], g1.shape[1] if g1 is not None else 0
    _timed("igrad", 2.0 * m * kp * (n0 + n1), lambda: check(
        _lib.load().gts_linear_bwd_input_f32(ptr(g0), ptr(w0), ptr(g1), ptr(w1), ptr(relu_mask), ptr(gin),
                                             m, kp, n0, n1, current_stream()), "gts_linear_bwd_input_f32"))
    return gin if kp == k else gin[:, :k].contiguous()


def transpose_batch(mats):
    """[w [R,C], ...] (one shape) -> [w^T [C,R], ...] as views of ONE new buffer, one launch per 32."""
    import ctypes

    if not mats:
        return []
    rows, cols = mats[0].shape
    mats = [m.contiguous() for m in mats]
    dev = _chk(*mats)
    for m in mats:
        _same(tuple(m.shape), (rows, cols), "matrices of one transpose batch")
    out = torch.empty((len(mats), cols, rows), dtype=torch.float32, device=dev)
    arr = ctypes.c_void_p * len(mats)
    check(_lib.load().gts_transpose_batch_f32(arr(*[ptr(m) for m in mats]), arr(*[ptr(out[q]) for q in range(len(mats))]),
                                              len(mats), rows, cols, current_stream()), "gts_transpose_batch_f32")
    return list(out.unbind(0))


def linear_bwd_input_t(g0, w0t, g1=None, w1t=None, relu_mask=None, relu_bits=None, packed=None):
    """linear_bwd_input from TRANSPOSED weights (w0t [K,N0] = w0.t(), from transpose_batch): the
    GEMM then runs in the forward kernel's form.  Same values, bit for bit.  All widths % 4 == 0.
    relu_bits: relu_mask > 0 as bits (see linear_fwd) — read instead of the floats where the kernel can."""
    _same(_mat(g0, "g0").shape[1], _mat(w0t, "w0t").shape[1], "inner dims of g0 @ w0t^T")
    if (g1 is None) != (w1t is None):
        raise _lib.GtsError("g1 and w1t go together")
    if g1 is not None:
        _same(_mat(g1, "g1").shape[1], _mat(w1t, "w1t").shape[1], "inner dims of g1 @ w1t^T")
        _same(g1.shape[0], g0.shape[0], "rows of g0 / g1")
        _same(w1t.shape[0], w0t.shape[0], "rows of w0t / w1t")
    m, k = g0.shape[0], w0t.shape[0]
    if relu_mask is not None:
        _same(tuple(relu_mask.shape), (m, k), "relu_mask vs result")
    n0, n1 = g0.shape[1], g1.shape[1] if g1 is not None else 0
    if k % 4 or n0 % 4 or n1 % 4:
        raise _lib.GtsError("linear_bwd_input_t needs widths that are multiples of 4 (use linear_bwd_input)")
    g0, w0t, g1, w1t, relu_mask = _dense(g0, w0t, g1, w1t, relu_mask)
    dev = _chk(g0, w0t, g1, w1t, relu_mask)
    relu_bits = relu_bits if relu_mask is not None else None
    _chk_bits(relu_bits, m, k, "relu_bits")
    gin = torch.empty((m, k), dtype=torch.float32, device=dev)
    table = _packed_array(packed, (w0t, w1t))
    _timed("igrad", 2.0 * m * k * (n0 + n1), lambda: check(
        _lib.load().gts_linear_bwd_input_t_f32(ptr(g0), ptr(w0t), ptr(g1), ptr(w1t), ptr(relu_mask), ptr(relu_bits),
                                               ptr(gin), m, k, n0, n1, table, current_stream()), "gts_linear_bwd_input_t_f32"))
    return gin


def linear_bwd_input_t_act(g0, w0t, act_out, activation, g1=None, w1t=None, want_bias_grad=True, packed=None):
    """(gin, g_bias): linear_bwd_input_t carried through the activation of the layer BELOW, whose output `act_out` [M,K] is
    this layer's input (activation 1 = ELU, 2 = ReLU): gin = (g0 @ w0t^T (+ g1 @ w1t^T)) * act'(act_out), g_bias = gin.sum(0)
    — at the tall 256-column-block shapes inside the GEMM's epilogue, otherwise as the pass of its own; the same gin
    either way."""
    _same(_mat(g0, "g0").shape[1], _mat(w0t, "w0t").shape[1], "inner dims of g0 @ w0t^T")
    if (g1 is None) != (w1t is None):
        raise _lib.GtsError("g1 and w1t go together")
    if g1 is not None:
        _same(_mat(g1, "g1").shape[1], _mat(w1t, "w1t").shape[1], "inner dims of g1 @ w1t^T")
        _same(g1.shape[0], g0.shape[0], "rows of g0 / g1")
        _same(w1t.shape[0], w0t.shape[0], "rows of w0t / w1t")
    m, k = g0.shape[0], w0t.shape[0]
    _same(tuple(_mat(act_out, "act_out").shape), (m, k), "act_out vs result")
    n0, n1 = g0.shape[1], g1.shape[1] if g1 is not None else 0
    if k % 4 or n0 % 4 or n1 % 4:
        raise _lib.GtsError("linear_bwd_input_t_act needs widths that are multiples of 4")
    g0, w0t, g1, w1t, act_out = _dense(g0, w0t, g1, w1t, act_out)
    dev = _chk(g0, w0t, g1, w1t, act_out)
    lib = _lib.load()
    gin = torch.empty((m, k), dtype=torch.float32, device=dev)
    g_bias = torch.empty(k, dtype=torch.float32, device=dev) if want_bias_grad else None
    nbytes = lib.gts_linear_bwd_input_t_act_workspace(m, k) if want_bias_grad else 0
    ws = _workspace(dev, nbytes) if want_bias_grad else None
    table = _packed_array(packed, (w0t, w1t))
    _timed("igrad", 2.0 * m * k * (n0 + n1), lambda: check(
        lib.gts_linear_bwd_input_t_act_f32(ptr(g0), ptr(w0t), ptr(g1), ptr(w1t), ptr(act_out), activation, ptr(gin),
                                           ptr(g_bias), ptr(ws), nbytes, m, k, n0, n1, table, current_stream()),
        "gts_linear_bwd_input_t_act_f32"))
    return gin, g_bias


MAX_WGRAD_PROBLEMS = 32   # kMaxProblems of csrc/gts_gemm.hip


def linear_bwd_weight_multi(problems):
    """[(g [M,N], a [M,K], want_bias_grad), ...] (problems of ONE shape) ->
    [(g^T @ a [N,K], column sums of g [N] or None), ...]; up to 32 problems per split-reduction
    launch (more are handled in groups)."""
    import ctypes

    if len(problems) > MAX_WGRAD_PROBLEMS:
        return (linear_bwd_weight_multi(problems[:MAX_WGRAD_PROBLEMS])
                + linear_bwd_weight_multi(problems[MAX_WGRAD_PROBLEMS:]))

    n, k = problems[0][0].shape[1], problems[0][1].shape[1]
    gs = [_pad4_cols(g) for g, _, _ in problems]
    acts = [_pad4_cols(a) for _, a, _ in problems]
    dev = _chk(*gs, *acts)
    m, n_p, k_p = gs[0].shape[0], gs[0].shape[1], acts[0].shape[1]
    for g, a in zip(gs, acts):
        if g.shape != (m, n_p) or a.shape != (m, k_p):
            raise _lib.GtsError("batched weight gradients need identical shapes")
    q = len(problems)
    gws = [torch.empty((n_p, k_p), dtype=torch.float32, device=dev) for _ in range(q)]
    gbs = [torch.empty(n_p, dtype=torch.float32, device=dev) if want else None for _, _, want in problems]
    if m == 0:
        for t in gws + [b for b in gbs if b is not None]:
            t.zero_()
    else:
        lib = _lib.load()
        nbytes = lib.gts_linear_bwd_weight_workspace(m, n_p, k_p, q)
        ws = _workspace(dev, nbytes)
        arr = ctypes.c_void_p * q
        _timed("wgrad", 2.0 * m * n_p * k_p * q, lambda: check(
            lib.gts_linear_bwd_weight_f32(arr(*[ptr(t) for t in gs]), arr(*[ptr(t) for t in acts]),
                                          arr(*[ptr(t) for t in gws]), arr(*[ptr(t) for t in gbs]), q,
                                          ptr(ws), ws.numel() * 4, m, n_p, k_p, current_stream()),
            "gts_linear_bwd_weight_f32"))
    out = []
    for gw, gb in zip(gws, gbs):
        if (n_p, k_p) != (n, k):
            gw = gw[:n, :k].contiguous()
            gb = gb[:n].contiguous() if gb is not None else None
        out.append((gw, gb))
    return out


def linear_bwd_weight(g, a, want_bias_grad=False):
    """(g^T @ a [N,K], column sums of g [N] or None).  g [M,N], a [M,K]."""
    return linear_bwd_weight_multi([(g, a, want_bias_grad)])[0]


def relu_bwd(g, out):
    """g * (out > 0) where `out` is the ReLU output (new tensor; g is left untouched)."""
    return g * (out > 0)


class _Linear(torch.autograd.Function):
    """y = x @ w^T (+ bias): nn.Linear on the MFMA kernels (used by the non-fused layers)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return linear_fwd(x, w, bias=bias)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = linear_bwd_input(gy, w) if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = linear_bwd_weight(gy, x, want_bias_grad=ctx.has_bias)
        return gx, gw, gb if ctx.has_bias else None


def linear(x, w, bias=None):
    """Drop-in for F.linear on 2-D inputs."""
    return _Linear.apply(x, w, bias)
