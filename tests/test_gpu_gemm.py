"""K11 fp32 MFMA GEMMs against torch (fp64 reference on the CPU).  The MFMA result is an
exact-fp32 fma chain in a permuted k order, so it agrees with a fp64 reference to fp32
rounding: |err| <= 2e-6 * sum_k |a_k b_k| is asserted (observed ~1e-7)."""
import pytest
import torch

from gts import dense

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _lib(hip_lib):
    assert torch.cuda.is_available()
    return hip_lib


def _rand(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _check(got, want64, bound64):
    err = (got.cpu().double() - want64).abs()
    assert torch.all(err <= 2e-6 * bound64 + 1e-30), f"max err {err.max():.3e}, bound {bound64.max():.3e}"


SHAPES = [(1, 4, 4), (37, 4, 8), (128, 256, 4), (129, 4, 256), (300, 256, 256), (1000, 20, 256),
          (515, 256, 260), (64, 8, 4), (2049, 128, 64), (250, 300, 132), (60000, 4, 4), (59999, 8, 8),
          (60000, 256, 4), (59999, 260, 8), (5, 64, 4), (4097, 1024, 8), (60000, 4, 256),
          # the reference's own first layer (in_feats 20, utils/hyperparam_helpers.py:22) at its batch size: fc_pool 20 -> 20 (one lane per
          # row), g @ W_neigh 256 -> 20 and a 256 -> 20 forward (rows streamed, two per wave), 20 -> 256
          (34992, 20, 20), (34992, 20, 256), (34992, 256, 20), (33, 20, 20), (7, 256, 20)]


@pytest.mark.parametrize("m,k,n", SHAPES)
@pytest.mark.parametrize("dual,relu,bias", [(False, False, False), (True, True, True), (False, True, True)])
def test_linear_forward(m, k, n, dual, relu, bias):
    a0, w0 = _rand(m, k, seed=1), _rand(n, k, seed=2)
    a1, w1 = (_rand(m, k + 4, seed=3), _rand(n, k + 4, seed=4)) if dual else (None, None)
    b = _rand(n, seed=5) if bias else None
    want = a0.double() @ w0.double().t()
    bound = a0.double().abs() @ w0.double().abs().t()
    if dual:
        want += a1.double() @ w1.double().t()
        bound += a1.double().abs() @ w1.double().abs().t()
    if bias:
        want += b.double()
        bound += b.double().abs()
    if relu:
        want = want.clamp(min=0)
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    got = dense.linear_fwd(dev(a0), dev(w0), dev(a1), dev(w1), bias=dev(b), relu=relu)
    assert got.shape == (m, n)
    _check(got, want, bound)


@pytest.mark.parametrize("m,k,n,dual", [(49999, 132, 260, True), (60000, 256, 256, False)])
def test_linear_forward_one_round_tile(m, k, n, dual):
    """Shapes that select the double-buffered 256x256 tile (>= 192 of them): ragged rows and
    columns, both reduction segments, bias + ReLU."""
    a0, w0 = _rand(m, k, seed=11), _rand(n, k, seed=12)
    a1, w1 = (_rand(m, k + 4, seed=13), _rand(n, k + 4, seed=14)) if dual else (None, None)
    b = _rand(n, seed=15)
    want = a0.double() @ w0.double().t() + b.double()
    bound = a0.double().abs() @ w0.double().abs().t() + b.double().abs()
    if dual:
        want += a1.double() @ w1.double().t()
        bound += a1.double().abs() @ w1.double().abs().t()
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    got = dense.linear_fwd(dev(a0), dev(w0), dev(a1), dev(w1), bias=dev(b), relu=True)
    _check(got, want.clamp(min=0), bound)


@pytest.mark.parametrize("m,f,dual", [(60000, 256, True), (49999, 1024, False)])
def test_forward_tile_variants_are_bitwise_identical_at_full_size(hip_lib, m, f, dual):
    """Every 32x32x2 tile variant walks the reduction in the same order, so the C2- / C3-sized forward
    GEMM of the one-round 256x256 tile equals the 64x256 / 128x256 tiles the small-graph parity tests
    exercise, bit for bit.  The 240-row panels (variant 10: what -1 selects at these sizes) run
    on the 16x16x4 MFMA, which adds four products per step instead of two: equal to fp32 rounding (checked
    against fp64 in the tests below), deterministic, and what the automatic choice returns."""
    a0, w0, b = _rand(m, f, seed=21).to(DEV), _rand(f, f, seed=23).to(DEV), _rand(f, seed=25).to(DEV)
    a1, w1 = (_rand(m, f, seed=22).to(DEV), _rand(f, f, seed=24).to(DEV)) if dual else (None, None)
    outs = {}
    try:
        for variant in (8, 3, 1, 10, -1, 10):
            assert hip_lib.gts_set_option(1, variant) == 0
            outs.setdefault(variant, []).append(dense.linear_fwd(a0, w0, a1, w1, bias=b, relu=True))
    finally:
        hip_lib.gts_set_option(1, -1)
    assert torch.equal(outs[8][0], outs[3][0]) and torch.equal(outs[8][0], outs[1][0])
    assert torch.equal(outs[10][0], outs[10][1]) and torch.equal(outs[10][0], outs[-1][0])
    scale = float(outs[8][0].abs().max())
    assert float((outs[10][0] - outs[8][0]).abs().max()) < 1e-5 * scale


ROWS240 = [(240, 256, 256, 0), (239, 64, 256, 64), (241, 32, 260, 0), (1000, 132, 132, 260), (60000, 256, 256, 256),
           (49999, 260, 1024, 0), (481, 4, 8, 4), (5, 36, 4, 0), (120000, 256, 256, 0)]


@pytest.mark.parametrize("variant", [10])
@pytest.mark.parametrize("m,k0,n,k1", ROWS240)
@pytest.mark.parametrize("relu,bias,mask", [(True, True, False), (False, False, True)])
def test_rows240_panels_against_fp64(hip_lib, m, k0, n, k1, relu, bias, mask, variant):
    """The 240 x 256 panel kernel on v_mfma_f32_16x16x4_f32 (10: operands loaded straight into the MFMA fragments
    through buffer loads; the forms measured and rejected are not in the library) forced on every shape class:
    ragged panels / columns / reduction tiles, both segments, bias + ReLU, and the ReLU-mask epilogue
    of the transposed-weight input gradient."""
    a0, w0 = _rand(m, k0, seed=1), _rand(n, k0, seed=2)
    a1, w1 = (_rand(m, k1, seed=3), _rand(n, k1, seed=4)) if k1 else (None, None)
    b = _rand(n, seed=5) if bias else None
    rm = _rand(m, n, seed=6) if mask else None
    want = a0.double() @ w0.double().t()
    bound = a0.double().abs() @ w0.double().abs().t()
    if k1:
        want += a1.double() @ w1.double().t()
        bound += a1.double().abs() @ w1.double().abs().t()
    if bias:
        want += b.double()
        bound += b.double().abs()
    if relu:
        want = want.clamp(min=0)
    if mask:
        want = want * (rm > 0)
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    try:
        assert hip_lib.gts_set_option(1, variant) == 0
        if mask:     # the mask lives on the input-gradient entry point: gin = g @ W from W^T (here w = W^T)
            got = dense.linear_bwd_input_t(dev(a0), dev(w0), dev(a1), dev(w1), relu_mask=dev(rm))
        else:
            got = dense.linear_fwd(dev(a0), dev(w0), dev(a1), dev(w1), bias=dev(b), relu=relu)
        again = dense.linear_fwd(dev(a0), dev(w0), dev(a1), dev(w1), bias=dev(b), relu=relu) if not mask else got
    finally:
        hip_lib.gts_set_option(1, -1)
    assert got.shape == (m, n) and torch.equal(got, again)
    _check(got, want, bound)


def test_rows240_narrow_output_through_the_c_abi(hip_lib):
    """n not a multiple of 4 (only reachable through the C ABI; the Python layer pads): scalar stores."""
    from gts._lib import current_stream, ptr

    m, k, n = 500, 64, 6
    a, w, b = _rand(m, k, seed=1).to(DEV), _rand(n, k, seed=2).to(DEV), _rand(n, seed=3).to(DEV)
    out = torch.full((m, n), float("nan"), device=DEV)
    try:
        assert hip_lib.gts_set_option(1, 10) == 0
        assert hip_lib.gts_linear_fwd_f32(ptr(a), ptr(w), None, None, ptr(b), ptr(out), m, n, k, 0, 1, None, None, current_stream()) == 0
    finally:
        hip_lib.gts_set_option(1, -1)
    want = (a.cpu().double() @ w.cpu().double().t() + b.cpu().double()).clamp(min=0)
    assert torch.allclose(out.cpu().double(), want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("m,k,n", SHAPES)
@pytest.mark.parametrize("dual", [False, True])
def test_linear_input_gradient(m, k, n, dual):
    g0, w0 = _rand(m, n, seed=1), _rand(n, k, seed=2)
    g1, w1 = (_rand(m, n + 8, seed=3), _rand(n + 8, k, seed=4)) if dual else (None, None)
    want = g0.double() @ w0.double()
    bound = g0.double().abs() @ w0.double().abs()
    if dual:
        want += g1.double() @ w1.double()
        bound += g1.double().abs() @ w1.double().abs()
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    got = dense.linear_bwd_input(dev(g0), dev(w0), dev(g1), dev(w1))
    assert got.shape == (m, k)
    _check(got, want, bound)


@pytest.mark.parametrize("m,k,n", SHAPES + [(60000, 256, 256), (60000, 4, 256), (60000, 256, 4)])
def test_linear_weight_and_bias_gradient(m, k, n):
    g, a = _rand(m, n, seed=1), _rand(m, k, seed=2)
    gw, gb = dense.linear_bwd_weight(g.to(DEV), a.to(DEV), want_bias_grad=True)
    assert gw.shape == (n, k) and gb.shape == (n,)
    _check(gw, g.double().t() @ a.double(), g.double().abs().t() @ a.double().abs())
    _check(gb, g.double().sum(0), g.double().abs().sum(0))
    gw2, none = dense.linear_bwd_weight(g.to(DEV), a.to(DEV))
    assert none is None and torch.equal(gw2, gw)          # bitwise reproducible (fixed-order slabs)


@pytest.mark.parametrize("m,n,k,count", [(3000, 256, 256, 19), (777, 128, 132, 5), (500, 64, 64, 35)])
def test_many_weight_gradients_in_one_launch(m, n, k, count):
    """A whole layer stack's weight gradients (19 problems of one shape at C2) in one
    split-reduction launch; more than 32 problems are handled in groups.  Distinct operands per
    problem, bias gradients for every other one, each checked against fp64."""
    gs = [_rand(m, n, seed=100 + q) for q in range(count)]
    acts = [_rand(m, k, seed=200 + q) for q in range(count)]
    out = dense.linear_bwd_weight_multi([(g.to(DEV), a.to(DEV), q % 2 == 0) for q, (g, a) in enumerate(zip(gs, acts))])
    assert len(out) == count
    for q, ((gw, gb), g, a) in enumerate(zip(out, gs, acts)):
        _check(gw, g.double().t() @ a.double(), g.double().abs().t() @ a.double().abs())
        assert (gb is not None) == (q % 2 == 0)
        if gb is not None:
            _check(gb, g.double().sum(0), g.double().abs().sum(0))
    again = dense.linear_bwd_weight_multi([(g.to(DEV), a.to(DEV), False) for g, a in zip(gs, acts)])
    assert all(torch.equal(x[0], y[0]) for x, y in zip(out, again))      # fixed-order slabs: reproducible


def test_linear_autograd_matches_torch():
    x = _rand(333, 20, seed=1).to(DEV).requires_grad_(True)
    w = _rand(64, 20, seed=2).to(DEV).requires_grad_(True)
    b = _rand(64, seed=3).to(DEV).requires_grad_(True)
    gy = _rand(333, 64, seed=4).to(DEV)
    dense.linear(x, w, b).backward(gy)
    got = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    torch.nn.functional.linear(x, w, b).backward(gy)
    for a, t in zip(got, (x, w, b)):
        assert torch.allclose(a, t.grad, rtol=1e-4, atol=1e-4)


def test_unaligned_width_is_padded_on_the_host():
    a, w = _rand(50, 5, seed=1), _rand(7, 5, seed=2)
    got = dense.linear_fwd(a.to(DEV), w.to(DEV))
    assert got.shape == (50, 7)
    assert torch.allclose(got.cpu(), a @ w.t(), rtol=1e-5, atol=1e-5)
    gx = dense.linear_bwd_input(_rand(50, 7, seed=3).to(DEV), w.to(DEV))
    assert gx.shape == (50, 5)
    assert torch.allclose(gx.cpu(), _rand(50, 7, seed=3) @ w, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("rows,cols,count", [(256, 256, 19), (4, 256, 2), (37, 50, 35), (1, 1, 1)])
def test_transpose_batch(rows, cols, count):
    mats = [_rand(rows, cols, seed=q).to(DEV) for q in range(count)]
    outs = dense.transpose_batch(mats)
    assert len(outs) == count
    for m, t in zip(mats, outs):
        assert t.shape == (cols, rows) and t.is_contiguous() and torch.equal(t, m.t())


@pytest.mark.parametrize("m,k,n0,n1,mask", [(60000, 256, 256, 256, True), (60000, 256, 256, 0, False),
                                            (49999, 132, 260, 64, True), (300, 256, 4, 256, False),
                                            (120000, 256, 256, 256, True)])
def test_input_gradient_from_transposed_weights(m, k, n0, n1, mask):
    """gts_linear_bwd_input_t_f32 (forward-form kernel on W^T) against gts_linear_bwd_input_f32 (bitwise on the
    same MFMA shape) and fp64."""
    g0, w0 = _rand(m, n0, seed=1).to(DEV), _rand(n0, k, seed=2).to(DEV)
    g1, w1 = (_rand(m, n1, seed=3).to(DEV), _rand(n1, k, seed=4).to(DEV)) if n1 else (None, None)
    rm = _rand(m, k, seed=5).to(DEV) if mask else None
    want = dense.linear_bwd_input(g0, w0, g1, w1, relu_mask=rm)
    w0t = dense.transpose_batch([w0])[0]
    w1t = dense.transpose_batch([w1])[0] if n1 else None
    lib = dense._lib.load()
    try:          # same 32x32x2 reduction order as the strided kernel when the 256x256 tile is forced
        assert lib.gts_set_option(1, 8) == 0
        assert torch.equal(dense.linear_bwd_input_t(g0, w0t, g1, w1t, relu_mask=rm), want)
    finally:
        lib.gts_set_option(1, -1)
    got = dense.linear_bwd_input_t(g0, w0t, g1, w1t, relu_mask=rm)       # automatic tile (240-row panels when large)
    ref = g0.cpu().double() @ w0.cpu().double()
    bound = g0.cpu().double().abs() @ w0.cpu().double().abs()
    if n1:
        ref += g1.cpu().double() @ w1.cpu().double()
        bound += g1.cpu().double().abs() @ w1.cpu().double().abs()
    if mask:
        ref = ref * (rm.cpu() > 0)
    _check(got, ref, bound)


@pytest.mark.parametrize("m,k,n,count", [(60000, 256, 256, 19), (3000, 256, 256, 3), (777, 132, 128, 5), (2049, 64, 128, 2),
                                         (515, 260, 256, 1), (37, 16, 12, 1), (4641, 256, 260, 4), (6, 32, 32, 1)])
@pytest.mark.parametrize("variant", [4, 6])
def test_direct_fragment_weight_gradient_against_fp64(hip_lib, m, k, n, count, variant):
    """The 256 x 256 weight-gradient tiles the library carries (GTS_OPT_WGRAD_TILE = 4: double-buffered through registers
    and LDS; 6: wgrad_stream_kernel — LDS-DMA tile copies, a main loop of MFMAs and LDS reads only, the automatic
    choice) forced on ragged node counts / widths / split boundaries, several problems per launch, bias sums."""
    gs = [_rand(m, n, seed=500 + q) for q in range(count)]
    acts = [_rand(m, k, seed=600 + q) for q in range(count)]
    dev = [(g.to(DEV), a.to(DEV), q % 2 == 0) for q, (g, a) in enumerate(zip(gs, acts))]
    try:
        assert hip_lib.gts_set_option(2, variant) == 0
        out = dense.linear_bwd_weight_multi(dev)
        again = dense.linear_bwd_weight_multi(dev)
    finally:
        hip_lib.gts_set_option(2, -1)
    for q, ((gw, gb), g, a) in enumerate(zip(out, gs, acts)):
        _check(gw, g.double().t() @ a.double(), g.double().abs().t() @ a.double().abs())
        assert (gb is not None) == (q % 2 == 0)
        if gb is not None:
            _check(gb, g.double().sum(0), g.double().abs().sum(0))
        assert torch.equal(gw, again[q][0])


@pytest.mark.parametrize("m,k,n,count", [(60000, 256, 256, 19), (120000, 256, 256, 19), (3000, 256, 256, 3), (777, 132, 128, 5),
                                         (2049, 64, 128, 2), (515, 260, 256, 1), (37, 16, 12, 1), (4641, 256, 260, 4),
                                         (6, 32, 32, 1), (33, 256, 256, 2), (64, 256, 256, 1), (96, 256, 256, 32),
                                         (15000, 1024, 1024, 2)])
def test_streaming_weight_gradient_equals_the_lds_tile_bit_for_bit(hip_lib, m, k, n, count):
    """GTS_OPT_WGRAD_TILE = 6 (wgrad_stream_kernel: fragment reads at immediate offsets, scalar-built DMA descriptors, bias
    sums in one wave per SIMD and only where asked for) walks the same tiles, slabs, MFMA steps and column-sum order as
    variant 4 (the double-buffered LDS tile): weight and bias gradients agree in every bit — one, two, three and many
    reduction tiles per split (both image parities and the odd tail), ragged node counts and widths, 32 problems."""
    gs = [_rand(m, n, seed=700 + q) for q in range(min(count, 4))]
    acts = [_rand(m, k, seed=800 + q) for q in range(min(count, 4))]
    dev_g, dev_a = [g.to(DEV) for g in gs], [a.to(DEV) for a in acts]
    dev = [(dev_g[q % len(gs)], dev_a[(q // 2) % len(acts)], q % 3 != 2) for q in range(count)]
    got = {}
    try:
        for variant in (4, 6):
            assert hip_lib.gts_set_option(2, variant) == 0
            got[variant] = dense.linear_bwd_weight_multi(dev)
    finally:
        hip_lib.gts_set_option(2, -1)
    for q, ((gw4, gb4), (gw6, gb6)) in enumerate(zip(got[4], got[6])):
        assert torch.equal(gw4, gw6), q
        assert (gb4 is None) == (gb6 is None) == (q % 3 == 2)
        if gb4 is not None:
            assert torch.equal(gb4, gb6), q
    g, a = gs[0], acts[0]
    _check(got[6][0][0], g.double().t() @ a.double(), g.double().abs().t() @ a.double().abs())
    _check(got[6][0][1], g.double().sum(0), g.double().abs().sum(0))


@pytest.mark.parametrize("m,count", [(60000, 19), (15013, 7)])
def test_streaming_weight_gradient_is_the_same_launch_after_launch(m, count):
    """wgrad_stream_kernel orders its LDS-DMA tile copies, its hand-placed LDS reads and the image hand-over with explicit
    wait counts and one barrier per tile: 150 launches on the same operands (full size; a ragged node count with an odd
    number of reduction tiles per split) must give the bits of the first — a missed wait would show as a stray difference."""
    gs = [_rand(m, 256, seed=900 + q).to(DEV) for q in range(3)]
    acts = [_rand(m, 256, seed=950 + q).to(DEV) for q in range(3)]
    problems = [(gs[q % 3], acts[(q + 1) % 3], q % 2 == 0) for q in range(count)]
    first = dense.linear_bwd_weight_multi(problems)
    first = [(gw.clone(), None if gb is None else gb.clone()) for gw, gb in first]
    for _ in range(150):
        again = dense.linear_bwd_weight_multi(problems)
        for (gw0, gb0), (gw1, gb1) in zip(first, again):
            assert torch.equal(gw0, gw1)
            assert gb0 is None or torch.equal(gb0, gb1)


@pytest.mark.parametrize("m,k0,k1,n,n2", [(60000, 256, 256, 256, 256), (60000, 4, 4, 256, 256), (49999, 132, 0, 256, 64),
                                          (60000, 256, 0, 132, 4), (1000, 64, 64, 256, 256), (120000, 256, 256, 256, 256)])
def test_chained_layer_gemms_equal_two_calls_bit_for_bit(m, k0, k1, n, n2):
    """gts_linear_fwd_chain_f32 / gts_linear_bwd_input_chain_t_f32: the second GEMM computed by the workgroup that
    has just stored those rows (one launch at the tall shapes, two at m = 1000) gives the bits of two separate calls."""
    a0, w0 = _rand(m, k0, seed=1).to(DEV), _rand(n, k0, seed=2).to(DEV)
    a1, w1 = (_rand(m, k1, seed=3).to(DEV), _rand(n, k1, seed=4).to(DEV)) if k1 else (None, None)
    b, w2, b2 = _rand(n, seed=5).to(DEV), _rand(n2, n, seed=6).to(DEV), _rand(n2, seed=7).to(DEV)
    out, out2 = dense.linear_fwd_chain(a0, w0, a1, w1, b, True, w2, b2, True)
    want = dense.linear_fwd(a0, w0, a1, w1, bias=b, relu=True)
    want2 = dense.linear_fwd(want, w2, bias=b2, relu=True)
    assert torch.equal(out, want) and torch.equal(out2, want2)
    ref = a0.cpu().double() @ w0.cpu().double().t() + b.cpu().double()
    if k1:
        ref += a1.cpu().double() @ w1.cpu().double().t()
    ref2 = (ref.clamp(min=0) @ w2.cpu().double().t() + b2.cpu().double()).clamp(min=0)
    assert torch.allclose(out2.cpu().double(), ref2, rtol=1e-4, atol=1e-4 * float(ref2.abs().max()))
    # input-gradient form on transposed weights, with the ReLU mask between the two products
    mask = _rand(m, n, seed=8).to(DEV)
    gin, gin2 = dense.linear_bwd_input_chain_t(a0, w0, a1, w1, mask, w2)
    want = dense.linear_bwd_input_t(a0, w0, a1, w1, relu_mask=mask)
    assert torch.equal(gin, want) and torch.equal(gin2, dense.linear_bwd_input_t(want, w2))


@pytest.mark.parametrize("rows,cols", [(256, 256), (256, 4), (256, 20), (4, 256), (132, 260), (17, 33), (1024, 1024)])
@pytest.mark.parametrize("transposed", [False, True])
def test_weights_in_fragment_order_are_the_layout_the_header_states(rows, cols, transposed):
    """gts_pack_weights_f32: packed[((T * G + g) * 64 + lane) * 4 + e] = B[16 T + (lane & 15)][16 g + 4 (lane >> 4) + e] with
    B = w (or w^T), zeros past the edges; with transposed=True the same launch also writes w^T row-major."""
    import numpy as np

    ws = [_rand(rows, cols, seed=10 + q).to(DEV) for q in range(3)]
    if transposed:
        plain, packed = dense.pack_weights(ws, transposed=True, want_plain=True)
        assert all(torch.equal(p, w.t().contiguous()) for p, w in zip(plain, ws))
    else:
        packed = dense.pack_weights(ws)
    for w, got in zip(ws, packed):
        b = (w.t() if transposed else w).cpu().numpy()
        n, k = b.shape
        tiles, groups = (n + 15) // 16, (k + 15) // 16
        pad = np.zeros((16 * tiles, 16 * groups), np.float32)
        pad[:n, :k] = b
        # [T, i16, g, q, e] -> [T, g, q, i16, e]: lane = i16 + 16 q
        want = pad.reshape(tiles, 16, groups, 4, 4).transpose(0, 2, 3, 1, 4).reshape(-1)
        assert got.numel() == want.size and np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("m,k0,k1,n,n2", [(34992, 256, 256, 256, 256), (34992, 20, 20, 256, 256), (60000, 256, 256, 256, 256),
                                          (60000, 4, 4, 256, 256), (35000, 256, 0, 256, 256), (30001, 132, 256, 256, 256),
                                          (1000, 64, 64, 256, 256), (49999, 132, 0, 256, 64)])
def test_gemms_reading_weights_in_fragment_order_give_the_same_bits(m, k0, k1, n, n2):
    """Forward pair, chained forward (mask bits written), transposed input gradient and its chained form with the
    `packed` copies of their weights (csrc/gts_gemm.hip: a fragment load = 1 KiB of consecutive bytes) against the same
    calls on the weights as torch stores them: bit for bit — at the reference's batch shape (35 000 rows: 144-row
    panels), at C2's (60 000: 240-row panels), with reduction tails (20, 132) and at a size where other kernels run and
    the copies are ignored — and against fp64."""
    a0, w0 = _rand(m, k0, seed=1).to(DEV), _rand(n, k0, seed=2).to(DEV)
    a1, w1 = (_rand(m, k1, seed=3).to(DEV), _rand(n, k1, seed=4).to(DEV)) if k1 else (None, None)
    b, w2, b2 = _rand(n, seed=5).to(DEV), _rand(n2, n, seed=6).to(DEV), _rand(n2, seed=7).to(DEV)
    mask = _rand(m, n, seed=8).to(DEV)
    p0, p2 = dense.pack_weights([w0])[0], dense.pack_weights([w2])[0]
    p1 = dense.pack_weights([w1])[0] if k1 else None
    bits_a, bits_b = dense.relu_bits_empty(m, n, a0.device).zero_(), dense.relu_bits_empty(m, n, a0.device).zero_()
    out, out2 = dense.linear_fwd_chain(a0, w0, a1, w1, b, True, w2, b2, True, relu_bits=bits_a)
    got, got2 = dense.linear_fwd_chain(a0, w0, a1, w1, b, True, w2, b2, True, relu_bits=bits_b, packed=(p0, p1, p2))
    assert torch.equal(got, out) and torch.equal(got2, out2) and torch.equal(bits_a, bits_b)
    assert torch.equal(dense.linear_fwd(a0, w0, a1, w1, bias=b, relu=True, packed=(p0, p1)), out)
    assert torch.equal(dense.linear_fwd(a0, w0, a1, w1, bias=b, relu=True, packed=(None, p1)), out)      # one copy only
    # the same operands as TRANSPOSED weights of an input gradient: w0 [n, k0] = (W0)^T with W0 [k0, n] ...
    gin, gin2 = dense.linear_bwd_input_chain_t(a0, w0, a1, w1, mask, w2, relu_bits=bits_a)
    hin, hin2 = dense.linear_bwd_input_chain_t(a0, w0, a1, w1, mask, w2, relu_bits=bits_a, packed=(p0, p1, p2))
    assert torch.equal(hin, gin) and torch.equal(hin2, gin2)
    assert torch.equal(dense.linear_bwd_input_t(a0, w0, a1, w1, relu_mask=mask, packed=(p0, p1)), gin.clone() if False else
                       dense.linear_bwd_input_t(a0, w0, a1, w1, relu_mask=mask))
    ref = a0.cpu().double() @ w0.cpu().double().t() + b.cpu().double()
    bound = a0.cpu().double().abs() @ w0.cpu().double().abs().t() + b.cpu().double().abs()
    if k1:
        ref += a1.cpu().double() @ w1.cpu().double().t()
        bound += a1.cpu().double().abs() @ w1.cpu().double().abs().t()
    _check(got, ref.clamp(min=0), bound)
    with pytest.raises(Exception, match="fragment-order copy"):
        dense.linear_fwd(a0, w0, a1, w1, bias=b, relu=True, packed=(p2[:-4], p1))


@pytest.mark.parametrize("m,k0,k1,n", [(60000, 256, 256, 256), (60000, 4, 4, 256), (49999, 132, 0, 128), (1000, 64, 64, 256),
                                       (46083, 256, 0, 64), (7, 8, 0, 64)])
@pytest.mark.parametrize("variant", [-1, 10, 8])
def test_relu_mask_as_bits_written_by_the_forward_and_read_by_the_input_gradient(hip_lib, m, k0, k1, n, variant):
    """`relu_bits`: the forward GEMM records out > 0 as one bit per element (panel kernels: in their epilogue;
    the other tiles: a pass of their own), and the transposed input gradient masked by those bits equals the one
    masked by the floats, bit for bit, plain and chained."""
    a0, w0 = _rand(m, k0, seed=1).to(DEV), _rand(n, k0, seed=2).to(DEV)
    a1, w1 = (_rand(m, k1, seed=3).to(DEV), _rand(n, k1, seed=4).to(DEV)) if k1 else (None, None)
    b, w2, b2 = _rand(n, seed=5).to(DEV), _rand(64, n, seed=6).to(DEV), _rand(64, seed=7).to(DEV)
    try:
        assert hip_lib.gts_set_option(1, variant) == 0
        bits = dense.relu_bits_empty(m, n, DEV).fill_(-1)
        out = dense.linear_fwd(a0, w0, a1, w1, bias=b, relu=True, relu_bits=bits)
        assert torch.equal(out, dense.linear_fwd(a0, w0, a1, w1, bias=b, relu=True))
        on = dense.unpack_relu_bits(bits, m, n)
        assert on.shape == (m, n) and torch.equal(on, out > 0)
        assert 0.2 < float(on.float().mean()) < 0.8              # a real mask, not all-on / all-off
        bits2 = dense.relu_bits_empty(m, n, DEV).fill_(-1)
        out_c, _ = dense.linear_fwd_chain(a0, w0, a1, w1, b, True, w2, b2, True, relu_bits=bits2)
        assert torch.equal(out_c, out) and torch.equal(dense.unpack_relu_bits(bits2, m, n), on)
        # consumer: g [m, n0] @ W^T-form weights [n, n0] masked by `out`
        g0, wt = _rand(m, 64, seed=9).to(DEV), _rand(n, 64, seed=10).to(DEV)
        want = dense.linear_bwd_input_t(g0, wt, relu_mask=out)
        assert torch.equal(dense.linear_bwd_input_t(g0, wt, relu_mask=out, relu_bits=bits), want)
        w3 = _rand(64, n, seed=11).to(DEV)
        gin, gin2 = dense.linear_bwd_input_chain_t(g0, wt, None, None, out, w3, relu_bits=bits)
        assert torch.equal(gin, want) and torch.equal(gin2, dense.linear_bwd_input_t(want, w3))
    finally:
        hip_lib.gts_set_option(1, -1)
    assert dense.relu_bits_pay(60000, 256) and not dense.relu_bits_pay(1000, 256) and not dense.relu_bits_pay(60000, 100)
    with pytest.raises(dense._lib.GtsError):
        dense.relu_bits_empty(m, 96, DEV)
    with pytest.raises(dense._lib.GtsError, match="shapes do not match"):
        dense.linear_fwd(a0, w0, a1, w1, bias=b, relu=True, relu_bits=bits[:-1])


def test_transposed_and_chained_entry_points_take_non_contiguous_operands():
    """linear_bwd_input_t / linear_fwd_chain / linear_bwd_input_chain_t read operands as dense row-major matrices:
    a .t() view, a column slice or an expanded gradient must be materialised by the wrapper, not mis-read."""
    from gts import dense

    torch.manual_seed(3)
    m, k, n = 300, 64, 32
    g_wide = torch.randn(m, 2 * n, device="cuda")
    g0 = g_wide[:, ::2]                                  # column slice: stride 2
    w = torch.randn(n, k, device="cuda")
    wt_view = w.t().contiguous().t().t()                 # [k, n] ... a view chain that is contiguous
    wt_strided = w.t()                                   # [k, n] with strides (1, k): NOT contiguous
    want = dense.linear_bwd_input_t(g0.contiguous(), wt_view.contiguous())
    assert torch.equal(dense.linear_bwd_input_t(g0, wt_strided), want)
    expanded = torch.randn(m, 1, device="cuda").expand(m, n)     # stride 0 along the columns
    assert torch.equal(dense.linear_bwd_input_t(expanded, wt_strided),
                       dense.linear_bwd_input_t(expanded.contiguous(), wt_view.contiguous()))

    a0 = torch.randn(k, m, device="cuda").t()            # [m, k] transposed view
    w0, w2 = torch.randn(n, k, device="cuda"), torch.randn(16, n, device="cuda")
    b0, b2 = torch.randn(n, device="cuda"), torch.randn(16, device="cuda")
    out, out2 = dense.linear_fwd_chain(a0, w0, None, None, b0, True, w2.t().contiguous().t().contiguous(), b2, False)
    ref1 = dense.linear_fwd(a0.contiguous(), w0, bias=b0, relu=True)
    assert torch.equal(out, ref1) and torch.equal(out2, dense.linear_fwd(ref1, w2, bias=b2))

    mask = torch.randn(k, m, device="cuda").t()          # non-contiguous ReLU source
    gin, gin2 = dense.linear_bwd_input_chain_t(g0, wt_strided, None, None, mask, torch.randn(8, k, device="cuda"))
    assert torch.equal(gin, dense.linear_bwd_input_t(g0.contiguous(), wt_view.contiguous(), relu_mask=mask.contiguous()))


@pytest.mark.parametrize("m,k,n0,n1", [(60000, 1024, 1024, 0), (60000, 1024, 4, 0), (35000, 256, 256, 256), (59999, 512, 64, 64),
                                        (4800, 512, 64, 0), (1000, 64, 32, 0), (5000, 260, 8, 0)])
@pytest.mark.parametrize("activation", [1, 2])
def test_input_gradient_through_the_activation_of_the_layer_below(m, k, n0, n1, activation):
    """gts_linear_bwd_input_t_act_f32: (g0 w0 + g1 w1) * act'(act_out) and its column sums — in the epilogue of the panel
    launch at the tall 256-column-block shapes (ELU), otherwise GEMM + gat_act_bwd in place.  gin must equal the two
    separate calls bit for bit on either route; the bias gradient is a different fixed-order sum (fp64 yardstick)."""
    from gts import ops
    g0, w0t = _rand(m, n0, seed=1).to(DEV), _rand(k, n0, seed=2).to(DEV)
    g1, w1t = (_rand(m, n1, seed=3).to(DEV), _rand(k, n1, seed=4).to(DEV)) if n1 else (None, None)
    act_out = torch.nn.functional.elu(_rand(m, k, seed=5)).to(DEV) if activation == 1 else _rand(m, k, seed=5).relu().to(DEV)
    plain = dense.linear_bwd_input_t(g0, w0t, g1, w1t)
    want, want_bias = ops.gat_act_bwd(plain, act_out, activation, want_bias_grad=True)
    got, got_bias = dense.linear_bwd_input_t_act(g0, w0t, act_out, activation, g1, w1t)
    assert torch.equal(got, want)
    sums64, bound = got.double().sum(0), got.double().abs().sum(0)
    for bias in (got_bias, want_bias):
        assert torch.all((bias.double() - sums64).abs() <= 2e-6 * bound + 1e-30)
    only, none = dense.linear_bwd_input_t_act(g0, w0t, act_out, activation, g1, w1t, want_bias_grad=False)
    assert none is None and torch.equal(only, want)
