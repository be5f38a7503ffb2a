"""GPU parity tests, one per kernel family of the C ABI (include/ssi_hip.h), against the CPU oracle / plain torch fp32 on
the same seeded inputs.  Integer results bit-exact; fp32 storage within 1e-5 relative; bf16 storage within bf16 rounding
of the fp32 result (tolerances written at each assert)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype):
    return dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


@pytest.fixture(scope="module")
def ops():
    from ssi import ops as o
    return o


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,dim", [(5, 128), (1024, 2048), (3, 256)])
def test_rmsnorm_fwd_bwd(ops, dtype, rows, dim):
    from oracle.llama_oracle import RMSNorm
    x = rnd(rows, dim, dtype=dtype, seed=1)
    w = (1 + 0.1 * rnd(dim, seed=2)).to(dtype)
    dy = rnd(rows, dim, dtype=dtype, seed=3)
    dres = rnd(rows, dim, dtype=dtype, seed=4)
    ref = RMSNorm(dim, 1e-5)
    ref.scale.data = w.clone()
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy)
    xg, wg, dyg = x.to(DEV), w.to(DEV), dy.to(DEV)
    y = torch.empty_like(xg)
    rstd = torch.empty(rows, dtype=torch.float32, device=DEV)
    ops.rmsnorm_fwd(xg, wg, y, rstd, 1e-5)
    torch.testing.assert_close(y.cpu().float(), yr.detach().float(), **tol(dtype))
    torch.testing.assert_close(rstd.cpu(), torch.rsqrt(x.float().pow(2).mean(-1) + 1e-5), rtol=1e-5, atol=1e-6)
    dx = torch.empty_like(xg)
    dscale = torch.zeros(dim, dtype=dtype, device=DEV)
    ops.rmsnorm_bwd(dyg, xg, wg, rstd, dres.to(DEV), dx, dscale)
    t = tol(dtype)
    torch.testing.assert_close(dx.cpu().float(), (xr.grad + dres).float(), rtol=t["rtol"], atol=max(t["atol"], 3e-5 if dtype == torch.float32 else 6e-2))
    torch.testing.assert_close(dscale.cpu().float(), ref.scale.grad.float(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-4,
                               atol=1e-4 * rows if dtype == torch.float32 else 0.05 * math.sqrt(rows))
    # accumulation into dscale + determinism
    d2 = dscale.clone()
    ops.rmsnorm_bwd(dyg, xg, wg, rstd, None, dx, d2)
    if dtype == torch.float32:
        torch.testing.assert_close(d2.cpu(), 2 * dscale.cpu(), rtol=1e-5, atol=1e-5)
    d3 = dscale.clone()
    ops.rmsnorm_bwd(dyg, xg, wg, rstd, None, dx, d3)
    assert torch.equal(d2, d3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_rope_matches_oracle_and_inverts(ops, dtype):
    from oracle.llama_oracle import apply_rope, llama3_scaled_theta, rope_cache
    from ssi.model import llama3_rope_table
    B, S, H, KV, hd = 2, 37, 4, 2, 64
    table = llama3_rope_table(hd, 128)
    torch.testing.assert_close(table, rope_cache(llama3_scaled_theta(hd), 128), rtol=0, atol=0)
    qkv = rnd(B * S, (H + 2 * KV) * hd, dtype=dtype, seed=5)
    ref = qkv.clone().view(B, S, H + 2 * KV, hd)
    ref[:, :, : H + KV] = apply_rope(ref[:, :, : H + KV], table)
    x = qkv.to(DEV)
    ops.rope_(x, S, H + KV, hd, table.to(DEV))
    torch.testing.assert_close(x.cpu().float(), ref.reshape(B * S, -1).float(), rtol=1e-6, atol=1e-6 if dtype == torch.float32 else 8e-3)
    assert torch.equal(x.cpu()[:, (H + KV) * hd:], qkv[:, (H + KV) * hd:])  # v heads untouched
    ops.rope_(x, S, H + KV, hd, table.to(DEV), inverse=True)
    torch.testing.assert_close(x.cpu().float(), qkv.float(), rtol=1e-5, atol=1e-5 if dtype == torch.float32 else 3e-2)
    # explicit positions == implicit row % S
    pos = (torch.arange(B * S) % S).to(torch.int32).to(DEV)
    x1, x2 = qkv.to(DEV), qkv.to(DEV)
    ops.rope_(x1, S, H + KV, hd, table.to(DEV))
    ops.rope_(x2, S, H + KV, hd, table.to(DEV), positions=pos)
    assert torch.equal(x1, x2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_swiglu_fwd_bwd(ops, dtype):
    rows, inter = 33, 512
    gu = rnd(rows, 2 * inter, dtype=dtype, seed=6, scale=2.0)
    da = rnd(rows, inter, dtype=dtype, seed=7)
    gr = gu.clone().requires_grad_(True)
    act_ref = F.silu(gr[:, :inter]) * gr[:, inter:]
    act_ref.backward(da)
    act = torch.empty(rows, inter, dtype=dtype, device=DEV)
    ops.swiglu_fwd(gu.to(DEV), act)
    torch.testing.assert_close(act.cpu().float(), act_ref.detach().float(), **tol(dtype))
    dgu = torch.empty(rows, 2 * inter, dtype=dtype, device=DEV)
    ops.swiglu_bwd(da.to(DEV), gu.to(DEV), dgu)
    torch.testing.assert_close(dgu.cpu().float(), gr.grad.float(), rtol=tol(dtype)["rtol"], atol=1e-5 if dtype == torch.float32 else 6e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_embedding_gather_and_deterministic_scatter(ops, dtype):
    V, D, T = 515, 256, 1000
    table = rnd(V, D, dtype=dtype, seed=8)
    g = torch.Generator().manual_seed(9)
    tok = torch.randint(0, V, (T,), generator=g)
    tok[100:400] = 7          # a heavily repeated id (pad-like) spanning several 64-wide scan chunks
    tok[-1] = 7
    out = torch.empty(T, D, dtype=dtype, device=DEV)
    ops.embed_fwd(tok.to(DEV), table.to(DEV), out, V)
    assert torch.equal(out.cpu(), table[tok])  # bit-exact row copies
    dout = rnd(T, D, dtype=dtype, seed=10)
    base = rnd(V, D, dtype=dtype, seed=11)
    ref = base.float().index_add(0, tok, dout.float())
    dt1 = base.to(DEV)
    ops.embed_bwd(tok.to(DEV), dout.to(DEV), dt1, V)
    torch.testing.assert_close(dt1.cpu().float(), ref, rtol=1e-5 if dtype == torch.float32 else 8e-3, atol=1e-4 if dtype == torch.float32 else 0.15)
    dt2 = base.to(DEV)
    ops.embed_bwd(tok.to(DEV), dout.to(DEV), dt2, V)
    assert torch.equal(dt1, dt2), "scatter-add must be bitwise reproducible"
    untouched = torch.ones(V, dtype=torch.bool)
    untouched[tok] = False
    assert torch.equal(dt1.cpu()[untouched], base[untouched])
    # empty input is a no-op
    ops.embed_bwd(torch.empty(0, dtype=torch.int64, device=DEV), torch.empty(0, D, dtype=dtype, device=DEV), dt2, V)
    assert torch.equal(dt1, dt2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("vocab,ld", [(515, 520), (133_258, 133_376), (136_450, 136_704), (130_306, 130_560), (20_000, 20_480)])
def test_cross_entropy_rows_reduce_and_grad(ops, dtype, vocab, ld):
    rows = 24
    logits = rnd(rows, ld, dtype=dtype, seed=12, scale=3.0)
    g = torch.Generator().manual_seed(13)
    labels = torch.randint(0, vocab, (rows,), generator=g)
    labels[3] = -100
    labels[rows - 1] = -100
    labels[5] = vocab - 1
    lr = logits[:, :vocab].float().clone().requires_grad_(True)
    nll = F.cross_entropy(lr, labels, ignore_index=-100, reduction="none")
    nll.sum().backward()
    work = logits.to(DEV)
    row_loss = torch.empty(rows, dtype=torch.float32, device=DEV)
    row_lse = torch.empty(rows, dtype=torch.float32, device=DEV)
    ops.ce_fwd(work, labels.to(DEV), vocab, -100, row_loss, row_lse, False)
    assert torch.equal(work.cpu(), logits), "write_grad=False must leave the logits untouched"
    torch.testing.assert_close(row_loss.cpu(), nll.detach(), rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(row_lse.cpu()[labels != -100], torch.logsumexp(lr.detach(), -1)[labels != -100], rtol=1e-6, atol=1e-5)
    out = torch.empty(4, dtype=torch.float32, device=DEV)
    ops.ce_reduce(row_loss, labels.to(DEV), vocab, -100, out)
    n_valid = int((labels != -100).sum())
    assert out.cpu()[2].item() == n_valid and out.cpu()[3].item() == 0
    assert out.cpu()[0].item() == pytest.approx(float(nll.detach().sum()) / n_valid, rel=1e-5)
    ops.ce_fwd(work, labels.to(DEV), vocab, -100, row_loss, None, True)
    grad = work.cpu().float()
    torch.testing.assert_close(grad[:, :vocab], lr.grad, rtol=1e-5 if dtype == torch.float32 else 1e-2, atol=1e-6 if dtype == torch.float32 else 4e-3)
    assert (grad[:, vocab:] == 0).all() and (grad[3] == 0).all() and (grad[rows - 1] == 0).all()
    # all rows ignored -> 0/0 = NaN like the reference
    allign = torch.full((rows,), -100, dtype=torch.int64, device=DEV)
    ops.ce_fwd(logits.to(DEV), allign, vocab, -100, row_loss, None, False)
    ops.ce_reduce(row_loss, allign, vocab, -100, out)
    assert math.isnan(out.cpu()[0].item()) and out.cpu()[2].item() == 0
    # a label outside [0, vocab) (torch would device-assert): zero loss, zero gradient row, NOT counted as valid, reported in out[3]
    bad = labels.clone()
    bad[0], bad[7] = vocab, -5
    work = logits.to(DEV)
    ops.ce_fwd(work, bad.to(DEV), vocab, -100, row_loss, None, True)
    ops.ce_reduce(row_loss, bad.to(DEV), vocab, -100, out)
    o = out.cpu()
    ok_rows = (bad != -100) & (bad >= 0) & (bad < vocab)
    assert o[3].item() == 2 and o[2].item() == int(ok_rows.sum())
    assert o[1].item() == pytest.approx(float(nll.detach()[ok_rows].sum()), rel=1e-5)
    assert (work.cpu().float()[0] == 0).all() and (work.cpu().float()[7] == 0).all() and row_loss.cpu()[0] == 0


def test_cross_entropy_register_resident_rows_many_rows_per_workgroup(ops):
    """More rows than workgroups (each of the 256 walks several rows, the next row's loads issued from inside the gradient pass),
    ignored rows in between, against fp32 torch on the CPU and against the 3-pass kernel's fp32 instance."""
    rows, vocab, ld = 700, 9000, 9216
    logits = rnd(rows, ld, dtype=torch.bfloat16, seed=120, scale=4.0)
    labels = torch.randint(0, vocab, (rows,), generator=torch.Generator().manual_seed(121))
    labels[::7] = -100
    labels[300:330] = -100
    lr = logits[:, :vocab].float().clone().requires_grad_(True)
    nll = F.cross_entropy(lr, labels, ignore_index=-100, reduction="none")
    nll.sum().backward()
    work = logits.to(DEV)
    row_loss = torch.empty(rows, dtype=torch.float32, device=DEV)
    row_lse = torch.empty(rows, dtype=torch.float32, device=DEV)
    ops.ce_fwd(work, labels.to(DEV), vocab, -100, row_loss, row_lse, True)
    torch.testing.assert_close(row_loss.cpu(), nll.detach(), rtol=1e-5, atol=2e-5)
    grad = work.cpu().float()
    torch.testing.assert_close(grad[:, :vocab], lr.grad, rtol=1e-2, atol=4e-3)
    assert (grad[:, vocab:] == 0).all() and (grad[labels == -100] == 0).all()
    work2 = logits.to(DEV)
    ops.ce_fwd(work2, labels.to(DEV), vocab, -100, row_loss, None, True)
    assert torch.equal(work, work2), "not bitwise reproducible"
    f32 = logits.float().to(DEV)                       # the 3-pass kernel (fp32 instance) on the same values
    rl32 = torch.empty(rows, dtype=torch.float32, device=DEV)
    ops.ce_fwd(f32, labels.to(DEV), vocab, -100, rl32, None, True)
    torch.testing.assert_close(row_loss.cpu(), rl32.cpu(), rtol=1e-5, atol=2e-5)
    assert float((f32.cpu() - grad).abs().max()) <= 4e-3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("vocab,ld", [(515, 520), (133_258, 133_376), (20_000, 20_480)])
def test_cross_entropy_with_a_weight_per_row(ops, dtype, vocab, ld):
    """``ssi_ce_fwd_weighted`` (ABI v8; an accumulation window run as one batch, ``ssi/data/window.py``): loss and gradient of row r times
    w[r], against fp32 torch; weights of exactly 1 give the bits of the unweighted call, a weight of 0 a zero row.  Both kernels: the
    register-resident bf16 one (the weight rides in the exponent) and the 3-pass one."""
    rows = 300
    logits = rnd(rows, ld, dtype=dtype, seed=31, scale=3.0)
    g = torch.Generator().manual_seed(32)
    labels = torch.randint(0, vocab, (rows,), generator=g)
    labels[::9] = -100
    w = 0.97 + 0.06 * torch.rand(rows, generator=g)        # (the real ones differ from 1 by a few 1e-4)
    w[5], w[6], w[7] = 0.0, 1.0, 2.5
    lr = logits[:, :vocab].float().clone().requires_grad_(True)
    nll = F.cross_entropy(lr, labels, ignore_index=-100, reduction="none")
    (w * nll).sum().backward()
    work = logits.to(DEV)
    row_loss = torch.empty(rows, dtype=torch.float32, device=DEV)
    ops.ce_fwd(work, labels.to(DEV), vocab, -100, row_loss, None, True, row_weight=w.to(DEV))
    torch.testing.assert_close(row_loss.cpu(), (w * nll).detach(), rtol=2e-5, atol=2e-5)
    grad = work.cpu().float()
    # (fp32: 5e-5 — one element of 40 M, a probability of 0.74 in a 133 258-column row, sits 2.8e-5 from torch's)
    torch.testing.assert_close(grad[:, :vocab], lr.grad, rtol=5e-5 if dtype == torch.float32 else 1e-2, atol=1e-6 if dtype == torch.float32 else 4e-3)
    assert (grad[:, vocab:] == 0).all() and (grad[labels == -100] == 0).all() and (grad[5] == 0).all() and float(row_loss[5]) == 0.0
    out = torch.empty(4, dtype=torch.float32, device=DEV)
    ops.ce_reduce(row_loss, labels.to(DEV), vocab, -100, out)
    n_valid = int((labels != -100).sum())
    assert out.cpu()[2].item() == n_valid and out.cpu()[0].item() == pytest.approx(float((w * nll).sum()) / n_valid, rel=2e-5)
    # the weighted sum of the exact (unrounded) gradient rows is what matters downstream: relative error of the row sums of |grad|
    plain, ones = logits.to(DEV), logits.to(DEV)
    rl_plain, rl_ones = torch.empty_like(row_loss), torch.empty_like(row_loss)
    ops.ce_fwd(plain, labels.to(DEV), vocab, -100, rl_plain, None, True)
    ops.ce_fwd(ones, labels.to(DEV), vocab, -100, rl_ones, None, True, row_weight=torch.ones(rows, device=DEV))
    assert torch.equal(plain, ones) and torch.equal(rl_plain, rl_ones), "weights of 1 must change no bit"
    assert torch.equal(work[6], plain[6]) and float(row_loss[6]) == float(rl_plain[6])
    with pytest.raises(AssertionError):
        ops.ce_fwd(work, labels.to(DEV), vocab, -100, row_loss, None, True, row_weight=torch.ones(rows - 1, device=DEV))


def test_count_tokens_matches_reference_counts(ops):
    from oracle.step_oracle import count_token_types, token_type_ranges
    from ssi.train_utils import count_token_types as ct_gpu, count_token_types_async
    r = token_type_ranges(128000, 5000, True, 256)
    g = torch.Generator().manual_seed(14)
    tok = torch.randint(0, 133_258, (8, 2048), generator=g)
    tok[:, -17:] = 133_006
    lab = tok.clone()
    lab[:, :25] = -100
    ref = count_token_types(tok, r, 133_006)
    assert ct_gpu(tok.to(DEV), r, 133_006) == ref  # bit-exact integers
    dev = count_token_types_async(tok.to(DEV), r, 133_006, lab.to(DEV), -100).cpu().tolist()
    assert dev[:4] == [ref[k] for k in r] and dev[4] == ref["total"] and dev[5] == int((lab != -100).sum())
    empty = torch.empty(0, 4, dtype=torch.int64, device=DEV)
    assert ct_gpu(empty, r, 0) == {"text": 0, "dsu": 0, "modality": 0, "special_text": 0, "total": 0}


@pytest.mark.parametrize("dtype", DTYPES)
def test_adamw_scale_sumsq(ops, dtype):
    n = 100_003  # odd tail
    p0, g0 = rnd(n, seed=15), rnd(n, seed=16, scale=50.0)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref_p], lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    p, m, v = p0.to(dtype).to(DEV), torch.zeros(n, dtype=dtype, device=DEV), torch.zeros(n, dtype=dtype, device=DEV)
    gs = torch.tensor([1 / 50.0], dtype=torch.float32, device=DEV)
    for step in range(1, 4):
        gstep = g0 * (1 + 0.1 * step)
        ref_p.grad = (gstep / 50.0).clone()
        opt.step()
        g = gstep.to(dtype).to(DEV)
        ops.adamw_step(p, g, m, v, lr=2e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=step, grad_scale_dev=gs, zero_grad=True)
        assert (g == 0).all()
    if dtype == torch.float32:
        torch.testing.assert_close(p.cpu(), ref_p.detach(), rtol=0, atol=2e-6)
        torch.testing.assert_close(m.cpu(), opt.state[ref_p]["exp_avg"], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(v.cpu(), opt.state[ref_p]["exp_avg_sq"], rtol=1e-4, atol=1e-7)
    else:
        torch.testing.assert_close(p.cpu().float(), ref_p.detach(), rtol=0, atol=2e-2)
    x = rnd(n, dtype=dtype, seed=17)
    out = torch.empty(1, dtype=torch.float32, device=DEV)
    ops.sumsq(x.to(DEV), out)
    assert out.item() == pytest.approx(float(x.float().pow(2).sum()), rel=1e-5)
    xs = x.to(DEV)
    ops.scale_(xs, 0.5, torch.tensor([4.0], device=DEV))
    torch.testing.assert_close(xs.cpu().float(), (x.float() * 2.0).to(dtype).float(), rtol=0, atol=0)


# ---------------------------------------------------------------------------------------------------------------------
def _gemm_ref(layout, a, b):
    a, b = a.double(), b.double()
    if layout == 0:
        return a @ b.T
    return a @ b if layout == 1 else a.T @ b


def _gemm_operands(layout, M, N, K, dtype, seed, integer=False):
    g = torch.Generator().manual_seed(seed)
    mk = (lambda *s: torch.randint(-3, 4, s, generator=g).float()) if integer else (lambda *s: torch.randn(*s, generator=g))
    a = mk(M, K) if layout in (0, 1) else mk(K, M)
    b = mk(N, K) if layout == 0 else mk(K, N)
    return a.to(dtype), b.to(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (70, 130, 33), (256, 64, 512)])
def test_gemm_generic(ops, dtype, layout, M, N, K):
    from ssi import _lib
    a, b = _gemm_operands(layout, M, N, K, dtype, 18)
    prev = ops.set_impl(_lib.IMPL_GENERIC)
    try:
        c = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
        ops.gemm(layout, a.to(DEV), b.to(DEV), c)
        ref = _gemm_ref(layout, a, b)
        torch.testing.assert_close(c.cpu().double(), ref, rtol=1e-5 if dtype == torch.float32 else 1e-2, atol=1e-4 if dtype == torch.float32 else 0.02 * math.sqrt(K))
        if dtype == torch.float32:
            r = rnd(M, N, seed=19)
            c0 = rnd(M, N, seed=20).to(DEV)
            c1 = c0.clone()
            ops.gemm(layout, a.to(DEV), b.to(DEV), c1, residual=r.to(DEV), alpha=0.5, alpha_dev=torch.tensor([4.0], device=DEV), accumulate=True)
            torch.testing.assert_close(c1.cpu().double(), c0.cpu().double() + 2.0 * ref + r.double(), rtol=1e-5, atol=1e-4)
    finally:
        ops.set_impl(prev)


@pytest.mark.parametrize("impl_name", ["IMPL_MFMA"])
@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 192), (1024, 256, 2048)])
def test_gemm_mfma_bf16(ops, impl_name, layout, M, N, K):
    from ssi import _lib
    prev = ops.set_impl(getattr(_lib, impl_name))
    try:
        # (1) small-integer operands: every product and partial sum is exact in fp32 and representable in bf16 when
        #     |sum| <= 256 — catches any row/column/k mapping error with asymmetric data
        a, b = _gemm_operands(layout, M, N, min(K, 64), torch.bfloat16, 21, integer=True)
        c = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.gemm(layout, a.to(DEV), b.to(DEV), c)
        ref = _gemm_ref(layout, a, b)
        exact = ref.abs() <= 256
        assert torch.equal(c.cpu().double()[exact], ref[exact]), "MFMA GEMM is not exact on small-integer operands"
        # (2) random operands, full K, with alpha / residual / accumulate
        a, b = _gemm_operands(layout, M, N, K, torch.bfloat16, 22)
        r = rnd(M, N, dtype=torch.bfloat16, seed=23)
        c0 = rnd(M, N, dtype=torch.bfloat16, seed=24)
        c1 = c0.to(DEV)
        ops.gemm(layout, a.to(DEV), b.to(DEV), c1, residual=r.to(DEV), alpha=0.5, alpha_dev=torch.tensor([2.0], device=DEV), accumulate=True)
        ref = (_gemm_ref(layout, a, b).float().bfloat16().double() + c0.double()) + r.double()
        torch.testing.assert_close(c1.cpu().double(), ref, rtol=2e-2, atol=0.03 * math.sqrt(K))
        # (3) against the generic kernel on the same inputs (same rounding points): near-identical
        c2 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ops.gemm(layout, a.to(DEV), b.to(DEV), c2)
        ops.set_impl(_lib.IMPL_GENERIC)
        c3 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ops.gemm(layout, a.to(DEV), b.to(DEV), c3)
        diff = (c2.float() - c3.float()).abs()
        assert float(diff.max()) <= 2 ** -6 * float(c3.float().abs().max()) and float((diff > 0).float().mean()) < 0.2
    finally:
        ops.set_impl(prev)


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("mode", ["plain", "accumulate", "residual"])
def test_gemm_persistent_kernel_walks_many_tiles(ops, layout, mode):
    """4-wave persistent kernel (NT and TN forms): 512 output tiles on 256 CUs, so every workgroup runs its tile loop twice
    and the operand stream crosses a tile boundary; exact on small integers and bit-identical to the 8-wave kernel."""
    from ssi import _lib
    M, N, K = 4096, 8192, 256
    a, b = _gemm_operands(layout, M, N, K, torch.bfloat16, 61, integer=True)
    a, b = a.to(DEV), b.to(DEV)
    r = torch.randint(-3, 4, (M, N), generator=torch.Generator().manual_seed(62)).to(torch.bfloat16).to(DEV)
    kw = {"accumulate": True, "alpha": 0.5} if mode == "accumulate" else ({"residual": r, "alpha": 0.25} if mode == "residual" else {})
    outs = []
    for impl in (_lib.IMPL_MFMA, _lib.IMPL_MFMA_WG8):
        prev = ops.set_impl(impl)
        try:
            c = r.clone() if mode == "accumulate" else torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            ops.gemm(layout, a, b, c, **kw)
            outs.append(c)
        finally:
            ops.set_impl(prev)
    assert torch.equal(outs[0], outs[1])
    if mode == "plain":
        ref = (a.float() @ b.float().t()) if layout == 0 else ((a.float() @ b.float()) if layout == 1 else (a.float().t() @ b.float()))
        exact = ref.abs() <= 256
        assert torch.equal(outs[0].float()[exact], ref[exact])
    # random data, longer K: still bit-identical (same accumulation order in both kernels)
    a, b = _gemm_operands(layout, 512, 768, 1024, torch.bfloat16, 63)
    cs = []
    for impl in (_lib.IMPL_MFMA, _lib.IMPL_MFMA_WG8):
        prev = ops.set_impl(impl)
        try:
            c = torch.empty(512, 768, dtype=torch.bfloat16, device=DEV)
            ops.gemm(layout, a.to(DEV), b.to(DEV), c)
            cs.append(c)
        finally:
            ops.set_impl(prev)
    assert torch.equal(cs[0], cs[1])


def test_gemm_persistent_kernels_share_the_gpu_with_the_dynamic_tile_order(ops):
    """Two persistent GEMMs launched back to back on two streams: each asks for every CU, so half of the workgroups of either
    start late or only when the other kernel ends (what RCCL's kernels do to the backward GEMMs during the gradient exchange).
    The dynamic tile scheduler must hand every tile out exactly once whatever the interleaving: results equal the serial ones."""
    ops.set_gemm_tile_order(dynamic=True)
    a1, b1 = _gemm_operands(0, 4096, 4096, 1024, torch.bfloat16, 81)
    a2, b2 = _gemm_operands(2, 2048, 8192, 2048, torch.bfloat16, 82)
    a1, b1, a2, b2 = a1.to(DEV), b1.to(DEV), a2.to(DEV), b2.to(DEV)
    ref1 = torch.empty(4096, 4096, dtype=torch.bfloat16, device=DEV)
    ref2 = torch.empty(2048, 8192, dtype=torch.bfloat16, device=DEV)
    ops.gemm(0, a1, b1, ref1)
    ops.gemm(2, a2, b2, ref2)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        c1 = torch.full_like(ref1, float("nan"))
        c2 = torch.full_like(ref2, float("nan"))
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            for _ in range(3):
                ops.gemm(0, a1, b1, c1)
        with torch.cuda.stream(s2):
            for _ in range(3):
                ops.gemm(2, a2, b2, c2)
        torch.cuda.synchronize()
        assert torch.equal(c1, ref1) and torch.equal(c2, ref2)
    ops.set_gemm_tile_order(dynamic=False)
    # the dynamic order on every layout / epilogue class, many tiles per workgroup, alone on the GPU: same bits as the static order
    for layout in (0, 1, 2):
        a, b = _gemm_operands(layout, 4096, 8192, 512, torch.bfloat16, 83 + layout)
        a, b = a.to(DEV), b.to(DEV)
        r = rnd(4096, 8192, dtype=torch.bfloat16, seed=86).to(DEV)
        outs = []
        for dyn in (False, True):
            ops.set_gemm_tile_order(dynamic=dyn)
            c = torch.full((4096, 8192), float("nan"), dtype=torch.bfloat16, device=DEV)
            ops.gemm(layout, a, b, c, residual=r)
            outs.append(c)
        ops.set_gemm_tile_order(dynamic=False)
        assert torch.equal(outs[0], outs[1])


def test_gemm_weight_gradient_form_beyond_2gib_of_k_offset(ops):
    """TN form on a column window of a [K, ld] operand whose K extent spans more than 2 GiB (the LM-head weight gradient is
    such a case): the K offset must not ride in a 32-bit buffer offset."""
    from ssi import _lib
    K, ld, M, N = 8192, 147_456, 512, 256
    big = torch.empty(K, ld, dtype=torch.bfloat16, device=DEV)
    big.normal_(generator=torch.Generator(device=DEV).manual_seed(64))
    a = big[:, 1024:1024 + M]          # rows k, leading dimension ld: K * ld * 2 B = 2.4 GB
    b = torch.randn(K, N, device=DEV, generator=torch.Generator(device=DEV).manual_seed(65)).bfloat16()
    cs = []
    for impl in (_lib.IMPL_MFMA, _lib.IMPL_MFMA_WG8):
        prev = ops.set_impl(impl)
        try:
            c = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            ops.gemm(ops.GEMM_TN, a, b, c)
            cs.append(c)
        finally:
            ops.set_impl(prev)
    assert torch.equal(cs[0], cs[1])
    ref = a.float().t() @ b.float()
    torch.testing.assert_close(cs[0].float(), ref, rtol=2e-2, atol=0.03 * math.sqrt(K))


def test_gemm_mfma_forced_on_bad_shape_fails_loudly(ops):
    from ssi import _lib
    prev = ops.set_impl(_lib.IMPL_MFMA)
    try:
        a, b = _gemm_operands(0, 100, 256, 64, torch.bfloat16, 25)
        with pytest.raises(RuntimeError):
            ops.gemm(0, a.to(DEV), b.to(DEV), torch.empty(100, 256, dtype=torch.bfloat16, device=DEV))
    finally:
        ops.set_impl(prev)


# ---------------------------------------------------------------------------------------------------------------------
def _sdpa_ref(qkv, B, S, H, KV, hd):
    q = qkv[:, : H * hd].view(B, S, H, hd).transpose(1, 2)
    k = qkv[:, H * hd:(H + KV) * hd].view(B, S, KV, hd).transpose(1, 2).repeat_interleave(H // KV, dim=1)
    v = qkv[:, (H + KV) * hd:].view(B, S, KV, hd).transpose(1, 2).repeat_interleave(H // KV, dim=1)
    o = F.scaled_dot_product_attention(q, k, v, is_causal=True)
    return o.transpose(1, 2).reshape(B * S, H * hd)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,H,KV,hd", [(2, 45, 8, 2, 16), (2, 128, 4, 1, 64), (1, 200, 4, 4, 64)])
def test_attention_generic_fwd_bwd(ops, dtype, B, S, H, KV, hd):
    from ssi import _lib
    prev = ops.set_impl(_lib.IMPL_GENERIC)
    try:
        qkv = rnd(B * S, (H + 2 * KV) * hd, dtype=dtype, seed=26)
        do = rnd(B * S, H * hd, dtype=dtype, seed=27)
        qr = qkv.float().clone().requires_grad_(True)
        oref = _sdpa_ref(qr, B, S, H, KV, hd)
        oref.backward(do.float())
        out = torch.empty(B * S, H * hd, dtype=dtype, device=DEV)
        lse = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
        ops.attn_fwd(qkv.to(DEV), out, lse, B, S, H, KV, hd)
        t = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(out.cpu().float(), oref.detach(), **t)
        dqkv = torch.full_like(qkv, float("nan")).to(DEV)
        delta = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
        ops.attn_bwd(qkv.to(DEV), out, do.to(DEV), lse, dqkv, delta, B, S, H, KV, hd)
        t = dict(rtol=1e-3, atol=1e-4) if dtype == torch.float32 else dict(rtol=5e-2, atol=8e-2)
        torch.testing.assert_close(dqkv.cpu().float(), qr.grad, **t)
    finally:
        ops.set_impl(prev)


@pytest.mark.parametrize("B,S,H,KV", [(2, 128, 4, 1), (1, 256, 4, 2), (2, 384, 2, 2), (1, 2048, 8, 2)])
def test_attention_mfma_fwd_bwd(ops, B, S, H, KV):
    """MFMA flash attention (bf16, head_dim 64) vs torch SDPA in fp32 on the same bf16 inputs, and vs the generic HIP
    kernel.  Includes a spiked key (forces the online-softmax rescale on a late tile) and a large-magnitude query."""
    from ssi import _lib
    hd = 64
    qkv = rnd(B * S, (H + 2 * KV) * hd, dtype=torch.bfloat16, seed=28)
    qkv[S // 2 + 3, H * hd: H * hd + hd] *= 6.0      # one key row of kv head 0 stands out for later queries
    qkv[S - 5, :hd] *= 4.0                            # one query row with large scores
    do = rnd(B * S, H * hd, dtype=torch.bfloat16, seed=29)
    qr = qkv.float().clone().requires_grad_(True)
    oref = _sdpa_ref(qr, B, S, H, KV, hd)
    oref.backward(do.float())
    outs = {}
    for impl in (_lib.IMPL_MFMA, _lib.IMPL_GENERIC):
        prev = ops.set_impl(impl)
        try:
            out = torch.full((B * S, H * hd), float("nan"), dtype=torch.bfloat16, device=DEV)
            lse = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
            ops.attn_fwd(qkv.to(DEV), out, lse, B, S, H, KV, hd)
            dqkv = torch.full_like(qkv, float("nan")).to(DEV)
            delta = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
            ops.attn_bwd(qkv.to(DEV), out, do.to(DEV), lse, dqkv, delta, B, S, H, KV, hd)
            outs[impl] = (out.cpu().float(), lse.cpu(), dqkv.cpu().float())
        finally:
            ops.set_impl(prev)
    out, lse, dqkv = outs[_lib.IMPL_MFMA]
    assert torch.isfinite(out).all() and torch.isfinite(dqkv).all()
    torch.testing.assert_close(out, oref.detach(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(lse, outs[_lib.IMPL_GENERIC][1], rtol=1e-4, atol=2e-3)
    # gradients: relative to the tensor's scale (bf16 P/dS operands)
    scale = float(qr.grad.abs().max())
    err = float((dqkv - qr.grad).abs().max())
    assert err <= 3e-2 * scale, f"dqkv max error {err} vs scale {scale}"
    rel_fro = float((dqkv - qr.grad).norm() / qr.grad.norm())
    assert rel_fro <= 1.5e-2, rel_fro
    # bitwise reproducible (no atomics)
    prev = ops.set_impl(_lib.IMPL_MFMA)
    try:
        out2 = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=DEV)
        lse2 = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
        ops.attn_fwd(qkv.to(DEV), out2, lse2, B, S, H, KV, hd)
        d2 = torch.empty_like(qkv).to(DEV)
        ops.attn_bwd(qkv.to(DEV), out2, do.to(DEV), lse2, d2, torch.empty(B * H * S, dtype=torch.float32, device=DEV), B, S, H, KV, hd)
        assert torch.equal(out2.cpu().float(), out) and torch.equal(d2.cpu().float(), dqkv)
    finally:
        ops.set_impl(prev)


@pytest.fixture
def attn_impl(ops):
    """Switch the attention backward kernels through the ABI's setter (``ssi_set_attn_impl``) and put the previous modes back afterwards."""
    from ssi import _lib
    saved = {}

    def choose(which, mode):
        prev = ops.set_attn_impl(which, mode)
        saved.setdefault(which, prev)

    yield choose
    for which, mode in saved.items():
        ops.set_attn_impl(which, mode)


@pytest.mark.parametrize("B,S,H,KV", [(3, 256, 4, 1), (2, 512, 8, 2), (1, 1024, 32, 8), (2, 2048, 8, 2), (8, 4096, 32, 8), (1, 8192, 8, 2),
                                      (5, 1280, 4, 1)])
@pytest.mark.parametrize("fused_rope", [False, True])
def test_attention_dkv_pipelined_kernel_matches_the_128_key_kernel(ops, B, S, H, KV, fused_rope, attn_impl):
    """Round 4: dK / dV on the one-wave-per-SIMD pipelined kernel (256-key groups, masked diagonal tiles first) against the round-1..3
    kernel on the same inputs.  Same products and operands, another order of the sums over the query tiles: equal to fp32 rounding of
    the sums (the bf16 results differ by at most one bf16 step), the dQ block untouched, and reproducible run to run.  S = 256 has no
    unmasked tile at all, 4 query heads per kv head as the kernel requires (other ratios fall back to the old kernel: covered by
    test_attention_mfma_fwd_bwd)."""
    from ssi import _lib
    hd = 64
    qkv = rnd(B * S, (H + 2 * KV) * hd, dtype=torch.bfloat16, seed=61)
    qkv[S // 2 + 3, H * hd: H * hd + hd] *= 6.0
    do = rnd(B * S, H * hd, dtype=torch.bfloat16, seed=62)
    table = pos = None
    if fused_rope:
        table = rnd(S + 8, hd // 2, 2, dtype=torch.float32, seed=63).to(DEV)  # [position][pair][cos, sin]; any values do: the map is linear
    prev = ops.set_impl(_lib.IMPL_MFMA)
    try:
        x, dout = qkv.to(DEV), do.to(DEV)
        out = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
        delta = torch.empty_like(lse)
        ops.attn_fwd(x, out, lse, B, S, H, KV, hd)
        res = {}
        for sel in ("1", "2", "2"):   # 1 = the 128-key kernel, 2 = the pipelined kernel even where its workgroups cannot fill the chip (small cases)
            attn_impl(_lib.ATTN_KERNEL_DKV, int(sel))
            d = torch.full_like(x, float("nan"))
            ops.attn_bwd(x, out, dout, lse, d, delta, B, S, H, KV, hd, rope_table=table, positions=pos)
            assert bool(ops.attn_last_dispatch() & _lib.ATTN_USED_DKV2) == (sel == "2"), "the dispatcher ignored the switch"
            res.setdefault(sel, []).append(d.cpu().float())
    finally:
        ops.set_impl(prev)
    old, (new, new2) = res["1"][0], res["2"]
    assert torch.isfinite(new).all()
    assert torch.equal(new, new2), "not reproducible"
    assert torch.equal(old[:, : H * hd], new[:, : H * hd]), "the dQ block belongs to the other kernel"
    a, b = old[:, H * hd:], new[:, H * hd:]
    assert not torch.equal(a, b) or S == 256, "the pipelined kernel did not run"
    rel = float((a - b).norm() / a.norm())
    assert rel <= 3e-4, rel
    assert float((a - b).abs().max()) <= 2.0 ** -7 * float(a.abs().max()), "more than a bf16 step apart"


@pytest.mark.parametrize("B,S,H,KV", [(1, 512, 4, 1), (3, 512, 8, 2), (2, 1024, 32, 8), (1, 2048, 8, 2), (8, 2048, 32, 8),
                                      (2, 128, 8, 2), (1, 256, 4, 1), (3, 384, 4, 1), (2, 2048, 32, 8), (8, 4096, 32, 8), (1, 8192, 8, 2),
                                      (3, 640, 4, 1), (1, 1152, 8, 2)])
@pytest.mark.parametrize("fused_rope", [False, True])
def test_attention_dq_pipelined_persistent_kernel_matches_the_round_1_kernel(ops, B, S, H, KV, fused_rope, attn_impl):
    """Round 4: dQ on the one-wave-per-SIMD pipelined kernel with persistent workgroups (8 query blocks of 64 per workgroup, the next block's
    Q / dO / O rows and the RoPE table rows staged through LDS) against the round-1..3 kernel on the same inputs.  Same products, same order
    of the sums over the key tiles; the exponent is one fma of the unscaled S (x log2(e)/8) where the old kernel scales Q by 1/8 first —
    the same value — so the two agree to the last bits of the bf16 result on all but a few elements.  delta (which the dK / dV kernel reads)
    and the dK / dV blocks must be IDENTICAL, the result reproducible.  S = 512: one workgroup per (batch, kv head) with all 8 blocks, B * KV
    not a multiple of 8 (second case: the plain workgroup -> pair map); S = 1024 / 2048: 2 / 4 workgroups per pair in zig-zag groups; the
    fifth case is the step's shape; S = 128, 256, 384: 2, 4 and 2 blocks per workgroup (a single pair of blocks; 3 workgroups per pair); the
    ninth is the reference's default SFT micro-batch, 2 x 2048, where the dispatcher takes 2 blocks per workgroup (256 workgroups).  Those two
    and (8, 4096) — BASELINE config C's one-GPU batch — the dispatcher picks by itself, the others are forced (mode NEW); S = 640 and 1152: 10
    and 18 query blocks, i.e. 2 blocks per workgroup with 5 and 9 workgroups per pair (odd counts)."""
    from ssi import _lib
    hd = 64
    qkv = rnd(B * S, (H + 2 * KV) * hd, dtype=torch.bfloat16, seed=81)
    qkv[S // 2 + 3, :hd] *= 6.0
    do = rnd(B * S, H * hd, dtype=torch.bfloat16, seed=82)
    table = rnd(S + 8, hd // 2, 2, dtype=torch.float32, seed=83).to(DEV) if fused_rope else None   # any values do: the map is linear
    prev = ops.set_impl(_lib.IMPL_MFMA)
    try:
        x, dout = qkv.to(DEV), do.to(DEV)
        out = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
        ops.attn_fwd(x, out, lse, B, S, H, KV, hd)
        res = {}
        for sel in ("1", "2", "2") + (("0",) if (B, S) in ((8, 2048), (2, 2048), (8, 4096)) else ()):
            attn_impl(_lib.ATTN_KERNEL_DQ, int(sel))
            d = torch.full_like(x, float("nan"))
            delta = torch.full_like(lse, float("nan"))
            ops.attn_bwd(x, out, dout, lse, d, delta, B, S, H, KV, hd, rope_table=table)
            assert bool(ops.attn_last_dispatch() & _lib.ATTN_USED_DQ2) == (sel != "1"), "the dispatcher ignored the switch"
            res.setdefault(sel, []).append((d.cpu().float(), delta.cpu()))
    finally:
        ops.set_impl(prev)
    (old, dl_old), ((new, dl_new), (new2, _)) = res["1"][0], res["2"]
    assert torch.isfinite(new).all()
    assert torch.equal(new, new2), "not reproducible"
    assert torch.equal(dl_old, dl_new), "delta differs"
    assert torch.equal(old[:, H * hd:], new[:, H * hd:]), "the dK / dV blocks belong to the other kernel"
    if "0" in res:
        assert torch.equal(res["0"][0][0], new), "the dispatcher did not pick the pipelined kernel for the step's shape"
    a, b = old[:, : H * hd], new[:, : H * hd]
    assert not torch.equal(a, b), "the pipelined kernel did not run"
    assert float((a != b).float().mean()) <= 0.02
    rel = float((a - b).norm() / a.norm())
    assert rel <= 3e-4, rel
    assert float((a - b).abs().max()) <= 2.0 ** -7 * float(a.abs().max()), "more than a bf16 step apart"


@pytest.mark.parametrize("B,S,H,KV,rows", [(2, 256, 4, 1, None), (1, 512, 8, 2, None), (2, 2048, 32, 8, None),
                                          (1, 1024, 8, 2, [[300, 37, 500, 187]]),
                                          (1, 11520, 32, 8, [[1807, 2038, 1909, 1924, 1500, 2048, 294]])])   # one long packed row: 720 workgroups, two heads each
@pytest.mark.parametrize("fused_rope", [False, True])
def test_attention_backward_with_workspace_splits_dkv_over_the_query_heads(ops, B, S, H, KV, rows, fused_rope):
    """ABI v6: with a caller-owned workspace, launches too small to fill the chip (the reference's default micro-batch of 2 x 2048 rows is
    one: third case) run dK / dV as one workgroup per query head + a reduction over the heads.  Against the same call without workspace:
    dQ untouched, dK / dV equal to the rounding of another summation order, reproducible; shapes that fill the chip ask for no workspace."""
    from ssi import _lib
    hd = 64
    assert ops.attn_bwd_workspace_bytes(8, 2048, 32, 8, hd, torch.bfloat16) == 0          # the headline shape needs none
    assert ops.attn_bwd_workspace_bytes(2, 2048, 32, 1, hd, torch.float32) == 0           # fp32: generic kernels
    want = ops.attn_bwd_workspace_bytes(B, S, H, KV, hd, torch.bfloat16)
    slots = (H // KV) if B * KV * (S // 128) < 512 else 2   # all heads apart below 512 workgroups, two halves below 1024
    assert want == slots * B * S * KV * 128 * 4
    qkv = rnd(B * S, (H + 2 * KV) * hd, dtype=torch.bfloat16, seed=71)
    do = rnd(B * S, H * hd, dtype=torch.bfloat16, seed=72)
    ds = de = None
    if rows is not None:
        ds, de = (t.to(DEV) for t in _doc_arrays(rows, S))
    table = rnd(S + 8, hd // 2, 2, dtype=torch.float32, seed=73).to(DEV) if fused_rope else None
    prev = ops.set_impl(_lib.IMPL_MFMA)
    try:
        x, dout = qkv.to(DEV), do.to(DEV)
        out = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
        delta = torch.empty_like(lse)
        ops.attn_fwd(x, out, lse, B, S, H, KV, hd, ds, de)
        res = []
        for ws in (None, torch.empty(want, dtype=torch.uint8, device=DEV), torch.full((want + 64,), 255, dtype=torch.uint8, device=DEV)):
            d = torch.full_like(x, float("nan"))
            ops.attn_bwd(x, out, dout, lse, d, delta, B, S, H, KV, hd, ds, de, rope_table=table, workspace=ws)
            res.append(d.cpu().float())
    finally:
        ops.set_impl(prev)
    plain, split, split2 = res
    assert torch.isfinite(split).all() and torch.equal(split, split2), "not reproducible / workspace contents leak into the result"
    assert torch.equal(plain[:, : H * hd], split[:, : H * hd])
    a, b = plain[:, H * hd:], split[:, H * hd:]
    assert not torch.equal(a, b), "the head-split form did not run"
    assert float((a - b).norm() / a.norm()) <= 3e-4
    assert float((a - b).abs().max()) <= 2.0 ** -7 * float(a.abs().max())


@pytest.mark.parametrize("B,S,H,KV", [(8, 2048, 32, 8), (1, 4096, 32, 8), (8, 4096, 32, 8), (1, 8192, 32, 8), (1, 16384, 32, 8), (2, 2048, 32, 8),
                                      (3, 640, 4, 1)])
def test_attention_pipelined_backward_kernels_against_fp32_sdpa(ops, B, S, H, KV, attn_impl):
    """Round 5 (the round-4 review: the pipelined kernels met an fp32 reference only through whole-model tests): attn_bwd_dq2_kernel and
    attn_bwd_dkv2_kernel, FORCED (mode NEW) and the dispatch asserted, against torch SDPA in fp32 on the same bf16 inputs, with the tolerances of
    test_attention_mfma_fwd_bwd — at every kind of shape the dispatcher sends them: the headline batch; BASELINE config C's rows at B = 1 and at
    its one-GPU batch B = 8 (16 key groups, 512-entry tile tables); S = 8192; S = 16 384 (DKV2_MAX_STEPS: the tile table full to its last
    entry); the reference's default micro-batch 2 x 2048 (2 query blocks per dQ workgroup); an odd number of workgroups per pair."""
    from ssi import _lib
    hd = 64
    qkv = rnd(B * S, (H + 2 * KV) * hd, dtype=torch.bfloat16, seed=131)
    qkv[S // 2 + 3, H * hd: H * hd + hd] *= 6.0
    qkv[S - 5, :hd] *= 4.0
    do = rnd(B * S, H * hd, dtype=torch.bfloat16, seed=132)
    qr = qkv.float().clone().requires_grad_(True)
    oref = _sdpa_ref(qr, B, S, H, KV, hd)
    oref.backward(do.float())
    prev = ops.set_impl(_lib.IMPL_MFMA)
    try:
        attn_impl(_lib.ATTN_KERNEL_DQ, _lib.ATTN_MODE_NEW)
        attn_impl(_lib.ATTN_KERNEL_DKV, _lib.ATTN_MODE_NEW)
        x, dout = qkv.to(DEV), do.to(DEV)
        out = torch.full((B * S, H * hd), float("nan"), dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
        ops.attn_fwd(x, out, lse, B, S, H, KV, hd)
        res = []
        for _ in range(2):
            dqkv = torch.full_like(x, float("nan"))
            ops.attn_bwd(x, out, dout, lse, dqkv, torch.empty(B * H * S, dtype=torch.float32, device=DEV), B, S, H, KV, hd)
            used = ops.attn_last_dispatch()
            assert used & _lib.ATTN_USED_DQ2 and not used & _lib.ATTN_USED_PLAN, hex(used)
            assert bool(used & _lib.ATTN_USED_DKV2) == (S % 256 == 0), hex(used)   # (256-key groups: S = 640 keeps the 128-key kernel)
            res.append(dqkv.cpu().float())
    finally:
        ops.set_impl(prev)
    dqkv = res[0]
    assert torch.equal(res[0], res[1]), "not reproducible"
    assert torch.isfinite(dqkv).all()
    torch.testing.assert_close(out.cpu().float(), oref.detach(), rtol=2e-2, atol=2e-2)
    scale = float(qr.grad.abs().max())
    assert float((dqkv - qr.grad).abs().max()) <= 3e-2 * scale
    for name, lo, hi in (("dq", 0, H * hd), ("dk", H * hd, (H + KV) * hd), ("dv", (H + KV) * hd, (H + 2 * KV) * hd)):
        rel = float((dqkv[:, lo:hi] - qr.grad[:, lo:hi]).norm() / qr.grad[:, lo:hi].norm())
        assert rel <= 1.5e-2, (name, rel)


def _random_documents(S, seed, lo, hi):
    g = torch.Generator().manual_seed(seed)
    lens, left = [], S
    while left > 0:
        n = min(left, int(torch.randint(lo, hi + 1, (1,), generator=g)))
        lens.append(n)
        left -= n
    return lens


PLAN_CASES = [
    (2, 256, 4, 1, [[100, 37, 119], [1, 63, 64, 128]]),              # boundaries off every tile size; a 1-token document
    (1, 512, 4, 1, [[512]]),                                         # one document: the plan's items are the plain kernels' workgroups
    (1, 384, 4, 1, [[5] * 76 + [4]]),                                # tiny documents: many items per 64 rows
    (1, 2048, 8, 2, [[700, 31, 1100, 217]]),                         # BASELINE-E-like documents
    (2, 1024, 8, 2, [[256, 512, 256], [33, 31, 64, 896]]),           # documents on the 256 / 64 / 32 grids, and starting at 33
    (1, 4096, 8, 2, [_random_documents(4096, 5, 1, 700)]),
    (2, 2048, 32, 8, [_random_documents(2048, 6, 200, 900), _random_documents(2048, 7, 30, 400)]),
    (1, 11520, 32, 8, [[1807, 2038, 1909, 1924, 1500, 2048, 294]]),  # a right-padded 8 x 2048 batch after the unpadding
]


@pytest.mark.parametrize("B,S,H,KV,rows", PLAN_CASES, ids=[f"{c[0]}x{c[1]}-{len(c[4][0])}docs" for c in PLAN_CASES])
@pytest.mark.parametrize("fused_rope,split_all", [(False, False), (True, False), (True, True)], ids=["plain-epilogue", "fused-rope", "fused-rope-every-chunk-split"])
def test_attention_plan_kernels_on_packed_rows(ops, B, S, H, KV, rows, fused_rope, split_all, attn_impl):
    """Round 5: packed rows on the pipelined backward kernels (document-aware forms: attn_bwd_dq2_kernel<0, true>, attn_bwd_dkv2_kernel<true>)
    from a host-built work plan (ssi_attn_plan_build): against torch SDPA with the dense block mask in fp32, against the plan-less call (the
    round-1..3 kernels: another order of the fp32 sums), reproducible, the dispatch asserted; with the fused RoPE backward the positions are
    document-relative (the plan's assumption).  Documents must not leak: perturbing one leaves the others' gradients bit-identical.
    ``split_all``: every dK/dV chunk split over the query heads (2 x 2 or 4 x 1 heads alternating; the builder does this to chunks heavier than
    the chip's share per compute unit): fp32 partial sums + the reduction pass, and — with 1 or 2 heads per workgroup — tile counts that are
    no multiple of 4, i.e. the dummy tiles that pad a loop to whole trips."""
    from ssi import _lib, attn_plan
    hd = 64
    qkv = rnd(B * S, (H + 2 * KV) * hd, dtype=torch.bfloat16, seed=141)
    do = rnd(B * S, H * hd, dtype=torch.bfloat16, seed=142)
    ds, de = (t.to(DEV) for t in _doc_arrays(rows, S))
    plan = attn_plan.plan_from_seq_lens(rows, H, KV, force=True, split_all=split_all)
    assert plan is not None and plan.matches(B, S, H, KV)
    assert plan.workspace_bytes > 0 or not split_all
    # every (key, query head) belongs to exactly one dK/dV item, every query to exactly one dQ item
    seen_k, seen_q = torch.zeros(B, S, dtype=torch.int32), torch.zeros(B, S, dtype=torch.int32)
    for b, k0, d0, d1, _, heads, slot in plan.dkv_items(with_heads=True):
        assert (slot >= 0) == (heads < 4)
        seen_k[b, max(k0, d0):min(k0 + 256, d1)] += heads
    seen_k //= 4
    for grp in plan.dq_groups():
        for b, q0, d0, d1 in grp:
            seen_q[b, max(q0, d0):min(q0 + 64, d1)] += 1
    assert bool((seen_k == 1).all()) and bool((seen_q == 1).all())
    plan = plan.to_device(DEV)
    table = pos = None
    if fused_rope:
        table = rnd(max(max(r) for r in rows) + 3, hd // 2, 2, dtype=torch.float32, seed=143).to(DEV)   # no longer than the longest document needs
        pos = torch.cat([torch.cat([torch.arange(n) for n in lens]) for lens in rows]).to(torch.int32).to(DEV)

    def run(x, plan_):
        out = torch.full((B * S, H * hd), float("nan"), dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
        ops.attn_fwd(x, out, lse, B, S, H, KV, hd, ds, de)
        d = torch.full_like(x, float("nan"))
        ops.attn_bwd(x, out, do.to(DEV), lse, d, torch.empty(B * H * S, dtype=torch.float32, device=DEV), B, S, H, KV, hd, ds, de, rope_table=table,
                     positions=pos, plan=plan_)
        used = ops.attn_last_dispatch()
        want = _lib.ATTN_USED_DQ2 | _lib.ATTN_USED_DKV2 | _lib.ATTN_USED_PLAN
        assert (used & want) == (want if plan_ is not None else 0), hex(used)
        if plan_ is not None:
            assert bool(used & _lib.ATTN_USED_HEAD_SPLIT) == (plan_.workspace_bytes > 0), hex(used)
        return d.cpu().float()

    prev = ops.set_impl(_lib.IMPL_MFMA)
    try:
        new, new2, old = run(qkv.to(DEV), plan), run(qkv.to(DEV), plan), run(qkv.to(DEV), None)
        n0 = rows[0][0]
        if n0 < S:
            pert = qkv.clone()
            pert[:n0] = rnd(n0, qkv.shape[1], dtype=torch.bfloat16, seed=144)
            newp = run(pert.to(DEV), plan)
    finally:
        ops.set_impl(prev)
    assert torch.isfinite(new).all(), f"{int((~torch.isfinite(new)).sum())} non-finite gradients (rows {torch.nonzero(~torch.isfinite(new).all(dim=1)).flatten()[:8].tolist()})"
    assert torch.equal(new, new2), "not reproducible"
    for name, lo, hi in (("dq", 0, H * hd), ("dk", H * hd, (H + KV) * hd), ("dv", (H + KV) * hd, (H + 2 * KV) * hd)):
        a, b = old[:, lo:hi], new[:, lo:hi]
        assert float((a - b).norm() / a.norm()) <= 3e-4, name
        assert float((a - b).abs().max()) <= 2.0 ** -7 * float(a.abs().max()), f"{name}: more than a bf16 step apart"
    if n0 < S:
        assert torch.equal(newp[n0:], new[n0:]) and not torch.equal(newp[:n0], new[:n0]), "documents leak into each other"
    if not fused_rope and B * S <= 8192:
        qr = qkv.float().clone().requires_grad_(True)
        _sdpa_block_ref(qr, B, S, H, KV, hd, rows).backward(do.float())
        scale = float(qr.grad.abs().max())
        assert float((new - qr.grad).abs().max()) <= 3e-2 * scale
        assert float((new - qr.grad).norm() / qr.grad.norm()) <= 1.5e-2


@pytest.mark.parametrize("B,S,H,KV", [(2, 2048, 32, 8), (8, 512, 8, 2), (3, 640, 4, 1)])
def test_attention_plan_for_plain_causal_rows(ops, B, S, H, KV):
    """Plain causal rows (no document arrays, no positions) with a plan whose documents are the rows: the document-aware kernels on the same
    work as the plain forms — dQ bit-identical to the plain pipelined kernel's (same products, same order per query block; only the blocks are
    dealt to the workgroups by load), dK / dV equal to the rounding of the sums over the query heads where the plan splits a chunk (B = 2:
    heavy chunks, the chip's share per compute unit is small) and bit-identical where it does not; a plan with several documents per row is
    refused without document arrays."""
    from ssi import _lib, attn_plan
    hd = 64
    x = rnd(B * S, (H + 2 * KV) * hd, dtype=torch.bfloat16, seed=171).to(DEV)
    dout = rnd(B * S, H * hd, dtype=torch.bfloat16, seed=172).to(DEV)
    table = rnd(S + 8, hd // 2, 2, dtype=torch.float32, seed=173).to(DEV)
    plan = attn_plan.plan_from_seq_lens([[S]] * B, H, KV, force=True).to_device(DEV)
    prev = ops.set_impl(_lib.IMPL_MFMA)
    saved = [ops.set_attn_impl(_lib.ATTN_KERNEL_DQ, _lib.ATTN_MODE_NEW), ops.set_attn_impl(_lib.ATTN_KERNEL_DKV, _lib.ATTN_MODE_NEW)]
    try:
        out = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
        ops.attn_fwd(x, out, lse, B, S, H, KV, hd)
        res = []
        for p_ in (None, plan, plan):
            d = torch.full_like(x, float("nan"))
            ops.attn_bwd(x, out, dout, lse, d, torch.empty_like(lse), B, S, H, KV, hd, rope_table=table, plan=p_)
            assert bool(ops.attn_last_dispatch() & _lib.ATTN_USED_PLAN) == (p_ is not None)
            res.append(d.cpu().float())
        two_docs = attn_plan.plan_from_seq_lens([[S // 2, S // 2]] * B, H, KV, force=True).to_device(DEV)
        with pytest.raises(AssertionError):
            ops.attn_bwd(x, out, dout, lse, torch.empty_like(x), torch.empty_like(lse), B, S, H, KV, hd, plan=two_docs)
    finally:
        ops.set_impl(prev)
        ops.set_attn_impl(_lib.ATTN_KERNEL_DQ, saved[0]), ops.set_attn_impl(_lib.ATTN_KERNEL_DKV, saved[1])
    plain, with_plan, again = res
    assert torch.isfinite(with_plan).all() and torch.equal(with_plan, again)
    assert torch.equal(plain[:, : H * hd], with_plan[:, : H * hd]), "dQ: same products in the same order per query block"
    a, b = plain[:, H * hd:], with_plan[:, H * hd:]
    if S % 256 == 0 and plan.workspace_bytes == 0:
        assert torch.equal(a, b), "dK / dV: unsplit chunks are the plain kernel's key groups"
    assert float((a - b).norm() / a.norm()) <= 3e-4 and float((a - b).abs().max()) <= 2.0 ** -7 * float(a.abs().max())


def test_attention_plan_builder_declines_what_the_old_kernels_do_better(ops):
    """ssi_attn_plan_build returns no plan for head ratios other than 4, mostly tiny documents, documents beyond the tile table; a plan
    that belongs to another batch is refused by the launch; documents that do not tile the rows are an error."""
    from ssi import attn_plan
    assert attn_plan.plan_from_seq_lens([[700, 31, 1100, 217]], 8, 4) is None
    assert attn_plan.plan_from_seq_lens([[5] * 76 + [4]], 4, 1) is None
    two_long = attn_plan.plan_from_seq_lens([[2048], [2048]], 32, 8)   # a few long documents: their heavy chunks are split over the query heads
    assert two_long is not None and two_long.workspace_bytes > 0 and max(h for *_, h, _ in two_long.dkv_items(True)) == 4
    assert min(h for *_, h, _ in two_long.dkv_items(True)) == 1
    assert attn_plan.plan_from_seq_lens([[2048]] * 8, 32, 8) is not None
    assert attn_plan.plan_from_seq_lens([[32768]], 32, 8) is None
    assert attn_plan.plan_from_seq_lens([[1807, 2038, 1909, 1924, 1500, 2048, 294]], 32, 8) is not None
    t = lambda x: torch.tensor(x, dtype=torch.int32)  # noqa: E731
    with pytest.raises(ValueError):
        attn_plan.build_plan(t([0, 0]), t([0, 100]), t([90, 256]), 1, 256, 4, 1)     # a gap
    B, S, H, KV, hd = 1, 512, 4, 1, 64
    plan = attn_plan.plan_from_seq_lens([[200, 312]], H, KV, force=True).to_device(DEV)
    other = attn_plan.plan_from_seq_lens([[100, 156]], H, KV, force=True).to_device(DEV)
    x = rnd(B * S, (H + 2 * KV) * hd, dtype=torch.bfloat16, seed=151).to(DEV)
    ds, de = (u.to(DEV) for u in _doc_arrays([[200, 312]], S))
    out = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
    ops.attn_fwd(x, out, lse, B, S, H, KV, hd, ds, de)
    with pytest.raises(AssertionError):
        ops.attn_bwd(x, out, out, lse, torch.empty_like(x), torch.empty_like(lse), B, S, H, KV, hd, ds, de, plan=other)
    with pytest.raises(AssertionError):
        ops.attn_bwd(x, out, out, lse, torch.empty_like(x), torch.empty_like(lse), B, S, H, KV, hd, None, None, plan=plan)


def _doc_arrays(seq_lens_rows, S):
    """doc_start / doc_end (int32 [B*S]) from per-row lists of document lengths (each row sums to S)."""
    ds, de = [], []
    for lens in seq_lens_rows:
        assert sum(lens) == S
        start = 0
        for n in lens:
            ds += [start] * n
            de += [start + n] * n
            start += n
    return torch.tensor(ds, dtype=torch.int32), torch.tensor(de, dtype=torch.int32)


def _sdpa_block_ref(qkv, B, S, H, KV, hd, seq_lens_rows):
    mask = torch.stack([torch.block_diag(*[torch.tril(torch.ones(n, n, dtype=torch.bool)) for n in lens]) for lens in seq_lens_rows])
    q = qkv[:, : H * hd].view(B, S, H, hd).transpose(1, 2)
    k = qkv[:, H * hd:(H + KV) * hd].view(B, S, KV, hd).transpose(1, 2).repeat_interleave(H // KV, dim=1)
    v = qkv[:, (H + KV) * hd:].view(B, S, KV, hd).transpose(1, 2).repeat_interleave(H // KV, dim=1)
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask[:, None])
    return o.transpose(1, 2).reshape(B * S, H * hd)


@pytest.mark.parametrize("B,S,H,KV,rows", [
    (2, 256, 4, 1, [[100, 37, 119], [1, 63, 64, 128]]),                # boundaries off every tile size; a 1-token document
    (1, 512, 4, 2, [[512]]),                                           # one document = plain causal attention
    (2, 384, 2, 2, [[128, 128, 128], [5] * 76 + [4]]),                 # tile-aligned documents; many tiny documents
    (1, 2048, 8, 2, [[700, 31, 1100, 217]]),                           # BASELINE-E-like documents, 4 query heads per kv head
])
def test_attention_varlen_documents(ops, B, S, H, KV, rows):
    """Packed rows: block-causal attention over the documents of a row (MFMA and generic kernels) vs torch SDPA with the dense
    mask; documents must not leak into each other (perturbing one document leaves the others' outputs bit-identical)."""
    from ssi import _lib
    hd = 64
    qkv = rnd(B * S, (H + 2 * KV) * hd, dtype=torch.bfloat16, seed=71)
    do = rnd(B * S, H * hd, dtype=torch.bfloat16, seed=72)
    ds, de = _doc_arrays(rows, S)
    qr = qkv.float().clone().requires_grad_(True)
    oref = _sdpa_block_ref(qr, B, S, H, KV, hd, rows)
    oref.backward(do.float())

    def run(impl, x):
        prev = ops.set_impl(impl)
        try:
            out = torch.full((B * S, H * hd), float("nan"), dtype=torch.bfloat16, device=DEV)
            lse = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
            ops.attn_fwd(x.to(DEV), out, lse, B, S, H, KV, hd, ds.to(DEV), de.to(DEV))
            dqkv = torch.full_like(x, float("nan")).to(DEV)
            ops.attn_bwd(x.to(DEV), out, do.to(DEV), lse, dqkv, torch.empty(B * H * S, dtype=torch.float32, device=DEV), B, S, H, KV, hd,
                         ds.to(DEV), de.to(DEV))
            return out.cpu().float(), lse.cpu(), dqkv.cpu().float()
        finally:
            ops.set_impl(prev)

    res = {impl: run(impl, qkv) for impl in (_lib.IMPL_MFMA, _lib.IMPL_GENERIC)}
    scale = float(qr.grad.abs().max())
    for impl, (out, lse, dqkv) in res.items():
        assert torch.isfinite(out).all() and torch.isfinite(dqkv).all() and torch.isfinite(lse).all()
        torch.testing.assert_close(out, oref.detach(), rtol=2e-2, atol=2e-2)
        assert float((dqkv - qr.grad).abs().max()) <= 3e-2 * scale
        assert float((dqkv - qr.grad).norm() / qr.grad.norm()) <= 1.5e-2
    torch.testing.assert_close(res[_lib.IMPL_MFMA][1], res[_lib.IMPL_GENERIC][1], rtol=1e-4, atol=2e-3)
    if len(rows[0]) == 1 and B == 1:  # a single document: identical to the plain causal entry points, bit for bit
        prev = ops.set_impl(_lib.IMPL_MFMA)
        try:
            out = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=DEV)
            lse = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
            ops.attn_fwd(qkv.to(DEV), out, lse, B, S, H, KV, hd)
            dq = torch.empty_like(qkv).to(DEV)
            ops.attn_bwd(qkv.to(DEV), out, do.to(DEV), lse, dq, torch.empty(B * H * S, dtype=torch.float32, device=DEV), B, S, H, KV, hd)
            assert torch.equal(out.cpu().float(), res[_lib.IMPL_MFMA][0]) and torch.equal(dq.cpu().float(), res[_lib.IMPL_MFMA][2])
        finally:
            ops.set_impl(prev)
    else:  # isolation: change the first document of row 0 only
        n0 = rows[0][0]
        pert = qkv.clone()
        pert[:n0] = rnd(n0, qkv.shape[1], dtype=torch.bfloat16, seed=73)
        out2, _, dq2 = run(_lib.IMPL_MFMA, pert)
        assert torch.equal(out2[n0:], res[_lib.IMPL_MFMA][0][n0:]) and torch.equal(dq2[n0:], res[_lib.IMPL_MFMA][2][n0:])
        assert not torch.equal(out2[:n0], res[_lib.IMPL_MFMA][0][:n0])


@pytest.mark.parametrize("packed", [False, True])
def test_attention_backward_with_fused_rope_backward(ops, packed):
    """ssi_attn_varlen_bwd_rope == ssi_attn_varlen_bwd followed by ssi_rope_inplace(inverse) on the q / k heads: MFMA kernels
    (rotation in the epilogues, to within one bf16 rounding of the two-step result) and generic kernels (second launch: identical)."""
    from ssi import _lib
    from ssi.model import llama3_rope_table
    B, S, H, KV, hd = 2, 256, 4, 2, 64
    qkv = rnd(B * S, (H + 2 * KV) * hd, dtype=torch.bfloat16, seed=95).to(DEV)
    do = rnd(B * S, H * hd, dtype=torch.bfloat16, seed=96).to(DEV)
    table = llama3_rope_table(hd, 512, 500_000, 32).to(DEV)
    ds = de = pos = None
    if packed:
        rows = [[100, 37, 119], [64, 192]]
        ds, de = (t.to(DEV) for t in _doc_arrays(rows, S))
        pos = torch.cat([torch.cat([torch.arange(n) for n in lens]) for lens in rows]).to(torch.int32).to(DEV)
    for impl in (_lib.IMPL_MFMA, _lib.IMPL_GENERIC):
        prev = ops.set_impl(impl)
        try:
            out = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=DEV)
            lse = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
            ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd, ds, de)
            delta = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
            two = torch.empty_like(qkv)
            ops.attn_bwd(qkv, out, do, lse, two, delta, B, S, H, KV, hd, ds, de)
            ops.rope_(two, S, H + KV, hd, table, inverse=True, positions=pos)
            one = torch.full_like(qkv, float("nan"))
            ops.attn_bwd(qkv, out, do, lse, one, delta, B, S, H, KV, hd, ds, de, rope_table=table, positions=pos)
        finally:
            ops.set_impl(prev)
        assert torch.isfinite(one).all()
        assert torch.equal(one[:, (H + KV) * hd:], two[:, (H + KV) * hd:])        # dV is not rotated
        if impl == _lib.IMPL_GENERIC:
            assert torch.equal(one, two)
        else:
            diff = (one.float() - two.float()).abs()
            assert float(diff.max()) <= 2 ** -7 * float(two.float().abs().max()) and float((diff > 0).float().mean()) < 0.05


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("splits", [2, 3, 8])
def test_gemm_splitk_matches_direct(ops, layout, splits):
    from ssi import _lib
    M, N, K = 512, 256, 1024  # 16 K-tiles: uneven slices for splits=3
    a, b = _gemm_operands(layout, M, N, K, torch.bfloat16, 30)
    c0 = rnd(M, N, dtype=torch.bfloat16, seed=31)
    alpha_dev = torch.tensor([0.5], device=DEV)
    direct, split = c0.to(DEV), c0.to(DEV)
    ops.gemm(layout, a.to(DEV), b.to(DEV), direct, alpha=2.0, alpha_dev=alpha_dev, accumulate=True)
    need = _lib.load().ssi_gemm_splitk_workspace_bytes(M, N, splits)
    assert need == splits * M * N * 4
    ws = torch.empty(need // 4, dtype=torch.float32, device=DEV)
    ops.gemm_splitk(layout, a.to(DEV), b.to(DEV), split, splits, ws, alpha=2.0, alpha_dev=alpha_dev, accumulate=True)
    ref = (_gemm_ref(layout, a, b).float().bfloat16().double() + c0.double())
    torch.testing.assert_close(split.cpu().double(), ref, rtol=2e-2, atol=0.03 * math.sqrt(K))
    diff = (split.float() - direct.float()).abs()  # fp32 partial sums in a different order: at most 1 bf16 ulp apart
    assert float(diff.max()) <= 2 ** -6 * float(direct.float().abs().max())
    with pytest.raises(RuntimeError):
        ops.gemm_splitk(layout, a.to(DEV), b.to(DEV), split, splits, ws[:16])
    assert ops.splitk_choice(2048, 2048, 16384) > 1 and ops.splitk_choice(16384, 2048, 16384) == 1


@pytest.mark.parametrize("splits", [2, 3, 4, 8])
@pytest.mark.parametrize("accumulate", [False, True])
def test_gemm_splitk_weight_gradient_on_the_persistent_kernel(ops, splits, accumulate):
    """Weight-gradient (TN) split-K with slices long enough for the persistent kernel (units = tile x K-slice, fp32 partials in
    the accumulator layout, nt4_splitk_reduce_kernel): exact on small integers whatever the slicing; close to the unsplit GEMM."""
    M, N, K = 768, 512, 4096   # 6 tiles x splits units; 64 K-steps: slices of 32 / 20-22-22 / 16 / 8
    a, b = _gemm_operands(2, M, N, K, torch.bfloat16, 91, integer=True)
    a, b = a.to(DEV), b.to(DEV)
    c0 = torch.randint(-3, 4, (M, N), generator=torch.Generator().manual_seed(92)).to(torch.bfloat16).to(DEV)
    ws = torch.empty(splits * M * N, dtype=torch.float32, device=DEV)
    c = c0.clone() if accumulate else torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.gemm_splitk(2, a, b, c, splits, ws, alpha=0.125, accumulate=accumulate)
    ref = (a.float().t() @ b.float()) * 0.125
    exact = ref.abs() <= 256
    want = ref.bfloat16().float() + (c0.float() if accumulate else 0)
    assert torch.equal(c.float()[exact], want.bfloat16().float()[exact])
    a2, b2 = _gemm_operands(2, M, N, K, torch.bfloat16, 93)
    direct = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    split = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(2, a2.to(DEV), b2.to(DEV), direct)
    ops.gemm_splitk(2, a2.to(DEV), b2.to(DEV), split, splits, ws)
    diff = (split.float() - direct.float()).abs()
    assert float(diff.max()) <= 2 ** -6 * float(direct.float().abs().max())


@pytest.mark.parametrize("accumulate", [False, True])
def test_gemm_batched_weight_gradients_equal_one_call_per_problem(ops, accumulate):
    """ssi_gemm_batched (TN, persistent kernel): `batch` problems in one launch, C a strided view (the model's flat gradient buffer with
    other weights between the problems).  Same K loop per tile as ssi_gemm -> bit-identical to one ssi_gemm per problem; exact on integers."""
    n, M, N, K = 5, 512, 768, 1024         # 5 x 6 tiles: problems end inside the XCD spans and inside a round
    g = torch.Generator().manual_seed(77)
    a = torch.randn(n, K, M, generator=g).bfloat16().to(DEV)
    b = torch.randn(n, K, N, generator=g).bfloat16().to(DEV)
    gap = 1000 * 8                         # elements between the problems' outputs (other parameters in the flat buffer)
    flat0 = torch.randn(n * (M * N + gap), generator=g).bfloat16().to(DEV)
    flat1, flat2 = flat0.clone(), flat0.clone()
    view = lambda f: torch.as_strided(f, (n, M, N), (M * N + gap, N, 1), 0)  # noqa: E731
    alpha = torch.tensor([0.25], device=DEV)
    ops.gemm_batched(2, a, b, view(flat1), alpha_dev=alpha, accumulate=accumulate)
    for i in range(n):
        ops.gemm(2, a[i], b[i], view(flat2)[i], alpha_dev=alpha, accumulate=accumulate)
    assert torch.equal(flat1, flat2)                                            # results AND the gaps (nothing written outside the views)
    ref = torch.bmm(a.float().transpose(1, 2), b.float()) * 0.25
    want = ref.bfloat16().float() + (view(flat0).float() if accumulate else 0)
    torch.testing.assert_close(view(flat1).float(), want.bfloat16().float(), rtol=2e-2, atol=0.05 * math.sqrt(K))
    ai = torch.randint(-2, 3, (n, K, M), generator=g).bfloat16().to(DEV)
    bi = torch.randint(-2, 3, (n, K, N), generator=g).bfloat16().to(DEV)
    ci = torch.full((n, M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.gemm_batched(2, ai, bi, ci, alpha=2.0 ** -4)
    refi = torch.bmm(ai.float().transpose(1, 2), bi.float()) * 2.0 ** -4
    exact = refi.abs() <= 16
    assert float(exact.float().mean()) > 0.9 and torch.equal(ci.float()[exact], refi[exact])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_gemm_batched_other_shapes_run_problem_by_problem(ops, dtype, layout):
    """Layouts / dtypes / shapes the batched MFMA form does not take (fp32, NT / NN, ragged sizes, batch 1 and 0): a loop over ssi_gemm."""
    n, M, N, K = 3, 70, 130, 33
    shp_a = (n, M, K) if layout < 2 else (n, K, M)
    shp_b = (n, N, K) if layout == 0 else (n, K, N)
    a, b = rnd(*shp_a, dtype=dtype, seed=5).to(DEV), rnd(*shp_b, dtype=dtype, seed=6).to(DEV)
    c = torch.full((n, M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm_batched(layout, a, b, c)
    af, bf = a.float().cpu(), b.float().cpu()
    A = af if layout < 2 else af.transpose(1, 2)
    Bm = bf.transpose(1, 2) if layout == 0 else bf
    torch.testing.assert_close(c.float().cpu(), torch.bmm(A, Bm), **(dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=0.1)))
    one = torch.full((1, M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm_batched(layout, a[:1], b[:1], one)
    assert torch.equal(one[0], c[0])
    ops.gemm_batched(layout, a[:0], b[:0], c[:0])   # empty batch: nothing to do


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,cols", [(64, 64), (200, 136), (2048, 3072)])
def test_transpose(ops, dtype, rows, cols):
    src = rnd(rows, cols, dtype=dtype, seed=32)
    dst = torch.empty(cols, rows, dtype=dtype, device=DEV)
    ops.transpose(src.to(DEV), dst)
    assert torch.equal(dst.cpu(), src.t().contiguous())  # bit-exact data movement


def test_fused_swiglu_gemms_equal_the_unfused_kernels(ops):
    """SwiGLU in the GEMM epilogue (MFMA path) must reproduce GEMM -> swiglu kernel bit for bit: same rounding points."""
    from ssi import _lib
    M, I, K = 512, 512, 256
    x = rnd(M, K, dtype=torch.bfloat16, seed=40).to(DEV)
    w13 = rnd(2 * I, K, dtype=torch.bfloat16, seed=41, scale=0.1).to(DEV)
    w2t = rnd(I, K, dtype=torch.bfloat16, seed=42, scale=0.1).to(DEV)   # [I, K] = transposed copy of W2 [K, I]
    dy = rnd(M, K, dtype=torch.bfloat16, seed=43).to(DEV)
    # unfused reference through the same library
    gu_ref = torch.empty(M, 2 * I, dtype=torch.bfloat16, device=DEV)
    act_ref = torch.empty(M, I, dtype=torch.bfloat16, device=DEV)
    ops.gemm(ops.GEMM_NT, x, w13, gu_ref)
    ops.swiglu_fwd(gu_ref, act_ref)
    dact = torch.empty(M, I, dtype=torch.bfloat16, device=DEV)
    ops.gemm(ops.GEMM_NT, dy, w2t, dact)
    dgu_ref = torch.empty_like(gu_ref)
    ops.swiglu_bwd(dact, gu_ref, dgu_ref)
    # fused
    gu = torch.full_like(gu_ref, float("nan"))
    act = torch.full_like(act_ref, float("nan"))
    ops.gemm_swiglu_fwd(x, w13, gu, act)
    assert torch.equal(gu, gu_ref) and torch.equal(act, act_ref)
    dgu = torch.full_like(gu_ref, float("nan"))
    ops.gemm_swiglu_bwd(ops.GEMM_NT, dy, w2t, gu_ref, dgu, None)
    assert torch.equal(dgu, dgu_ref)
    dgu_nn = torch.full_like(gu_ref, float("nan"))   # same product against the untransposed weight [K, I] (NN form), also fused
    ops.gemm_swiglu_bwd(ops.GEMM_NN, dy, w2t.t().contiguous(), gu_ref, dgu_nn, None)
    assert torch.equal(dgu_nn, dgu_ref)
    # against fp32 math on the CPU
    xr, wr = x.float().cpu(), w13.float().cpu()
    g = (xr @ wr.T).bfloat16().float()
    ref_act = (F.silu(g[:, :I]).bfloat16().float() * g[:, I:]).bfloat16().float()
    torch.testing.assert_close(act.float().cpu(), ref_act, rtol=2e-2, atol=2e-2)
    # unfused fallback (generic implementation, NN layout with a scratch buffer) agrees within bf16 rounding
    prev = ops.set_impl(_lib.IMPL_GENERIC)
    try:
        dgu2 = torch.empty_like(gu_ref)
        ops.gemm_swiglu_bwd(ops.GEMM_NN, dy, w2t.t().contiguous(), gu_ref, dgu2, torch.empty(M, I, dtype=torch.bfloat16, device=DEV))
        diff = (dgu2.float() - dgu_ref.float()).abs()
        assert float(diff.max()) <= 2 ** -6 * float(dgu_ref.float().abs().max()) + 1e-3
    finally:
        ops.set_impl(prev)


@pytest.mark.parametrize("explicit_positions", [False, True])
def test_qkv_gemm_with_rope_in_the_epilogue_equals_gemm_then_rope(ops, explicit_positions):
    """ssi_gemm_rope (persistent MFMA kernel, rotation in the epilogue) == ssi_gemm + ssi_rope_inplace bit for bit: the rotation
    is applied to the bf16-rounded projection in both.  Three m-tiles x three n-tiles, the last n-tile (v heads) not rotated."""
    from ssi import _lib
    from ssi.model import llama3_rope_table
    B, S, H, KV, hd, K = 3, 256, 6, 2, 64, 256       # N = (6 + 2*2) * 64 = 640 -> padded weight rows: use H + KV = 8 heads = 512 rotated
    N = 768
    table = llama3_rope_table(hd, 4096).to(DEV)
    x = rnd(B * S, K, dtype=torch.bfloat16, seed=50).to(DEV)
    w = rnd(N, K, dtype=torch.bfloat16, seed=51, scale=0.1).to(DEV)
    pos = None
    if explicit_positions:   # packed rows: positions restart inside a row and reach beyond S
        pos = torch.cat([torch.arange(100), torch.arange(3000, 3000 + 412), torch.arange(B * S - 512)]).to(torch.int32).to(DEV)
    ref = torch.empty(B * S, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(ops.GEMM_NT, x, w, ref)
    plain = ref.clone()
    ops.rope_(ref, S, 8, hd, table, positions=pos)
    out = torch.full_like(ref, float("nan"))
    ops.gemm_rope(x, w, out, S, 8, hd, table, positions=pos)
    assert torch.equal(out, ref)
    assert torch.equal(out[:, 512:], plain[:, 512:]) and not torch.equal(out[:, :512], plain[:, :512])
    # shapes the fused kernel does not take (rotated width not a multiple of 256) and the generic implementation: two launches, same result
    out2 = torch.full_like(ref, float("nan"))
    ops.gemm_rope(x, w, out2, S, 7, hd, table, positions=pos)
    ref2 = plain.clone()
    ops.rope_(ref2, S, 7, hd, table, positions=pos)
    assert torch.equal(out2, ref2)
    prev = ops.set_impl(_lib.IMPL_GENERIC)
    try:
        out3 = torch.full_like(ref, float("nan"))
        ops.gemm_rope(x, w, out3, S, 8, hd, table, positions=pos)
    finally:
        ops.set_impl(prev)
    torch.testing.assert_close(out3.float(), ref.float(), rtol=2e-2, atol=2e-2)


def test_doc_ranges_kernel_matches_the_torch_restatement(ops):
    """Packed rows: positions / doc_start / doc_end from input_pos in one launch — bit-exact against the torch ops the model used
    before, on rows with 1-token documents, a document crossing every 256-thread span boundary, S not a multiple of anything."""
    from ssi.model import HipLlamaDecoder
    g = torch.Generator().manual_seed(130)
    for B, S in ((1, 1), (2, 37), (3, 300), (2, 8192), (1, 5000)):
        rows = []
        for _ in range(B):
            lens, left = [], S
            while left > 0:
                n = min(left, int(torch.randint(1, max(2, S // 3), (1,), generator=g)))
                lens.append(n)
                left -= n
            rows.append(torch.cat([torch.arange(n) for n in lens]))
        ip = torch.stack(rows).to(torch.int64)
        ip[0, 0] = 5  # a row whose first position is not 0 still starts a document there
        pos, ds, de = ops.doc_ranges(ip.to(DEV), 4095)
        rpos, rds, rde = HipLlamaDecoder._document_ranges(ip, 4095)
        assert torch.equal(pos.cpu(), rpos) and torch.equal(ds.cpu(), rds) and torch.equal(de.cpu(), rde), (B, S)
    big = torch.arange(6000).reshape(1, 6000)
    pos, _, _ = ops.doc_ranges(big.to(DEV), 4095)
    assert int(pos.max()) == 4095   # clamped to the RoPE table
    # ... and counted (ABI v5): 6000 - 4096 positions beyond the table, plus one negative one; the counter is ADDED to
    n_clamped = torch.full((1,), 7, dtype=torch.int32, device=DEV)
    big[0, 3] = -2
    pos, _, _ = ops.doc_ranges(big.to(DEV), 4095, n_clamped)
    assert int(n_clamped) == 7 + (6000 - 4096) + 1 and int(pos.min()) == 0
    n_clamped.zero_()
    ops.doc_ranges(torch.arange(300).reshape(1, 300).to(DEV), 4095, n_clamped)
    assert int(n_clamped) == 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_lmhead_ce_one_call_entries_equal_the_launch_sequence(ops, dtype):
    """ssi_lmhead_ce_fwd / ssi_lmhead_ce_bwd (SURVEY.md §8b's export list) == head GEMM -> ssi_ce_fwd -> ssi_ce_reduce and the two
    gradient GEMMs, bit for bit; and the loss against fp32 torch on the CPU."""
    rows, dim, vocab, vpad = 256, 256, 700, 768
    hn = rnd(rows, dim, dtype=dtype, seed=140)
    table = rnd(vpad, dim, dtype=dtype, seed=141, scale=0.2)
    table[vocab:] = 0
    labels = torch.randint(0, vocab, (rows,), generator=torch.Generator().manual_seed(142))
    labels[::9] = -100
    h, e, lab = hn.to(DEV), table.to(DEV), labels.to(DEV)
    # the sequence
    logits = torch.empty(rows, vpad, dtype=dtype, device=DEV)
    ops.gemm(ops.GEMM_NT, h, e, logits)
    row_loss = torch.empty(rows, dtype=torch.float32, device=DEV)
    ops.ce_fwd(logits, lab, vocab, -100, row_loss, None, True)
    stats = torch.empty(4, dtype=torch.float32, device=DEV)
    ops.ce_reduce(row_loss, lab, vocab, -100, stats)
    alpha = torch.tensor([0.37], dtype=torch.float32, device=DEV)
    d_h = torch.empty_like(h)
    d_e = rnd(vpad, dim, dtype=dtype, seed=143).to(DEV)
    d_e0 = d_e.clone()
    ops.gemm(ops.GEMM_NN, logits, e, d_h, alpha_dev=alpha)
    ops.gemm(ops.GEMM_TN, logits, h, d_e, alpha_dev=alpha, accumulate=True)
    # one call each
    ws = torch.full((rows, vpad), float("nan"), dtype=dtype, device=DEV)
    rl2 = torch.empty_like(row_loss)
    st2 = torch.empty_like(stats)
    ops.lmhead_ce_fwd(h, e, lab, vocab, -100, ws, rl2, st2, True)
    assert torch.equal(ws, logits) and torch.equal(rl2, row_loss) and torch.equal(st2, stats)
    d_h2 = torch.full_like(h, float("nan"))
    d_e2 = d_e0.clone()
    ops.lmhead_ce_bwd(ws, h, e, alpha, d_h2, d_e2, True)
    assert torch.equal(d_h2, d_h) and torch.equal(d_e2, d_e)
    ref = F.cross_entropy((hn.float() @ table.float().t())[:, :vocab].to(dtype).float(), labels, ignore_index=-100, reduction="sum")
    n_valid = int((labels != -100).sum())
    assert st2.cpu()[2].item() == n_valid and st2.cpu()[3].item() == 0
    assert st2.cpu()[1].item() == pytest.approx(float(ref), rel=1e-5 if dtype == torch.float32 else 2e-3)
