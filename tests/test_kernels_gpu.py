"""Parity of every non-GEMM HIP kernel (through the C ABI) against plain torch fp32 references /
the oracle.  Tolerances: fp32 kernels 1e-5..1e-4 relative to the tensor scale; bf16-output kernels
one bf16 ulp (8e-3)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from scl_amd import ops  # noqa: E402
from scl_amd.lib import SclError  # noqa: E402


def rel(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("C,M", [(512, 130), (1024, 67), (32, 50), (64, 9)])
@pytest.mark.parametrize("xdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", [0, 1])
def test_layernorm_fwd_bwd(dev, C, M, xdt, act):
    x = torch.randn(M, C, generator=g(1)).to(xdt).to(dev)
    gamma = (1 + 0.1 * torch.randn(C, generator=g(2))).to(dev)
    beta = (0.1 * torch.randn(C, generator=g(3))).to(dev)
    y_bf = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    y_f = torch.empty(M, C, dtype=torch.float32, device=dev)
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    ops.layernorm_fwd(x, gamma, beta, y_bf, y_f, mean, rstd, M, C, act=act)
    xr = x.float().clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-5)
    if act:
        ref = F.gelu(ref)
    assert rel(y_f, ref) < 2e-5
    assert rel(y_bf, ref) < 8e-3
    assert rel(mean, x.float().mean(1)) < 1e-5
    dy = torch.randn(M, C, generator=g(4)).to(dev)
    dres = torch.randn(M, C, generator=g(5)).to(dev)
    ref.backward(dy)
    nparts = ops.layernorm_bwd_nparts(M)
    part = torch.empty(nparts, 2 * C, device=dev)
    dx_f = torch.empty(M, C, device=dev); dx_b = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    ops.layernorm_bwd(dy, x, mean, rstd, gamma, beta, dres, dx_f, dx_b, part, M, C, act=act)
    both = torch.empty(2 * C, device=dev)
    ops.colreduce(part, both, nparts, 2 * C)
    dgam, dbet = both[:C], both[C:]
    assert rel(dx_f, xr.grad + dres) < 5e-5
    assert rel(dx_b, xr.grad + dres) < 8e-3
    assert rel(dgam, gr.grad) < 5e-5 and rel(dbet, br.grad) < 5e-5
    # third partial row: column sums of the residual gradient (bias gradient of the linear feeding the residual), split output
    part3 = torch.empty(nparts, 3 * C, device=dev)
    dx_f3 = torch.empty(M, C, device=dev)
    ops.layernorm_bwd(dy, x, mean, rstd, gamma, beta, dres, dx_f3, None, part3, M, C, act=act, sum_dres=True)
    assert torch.equal(dx_f3, dx_f)
    lnout = torch.full((2 * C,), float("nan"), device=dev); bias_out = torch.full((C,), float("nan"), device=dev)
    ops.colreduce_seg(part3, lnout, nparts, 3 * C, out2=bias_out, split=2 * C)
    assert rel(lnout, both) < 1e-5 and rel(bias_out, dres.sum(0)) < 2e-5
    # sum mode 2: column sums of the OUTPUT (conv bias gradient), padded bf16 output rows (3 rows per "utterance" of rpb frames)
    rpb = 5
    if M % rpb == 0:
        padded = torch.zeros((M // rpb) * (rpb + 3), C, dtype=torch.bfloat16, device=dev)
        ops.layernorm_bwd(dy, x, mean, rstd, gamma, beta, None, None, padded, part3, M, C, act=act, sum_dres=2,
                          out_rpb=rpb, out_rbstride=(rpb + 3) * C, out_off=2 * C)
        ops.colreduce_seg(part3, lnout, nparts, 3 * C, out2=bias_out, split=2 * C)
        pv = padded.view(M // rpb, rpb + 3, C)
        assert rel(pv[:, 2:2 + rpb].reshape(M, C), xr.grad) < 8e-3 and pv[:, :2].abs().max() == 0 and pv[:, 2 + rpb:].abs().max() == 0
        assert rel(bias_out, xr.grad.sum(0)) < 2e-4
    # bf16 dy variant
    dyb = dy.to(torch.bfloat16)
    ops.layernorm_bwd(dyb, x, mean, rstd, gamma, beta, None, dx_f, None, part, M, C, act=act)
    xr.grad = None
    ref2 = F.layer_norm(xr, (C,), gamma, beta, 1e-5)
    if act:
        ref2 = F.gelu(ref2)
    ref2.backward(dyb.float())
    assert rel(dx_f, xr.grad) < 5e-5


def test_colsum_and_cast_and_add(dev):
    M, N = 1000, 264
    x = torch.randn(M, N, generator=g(1)).to(dev)
    for t in (x, x.to(torch.bfloat16)):
        n = ops.colsum_nparts(M)
        part = torch.empty(n, N, device=dev); out = torch.empty(N, device=dev)
        ops.colsum(t, part, M, N)
        ops.colreduce(part, out, n, N)
        assert rel(out, t.float().sum(0)) < 1e-5
    ops.colreduce(part, out, n, N, accumulate=True)
    assert rel(out, 2 * x.to(torch.bfloat16).float().sum(0)) < 1e-5
    d = torch.empty(M * N + 3, dtype=torch.bfloat16, device=dev)
    src = torch.randn(M * N + 3, generator=g(2)).to(dev)
    ops.cast_bf16(src, d)
    assert torch.equal(d, src.to(torch.bfloat16))
    a, b = torch.randn(999, generator=g(3)).to(dev), torch.randn(999, generator=g(4)).to(dev)
    o = torch.empty(999, device=dev); ob = torch.empty(999, dtype=torch.bfloat16, device=dev)
    ops.add_f32(a, b, o, ob, 999)
    assert torch.equal(o, a + b) and torch.equal(ob, (a + b).to(torch.bfloat16))


@pytest.mark.parametrize("M,N", [(1000, 264), (6368, 4096), (300, 8), (70000, 512)])
def test_self_finishing_reductions(dev, M, N):
    """colsum_reduce / colreduce_seg finish their sum inside the launch (ticket + agent-scope release/acquire): same numbers as
    the two-launch path bit for bit, repeatable (the counters re-arm themselves), deterministic."""
    x = torch.randn(M, N, generator=g(5)).to(dev).to(torch.bfloat16)
    n = ops.colsum_nparts(M)
    part = torch.empty(n, N, device=dev); ref = torch.empty(N, device=dev)
    ops.colsum(x, part, M, N)
    ops.colreduce(part, ref, n, N)
    for _ in range(4):
        part2 = torch.full((n, N), float("nan"), device=dev); out = torch.full((N,), float("nan"), device=dev)
        ops.colsum_reduce(x, part2, out, M, N)
        assert torch.equal(part2, part)
        assert rel(out, x.float().sum(0)) < 1e-5
        first = out.clone() if _ == 0 else first
        assert torch.equal(out, first)
    # LayerNorm-style partial rows [nparts, C]
    nparts, C = 768, min(N, 2048)
    pr = torch.randn(nparts, C, generator=g(6)).to(dev)
    want = torch.empty(C, device=dev)
    ops.colreduce(pr, want, nparts, C)
    for _ in range(3):
        got = torch.full((C,), float("nan"), device=dev)
        ops.colreduce_seg(pr, got, nparts, C)
        assert rel(got, pr.double().sum(0).float()) < 1e-5 and rel(got, want) < 1e-5
    ops.colreduce_seg(pr, got, nparts, C, accumulate=True)
    assert rel(got, 2 * pr.double().sum(0).float()) < 1e-5
    small = torch.randn(5, C, generator=g(7)).to(dev)           # fewer than 64 partial rows: single segment
    ops.colreduce_seg(small, got, 5, C)
    assert rel(got, small.sum(0)) < 1e-5


def test_pad_rows_col2im_weight_packs(dev):
    B, T, C = 3, 21, 32
    x = torch.randn(B, T, C, generator=g(1)).to(dev)
    dst = torch.full((B, T + 10, C), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.pad_rows(x, dst, B, T, C, T + 10, 4)
    ref = torch.zeros(B, T + 10, C, device=dev); ref[:, 4:4 + T] = x
    assert torch.equal(dst, ref.to(torch.bfloat16))
    pre = torch.randn(B, T, C, generator=g(2)).to(torch.bfloat16).to(dev)
    ops.pad_rows(x, dst, B, T, C, T + 10, 3, pre=pre, ract=1)
    pr = pre.float().requires_grad_(True); F.gelu(pr).sum().backward()
    ref = torch.zeros(B, T + 10, C, device=dev); ref[:, 3:3 + T] = x * pr.grad
    assert rel(dst, ref) < 8e-3
    # col2im == conv1d backward-data given dcol = dy @ Wflat
    for (k, s, Tin) in ((3, 2, 41), (2, 2, 40), (2, 2, 41), (3, 2, 42)):
        Tout = (Tin - k) // s + 1
        dcol = torch.randn(B, Tout, k, C, generator=g(3)).to(torch.bfloat16).to(dev)
        dz = torch.full((B, Tin, C), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.col2im(dcol, dz, B, Tin, Tout, C, k, s)
        ref = torch.zeros(B, Tin, C, device=dev)
        for t in range(Tout):
            for j in range(k):
                ref[:, s * t + j] += dcol[:, t, j].float()
        assert rel(dz, ref) < 8e-3
    Co, Ci, k = 16, 8, 3
    w = torch.randn(Co, Ci, k, generator=g(4)).to(dev)
    wk = torch.empty(Co, k * Ci, dtype=torch.bfloat16, device=dev)
    ops.conv_weight_pack(w, wk, Co, Ci, k)
    assert torch.equal(wk, w.permute(0, 2, 1).reshape(Co, k * Ci).to(torch.bfloat16))
    dwk = torch.randn(Co, k * Ci, generator=g(5)).to(dev); dw = torch.empty(Co, Ci, k, device=dev)
    ops.conv_weight_unpack_grad(dwk, dw, Co, Ci, k)
    assert torch.equal(dw, dwk.view(Co, k, Ci).permute(0, 2, 1))


@pytest.mark.parametrize("E,G,K", [(32, 4, 8), (64, 4, 16), (1024, 16, 128)])
def test_posconv_weight_norm_pack_and_bwd(dev, E, G, K):
    Cg = E // G
    v = torch.randn(E, Cg, K, generator=g(1)).to(dev)
    gg = (1 + 0.3 * torch.randn(1, 1, K, generator=g(2))).to(dev)
    norm = torch.empty(K, device=dev)
    wf = torch.empty(G, Cg, K * Cg, dtype=torch.bfloat16, device=dev); wd = torch.empty_like(wf)
    ops.posconv_weight_pack(v, gg, norm, wf, wd, E, Cg, K)
    vr, gr = v.clone().requires_grad_(True), gg.clone().requires_grad_(True)
    w = gr * vr / vr.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    assert rel(norm, v.pow(2).sum(dim=(0, 1)).sqrt()) < 1e-5
    ref_wf = w.view(G, Cg, Cg, K).permute(0, 1, 3, 2).reshape(G, Cg, K * Cg)          # [g][co][j][ci]
    ref_wd = w.view(G, Cg, Cg, K).flip(-1).permute(0, 2, 3, 1).reshape(G, Cg, K * Cg)  # [g][ci][j'][co]
    assert rel(wf, ref_wf) < 8e-3 and rel(wd, ref_wd) < 8e-3
    dwf = torch.randn(G, Cg, K * Cg, generator=g(3)).to(dev)
    dw_torch = dwf.view(G, Cg, K, Cg).permute(0, 1, 3, 2).reshape(E, Cg, K)
    w.backward(dw_torch)
    dv = torch.empty_like(v); dg = torch.empty(K, device=dev); ws = torch.empty(K + E * K, device=dev)
    ops.posconv_weight_bwd(dwf, v, gg, norm, ws, dv, dg, E, Cg, K)
    assert rel(dv, vr.grad) < 1e-4 and rel(dg, gr.grad.view(-1)) < 1e-4


def test_head_tail(dev):
    B, T, C, NC = 5, 49, 128, 2
    h = torch.randn(B, T, C, generator=g(1)).to(torch.bfloat16).to(dev)
    emb = torch.empty(B, C, device=dev)
    ops.meanpool_fwd(h, emb, B, T, C)
    assert rel(emb, h.float().mean(1)) < 1e-5
    W = (0.1 * torch.randn(NC, C, generator=g(2))).to(dev); b = torch.randn(NC, generator=g(3)).to(dev)
    logp = torch.empty(B, NC, device=dev)
    ops.utt_head_fwd(emb, W, b, logp, B, C, NC)
    er, Wr, br = emb.clone().requires_grad_(True), W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.log_softmax(F.linear(er, Wr, br), 1)
    assert rel(logp, ref) < 1e-5
    dlogp = torch.randn(B, NC, generator=g(4)).to(dev); demb_in = torch.randn(B, C, generator=g(5)).to(dev)
    ref.backward(dlogp)
    demb = torch.empty(B, C, device=dev); dW = torch.empty(NC, C, device=dev); db = torch.empty(NC, device=dev)
    ws = torch.empty(B * NC, device=dev)
    ops.utt_head_bwd(dlogp, logp, emb, W, demb_in, demb, dW, db, ws, B, C, NC)
    assert rel(demb, er.grad + demb_in) < 1e-5 and rel(dW, Wr.grad) < 1e-5 and rel(db, br.grad) < 1e-5
    pre = torch.randn(B, T, C, generator=g(6)).to(torch.bfloat16).to(dev)
    dpre = torch.empty(B, T, C, dtype=torch.bfloat16, device=dev)
    ops.meanpool_bwd(demb, pre, dpre, B, T, C, 3)
    refd = (demb[:, None, :] / T) * torch.where(pre.float() > 0, 1.0, 0.01)
    assert rel(dpre, refd) < 8e-3
    ops.meanpool_bwd(demb, pre, dpre, B, T, C, 3, drop_p=0.5, seed=7)
    ratio = (dpre.float() / refd)
    vals = ratio[refd.abs() > 1e-3].round().unique().tolist()
    assert set(vals) <= {0.0, 2.0}


def test_softmax_fwd_bwd(dev):
    R, T, Tp = 300, 199, 208
    S = (3 * torch.randn(R, T, generator=g(1))).to(dev)
    P = torch.full((R, Tp), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.softmax_fwd(S, P, R, T, T, Tp)
    Sr = S.clone().requires_grad_(True)
    ref = torch.softmax(Sr, -1)
    assert rel(P[:, :T], ref) < 8e-3 and (P[:, T:] == 0).all()
    dP = torch.randn(R, T, generator=g(2)).to(dev)
    dS = torch.full((R, Tp), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.softmax_bwd(P, dP, dS, R, T, T, Tp)
    pf = P[:, :T].float()
    refd = pf * (dP - (dP * pf).sum(-1, keepdim=True))
    assert rel(dS[:, :T], refd) < 8e-3 and (dS[:, T:] == 0).all()


@pytest.mark.parametrize("C,L,B,saved", [(512, 16000, 2, True), (32, 4000, 3, False), (512, 64000, 3, True), (64, 2000, 40, False)])
def test_conv0_fwd_bwd(dev, C, L, B, saved):
    k, s = 10, 5
    x = (0.5 * torch.randn(B, L, generator=g(1))).to(dev)
    w = (torch.randn(C, 1, k, generator=g(2)) * (2.0 / k) ** 0.5).to(dev)
    b = (0.1 * torch.randn(C, generator=g(3))).to(dev)
    gamma = (1 + 0.1 * torch.randn(C, generator=g(4))).to(dev); beta = (0.1 * torch.randn(C, generator=g(5))).to(dev)
    T0 = (L - k) // s + 1
    z = torch.empty(B, T0, C, dtype=torch.bfloat16, device=dev)
    stats = torch.full((B * T0, 2), float("nan"), device=dev) if saved else None   # forward LayerNorm (mean, rstd) kept for the backward
    ops.conv0_fwd(x, w, b, gamma, beta, z, B, L, C, k, s, stats=stats)
    wr, br_, gr, ber = (t.clone().requires_grad_(True) for t in (w, b, gamma, beta))
    ref = F.gelu(F.layer_norm(F.conv1d(x[:, None], wr, br_, stride=s).transpose(1, 2), (C,), gr, ber, 1e-5))
    assert rel(z, ref) < 8e-3
    dz = (torch.randn(B, T0, C, generator=g(6))).to(torch.bfloat16).to(dev)
    ref.backward(dz.float())
    nparts = ops.conv0_bwd_nparts(B, L, k, s)
    ws = torch.empty(nparts * C * (k + 3), device=dev)
    dW = torch.empty(C, 1, k, device=dev); db = torch.empty(C, device=dev); dg = torch.empty(C, device=dev); dbe = torch.empty(C, device=dev)
    ops.conv0_bwd(x, w, b, gamma, beta, dz, ws, dW, db, dg, dbe, B, L, C, k, s, stats=stats)
    if saved:
        y = F.conv1d(x[:, None], w, b, stride=s).transpose(1, 2).reshape(B * T0, C)
        assert rel(stats[:, 0], y.mean(1)) < 1e-4 and rel(stats[:, 1], (y.var(1, unbiased=False) + 1e-5).rsqrt()) < 1e-4
    assert rel(dW, wr.grad) < 2e-4 and rel(db, br_.grad) < 2e-4 and rel(dg, gr.grad) < 2e-4 and rel(dbe, ber.grad) < 2e-4


def test_supcon_batches_beyond_128_match_oracle(dev):
    """The reference has no batch limit (nn.DataParallel hands Model.loss the gathered batch of all GPUs: 8 x 64 = 512 rows at
    BASELINE configs[2] / [3]; a 12-pack step is 132 utterances).  Above 128 utterances the loss takes a wave per row with S in global
    memory; GEMM form (bz, K multiples of 4) and the scalar fall-backs (ragged bz / K) against supcon_loss (loss_metrics.py:85-209)."""
    from oracle import head as OH
    for bz, T2, d, seed in ((192, 49, 128, 0), (512, 199, 128, 1), (132, 199, 128, 2), (130, 7, 16, 3), (257, 5, 3, 4), (512, 128, 1, 5)):
        torch.manual_seed(seed)
        f = torch.randn(bz, 1, T2, d)
        lab = torch.tensor(([1] * ((5 * bz + 10) // 11) + [0] * bz)[:bz])
        fr = f.clone().requires_grad_(True)
        ref = OH.supcon_loss(fr, lab)
        (ref * 0.37).backward()
        K = T2 * d
        Fd = f.view(bz, K).to(dev).contiguous()
        ws = torch.empty(ops.supcon_ws_floats(bz, K), device=dev)
        G = torch.empty(2 * bz * bz, device=dev); loss = torch.empty(1, device=dev)
        S = torch.empty(bz, bz, device=dev)
        ops.supcon_fwd(Fd, lab.to(dev), bz, K, K, T2, 0.07, ws, G, loss, S_out=S)
        Sref = (f.view(bz, K).double() @ f.view(bz, K).double().t()) / (T2 * 0.07)
        assert rel(S, Sref.float()) < 1e-5, (bz, T2, d)
        assert abs(loss.item() - ref.item()) <= 5e-5 * max(1.0, abs(ref.item())), (bz, T2, d, loss.item(), ref.item())
        loss2 = torch.empty(1, device=dev); G2 = torch.empty(2 * bz * bz, device=dev)
        ops.supcon_fwd(Fd, lab.to(dev), bz, K, K, T2, 0.07, ws, G2, loss2)          # S parked in the G buffer's scratch half
        assert torch.equal(loss2, loss) and torch.equal(G2[: bz * bz], G[: bz * bz])
        dF = torch.empty(bz, K, device=dev)
        up = torch.tensor([0.37], device=dev)
        ops.supcon_bwd(Fd, G2, up, 1.0, bz, K, K, T2, 0.07, dF)
        assert rel(dF, fr.grad.view(bz, K)) < 3e-4, (bz, T2, d, rel(dF, fr.grad.view(bz, K)))
    with pytest.raises(SclError):
        bz = 1028
        ops.supcon_fwd(torch.zeros(bz, 64, device=dev), torch.zeros(bz, dtype=torch.int64, device=dev), bz, 64, 64, 8, 0.07,
                       torch.empty(ops.supcon_nchunks(64) * bz * bz + bz, device=dev), torch.empty(2 * bz * bz, device=dev), torch.empty(1, device=dev))


def test_supcon_and_nll_match_oracle(dev):
    from oracle import head as OH
    for bz, T2, d, seed in ((4, 49, 128, 0), (11, 199, 128, 1), (64, 49, 128, 2), (11, 128, 1, 3), (32, 199, 128, 4), (100, 7, 16, 5), (64, 199, 128, 6),
                            (12, 199, 128, 7)):
        torch.manual_seed(seed)
        f = torch.randn(bz, 1, T2, d)
        lab = torch.tensor(([1] * ((5 * bz + 10) // 11) + [0] * bz)[:bz])
        fr = f.clone().requires_grad_(True)
        ref = OH.supcon_loss(fr, lab)
        (ref * 0.37).backward()
        K = T2 * d
        Fd = f.view(bz, K).to(dev).contiguous()
        ws = torch.empty(ops.supcon_nchunks(K) * bz * bz, device=dev)
        G = torch.empty(2 * bz * bz, device=dev); loss = torch.empty(1, device=dev)
        ops.supcon_fwd(Fd, lab.to(dev), bz, K, K, T2, 0.07, ws, G, loss)
        assert abs(loss.item() - ref.item()) <= 2e-5 * max(1.0, abs(ref.item())), (bz, T2, d)
        dF = torch.empty(bz, K, device=dev)
        up = torch.tensor([0.37], device=dev)
        ops.supcon_bwd(Fd, G, up, 1.0, bz, K, K, T2, 0.07, dF)
        assert rel(dF, fr.grad.view(bz, K)) < 2e-4, (bz, T2, d)
        # accumulate onto an existing gradient, and the bf16 copy (both forms: scalar kernels for short / ragged rows, fp32 GEMMs else)
        acc = torch.ones(bz, K, device=dev)
        ops.supcon_bwd(Fd, G, up, 1.0, bz, K, K, T2, 0.07, acc, accumulate=True)
        assert rel(acc - 1.0, fr.grad.view(bz, K)) < 1e-3, (bz, T2, d)
        dF2 = torch.empty(bz, K, device=dev); dFb = torch.empty(bz, K, device=dev, dtype=torch.bfloat16)
        ops.supcon_bwd(Fd, G, up, 1.0, bz, K, K, T2, 0.07, dF2, dF_bf16=dFb)
        assert torch.equal(dF2, dF) and torch.equal(dFb, dF.to(torch.bfloat16)), (bz, T2, d)
    # single-member class -> NaN, as loss_metrics.py:202
    torch.manual_seed(9)
    f = torch.randn(4, 80).to(dev)
    ws = torch.empty(ops.supcon_nchunks(80) * 16, device=dev); G = torch.empty(32, device=dev); loss = torch.empty(1, device=dev)
    ops.supcon_fwd(f, torch.tensor([1, 0, 0, 0], device=dev), 4, 80, 80, 10, 0.07, ws, G, loss)
    assert torch.isnan(loss).item()
    # NLL term: CE on log-probs, mean, then / bz
    bz = 11
    logits = torch.randn(bz, 2, generator=g(7))
    logp = F.log_softmax(logits, 1).requires_grad_(True)
    y = torch.tensor([1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0])
    ref = F.cross_entropy(logp, y) / bz
    ref.backward()
    lo = torch.empty(1, device=dev); coef = torch.empty(bz, 2, device=dev)
    ops.nll_fwd(logp.detach().to(dev), y.to(dev), bz, 2, lo, coef)
    assert abs(lo.item() - ref.item()) < 1e-6 and rel(coef, logp.grad) < 1e-5


def test_adamw_matches_torch(dev):
    n = 10007
    p = torch.randn(n, generator=g(1)).to(dev); p_ref = p.clone().requires_grad_(True)
    m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    pb = torch.empty(n + 1, dtype=torch.bfloat16, device=dev)
    opt = torch.optim.AdamW([p_ref], lr=1e-3, weight_decay=1e-2)
    for step in range(1, 4):
        gr = torch.randn(n, generator=g(10 + step)).to(dev)
        p_ref.grad = gr.clone()
        opt.step()
        ops.adamw_flat(p, gr, m, v, pb, n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step)
        assert rel(p, p_ref.detach()) < 1e-6
    assert torch.equal(pb[:n], p.to(torch.bfloat16))


@pytest.mark.parametrize("B,T,H", [(2, 199, 3), (1, 201, 2), (3, 49, 2), (2, 16, 1), (1, 224, 1)])
def test_fused_attention_fwd_bwd(dev, B, T, H):
    """scl_attn_fwd / scl_attn_bwd (head dim 64) vs fp32 softmax attention on the same bf16-rounded q, k, v."""
    D, E = 64, H * 64
    qkv = (0.7 * torch.randn(B, T, 3, H, D, generator=g(1))).to(torch.bfloat16).to(dev)
    ctx = torch.full((B, T, E), float("nan"), dtype=torch.bfloat16, device=dev)
    lse = torch.full((B, H, T), float("nan"), device=dev)
    scale = D ** -0.5
    ops.attn_fwd(qkv, ctx, lse, B, T, H, D, scale)
    q, k, v = (qkv[:, :, i].float().permute(0, 2, 1, 3).clone().requires_grad_(True) for i in range(3))   # [B,H,T,D]
    s = (q @ k.transpose(-1, -2)) * scale
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, T, E)
    assert rel(ctx, ref) < 1.2e-2
    assert rel(lse, torch.logsumexp(s, -1)) < 1e-5
    dctx = torch.randn(B, T, E, generator=g(2)).to(torch.bfloat16).to(dev)
    ref.backward(dctx.float())
    dqkv = torch.full((B, T, 3, H, D), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.attn_bwd(qkv, ctx, dctx, lse, dqkv, B, T, H, D, scale)
    for i, gr in enumerate((q.grad, k.grad, v.grad)):
        got = dqkv[:, :, i].float().permute(0, 2, 1, 3)
        assert rel(got, gr) < 2.5e-2, ("qkv"[i], rel(got, gr))
        cos = torch.nn.functional.cosine_similarity(got.flatten().cpu(), gr.flatten().cpu(), dim=0).item()
        assert cos > 0.999, ("qkv"[i], cos)
    # the same launch with the fused bias-gradient sums: dqkv unchanged bit for bit, bias_part[b] = column sums of utterance b's
    # dqkv rows (from the f32 accumulators: compared with the sum of the bf16-rounded rows at the rounding of T addends)
    dq2 = torch.full_like(dqkv, float("nan"))
    part = torch.full((B, 3 * E), float("nan"), device=dev)
    ops.attn_bwd(qkv, ctx, dctx, lse, dq2, B, T, H, D, scale, bias_part=part)
    assert torch.equal(dq2, dqkv)
    ref_cs = dqkv.float().reshape(B, T, 3 * E).sum(1)
    bound = 2.0 ** -8 * dqkv.float().abs().reshape(B, T, 3 * E).sum(1) + 1e-6
    assert ((part - ref_cs).abs() <= bound).all()
    exact = torch.stack([(q.grad * 1.0).permute(0, 2, 1, 3).reshape(B, T, E), k.grad.permute(0, 2, 1, 3).reshape(B, T, E),
                         v.grad.permute(0, 2, 1, 3).reshape(B, T, E)], dim=2).reshape(B, T, 3 * E).sum(1)
    assert rel(part, exact) < 2.5e-2


@pytest.mark.parametrize("B,T,H", [(2, 199, 3), (1, 201, 2), (3, 49, 2), (2, 16, 1), (1, 256, 1), (2, 208, 16)])
def test_fp8_attention_forward_against_fp32(dev, B, T, H):
    """scl_attn_fwd_fp8 (BASELINE.json configs[4]: K, V, Q and the probabilities as e4m3 MFMA operands, fp32 accumulation) against fp32
    soft-max attention on the same q, k, v: 6e-2 relative L2 on the context (e4m3 carries 3 mantissa bits: 2^-4 per operand element,
    averaged over the 64-deep and T-deep contractions), 2e-2 absolute on the row log-sum-exp; and within 6e-2 of the bf16 kernel."""
    D, E = 64, H * 64
    qkv = (0.7 * torch.randn(B, T, 3, H, D, generator=g(11))).to(torch.bfloat16).to(dev)
    ctx = torch.full((B, T, E), float("nan"), dtype=torch.bfloat16, device=dev)
    lse = torch.full((B, H, T), float("nan"), device=dev)
    scale = D ** -0.5
    ops.attn_fwd_fp8(qkv, ctx, lse, B, T, H, D, scale)
    q, k, v = (qkv[:, :, i].float().permute(0, 2, 1, 3) for i in range(3))
    s = (q @ k.transpose(-1, -2)) * scale
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, T, E)
    assert torch.isfinite(ctx.float()).all() and torch.isfinite(lse).all()
    assert rel(ctx, ref) < 6e-2, rel(ctx, ref)
    assert float((lse - torch.logsumexp(s, -1)).abs().max()) < 2e-2
    ctx16 = torch.empty_like(ctx); lse16 = torch.empty_like(lse)
    ops.attn_fwd(qkv, ctx16, lse16, B, T, H, D, scale)
    assert rel(ctx, ctx16.float()) < 6e-2


@pytest.mark.parametrize("B,T,K,G", [(3, 199, 128, 16), (2, 208, 128, 2), (2, 49, 128, 16), (1, 7, 16, 1), (2, 100, 32, 3)])
def test_posconv_mfma_equals_the_grouped_gemm(dev, B, T, K, G):
    """csrc/posconv.hip (utterance slab resident in LDS, weights streamed tap by tap) against the same contraction through
    scl_gemm_bf16 (2-level contiguous index over the padded rows), forward form (bias + GELU + pre-activation copy + f32 residual) and
    data-gradient form (+ residual): bit for bit, including T = 208 (all 13 row tiles full), short utterances, few taps, and buffers
    followed by NaN (a fetch outside the utterance's padded rows would show)."""
    from scl_amd.ops import Op
    Cg = 64
    E, M = G * Cg, B * T
    gen = torch.Generator().manual_seed(5)
    def nan_tail(t, extra=65536):
        buf = torch.full((t.numel() + extra,), float("nan"), dtype=t.dtype, device=dev)
        buf[:t.numel()] = t.reshape(-1).to(dev)
        return buf
    xpad = torch.zeros(B, T + K, E)
    xpad[:, K // 2: K // 2 + T] = torch.randn(B, T, E, generator=gen) * 0.5
    xpad = nan_tail(xpad.to(torch.bfloat16))
    w = nan_tail((torch.randn(G, Cg, K * Cg, generator=gen) * 0.02).to(torch.bfloat16))
    bias = torch.randn(E, generator=gen).to(dev)
    R = torch.randn(M, E, generator=gen).to(dev)
    assert ops.posconv_supported(T, K, G, Cg)
    outs = []
    for kernel in ("gemm", "mfma"):
        for fwd in (True, False):
            C = torch.full((M, E), float("nan"), device=dev)
            c2 = torch.full((M, E), float("nan"), dtype=torch.bfloat16, device=dev) if fwd else None
            if kernel == "gemm":
                kw = dict(bias=bias, bias_bs2=Cg, act=1, c2=c2) if fwd else {}
                ops.gemm(Op(xpad, E, rpb=T, rbstride=(T + K) * E, cin=Cg, cout=E, bs2=Cg), Op(w, K * Cg, bs2=Cg * K * Cg), C, M, Cg, K * Cg,
                         nb2=G, ldc=E, c_bs2=Cg, R=R, rmode=1, **kw)
            else:
                ops.posconv_mfma(xpad, w, C, R, B, T, K, G, Cg, bias=bias if fwd else None, c2=c2)
            outs.append((C, c2))
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.isfinite(outs[i][0]).all()
        assert torch.equal(outs[i][0], outs[2 + i][0]), ("fwd" if i == 0 else "dgrad", (outs[i][0] - outs[2 + i][0]).abs().max().item())
    assert torch.equal(outs[0][1], outs[2][1])
    assert not ops.posconv_supported(209, 128, 16, 64) and not ops.posconv_supported(199, 128, 16, 32) and not ops.posconv_supported(199, 127, 16, 64)


@pytest.mark.parametrize("B,T,K,G", [(5, 199, 128, 16), (2, 224, 128, 2), (3, 49, 128, 16), (1, 7, 16, 1), (3, 100, 32, 3)])
def test_posconv_wgrad_equals_the_transposed_gemm(dev, B, T, K, G):
    """csrc/posconv.hip's weight gradient (one wave per tap, accumulators resident over the utterances) against the same contraction
    through scl_gemm_bf16 (both operands transposed, f32 output) and against torch in f64: the summation orders differ (utterance by
    utterance here, one pass over all rows there), so 2e-5 of the largest entry; buffers are followed by NaN and the rows of the
    gradient image outside [dy_row0, dy_row0 + T) hold NaN-free garbage that must not be read as gradient."""
    from scl_amd.ops import Op
    Cg = 64
    E, M = G * Cg, B * T
    pb = K // 2 - 1
    gen = torch.Generator().manual_seed(7)
    def nan_tail(t, extra=65536):
        buf = torch.full((t.numel() + extra,), float("nan"), dtype=t.dtype, device=dev)
        buf[:t.numel()] = t.reshape(-1).to(dev)
        return buf
    x = torch.zeros(B, T + K, E); x[:, K // 2: K // 2 + T] = torch.randn(B, T, E, generator=gen) * 0.5
    dy = torch.zeros(B, T + K, E); dy[:, pb: pb + T] = torch.randn(B, T, E, generator=gen) * 0.1
    xb, dyb = x.to(torch.bfloat16), dy.to(torch.bfloat16)
    xpad, dypad = nan_tail(xb), nan_tail(dyb)
    assert ops.posconv_wgrad_supported(T, K, G, Cg)
    got = torch.full((G, Cg, K * Cg), float("nan"), device=dev)
    ops.posconv_wgrad(dypad, pb, xpad, got, B, T, K, G, Cg)
    ref = torch.full((G, Cg, K * Cg), float("nan"), device=dev)
    ops.gemm(Op(dypad, E, rpb=T, rbstride=(T + K) * E, bs2=Cg, offset=pb * E), Op(xpad, E, rpb=T, rbstride=(T + K) * E, cin=Cg, cout=E, bs2=Cg),
             ref, Cg, K * Cg, M, a_t=True, b_t=True, nb2=G, c_bs2=Cg * K * Cg, ldc=K * Cg)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 2e-5 * scale, ((got - ref).abs().max().item(), scale)
    # torch, f64: dw[g][o][tap][c] = sum_{b,t} dy[b][pb+t][g,o] x[b][t+tap][g,c]
    xd, dyd = xb.double().view(B, T + K, G, Cg), dyb.double().view(B, T + K, G, Cg)[:, pb: pb + T]
    want = torch.stack([torch.einsum("btgo,btgc->goc", dyd, xd[:, tap: tap + T]) for tap in range(K)], dim=2).reshape(G, Cg, K * Cg)
    assert (got.cpu().double() - want).abs().max().item() <= 2e-5 * want.abs().max().item()


@pytest.mark.parametrize("M,N", [(64 * 42 * 66, 64), (70000, 32), (200000, 128), (5000, 8)])
def test_self_finishing_column_sum_cuts_narrow_matrices_finer(dev, M, N):
    """scl_colsum_reduce over a narrow matrix (N <= 128: a single column group) uses up to 512 row slabs instead of 64 — the AASIST
    back-end's [177408, 64] bias-gradient sums ran on 44 workgroups.  Sum against torch in f64, repeatable bit for bit, and the
    partial-row count is what scl_colsum_reduce_nparts promises (rows beyond it stay untouched)."""
    x = torch.randn(M, N, generator=g(9)).to(dev)
    n = ops.colsum_reduce_nparts(M, N)
    assert n >= ops.colsum_nparts(M) and n <= 512
    first = None
    for _ in range(3):
        part = torch.full((n + 1, N), float("nan"), device=dev); out = torch.full((N,), float("nan"), device=dev)
        ops.colsum_reduce(x, part, out, M, N)
        assert torch.isfinite(part[:n]).all() and torch.isnan(part[n]).all()
        assert ((out.double().cpu() - x.double().sum(0).cpu()).abs() / x.double().abs().sum(0).cpu()).max().item() < 1e-6
        first = out.clone() if first is None else first
        assert torch.equal(out, first)


def test_reduce_slabs_multi_equals_the_single_launches(dev):
    """scl_reduce_slabs_multi (one launch for the split-K combines of an encoder layer's four weight gradients, csrc/norm.hip) keeps every
    job's grid, element-to-thread map and slab order: bit-identical to scl_reduce_slabs_f32 per job — ragged tails, a stride larger than
    n, one slab, sixteen slabs, more elements than the 2048-block grid covers in one sweep."""
    specs = [(4096 * 1024, 4, 4096 * 1024), (1024 * 1024, 16, 1024 * 1024), (3072 * 1024, 5, 3072 * 1024), (1003, 3, 1004),
             (8, 1, 8), (2048 * 256 * 4 * 3 + 7, 2, 2048 * 256 * 4 * 3 + 8)]
    jobs, refs = [], []
    for i, (n, sk, stride) in enumerate(specs):
        slabs = torch.randn(sk * stride, generator=g(70 + i)).to(dev)
        out = torch.full((n,), float("nan"), device=dev)
        ref = torch.full((n,), float("nan"), device=dev)
        ops.reduce_slabs(slabs, ref, n, sk, stride)
        jobs.append((slabs, out, n, sk, stride)); refs.append(ref)
    ops.reduce_slabs_multi(jobs)
    for (slabs, out, n, sk, stride), ref in zip(jobs, refs):
        assert torch.equal(out, ref), (n, sk)
        acc = slabs.view(sk, stride)[:, :n].double().sum(0)
        assert rel(out, acc.float()) < 1e-6
    with pytest.raises(Exception):
        ops.reduce_slabs_multi(jobs + jobs)      # more than 8 jobs
