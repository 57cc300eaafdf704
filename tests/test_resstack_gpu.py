"""csrc/resstack.hip — the fused Residual_block stack of the AASIST encoder (model/wav2vec2_aasist.py:377-433) — against a float64
torch restatement of the reference block built from F.conv2d / F.batch_norm / F.selu on the CPU: outputs, input gradient, every
parameter gradient and the BatchNorm buffers, in training and in eval mode, on the real map size (42 x 66) and on an odd small one.
Tolerance 1e-4 of each tensor's largest magnitude (fp32 matrix cores vs float64)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scl_amd import resstack  # noqa: E402
from scl_amd.aasist_head import Residual_block  # noqa: E402

pytestmark = pytest.mark.gpu


def ref_stack(x, blocks, training):
    """x [B, 1, H, W] float64; blocks: list of dicts of float64 tensors (+ running buffers, updated in place)."""
    for i, p in enumerate(blocks):
        if i > 0:       # bn1: result discarded, running statistics updated (training)
            F.batch_norm(x, p["bn1.running_mean"], p["bn1.running_var"], p["bn1.weight"], p["bn1.bias"], training, 0.1, 1e-5)
        out = F.conv2d(x, p["conv1.weight"], p["conv1.bias"], padding=(1, 1))
        out = F.selu(F.batch_norm(out, p["bn2.running_mean"], p["bn2.running_var"], p["bn2.weight"], p["bn2.bias"], training, 0.1, 1e-5))
        out = F.conv2d(out, p["conv2.weight"], p["conv2.bias"], padding=(0, 1))
        idn = F.conv2d(x, p["conv_downsample.weight"], p["conv_downsample.bias"], padding=(0, 1)) if "conv_downsample.weight" in p else x
        x = out + idn
    return x


def close(got, want, name, tol=1e-4, kinks=0):
    """max |got - want| < tol * max |want|.  `kinks` > 0 (maps downstream of a SELU in the backward): SELU's derivative jumps from
    1.758 to 1.051 at 0, so a pre-activation within fp32 rounding of zero (|z| ~ 1e-6; one in ~1e5 elements) flips the slope against
    float64 and leaves an O(1) error in the 2 x 3 footprint of that element (tools/rs_debug.py shows exactly that: z = -6.9e-7 in
    float64, +3.9e-7 here).  Up to `kinks` footprints may exceed the bound; the relative L2 error must still be below 10 tol."""
    got = np.asarray(torch.as_tensor(got).detach().double().cpu())
    want = np.asarray(torch.as_tensor(want).detach().double().cpu())
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = np.abs(got - want) / max(np.abs(want).max(), 1e-9)
    nbad = int((err >= tol).sum())
    rl2 = float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))
    assert nbad <= kinks * 6 * 64 and rl2 < 10 * tol, "%s: %d elements above %.0e (max rel err %.3e), rel L2 %.3e" % (name, nbad, tol, err.max(), rl2)


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("B,H,W,filts", [(3, 42, 66, [[1, 32], [32, 32], [32, 64], [64, 64]]), (2, 5, 9, [[1, 32], [32, 64], [64, 64], [64, 64]]),
                                        (5, 42, 67, [[1, 32], [32, 32]])])
def test_fused_stack_matches_float64_reference(B, H, W, filts, training):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 1000 + W)
    blocks = [Residual_block(f, first=(i == 0)) for i, f in enumerate(filts)]
    refp = []
    for blk in blocks:
        with torch.no_grad():
            for n, p in blk.named_parameters():
                # fan-in scaled filters: activations stay O(1) through the stack in eval mode too (running statistics do not re-normalise)
                std = (1.0 / np.sqrt(p[0].numel())) if p.dim() > 1 else 0.2
                p.copy_(torch.randn(p.shape, generator=g) * std + (1.0 if n.endswith("bn2.weight") or n.endswith("bn1.weight") else 0.0))
            for n, b in blk.named_buffers():
                if n.endswith("running_mean"):
                    b.copy_(0.1 * torch.randn(b.shape, generator=g))
                elif n.endswith("running_var"):
                    b.copy_(0.5 + torch.rand(b.shape, generator=g))
        d = {n: p.detach().double().clone().requires_grad_(True) for n, p in blk.named_parameters()}
        d.update({n: b.detach().double().clone() for n, b in blk.named_buffers() if b.dtype.is_floating_point})
        refp.append(d)
        blk.to(dev)
        blk.train(training)
    assert resstack.supported(blocks)
    x = torch.randn(B, H, W, 1, generator=g)
    wout = torch.randn(B, H, W, filts[-1][1], generator=g)
    xg = x.to(dev).requires_grad_(True)
    out = resstack.res_stack(xg, blocks)
    (out * wout.to(dev)).sum().backward()
    torch.cuda.synchronize()
    xr = x.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
    ro = ref_stack(xr, refp, training)
    (ro * wout.double().permute(0, 3, 1, 2)).sum().backward()
    close(out.permute(0, 3, 1, 2), ro, "output")
    close(xg.grad.permute(0, 3, 1, 2), xr.grad, "grad_x", kinks=2)
    for i, (blk, d) in enumerate(zip(blocks, refp)):
        for n, p in blk.named_parameters():
            if n.startswith("bn1."):
                assert p.grad is None or float(p.grad.abs().max()) == 0.0      # discarded branch: no gradient
                continue
            # conv1's bias sits in front of a BatchNorm: its true gradient is 0 in training and both sides compute round-off
            tol = 1e-4
            if n == "conv1.bias" and training:
                assert float(p.grad.abs().max()) < 1e-3 * float(blk.conv2.bias.grad.abs().max())
                continue
            close(p.grad, d[n].grad, "block %d %s" % (i, n), tol * (5 if i < len(blocks) - 1 else 1))      # a slope flip downstream shifts these sums by ~1e-4
        for n, b in blk.named_buffers():
            if b.dtype.is_floating_point:
                close(b, d[n], "block %d %s" % (i, n), 1e-5)
            elif training:
                assert int(b) == 1, (i, n, int(b))


def test_two_forwards_then_backwards_keep_their_own_activations():
    """A plan (the stack's saved activations) is busy until its backward: a second forward in between must not overwrite them."""
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    blocks = [Residual_block(f, first=(i == 0)).to(dev) for i, f in enumerate([[1, 32], [32, 32]])]
    for b in blocks:
        b.train()
    xa = torch.randn(2, 6, 7, 1, device=dev, requires_grad=True)
    xb = torch.randn(2, 6, 7, 1, device=dev, requires_grad=True)
    oa = resstack.res_stack(xa, blocks)
    ob = resstack.res_stack(xb, blocks)
    oa.sum().backward()
    ga = xa.grad.clone()
    xa.grad = None
    for b in blocks:
        b.zero_grad()
    oa2 = resstack.res_stack(xa, blocks)
    oa2.sum().backward()
    ob.sum().backward()
    torch.cuda.synchronize()
    assert torch.allclose(oa, oa2, rtol=1e-5, atol=1e-6) and torch.allclose(ga, xa.grad, rtol=1e-4, atol=1e-6)
    assert torch.isfinite(xb.grad).all() and float(xb.grad.abs().max()) > 0


def test_many_pending_forwards_never_share_a_plan():
    """Five grad-mode forwards before any backward (micro-batch loops, retained graphs): every node keeps its own plan, so each
    backward reproduces the gradient of a lone forward + backward of the same input; graphs that are dropped without a backward give
    their plans back; past the hard cap a forward raises instead of overwriting somebody's activations."""
    import gc
    dev = torch.device("cuda:0")
    torch.manual_seed(6)
    blocks = [Residual_block(f, first=(i == 0)).to(dev) for i, f in enumerate([[1, 32], [32, 32]])]
    for b in blocks:
        b.eval()          # batch statistics would couple the comparison to the running buffers' history
    xs = [torch.randn(2, 5, 9, 1, device=dev, requires_grad=True) for _ in range(5)]
    lone = []
    for x in xs:
        resstack.res_stack(x, blocks).square().sum().backward()
        lone.append(x.grad.clone())
        x.grad = None
    busy = lambda: sum(1 for pl in resstack._PLANS if pl.busy)
    busy0 = busy()          # plans other tests' still-alive graphs hold
    outs = [resstack.res_stack(x, blocks) for x in xs]
    assert busy() == busy0 + 5
    for o in reversed(outs):
        o.square().sum().backward()
    torch.cuda.synchronize()
    for x, g in zip(xs, lone):
        assert torch.allclose(x.grad, g, rtol=1e-4, atol=1e-6)
    del outs, o
    gc.collect()
    assert busy() == busy0
    held = [resstack.res_stack(x, blocks) for x in xs[:3]]
    assert busy() == busy0 + 3
    del held
    gc.collect()
    assert busy() == busy0          # released by the nodes' finalizers
    held = [resstack.res_stack(xs[0], blocks) for _ in range(resstack.MAX_LIVE_PLANS)]
    with pytest.raises(RuntimeError, match="waiting for their backward"):
        resstack.res_stack(xs[0], blocks)
    del held
    gc.collect()
    resstack.res_stack(xs[0], blocks).sum().backward()


# ---- kernel level: scl_rs_conv / scl_rs_wgrad on random bordered maps against the flat-shift definition evaluated with torch in float64 ----
import ctypes  # noqa: E402

from scl_amd import lib as L  # noqa: E402


def _flat_case(B, H, W, cin, cout, nt, seed, transposed=False):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    Wp, RPU = W + 2, H + 2
    G = B * RPU * Wp
    slack = Wp + 2 + 256 + 6
    full = lambda c: torch.randn(G + 2 * slack, c, generator=g)
    shifts = ([kh * Wp + kw - 1 for kh in (0, 1) for kw in (0, 1, 2)] if nt == 6 else [-1, 0, 1])
    if transposed:
        shifts = [-s for s in shifts]
    return dev, g, Wp, RPU, G, slack, full, shifts


def _valid_mask(B, H, W, r_lo, r_hi):
    m = torch.zeros(B, H + 2, W + 2, dtype=torch.bool)
    m[:, r_lo:r_hi + 1, 1:W + 1] = True
    return m.reshape(-1)


@pytest.mark.parametrize("cin,cout,nt", [(16, 32, 6), (32, 32, 6), (32, 64, 6), (64, 64, 6), (16, 32, 3), (32, 64, 3), (32, 16, 6), (64, 32, 6), (32, 16, 3), (64, 32, 3)])
@pytest.mark.parametrize("mode", ["plain", "stats", "bnbwd"])
def test_rs_conv_kernel_against_the_flat_shift_definition(cin, cout, nt, mode):
    B, H, W = 3, 42, 66
    transposed = cout < cin or (cin, cout, nt) in ()
    dev, g, Wp, RPU, G, slack, full, shifts = _flat_case(B, H, W, cin, cout, nt, 7 * cin + cout + nt, transposed)
    lib = L.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    xin, add, act, y1 = full(cin), full(cout), full(cout), full(cout)
    r_lo, r_hi = (0, H) if mode == "bnbwd" else (1, H)
    wt = torch.randn(nt, cin, cout, generator=g) / np.sqrt(nt * cin)           # Wt[t][c][n]
    # the torch-layout weight whose (transposed) image is Wt: forward W[n][c][t] = Wt[t][c][n]; transposed W[c][n][t] = Wt[t][c][n]
    w_torch = (wt.permute(2, 1, 0) if not transposed else wt.permute(1, 2, 0)).contiguous()
    wpk = torch.empty(nt * cin * cout, device=dev)
    wd = w_torch.to(dev)
    job = (L.SclRsPackJob * 1)(L.SclRsPackJob(wd.data_ptr(), wpk.data_ptr(), wd.shape[0], wd.shape[1], nt, cin, cout, 1 if transposed else 0))
    assert lib.scl_rs_pack_weights(job, 1, st) == 0
    bias = torch.randn(cout, generator=g)
    stats = torch.cat([0.1 * torch.randn(cout, generator=g), 0.5 + torch.rand(cout, generator=g), torch.randn(2 * cout, generator=g)])
    d_in, d_add, d_act, d_y1, d_out = xin.to(dev), add.to(dev), act.to(dev), y1.to(dev), torch.full((G + 2 * slack, cout), 7.0, device=dev)
    d_bias, d_stats = bias.to(dev), stats.to(dev)
    acc = torch.zeros(16 * 2 * cout, dtype=torch.float64, device=dev)
    ticket = torch.zeros(1, dtype=torch.int32, device=dev)
    sout = torch.zeros(4 * cout, device=dev)
    gam, bet = torch.randn(cout, generator=g).to(dev), torch.randn(cout, generator=g).to(dev)
    rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    nbt = torch.zeros(1, dtype=torch.int64, device=dev)
    dg, db = torch.zeros(cout, device=dev), torch.zeros(cout, device=dev)
    d = L.SclRsConv()
    off = lambda t, c: t.data_ptr() + 4 * slack * c
    d.inp, d.wpk, d.out, d.bias, d.addend = off(d_in, cin), wpk.data_ptr(), off(d_out, cout), d_bias.data_ptr(), off(d_add, cout)
    d.acc, d.ticket = acc.data_ptr(), ticket.data_ptr()
    d.geom = L.SclRsGeom(B, H, W, r_lo, r_hi, 0)
    for i, s in enumerate(shifts):
        d.shift[i] = s
    d.cin, d.cout, d.ntaps, d.training, d.eps, d.momentum = cin, cout, nt, 1, 1e-5, 0.1
    mask = _valid_mask(B, H, W, r_lo, r_hi)
    nvalid = float(mask.sum())
    d.nvalid = nvalid
    if mode == "stats":
        d.stat_mode, d.stats_out, d.gamma, d.beta, d.run_mean, d.run_var, d.nbt = 1, sout.data_ptr(), gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr()
    if mode == "bnbwd":
        d.stat_mode, d.stats_out, d.act_a, d.y1, d.bnstats, d.dgamma, d.dbeta = 2, sout.data_ptr(), off(d_act, cout), off(d_y1, cout), d_stats.data_ptr(), dg.data_ptr(), db.data_ptr()
    instantiated = mode == "plain" or (nt == 6 and ((mode == "stats" and cout >= cin) or (mode == "bnbwd" and cin == cout)))
    if not instantiated:          # the statistics epilogues exist where the stack uses them; anything else is refused, not mis-computed
        assert lib.scl_rs_conv(ctypes.byref(d), st) == -1 and b"not instantiated" in lib.scl_last_error()
        return
    for rep in range(2):          # twice: the accumulators / ticket must come back zeroed
        assert lib.scl_rs_conv(ctypes.byref(d), st) == 0, lib.scl_last_error()
    torch.cuda.synchronize()
    # float64 definition
    x64 = xin.double()
    ref = bias.double()[None, :] + add.double()[slack:slack + G]
    for t, s in enumerate(shifts):
        ref = ref + x64[slack + s: slack + s + G] @ wt[t].double()
    if mode == "bnbwd":
        a = act.double()[slack:slack + G]
        ref = ref * torch.where(a > 0, torch.full_like(a, 1.0507009873554805), a + 1.0507009873554805 * 1.6732632423543772)
    ref = torch.where(mask[:, None], ref, torch.zeros_like(ref))
    got = d_out[slack:slack + G].double().cpu()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 2e-5 * scale, float((got - ref).abs().max()) / scale
    assert float((d_out[:slack] - 7.0).abs().max()) == 0 and float((d_out[slack + G:] - 7.0).abs().max()) == 0      # nothing written outside [0, G)
    assert int(ticket) == 0 and float(acc.abs().max()) == 0.0
    if mode == "stats":
        m = ref.sum(0) / nvalid
        var = (ref * ref).sum(0) / nvalid - m * m
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        so = sout.double().cpu().view(4, cout)
        assert torch.allclose(so[0], m, rtol=1e-5, atol=1e-6 * scale) and torch.allclose(so[1], rstd, rtol=2e-5)
        assert torch.allclose(so[2], gam.double().cpu() * rstd, rtol=2e-5, atol=1e-6) and torch.equal(so[3], bet.double().cpu())
        assert int(nbt) == 2
        unb = var * nvalid / (nvalid - 1)
        want_rv = 0.9 * (0.9 * 1.0 + 0.1 * unb) + 0.1 * unb
        assert torch.allclose(rv.double().cpu(), want_rv, rtol=2e-5)
    if mode == "bnbwd":
        xh = (y1.double()[slack:slack + G] - stats[:cout].double()) * stats[cout:2 * cout].double()
        s0, s1 = ref.sum(0), (ref * torch.where(mask[:, None], xh, torch.zeros_like(xh))).sum(0)
        sa = float(ref.abs().sum(0).max())
        assert float((db.double().cpu() - 2 * s0).abs().max()) < 1e-5 * sa, float((db.double().cpu() - 2 * s0).abs().max()) / sa      # two launches accumulated
        assert float((dg.double().cpu() - 2 * s1).abs().max()) < 1e-5 * float((ref * xh).abs().sum(0).max())
        so = sout.double().cpu()
        assert torch.allclose(so[:cout], s0 / nvalid, rtol=1e-4, atol=1e-6 * sa / nvalid) and torch.allclose(so[cout:2 * cout], s1 / nvalid, rtol=1e-4, atol=1e-5 * sa / nvalid)


@pytest.mark.parametrize("cin,cout,nt", [(16, 32, 6), (32, 32, 6), (32, 64, 6), (64, 64, 6), (16, 32, 3), (32, 64, 3)])
def test_rs_wgrad_kernel_against_the_flat_shift_definition(cin, cout, nt):
    B, H, W = 3, 42, 66
    dev, g, Wp, RPU, G, slack, full, shifts = _flat_case(B, H, W, cin, cout, nt, 11 * cin + cout + nt)
    lib = L.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    xin = full(cin)
    mask = _valid_mask(B, H, W, 1, H)
    dout = torch.zeros(G + 2 * slack, cout)
    dout[slack:slack + G] = torch.where(mask[:, None], torch.randn(G, cout, generator=g), torch.zeros(G, cout))
    dout[slack + G:] = float("nan")                 # past the map: the kernel must not multiply what lies there
    d_in, d_dout = xin.to(dev), dout.to(dev)
    nsl = lib.scl_rs_wgrad_nslabs(cin, cout)
    part = torch.empty(nsl * nt * cin * cout, device=dev)
    acc = torch.zeros(16 * cout, dtype=torch.float64, device=dev)
    ticket = torch.zeros(1, dtype=torch.int32, device=dev)
    ci_real = 1 if cin == 16 else cin
    dw = torch.zeros(cout, ci_real, nt, device=dev)
    dbias = torch.zeros(cout, device=dev)
    arr = (ctypes.c_int32 * 6)(*(shifts + [0] * (6 - len(shifts))))
    geom = L.SclRsGeom(B, H, W, 1, H, 0)
    off = lambda t, c: t.data_ptr() + 4 * slack * c
    for rep in range(2):
        assert lib.scl_rs_wgrad(off(d_in, cin), off(d_dout, cout), cin, cout, nt, arr, ctypes.byref(geom), part.data_ptr(), acc.data_ptr(), ticket.data_ptr(), dbias.data_ptr(), st) == 0
        assert lib.scl_rs_wgrad_reduce(part.data_ptr(), nsl, nt, cin, cout, cout, ci_real, 0, dw.data_ptr(), st) == 0
    torch.cuda.synchronize()
    do64 = dout.double()[slack:slack + G]
    ref = torch.stack([xin.double()[slack + s: slack + s + G].t() @ do64 for s in shifts])      # [t][c][n]
    want = 2 * ref.permute(2, 1, 0)[:, :ci_real]                                                # torch layout [n][c][t], two accumulating launches
    got = dw.double().cpu()
    assert float((got - want).abs().max()) < 3e-5 * float(want.abs().max()), float((got - want).abs().max()) / float(want.abs().max())
    assert float((dbias.double().cpu() - 2 * do64.sum(0)).abs().max()) < 1e-5 * float(do64.abs().sum(0).max())
    assert int(ticket) == 0 and float(acc.abs().max()) == 0.0
