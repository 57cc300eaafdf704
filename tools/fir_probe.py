"""FIR kernel micro-benchmark against fp64 on the CPU: (a) the RawBoost LnL shape — 64 clips x 64000 samples, 5 power branches with
121-411 taps each; (b) the RIR shape — one 8000-tap (and 16000-tap) filter per clip.  us per launch, fp32 TFLOP/s (2 x taps x samples)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scl_amd import ops
dev = torch.device("cuda:0")


def run(nclip, L, lens_per_clip, use_pow, check_clip=0):
    rs = np.random.RandomState(0)
    nf = len(lens_per_clip[0])
    x = (0.1 * torch.randn(nclip, L, generator=torch.Generator().manual_seed(1))).to(dev)
    taps, off, ln, hh = [], [], [], []
    pos = 0
    for c in range(nclip):
        for f in range(nf):
            n = lens_per_clip[c][f]
            b = (rs.randn(n) / np.sqrt(n)).astype(np.float32)
            taps.append(b); off.append(pos); ln.append(n); hh.append((n + 1) // 2); pos += n
    taps_t = torch.from_numpy(np.concatenate(taps)).to(dev)
    i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=dev)
    toff, tlen, th = i32(off), i32(ln), i32(hh)
    y = torch.empty(nclip, L, device=dev)
    part = torch.empty(nclip * ops.fir_nblocks(L) * 4, device=dev)
    call = lambda: ops.fir_multi(x, L, L, taps_t, toff, tlen, th, nclip, nf, use_pow, y, L, L, part)
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    flops = 2.0 * sum(sum(l) for l in lens_per_clip) * L
    # fp64 check of one clip
    xc = x[check_clip].double().cpu().numpy()
    ref = np.zeros(L)
    for f in range(nf):
        b = taps[check_clip * nf + f].astype(np.float64); h = hh[check_clip * nf + f]
        xp = xc ** (f + 1) if use_pow else xc
        full = np.convolve(xp, b)              # full[m] = sum_k b[k] xp[m - k]; y[n] = full[n + h]
        ref += full[h:h + L]
    err = np.abs(y[check_clip].double().cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-12)
    return us, flops / us / 1e6, err


rs = np.random.RandomState(7)
lens = [[int(rs.randint(121, 412)) | 1 for _ in range(5)] for _ in range(64)]
us, tf, err = run(64, 64000, lens, 1)
print("LnL  64 clips x 64000, 5 branches, %d taps per clip on average: %8.1f us  %6.1f TFLOP/s  max rel err vs fp64 %.1e" % (np.mean([sum(l) for l in lens]), us, tf, err))
for R in (8000, 16000):
    us, tf, err = run(16, 64000, [[R]] * 16, 0)
    print("RIR  16 clips x 64000, %5d taps:                                   %8.1f us  %6.1f TFLOP/s  max rel err vs fp64 %.1e" % (R, us, tf, err))
