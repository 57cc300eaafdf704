"""A/B of the 256x256 ping-pong GEMM (FORCE_P8) against the 128x128 LDS-DMA kernel (NO_P8 | NO_BIG) on the encoder shapes:
interleaved rounds in one process, median per variant, bitwise comparison of the outputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = B * 199
bf = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)


def run(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cases = []
for name, N_, K_ in (("qkv", 3072, 1024), ("out", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
    cases.append((name + " fwd", M, N_, K_, False, False, 1))
    cases.append((name + " dgrad", M, K_, N_, False, True, 1))
    for sk in (1, 2, 4):
        cases.append((name + " wgrad sk%d" % sk, N_, K_, M, True, True, sk))
cases.append(("conv2 fwd", 32 * 6399, 512, 1536, False, False, 1))
cases.append(("conv2 dgrad", 32 * 6399, 1536, 512, False, True, 1))
for name, Mm, Nn, Kk, at, bt, sk in cases:
    A = bf(Kk, Mm) if at else bf(Mm, Kk)
    Bm = bf(Kk, Nn) if bt else bf(Nn, Kk)
    outs, times = {}, {}
    for var, kw in (("t128", dict(no_p8=True, no_big=True)), ("p8", dict(force_p8=True))):
        C = torch.zeros(sk, Mm, Nn, dtype=torch.float32, device=dev) if sk > 1 else torch.zeros(Mm, Nn, dtype=torch.bfloat16, device=dev)
        fn = (lambda C=C, kw=kw: ops.gemm(Op(A, Mm if at else Kk), Op(Bm, Nn if bt else Kk), C, Mm, Nn, Kk, a_t=at, b_t=bt, splitk=sk,
                                          c_split_stride=Mm * Nn if sk > 1 else 0, **kw))
        fn(); torch.cuda.synchronize()
        outs[var] = C
        times[var] = fn
    meas = {v: [] for v in times}
    for _ in range(5):
        for v, fn in times.items():
            meas[v].append(run(fn))
    med = {v: sorted(x)[len(x) // 2] for v, x in meas.items()}
    fl = 2.0 * Mm * Nn * Kk
    same = torch.equal(outs["t128"], outs["p8"])
    print("%-16s M=%6d N=%5d K=%5d  128: %7.1f us %6.0f TF | p8: %7.1f us %6.0f TF | x%.2f  bitwise %s" % (
        name, Mm, Nn, Kk, med["t128"], fl / med["t128"] / 1e6, med["p8"], fl / med["p8"] / 1e6, med["t128"] / med["p8"], same))
