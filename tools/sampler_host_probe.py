"""Host cost of the throughput-mode RawBoost sampler (no GPU involved): the batched notch-filter design (zero-phase route vs the FFT route it
replaced) and the ISD position draw, per batch of clips, on this machine's cores."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scl_amd import augment as A
from scl_amd.datautils_common import default_rawboost_args

a = default_rawboost_args()
r = np.random.default_rng(0)


def mk(n, nb=5):
    fc = r.uniform(a.minF, a.maxF, (n, nb)); bw = r.uniform(a.minBW, a.maxBW, (n, nb))
    c = r.uniform(a.minCoeff, a.maxCoeff, (n, nb)).astype(np.int64); c = np.where(c % 2 == 0, c + 1, c)
    return fc, bw, c, r.uniform(-20, -5, n)


def tm(f, k=40):
    f(); f()
    ts = []
    for _ in range(k):
        t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    return ts[k // 2]


print("# median ms per call, %d logical cores" % os.cpu_count())
for n in (55, 320, 330):
    fc, bw, c, G = mk(n)
    f1 = fc - bw / 2; f2 = fc + bw / 2
    f1 = np.where(f1 <= 0, 1 / 1000, f1) / 8000.0; f2 = np.where(f2 >= 8000.0, 8000.0 - 1 / 1000, f2) / 8000.0
    print("design %3d filters: zero-phase route %.2f ms | FFT route %.2f ms" % (n, tm(lambda: A.design_notch_filters(fc, bw, c, G, 16000)),
                                                                           tm(lambda: A._design_notch_filters_general(f1, f2, c, G))))
A.seed_fast_sampler(1)
for n in (11, 64, 66):
    print("%2d clips: LnL draw + design %.2f ms, ISD draw %.2f ms" % (n, tm(lambda: A._fast_lnl(a, n, 16000)), tm(lambda: A._fast_isd(a, n, 64000))))
