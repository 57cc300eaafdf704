"""Micro-benchmark of csrc/resstack.hip's convolution / weight-gradient kernels at the AASIST map size (B x 44 x 68 positions), per
variant of the epilogue; optional alternative builds of the one source (tools/rs_probe_build.sh) are timed in the same process."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scl_amd import lib as L  # noqa: E402


def load(path):
    lib = ctypes.CDLL(path)
    for name, (argtypes, restype) in L._protos().items():
        if name.startswith("scl_rs_"):
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = argtypes, restype
    return lib


def run(lib, tag, B=64, H=42, W=66, cin=64, cout=64, nt=6, reps=30):
    dev = torch.device("cuda:0")
    Wp = W + 2
    G = B * (H + 2) * Wp
    slack = Wp + 2 + 256 + 6
    z = lambda c: torch.randn((G + 2 * slack) * c, device=dev) * 0.1
    xin, out, add, a, y1 = z(cin), z(cout), z(cout), z(cout), z(cout)
    p = lambda t, c: t.data_ptr() + 4 * slack * c
    wpk = torch.randn(nt * cin * cout, device=dev) * 0.05
    acc = torch.zeros(16 * 2 * 64, dtype=torch.float64, device=dev)
    ticket = torch.zeros(1, dtype=torch.int32, device=dev)
    stats = torch.rand(4 * cout, device=dev) + 0.5
    sout = torch.zeros(4 * cout, device=dev)
    gam = torch.ones(cout, device=dev)
    shifts = [kh * Wp + kw - 1 for kh in (0, 1) for kw in (0, 1, 2)][:nt] if nt == 6 else [-1, 0, 1]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    geom = L.SclRsGeom(B, H, W, 1, H, 0)
    flops = 2.0 * G * cin * cout * nt
    res = {}
    for name, kw in (("plain", {}), ("stats", dict(stat=1)), ("addend", dict(add=1)), ("addend+stats", dict(add=1, stat=1)), ("bn-bwd", dict(stat=2))):
        d = L.SclRsConv()
        d.inp, d.wpk, d.out = p(xin, cin), wpk.data_ptr(), p(out, cout)
        d.acc, d.ticket = acc.data_ptr(), ticket.data_ptr()
        d.geom = geom
        for i, s in enumerate(shifts):
            d.shift[i] = s
        d.cin, d.cout, d.ntaps = cin, cout, nt
        d.nvalid = float(B * H * W)
        d.eps, d.momentum, d.training = 1e-5, 0.1, 1
        if kw.get("add"):
            d.addend = p(add, cout)
        if kw.get("stat") == 1:
            d.stat_mode, d.stats_out, d.gamma, d.beta = 1, sout.data_ptr(), gam.data_ptr(), gam.data_ptr()
        if kw.get("stat") == 2:
            d.stat_mode, d.stats_out, d.act_a, d.y1, d.bnstats = 2, sout.data_ptr(), p(a, cout), p(y1, cout), stats.data_ptr()
        if lib.scl_rs_conv(ctypes.byref(d), st) != 0:
            continue          # this statistics mode is not instantiated for the shape
        for _ in range(3):
            assert lib.scl_rs_conv(ctypes.byref(d), st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.scl_rs_conv(ctypes.byref(d), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        res[name] = us
        print("%-14s conv %d->%d x%d  %-13s %7.1f us  %6.1f TFLOP/s (of 157)" % (tag, cin, cout, nt, name, us, flops / us / 1e6))
    # weight gradient
    if (cin, cout, nt) in ((64, 64, 6), (32, 32, 6), (32, 64, 6), (16, 32, 6), (16, 32, 3), (32, 64, 3)):
        nsl = lib.scl_rs_wgrad_nslabs(cin, cout)
        part = torch.empty(nsl * nt * cin * cout, device=dev)
        arr = (ctypes.c_int32 * 6)(*(shifts + [0] * (6 - len(shifts))))
        db = torch.zeros(cout, device=dev)
        for _ in range(3):
            assert lib.scl_rs_wgrad(p(xin, cin), p(out, cout), cin, cout, nt, arr, ctypes.byref(geom), part.data_ptr(), acc.data_ptr(), ticket.data_ptr(), db.data_ptr(), st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.scl_rs_wgrad(p(xin, cin), p(out, cout), cin, cout, nt, arr, ctypes.byref(geom), part.data_ptr(), acc.data_ptr(), ticket.data_ptr(), db.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print("%-14s wgrad %d->%d x%d %21.1f us  %6.1f TFLOP/s" % (tag, cin, cout, nt, us, flops / us / 1e6))
        dw = torch.zeros(cout * cin * nt, device=dev)
        e0.record()
        for _ in range(reps):
            lib.scl_rs_wgrad_reduce(part.data_ptr(), nsl, nt, cin, cout, cout, cin, 0, dw.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        print("%-14s wgrad reduce %d slabs %21.1f us" % (tag, nsl, e0.elapsed_time(e1) * 1e3 / reps))
    return res


if __name__ == "__main__":
    libs = [("default", os.path.join(ROOT, "scl-deepfake-audio-detection_amd", "libscl_hip.so"))]
    vdir = os.path.join(ROOT, "tools", "rs_variants")
    if os.path.isdir(vdir):
        libs += [(f[6:-3], os.path.join(vdir, f)) for f in sorted(os.listdir(vdir)) if f.endswith(".so")]
    for tag, path in libs:
        lib = load(path)
        run(lib, tag, 64, 42, 66, 64, 64, 6)
        if tag == "default":
            run(lib, tag, 64, 42, 66, 32, 32, 6)
            run(lib, tag, 64, 42, 66, 32, 64, 6)
            run(lib, tag, 64, 42, 66, 16, 32, 6)
