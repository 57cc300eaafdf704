"""Fused attention forward (csrc/attention.hip) at the encoder's shape: alternative builds of the one source (tools/attn_variants_build.sh:
ablations and variants) timed in one process, interleaved rounds."""
import ctypes
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scl_amd import lib as L  # noqa: E402

dev = torch.device("cuda:0")
T, H, D = 199, 16, 64
E = H * D
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
libs = {}
for path in sorted(glob.glob(os.path.join(ROOT, "tools", "attn_variants", "libattn_*.so"))):
    lib = ctypes.CDLL(path)
    a, r = L._protos()["scl_attn_fwd"]
    lib.scl_attn_fwd.argtypes, lib.scl_attn_fwd.restype = a, r
    libs[os.path.basename(path)[8:-3]] = lib
sets = []
for i in range(3):
    qkv = (0.5 * torch.randn(B * T, 3 * E, device=dev)).bfloat16()
    sets.append((qkv, torch.empty(B * T, E, device=dev, dtype=torch.bfloat16), torch.empty(B * H * T, device=dev)))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(lib, i):
    qkv, ctx, lse = sets[i % 3]
    rc = lib.scl_attn_fwd(qkv.data_ptr(), ctx.data_ptr(), lse.data_ptr(), B, T, H, D, D ** -0.5, 0.0, 0, st)
    assert rc == 0


res = {k: [] for k in libs}
for rnd in range(4):
    for name, lib in libs.items():
        for i in range(5):
            run(lib, i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(50):
            run(lib, i)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) * 1000 / 50)
for name, v in res.items():
    print("B=%d attn fwd %-10s %s  (min %.1f us)" % (B, name, " ".join("%.1f" % x for x in v), min(v)))
