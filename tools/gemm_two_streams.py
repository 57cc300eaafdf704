"""Does running a layer's wgrad GEMMs on a second stream next to its dgrad chain recover the tile-quantisation tails?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op
dev = torch.device("cuda:0")
M, E, Fd = 6368, 1024, 4096
bf = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)
f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
dx, a, W2, d_f, h2, W1, d_h = bf(M, E), bf(M, Fd), bf(E, Fd), bf(M, Fd), bf(M, E), bf(Fd, E), bf(M, E)
ctx, Wo, d_ctx, dqkv, h1, Wqkv = bf(M, E), bf(E, E), bf(M, E), bf(M, 3 * E), bf(M, E), bf(3 * E, E)
gW2, gW1, gWo, gWq = f32(2, E, Fd), f32(2, Fd, E), f32(8, E, E), f32(2, 3 * E, E)
def dgrads():
    ops.gemm(Op(dx, E), Op(W2, Fd), d_f, M, Fd, E, b_t=True)
    ops.gemm(Op(d_f, Fd), Op(W1, E), d_h, M, E, Fd, b_t=True)
    ops.gemm(Op(dx, E), Op(Wo, E), d_ctx, M, E, E, b_t=True)
    ops.gemm(Op(dqkv, 3 * E), Op(Wqkv, E), d_h, M, E, 3 * E, b_t=True)
def wgrads():
    ops.gemm(Op(dx, E), Op(a, Fd), gW2, E, Fd, M, a_t=True, b_t=True, splitk=2, c_split_stride=E * Fd)
    ops.gemm(Op(d_f, Fd), Op(h2, E), gW1, Fd, E, M, a_t=True, b_t=True, splitk=2, c_split_stride=E * Fd)
    ops.gemm(Op(dx, E), Op(ctx, E), gWo, E, E, M, a_t=True, b_t=True, splitk=8, c_split_stride=E * E)
    ops.gemm(Op(dqkv, 3 * E), Op(h1, E), gWq, 3 * E, E, M, a_t=True, b_t=True, splitk=2, c_split_stride=3 * E * E)
side = torch.cuda.Stream()
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def serial():
    dgrads(); wgrads()
def overlapped():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        wgrads()
        ev2 = torch.cuda.Event(); ev2.record()
    dgrads()
    torch.cuda.current_stream().wait_event(ev2)
plan_s = None
print("dgrads only %.1f us, wgrads only %.1f us" % (timeit(dgrads), timeit(wgrads)))
print("serial %.1f us   two streams %.1f us" % (timeit(serial), timeit(overlapped)))
