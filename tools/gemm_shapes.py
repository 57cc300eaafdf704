"""Micro-benchmark of every GEMM shape the XLS-R-300M train step launches (B=32, L=64000): TFLOP/s per shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op

dev = torch.device("cuda:0")
B, T, E, C, Fd, H, D, K, G = 32, 199, 1024, 512, 4096, 16, 64, 128, 16
M = B * T
bf = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)
f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)


def timeit(fn, flops, name, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("%-46s %8.3f ms %8.1f TFLOP/s" % (name, ms, flops / ms / 1e9))
    return ms

tot = 0.0
# linears fwd (NT)
for name, N_, K_ in (("qkv fwd", 3 * E, E), ("out fwd", E, E), ("fc1 fwd", Fd, E), ("fc2 fwd", E, Fd)):
    A, W, Cc = bf(M, K_), bf(N_, K_), torch.empty(M, N_, dtype=torch.bfloat16, device=dev)
    tot += 24 * timeit(lambda: ops.gemm(Op(A, K_), Op(W, K_), Cc, M, N_, K_), 2.0 * M * N_ * K_, name + " NT %dx%dx%d" % (M, N_, K_))
    dY = bf(M, N_); dX = torch.empty(M, K_, dtype=torch.bfloat16, device=dev)
    tot += 24 * timeit(lambda: ops.gemm(Op(dY, N_), Op(W, K_), dX, M, K_, N_, b_t=True), 2.0 * M * N_ * K_, name.replace("fwd", "dgrad") + " NN")
    dW = f32(N_, K_)
    tiles = ((N_ + 127) // 128) * ((K_ + 127) // 128)
    sk = max(1, min(32, 512 // tiles, (M + 63) // 64))
    slab = f32(sk, N_, K_)
    def wg():
        if sk == 1:
            ops.gemm(Op(dY, N_), Op(A, K_), dW, N_, K_, M, a_t=True, b_t=True)
        else:
            ops.gemm(Op(dY, N_), Op(A, K_), slab, N_, K_, M, a_t=True, b_t=True, splitk=sk, c_split_stride=N_ * K_)
            ops.reduce_slabs(slab, dW, N_ * K_, sk, N_ * K_)
    tot += 24 * timeit(wg, 2.0 * M * N_ * K_, name.replace("fwd", "wgrad") + " TT splitk=%d" % sk)
# attention
qkv = bf(M, 3 * E); S = f32(B * H * T * 200); P = bf(B * H * T * 208 + 1024); ctx = torch.empty(M, E, dtype=torch.bfloat16, device=dev)
fl = 2.0 * B * H * T * T * D
tot += 48 * timeit(lambda: ops.gemm(Op(qkv, 3 * E, bs1=T * 3 * E, bs2=D), Op(qkv, 3 * E, bs1=T * 3 * E, bs2=D, offset=E), S, T, T, D, nb1=B, nb2=H, ldc=200, c_bs1=H * T * 200, c_bs2=T * 200), fl, "QK^T (also dP) batched 512x 199x199x64")
tot += 48 * timeit(lambda: ops.gemm(Op(P, 208, bs1=H * T * 208, bs2=T * 208), Op(qkv, 3 * E, bs1=T * 3 * E, bs2=D, offset=2 * E), ctx, T, D, T, b_t=True, nb1=B, nb2=H, ldc=E, c_bs1=T * E, c_bs2=D), fl, "PV (also dQ) NN batched")
tot += 48 * timeit(lambda: ops.gemm(Op(P, 208, bs1=H * T * 208, bs2=T * 208), Op(ctx, E, bs1=T * E, bs2=D), qkv, T, D, T, a_t=True, b_t=True, nb1=B, nb2=H, ldc=3 * E, c_bs1=T * 3 * E, c_bs2=D), fl, "dV (also dK) TT batched")
# conv stack
Ts = [12799, 6399, 3199, 1599, 799, 399, 199]
ks = [10, 3, 3, 3, 3, 2, 2]
for i in range(1, 7):
    k, Tin, Tout = ks[i], Ts[i - 1], Ts[i]
    z = bf(B * Tin * C + 65536); wk = bf(C, k * C); y = f32(B * Tout, C); Mi = B * Tout
    fl = 2.0 * Mi * C * k * C
    tot += timeit(lambda: ops.gemm(Op(z, 2 * C, rpb=Tout, rbstride=Tin * C), Op(wk, k * C), y, Mi, C, k * C), fl, "conv%d fwd M=%d K=%d" % (i, Mi, k * C))
    dy = bf(Mi, C); dcol = torch.empty(Mi, k * C, dtype=torch.bfloat16, device=dev)
    tot += timeit(lambda: ops.gemm(Op(dy, C), Op(wk, k * C), dcol, Mi, k * C, C, b_t=True), fl, "conv%d dgrad" % i)
    tiles = 4 * ((k * C + 127) // 128)
    sk = max(1, min(32, 512 // tiles, (Mi + 63) // 64))
    slab = f32(sk, C, k * C); dW = f32(C, k * C)
    def cw():
        ops.gemm(Op(dy, C), Op(z, 2 * C, rpb=Tout, rbstride=Tin * C), slab, C, k * C, Mi, a_t=True, b_t=True, splitk=sk, c_split_stride=C * k * C)
        ops.reduce_slabs(slab, dW, C * k * C, sk, C * k * C)
    tot += timeit(cw, fl, "conv%d wgrad splitk=%d" % (i, sk))
# pos conv
Cg = E // G
xpad = bf(B * (T + K) * E + 65536); wf = bf(G, Cg, K * Cg); xo = f32(M, E); x0 = f32(M, E); pre = torch.empty(M, E, dtype=torch.bfloat16, device=dev)
bias = f32(E).zero_()
fl = 2.0 * M * E * K * Cg
tot += 2 * timeit(lambda: ops.gemm(Op(xpad, E, rpb=T, rbstride=(T + K) * E, cin=Cg, cout=E, bs2=Cg), Op(wf, K * Cg, bs2=Cg * K * Cg), xo, M, Cg, K * Cg, nb2=G, ldc=E, c_bs2=Cg, bias=bias, bias_bs2=Cg, act=1, c2=pre, R=x0, rmode=1), fl, "pos-conv fwd (also dgrad) 16 groups N=64 K=8192")
dwf = f32(G, Cg, K * Cg)
tot += timeit(lambda: ops.gemm(Op(xpad, E, rpb=T, rbstride=(T + K) * E, bs2=Cg, offset=63 * E), Op(xpad, E, rpb=T, rbstride=(T + K) * E, cin=Cg, cout=E, bs2=Cg), dwf, Cg, K * Cg, M, a_t=True, b_t=True, nb2=G, c_bs2=Cg * K * Cg, ldc=K * Cg), fl, "pos-conv wgrad")
print("sum of GEMM time per train step (weighted by call counts): %.2f ms" % tot)
