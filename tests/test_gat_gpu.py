"""Fused AASIST pairwise attention score (csrc/gat.hip) against the torch formulation the reference uses
(model/wav2vec2_aasist.py:107-135, 259-291), fp32 on both sides: forward 1e-5, gradients 1e-4 (relative to the tensor max)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scl_amd.gat import gat_score  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12)).item()


def ref_score(x, W, bias, a, n1):
    pair = x.unsqueeze(2) * x.unsqueeze(1)                       # [B, N, N, D]
    h = torch.tanh(torch.nn.functional.linear(pair, W, bias))    # [B, N, N, Do]
    N = x.shape[1]
    if a.shape[0] == 1:
        return (h @ a[0].unsqueeze(-1)).squeeze(-1)
    board = torch.zeros_like(h[..., 0])
    board[:, :n1, :n1] = (h[:, :n1, :n1] @ a[0].unsqueeze(-1)).squeeze(-1)
    board[:, n1:, n1:] = (h[:, n1:, n1:] @ a[1].unsqueeze(-1)).squeeze(-1)
    board[:, :n1, n1:] = (h[:, :n1, n1:] @ a[2].unsqueeze(-1)).squeeze(-1)
    board[:, n1:, :n1] = (h[:, n1:, :n1] @ a[2].unsqueeze(-1)).squeeze(-1)
    return board


@pytest.mark.parametrize("B,N,D,Do,n1,na", [(3, 66, 64, 64, 66, 1), (2, 42, 64, 64, 42, 1), (4, 54, 64, 32, 33, 3), (2, 26, 32, 32, 16, 3),
                                            (1, 1, 64, 64, 1, 1), (2, 128, 32, 17, 100, 3)])
def test_gat_score_fwd_bwd(dev, B, N, D, Do, n1, na):
    g = torch.Generator().manual_seed(N * 7 + Do)
    x = torch.randn(B, N, D, generator=g).to(dev)
    W = (torch.randn(Do, D, generator=g) / D ** 0.5).to(dev)
    bias = (0.1 * torch.randn(Do, generator=g)).to(dev)
    a = torch.randn(na, Do, generator=g).to(dev)
    gs = torch.randn(B, N, N, generator=g).to(dev)
    xr, Wr, br, ar = (t.clone().requires_grad_(True) for t in (x, W, bias, a))
    ref = ref_score(xr, Wr, br, ar, n1)
    ref.backward(gs)
    xs, Ws, bs, as_ = (t.clone().requires_grad_(True) for t in (x, W, bias, a))
    got = gat_score(xs, Ws, bs, as_, n1)
    assert got.shape == (B, N, N) and rel(got, ref) < 1e-5
    got.backward(gs)
    assert rel(xs.grad, xr.grad) < 1e-4 and rel(Ws.grad, Wr.grad) < 1e-4 and rel(bs.grad, br.grad) < 1e-4 and rel(as_.grad, ar.grad) < 1e-4
    # deterministic: the same call twice gives the same bits
    xs2, Ws2 = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
    got2 = gat_score(xs2, Ws2, bias, a, n1)
    got2.backward(gs)
    assert torch.equal(got2, got) and torch.equal(xs2.grad, xs.grad) and torch.equal(Ws2.grad, Ws.grad)
