"""Pins oracle/conformer.py to the reference's own ConformerBlock (model/conformer.py:180-216) through tests/golden/conformer.npz —
outputs, input gradient, every parameter gradient (whole or as a fingerprint), BatchNorm buffers; train / eval + mask / causal /
clamped relative positions.  CPU only; fp32 on both sides, so the bound is fp32 round-off of differently ordered sums."""
import numpy as np
import pytest
import torch

from oracle import conformer as OC
from tests import conformer_cases as CC


@pytest.mark.parametrize("case", CC.CASES)
def test_oracle_conformer_block_matches_the_reference(case):
    g = CC.load()
    cfg = CC.cfg_of(g, case)
    st = CC.state_of(g, case)
    t = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in st.items()}
    x = torch.from_numpy(g[case + ":x"]).clone().requires_grad_(True)
    mask = torch.from_numpy(g[case + ":mask"]) if case + ":mask" in g.files else None
    y = OC.forward(t, x, cfg.get("heads", 8), CC.TRAIN[case], mask=mask)
    (y * torch.from_numpy(g[case + ":wout"])).sum().backward()
    want = g[case + ":y"]
    assert np.abs(y.detach().numpy() - want).max() < 2e-5 * np.abs(want).max()
    want = g[case + ":grad_x"]
    assert np.abs(x.grad.numpy() - want).max() < 5e-5 * np.abs(want).max()
    n = CC.check_grads(g, case, {k: v.grad for k, v in t.items() if v.requires_grad}, 1e-4, "oracle")
    assert n == sum(1 for v in t.values() if v.requires_grad)
    for key in g.files:
        if key.startswith(case + ":buf:"):
            np.testing.assert_allclose(t[key[len(case) + 5:]].detach().numpy(), g[key], rtol=1e-5, atol=1e-6)


def test_shapes_table_is_the_reference_state_dict():
    g = CC.load()
    assert sorted(g["init:keys"]) == sorted(CC.shapes_of(dict(dim=64, dim_head=16, heads=4)))


def test_hip_block_has_the_reference_state_dict_and_refuses_cpu_tensors():
    """Host logic of scl_amd/conformer.py (no kernel runs): the module tree gives the reference's state-dict keys in the reference's order,
    a seeded construction gives the reference's initial weights (same torch modules created in the same order), and a CPU tensor is
    refused — there is no CPU fallback."""
    from scl_amd import conformer as C
    g = CC.load()
    torch.manual_seed(1234)
    blk = C.ConformerBlock(dim=64, dim_head=16, heads=4)
    sd = blk.state_dict()
    assert list(sd.keys()) == list(g["init:keys"])
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], g["init:fp"], rtol=0, atol=0)
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in CC.shapes_of(dict(dim=64, dim_head=16, heads=4)).items()}
    causal = C.ConformerBlock(dim=32, dim_head=32, heads=2, ff_mult=2, conv_kernel_size=8, conv_causal=True)
    assert sorted(causal.state_dict().keys()) == sorted(CC.shapes_of(CC.cfg_of(g, "c")))
    assert causal.pad == (7, 0) and blk.pad == (15, 15) and C.ConformerBlock(dim=32, conv_kernel_size=8).pad == (4, 3)
    with pytest.raises(RuntimeError):
        blk(torch.zeros(1, 4, 64))
    net = C.Conformer(32, depth=2, dim_head=16, heads=2)
    assert len(net.layers) == 2 and list(net.state_dict().keys())[0] == "layers.0.ff1.fn.fn.net.0.weight"
