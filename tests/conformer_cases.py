"""Shared by tests/test_conformer_cpu.py and tests/test_conformer_gpu.py: the cases of tests/golden/conformer.npz (written by
oracle/gen_golden.py::gen_conformer from the reference's own ConformerBlock) and the parameters they were run with."""
import ast
import os

import numpy as np
import torch

from oracle.aasist import fill_state

G = os.path.join(os.path.dirname(__file__), "golden", "conformer.npz")
CASES = ("a", "b", "c", "d")
TRAIN = {"a": True, "b": False, "c": True, "d": True}


def load():
    return np.load(G, allow_pickle=False)


def cfg_of(g, case):
    return dict(ast.literal_eval(str(g[case + ":cfg"])))


def shapes_of(cfg):
    """State-dict shapes of ConformerBlock(**cfg) (model/conformer.py:180-206), written out so that the CPU tests need no module."""
    dim, dh, H = cfg["dim"], cfg.get("dim_head", 64), cfg.get("heads", 8)
    Fd, inner, k = dim * cfg.get("ff_mult", 4), dim * cfg.get("conv_expansion_factor", 2), cfg.get("conv_kernel_size", 31)
    s = {}
    for pre in ("ff1", "ff2"):
        s[pre + ".fn.fn.net.0.weight"], s[pre + ".fn.fn.net.0.bias"] = (Fd, dim), (Fd,)
        s[pre + ".fn.fn.net.3.weight"], s[pre + ".fn.fn.net.3.bias"] = (dim, Fd), (dim,)
        s[pre + ".fn.norm.weight"], s[pre + ".fn.norm.bias"] = (dim,), (dim,)
    s["attn.fn.to_q.weight"], s["attn.fn.to_kv.weight"] = (H * dh, dim), (2 * H * dh, dim)
    s["attn.fn.to_out.weight"], s["attn.fn.to_out.bias"] = (dim, H * dh), (dim,)
    s["attn.fn.rel_pos_emb.weight"] = (1025, dh)
    s["attn.norm.weight"], s["attn.norm.bias"] = (dim,), (dim,)
    s["conv.net.0.weight"], s["conv.net.0.bias"] = (dim,), (dim,)
    s["conv.net.2.weight"], s["conv.net.2.bias"] = (2 * inner, dim, 1), (2 * inner,)
    s["conv.net.4.conv.weight"], s["conv.net.4.conv.bias"] = (inner, 1, k), (inner,)
    if not cfg.get("conv_causal", False):
        s["conv.net.5.weight"], s["conv.net.5.bias"] = (inner,), (inner,)
        s["conv.net.5.running_mean"], s["conv.net.5.running_var"], s["conv.net.5.num_batches_tracked"] = (inner,), (inner,), ()
    s["conv.net.7.weight"], s["conv.net.7.bias"] = (dim, inner, 1), (dim,)
    s["post_norm.weight"], s["post_norm.bias"] = (dim,), (dim,)
    return s


def state_of(g, case):
    """{state-dict key: float32 / int64 tensor} the reference ran this case with."""
    filled = fill_state(shapes_of(cfg_of(g, case)), seed=int(g[case + ":seed"]))
    return {k: torch.from_numpy(v) for k, v in filled.items()}


def fingerprint(t):
    t = t.detach().double().cpu()
    return np.concatenate([[float(t.pow(2).sum().sqrt()), float(t.sum())], t.flatten()[:16].numpy()])


def check_grads(g, case, grads, tol, what):
    """grads: {parameter name: tensor}.  Whole tensors where the fixture holds them, (norm, sum, first 16 values) otherwise."""
    n = 0
    for key in g.files:
        if key.startswith(case + ":grad:"):
            name = key[len(case) + 6:]
            want = g[key]
            got = grads[name].detach().double().cpu().numpy().reshape(want.shape)
            if name == "conv.net.4.conv.bias" and TRAIN[case] and case + ":buf:conv.net.5.running_mean" in g.files:
                # a bias in front of a BatchNorm on batch statistics: the true gradient is 0, both sides hold round-off
                ref = float(grads["conv.net.4.conv.weight"].detach().abs().max())
                assert np.abs(got).max() < 1e-4 * ref and np.abs(want).max() < 1e-4 * ref, "%s %s %s" % (what, case, name)
                n += 1
                continue
            err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-9)
            assert err < tol, "%s %s grad %s: rel err %.3e" % (what, case, name, err)
            n += 1
        elif key.startswith(case + ":gradfp:"):
            name = key[len(case) + 8:]
            want, got = g[key], fingerprint(grads[name])
            assert abs(got[0] - want[0]) < tol * want[0], "%s %s grad norm %s" % (what, case, name)
            assert np.abs(got[2:] - want[2:]).max() < tol * want[0], "%s %s grad head %s" % (what, case, name)
            n += 1
    return n
