"""Flat parameter storage laid out for the GPU, exposed under the reference's state-dict names.

All parameters that receive gradients live in ONE contiguous fp32 buffer (`flat[:n_train]`) with a
parallel fp32 gradient buffer and a bf16 working copy at the SAME element offsets, so that
  * AdamW + the bf16 refresh is a single kernel over the buffer (csrc/optim.hip),
  * the data-parallel gradient all-reduce is a handful of large contiguous buckets,
  * q/k/v projection weights of a layer sit back to back and are used as one [3E, E] GEMM operand.
Parameters that exist only for checkpoint compatibility (fairseq's pre-training heads, SURVEY.md
§8b) follow after `n_train` and are never touched by the optimizer, exactly like torch.optim.AdamW
skips parameters whose .grad is None (main.py:339).
"""
import torch
from torch import nn

ALIGN = 8  # elements: 16 B in bf16, 32 B in f32 — every GEMM operand offset stays vector-aligned


def _aligned(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


class FlatParams:
    def __init__(self, specs, device):
        """specs: list of (name, shape, trainable) in the desired memory order."""
        self.device = torch.device(device)
        self.index = {}
        off = 0
        for name, shape, tr in specs:
            if not tr:
                continue
            n = 1
            for s in shape:
                n *= s
            self.index[name] = (off, n, tuple(shape), True)
            off += _aligned(n)
        self.n_train = off
        for name, shape, tr in specs:
            if tr:
                continue
            n = 1
            for s in shape:
                n *= s
            self.index[name] = (off, n, tuple(shape), False)
            off += _aligned(n)
        self.n_total = off
        self.flat = torch.zeros(self.n_total, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(self.n_train, dtype=torch.float32, device=self.device)
        self.bf16 = torch.zeros(self.n_train, dtype=torch.bfloat16, device=self.device)
        self.version = 0        # bumped whenever `flat` changes (optimizer step, load_state_dict)
        self.bf16_version = -1  # version the bf16 copy / derived weights were built from
        self.params = {}
        for name, (o, n, shape, tr) in self.index.items():
            p = nn.Parameter(self.flat[o:o + n].view(shape), requires_grad=tr)
            if tr:
                p.grad = self.grad[o:o + n].view(shape)
            self.params[name] = p

    # views --------------------------------------------------------------------------------------
    def off(self, name):
        return self.index[name][0]

    def f32(self, name):
        o, n, shape, _ = self.index[name]
        return self.flat[o:o + n].view(shape)

    def g(self, name):
        o, n, shape, tr = self.index[name]
        assert tr, name
        return self.grad[o:o + n].view(shape)

    def rebind_grads(self):
        """(Re-)attach .grad views after something (e.g. optimizer.zero_grad(set_to_none=True)) dropped them."""
        for name, p in self.params.items():
            o, n, shape, tr = self.index[name]
            if tr and (p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o):
                p.grad = self.grad[o:o + n].view(shape)

    def mark_dirty(self):
        self.version += 1


def register_by_name(root, dotted, param):
    """Attach `param` to `root` under a dotted state-dict name, creating bare nn.Modules on the way."""
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, nn.Module())
        m = m._modules[p]
    m.register_parameter(parts[-1], param)
