"""AdamW over the model's flat parameter buffer (one fused HIP kernel per step).

Mirror of `torch.optim.AdamW(model.parameters(), lr=args.max_lr, weight_decay=args.weight_decay)`
(main.py:339): same update rule and defaults; parameters without gradients (fairseq's pre-training
heads, the unused BatchNorms) are left untouched, as torch does for `.grad is None`.  It is a real
torch.optim.Optimizer so that `torch.optim.lr_scheduler.CyclicLR(..., cycle_momentum=False)`
(main.py:341) drives `param_groups[0]['lr']` unchanged.
"""
import os

import torch

from . import ops


class _StepOverlap:
    """Single-GPU optimizer-in-backward.  The hand-written backward finishes gradients from the END of the flat buffer and
    announces it (`ready_above`, the same hook the data-parallel bucket all-reduce uses); each finished 16 Mi-element slice is
    handed to the AdamW kernel on a side stream, where the HBM-bound update (30 B / parameter) runs underneath the MFMA-bound
    rest of the backward instead of after it.  `optimizer.step()` then only joins the streams.  Contract: exactly one backward
    per step (main.py:74-80 does that); hyper-parameters are read when a slice is launched."""

    world = 1

    def __init__(self, opt, seg_elems=16 * 1024 * 1024):
        self.opt = opt
        # the trainable range only: under flag_fix_ssl the frozen encoder gets neither weight decay nor Adam updates from a
        # gradient slice nothing writes (torch.optim.AdamW skips parameters whose .grad is None)
        lo0, hi = opt.model.trainable_range() if hasattr(opt.model, "trainable_range") else (0, opt.P.n_train)
        self.bounds = []
        while hi > lo0:
            lo = max(lo0, hi - seg_elems)
            self.bounds.append((lo, hi))
            hi = lo
        self.side = torch.cuda.Stream(device=opt.P.device)
        self.events = [torch.cuda.Event() for _ in self.bounds]
        self.launched = 0

    def begin(self):
        pass

    def ready_above(self, lo_offset):
        while self.launched < len(self.bounds) and self.bounds[self.launched][0] >= lo_offset:
            self._launch(self.launched)
            self.launched += 1

    def _launch(self, i):
        lo, hi = self.bounds[i]
        opt, P = self.opt, self.opt.P
        g = opt.param_groups[0]
        self.events[i].record(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.events[i])
            ops.adamw_flat(P.flat[lo:hi], P.grad[lo:hi], opt.exp_avg[lo:hi], opt.exp_avg_sq[lo:hi], P.bf16[lo:hi], hi - lo, float(g["lr"]),
                           g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], opt.step_count + 1, 1.0)

    def finish(self):
        launched_any = self.launched > 0
        self.ready_above(0)
        torch.cuda.current_stream().wait_stream(self.side)
        self.launched = 0
        return launched_any


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_sync=None, overlap=None):
        self.model = model
        self.P = model.P
        params = [p for p in self.P.params.values() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        n = self.P.n_train
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=self.P.device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=self.P.device)
        self.step_count = 0
        self._consolidated_at = 0      # shard mode: the step count at which every rank last held the WHOLE moments (consolidate_state)
        self.grad_sync = grad_sync
        model.grad_sync = grad_sync
        # optional on a single GPU: run the update underneath the backward (see _StepOverlap).  Off by default — measured on
        # MI355X at batch 32 x 64000 it is a wash (38.6 vs 38.5 ms/step): the 9.5 GB of optimizer traffic slows the
        # L2/Infinity-Cache-latency-bound GEMMs it runs beside by as much as it saves after them.
        if overlap is None:
            overlap = grad_sync is None and os.environ.get("SCL_ADAMW_OVERLAP", "0") == "1"
        self.overlap = _StepOverlap(self) if (overlap and grad_sync is None) else None
        if self.overlap is not None:
            model.grad_sync = self.overlap

    def zero_grad(self, set_to_none=True):
        # gradients live in one flat buffer that every backward overwrites in full; nothing to clear,
        # and the .grad views must stay attached.  A model whose back-end accumulates through autograd's .grad views clears
        # that slice (model_front.zero_torch_grads).
        if hasattr(self.model, "zero_torch_grads"):
            self.model.zero_torch_grads()
        return None

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        if self.overlap is not None:
            self.overlap.finish()          # slices not announced by the backward (the conv stack) + join the side stream
            self.step_count += 1
            self.model.optimizer_stepped(bf16_fresh=True)
            return None
        scale = 1.0
        if self.grad_sync is not None:
            scale = self.grad_sync.finish()
        self.step_count += 1
        # parameters that received no gradient (the whole encoder under flag_fix_ssl) are skipped, as torch.optim.AdamW skips
        # .grad is None: no weight decay on frozen weights
        lo, hi = self.model.trainable_range() if hasattr(self.model, "trainable_range") else (0, self.P.n_train)
        gs = self.grad_sync
        if gs is not None and getattr(gs, "mode", "allreduce") == "shard" and gs.active:
            # sharded step (GradSync mode "shard"): this rank holds the summed gradient of its shards only; update those, then
            # all-gather the parameters and refresh the bf16 working copy of the shards other ranks updated
            for a, b in gs.owned():
                a, b = lo + a, lo + b
                ops.adamw_flat(self.P.flat[a:b], self.P.grad[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], self.P.bf16[a:b], b - a,
                               float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.step_count, scale)
            gs.gather_params(self.P.flat[lo:hi])
            ops.cast_bf16(self.P.flat[lo:hi], self.P.bf16[lo:hi], hi - lo)
            self.model.optimizer_stepped(bf16_fresh=True)
            return None
        ops.adamw_flat(self.P.flat[lo:hi], self.P.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.P.bf16[lo:hi], hi - lo,
                       float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.step_count, scale)
        self.model.optimizer_stepped(bf16_fresh=True)
        return None

    def consolidate_state(self):
        """COLLECTIVE (every rank must call it): after sharded steps (SCL_DP_MODE=shard) a rank has updated the moments of its own
        shards only (the buffers are full-size on every rank: what the mode divides by the world size is the optimizer's HBM traffic
        and arithmetic, not its allocation) — collect the owners' values, so that a following state_dict() holds the whole optimizer.
        A no-op in the all-reduce mode.  state_dict() itself is local, so `if rank == 0: torch.save(opt.state_dict())` cannot deadlock."""
        gs = self.grad_sync
        if gs is not None and getattr(gs, "mode", "allreduce") == "shard" and gs.active:
            lo, hi = self.model.trainable_range() if hasattr(self.model, "trainable_range") else (0, self.P.n_train)
            gs.gather_params(self.exp_avg[lo:hi])
            gs.gather_params(self.exp_avg_sq[lo:hi])
        self._consolidated_at = self.step_count

    def _sharded(self):
        gs = self.grad_sync
        return gs is not None and getattr(gs, "mode", "allreduce") == "shard" and gs.active

    def state_dict(self):
        # After sharded steps the moments outside this rank's shards are stale: a checkpoint written from them would resume wrongly and
        # silently (round-5 advisory).  state_dict() stays local (no collective: `if rank == 0: save` cannot deadlock) and refuses instead.
        if self._sharded() and self._consolidated_at != self.step_count:
            raise RuntimeError("FusedAdamW.state_dict(): SCL_DP_MODE=shard and %d step(s) since the moments were last whole on this rank — call "
                               "consolidate_state() on EVERY rank first (it is a collective), then state_dict() on the rank that saves"
                               % (self.step_count - self._consolidated_at))
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self._consolidated_at = self.step_count
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for k, v in sd["param_groups"][0].items():
            self.param_groups[0][k] = v
