"""Data parallelism: one process per GPU, gradients all-reduced over RCCL / xGMI.

The reference has no working multi-GPU training path (nn.DataParallel + model.loss);
the semantics implemented here are §8(e)'s: every rank runs augmentation, forward, the rank-local
loss and backward on its own shard of packs, then ONE exchange — an all-reduce (mean) of the flat
fp32 gradient buffer — before the (replicated) AdamW step.  The buffer is cut into a few large
buckets; because the hand-written backward finishes gradients from the END of the buffer (head,
then encoder layers 23..0, then the conv stack), each bucket's all-reduce is launched as soon as the
backward has passed its lower edge and overlaps with the rest of the backward.  xGMI is
point-to-point, so buckets are large (16 Mi elements = 64 MiB of fp32: per-collective latency is amortised, and only the
last bucket — the conv stack, finished at the very end of backward — is exposed).
Works with any torch.distributed backend ("nccl" = RCCL on the GPU box, "gloo" in the CPU tests).
"""
import time

import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world, epoch_seed=None, drop_last=True):
    """Rank-strided sharding of a pack list (optionally shuffled with a seed shared by all ranks)."""
    idx = torch.arange(n_items)
    if epoch_seed is not None:
        g = torch.Generator().manual_seed(int(epoch_seed))
        idx = idx[torch.randperm(n_items, generator=g)]
    if drop_last:
        idx = idx[: (n_items // world) * world]
    return idx[rank::world].tolist()


class GatherForGlobalLoss(torch.autograd.Function):
    """all_gather of a per-rank tensor [b, ...] into [world * b, ...] (rank order) for a loss term that EVERY rank evaluates identically on
    the gathered tensor — the optional global-batch SupCon of SURVEY.md 8(e): nn.DataParallel in the reference would have gathered
    `feats` onto one device before Model.loss.  Backward: the rank keeps the rows it contributed and multiplies them by the world
    size — the N identical copies of the term each back-propagate only through their own rows, the gradient exchange then averages over
    ranks, and N x (1/N) leaves exactly the gradient of ONE global term.  No second collective."""

    @staticmethod
    def forward(ctx, x, group=None):
        world = dist.get_world_size(group)
        ctx.rank, ctx.world, ctx.b = dist.get_rank(group), world, x.shape[0]
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.b:(ctx.rank + 1) * ctx.b] * float(ctx.world), None


def gather_for_global_loss(x, group=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return x
    if x.requires_grad:
        return GatherForGlobalLoss.apply(x, group)
    out = torch.empty((dist.get_world_size(group) * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out


class GradSync:
    """Bucketed all-reduce of the flat gradient slice, launched in backward order.

    Streams: on the GPU every bucket is reduced on a DEDICATED communication stream — the compute stream records an event when the
    bucket's slice is final, the communication stream waits for it, and the optimizer waits for the communication stream at
    `finish()`.  RCCL's kernels therefore never sit in the compute queue in front of backward GEMMs (with torch's default, an
    async_op collective is ordered on the current stream's side, and a 64-MiB ring step issued mid-backward would delay the next
    launches).  `wire="bf16"` halves the bytes on the xGMI links: each bucket is cast to a bf16 staging buffer, summed in bf16 by
    the collective and accumulated back in fp32 (the sum of `world` bf16 values carries ~3 significant digits per element — the
    fp32 wire stays the default).  Per-bucket timing (`report()`): when the bucket's all-reduce was issued and how long the
    optimizer had to wait for it at the end — the exposed part a scaling run needs to see."""

    def __init__(self, flat_grad, group=None, bucket_elems=16 * 1024 * 1024, base=0, wire="fp32", force=False, mode="allreduce"):
        """flat_grad: the slice of the flat gradient buffer that receives gradients (model.trainable_range()); `base` = its first
        element's offset in the whole buffer (ready_above() is called with whole-buffer offsets).  `force`: run the collectives
        even in a one-rank group (a single-GPU box can then rehearse the whole RCCL choreography: tests/test_dp_gpu.py)."""
        self.grad = flat_grad
        self.base = int(base)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.wire = wire
        assert wire in ("fp32", "bf16")
        # mode "shard" (optional): every bucket is REDUCE-SCATTERED instead of all-reduced — rank r ends up with the
        # summed gradient of the r-th 1/world of the bucket only, runs AdamW on that shard (optimizer HBM traffic and state / world)
        # and the updated parameters are all-gathered afterwards (gather_params).  Same bytes on the links as the all-reduce
        # (a ring all-reduce IS reduce-scatter + all-gather).  fp32 wire only.
        assert mode in ("allreduce", "shard") and not (mode == "shard" and wire != "fp32")
        self.mode = mode
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if mode == "shard":
            bucket_elems = max(self.world, bucket_elems // self.world * self.world)
        n = flat_grad.numel()
        self.bounds = []
        hi = n
        while hi > 0:
            lo = max(0, hi - bucket_elems)
            self.bounds.append((lo, hi))     # ordered from the END of the buffer (backward order)
            hi = lo
        self.launched = 0
        self.works = []
        self.on_gpu = flat_grad.is_cuda
        self.comm = torch.cuda.Stream(device=flat_grad.device) if (self.on_gpu and self.active) else None
        self.stage = torch.empty(min(bucket_elems, n), dtype=torch.bfloat16, device=flat_grad.device) if wire == "bf16" else None
        self.final_check = None              # tests: callable(lo, hi) run right before a bucket is handed to the collective
        self._t_issue, self._ev_done, self._exposed_ms = [], [], 0.0
        self.last_report = None

    def begin(self):
        self.launched = 0
        self.works = []
        self._t_issue, self._ev_done = [], []

    def shard_of(self, lo, hi):
        """The part of bucket [lo, hi) this rank owns in mode "shard" (the whole bucket when its size does not divide by the world
        size — at most one, the lowest: it is all-reduced and updated on every rank)."""
        if self.mode != "shard" or not self.active or (hi - lo) % self.world:
            return lo, hi
        sh = (hi - lo) // self.world
        return lo + self.rank * sh, lo + (self.rank + 1) * sh

    def owned(self):
        """Slice-relative [lo, hi) ranges whose summed gradient this rank holds after finish() — what its optimizer updates."""
        return [self.shard_of(lo, hi) for lo, hi in self.bounds]

    def gather_params(self, flat_params):
        """Mode "shard", after the optimizer step: all-gather every bucket of the parameter slice (same slicing as the gradient)
        from the ranks that own its shards."""
        if self.mode != "shard" or not self.active:
            return
        works = []
        for lo, hi in self.bounds:
            if (hi - lo) % self.world:
                continue                                    # replicated bucket: every rank made the same update
            a, b = self.shard_of(lo, hi)
            works.append(dist.all_gather_into_tensor(flat_params[lo:hi], flat_params[a:b].clone(), group=self.group, async_op=True))
        for w in works:
            w.wait()

    def _reduce(self, lo, hi):
        g = self.grad[lo:hi]
        if self.mode == "shard" and (hi - lo) % self.world == 0:
            a, b = self.shard_of(lo, hi)
            out = torch.empty(b - a, dtype=g.dtype, device=g.device)      # not in place: only NCCL documents output == input chunk
            w = dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            return w, ("shard", out, a, b)
        if self.stage is None:
            return dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None
        st = self.stage[: hi - lo]
        st.copy_(g)                                            # fp32 -> bf16 on the wire
        return dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group, async_op=True), st

    def ready_above(self, lo_offset):
        """All gradient elements at offsets >= lo_offset are final: launch every bucket above it."""
        if not self.active:
            return
        lo_offset = max(0, lo_offset - self.base)
        while self.launched < len(self.bounds) and self.bounds[self.launched][0] >= lo_offset:
            lo, hi = self.bounds[self.launched]
            if self.final_check is not None:
                self.final_check(lo, hi)
            if self.comm is not None:
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream())
                with torch.cuda.stream(self.comm):
                    self.comm.wait_event(ready)                # the slice is final on the compute stream
                    w, st = self._reduce(lo, hi)
                    if st is not None:
                        w.wait()                               # orders the copy-back behind the collective on the comm stream
                        if isinstance(st, tuple):
                            self.grad[st[2]:st[3]].copy_(st[1])    # this rank's shard of the summed gradient
                        else:
                            self.grad[lo:hi].copy_(st)
                        w = None
                    done = torch.cuda.Event(enable_timing=False)
                    done.record(self.comm)
                self._ev_done.append(done)
            else:
                w, st = self._reduce(lo, hi)
                if st is not None:
                    w.wait()
                    if isinstance(st, tuple):
                        self.grad[st[2]:st[3]].copy_(st[1])
                    else:
                        self.grad[lo:hi].copy_(st)
                    w = None
            self.works.append(w)
            self._t_issue.append(time.perf_counter())
            self.launched += 1

    def finish(self):
        """Launch what is left, wait for everything, return the 1/world scale the optimizer applies."""
        if not self.active:
            return 1.0
        self.ready_above(0)
        t0 = time.perf_counter()
        for w in self.works:
            if w is not None:
                w.wait()
        if self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)     # the optimizer kernel is ordered behind every bucket
        self.last_report = {"buckets": len(self.bounds), "bucket_mib": [round((hi - lo) * (2 if self.stage is not None else 4) / 2 ** 20, 1)
                                                                         for lo, hi in self.bounds],
                            "issue_ms_before_finish": [round((t0 - t) * 1e3, 3) for t in self._t_issue],
                            "host_wait_ms_at_finish": round((time.perf_counter() - t0) * 1e3, 3), "wire": self.wire, "mode": self.mode}
        self.works = []
        self.launched = 0
        return 1.0 / self.world

    def report(self):
        """Diagnostics of the last step (bench.py prints them for N > 1): bucket sizes, how long before the end of the backward each
        bucket's all-reduce was issued (host clock), and the host-side wait at finish()."""
        return self.last_report
