"""Data parallelism: one process per GPU, gradients all-reduced over RCCL / xGMI.

The reference has no working multi-GPU training path (nn.DataParallel + model.loss);
the semantics implemented here are §8(e)'s: every rank runs augmentation, forward, the rank-local
loss and backward on its own shard of packs, then ONE exchange — an all-reduce (mean) of the flat
fp32 gradient buffer — before the (replicated) AdamW step.  The buffer is cut into a few large
buckets; because the hand-written backward finishes gradients from the END of the buffer (head,
then encoder layers 23..0, then the conv stack), each bucket's all-reduce is launched as soon as the
backward has passed its lower edge and overlaps with the rest of the backward.  xGMI is
point-to-point, so buckets are large (16 Mi elements = 64 MiB: per-collective latency is amortised, and only the
last bucket — the conv stack, finished at the very end of backward — is exposed).
Works with any torch.distributed backend ("nccl" = RCCL on the GPU box, "gloo" in the CPU tests).
"""
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


class GradSync:
    def __init__(self, flat_grad, group=None, bucket_elems=16 * 1024 * 1024, base=0):
        """flat_grad: the slice of the flat gradient buffer that receives gradients (model.trainable_range()); `base` = its first
        element's offset in the whole buffer (ready_above() is called with whole-buffer offsets)."""
        self.grad = flat_grad
        self.base = int(base)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = flat_grad.numel()
        self.bounds = []
        hi = n
        while hi > 0:
            lo = max(0, hi - bucket_elems)
            self.bounds.append((lo, hi))     # ordered from the END of the buffer (backward order)
            hi = lo
        self.launched = 0
        self.works = []

    def begin(self):
        self.launched = 0
        self.works = []

    def ready_above(self, lo_offset):
        """All gradient elements at offsets >= lo_offset are final: launch every bucket above it."""
        if self.world == 1:
            return
        lo_offset = max(0, lo_offset - self.base)
        while self.launched < len(self.bounds) and self.bounds[self.launched][0] >= lo_offset:
            lo, hi = self.bounds[self.launched]
            self.works.append(dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.launched += 1

    def finish(self):
        """Launch what is left, wait for everything, return the 1/world scale the optimizer applies."""
        if self.world == 1:
            return 1.0
        self.ready_above(0)
        for w in self.works:
            w.wait()
        self.works = []
        self.launched = 0
        return 1.0 / self.world
