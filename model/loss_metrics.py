"""SupCon + NLL on the HIP path, under the reference's module name (model/loss_metrics.py:85-209, 498-532)."""
import torch

from scl_amd import ops
from scl_amd.model_linear import loss_custom  # noqa: F401


def supcon_loss(input_feat, labels=None, mask=None, sim_metric=None, t=0.07, contra_mode="all", length_norm=False):
    """supcon_loss(feat [bs, 1, T', d], labels) as Model.loss calls it (n_views = 1, 'all', sequence similarity);
    forward only — use loss_custom / Model.loss for training."""
    if mask is not None or contra_mode != "all" or length_norm or input_feat.shape[1] != 1:
        raise NotImplementedError("only the call pattern of Model.loss is accelerated")
    bz, _, Tq, dq = input_feat.shape
    F = input_feat.reshape(bz, Tq * dq).float().contiguous()
    ws = torch.empty(ops.supcon_nchunks(Tq * dq) * bz * bz, device=F.device)
    G, out = torch.empty(bz * bz, device=F.device), torch.empty(1, device=F.device)
    ops.supcon_fwd(F, labels.to(torch.int64).contiguous(), bz, Tq * dq, Tq * dq, Tq, float(t), ws, G, out)
    return out[0]
