"""`wav2vec2_resnet_nll` model plugin — host-side mirror of model/wav2vec2_resnet_nll.py::Model.

    Model(args: dict with `flag_fix_ssl`, `contra_mode`, `loss_type` and a `resnet` section, device, is_train=True)
    forward(x [bz, L]) -> (logits [bz, 2], feats [bz, T, 128], emb [bz, 256])   (logits only when not is_train)
    loss(output, feats, emb, labels, config, info=None) -> dict of 0-d tensors

Encoder + LL + losses: HIP kernels; back-end: `resnet_head.ResNetHead` on flat-buffer parameter views, replayed as hipGraphs in
training (scl_amd/model_front.py).  Differences from the linear plugin that the reference has and this keeps
(wav2vec2_resnet_nll.py:91-121): CrossEntropy on raw logits WITHOUT the extra 1/bz, SupCon terms WITHOUT 1/bz.  The reference's
`loss` takes no `info` argument although main.py:68 passes one; it is accepted and ignored here.
"""
from .model_front import FrontHeadModel
from .model_linear import loss_custom
from .resnet_head import DEFAULT_RESNET, ResNetHead


class Model(FrontHeadModel):
    def __init__(self, args, device, is_train=True, w2v_cfg=None, seed=0, rank=0):
        super().__init__(args, device, is_train=is_train, w2v_cfg=w2v_cfg, seed=seed, rank=rank)
        self.flag_fix_ssl = bool(args.get("flag_fix_ssl", False))

    def trainable_range(self):
        return (self.P.off("LL.weight") if self.flag_fix_ssl else 0), self.P.n_train

    def _build_head(self, args):
        return ResNetHead(args.get("resnet") or DEFAULT_RESNET)

    _head_forward = staticmethod(ResNetHead.forward)

    def _ssl_train(self):
        # wav2vec2_resnet_nll.py:54-58 -> xlsr.py:23-42: eval mode when frozen, else train/eval follows `is_train`
        return False if self.flag_fix_ssl else bool(self.is_train and self.training)

    def loss(self, output, feats, emb, labels, config, info=None):
        bz = output.shape[0]
        terms = loss_custom(output, feats, emb, labels, config)   # the linear plugin's terms carry Model.loss's 1/bz (linear_nll:163-176)
        return {k: v * float(bz) for k, v in terms.items()}
