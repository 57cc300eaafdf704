"""`wav2vec2_aasist` model plugin — host-side mirror of model/wav2vec2_aasist.py::Model.

    Model(args: dict with an `aasist` section, device, is_train=True)
    forward(x [bz, L]) -> (logits [bz, 2], feats [bz, T, 128], last_hidden [bz, 160])   (logits only when not is_train)
    loss(output, feats, emb, labels, config, info=None) -> dict of 0-d tensors

Encoder + LL + losses: HIP kernels; graph back-end: `aasist_head.AasistHead` on flat-buffer parameter views, replayed as
hipGraphs in training (scl_amd/model_front.py).

The reference's train-mode forward returns a 2-tuple while main.py:63 unpacks three values: this plugin
returns the triple main.py and Model.loss need — feats is the LL output, exactly what the linear plugin hands to SupCon.
"""
from .aasist_head import UPSTREAM_AASIST, AasistHead
from .model_front import FrontHeadModel
from .model_linear import dropout_stream_seed, loss_custom


class Model(FrontHeadModel):
    def __init__(self, args, device, is_train=True, w2v_cfg=None, seed=0, rank=0):
        super().__init__(args, device, is_train=is_train, w2v_cfg=w2v_cfg, seed=seed, rank=rank)
        # the fused graph module draws its dropout masks (input_drop, GraphPool, drop_way, read-out) from a counter-hash stream:
        # start it from --seed and the data-parallel rank, like the encoder's (model_front.py)
        from . import graph
        graph.seed(dropout_stream_seed(seed, rank) ^ 0x3C6EF372)

    def _build_head(self, args):
        return AasistHead(args.get("aasist") or UPSTREAM_AASIST)

    _head_forward = staticmethod(AasistHead.forward)

    # aasist.py:33-52: the SSL model simply follows the parent's train / eval mode (FrontHeadModel._ssl_train default)

    def loss(self, output, feats, emb, labels, config, info=None):
        # aasist.py:607-640: unweighted CrossEntropy on raw logits (the NLL kernel folds the log-softmax), SupCon over the
        # frame features and over last_hidden viewed as [bz, 1, 160, 1]
        return loss_custom(output, feats, emb, labels, config)
