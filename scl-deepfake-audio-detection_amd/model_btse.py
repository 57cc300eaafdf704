"""`wav2vec2_btse` model plugin — host-side mirror of model/wav2vec2_btse/model.py::Model (BASELINE.json configs[4]).

    Model(args: the YAML `model:` block of conf-5-btse-trans64.yaml, device, is_train=True)
    forward(x [bz, L], bio=None, bio_lengths=None, y=None) -> (log_probs [bz, 2], ssl_feat [bz, T, 128], b [bz, 128 + bio_out])
                                                              (log_probs only when not is_train)          (model.py:321-343)
    loss(output, feats, emb, labels, config, info=None) -> dict of 0-d tensors

What the plugin IS in the reference (not a Conformer): XLS-R -> LL 1024 -> 128 -> a 3-layer MLP, mean over time (backend.py:17-47,
linear.py:5-67) joined with a 3-layer relative-position transformer over "bio" tokens (model.py:210-238, transformer.py) -> fc2 ->
log_softmax.  State-dict keys equal the reference's: backend.ssl_model.model.<fairseq keys>, backend.LL.*, backend.mlp.m_frame_level.linear_{0,1,2}.*,
backend.mlp.m_utt_level.*, bioScoring.bio_embedding.weight, bioScoring.encoder.{attn_layers,norm_layers_1,ffn_layers,norm_layers_2}.<i>.*,
bioScoring.bio_scoring.*, [fc1.*,] fc2.*  (tests/test_btse_gpu.py compares them with the key list of the reference's own Model).

Two things the reference does not hold, and what this plugin does about them:
  * the TOKENISER.  model.py:306-319 calls `Wav2bioCNN.wav2bio` from a `biosegment` package that is a dangling symlink in the reference;
    `forward` takes the bio tokens as an argument (model.py:321), and that is the supported way in.  Without them, a tokeniser named by
    the optional YAML key `bio_tokenizer: "<module>:<callable>"` (signature (waveforms [bz, L] numpy, sample_rate) -> list of equal-length
    int sequences, as get_Bio expects) is called; with neither, forward raises.
  * `loss`.  model.py:345-375 reads self.loss_CE / self.contra_mode / self.sim_metric_seq, none of which is ever set: it cannot run.  This
    plugin uses the linear plugin's loss (model/wav2vec2_linear_nll.py:158-192), which is what that code was copied from.
The MLP's own logits (m_utt_level) are computed and dropped by the reference (model.py:324): here they are not computed, and the two
tensors sit behind the trainable range so that AdamW leaves them untouched exactly as torch.optim.AdamW skips a parameter without .grad.
The SSL encoder always runs in train mode (backend.py:29,39: the back-end's own `is_train` stays True whatever the outer flag says).
"""
import importlib

import numpy as np
import torch

from .btse_head import BtseHead
from .model_front import FrontHeadModel
from .model_linear import dropout_stream_seed, loss_custom


class Model(FrontHeadModel):
    front_prefix = "backend."

    def __init__(self, args, device, is_train=True, w2v_cfg=None, seed=0, rank=0):
        super().__init__(args, device, is_train=is_train, w2v_cfg=w2v_cfg, seed=seed, rank=rank)
        self.is_add = bool(self.btse_args["is_add"])
        self.use_graphs = False          # the head is one hand-scheduled autograd node already
        BtseHead.reseed(self, dropout_stream_seed(seed, rank) ^ 0x2B7E15)      # MLP dropout masks follow --seed and the data-parallel rank
        self.bio_tokenizer = None
        spec = args.get("bio_tokenizer") if hasattr(args, "get") else None
        if spec:
            mod, fn = spec.split(":")
            self.bio_tokenizer = getattr(importlib.import_module(mod), fn)

    def _build_head(self, args):
        head = BtseHead(args)
        self.btse_args = head.btse_args
        return head

    _head_forward = staticmethod(BtseHead.forward)

    def _ssl_train(self):
        return True      # backend.py:29,39 -> xlsr.py:30-31: self.model.train() on every call

    def get_Bio(self, X_pad, fs):
        """model.py:306-319, with the absent Wav2bioCNN replaced by the configured callable."""
        if self.bio_tokenizer is None:
            raise RuntimeError("wav2vec2_btse: forward() needs bio tokens (bio, bio_lengths): the reference's tokeniser package "
                               "`model/wav2vec2_btse/biosegment` is absent from the reference itself; pass the tokens, or name a "
                               "tokeniser with the YAML key model.bio_tokenizer = '<module>:<callable>'")
        toks = self.bio_tokenizer(X_pad.detach().cpu().numpy(), fs)
        lens = torch.tensor([len(t) for t in toks], dtype=torch.int32)
        return torch.from_numpy(np.asarray(toks, dtype=np.int32)), lens          # ragged lists fail here, as torch.IntTensor(bio) does

    def forward(self, x, bio=None, bio_lengths=None, y=None):
        if bio is None:
            bio, bio_lengths = self.get_Bio(x if x.dim() == 2 else x[:, :, 0], 16000)
        self.__dict__["_bio"] = (bio, bio_lengths)
        try:
            return super().forward(x)
        finally:
            self.__dict__["_bio"] = None

    def loss(self, output, feats, emb, labels, config, info=None):
        return loss_custom(output, feats, emb, labels, config)
