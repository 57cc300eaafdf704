"""`wav2vec2_linear_nll` model plugin — host-side mirror of the reference's
model/wav2vec2_linear_nll.py::Model (ctor, `is_train`, forward, loss), with every
numeric step executed by the HIP kernels behind the C ABI.

    Model(args: dict, device, is_train=True)
    forward(x [bz, L] fp32) -> (log_probs [bz,2], feats [bz,T,128], emb [bz,128])   (or log_probs when not is_train)
    loss(output, feats, emb, labels, config, info=None) -> dict of 0-d tensors, summed and .backward()-ed by the caller

State-dict keys equal the reference's (ssl_model.model.<fairseq keys>, LL.*, first_bn.*, first_bn1.*,
backend.m_frame_level.{0,3,6}.*, backend.m_utt_level.*), so its checkpoints load unchanged.
`train_loss.backward()` reaches the hand-written backward through two torch.autograd.Function
boundaries (model, loss); parameter gradients are written straight into the flat gradient buffer.
"""
import math

import torch
from torch import nn

from . import ops
from .encoder import Encoder, W2VConfig, param_specs
from .lib import ACT_LEAKY, ACT_RELU
from .ops import Op
from .params import FlatParams, register_by_name

import os

HEAD_DIM = 128
N_CLASS = 2
# scoring (model.eval() under torch.no_grad(): main.py --eval / --predict / --emb, the validation pass) runs fp32 end to end unless
# SCL_SCORE_FP32=0 asks for the bf16-operand training kernels (about 8x faster forward, scores to ~1e-2)
SCORE_FP32 = os.environ.get("SCL_SCORE_FP32", "1") != "0"
# the 128-wide frame-level head in f32 on the exact-fp32 GEMM (SCL_HEAD_F32=0: bf16 operands, the A/B switch of tests / probes)
HEAD_F32 = os.environ.get("SCL_HEAD_F32", "1") != "0"
X3_OFF = {"x3": False} if HEAD_F32 else {}      # the training head's f32 GEMMs stay on the exact kernel (its backward is checked at 4e-6)
DROP_P = 0.5  # BackEnd(128, 128, 2, 0.5, False): torch.nn.Dropout(0.5) after each frame-level layer


def head_specs(embed):
    return [("LL.weight", (HEAD_DIM, embed), True), ("LL.bias", (HEAD_DIM,), True),
            ("backend.m_frame_level.0.weight", (HEAD_DIM, HEAD_DIM), True), ("backend.m_frame_level.0.bias", (HEAD_DIM,), True),
            ("backend.m_frame_level.3.weight", (HEAD_DIM, HEAD_DIM), True), ("backend.m_frame_level.3.bias", (HEAD_DIM,), True),
            ("backend.m_frame_level.6.weight", (HEAD_DIM, HEAD_DIM), True), ("backend.m_frame_level.6.bias", (HEAD_DIM,), True),
            ("backend.m_utt_level.weight", (N_CLASS, HEAD_DIM), True), ("backend.m_utt_level.bias", (N_CLASS,), True)]


def init_parameters_(P, cfg, seed=0):
    """Seeded random init at the right shapes (no checkpoint ships with the repo): kaiming-normal convs
    and N(0, 0.02) linears for the encoder, torch's default uniform for the head."""
    g = torch.Generator().manual_seed(seed)
    for name, (o, n, shape, tr) in P.index.items():
        short = name.split("ssl_model.model.")[-1]
        if short.endswith("2.1.weight") or "layer_norm.weight" in short:
            t = torch.ones(shape)
        elif short.endswith("2.1.bias") or "layer_norm.bias" in short:
            t = torch.zeros(shape)
        elif "conv_layers" in short and short.endswith("0.weight"):
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[1] * shape[2]))
        elif short == "encoder.pos_conv.0.weight_v":
            t = torch.randn(shape, generator=g) * math.sqrt(4.0 / (cfg.pos_k * cfg.embed))
        elif short == "encoder.pos_conv.0.weight_g":
            continue
        elif name.startswith("LL.") or name.startswith("backend."):
            fan_in = shape[-1] if len(shape) > 1 else HEAD_DIM
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        elif short.endswith(".bias"):
            t = torch.zeros(shape)
        else:
            t = 0.02 * torch.randn(shape, generator=g)
        P.f32(name).copy_(t.to(P.device))
    v = P.f32("ssl_model.model.encoder.pos_conv.0.weight_v")
    P.f32("ssl_model.model.encoder.pos_conv.0.weight_g").copy_(v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
    P.mark_dirty()


def dropout_stream_seed(seed, rank):
    return (int(seed) * 2654435761 + int(rank) * 40503 + 12345) & 0x7FFFFFFF


def maybe_load_pretrained(model, args):
    """The reference's SSLModel loads pretrained/xlsr2_300m.pt at construction (model/xlsr.py:14-16).  Do the same when the file
    (or the YAML's optional `pretrained:` path) exists; take encoder_layerdrop from its cfg unless the YAML overrides it."""
    import os
    from . import checkpoint
    explicit = hasattr(args, "get") and args.get("pretrained")
    path = explicit or checkpoint.DEFAULT_PRETRAINED
    if not os.path.exists(path) or (not explicit and model.cfg.layers != 24):     # toy encoders never pick up the default file
        return False
    probs = checkpoint.load_pretrained_into(model, path)
    if "encoder_layerdrop" in probs and "encoder_layerdrop" not in args:
        model.cfg.encoder_layerdrop = probs["encoder_layerdrop"]
    print("[scl] XLS-R weights loaded from %s (encoder_layerdrop %.3f)" % (path, model.cfg.encoder_layerdrop))
    return True


class _ModelFn(torch.autograd.Function):
    """Autograd boundary around the whole network: forward runs the kernels and keeps the activation
    buffers; backward consumes (d_logp, d_feats, d_emb) and fills the flat gradient buffer."""

    @staticmethod
    def forward(ctx, model, x, anchor):
        out, feats, emb, saved = model._run_forward(x)
        ctx.model, ctx.saved = model, saved
        # the outputs live in static per-(B, L) buffers (launch plans point at them): hand out copies, so results
        # of successive forward calls do not alias (3 small device copies)
        return out.clone(), feats.clone(), emb.clone()

    @staticmethod
    def backward(ctx, d_out, d_feats, d_emb):
        ctx.model._run_backward(ctx.saved, d_out, d_feats, d_emb)
        return None, None, None


class Model(nn.Module):
    def __init__(self, args, device, is_train=True, w2v_cfg=None, seed=0, rank=0):
        super().__init__()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("scl_amd.Model needs an MI355X device: the product path has no CPU fallback")
        self.is_train = is_train
        self.flag_fix_ssl = args["flag_fix_ssl"]
        self.contra_mode = args["contra_mode"]
        self.loss_type = args["loss_type"]
        if self.contra_mode not in ("all", "one"):
            raise ValueError("Unknown mode: {}".format(self.contra_mode))     # loss_metrics.py:161
        if w2v_cfg is None:
            # optional YAML key `w2v_arch`: "xlsr_300m" (default, the reference's only encoder) or "tiny" (tests / plumbing)
            w2v_cfg = W2VConfig.tiny() if args.get("w2v_arch", "xlsr_300m") == "tiny" else \
                W2VConfig(encoder_layerdrop=float(args.get("encoder_layerdrop", 0.0)))
        self.cfg = w2v_cfg
        specs = param_specs(self.cfg) + head_specs(self.cfg.embed)
        # memory order: trainable encoder, head, then the checkpoint-only tensors (FlatParams puts those last)
        self.P = FlatParams(specs, self.device)
        for name, p in self.P.params.items():
            register_by_name(self, name, p)
        # defined-but-unused modules of the reference (linear_nll:108-109) — state-dict compatibility only
        self.first_bn = nn.BatchNorm2d(num_features=1).to(self.device)
        self.first_bn1 = nn.BatchNorm2d(num_features=64).to(self.device)
        init_parameters_(self.P, self.cfg, seed)
        self.encoder = Encoder(self.cfg, self.P)
        self._anchor = torch.zeros((), device=self.device, requires_grad=True)
        self._hbufs = {}
        self._states = {}
        self._step_seed = dropout_stream_seed(seed, rank)   # head dropout masks differ per --seed and per data-parallel rank
        self.out_dim = self.cfg.embed
        self.grad_sync = None   # scl_amd.parallel.GradSync when data-parallel (set by FusedAdamW)
        self.pretrained_loaded = maybe_load_pretrained(self, args)

    def trainable_range(self):
        """[lo, hi) of the flat buffer that receives gradients: with flag_fix_ssl the encoder backward is skipped, and
        torch.optim.AdamW in the reference leaves parameters whose .grad is None untouched (no weight decay, no all-reduce)."""
        return (self.P.off("LL.weight") if self.flag_fix_ssl else 0), self.P.n_train

    # nn.Module plumbing ----------------------------------------------------------------------------
    def _apply(self, fn, recurse=True):
        # .to(device) / .cuda() on an already-placed model must not break the flat-buffer views
        probe = fn(torch.zeros(1, device=self.device))
        if probe.device != self.device or probe.dtype != torch.float32:
            raise RuntimeError("scl_amd.Model lives in one flat fp32 device buffer; construct it on the target device")
        return self

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.P.mark_dirty()
        return r

    def optimizer_stepped(self, bf16_fresh):
        self.P.mark_dirty()
        if bf16_fresh:
            self.P.bf16_version = self.P.version

    # forward ---------------------------------------------------------------------------------------
    def _head_bufs(self, B, T):
        key = (B, T)
        if key not in self._hbufs:
            M, dev = B * T, self.device
            bf = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
            f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            # the frame-level head (three 128 x 128 linears) runs in f32 on the exact-fp32 GEMM with the fp32 master weights, forward and
            # backward: 3.8 GFLOP per step at batch 64, and its gradients then meet the same bound as every other tensor (the
            # mean-pool backward hands every frame of an utterance the SAME row — rounded to bf16 its error is systematic over T)
            hd = f32 if HEAD_F32 else bf
            self._hbufs[key] = dict(r0=hd(M * HEAD_DIM + 1024), pre=[hd(M * HEAD_DIM) for _ in range(3)],
                                    h=[hd(M * HEAD_DIM + 1024) for _ in range(3)], dpre=[hd(M * HEAD_DIM + 1024) for _ in range(3)],
                                    dfe=f32(M * HEAD_DIM), dfe_bf=bf(M * HEAD_DIM + 1024), denc=bf(M * self.cfg.embed),
                                    demb=f32(B * HEAD_DIM), ws=f32(B * N_CLASS + 16),
                                    cs=f32(ops.colsum_reduce_nparts(M, 8) * max(HEAD_DIM, 8)), dW=f32(HEAD_DIM * max(HEAD_DIM, self.cfg.embed)))
        return self._hbufs[key]

    def _state(self, B, L):
        """Static per-(B, L) buffers at the autograd boundary, so recorded launch plans stay valid."""
        key = (B, L)
        if key not in self._states:
            T = self.cfg.conv_lens(L)[-1]
            f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.device)
            self._states[key] = dict(T=T, x=f32(B, L), feats=f32(B, T, HEAD_DIM), emb=f32(B, HEAD_DIM), logp=f32(B, N_CLASS),
                                     d_logp=f32(B, N_CLASS), d_feats=f32(B, T, HEAD_DIM), d_emb=f32(B, HEAD_DIM), plans={})
        return self._states[key]

    def _run_forward(self, x):
        B, L = x.shape
        training = bool(self.training)
        ssl_train = False if self.flag_fix_ssl else bool(self.is_train and training)
        st = self._state(B, L)
        st["x"].copy_(x)
        drop = DROP_P if training else 0.0
        self._step_seed = (self._step_seed * 1664525 + 1013904223) & 0x7FFFFFFF
        seeds = [(self._step_seed + 7919 * j) & 0x7FFFFFFF for j in range(3)]
        self.encoder.refresh_weights()
        use_plan = self.cfg.encoder_layerdrop == 0 or not ssl_train
        pk = ("fwd", training, ssl_train)
        plan = st["plans"].get(pk) if use_plan else None
        if plan is not None:
            for dsc, sd in zip(plan["drop_descs"], seeds):
                dsc.drop_seed = sd
            ectx = plan["saved"]["ectx"]
            self.encoder.apply_seeds(ectx["drop_slots"], self._step_seed)      # the encoder's element-dropout sites (none at p = 0)
            ops.replay(plan["calls"])
            saved = dict(plan["saved"], seeds=seeds, ectx=dict(ectx, step_seed=self._step_seed))
            return st["logp"], st["feats"], st["emb"], saved
        if use_plan:
            ops.start_recording()
        saved, drop_descs = self._forward_kernels(st, B, L, ssl_train, drop, seeds)
        if use_plan:
            st["plans"][pk] = dict(calls=ops.stop_recording(), drop_descs=drop_descs, saved=saved)
        return st["logp"], st["feats"], st["emb"], dict(saved, seeds=seeds)

    def _forward_kernels(self, st, B, L, ssl_train, drop, seeds):
        P, E = self.P, self.cfg.embed
        enc_out, ectx = self.encoder.forward(st["x"], training=ssl_train, refresh=False, step_seed=self._step_seed)
        T = ectx["d"]["T"]
        M = B * T
        hb = self._head_bufs(B, T)
        feats, emb, logp = st["feats"], st["emb"], st["logp"]
        W = lambda name, ld: Op(P.bf16, ld, offset=P.off(name))
        # feats = LL(x) (pre-ReLU tensor is what SupCon sees, linear_nll:127-129), r0 = relu(feats)
        ops.gemm(Op(enc_out, E), W("LL.weight", E), hb["r0"], M, HEAD_DIM, E, bias=P.f32("LL.bias"), act=ACT_RELU, c2=feats)
        Wf = (lambda name, ld: Op(P.flat, ld, offset=P.off(name))) if HEAD_F32 else W
        prev, drop_descs = hb["r0"], []
        for j, idx in enumerate((0, 3, 6)):
            dsc = ops.gemm(Op(prev, HEAD_DIM), Wf("backend.m_frame_level.%d.weight" % idx, HEAD_DIM), hb["h"][j], M, HEAD_DIM, HEAD_DIM,
                           bias=P.f32("backend.m_frame_level.%d.bias" % idx), act=ACT_LEAKY, c2=hb["pre"][j], drop_p=drop,
                           drop_seed=seeds[j], x3=False)
            drop_descs.append(dsc)
            prev = hb["h"][j]
        ops.meanpool_fwd(prev, emb, B, T, HEAD_DIM)
        ops.utt_head_fwd(emb, P.f32("backend.m_utt_level.weight"), P.f32("backend.m_utt_level.bias"), logp, B, HEAD_DIM, N_CLASS)
        saved = dict(ectx=ectx, hb=hb, B=B, L=L, T=T, drop=drop, feats=feats, emb=emb, logp=logp, enc_out=enc_out, st=st)
        return saved, drop_descs

    def _run_backward(self, sv, d_logp, d_feats, d_emb):
        P = self.P
        P.rebind_grads()
        st, seeds = sv["st"], sv["seeds"]
        for name, g in (("d_logp", d_logp), ("d_feats", d_feats), ("d_emb", d_emb)):
            if g is None:
                st[name].zero_()
            else:
                st[name].copy_(g)
        if self.grad_sync is not None:
            self.encoder.on_grads_ready = self.grad_sync.ready_above
        use_plan = not sv["ectx"]["skipped"] and self.cfg.encoder_layerdrop == 0
        pk = ("bwd", sv["drop"] > 0, self.grad_sync is not None)
        plan = st["plans"].get(pk) if use_plan else None
        if plan is not None:
            plan["drop_descs"][0].drop_seed = seeds[1]
            plan["drop_descs"][1].drop_seed = seeds[0]
            plan["meanpool_entry"][1][8] = seeds[2]
            self.encoder.apply_seeds(plan["enc_slots"], sv["ectx"]["step_seed"])
            ops.replay(plan["calls"])
            return
        if use_plan:
            ops.start_recording()
        extra = self._backward_kernels(sv, st, seeds)
        if use_plan:
            st["plans"][pk] = dict(calls=ops.stop_recording(), **extra)

    def _backward_kernels(self, sv, st, seeds):
        P, E = self.P, self.cfg.embed
        B, T, hb = sv["B"], sv["T"], sv["hb"]
        M = B * T
        W = lambda name, ld: Op(P.bf16, ld, offset=P.off(name))
        ops.utt_head_bwd(st["d_logp"], sv["logp"], sv["emb"], P.f32("backend.m_utt_level.weight"), st["d_emb"], hb["demb"],
                         P.g("backend.m_utt_level.weight"), P.g("backend.m_utt_level.bias"), hb["ws"], B, HEAD_DIM, N_CLASS)
        mp_entry = ops.meanpool_bwd(hb["demb"], hb["pre"][2], hb["dpre"][2], B, T, HEAD_DIM, ACT_LEAKY, sv["drop"], seeds[2])
        drop_descs, enc_slots = [], []
        for j, idx in reversed(list(enumerate((0, 3, 6)))):
            dpre = hb["dpre"][j]
            inp = hb["h"][j - 1] if j > 0 else hb["r0"]
            ops.colsum_reduce(dpre, hb["cs"], P.g("backend.m_frame_level.%d.bias" % idx), M, HEAD_DIM)
            self.encoder._wgrad(sv["ectx"]["d"], Op(dpre, HEAD_DIM), Op(inp, HEAD_DIM), P.g("backend.m_frame_level.%d.weight" % idx),
                                HEAD_DIM, HEAD_DIM, M, **X3_OFF)
            # fp32 master weights on the exact-fp32 GEMM (bf16 working copy when SCL_HEAD_F32=0)
            wj = Op(P.flat if HEAD_F32 else P.bf16, HEAD_DIM, offset=P.off("backend.m_frame_level.%d.weight" % idx))
            if j > 0:
                drop_descs.append(ops.gemm(Op(dpre, HEAD_DIM), wj, hb["dpre"][j - 1], M, HEAD_DIM, HEAD_DIM, b_t=True, R=hb["pre"][j - 1],
                                           rmode=2, ract=ACT_LEAKY, drop_p=sv["drop"], drop_seed=seeds[j - 1], x3=False))
            else:
                ops.gemm(Op(dpre, HEAD_DIM), wj, hb["dfe"], M, HEAD_DIM, HEAD_DIM, b_t=True, R=sv["feats"], rmode=2, ract=ACT_RELU, x3=False)
        # total gradient at feats = ReLU path + SupCon path
        ops.add_f32(hb["dfe"], st["d_feats"], hb["dfe"], hb["dfe_bf"], M * HEAD_DIM)
        ops.colsum_reduce(hb["dfe"], hb["cs"], P.g("LL.bias"), M, HEAD_DIM)
        self.encoder._wgrad(sv["ectx"]["d"], Op(hb["dfe_bf"], HEAD_DIM), Op(sv["enc_out"], E), P.g("LL.weight"), HEAD_DIM, E, M)
        if not self.flag_fix_ssl:
            ops.gemm(Op(hb["dfe_bf"], HEAD_DIM), W("LL.weight", E), hb["denc"], M, E, HEAD_DIM, b_t=True)
            if self.grad_sync is not None:      # head gradients (the END of the flat buffer) are final: start their all-reduce
                ops.host_callback(self.grad_sync.ready_above, P.off("LL.weight"))
            enc_slots = self.encoder.backward(sv["ectx"], hb["denc"])
        return dict(drop_descs=drop_descs, meanpool_entry=mp_entry, enc_slots=enc_slots)

    def _score_fp32(self, x):
        """Scoring forward (no grad, eval mode): fp32 activations, fp32 master weights, exact-fp32 GEMMs end to end — the
        reference's precision (main.py:161-214), for scores / embeddings within 1e-3 of it."""
        P, E = self.P, self.cfg.embed
        B = x.shape[0]
        enc, T = self.encoder.forward_f32(x)
        M = B * T
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.device)
        Wf = lambda name, ld: Op(P.flat, ld, offset=P.off(name))
        feats, h0, h1 = f32(B, T, HEAD_DIM), f32(M, HEAD_DIM), f32(M, HEAD_DIM)
        ops.gemm(Op(enc, E), Wf("LL.weight", E), h0, M, HEAD_DIM, E, bias=P.f32("LL.bias"), act=ACT_RELU, c2=feats)
        for idx in (0, 3, 6):
            ops.gemm(Op(h0, HEAD_DIM), Wf("backend.m_frame_level.%d.weight" % idx, HEAD_DIM), h1, M, HEAD_DIM, HEAD_DIM,
                     bias=P.f32("backend.m_frame_level.%d.bias" % idx), act=ACT_LEAKY)
            h0, h1 = h1, h0
        emb = h0.view(B, T, HEAD_DIM).mean(dim=1)
        logp = f32(B, N_CLASS)
        ops.utt_head_fwd(emb.contiguous(), P.f32("backend.m_utt_level.weight"), P.f32("backend.m_utt_level.bias"), logp, B, HEAD_DIM, N_CLASS)
        return logp, feats, emb

    def _forward(self, x):
        if x.dim() == 3:
            x = x[:, :, 0]
        x = x.to(device=self.device, dtype=torch.float32).contiguous()   # main.py:60 hands over a transposed view
        if torch.is_grad_enabled() and any(p.requires_grad for p in (self._anchor,)):
            out, feats, emb = _ModelFn.apply(self, x, self._anchor)
        elif not self.training and SCORE_FP32:
            out, feats, emb = self._score_fp32(x)
        else:
            out, feats, emb, _ = self._run_forward(x)
            out, feats, emb = out.clone(), feats.clone(), emb.clone()
        if self.is_train:
            return out, feats, emb
        return out

    def forward(self, x_big):
        if not self.is_train:
            print("Inference mode")  # linear_nll:152
        return self._forward(x_big)

    # loss ------------------------------------------------------------------------------------------
    def loss(self, output, feats, emb, labels, config, info=None):
        return loss_custom(output, feats, emb, labels, config)


_LOSS_SCALE = {}


def _loss_scale(bz, dev):
    """[1, 1/bz, 1/bz] on the device, built once per (batch, device): torch.tensor(..., device=) is a blocking host-to-device copy, and
    one per step in the middle of the forward -> backward hand-over stalled the launch thread until the GPU had caught up."""
    key = (bz, dev.type, dev.index)
    if key not in _LOSS_SCALE:
        _LOSS_SCALE[key] = torch.tensor([1.0, 1.0 / bz, 1.0 / bz], device=dev)
    return _LOSS_SCALE[key]


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, feats, emb, labels):
        dev = output.device
        bz = output.shape[0]
        Tq, dq = feats.shape[1], feats.shape[2]
        res = torch.empty(3, device=dev)
        coef = torch.empty(bz, output.shape[1], device=dev)
        labels = labels.to(torch.int64).contiguous()
        outc, F1, F2 = output.contiguous(), feats.contiguous().view(bz, -1), emb.contiguous()
        ops.nll_fwd(outc, labels, bz, output.shape[1], res[0:1], coef)
        K1, K2 = Tq * dq, emb.shape[1]
        ws = torch.empty(max(ops.supcon_ws_floats(bz, K1), ops.supcon_ws_floats(bz, K2)), device=dev)
        G1, G2 = torch.empty(2 * bz * bz, device=dev), torch.empty(2 * bz * bz, device=dev)      # dL/dS + the backward's scratch
        ops.supcon_fwd(F1, labels, bz, K1, K1, Tq, 0.07, ws, G1, res[1:2])
        ops.supcon_fwd(F2, labels, bz, K2, K2, K2, 0.07, ws, G2, res[2:3])   # emb as [bz,1,128,1]: T' = 128, d = 1
        ctx.save_for_backward(coef, G1, G2, F1, F2)
        ctx.dims = (bz, Tq, dq, K1, K2)
        scale = _loss_scale(bz, dev)   # Model.loss multiplies the SupCon terms by 1/bz
        out = res * scale
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_ce, g_cf1, g_cf2):
        coef, G1, G2, F1, F2 = ctx.saved_tensors
        bz, Tq, dq, K1, K2 = ctx.dims
        dev = coef.device
        d_out = coef * g_ce
        dF1 = torch.empty(bz, K1, device=dev)
        dF2 = torch.empty(bz, K2, device=dev)
        ops.supcon_bwd(F1, G1, g_cf1.contiguous().view(1), 1.0 / bz, bz, K1, K1, Tq, 0.07, dF1)
        ops.supcon_bwd(F2, G2, g_cf2.contiguous().view(1), 1.0 / bz, bz, K2, K2, K2, 0.07, dF2)
        return d_out, dF1.view(bz, Tq, dq), dF2, None


GLOBAL_SUPCON = os.environ.get("SCL_GLOBAL_SUPCON", "0") == "1"


def loss_custom(output, feats, emb, labels, config):
    """model/loss_metrics.py:498-532 / Model.loss: dict of loss terms selected by loss_type.
    SCL_GLOBAL_SUPCON=1 under data parallelism (optional): the loss nn.DataParallel computes (main.py:62-66: Model.loss on
    the outputs of ALL GPUs) — the two SupCon terms see the batch of all ranks (features and labels all-gathered, positives / negatives
    across ranks, 1/bz with the global bz), and the NLL term, a sum over utterances, is the rank's own share of CE_global / bz_global
    (= its local term / world size: averaged over the ranks by the gradient exchange it is the global term).  Default: every term
    rank-local, the reference's semantics at --batch_size 1."""
    L_CE, L_CF1, L_CF2 = _LossFn.apply(output, feats, emb, labels)
    if GLOBAL_SUPCON and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        from .parallel import gather_for_global_loss as gather
        _, L_CF1, L_CF2 = _LossFn.apply(gather(output.detach()), gather(feats), gather(emb), gather(labels))
        L_CE = L_CE / torch.distributed.get_world_size()
    lt = config["model"]["loss_type"]
    # contra_mode 'one' anchors only the first VIEW (loss_metrics.py:155-157); Model.loss hands supcon_loss exactly one view per
    # utterance (feats.unsqueeze(1), wav2vec2_linear_nll.py:176-180), so anchor_feature == contrast_feature and anchor_count == nv == 1:
    # both modes are the same computation here
    if config["model"].get("contra_mode", "all") not in ("all", "one"):
        raise ValueError("Unknown mode: {}".format(config["model"]["contra_mode"]))
    if lt == 1:
        return {"L_CE": L_CE, "L_CF1": L_CF1, "L_CF2": L_CF2}
    if lt == 2:
        return {"L_CE": L_CE, "L_CF1": L_CF1}
    if lt == 3:
        return {"L_CE": L_CE, "L_CF2": L_CF2}
    if lt == 4:
        return {"L_CE": L_CE}
    if lt == 5:
        return {"L_CF1": L_CF1, "L_CF2": L_CF2}
    return None
