"""HIP front end (XLS-R encoder + LL projection, losses) in front of a back-end module: the shared machinery of the
`wav2vec2_aasist`, `wav2vec2_resnet_nll` and `wav2vec2_btse` plugins.

What runs where: the encoder, the LL projection and the three loss terms are the HIP kernels of the wav2vec2_linear_nll
path (one autograd boundary around encoder + LL, launch plans recorded per (B, L)).  The back-end on the [bz, T, 128]
features is an nn.Module (a parameter container whose forward is hand-written HIP behind its own autograd Function: resstack.py +
graph.py, resnet_head.py, btse_head.py) whose parameters are views into the same flat fp32 buffer, so the fused AdamW kernel and the
data-parallel gradient buckets cover them too.  SCL_HEAD_GRAPH=1 replays a back-end as two captured hipGraphs per feature shape (off by
default: the HIP back-ends are a few dozen launches).
Sub-classes supply `_build_head(args)` (an nn.Module whose children / parameters are grafted at the root under the
reference's state-dict names) and `_head_forward(mod, feats) -> (output, emb)`.
"""
import os

import torch
from torch import nn

from . import ops
from .encoder import Encoder, W2VConfig, param_specs
from .model_linear import SCORE_FP32, dropout_stream_seed, init_parameters_, loss_custom, maybe_load_pretrained
from .ops import Op
from .params import FlatParams, register_by_name

FEAT_DIM = 128


class _FrontFn(torch.autograd.Function):
    """Autograd boundary around encoder + LL: waveform -> feats [bz, T, 128] fp32."""

    @staticmethod
    def forward(ctx, model, x, anchor):
        feats, saved = model._front_forward(x)
        ctx.model, ctx.saved = model, saved
        return feats.clone()

    @staticmethod
    def backward(ctx, d_feats):
        ctx.model._front_backward(ctx.saved, d_feats)
        return None, None, None


class _HeadRunner(nn.Module):
    """The graph back-end's sub-modules and parameters (shared objects, not copies) under one nn.Module: the unit that
    torch.cuda.make_graphed_callables captures into a forward and a backward hipGraph."""

    def __init__(self, owner):
        super().__init__()
        for n in owner._head_children:
            self.add_module(n, owner._modules[n])
        for n in owner._head_root_params:
            self.register_parameter(n, owner._parameters[n])
        self.__dict__["_fwd"] = owner._head_forward

    def forward(self, feats):
        return self._fwd(self, feats)


class FrontHeadModel(nn.Module):
    flag_fix_ssl = False
    front_prefix = ""        # module path of `ssl_model` / `LL` in the state dict ("backend." for wav2vec2_btse, backend.py:31-33)

    def _build_head(self, args):
        raise NotImplementedError

    @staticmethod
    def _head_forward(mod, feats):
        raise NotImplementedError

    def _ssl_train(self):
        return bool(self.training)

    def __init__(self, args, device, is_train=True, w2v_cfg=None, seed=0, rank=0):
        super().__init__()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("scl_amd.Model needs an MI355X device: the product path has no CPU fallback")
        self.is_train = is_train
        self.contra_mode = args.get("contra_mode", "all")
        self.loss_type = args.get("loss_type", 1)
        if w2v_cfg is None:
            w2v_cfg = W2VConfig.tiny() if args.get("w2v_arch", "xlsr_300m") == "tiny" else \
                W2VConfig(encoder_layerdrop=float(args.get("encoder_layerdrop", 0.0)))
        self.cfg = w2v_cfg
        rng_state = torch.get_rng_state()          # seeded head init without disturbing the caller's RNG stream
        torch.manual_seed(seed + 1)
        head = self._build_head(args)
        torch.set_rng_state(rng_state)
        head_params = list(head.named_parameters())
        specs = param_specs(self.cfg) + [("LL.weight", (FEAT_DIM, self.cfg.embed), True), ("LL.bias", (FEAT_DIM,), True)] + \
            [(n, tuple(p.shape), n not in getattr(head, "frozen_names", ())) for n, p in head_params]      # frozen: no gradient in the reference
        self.P = FlatParams(specs, self.device)
        self._head_lo = self.P.off(next(n for n, _ in head_params if n not in getattr(head, "frozen_names", ())))
        init_parameters_(self.P, self.cfg, seed)
        for n, p in head_params:
            self.P.f32(n).copy_(p.detach().to(self.device))
        # graft the head's sub-modules at the root (reference state-dict names: encoder.0.0.conv1.weight, GAT_layer_S.…) and
        # swap every head parameter for its flat-buffer view
        head.to(self.device)
        self._head_children = [n for n, _ in head.named_children()]
        self._head_root_params = list(head._parameters.keys())
        for n, child in head.named_children():
            self.add_module(n, child)
        for name, p in self.P.params.items():
            if name.startswith("ssl_model.") or name.startswith("LL."):
                register_by_name(self, self.front_prefix + name, p)
            else:
                mod = self
                parts = name.split(".")
                for part in parts[:-1]:
                    mod = mod._modules[part]
                mod._parameters[parts[-1]] = p
        self.P.mark_dirty()
        self.ssl = Encoder(self.cfg, self.P)
        self._anchor = torch.zeros((), device=self.device, requires_grad=True)
        self._states = {}
        self.out_dim = self.cfg.embed
        self.grad_sync = None
        self._drop_step = dropout_stream_seed(seed, rank)      # encoder element-dropout masks differ per --seed and per data-parallel rank
        # Optional (SCL_HEAD_GRAPH=1): replay the back-end's training forward / backward as two captured hipGraphs per feature
        # shape.  Both HIP back-ends capture and replay correctly (tests/test_aasist_gpu.py), but the step is kernel-bound: AASIST
        # 45.9 -> 45.5 ms/step at batch 32, ResNet +-0.  Off by default.
        self.use_graphs = os.environ.get("SCL_HEAD_GRAPH", "0") == "1"
        self.__dict__["_graphed"] = {}
        self.pretrained_loaded = maybe_load_pretrained(self, args)
        if rank:      # torch dropout in the back-end: decorrelate the data-parallel ranks' masks
            torch.cuda.manual_seed(int(seed) * 1000003 + int(rank))

    def trainable_range(self):
        return 0, self.P.n_train

    # nn.Module plumbing ----------------------------------------------------------------------------
    def _apply(self, fn, recurse=True):
        probe = fn(torch.zeros(1, device=self.device))
        if probe.device != self.device or probe.dtype != torch.float32:
            raise RuntimeError("scl_amd.Model lives in one flat fp32 device buffer; construct it on the target device")
        return self

    def load_state_dict(self, state_dict, strict=True):
        from . import hipnn
        r = super().load_state_dict(state_dict, strict=strict)
        hipnn.weights_changed()
        self.P.mark_dirty()
        return r

    def optimizer_stepped(self, bf16_fresh):
        from . import hipnn
        hipnn.weights_changed()      # the back-end's re-laid-out convolution weights are rebuilt on next use
        self.P.mark_dirty()
        if bf16_fresh:
            self.P.bf16_version = self.P.version

    def zero_torch_grads(self):
        """The head's gradients are ACCUMULATED by torch autograd into the flat gradient buffer (the HIP backward
        overwrites its part), so FusedAdamW.zero_grad clears that 0.4 M-element tail."""
        self.P.rebind_grads()
        self.P.grad[self._head_lo:].zero_()

    # encoder + LL ----------------------------------------------------------------------------------
    def _state(self, B, L):
        key = (B, L)
        if key not in self._states:
            T = self.cfg.conv_lens(L)[-1]
            M, E, dev = B * T, self.cfg.embed, self.device
            f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            bf = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
            self._states[key] = dict(T=T, x=f32(B, L), feats=f32(B, T, FEAT_DIM), d_feats=f32(B, T, FEAT_DIM),
                                     dfe_bf=bf(M * FEAT_DIM + 1024), denc=bf(M * E), cs=f32(ops.colsum_reduce_nparts(M, 8) * FEAT_DIM), plans={})
        return self._states[key]

    def _front_forward(self, x):
        B, L = x.shape
        ssl_train = self._ssl_train()
        st = self._state(B, L)
        st["x"].copy_(x)
        self.ssl.refresh_weights()
        use_plan = self.cfg.encoder_layerdrop == 0 or not ssl_train
        pk = ("fwd", ssl_train)
        plan = st["plans"].get(pk) if use_plan else None
        self._drop_step = (self._drop_step * 1664525 + 1013904223) & 0x7FFFFFFF
        if plan is not None:
            ectx = plan["saved"]["ectx"]
            self.ssl.apply_seeds(ectx["drop_slots"], self._drop_step)      # the encoder's element-dropout sites (none at p = 0)
            ops.replay(plan["calls"])
            return st["feats"], dict(plan["saved"], ectx=dict(ectx, step_seed=self._drop_step))
        if use_plan:
            ops.start_recording()
        P, E = self.P, self.cfg.embed
        enc_out, ectx = self.ssl.forward(st["x"], training=ssl_train, refresh=False, step_seed=self._drop_step)
        M = B * st["T"]
        # the bf16 primary output of the GEMM is not needed here: it lands in dfe_bf, which the backward overwrites
        ops.gemm(Op(enc_out, E), Op(P.bf16, E, offset=P.off("LL.weight")), st["dfe_bf"], M, FEAT_DIM, E, bias=P.f32("LL.bias"),
                 c2=st["feats"])
        saved = dict(ectx=ectx, st=st, enc_out=enc_out, B=B)
        if use_plan:
            st["plans"][pk] = dict(calls=ops.stop_recording(), saved=saved)
        return st["feats"], saved

    def _front_backward(self, sv, d_feats):
        P, E = self.P, self.cfg.embed
        P.rebind_grads()
        st = sv["st"]
        st["d_feats"].copy_(d_feats)
        if self.grad_sync is not None:
            self.ssl.on_grads_ready = self.grad_sync.ready_above
        use_plan = not sv["ectx"]["skipped"] and self.cfg.encoder_layerdrop == 0
        pk = ("bwd", self.grad_sync is not None)
        plan = st["plans"].get(pk) if use_plan else None
        if plan is not None:
            self.ssl.apply_seeds(plan["enc_slots"], sv["ectx"]["step_seed"])
            ops.replay(plan["calls"])
            return
        if use_plan:
            ops.start_recording()
        enc_slots = []
        M = sv["B"] * st["T"]
        ops.cast_bf16(st["d_feats"], st["dfe_bf"], M * FEAT_DIM)
        ops.colsum_reduce(st["d_feats"], st["cs"], P.g("LL.bias"), M, FEAT_DIM)
        self.ssl._wgrad(sv["ectx"]["d"], Op(st["dfe_bf"], FEAT_DIM), Op(sv["enc_out"], E), P.g("LL.weight"), FEAT_DIM, E, M)
        if not self.flag_fix_ssl:
            ops.gemm(Op(st["dfe_bf"], FEAT_DIM), Op(P.bf16, E, offset=P.off("LL.weight")), st["denc"], M, E, FEAT_DIM, b_t=True)
            if self.grad_sync is not None:     # LL and the torch head's gradients (the END of the flat buffer) are final here
                ops.host_callback(self.grad_sync.ready_above, P.off("LL.weight"))
            enc_slots = self.ssl.backward(sv["ectx"], st["denc"])
        if use_plan:
            st["plans"][pk] = dict(calls=ops.stop_recording(), enc_slots=enc_slots)

    # forward / loss --------------------------------------------------------------------------------
    def forward(self, x):
        if x.dim() == 3:
            x = x[:, :, 0]
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        if torch.is_grad_enabled():
            feats = _FrontFn.apply(self, x, self._anchor)
        elif not self.training and SCORE_FP32:
            # scoring: fp32 activations / master weights / exact-fp32 GEMMs through the encoder and LL (the back-end is fp32 anyway)
            enc, T = self.ssl.forward_f32(x)
            feats = torch.empty(x.shape[0], T, FEAT_DIM, device=self.device)
            ops.gemm(Op(enc, self.cfg.embed), Op(self.P.flat, self.cfg.embed, offset=self.P.off("LL.weight")), feats, x.shape[0] * T, FEAT_DIM,
                     self.cfg.embed, bias=self.P.f32("LL.bias"))
        else:
            feats = self._front_forward(x)[0].clone()
        output, last_hidden = self._head(feats)
        if self.is_train:
            return output, feats, last_hidden
        return output

    def _head(self, feats):
        if not (self.use_graphs and self.training and torch.is_grad_enabled() and feats.requires_grad):
            return self._head_forward(self, feats)
        key = tuple(feats.shape)
        runner = self._graphed.get(key)
        if runner is None:
            runner = self._capture_head(feats)
            self._graphed[key] = runner
        if runner is False:          # capture failed once for this shape: stay eager
            return self._head_forward(self, feats)
        out, hid = runner(feats)          # static graph outputs: hand out copies, successive calls must not alias
        return out.clone(), hid.clone()

    def _capture_head(self, feats):
        """Warm-up + capture (torch runs the callable a few times on a side stream): BatchNorm running statistics and the RNG
        offset it consumes are put back afterwards, so capturing is invisible to the training trajectory."""
        runner = _HeadRunner(self)
        runner.train()
        saved = [b.detach().clone() for b in runner.buffers()]
        rng = torch.cuda.get_rng_state(self.device)
        try:
            sample = feats.detach().clone().requires_grad_(True)
            runner = torch.cuda.make_graphed_callables(runner, (sample,), allow_unused_input=True)   # e.g. AASIST's bn1 of Residual_block is unused
        except Exception as e:   # noqa: BLE001 - any capture problem degrades to the eager path, loudly
            print("[scl_amd] torch back-end: hipGraph capture failed (%s: %s); running it eagerly" % (type(e).__name__, e))
            runner = False
        finally:
            with torch.no_grad():
                for b, v in zip(_HeadRunner(self).buffers(), saved):
                    b.copy_(v)
            torch.cuda.set_rng_state(rng, self.device)
        return runner
