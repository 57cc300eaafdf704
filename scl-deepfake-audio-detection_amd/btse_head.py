"""The back-end of the `wav2vec2_btse` plugin (BASELINE.json configs[4]) behind the XLS-R encoder + LL projection, as HIP kernels.

Reference: model/wav2vec2_btse/backend.py:17-47 (LL -> MLP), model/wav2vec2_btse/linear.py:5-67 (MLP: 3 x (Linear 128 -> 128, LeakyReLU,
Dropout 0.5), mean over T; its own logits are discarded by model.py:324), model/wav2vec2_btse/model.py:210-238 (bioEncoderTransformersmall:
Embedding -> transformer.Encoder with window-4 relative attention -> 1x1 conv -> last position), model.py:321-343 (concat or fc1 + add,
fc2, log_softmax).  There is NO Conformer in this plugin (model/conformer.py has no caller in the reference).

`BtseHead` is the parameter container under the reference's state-dict names (backend.mlp.*, bioScoring.*, fc1 / fc2); its forward is ONE
autograd node whose forward and hand-written backward are C-ABI calls only:
  * the frame-level MLP on the exact-fp32 MFMA GEMM (bias + LeakyReLU + counter-hash dropout in the epilogue, as the linear plugin's head),
  * the whole bio transformer in one launch per direction, one workgroup per utterance (csrc/btse.hip),
  * the join + fc2 + log_softmax tail (csrc/btse.hip, csrc/elementwise.hip).
"""
import ctypes
import math

import torch
from torch import nn

from . import lib as L
from . import ops
from .lib import ACT_LEAKY
from .ops import Op

HID = 128          # backend.py:33-35: LL out = MLP in = MLP out
N_CLASS = 2
DROP_P = 0.5       # backend.py:34: MLP(128, 128, 3, 2, 0.5)
WINDOW = 4         # transformer.py:18
DEFAULT_BTSE = dict(bio_out=64, nb_classes=2, is_add=False, bio_hid=256, n_heads=4, pf_dim=128, n_layers=3, n_bios=3, bio_dim=32)   # conf-5-btse-trans64.yaml


class _Attn(nn.Module):
    """Parameter container of transformer.py:105-136 (same tensors, same initialisers; never called as a module)."""

    def __init__(self, D, H):
        super().__init__()
        dk = D // H
        self.conv_q, self.conv_k, self.conv_v, self.conv_o = (nn.Conv1d(D, D, 1) for _ in range(4))
        self.emb_rel_k = nn.Parameter(torch.randn(1, 2 * WINDOW + 1, dk) * dk ** -0.5)
        self.emb_rel_v = nn.Parameter(torch.randn(1, 2 * WINDOW + 1, dk) * dk ** -0.5)
        for m in (self.conv_q, self.conv_k, self.conv_v):
            nn.init.xavier_uniform_(m.weight)


class _ChanNorm(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.gamma, self.beta = nn.Parameter(torch.ones(D)), nn.Parameter(torch.zeros(D))


class _FFN(nn.Module):
    def __init__(self, D, Fd):
        super().__init__()
        self.conv_1, self.conv_2 = nn.Conv1d(D, Fd, 1), nn.Conv1d(Fd, D, 1)


class BtseHead(nn.Module):
    # parameters that never receive a gradient in the reference (model.py:324 drops the MLP's logits): torch.optim.AdamW skips them, so
    # they live behind the trainable range of the flat buffer
    frozen_names = ("backend.mlp.m_utt_level.weight", "backend.mlp.m_utt_level.bias")

    def __init__(self, args):
        super().__init__()
        a = dict(DEFAULT_BTSE)
        a.update({k: args[k] for k in DEFAULT_BTSE if k in args})
        self.btse_args = a
        D, Fd, NL, H = a["bio_dim"], a["pf_dim"], a["n_layers"], a["n_heads"]
        if a["is_add"] and a["bio_out"] != HID:
            raise ValueError("wav2vec2_btse: is_add needs bio_out == %d (fc2 takes %d inputs, model.py:296-299; the reference fails in "
                             "fc2 otherwise)" % (HID, HID))
        self.backend = nn.Module()
        mlp = nn.Module()
        mlp.m_frame_level = nn.Sequential()
        for i in range(3):                                                       # linear.py:27-36
            mlp.m_frame_level.add_module("linear_%d" % i, nn.Linear(HID, HID))
            mlp.m_frame_level.add_module("relu_%d" % i, nn.LeakyReLU())
            mlp.m_frame_level.add_module("dropout_%d" % i, nn.Dropout(DROP_P))
        mlp.m_utt_level = nn.Linear(HID, N_CLASS)
        self.backend.mlp = mlp
        bs = nn.Module()
        bs.bio_embedding = nn.Embedding(a["n_bios"], D)
        nn.init.normal_(bs.bio_embedding.weight, 0.0, D ** -0.5)                  # model.py:217
        enc = nn.Module()
        enc.attn_layers, enc.norm_layers_1, enc.ffn_layers, enc.norm_layers_2 = nn.ModuleList(), nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for _ in range(NL):                                                      # transformer.py:33-37
            enc.attn_layers.append(_Attn(D, H))
            enc.norm_layers_1.append(_ChanNorm(D))
            enc.ffn_layers.append(_FFN(D, Fd))
            enc.norm_layers_2.append(_ChanNorm(D))
        bs.encoder = enc
        bs.bio_scoring = nn.Conv1d(D, a["bio_out"], 1)
        self.bioScoring = bs
        if a["is_add"]:
            self.fc1 = nn.Linear(HID, a["bio_out"])
        self.fc2 = nn.Linear(HID if a["is_add"] else HID + a["bio_out"], a["nb_classes"])
        self.__dict__["_seed"] = 0x51ED27

    # ---- host-side plumbing (the receiver is the BtseHead itself or the plugin Model its sub-modules were grafted onto) ---------------
    @staticmethod
    def reseed(mod, seed):
        mod.__dict__["_seed"] = int(seed) & 0x7FFFFFFF

    @staticmethod
    def bio_params(mod):
        """The bio transformer's tensors in the order of the kernel descriptor (SclBtseBio.lw[l][0..17]) = the slab order of its backward."""
        bs, enc = mod.bioScoring, mod.bioScoring.encoder
        out = [bs.bio_embedding.weight]
        for i in range(len(enc.attn_layers)):
            at, f, n1, n2 = enc.attn_layers[i], enc.ffn_layers[i], enc.norm_layers_1[i], enc.norm_layers_2[i]
            out += [at.conv_q.weight, at.conv_q.bias, at.conv_k.weight, at.conv_k.bias, at.conv_v.weight, at.conv_v.bias, at.conv_o.weight,
                    at.conv_o.bias, at.emb_rel_k, at.emb_rel_v, n1.gamma, n1.beta, f.conv_1.weight, f.conv_1.bias, f.conv_2.weight, f.conv_2.bias,
                    n2.gamma, n2.beta]
        return out + [bs.bio_scoring.weight, bs.bio_scoring.bias]

    @staticmethod
    def check_tokens(mod):
        """Waits for the device-side range check of the last forward with device-resident tokens and raises IndexError if it failed."""
        _raise_if_bad_tokens(mod, mod.btse_args["n_bios"], wait=True)

    @staticmethod
    def forward(mod, feats, bio=None, bio_lengths=None):
        if bio is None:
            bio, bio_lengths = mod.__dict__["_bio"]
        a = mod.btse_args
        mlp = mod.backend.mlp.m_frame_level
        params = [mlp.linear_0.weight, mlp.linear_0.bias, mlp.linear_1.weight, mlp.linear_1.bias, mlp.linear_2.weight, mlp.linear_2.bias]
        params += BtseHead.bio_params(mod)
        if a["is_add"]:
            params += [mod.fc1.weight, mod.fc1.bias]
        params += [mod.fc2.weight, mod.fc2.bias]
        return _BtseFn.apply(mod, feats, bio, bio_lengths, *params)


def _plan(mod, B, T, Lt, dev):
    """Static buffers + the kernel descriptor of a (batch, frames, tokens) combination."""
    plans = mod.__dict__.setdefault("_btse_plans", {})
    key = (B, T, Lt, torch.cuda.current_stream(dev).cuda_stream)
    pl = plans.get(key)
    if pl is not None:
        return pl
    a = mod.btse_args
    if not ops.btse_bio_supported(a["bio_dim"], a["n_heads"], a["pf_dim"], a["n_layers"], WINDOW, a["bio_out"], Lt):
        raise L.SclError("wav2vec2_btse: the fused bio transformer serves bio_dim 32, 4 heads, pf_dim 128, 1-8 layers, bio_out <= 256 and "
                         "1 <= tokens <= 512 (got %r with %d tokens); there is no fallback path" % (a, Lt))
    M, bo, NL = B * T, a["bio_out"], a["n_layers"]
    KB = bo if a["is_add"] else HID + bo
    f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    bio_p = BtseHead.bio_params(mod)
    sizes = [p.numel() for p in bio_p]
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + n)
    NP = offs[-1]
    wsf = ops.btse_bio_ws_floats(NL, Lt)
    pl = dict(B=B, T=T, Lt=Lt, KB=KB, NP=NP, sizes=sizes, h=[f32(M * HID) for _ in range(3)], pre=[f32(M * HID) for _ in range(3)],
              dpre=[f32(M * HID) for _ in range(3)], emb=f32(B, HID), bvec=f32(B, KB), s=f32(B, bo), logp=f32(B, N_CLASS),
              ws=f32(B * wsf), bio=torch.empty(B, Lt, dtype=torch.int32, device=dev), lens=torch.empty(B, dtype=torch.int32, device=dev),
              slab=f32(B, NP), gbio=f32(NP), db=f32(B, KB), demb=f32(B, HID), ds=f32(B, bo), tail_ws=f32(B * N_CLASS + 16),
              cs=f32(ops.colsum_reduce_nparts(M, HID) * HID), wslab=f32(32 * HID * HID))
    d = L.SclBtseBio()
    d.emb = bio_p[0].data_ptr()
    k = 1
    for l in range(NL):
        for i in range(18):
            d.lw[l][i] = bio_p[k].data_ptr()
            d.go[l][i] = offs[k]
            k += 1
    d.Ws, d.bs = bio_p[k].data_ptr(), bio_p[k + 1].data_ptr()
    d.go_emb, d.go_Ws, d.go_bs = 0, offs[k], offs[k + 1]
    d.bio, d.lens, d.ws = pl["bio"].data_ptr(), pl["lens"].data_ptr(), pl["ws"].data_ptr()
    if a["is_add"]:
        d.out, d.out_ld = pl["s"].data_ptr(), bo
    else:                                       # the bio score lands in its columns of the concatenated row (model.py:333)
        d.out, d.out_ld = pl["bvec"].data_ptr() + 4 * HID, KB
    d.d_out, d.dout_ld, d.slab, d.slab_ld, d.ws_stride = pl["ds"].data_ptr(), bo, pl["slab"].data_ptr(), NP, wsf
    d.n_layers, d.n_bios, d.bio_out, d.L, d.B = NL, a["n_bios"], bo, Lt, B
    d.bio_dim, d.n_heads, d.pf_dim, d.window = a["bio_dim"], a["n_heads"], a["pf_dim"], WINDOW
    pl["desc"] = d
    pl["ptrs"] = [p.data_ptr() for p in bio_p]
    plans[key] = pl
    return pl


# Device-resident tokens (bench.py, real training) are range-checked WITHOUT a host round trip: the forward queues "any token outside
# [0, n_bios)?" into a pinned flag behind an event, and the next forward (or check_tokens()) raises nn.Embedding's IndexError once that
# event has completed — one step late instead of never (the kernel clamps for memory safety; model.py:228 would have raised).
def _queue_token_check(mod, bio_i32, n_bios):
    st = mod.__dict__.setdefault("_tokchk", {"host": torch.zeros(1, dtype=torch.int32).pin_memory(), "ev": torch.cuda.Event(), "pending": False, "bad": False})
    if st["pending"]:
        if not st["ev"].query():
            return                      # the previous check is still in flight: keep it (its pinned flag must not be overwritten)
        st["bad"] = st["bad"] or bool(st["host"][0].item())
    st["host"].copy_(((bio_i32 < 0) | (bio_i32 >= n_bios)).any().to(torch.int32).view(1), non_blocking=True)
    st["ev"].record()
    st["pending"] = True


def _raise_if_bad_tokens(mod, n_bios, wait=False):
    st = mod.__dict__.get("_tokchk")
    if st is None:
        return
    if st["pending"] and (wait or st["ev"].query()):
        st["ev"].synchronize()
        st["bad"] = st["bad"] or bool(st["host"][0].item())
        st["pending"] = False
    if st["bad"]:
        st["bad"] = False
        raise IndexError("wav2vec2_btse: a bio token of an earlier forward was out of range [0, %d) (the kernel clamped it; the reference's "
                         "nn.Embedding raises: model/wav2vec2_btse/model.py:228)" % n_bios)


class _BtseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, feats, bio, bio_lengths, *params):
        a = mod.btse_args
        dev = feats.device
        B, T, C = feats.shape
        assert C == HID
        if bio.dim() != 2 or bio.shape[0] != B or bio_lengths.shape[0] != B:
            raise ValueError("wav2vec2_btse: bio must be [batch, tokens] and bio_lengths [batch] (got %r, %r for batch %d)"
                             % (tuple(bio.shape), tuple(bio_lengths.shape), B))
        if not bio.is_cuda and bio.numel() and (int(bio.min()) < 0 or int(bio.max()) >= a["n_bios"]):
            raise IndexError("wav2vec2_btse: bio token out of range [0, %d)" % a["n_bios"])          # nn.Embedding's error (model.py:228)
        _raise_if_bad_tokens(mod, a["n_bios"])          # a device-side check queued by an EARLIER forward, if it has completed
        Lt = bio.shape[1]
        pl = _plan(mod, B, T, Lt, dev)
        if pl["ptrs"] != [p.data_ptr() for p in BtseHead.bio_params(mod)]:
            raise RuntimeError("wav2vec2_btse: parameter storage moved after the kernel descriptor was built")
        M = B * T
        training = bool(mod.training)
        drop = DROP_P if training else 0.0
        s0 = (mod.__dict__["_seed"] * 1664525 + 1013904223) & 0x7FFFFFFF
        mod.__dict__["_seed"] = s0
        seeds = [(s0 + 7919 * j) & 0x7FFFFFFF for j in range(3)]
        pl["bio"].copy_(bio.to(torch.int32), non_blocking=True)
        pl["lens"].copy_(bio_lengths.to(torch.int32), non_blocking=True)
        if bio.is_cuda and bio.numel():
            _queue_token_check(mod, pl["bio"], a["n_bios"])
        x = feats.contiguous()
        w = params[:6]
        prev = x
        for j in range(3):       # linear.py:27-36 on the exact-fp32 GEMM: bias + LeakyReLU(0.01) + dropout in the epilogue, pre-activation kept
            ops.gemm(Op(prev, HID), Op(w[2 * j].detach(), HID), pl["h"][j], M, HID, HID, bias=w[2 * j + 1].detach(), act=ACT_LEAKY, c2=pl["pre"][j],
                     drop_p=drop, drop_seed=seeds[j], x3=False)       # x3=False: the EXACT f32 kernel, as model_linear's head (not the SCL_F32X3 default)
            prev = pl["h"][j]
        ops.meanpool_fwd(prev, pl["emb"], B, T, HID)                              # linear.py:62
        ops.btse_bio_fwd(pl["desc"])                                              # model.py:328
        is_add = a["is_add"]
        fc1w, fc1b = (params[-4].detach(), params[-3].detach()) if is_add else (None, None)
        ops.btse_join_fwd(pl["emb"], pl["s"] if is_add else None, fc1w, fc1b, pl["bvec"], B, HID, a["bio_out"], is_add)
        ops.utt_head_fwd(pl["bvec"], params[-2].detach(), params[-1].detach(), pl["logp"], B, pl["KB"], a["nb_classes"])   # model.py:336-338
        ctx.mod, ctx.pl, ctx.x, ctx.seeds, ctx.drop, ctx.params = mod, pl, x, seeds, drop, params
        ctx.gen = pl["gen"] = pl.get("gen", 0) + 1
        return pl["logp"].clone(), pl["bvec"].clone()

    @staticmethod
    def backward(ctx, d_logp, d_b):
        mod, pl, x, seeds, drop, params = ctx.mod, ctx.pl, ctx.x, ctx.seeds, ctx.drop, ctx.params
        if pl["gen"] != ctx.gen:
            raise RuntimeError("wav2vec2_btse: a later forward of the same (batch, frames, tokens) shape overwrote this node's saved "
                               "activations; run each backward before the next forward of that shape")
        a = mod.btse_args
        dev = x.device
        B, T, bo, KB, NC = pl["B"], pl["T"], a["bio_out"], pl["KB"], a["nb_classes"]
        M = B * T
        is_add = a["is_add"]
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        d_logp = torch.zeros(B, NC, device=dev) if d_logp is None else d_logp.contiguous()
        dfc2w, dfc2b = f32(NC, KB), f32(NC)
        ops.utt_head_bwd(d_logp, pl["logp"], pl["bvec"], params[-2].detach(), None if d_b is None else d_b.contiguous(), pl["db"], dfc2w, dfc2b,
                         pl["tail_ws"], B, KB, NC)
        dfc1w, dfc1b = (f32(bo, HID), f32(bo)) if is_add else (None, None)
        ops.btse_join_bwd(pl["db"], pl["emb"], params[-4].detach() if is_add else None, pl["demb"], pl["ds"], dfc1w, dfc1b, B, HID, bo, is_add)
        ops.btse_bio_bwd(pl["desc"])
        ops.reduce_slabs(pl["slab"], pl["gbio"], pl["NP"], B, pl["NP"])          # utterance rows summed in index order
        gbio = [g.clone().view(p.shape) for g, p in zip(torch.split(pl["gbio"], pl["sizes"]), BtseHead.bio_params(mod))]
        # the frame-level MLP, in reverse (as scl_amd/model_linear.py::_backward_kernels, f32 operands)
        ops.meanpool_bwd(pl["demb"], pl["pre"][2], pl["dpre"][2], B, T, HID, ACT_LEAKY, drop, seeds[2])
        gw = [None] * 6
        d_feats = f32(B, T, HID)
        sk = max(1, min(32, M // 128))
        for j in (2, 1, 0):
            dpre = pl["dpre"][j]
            inp = pl["h"][j - 1] if j > 0 else x
            gw[2 * j + 1] = f32(HID)
            ops.colsum_reduce(dpre, pl["cs"], gw[2 * j + 1], M, HID)
            gw[2 * j] = f32(HID, HID)
            if sk > 1:
                ops.gemm(Op(dpre, HID), Op(inp, HID), pl["wslab"], HID, HID, M, a_t=True, b_t=True, splitk=sk, c_split_stride=HID * HID, x3=False)
                ops.reduce_slabs(pl["wslab"], gw[2 * j], HID * HID, sk, HID * HID)
            else:
                ops.gemm(Op(dpre, HID), Op(inp, HID), gw[2 * j], HID, HID, M, a_t=True, b_t=True, x3=False)
            wj = Op(params[2 * j].detach(), HID)
            if j > 0:
                ops.gemm(Op(dpre, HID), wj, pl["dpre"][j - 1], M, HID, HID, b_t=True, R=pl["pre"][j - 1], rmode=2, ract=ACT_LEAKY, drop_p=drop,
                         drop_seed=seeds[j - 1], x3=False)
            else:
                ops.gemm(Op(dpre, HID), wj, d_feats, M, HID, HID, b_t=True, x3=False)      # backend.py:41-43: no activation between LL and the MLP
        grads = gw + gbio + ([dfc1w, dfc1b] if is_add else []) + [dfc2w, dfc2b]
        return (None, d_feats, None, None) + tuple(grads)
