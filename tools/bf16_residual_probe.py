"""Would a bf16 residual stream keep the forward inside north_star's 1e-2 bar?  CPU experiment on the oracle's encoder (test
infrastructure; the product is not involved) at XLS-R-300M shape, seeded random weights, 2 x 16000 samples:
  (a) the HIP path's numerics emulated: every GEMM operand (activation and weight) rounded to bf16, fp32 accumulation, fp32 residual
      stream, fp32 LayerNorm / soft-max — against the fp32 oracle;
  (b) the same with the residual stream stored in bf16 (rounded after the positional-conv add and after each of the 48 residual adds;
      LayerNorm statistics still fp32 on the bf16 values) — what the round-4 review proposed as an A/B (its item 3a)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import wav2vec2 as W

torch.manual_seed(0)
cfg = W.W2VConfig()
sd = W.init_state(cfg, seed=61)
x = 0.1 * torch.randn(2, 16000, generator=torch.Generator().manual_seed(1234))
bf = lambda t: t.to(torch.bfloat16).float()
real_linear = F.linear

def run(bf16_gemm, bf16_resid):
    def lin(inp, w, b=None):
        if bf16_gemm and w.dim() == 2 and w.shape[0] >= 1024 or (bf16_gemm and w.shape[-1] in (512, 1024, 4096)):
            return real_linear(bf(inp), bf(w), b)
        return real_linear(inp, w, b)
    F.linear = lin
    try:
        with torch.no_grad():
            feats = W.conv_stack(sd, cfg, x)
            h = F.layer_norm(feats[-1], (cfg.conv_dim,), sd["layer_norm.weight"], sd["layer_norm.bias"], 1e-5)
            h = F.linear(h, sd["post_extract_proj.weight"], sd["post_extract_proj.bias"])
            w = W.pos_conv_weight(sd)
            pc = F.conv1d(h.transpose(1, 2), w, sd["encoder.pos_conv.0.bias"], padding=cfg.pos_k // 2, groups=cfg.pos_groups)[:, :, :-1]
            h = h + F.gelu(pc).transpose(1, 2)
            outs = []
            for n in range(cfg.layers):
                if bf16_resid:
                    h = bf(h)
                p = "encoder.layers.%d." % n
                B, T, E = h.shape
                H, D = cfg.heads, E // cfg.heads
                res = h
                y = F.layer_norm(h, (E,), sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"], 1e-5)
                q = F.linear(y, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * (D ** -0.5)
                k = F.linear(y, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
                v = F.linear(y, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
                if bf16_gemm:
                    q, k, v = bf(q), bf(k), bf(v)
                q, k, v = (t.view(B, T, H, D).transpose(1, 2) for t in (q, k, v))
                a = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
                ctx = ((bf(a) if bf16_gemm else a) @ v).transpose(1, 2).reshape(B, T, E)
                h = res + F.linear(ctx, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
                if bf16_resid:
                    h = bf(h)
                res = h
                y = F.layer_norm(h, (E,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], 1e-5)
                y = F.gelu(F.linear(y, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
                h = res + F.linear(y, sd[p + "fc2.weight"], sd[p + "fc2.bias"])
                outs.append(h)
            if bf16_resid:
                h = bf(h)
            return F.layer_norm(h, (cfg.embed,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], 1e-5), outs
    finally:
        F.linear = real_linear

ref, ro = run(False, False)
rl2 = lambda a, b: float((a - b).norm() / b.norm())
for name, args in (("bf16 GEMM operands, fp32 residual stream (the product's numerics)", (True, False)),
                   ("bf16 GEMM operands, bf16 residual stream", (True, True)), ("fp32 GEMMs, bf16 residual stream only", (False, True))):
    out, oo = run(*args)
    print("%-72s encoder output rel-L2 %.2e | residual stream after layer 6 / 12 / 24: %.2e / %.2e / %.2e" % (name, rl2(out, ref), rl2(oo[5], ro[5]), rl2(oo[11], ro[11]), rl2(oo[23], ro[23])))
