"""Sanity: a fixed synthetic batch is over-fitted by a few dozen train steps at the PRODUCTION model size (loss must fall, no NaN),
for each model plugin.  Not a parity test — a smoke check that forward, hand-written backward and the fused optimizer move together."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd.optim import FusedAdamW

dev = torch.device("cuda:0")
conf = {"model": {"contra_mode": "all", "loss_type": 1}}
for name, nsteps, lr in (("linear", 40, 2e-5), ("aasist", 40, 2e-5), ("resnet", 40, 2e-5), ("linear", 240, 1e-4)):      # the last: long enough for the CE term to move
    if name == "linear":
        from scl_amd.model_linear import Model
    elif name == "aasist":
        from scl_amd.model_aasist import Model
    else:
        from scl_amd.model_resnet import Model
    m = Model({"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1}, dev, seed=0)
    m.train()
    opt = FusedAdamW(m, lr=lr, weight_decay=1e-4)
    g = torch.Generator().manual_seed(3)
    B = 8
    t = torch.arange(64000) / 16000.0
    # class 1: low tones, class 0: high tones (+ noise): separable
    x = torch.stack([0.3 * torch.sin(2 * torch.pi * (200 + 40 * i if i < 4 else 2000 + 300 * i) * t) + 0.02 * torch.randn(64000, generator=g)
                     for i in range(B)]).to(dev)
    y = torch.tensor([1, 1, 1, 1, 0, 0, 0, 0], device=dev)
    hist = []
    for step in range(nsteps):
        out, feats, emb = m(x)
        losses = m.loss(out, feats, emb, y, conf)
        tot = sum(losses.values())
        opt.zero_grad()
        tot.backward()
        opt.step()
        if step % max(5, nsteps // 8) == 0 or step == nsteps - 1:
            acc = (out.argmax(1) == y).float().mean().item()
            hist.append((step, round(tot.item(), 4), round(losses["L_CE"].item(), 4), acc))
    print(name, "%d steps lr %g" % (nsteps, lr), hist)
    assert all(torch.isfinite(torch.tensor(h[1])) for h in hist) and hist[-1][1] < hist[0][1], name
    del m, opt
    torch.cuda.empty_cache()
print("ok")
