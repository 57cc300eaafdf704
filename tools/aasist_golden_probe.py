"""Per-tensor error of the HIP AASIST back-end against tests/golden/aasist.npz at every size tag (prints tensors above 1e-4)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_aasist_gpu as T  # noqa: E402
from test_aasist_cpu import analytically_zero  # noqa: E402

G = T.G
dev = torch.device("cuda:0")
for tag in ("", "199:", "202:"):
    for case in ("eval", "train"):
        m = T._HeadWithLL().to(dev)
        sd = m.state_dict()
        filled = T.fill_state({k: tuple(v.shape) for k, v in sd.items()}, seed=5)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
        if case == "eval":
            m.eval()
        else:
            m.train()
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
        x = torch.from_numpy(G[tag + "x"]).to(dev).requires_grad_(True)
        logits, hidden = m(x)
        (logits * torch.from_numpy(G[tag + "w_logits"]).to(dev)).sum().add((hidden * torch.from_numpy(G[tag + "w_hidden"]).to(dev)).sum()).backward()
        torch.cuda.synchronize()
        pre = tag + case
        params = dict(m.named_parameters())

        def err(a, b):
            a, b = np.asarray(torch.as_tensor(a).detach().cpu(), dtype=np.float64), np.asarray(b, dtype=np.float64)
            return np.abs(a - b).max() / max(np.abs(b).max(), 1e-6), np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30))
        for nm, a, b in (("logits", logits, G[pre + ":logits"]), ("hidden", hidden, G[pre + ":hidden"]), ("grad_x", x.grad, G[pre + ":grad_x"])):
            print(pre, nm, "max-rel %.2e rel-L2 %.2e" % err(a, b))
        for k in G.files:
            if k.startswith(pre + ":grad:"):
                name = k.split(":")[-1]
                if analytically_zero(name, case == "train"):
                    continue
                e = err(params[name].grad, G[k])
                if e[0] > 1e-4:
                    print(pre, name, "max-rel %.2e rel-L2 %.2e" % e)
            elif k.startswith(pre + ":gradfp:"):
                name = k.split(":")[-1]
                g = np.asarray(params[name].grad.detach().cpu(), dtype=np.float64)
                ref = G[k]
                e = abs(np.sqrt((g ** 2).sum()) - ref[0]) / ref[0], np.abs(g.flatten()[:16] - ref[2:]).max() / max(np.abs(g).max(), 1e-12)
                if max(e) > 1e-4:
                    print(pre, name, "fp: norm %.2e first16 %.2e" % e)
