import os, sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "./tests")
os.environ["SCL_RESNET_CONV"] = sys.argv[1]
from oracle import head as OH
from oracle.aasist import fill_state
import test_resnet_gpu as T
dev = torch.device("cuda:0")
G = T.G
for case in ("eval", "train"):
    m = T._HeadWithLL().to(dev)
    sd = m.state_dict()
    filled = fill_state({k: tuple(v.shape) for k, v in sd.items()}, seed=7)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    m.eval() if case == "eval" else m.train()
    x = torch.from_numpy(G["x"]).to(dev).requires_grad_(True); y = torch.from_numpy(G["y"])
    out, feats, emb = m(x)
    losses = {k: v * 4 for k, v in OH.model_loss(out.cpu(), feats.cpu(), emb.cpu(), y, 1).items()}
    sum(losses.values()).backward()
    def err(a, b):
        a, b = np.asarray(torch.as_tensor(a).detach().cpu(), dtype=np.float64), np.asarray(b, dtype=np.float64)
        return np.abs(a - b).max() / max(np.abs(b).max(), 1e-6)
    print(case, "logits %.2e emb %.2e grad_x %.2e" % (err(out, G[case+":logits"]), err(emb, G[case+":emb"]), err(x.grad, G[case+":grad_x"])))
    params = dict(m.named_parameters())
    for k in G.files:
        if k.startswith(case + ":grad:"):
            print("   ", k, "%.2e" % err(params[k.split(":", 2)[2]].grad, G[k]))
        if k.startswith(case + ":buf:"):
            print("   ", k, "%.2e" % err(m.state_dict()[k.split(":", 2)[2]], G[k]))
