"""Where the launch thread spends a train step (no GPU sync inside): host time stamps around augment / forward / loss / backward / step of the
default bench workload.  A phase whose host time is of the order of the GPU step time contains a blocking call."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scl_amd import augment  # noqa: E402
from scl_amd.datautils_common import default_rawboost_args  # noqa: E402
from scl_amd.encoder import W2VConfig  # noqa: E402
from scl_amd.model_linear import Model  # noqa: E402
from scl_amd.optim import FusedAdamW  # noqa: E402

dev = torch.device("cuda:0")
B, L = int(os.environ.get("B", "64")), 64000
model = Model({"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1}, dev, w2v_cfg=W2VConfig(), seed=0)
model.train()
opt = FusedAdamW(model, lr=1e-5, weight_decay=1e-4)
conf = {"model": {"contra_mode": "all", "loss_type": 1}}
x = (0.1 * torch.randn(B, L, generator=torch.Generator().manual_seed(1234))).to(dev)
y = torch.tensor(([1] * ((5 * B + 10) // 11) + [0] * B)[:B], device=dev)
rb = default_rawboost_args()
names = ["augment", "forward", "loss", "zero_grad", "backward", "opt.step"]
acc = [0.0] * len(names)
N = 12
for it in range(5 + N):
    if it == 5:
        torch.cuda.synchronize()
        t_all = time.perf_counter()
        acc = [0.0] * len(names)
    t = [time.perf_counter()]
    xs = augment.rawboost_batch(x, rb, 5, 16000, sampler="fast"); t.append(time.perf_counter())
    out, feats, emb = model(xs); t.append(time.perf_counter())
    losses = model.loss(out, feats, emb, y, conf)
    total = sum(losses.values()); t.append(time.perf_counter())
    opt.zero_grad(); t.append(time.perf_counter())
    total.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    for i in range(len(names)):
        acc[i] += t[i + 1] - t[i]
host_all = time.perf_counter() - t_all
torch.cuda.synchronize()
wall = time.perf_counter() - t_all
print("per step: host %.2f ms (launch thread, un-synchronised), wall %.2f ms" % (host_all / N * 1e3, wall / N * 1e3))
for n, a in zip(names, acc):
    print("  %-10s %7.2f ms" % (n, a / N * 1e3))
