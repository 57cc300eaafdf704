"""Where does the RawBoost chain of the un-profiled default step lose its ~1 ms (profiles/r6_bench_default_phase_gaps.txt: 0.45 ms of kernels
spanning 1.44 ms)?  HIP events after every operation of the chain inside the full train step (the launch thread runs ~3 steps ahead, so
host time does not show): per operation, time from the previous event to this one = idle + kernel."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from scl_amd import augment, ops
from scl_amd.datautils_common import default_rawboost_args
from scl_amd.model_linear import Model
from scl_amd.optim import FusedAdamW

dev = torch.device("cuda:0")
model = Model({"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1}, dev, seed=0)
model.train()
opt = FusedAdamW(model, lr=1e-5, weight_decay=1e-4)
conf = {"model": {"contra_mode": "all", "loss_type": 1}}
B, L = 64, 64000
x = (0.1 * torch.randn(B, L, generator=torch.Generator().manual_seed(1234))).to(dev)
y = torch.tensor(([1] * ((5 * B + 10) // 11) + [0] * B)[:B], device=dev)
rb = default_rawboost_args()
marks = None


def mark(name):
    if marks is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, e))


def wrap(mod, name):
    f = getattr(mod, name)

    def g(*a, **k):
        r = f(*a, **k)
        mark(name)
        return r
    setattr(mod, name, g)


for nm in ("fir_multi", "clip_affine", "isd_scatter", "clip_stats"):
    wrap(ops, nm)
wrap(augment, "_h2d_pack")
_clone = torch.Tensor.clone


def step(rec):
    global marks
    marks = [] if rec else None
    mark("step start")
    xs = augment.rawboost_batch(x, rb, 5, 16000, sampler="fast")
    mark("augment done")
    m = marks
    marks = None
    out, feats, emb = model(xs)
    losses = model.loss(out, feats, emb, y, conf)
    total = None
    for v in losses.values():
        total = v if total is None else total + v
    opt.zero_grad()
    total.backward()
    opt.step()
    return m


for _ in range(5):
    step(False)
torch.cuda.synchronize()
runs = [step(True) for _ in range(10)]
torch.cuda.synchronize()
agg = collections.OrderedDict()
for m in runs:
    for i in range(1, len(m)):
        key = "%2d %s" % (i, m[i][0])
        agg.setdefault(key, []).append(m[i - 1][1].elapsed_time(m[i][1]) * 1e3)
print("# un-profiled default step, RawBoost algo 5 chain: us from the previous event to the one recorded after each operation; the FIRST step after the")
print("# synchronize apart (the GPU idles through the launch thread's parameter draw + filter design), then the mean of the 9 that follow (SCL_UPLOAD_STREAM=%s)" % os.environ.get("SCL_UPLOAD_STREAM", "1"))
tot0 = tot = 0.0
for k, v in agg.items():
    rest = v[1:]
    print("  %-28s first %9.1f us   steps 2..10 %8.1f us" % (k, v[0], sum(rest) / len(rest)))
    tot0 += v[0]; tot += sum(rest) / len(rest)
print("  %-28s first %9.1f us   steps 2..10 %8.1f us" % ("sum", tot0, tot))

# ---- is the launch thread ahead of the GPU when a step starts?  (host clock per phase; was the PREVIOUS step's last event already complete?)
import time
done_at_start, host = [], collections.OrderedDict()
prev_end = None
torch.cuda.synchronize()
for it in range(12):
    t0 = time.perf_counter()
    if prev_end is not None:
        done_at_start.append(prev_end.query())
    xs = augment.rawboost_batch(x, rb, 5, 16000, sampler="fast")
    t1 = time.perf_counter()
    out, feats, emb = model(xs)
    t2 = time.perf_counter()
    losses = model.loss(out, feats, emb, y, conf)
    total = None
    for v in losses.values():
        total = v if total is None else total + v
    t3 = time.perf_counter()
    opt.zero_grad()
    total.backward()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    prev_end = torch.cuda.Event(); prev_end.record()
    if it >= 2:
        for k, v in (("augment", t1 - t0), ("forward", t2 - t1), ("losses", t3 - t2), ("backward", t4 - t3), ("optimizer", t5 - t4)):
            host.setdefault(k, []).append(v * 1e3)
torch.cuda.synchronize()
print("# host clock per phase (ms, mean of 10 steps): " + ", ".join("%s %.2f" % (k, sum(v) / len(v)) for k, v in host.items()) + "; sum %.2f" % sum(sum(v) / len(v) for v in host.values()))
print("# previous step's last event already complete when the next step starts: %d of %d steps" % (sum(done_at_start), len(done_at_start)))
