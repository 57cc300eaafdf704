"""Data-parallel train step on the GPU path with two processes (both on cuda:0, gloo transport — the box has one GPU;
the code path is the one RCCL takes on a multi-GPU node): bucketed all-reduce launched from the backward's callbacks
(also when the step is REPLAYED from a recorded launch plan), 1/world scaling inside AdamW, replicated weights."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ARGS = {"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1, "w2v_arch": "tiny"}
CONF = {"model": {"contra_mode": "all", "loss_type": 1}}


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return 0.1 * torch.randn(4, 4000, generator=g), torch.tensor([1, 1, 0, 0])


def _step(model, opt, x, y, sync=None):
    out, feats, emb = model(x)
    losses = model.loss(out, feats, emb, y, CONF)
    total = sum(losses.values())
    opt.zero_grad()
    if sync is not None:
        sync.begin()
    total.backward()
    opt.step()


def _worker(rank, world, port, q, mode="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scl_amd.model_linear import Model
    from scl_amd.optim import FusedAdamW
    from scl_amd.parallel import GradSync
    dev = torch.device("cuda:0")
    m = Model(ARGS, dev, seed=0)
    m.eval()                                   # dropout off so the single-process reference below is comparable
    sync = GradSync(m.P.grad[: m.P.n_train], bucket_elems=40000, mode=mode)   # several buckets even for the tiny model
    opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4, grad_sync=sync)
    x, y = _data(rank)
    for _ in range(3):                         # step 2 and 3 replay the recorded plan (with its bucket callbacks)
        _step(m, opt, x.to(dev), y.to(dev), sync)
    torch.cuda.synchronize()
    q.put((rank, m.P.flat[: m.P.n_train].cpu().numpy(), len(sync.bounds)))      # by value: the worker may exit before the parent reads
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "shard"])
def test_two_rank_step_equals_single_process_mean_gradient_step(dev, mode):
    """mode "shard": reduce-scatter of every bucket, AdamW on the rank's own shards, all-gather of the parameters (SURVEY.md 8e's
    optional ZeRO-1 form) — must land on the same weights as the all-reduce + replicated step and as the single-process reference."""
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (torch.from_numpy(w), nb)) for r, w, nb in (q.get(timeout=300) for _ in range(world)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][1] > 3                                    # really bucketed
    assert torch.equal(res[0][0], res[1][0])                # replicas stay bit-identical
    # single-process reference: same three steps on the MEAN of the two ranks' gradients
    from scl_amd.model_linear import Model
    from scl_amd.optim import FusedAdamW
    m = Model(ARGS, dev, seed=0)
    m.eval()
    opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4, overlap=False)   # gradients are averaged by hand between backward and step
    for _ in range(3):
        grads = []
        for r in range(world):
            x, y = _data(r)
            out, feats, emb = m(x.to(dev))
            sum(m.loss(out, feats, emb, y.to(dev), CONF).values()).backward()
            grads.append(m.P.grad.clone())
        m.P.grad.copy_((grads[0] + grads[1]) / world)
        opt.step()
    ref = m.P.flat[: m.P.n_train].cpu()
    err = (res[0][0] - ref).abs().max().item()
    assert err < 2e-6, err


TRAJ_LR = (4e-4, 1e-4)      # max / base rate of tests/golden/trajectory.npz (the reference's AdamW + CyclicLR, main.py:339-341)


def _traj_data(step, rank):
    g = torch.Generator().manual_seed(1000 + 10 * step + rank)
    return 0.1 * torch.randn(4, 4000, generator=g), torch.tensor([1, 1, 0, 0])


def _traj_sched(opt):
    return torch.optim.lr_scheduler.CyclicLR(opt, base_lr=TRAJ_LR[1], max_lr=TRAJ_LR[0], step_size_up=3, mode="exp_range", gamma=0.85,
                                             cycle_momentum=False)


def _worker_traj(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scl_amd.model_linear import Model
    from scl_amd.optim import FusedAdamW
    from scl_amd.parallel import GradSync
    dev = torch.device("cuda:0")
    m = Model(ARGS, dev, seed=0)
    m.eval()
    sync = GradSync(m.P.grad[: m.P.n_train], bucket_elems=40000)
    opt = FusedAdamW(m, lr=TRAJ_LR[0], weight_decay=1e-4, grad_sync=sync)
    sched = _traj_sched(opt)
    lrs, step = [], 0
    for ep in range(2):
        for i in range(3):
            x, y = _traj_data(step, rank)
            lrs.append(opt.param_groups[0]["lr"])
            _step(m, opt, x.to(dev), y.to(dev), sync)
            step += 1
        sched.step()                                           # main.py:416: once per epoch
    torch.cuda.synchronize()
    q.put((rank, m.P.flat[: m.P.n_train].cpu().numpy(), lrs))
    dist.destroy_process_group()


def test_two_rank_six_step_trajectory_with_the_epoch_scheduler_equals_the_mean_gradient_trajectory(dev):
    """The trajectory of tests/test_model_gpu.py (six steps, AdamW, CyclicLR stepped between step 3 and 4) under data parallelism: two
    ranks on different packs, bucketed all-reduce from the backward's callbacks, steps 2..6 replayed from the recorded launch plan while
    the scheduler changes the rate the fused AdamW kernel reads.  Equal to one process stepping on the mean of the two gradients."""
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_traj, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (torch.from_numpy(w), lrs)) for r, w, lrs in (q.get(timeout=300) for _ in range(world)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(res[0][0], res[1][0])
    from scl_amd.model_linear import Model
    from scl_amd.optim import FusedAdamW
    m = Model(ARGS, dev, seed=0)
    m.eval()
    opt = FusedAdamW(m, lr=TRAJ_LR[0], weight_decay=1e-4, overlap=False)
    sched = _traj_sched(opt)
    w0 = m.P.flat[: m.P.n_train].cpu().clone()
    lrs, step = [], 0
    for ep in range(2):
        for i in range(3):
            lrs.append(opt.param_groups[0]["lr"])
            grads = []
            for r in range(world):
                x, y = _traj_data(step, r)
                out, feats, emb = m(x.to(dev))
                opt.zero_grad()
                sum(m.loss(out, feats, emb, y.to(dev), CONF).values()).backward()
                grads.append(m.P.grad.clone())
            m.P.grad.copy_((grads[0] + grads[1]) / world)
            opt.step()
            step += 1
        sched.step()
    assert lrs == res[0][1] == res[1][1] and len(set(lrs)) == 2, (lrs, res[0][1])          # the rate changed once, identically everywhere
    ref = m.P.flat[: m.P.n_train].cpu()
    err = (res[0][0] - ref).abs().max().item()
    upd = (ref - w0).abs().max().item()
    assert upd > 5e-4 and err < 2e-5, (err, upd)      # six Adam steps move weights by up to ~6 lr = 2e-3; round-off sign flips are bounded by lr each


def _worker_aasist(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scl_amd.model_aasist import Model
    from scl_amd.optim import FusedAdamW
    from scl_amd.parallel import GradSync
    dev = torch.device("cuda:0")
    m = Model(ARGS, dev, seed=0)
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    lo, hi = m.trainable_range()
    sync = GradSync(m.P.grad[lo:hi], bucket_elems=40000, base=lo)
    # every bucket must be handed to the collective only after its slice is final: the back-end's gradients are ACCUMULATED by torch
    # autograd (not written by the hand-scheduled backward), so snapshot each bucket at hand-over and compare after the backward
    snaps = []
    sync.final_check = lambda a, b: snaps.append((a, b, m.P.grad[lo + a: lo + b].clone()))
    opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4, grad_sync=sync)
    x, y = _data(rank)
    final_ok = True
    for _ in range(2):
        snaps.clear()
        out, feats, emb = m(x.to(dev))
        total = sum(m.loss(out, feats, emb, y.to(dev), CONF).values())
        opt.zero_grad()
        sync.begin()
        total.backward()
        # local gradients as they stand after the whole backward, before finish() lets late buckets go
        torch.cuda.synchronize()
        launched = sync.launched
        opt.step()
        torch.cuda.synchronize()
        final_ok = final_ok and launched >= 1 and len(snaps) == len(sync.bounds)
    q.put((rank, m.P.flat[: m.P.n_train].cpu().numpy(), final_ok, m.state_dict()["first_bn1.running_mean"].cpu().numpy()))
    dist.destroy_process_group()


def test_two_rank_aasist_step_equals_mean_gradient_step(dev):
    """The BatchNorm-carrying plugin under data parallelism: per-rank batch statistics (as nn.DataParallel replicas would have), the
    torch-autograd-accumulated back-end gradients inside the buckets, replicas bit-identical after the step and equal to a
    single-process step on the mean of the two ranks' gradients."""
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_aasist, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (torch.from_numpy(w), ok, torch.from_numpy(bn))) for r, w, ok, bn in (q.get(timeout=600) for _ in range(world)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]
    assert torch.equal(res[0][0], res[1][0])                 # replicated weights stay bit-identical
    assert not torch.equal(res[0][2], res[1][2])             # BatchNorm running statistics are per rank (different shards)
    from scl_amd.model_aasist import Model
    from scl_amd.optim import FusedAdamW
    refs = []
    m = Model(ARGS, dev, seed=0)
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4, overlap=False)
    for _ in range(2):
        grads = []
        for r in range(world):
            x, y = _data(r)
            out, feats, emb = m(x.to(dev))
            opt.zero_grad()
            sum(m.loss(out, feats, emb, y.to(dev), CONF).values()).backward()
            grads.append(m.P.grad.clone())
        m.P.grad.copy_((grads[0] + grads[1]) / world)
        opt.step()
    ref = m.P.flat[: m.P.n_train].cpu()
    err = (res[0][0] - ref).abs().max().item()
    assert err < 5e-5, err          # Adam's update is +-lr for any gradient: 2 steps x 1e-3 bound the effect of round-off sign flips


def _rccl_worker(port, q, wire):
    """One rank on the RCCL ("nccl") backend: the all-reduce of a one-rank group is the identity, so the forced exchange must
    leave the training trajectory unchanged (fp32 wire) — while every stream hand-off, async work handle and bucket callback of
    GradSync runs against the real library."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from scl_amd.model_linear import Model
    from scl_amd.optim import FusedAdamW
    from scl_amd.parallel import GradSync
    dev = torch.device("cuda:0")
    res = []
    for use_sync in (False, True):
        m = Model(ARGS, dev, seed=0)
        m.eval()
        sync = GradSync(m.P.grad, bucket_elems=40000, wire=wire, force=True) if use_sync else None
        opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4, grad_sync=sync, overlap=False)
        x, y = _data(0)
        for _ in range(3):
            _step(m, opt, x.to(dev), y.to(dev), sync)
        torch.cuda.synchronize()
        res.append(m.P.flat[: m.P.n_train].cpu())
        if use_sync:
            rep = sync.report()
    q.put((res[0].numpy(), res[1].numpy(), rep, dist.get_backend()))
    dist.destroy_process_group()


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_gradsync_choreography_on_rccl_with_one_rank(dev, wire):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(port, q, wire))
    p.start()
    try:
        plain, synced, rep, backend = q.get(timeout=240)
        plain, synced = torch.from_numpy(plain), torch.from_numpy(synced)
    finally:
        p.join(timeout=30)
        if p.is_alive():
            p.terminate()
    assert backend == "nccl" and rep["buckets"] >= 3 and rep["wire"] == wire and len(rep["issue_ms_before_finish"]) == rep["buckets"]
    if wire == "fp32":
        assert torch.equal(plain, synced)
    else:       # gradients rounded to bf16 on the wire: three AdamW steps of lr 1e-3 later the weights agree to a few 1e-4
        assert (plain - synced).abs().max().item() < 3.1e-3 and not torch.equal(plain, synced)


def _worker_global_supcon(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SCL_GLOBAL_SUPCON="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scl_amd.model_linear import Model
    from scl_amd.parallel import GradSync
    dev = torch.device("cuda:0")
    m = Model(ARGS, dev, seed=0)
    m.eval()
    sync = GradSync(m.P.grad[: m.P.n_train], bucket_elems=40000)
    m.grad_sync = sync
    x, y = _data(rank)
    out, feats, emb = m(x.to(dev))
    losses = m.loss(out, feats, emb, y.to(dev), CONF)
    sync.begin()
    sum(losses.values()).backward()
    scale = sync.finish()
    torch.cuda.synchronize()
    q.put((rank, (m.P.grad[: m.P.n_train] * scale).cpu().numpy(), {k: v.item() for k, v in losses.items()}))
    dist.destroy_process_group()


def test_global_batch_supcon_over_two_ranks_equals_the_one_process_global_loss(dev):
    """SCL_GLOBAL_SUPCON=1 (optional) = the loss nn.DataParallel computes (main.py:62-66): the SupCon terms see the features
    of both ranks (all-gathered) and each rank's NLL term is its share of CE_global / bz_global.  The averaged gradient of the two ranks
    must equal the gradient of Model.loss computed in ONE process on the concatenated batch of 8 utterances."""
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_global_supcon, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (torch.from_numpy(g), l)) for r, g, l in (q.get(timeout=300) for _ in range(world)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(res[0][0], res[1][0])
    assert abs(res[0][1]["L_CF1"] - res[1][1]["L_CF1"]) < 1e-6 and res[0][1]["L_CE"] != res[1][1]["L_CE"]      # one global term, two local ones
    from scl_amd.model_linear import Model, _LossFn
    m = Model(ARGS, dev, seed=0)
    m.eval()
    (x0, y0), (x1, y1) = _data(0), _data(1)
    x, y = torch.cat([x0, x1]).to(dev), torch.cat([y0, y1]).to(dev)
    out, feats, emb = m(x)
    ce, cf1, cf2 = _LossFn.apply(out, feats, emb, y)          # ONE process on the concatenated batch: what nn.DataParallel's Model.loss sees
    assert abs(cf1.item() - res[0][1]["L_CF1"]) < 1e-3 * abs(cf1.item()) + 1e-6
    assert abs(ce.item() - (res[0][1]["L_CE"] + res[1][1]["L_CE"]) / 2) < 2e-2 * abs(ce.item())      # the ranks' shares average to the global term
    (ce + cf1 + cf2).backward()
    torch.cuda.synchronize()
    ref = m.P.grad[: m.P.n_train].cpu()
    got = res[0][0]
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 2e-2, rel          # the batch-8 forward and the two batch-4 forwards pick different GEMM tiles: bf16 round-off only
