"""Data-parallel pieces on CPU with the gloo backend, world_size 2: bucketed gradient all-reduce launched
in backward order (GradSync), mean-gradient semantics, rank-strided pack sharding."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scl_amd.parallel import GradSync, shard_indices
    n = 1000
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    sync = GradSync(g, bucket_elems=300)
    assert sync.bounds[0] == (700, 1000) and sync.bounds[-1] == (0, 100)
    sync.begin()
    sync.ready_above(800)          # nothing complete yet (bucket 0 starts at 700)
    assert sync.launched == 0
    sync.ready_above(650)          # head + last layers done -> bucket [700, 1000) goes out
    assert sync.launched == 1
    sync.ready_above(100)
    assert sync.launched == 3
    scale = sync.finish()
    expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = torch.allclose(g, expect) and abs(scale - 1.0 / world) < 1e-12
    # second step reuses the object
    g.fill_(float(rank))
    sync.begin(); s2 = sync.finish()
    ok = ok and torch.allclose(g * s2, torch.full((n,), sum(range(world)) / world))
    sh = shard_indices(11, rank, world, epoch_seed=5)
    q.put((rank, ok, sh))
    dist.destroy_process_group()


def test_gradsync_and_sharding_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    shards = {r: s for r, _, s in res}
    assert len(shards[0]) == len(shards[1]) == 5 and not (set(shards[0]) & set(shards[1]))


def test_single_process_is_a_noop():
    from scl_amd.parallel import GradSync
    g = torch.ones(10)
    s = GradSync(g)
    s.begin(); s.ready_above(0)
    assert s.finish() == 1.0 and torch.equal(g, torch.ones(10))


def _worker_wire(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scl_amd.parallel import GradSync
    n = 1000
    torch.manual_seed(3 + rank)
    whole = torch.randn(n + 64)
    g = whole[64:]                               # a trainable slice that starts at whole-buffer offset 64 (frozen encoder in front)
    mine = g.clone()
    gathered = [torch.empty(n) for _ in range(world)]
    dist.all_gather(gathered, mine)
    exact = sum(gathered)
    seen = []
    sync = GradSync(g, bucket_elems=256, base=64, wire="bf16")
    sync.final_check = lambda lo, hi: seen.append((lo, hi, sync.launched))
    sync.begin()
    sync.ready_above(64 + 744)                   # whole-buffer offsets: bucket [744, 1000) of the slice is final
    first = list(seen)
    scale = sync.finish()
    rep = sync.report()
    # bf16 on the wire, fp32 accumulate back: each addend is rounded to bf16 once, the sum again
    ok = torch.allclose(g, exact, rtol=2e-2, atol=2e-2) and not torch.equal(g, exact) and abs(scale - 0.5) < 1e-12
    ok = ok and first == [(744, 1000, 0)] and [s[:2] for s in seen] == sync.bounds and rep["buckets"] == 4 and rep["wire"] == "bf16"
    ok = ok and torch.equal(whole[:64], whole[:64]) and len(rep["issue_ms_before_finish"]) == 4
    q.put((rank, ok))
    dist.destroy_process_group()


def test_bf16_wire_slice_base_and_bucket_order_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_wire, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _worker_shard(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scl_amd.parallel import GradSync
    n = 1003                                     # the lowest bucket (3 elements at world 2 x bucket 250: 1003 = 4 * 250 + 3) does not divide
    torch.manual_seed(11 + rank)
    g = torch.randn(n)
    gathered = [torch.empty(n) for _ in range(world)]
    dist.all_gather(gathered, g.clone())
    total = sum(gathered)
    p = torch.arange(n, dtype=torch.float32)     # replicated "parameters"
    sync = GradSync(g, bucket_elems=251, mode="shard")      # rounded down to a multiple of the world size
    ok = all((hi - lo) % world == 0 for lo, hi in sync.bounds[:-1]) and sync.bounds[0][1] == n
    sync.begin()
    sync.ready_above(600)
    launched_early = sync.launched
    scale = sync.finish()
    owned = sync.owned()
    # the rank holds the SUM on the ranges it owns (whole lowest bucket: replicated) ...
    for a, b in owned:
        ok = ok and torch.allclose(g[a:b], total[a:b], atol=1e-5)
    cover = torch.zeros(n)
    for a, b in owned:
        cover[a:b] += 1
    allc = [torch.empty(n) for _ in range(world)]
    dist.all_gather(allc, cover)
    ok = ok and launched_early >= 1 and abs(scale - 1.0 / world) < 1e-12
    lo_b, hi_b = sync.bounds[-1]
    expect_cover = torch.ones(n); expect_cover[lo_b:hi_b] = world if (hi_b - lo_b) % world else 1
    ok = ok and torch.equal(sum(allc), expect_cover)      # the shards tile the slice exactly once (the replicated remainder on every rank)
    # ... updates them (a plain SGD step stands in for the HIP AdamW) and the all-gather restores replicated parameters
    for a, b in owned:
        p[a:b] -= 0.1 * scale * g[a:b]
    sync.gather_params(p)
    ok = ok and torch.allclose(p, torch.arange(n, dtype=torch.float32) - 0.1 * total / world, atol=1e-5)
    q.put((rank, ok))
    dist.destroy_process_group()


def test_reduce_scatter_sharded_update_and_all_gather_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_shard, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _worker_gather(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scl_amd.parallel import gather_for_global_loss
    torch.manual_seed(5)
    full = torch.randn(world * 3, 4, 2)                       # the global batch every rank would see
    w = torch.randn(world * 3, 4, 2)
    local = full[rank * 3:(rank + 1) * 3].clone().requires_grad_(True)
    g = gather_for_global_loss(local)
    ok = torch.equal(g.detach(), full)
    # a term every rank evaluates identically on the gathered tensor; the exchange then AVERAGES the ranks' gradients
    loss = (g * g * w).sum()
    loss.backward()
    mine = torch.zeros_like(full)
    mine[rank * 3:(rank + 1) * 3] = local.grad
    dist.all_reduce(mine)
    mean_over_ranks = mine / world
    ok = ok and torch.allclose(mean_over_ranks, 2 * full * w, atol=1e-6)        # = the gradient of ONE global term
    lab = gather_for_global_loss(torch.tensor([rank, rank + 10]))
    ok = ok and lab.tolist() == [0, 10, 1, 11][: 2 * world]
    q.put((rank, ok))
    dist.destroy_process_group()


def test_global_loss_gather_backward_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_gather, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _worker_consolidate(rank, world, port, q):
    """FusedAdamW.state_dict() in the sharded mode: refuses while the moments outside the rank's shards are stale, whole after the
    collective consolidate_state() (the HIP update itself is not run here: the rank writes its own shards of the moments by hand)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from scl_amd.optim import FusedAdamW
    from scl_amd.parallel import GradSync
    n = 1003
    w = torch.nn.Parameter(torch.zeros(n))
    P = types.SimpleNamespace(params={"w": w}, n_train=n, device=torch.device("cpu"), flat=w.data, grad=torch.zeros(n))
    model = types.SimpleNamespace(P=P, grad_sync=None)
    sync = GradSync(P.grad, bucket_elems=251, mode="shard")
    opt = FusedAdamW(model, lr=1e-3, grad_sync=sync, overlap=False)
    ok = isinstance(opt.state_dict()["exp_avg"], torch.Tensor)          # nothing stepped yet: whole by construction
    sync.begin(); sync.finish()
    for a, b in sync.owned():                                           # what a sharded step leaves behind: only the owner's ranges are current
        opt.exp_avg[a:b] = torch.arange(a, b, dtype=torch.float32) + 1.0
        opt.exp_avg_sq[a:b] = 2.0 * torch.arange(a, b, dtype=torch.float32) + 1.0
    opt.step_count = 3
    try:
        opt.state_dict()
        ok = False
    except RuntimeError as e:
        ok = ok and "consolidate_state" in str(e)
    opt.consolidate_state()                                             # collective
    sd = opt.state_dict()
    lo_b, hi_b = sync.bounds[-1]                                        # the non-dividing remainder bucket is replicated (all-reduce fallback): both ranks wrote it
    ref = torch.arange(n, dtype=torch.float32)
    ok = ok and sd["step"] == 3 and torch.equal(sd["exp_avg"], ref + 1.0) and torch.equal(sd["exp_avg_sq"], 2.0 * ref + 1.0)
    opt.step_count = 4
    try:
        opt.state_dict()
        ok = False
    except RuntimeError:
        pass
    opt.load_state_dict(sd)
    ok = ok and opt.step_count == 3 and isinstance(opt.state_dict(), dict)
    q.put((rank, ok))
    dist.destroy_process_group()


def test_sharded_optimizer_state_needs_consolidation_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_consolidate, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)
