#!/usr/bin/env python
"""bench.py — train-step throughput of the MI355X hot path on BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    N > 1 either way: under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` (RANK / WORLD_SIZE in
    the environment: this process IS a rank), or as plain `python bench.py --gpus N ...` — then this process makes no GPU call at all,
    starts N child ranks of itself (the reference goes multi-GPU in-process: main.py:350-355), relays rank 0's JSON line and exits
    non-zero if any rank did.

A step = one optimizer step of `wav2vec2_linear_nll` at XLS-R-300M shape on a synthetic batch that is
already resident in HBM: [RawBoost on the GPU if the config says so ->] forward -> NLL + SupCon losses
-> backward -> gradient all-reduce (N > 1) -> fused AdamW.  Default workload = BASELINE.json
configs[2], the north_star's target configuration and the largest single-GPU config: wav2vec2_linear_nll + SupCon
(conf-3 loss), batch 64 x 64000-sample clips per GPU, RawBoost algo 5 (LnL + ISD) on the GPU INSIDE the timed step,
bf16 GEMM operands.  configs[1] (batch 32, RawBoost off) = `--batch 32 --rawboost 0`.
Weak scaling: every rank runs the same per-GPU batch; value = all ranks' utterances / max-over-ranks time.

The JSON line also carries
  roofline     — the dominant kernel family (scl_gemm_*, bf16 MFMA): algorithmic FLOPs per launch
                 / average launch duration, both measured live with HIP events on the launch stream
                 over the timed region, against the dense bf16 MFMA peak (2.5 PFLOP/s);
  roofline_aug — the RawBoost chain (FIR + normalise + ISD scatter kernels): SURVEY.md 8(d)'s 8 B/sample
                 = 512 kB per clip / the chain's kernel time (HIP events), against the 8 TB/s HBM peak, and its
                 fp32 FLOP/s against the 157.3 TFLOP/s vector peak (the direct-form FIR is VALU-bound);
  cpu_baseline — the oracle's CPU path (RawBoost per clip as the reference does it + torch fp32 train step,
                 all host cores) on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0
FLOP_PER_UTT_STEP_64000 = 444e9  # 74.0 GMAC forward x 2 x 3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU (configs[2]: 64; configs[1]: 32)")
    ap.add_argument("--samples", type=int, default=64000)
    ap.add_argument("--rawboost", type=int, default=5, help="RawBoost algo applied on the GPU inside the step (configs[2]: 5; 0 = off)")
    ap.add_argument("--model", choices=["wav2vec2_linear_nll", "wav2vec2_aasist", "wav2vec2_resnet_nll", "wav2vec2_btse"], default="wav2vec2_linear_nll",
                    help="model plugin; the headline (BASELINE.json) is wav2vec2_linear_nll, aasist / resnet / btse (configs[3], configs[4]) are extra workloads")
    ap.add_argument("--bio-tokens", type=int, default=199, help="wav2vec2_btse: synthetic bio tokens per utterance (the reference's tokeniser is absent)")
    ap.add_argument("--eval", action="store_true", help="scoring throughput instead of the train step: main.py --eval / 03_eval.sh's forward (model.eval(), "
                    "is_train False, no grad) on batches of 64600-sample clips (Dataset_for_eval's pad length); both scoring precisions are timed")
    ap.add_argument("--tiny", action="store_true", help="2-layer toy encoder (plumbing check only; not a valid bench)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-gemm-launches", default=None, metavar="FILE", help="write every profiled GEMM-family launch (ms, M, N, K, flags, z, variant) in "
                    "issue order as JSON: the input of tools/gemm_classes.py (profiles/r6_gemm_classes.txt)")
    ap.add_argument("--cpu-batch", type=int, default=4)
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: N child processes of this file, one per GPU, rendezvous on 127.0.0.1.
    The parent has not imported torch and never touches HIP (a process that initialised the GPU must not be replaced, and has no
    business holding a context on GPU 0 while rank 0 is timed).  stdout of rank 0 is relayed; its last line is the JSON line."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else None))
    # rank 0's stdout is drained on a thread, so that the poll loop below runs WHILE the ranks run: a rank (any rank) that dies before
    # the rendezvous leaves the others blocked in it — once one child has failed, the rest get a grace period and are killed
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * len(procs)
    deadline = None
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes) and deadline is None:
            deadline = time.time() + float(os.environ.get("SCL_BENCH_GRACE_S", "30"))
        if deadline is not None and time.time() > deadline:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.kill()                      # exactly the children started above
                    codes[i] = p.wait()
        time.sleep(0.05)
    reader.join(10.0)
    out = b"".join(chunks).decode("utf-8", "replace")
    lines = [l for l in out.splitlines() if l.strip()]
    for l in lines[:-1]:
        print(l)
    bad = [(i, c) for i, c in enumerate(codes) if c != 0]
    if bad:
        if lines:
            print(lines[-1])
        print("bench.py: ranks failed (rank, exit code): %r" % bad, file=sys.stderr)
        sys.exit(1)
    if not lines:
        print("bench.py: rank 0 printed nothing", file=sys.stderr)
        sys.exit(1)
    sys.stdout.flush()
    print(lines[-1], flush=True)
    sys.exit(0)


def gemm_source_sha():
    """Fingerprint of the GEMM kernel sources: a committed PMC pass is quoted only for the sources it was taken with."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "scl-deepfake-audio-detection_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "gemm*")) + glob.glob(os.path.join(csrc, "posconv*"))):
        if f.endswith((".hip", ".h")):
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(args):
    """Oracle CPU path (the restatement pinned to the reference) on a bounded sample: `cpu_batch` utterances of the same
    length — RawBoost per clip with the reference's own call sequence (one clip after the other, as a DataLoader worker
    does), then one DISCARDED train step (thread pool, allocator and oneDNN primitive caches warm) and one timed train step."""
    import numpy as np
    import torch
    from oracle import head as OH
    from oracle import rawboost as RB
    from oracle import wav2vec2 as W
    cfg = W.W2VConfig.tiny() if args.tiny else W.W2VConfig()
    ssl, head = W.init_state(cfg, seed=0), OH.init_head(cfg.embed, seed=1)
    B = args.cpu_batch
    x = 0.1 * torch.randn(B, args.samples, generator=torch.Generator().manual_seed(1234))
    y = torch.tensor(([1] * ((5 * B + 10) // 11) + [0] * B)[:B])
    t0 = time.time()
    t_aug = 0.0
    if args.rawboost:
        np.random.seed(1234)
        xa = np.stack([RB.process_rawboost_feature(x[i].numpy(), 16000, RB.RawBoostArgs(), args.rawboost) for i in range(B)])
        x = torch.from_numpy(xa.astype(np.float32))
        t_aug = time.time() - t0
    OH.train_step(ssl, head, cfg, x, y)          # discarded
    t1 = time.time()
    OH.train_step(ssl, head, cfg, x, y)
    t_step = time.time() - t1
    dt = t_aug + t_step
    return {"value": B / dt, "unit": "utterances/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d x %d-sample clips: oracle RawBoost algo %d per clip on 1 core (%.2f s) + the second of 2 train steps (fwd+loss+bwd+AdamW, "
                      "fp32 torch CPU oracle, %d threads; %.1f s), the first discarded" % (B, args.samples, args.rawboost, t_aug, torch.get_num_threads(), t_step)}


def bench_eval(args, model, dev, world, rank):
    """Scoring path (reference main.py:161-214: produce_evaluation_file over Dataset_for_eval's 64600-sample clips, batch_size from the CLI):
    forward only, model.eval(), is_train False, torch.no_grad().  Times the default scoring precision (fp32 activations + fp32 master weights:
    scores within 1e-3 of the fp32 reference) and the bf16-operand training kernels (SCL_SCORE_FP32=0: scores to ~1e-2), same weights, same
    input, and reports how far the two sets of log-probs are apart."""
    import torch
    from scl_amd import model_linear
    assert world == 1, "--eval is a single-GPU measurement (03_eval.sh shards the file list, every rank scores its own part)"
    B, L = args.batch, (64600 if args.samples == 64000 else args.samples)
    x = (0.1 * torch.randn(B, L, generator=torch.Generator().manual_seed(4321))).to(dev)
    model.eval()
    model.is_train = False
    extra = ()
    if args.model == "wav2vec2_btse":
        extra = (torch.randint(0, 3, (B, args.bio_tokens), dtype=torch.int32).to(dev), torch.full((B,), args.bio_tokens, dtype=torch.int32, device=dev))
    res, outs = {}, {}
    for name, flag in (("fp32", True), ("bf16", False)):
        model_linear.SCORE_FP32 = flag
        with torch.no_grad():
            for _ in range(max(1, args.warmup)):
                out = model(x, *extra)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out = model(x, *extra)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        res[name] = {"ms_per_batch": dt * 1e3, "utterances_per_s": B / dt}
        outs[name] = out.float().cpu()
    model_linear.SCORE_FP32 = True
    d = (outs["bf16"] - outs["fp32"]).abs()
    line = {"metric": "scoring utterances/sec (64600-sample clips)", "value": res["fp32"]["utterances_per_s"], "unit": "utterances/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["fp32"]["ms_per_batch"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s (XLS-R-300M shape, random init) scoring forward, batch %d x %d-sample clips, fp32 scoring path (the default of main.py --eval)" % (args.model, B, L),
                       "per_gpu_batch": B, "samples": L, "tiny": bool(args.tiny)},
            "fp32": res["fp32"], "bf16": res["bf16"],
            "bf16_vs_fp32": {"max_abs_logprob_diff": float(d.max()), "max_rel_to_largest": float(d.max() / outs["fp32"].abs().max()),
                             "argmax_agree": float((outs["bf16"].argmax(1) == outs["fp32"].argmax(1)).float().mean())},
            "eval_set_hours": {k: 71933 / v["utterances_per_s"] / 3600 for k, v in res.items()}}
    print(json.dumps(line), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                      # never returns
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank != 0:
        os.dup2(2, 1)       # only rank 0 owns stdout (library banners of the other ranks must not trail the JSON line)
    # SCL_BENCH_FORCE_DP=1: initialise the backend and run the gradient exchange even with one rank (a one-rank RCCL all-reduce is the
    # identity): the whole choreography — communication stream, events, async work handles — against the real library on a 1-GPU box
    dp = world > 1 or os.environ.get("SCL_BENCH_FORCE_DP") == "1"
    if dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # SCL_RCCL_CHANNELS=<n> caps RCCL's persistent blocks (NCCL_MAX_NCHANNELS) so that they and the one-per-CU GEMM blocks share the
        # 256 CUs (pair it with SCL_GEMM_CUS=256-n); opt-in: no multi-GPU A/B backs a default, and few channels can cap xGMI bandwidth
        if os.environ.get("SCL_RCCL_CHANNELS"):
            os.environ.setdefault("NCCL_MAX_NCHANNELS", os.environ["SCL_RCCL_CHANNELS"])
        # "nccl" is RCCL on ROCm.  SCL_BENCH_BACKEND=gloo + SCL_BENCH_ONE_DEVICE=1 rehearse the N > 1 code path on a one-GPU box
        dist.init_process_group(os.environ.get("SCL_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world)
    if os.environ.get("SCL_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from scl_amd import augment, ops
    from scl_amd.encoder import W2VConfig
    from scl_amd.lib import KID_AUG, KID_GEMM
    if args.model == "wav2vec2_aasist":
        from scl_amd.model_aasist import Model
    elif args.model == "wav2vec2_resnet_nll":
        from scl_amd.model_resnet import Model
    elif args.model == "wav2vec2_btse":
        from scl_amd.model_btse import Model
    else:
        from scl_amd.model_linear import Model
    from scl_amd.optim import FusedAdamW
    from scl_amd.parallel import GradSync

    cfg = W2VConfig.tiny() if args.tiny else W2VConfig()
    margs = {"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1}     # the aasist plugin falls back to the upstream AASIST-SSL sizes
    conf = {"model": {"contra_mode": "all", "loss_type": 1}}
    model = Model(margs, dev, w2v_cfg=cfg, seed=0)          # same seed on every rank = replicated weights
    model.train()                                            # dropout on, as train_epoch does (main.py:48)
    g_lo, g_hi = model.trainable_range()
    sync = GradSync(model.P.grad[g_lo:g_hi], base=g_lo, wire=os.environ.get("SCL_DP_WIRE", "fp32"), force=True,
                    mode=os.environ.get("SCL_DP_MODE", "allreduce")) if dp else None
    model.grad_sync = sync
    opt = FusedAdamW(model, lr=1e-5, weight_decay=1e-4, grad_sync=sync)   # main.py:339 defaults (max_lr, weight_decay)

    if args.eval:
        return bench_eval(args, model, dev, world, rank)
    B, L = args.batch, args.samples
    g = torch.Generator().manual_seed(1234 + rank)
    x = (0.1 * torch.randn(B, L, generator=g)).to(dev)       # resident in HBM before the timed region
    y = torch.tensor(([1] * ((5 * B + 10) // 11) + [0] * B)[:B], device=dev)
    fwd_extra = ()
    if args.model == "wav2vec2_btse":       # bio tokens are an input of the plugin (model/wav2vec2_btse/model.py:321); synthetic, resident like x
        fwd_extra = (torch.randint(0, 3, (B, args.bio_tokens), generator=g, dtype=torch.int32).to(dev), torch.full((B,), args.bio_tokens, dtype=torch.int32, device=dev))
    rb_args = None
    if args.rawboost:
        from scl_amd.datautils_common import default_rawboost_args
        rb_args = default_rawboost_args()

    def step():
        xs = augment.rawboost_batch(x, rb_args, args.rawboost, 16000, sampler="fast") if args.rawboost else x
        out, feats, emb = model(xs, *fwd_extra)
        losses = model.loss(out, feats, emb, y, conf)
        total = None
        for v in losses.values():
            total = v if total is None else total + v
        opt.zero_grad()
        if sync is not None:
            sync.begin()
        total.backward()
        opt.step()
        return total

    def fence():
        if dp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # roofline: every GEMM launch of the LAST n_prof timed steps (5, or all of a shorter run) carries two HIP events on its launch stream
    # (one step = ~320 launches; the kernel's own dispatch packet takes the time stamps).  The event pairs exist before the timed
    # region starts; `frac` is the mean over those steps, not one step on one box.
    n_prof = max(1, min(5, args.steps))
    ops.prof_reserve(KID_GEMM, 1024 * n_prof)
    ops.prof_reserve(KID_AUG, 64 * n_prof)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - n_prof:
            ops.prof_enable(KID_GEMM, True)
            ops.prof_enable(KID_AUG, True)
        last = step()
    fence()
    dt = time.perf_counter() - t0
    ops.prof_enable(KID_GEMM, False)
    ops.prof_enable(KID_AUG, False)
    if args.dump_gemm_launches and rank == 0:
        with open(args.dump_gemm_launches, "w") as f:
            json.dump({"batch": B, "samples": L, "model": args.model, "profiled_steps": n_prof, "gemm_src_sha": gemm_source_sha(),
                       "launches": [[ms] + list(meta) for ms, meta in ops.prof_read_launches(KID_GEMM)]}, f)
    n_launch, gemm_ms, gemm_flops = ops.prof_read(KID_GEMM)
    aug_launch, aug_ms, _ = ops.prof_read(KID_AUG)
    loss_val = float(last.item())
    # seeded random init, labels 5:6: CE/bz + 2 x SupCon/bz is a few tenths at batch 64 and grows as the batch shrinks (AASIST after 5 steps:
    # 0.58 / 0.92 / 2.4 at batch 64 / 32 / 16); the resnet plugin's loss has no 1/bz (x batch)
    hi = 2.0 * (B if args.model == "wav2vec2_resnet_nll" else max(1.0, 48.0 / B))
    assert loss_val == loss_val and 0.0 < loss_val < hi, "final loss %r outside the band of a seeded-random-init step" % loss_val
    if dp:
        t = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        torch.distributed.destroy_process_group()
        return
    utt_s = world * B * args.steps / dt
    flop_per_utt = FLOP_PER_UTT_STEP_64000 if (L == 64000 and not args.tiny) else None
    achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    # HBM traffic per GEMM launch cannot be read live (PMC needs rocprofv3): take it from the committed counter pass of this
    # same command (profiles/r5_pmc_hbm_traffic.json, or a previous round's; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) — but only
    # while the GEMM sources are the ones that pass was taken with (its "gemm_src_sha"), else null
    traffic = None
    for tag in ("r6", "r5", "r4", "r3"):
        try:
            with open(os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.json" % tag)) as f:
                pmc = json.load(f)
            if (B == pmc.get("batch", 32) and L == 64000 and not args.tiny and args.model == "wav2vec2_linear_nll"
                    and pmc.get("gemm_src_sha") == gemm_source_sha()):
                traffic = (2.0 * pmc["FETCH_SIZE"]["gemm"][0] + pmc["WRITE_SIZE"]["gemm"][0]) * 1024.0
                break
        except (OSError, KeyError, ValueError):
            continue
    res = {
        "metric": "train-step utterances/sec (64000-sample clips)", "value": utt_s, "unit": "utterances/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "%s (XLS-R-300M shape, random init) full train step: fwd + NLL/SupCon + bwd + "
                               "AdamW, batch %d x %d-sample clips per GPU, RawBoost %s" % (args.model, B, L, ("algo %d on-GPU" % args.rawboost) if args.rawboost else "off"),
                   "per_gpu_batch": B, "samples": L, "parallelism": "dp%d" % world, "tiny": bool(args.tiny)},
        "final_loss": loss_val,
        "model_flops_fraction_of_bf16_peak": (utt_s * flop_per_utt / (world * PEAK_BF16_TFLOPS * 1e12)) if flop_per_utt else None,
        "roofline": {"bound": "mfma", "kernel": "scl_gemm_{w8,w8s,dma}_kernel + posconv_mfma_kernel family (bf16 MFMA 16x16x32, all layouts)", "achieved": achieved,
                     "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                     "traffic_note": "bytes per launch from the committed rocprofv3 PMC pass, not live" if traffic else None,
                     "launches": n_launch, "profiled_steps": n_prof, "launches_per_step": n_launch / n_prof, "avg_launch_us": (gemm_ms * 1e3 / n_launch) if n_launch else None,
                     "avg_flops_per_launch": (gemm_flops / n_launch) if n_launch else None,
                     "gemm_share_of_step_time": gemm_ms * 1e-3 / n_prof / (dt / args.steps)},
    }
    if args.rawboost and aug_ms > 0:
        # 8 B/sample (fp32 in + out) per fused chain = 512 kB per 64000-sample clip; arithmetic = 2 * sum(taps) per
        # sample of the five LnL branches (+ a few FLOP/sample for the powers and the ISD / normalise passes)
        aug_ms /= n_prof                                   # per step (chain_ms = one chain over the batch)
        aug_launch //= n_prof
        aug_bytes = 8.0 * B * L
        aug_flops = 2.0 * augment.last_tap_total() * L
        gbs = aug_bytes / (aug_ms * 1e-3) / 1e9
        res["roofline_aug"] = {"bound": "hbm", "kernel": "RawBoost algo %d chain: fir_kernel (LnL, 5 power branches fused) + clip_affine + "
                               "isd_scatter + clip_stats + clip_affine" % args.rawboost, "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                               "frac": gbs / 8000.0, "traffic": None, "launches": aug_launch, "chain_ms": aug_ms,
                               "clips_per_s_kernel_only": B / (aug_ms * 1e-3), "profiled_steps": n_prof,
                               "achieved_fp32_tflops": aug_flops / (aug_ms * 1e-3) / 1e12, "fp32_vector_peak_tflops": 157.3,
                               "note": "direct-form FIR, ~2.7 kFLOP/sample: VALU-bound, HBM traffic is the minimal 8 B/sample"}
    if sync is not None:
        # self-diagnosing multi-GPU line: ranks, buckets, when each all-reduce was issued relative to the end of the backward and
        # how long the optimizer waited for the last one (the exposed part of the exchange)
        res["rccl"] = dict(sync.report() or {}, rccl_ranks=world, backend=torch.distributed.get_backend(), gemm_cus=os.environ.get("SCL_GEMM_CUS", "256"), nccl_max_nchannels=os.environ.get("NCCL_MAX_NCHANNELS"),
                           grad_bytes_per_step=(g_hi - g_lo) * (2 if sync.wire == "bf16" else 4))
    if world == 1 and not args.no_cpu_baseline and args.model == "wav2vec2_linear_nll":
        res["cpu_baseline"] = cpu_baseline(args)
    if dp:
        torch.distributed.destroy_process_group()
    # RCCL writes a version banner to the C stdout, which is flushed at exit — i.e. AFTER a Python print.  The JSON line must be the
    # last thing on stdout: tear the backend down, flush the C stream, then print.
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
