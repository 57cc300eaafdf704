#!/usr/bin/env python
"""Training / scoring driver on the MI355X hot path — drop-in for the reference's main.py
(same flags main.py:217-298, same plugin resolution :325-333, same score-file formats :178/:212,
same checkpoint naming :310-318,44-45), so `02_train.sh` / `03_eval.sh` keep working.

What is different underneath:
  * model, losses, backward and AdamW run in the HIP kernels of libscl_hip.so; augmentation runs on the
    GPU inside the dataset, so the loaders use no worker processes;
  * loss terms are accumulated on the device and read back once per epoch instead of 3-4 `.item()`
    host syncs per step (main.py:71,73,76) — the epoch sums are the same numbers;
  * multi-GPU = one process per GPU (`python -m torch.distributed.run --nproc-per-node N main.py ...`):
    packs are sharded rank-strided, gradients all-reduced over RCCL/xGMI overlapped with backward
    (the reference's nn.DataParallel training path cannot run).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch
import yaml
from torch.utils.data import DataLoader, Subset

from model.wav2vec2_aasist import Model as wav2vec2_aasist
from model.wav2vec2_linear_nll import Model as wav2vec2_linear_nll
from model.wav2vec2_resnet_nll import Model as wav2vec2_resnet_nll
from model.wav2vec2_btse import wav2vec2_btse

MODEL_REGISTRY = {"wav2vec2_linear_nll": wav2vec2_linear_nll, "wav2vec2_aasist": wav2vec2_aasist, "wav2vec2_resnet_nll": wav2vec2_resnet_nll,
                  "wav2vec2_btse": wav2vec2_btse}


class EarlyStop:
    """Patience on validation accuracy; saves state_dict when it improves (reference main.py:23-45)."""

    def __init__(self, patience=5, delta=0, init_best=60, save_dir="", is_main=True):
        self.patience, self.delta, self.best_score, self.save_dir = patience, delta, init_best, save_dir
        self.counter, self.early_stop, self.is_main = 0, False, is_main

    def __call__(self, score, model, epoch):
        if self.best_score is None:
            self.best_score = score
        elif score < self.best_score + self.delta:
            self.counter += 1
            self.early_stop = self.counter >= self.patience
        else:
            print("Best epoch: {}".format(epoch))
            self.best_score, self.counter = score, 0
            if self.is_main:
                torch.save(model.state_dict(), os.path.join(self.save_dir, "epoch_{}.pth".format(epoch)))


class ScalarLog:
    """tensorboardX.SummaryWriter when installed, else one JSON line per scalar under logs/<tag>/scalars.jsonl."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        try:
            from tensorboardX import SummaryWriter
            self.tb, self.f = SummaryWriter(logdir), None
        except ImportError:
            self.tb, self.f = None, open(os.path.join(logdir, "scalars.jsonl"), "a")

    def add_scalar(self, name, value, step):
        if self.tb is not None:
            self.tb.add_scalar(name, value, step)
        else:
            self.f.write(json.dumps({"tag": name, "value": float(value), "step": int(step)}) + "\n")
            self.f.flush()


def _as_model_input(batch_x, device):
    """[1, L, V] pack -> [V, L]; [bz, L] batch stays (reference main.py:57-63).  Returns (x, n_utts, n_packs).
    Extension over the reference (whose reshape only works for --batch_size 1): [k, L, V] with k > 1 packs becomes [k*V, L] —
    the encoder sees one k*V batch (k = 3 packs of 11 views run at 935 instead of 507 utterances/s on one MI355X), the loss is
    still computed pack by pack and summed, i.e. one optimizer step on the summed gradient of k packs."""
    batch_x = batch_x.to(device)
    if batch_x.dim() == 3:
        k, L, V = batch_x.shape
        return batch_x.permute(0, 2, 1).reshape(k * V, L), k * V, k
    return batch_x, batch_x.shape[0], 1


def run_epoch(loader, model, optimizer, device, config, train):
    model.train(train)
    sums, total_sum, correct, n_total = {}, torch.zeros((), device=device), torch.zeros((), device=device), 0.0
    ctx = torch.enable_grad() if train else torch.no_grad()
    with ctx:
        for info, batch_x, batch_y in loader:
            x, n, n_packs = _as_model_input(batch_x, device)
            n_total += n
            y = batch_y.view(-1).type(torch.int64).to(device)
            out, feat, emb = model(x)
            step_loss = None
            V = n // n_packs
            for pk in range(n_packs):        # SupCon positives / negatives stay inside a pack, as with --batch_size 1
                sl = slice(pk * V, (pk + 1) * V)
                losses = model.loss(out[sl], feat[sl], emb[sl], y[sl], config, info) if n_packs > 1 else \
                    model.loss(out, feat, emb, y, config, info)
                for k, v in losses.items():
                    step_loss = v if step_loss is None else step_loss + v
                    sums[k] = sums.get(k, 0) + v.detach()
            total_sum += step_loss.detach()
            correct += (out.argmax(dim=1) == y).sum()
            if train:
                optimizer.zero_grad()
                if getattr(optimizer, "grad_sync", None) is not None:
                    optimizer.grad_sync.begin()
                step_loss.backward()
                optimizer.step()
    # one host sync per epoch.  Data-parallel validation: the shards differ in size (drop_last=False), so the sums — not the per-rank
    # ratios — are all-reduced: accuracy and losses are the same global numbers a single-GPU run logs
    if not train and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        keys = sorted(sums)
        t = torch.stack([total_sum, correct, torch.tensor(float(n_total), device=device)] + [sums[k].float() for k in keys])
        torch.distributed.all_reduce(t)
        total_sum, correct, n_total = t[0], t[1], float(t[2])
        sums = {k: t[3 + i] for i, k in enumerate(keys)}
    detail = {k: float(v) for k, v in sums.items()}
    return float(total_sum), float(correct) / max(n_total, 1.0) * 100.0, detail


def _score_loop(dataset, model, device, batch_size, fn, workers=None):
    """Scoring loop of produce_evaluation_file / _prediction_file / _emb_file (reference main.py:120-214: a DataLoader over Dataset_for_eval).
    Round 6: the reference decodes on DataLoader workers; here a thread pool decodes and pads the files (the FLAC decoder is C behind ctypes:
    no GIL), batches are assembled in pinned memory and uploaded asynchronously, and `fn` — which reads results back to the host — runs for
    batch i - 1 AFTER batch i has been launched, so the GPU never waits for a file or for the host's result formatting.
    Order of `fn` calls and of the lines they write = protocol order, as before.  SCL_EVAL_THREADS=1: the plain loop."""
    from concurrent.futures import ThreadPoolExecutor
    from collections import deque
    if workers is None:
        workers = max(1, int(os.environ.get("SCL_EVAL_THREADS", "8")))
    model.eval()
    n = len(dataset)
    if workers <= 1 or n == 0:
        loader = DataLoader(dataset, batch_size, shuffle=False, drop_last=False)
        with torch.no_grad():
            for batch_x, utt_id in loader:
                fn(model(batch_x.to(device)), list(utt_id))
        return
    use_gpu = torch.device(device).type == "cuda"
    ahead = max(4 * batch_size, 4 * workers)            # items decoded or being decoded beyond the batch in flight
    old_switch = sys.getswitchinterval()
    sys.setswitchinterval(min(old_switch, 5e-4))        # the launch thread shares the interpreter lock with the decoders (scl_amd/prefetch.py)
    with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="scl-eval") as pool, torch.no_grad():
        futs = deque()
        nxt = 0

        def top_up():
            nonlocal nxt
            while nxt < n and len(futs) < ahead:
                futs.append(pool.submit(dataset.__getitem__, nxt))
                nxt += 1
        direct = hasattr(dataset, "load_into") and hasattr(dataset, "cut")      # scl_amd.pack.EvalDataset: decode straight into the batch's pinned rows
        if not direct:
            top_up()
        pending = None                                   # (host copies of the previous batch's results, their event, its ids)

        def to_host(res):
            # results -> pinned host tensors by an ASYNCHRONOUS copy queued right behind the batch's own kernels.  (A `.cpu()` inside
            # `fn` is a blocking copy that is ordered behind everything queued so far — i.e. behind the NEXT batch's forward: the
            # host then sat out every forward and assembled the following batch with the GPU idle: 70 instead of 44 ms per batch.)
            def one(t):
                if not (torch.is_tensor(t) and t.is_cuda):
                    return t
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                h.copy_(t, non_blocking=True)
                return h
            out = tuple(one(t) for t in res) if isinstance(res, (tuple, list)) else one(res)
            ev = torch.cuda.Event() if use_gpu else None
            if ev is not None:
                ev.record()
            return out, ev
        if direct:
            batches = deque()

            def submit_batch():
                nonlocal nxt
                cnt = min(batch_size, n - nxt)
                host = torch.empty((cnt, dataset.cut), dtype=torch.float32, pin_memory=use_gpu)
                hv = host.numpy()
                fs = [pool.submit(dataset.load_into, nxt + j, hv[j]) for j in range(cnt)]
                nxt += cnt
                batches.append((host, fs))
            while nxt < n and len(batches) < 4:          # three batches being decoded beyond the one in flight
                submit_batch()
        while (batches if direct else futs):
            if direct:
                host, fs = batches.popleft()
                ids = [f.result() for f in fs]
                if nxt < n:
                    submit_batch()
            else:
                items = [futs.popleft().result() for _ in range(min(batch_size, len(futs)))]
                top_up()
                xs, ids = [it[0] for it in items], [it[1] for it in items]
                host = torch.empty((len(xs),) + tuple(xs[0].shape), dtype=xs[0].dtype, pin_memory=use_gpu)
                torch.stack(xs, out=host)
            res, ev = to_host(model(host.to(device, non_blocking=True)))
            if pending is not None:
                if pending[1] is not None:
                    pending[1].synchronize()             # batch i - 1's copy: done long ago, batch i is what the GPU works on now
                fn(pending[0], pending[2])
            pending = (res, ev, ids)
        if pending is not None:
            if pending[1] is not None:
                pending[1].synchronize()
            fn(pending[0], pending[2])
    sys.setswitchinterval(old_switch)


def produce_evaluation_file(dataset, model, device, save_path, batch_size=10):
    """'<utt> <logp0> <logp1>' per line (reference main.py:161-180)."""
    def emit(res, ids):
        scores = res[0].cpu().numpy().tolist()
        with open(save_path, "a+") as fh:
            for f, cm in zip(ids, scores):
                fh.write("{} {} {}\n".format(f, cm[0], cm[1]))
    _score_loop(dataset, model, device, batch_size, emit)
    print("Scores saved to {}".format(save_path))


def produce_prediction_file(dataset, model, device, save_path, batch_size=10):
    """'<utt> <logp1> <argmax>' per line, model.is_train = False (reference main.py:182-214)."""
    model.is_train = False

    def emit(out, ids):
        score = out[:, 1].cpu().numpy().ravel().tolist()
        pred = out.argmax(dim=1).tolist()
        with open(save_path, "a+") as fh:
            for f, cm, p in zip(ids, score, pred):
                fh.write("{} {} {}\n".format(f, cm, p))
    _score_loop(dataset, model, device, batch_size, emit)
    print("Scores saved to {}".format(save_path))


def produce_emb_file(dataset, model, device, save_path, batch_size=10, scores_name="scores.txt"):
    """<save_path>/<utt>.npy embeddings + scores.txt (reference main.py:120-159)."""
    model.is_train = True
    os.makedirs(save_path, exist_ok=True)

    def emit(res, ids):
        out, _, emb = res
        for f, e in zip(ids, emb):
            np.save(os.path.join(save_path, f.split("/")[-1].split(".")[0]), e.cpu().numpy())
        with open(os.path.join(save_path, scores_name), "a+") as fh:
            for f, cm in zip(ids, out.cpu().numpy().tolist()):
                fh.write("{} {} {}\n".format(f, cm[0], cm[1]))
    _score_loop(dataset, model, device, batch_size, emit)
    print("Scores saved to {}".format(save_path))


def merge_rank_outputs(path, world, n_items):
    """Rank r scored items r, r + world, ...: interleave the shards back into protocol order and drop them."""
    shards = []
    for r in range(world):
        with open("%s.rank%d" % (path, r)) as fh:
            shards.append(fh.read().splitlines())
        want = len(range(r, n_items, world))
        if len(shards[r]) != want:      # a shard of an aborted run that was appended to, or a rank that died: never merge it silently
            raise RuntimeError("%s.rank%d holds %d lines, expected %d" % (path, r, len(shards[r]), want))
    with open(path, "a+") as out:
        for i in range(n_items):
            out.write(shards[i % world][i // world] + "\n")
    for r in range(world):
        os.remove("%s.rank%d" % (path, r))
    print("Scores of %d ranks merged into %s" % (world, path))


def build_parser():
    p = argparse.ArgumentParser(description="SCL deepfake-audio detection on MI355X")
    p.add_argument("--database_path", type=str, default="/your/path/to/data/")
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--num_epochs", type=int, default=100)
    p.add_argument("--start_epoch", type=int, default=0)
    p.add_argument("--min_lr", type=float, default=0.00000001)
    p.add_argument("--max_lr", type=float, default=0.00001)
    p.add_argument("--weight_decay", type=float, default=0.0001)
    p.add_argument("--loss", type=str, default="weighted_CCE")
    p.add_argument("--config", type=str, default="configs/config.yaml")
    p.add_argument("--padding_type", type=str, default="zero", help="zero or repeat")
    p.add_argument("--is_train", type=bool, default=True)
    p.add_argument("--seed", type=int, default=1234)
    p.add_argument("--model_path", type=str, default=None)
    p.add_argument("--comment", type=str, default=None)
    p.add_argument("--eval_output", type=str, default=None)
    p.add_argument("--eval", action="store_true", default=False)
    p.add_argument("--predict", action="store_true", default=False)
    p.add_argument("--emb", action="store_true", default=False)
    # RawBoost hyper-parameters (names and defaults of the reference CLI)
    p.add_argument("--algo", type=int, default=5)
    for name, default in (("nBands", 5), ("minF", 20), ("maxF", 8000), ("minBW", 100), ("maxBW", 1000), ("minCoeff", 10),
                          ("maxCoeff", 100), ("minG", 0), ("maxG", 0), ("minBiasLinNonLin", 5), ("maxBiasLinNonLin", 20),
                          ("N_f", 5), ("P", 10), ("g_sd", 2), ("SNRmin", 10), ("SNRmax", 40)):
        p.add_argument("--" + name, type=int, default=default)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("SCL_RCCL_CHANNELS"):      # opt-in cap on RCCL's persistent blocks (no multi-GPU A/B backs a default)
            os.environ.setdefault("NCCL_MAX_NCHANNELS", os.environ["SCL_RCCL_CHANNELS"])
        torch.distributed.init_process_group("nccl")
    os.makedirs("out", exist_ok=True)
    torch.manual_seed(args.seed)
    model_tag = "model_{}_{}_{}_{}".format(args.loss, args.num_epochs, args.batch_size, args.min_lr)
    if args.comment:
        model_tag += "_{}".format(args.comment)
    model_save_path = os.path.join("out", model_tag)
    os.makedirs(model_save_path, exist_ok=True)
    if not torch.cuda.is_available():
        sys.exit("main.py needs an MI355X: the product path has no CPU fallback (use the oracle/ for CPU experiments)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    args.device = str(device)
    print("Number of GPUs available: ", torch.cuda.device_count(), "| processes:", world)

    with open(args.config) as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    data_mod = importlib.import_module("datautils." + config["data"]["name"])
    genList, Dataset_for, Dataset_for_eval = data_mod.genList, data_mod.Dataset_for, data_mod.Dataset_for_eval
    # --seed initialises the weights (02_train.sh: "random seed to initialize the weight") and, with the rank, the dropout stream
    if config["model"]["name"] == "wav2vec2_btse" and not config["model"].get("bio_tokenizer"):
        # run_epoch / the scoring loops call model(x) as the reference's train_epoch does (main.py:60): the plugin then needs its tokeniser,
        # and the reference's own (model/wav2vec2_btse/biosegment) is a dangling symlink.  Say so now, not at the first forward.
        sys.exit("main.py: model wav2vec2_btse needs the YAML key model.bio_tokenizer = '<module>:<callable>' (waveforms [bz, L] numpy, "
                 "sample_rate) -> equal-length token rows; the reference's biosegment package does not exist, so %s cannot run as shipped "
                 "(scl_amd/model_btse.py; bench.py --model wav2vec2_btse feeds synthetic tokens)" % args.config)
    model = MODEL_REGISTRY[config["model"]["name"]](config["model"], device, seed=args.seed, rank=rank)
    print("nb_params:", sum(p.numel() for p in model.parameters()))

    from scl_amd.optim import FusedAdamW
    from scl_amd.parallel import GradSync, shard_indices
    from scl_amd.prefetch import Prefetcher
    g_lo, g_hi = model.trainable_range()
    # SCL_DP_MODE=shard: reduce-scatter + AdamW on the rank's shards + all-gather of the parameters instead of all-reduce + replicated AdamW
    sync = GradSync(model.P.grad[g_lo:g_hi], base=g_lo, mode=os.environ.get("SCL_DP_MODE", "allreduce")) if world > 1 else None
    optimizer = FusedAdamW(model, lr=args.max_lr, weight_decay=args.weight_decay, grad_sync=sync)
    scheduler = torch.optim.lr_scheduler.CyclicLR(optimizer, base_lr=args.min_lr, max_lr=args.max_lr, step_size_up=3,
                                                  mode="exp_range", gamma=0.85, cycle_momentum=False)
    if args.model_path:
        sd = torch.load(args.model_path, map_location=device)
        sd = {k.replace("module.", "").replace("_orig_mod.", ""): v for k, v in sd.items()}
        model.load_state_dict(sd)
        print("Model loaded")
    else:
        print("Model initialized")
        if not getattr(model, "pretrained_loaded", False):
            msg = ("no --model_path and no pre-trained XLS-R checkpoint (pretrained/xlsr2_300m.pt, model/xlsr.py:14): the 300 M-parameter "
                   "encoder starts from seeded random noise")
            if args.eval:
                sys.exit("main.py --eval: " + msg + "; refusing to score with it")
            print("WARNING: " + msg + " — the reference fine-tunes the pre-trained encoder")

    proto = os.path.join(args.database_path, "protocol.txt")
    if args.eval:
        if not args.eval_output:
            sys.exit("main.py --eval needs --eval_output")
        _, file_eval = genList(dir_meta=proto, is_train=False, is_eval=True)
        print("no. of eval trials", len(file_eval))
        eval_set = Dataset_for_eval(list_IDs=file_eval, base_dir=os.path.join(args.database_path + "/"), padding_type=args.padding_type)
        final_output = args.eval_output
        if world > 1:
            eval_set = Subset(eval_set, list(range(rank, len(eval_set), world)))
            # per-rank score shards, merged below in protocol order (embeddings are one file per utterance; their scores.txt is a shard too)
            shard = "%s.rank%d" % (os.path.join(final_output, "scores.txt") if args.emb else final_output, rank)
            if os.path.exists(shard):       # left behind by an aborted run: the writers append
                os.remove(shard)
            if not args.emb:
                args.eval_output = shard
        if args.predict:
            produce_prediction_file(eval_set, model, device, args.eval_output, batch_size=args.batch_size)
        elif args.emb:
            produce_emb_file(eval_set, model, device, args.eval_output, batch_size=args.batch_size,
                             scores_name="scores.txt.rank%d" % rank if world > 1 else "scores.txt")
        else:
            produce_evaluation_file(eval_set, model, device, args.eval_output, batch_size=args.batch_size)
        if world > 1:
            torch.distributed.barrier()
            if rank == 0:
                merge_rank_outputs(os.path.join(final_output, "scores.txt") if args.emb else final_output, world, len(file_eval))
        return 0

    repeat = args.padding_type == "repeat"
    # RawBoost parameter draws of the pack builder: "fast" = every builder thread's own numpy Generator, batched closed-form filter design,
    # ISD positions without a 64000-element permutation per clip (same distributions; scl_amd/augment.py); SCL_PACK_SAMPLER=reference =
    # the reference's draw-for-draw order on the global np.random stream (what the pack goldens pin; 0.5 ms more host time per clip)
    args.rawboost_sampler = os.environ.get("SCL_PACK_SAMPLER", "fast")
    d_label_trn, file_train = genList(dir_meta=proto, is_train=True, is_eval=False, is_dev=False)
    print("no. of training trials", len(file_train))
    train_set = Dataset_for(args, list_IDs=file_train, labels=d_label_trn, base_dir=args.database_path + "/", algo=args.algo,
                            repeat_pad=repeat, **config["data"]["kwargs"])
    d_label_dev, file_dev = genList(dir_meta=proto, is_train=False, is_eval=False, is_dev=True)
    print("no. of validation trials", len(file_dev))
    args.is_train = False
    dev_set = Dataset_for(args, list_IDs=file_dev, labels=d_label_dev, base_dir=args.database_path + "/", algo=args.algo,
                          repeat_pad=repeat, **config["data"]["kwargs"])
    if world > 1:
        dev_set = Subset(dev_set, shard_indices(len(dev_set), rank, world, drop_last=False))
    dev_loader = DataLoader(dev_set, batch_size=args.batch_size, num_workers=0, shuffle=False)

    writer = ScalarLog("logs/{}".format(model_tag)) if rank == 0 else None
    early = EarlyStop(patience=10, delta=0.01, init_best=90.0, save_dir=model_save_path, is_main=rank == 0)
    t_start = time.time()
    for epoch in range(args.start_epoch, args.start_epoch + args.num_epochs):
        print("Epoch {}/{}. Current LR: {}".format(epoch, args.num_epochs - 1, optimizer.param_groups[0]["lr"]))
        if world > 1:   # same permutation on every rank, rank-strided shards, equal step counts
            epoch_set = Subset(train_set, shard_indices(len(train_set), rank, world, epoch_seed=args.seed + epoch))
            train_loader = DataLoader(epoch_set, batch_size=args.batch_size, num_workers=0, shuffle=False, drop_last=True)
        else:
            train_loader = DataLoader(train_set, batch_size=args.batch_size, num_workers=0, shuffle=True, drop_last=True)
        # pack i+1 is decoded / sampled / augmented by a prefetch thread on its own HIP stream while pack i trains (SCL_PREFETCH=0: inline)
        wrap = (lambda ld: Prefetcher(ld, depth=2, device=device)) if os.environ.get("SCL_PREFETCH", "1") != "0" else (lambda ld: ld)
        running_loss, train_acc, train_detail = run_epoch(wrap(train_loader), model, optimizer, device, config, train=True)
        val_loss, val_acc, val_detail = run_epoch(wrap(dev_loader), model, None, device, config, train=False)
        if writer is not None:
            for k, v in (("train_accuracy", train_acc), ("val_accuracy", val_acc), ("val_loss", val_loss), ("loss", running_loss)):
                writer.add_scalar(k, v, epoch)
            for k, v in train_detail.items():
                writer.add_scalar("train_{}".format(k), v, epoch)
            for k, v in val_detail.items():
                writer.add_scalar("val_{}".format(k), v, epoch)
        print("\n{} - {} - {} ".format(epoch, running_loss, val_loss))
        scheduler.step()
        early(val_acc, model, epoch)
        if early.early_stop:
            print("Early stopping activated.")
            break
    print("Total training time: {}s".format(time.time() - t_start))
    return 0


if __name__ == "__main__":
    sys.exit(main())
