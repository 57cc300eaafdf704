"""Round 6, verdict item 2: does the host data path keep up with the GPU it feeds?  (-> profiles/r6_pack_builder.txt)

  1. pack builder alone: conf-3 packs (11 views x 64000; RawBoost12 + MUSAN + RIR) from a FLAC corpus through scl_amd.prefetch.Prefetcher
     with 1 / 2 / 4 / 8 builder threads, reference and fast RawBoost samplers -> packs / s, utterances / s;
  2. training end to end: main.run_epoch over that corpus at PACKS = 1 / 3 / 6 packs per optimizer step with the prefetcher, against
     the same steps on a batch that is already resident (the synthetic step rate);
  3. scoring end to end: main.produce_evaluation_file over 2048 FLAC files (64 distinct encodings, copied) against the forward alone on a
     resident batch (what `bench.py --eval` times).
The FLAC files come from tests/flac_writer.py (fixed order-2 predictor, Rice partitions, MD5) and are decoded by csrc/flac.hip.
"""
import os
import shutil
import sys
import tempfile
import time
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

import flac_writer  # noqa: E402
import main as M  # noqa: E402
from scl_amd.datautils_common import default_rawboost_args  # noqa: E402
from scl_amd.model_linear import Model  # noqa: E402
from scl_amd.optim import FusedAdamW  # noqa: E402
from scl_amd.prefetch import Prefetcher  # noqa: E402
import importlib  # noqa: E402

dev = torch.device("cuda:0")
QUICK = os.environ.get("QUICK") == "1"
PARTS = os.environ.get("PROBE_PARTS", "123")


def write_wav(path, x, sr=16000):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


def write_flac(path, x, sr=16000):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(flac_writer.write_flac((np.clip(x, -1, 1) * 32767).astype(np.int64), sample_rate=sr, bps=16))


def speechlike(rs, n):
    e = rs.randn(n)
    from scipy.signal import lfilter
    x = lfilter([1.0], [1.0, -1.6, 0.7], e)
    return 0.3 * x / np.abs(x).max()


tmp = tempfile.mkdtemp()
root = os.path.join(tmp, "data") + "/"
rs = np.random.RandomState(0)
NID = 8 if QUICK else 24
t0 = time.time()
ids = ["u%d.flac" % i for i in range(NID)]
for k, u in enumerate(ids):
    write_flac(os.path.join(root, "bonafide", u), speechlike(rs, 66000 + 1500 * (k % 8)))
    for v in ("hifigan", "hn-sinc-nsf-hifi", "waveglow"):
        if k < 8:
            write_flac(os.path.join(root, "vocoded", v + "_" + u), speechlike(rs, 66000))
        else:      # the Python encoder takes 0.4 s per file: later utterances reuse the first eight vocoded encodings (the decoder still runs on every read)
            shutil.copy(os.path.join(root, "vocoded", v + "_u%d.flac" % (k % 8)), os.path.join(root, "vocoded", v + "_" + u))
for i in range(4):
    write_wav(os.path.join(root, "musan", "n%d.wav" % i), 0.05 * rs.randn(160000))
    write_wav(os.path.join(root, "rirs", "r%d.wav" % i), np.exp(-np.arange(8000) / 1200.0) * rs.randn(8000) * 0.3)
print("corpus written in %.1f s" % (time.time() - t0), flush=True)

D = importlib.import_module("datautils.asvspoof_2019_augall_3")


def make_ds(sampler):
    args = default_rawboost_args()
    args.rawboost_sampler = sampler
    return D.Dataset_for(args, list_IDs=ids, labels=[], base_dir=root, algo=5, vocoders=["hifigan", "hn-sinc-nsf-hifi", "waveglow"],
                         augmentation_methods=["RawBoost12", "background_noise_wrapper", "reverb_wrapper"], num_additional_real=1, trim_length=64000,
                         wav_samp_rate=16000, online_aug=True, aug_dir=os.path.join(tmp, "aug"), noise_path=os.path.join(root, "musan"),
                         rir_path=os.path.join(root, "rirs"), repeat_pad=True)


class Rep(torch.utils.data.Dataset):
    def __init__(self, ds, n):
        self.ds, self.n = ds, n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self.ds[i % len(ids)]

    def decode_ahead(self, idxs):
        self.ds.decode_ahead([i % len(ids) for i in idxs])


print("== 1. pack builder alone (11-view conf-3 packs from FLAC; builder threads of scl_amd.prefetch, each on its own HIP stream)")
for sampler in (("reference", "fast") if "1" in PARTS else ()):
    ds = make_ds(sampler)
    for i in range(3):
        ds[i]
    torch.cuda.synchronize()
    for workers in ((1, 4) if QUICK else (1, 2, 4, 8)):
        n = 48 if QUICK else 160
        ld = DataLoader(Rep(ds, n), batch_size=1, shuffle=False, num_workers=0)
        torch.cuda.synchronize(); t0 = time.time(); k = 0
        for uid, bx, by in Prefetcher(ld, depth=2, device=dev, workers=workers):
            k += 1
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("sampler %-9s builder threads %d: %6.1f packs/s = %6.0f utterances/s  (%.2f ms per pack)" % (sampler, workers, k / dt, 11 * k / dt, dt / k * 1e3), flush=True)

print("== 2. training end to end through main.run_epoch (prefetcher, SCL_PREFETCH_THREADS default) vs the same step on a resident batch")
model = Model({"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1}, dev)
opt = FusedAdamW(model, lr=1e-5, weight_decay=1e-4)
conf = {"model": {"contra_mode": "all", "loss_type": 1}}
ds = make_ds("fast")
for packs in (() if "2" not in PARTS else ((3,) if QUICK else (1, 3, 6))):
    nsteps = 12 if QUICK else 30
    ld = DataLoader(Rep(ds, packs * nsteps), batch_size=packs, shuffle=False, num_workers=0, drop_last=True)
    M.run_epoch(Prefetcher(DataLoader(Rep(ds, packs * 4), batch_size=packs, num_workers=0), depth=2, device=dev), model, opt, dev, conf, train=True)      # warm-up: plans recorded
    res = {}
    for workers in (1, 2, 3):
        torch.cuda.synchronize(); t0 = time.time()
        M.run_epoch(Prefetcher(ld, depth=2, device=dev, workers=workers), model, opt, dev, conf, train=True)
        torch.cuda.synchronize()
        res[workers] = (time.time() - t0) / nsteps * 1e3
    # the same optimizer steps on resident input (what bench.py times at this batch size)
    uid, bx, by = next(iter(DataLoader(Rep(ds, packs), batch_size=packs, num_workers=0)))
    x, nutt, npk = M._as_model_input(bx, dev)
    y = by.view(-1).long().to(dev)
    V = nutt // npk
    model.train()

    def step():
        out, feats, emb = model(x)
        tot = None
        for pk in range(npk):
            sl = slice(pk * V, (pk + 1) * V)
            for v in (model.loss(out[sl], feats[sl], emb[sl], y[sl], conf) if npk > 1 else model.loss(out, feats, emb, y, conf)).values():
                tot = v if tot is None else tot + v
        opt.zero_grad(); tot.backward(); opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(nsteps):
        step()
    torch.cuda.synchronize()
    t_res = (time.time() - t0) / nsteps * 1e3
    print("PACKS=%d (%2d utterances per step): resident batch %.1f ms/step = %5.0f utt/s | end to end with 1 / 2 / 3 builder threads: %.1f ms (%.2f x) / %.1f ms (%.2f x) / %.1f ms (%.2f x); 1 thread (the default) = %5.0f utt/s"
          % (packs, nutt, t_res, nutt / t_res * 1e3, res[1], res[1] / t_res, res[2], res[2] / t_res, res[3], res[3] / t_res, nutt / res[1] * 1e3), flush=True)

if "3" not in PARTS:
    shutil.rmtree(tmp, ignore_errors=True)
    sys.exit(0)
print("== 3. scoring end to end: main.produce_evaluation_file over FLAC files vs the forward alone on a resident batch")
NEVAL = int(os.environ.get("PROBE_NEVAL", "256" if QUICK else "2048"))
evdir = os.path.join(root, "eval")
os.makedirs(evdir, exist_ok=True)
t0 = time.time()
for i in range(min(64, NEVAL)):
    write_flac(os.path.join(evdir, "e%d.flac" % i), speechlike(rs, 48000 + 700 * i)) if i < 16 else shutil.copy(os.path.join(evdir, "e%d.flac" % (i % 16)), os.path.join(evdir, "e%d.flac" % i))
for i in range(64, NEVAL):
    shutil.copy(os.path.join(evdir, "e%d.flac" % (i % 64)), os.path.join(evdir, "e%d.flac" % i))
E = importlib.import_module("datautils.asvspoof_2019_augall_3")
ev_ids = ["e%d.flac" % i for i in range(NEVAL)]
eval_set = E.Dataset_for_eval(list_IDs=ev_ids, base_dir=root)
B = 64
model.eval()
xb = torch.stack([eval_set[i][0] for i in range(B)]).to(dev)
with torch.no_grad():
    for _ in range(2):
        model(xb)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(NEVAL // B):
        model(xb)
    torch.cuda.synchronize()
t_fwd = time.time() - t0
for thr in os.environ.get("PROBE_EVAL_THREADS", "1,8").split(","):
    os.environ["SCL_EVAL_THREADS"] = thr
    out = os.path.join(tmp, "scores_%s.txt" % thr)
    torch.cuda.synchronize(); t0 = time.time()
    M.produce_evaluation_file(eval_set, model, dev, out, batch_size=B)
    torch.cuda.synchronize()
    dt = time.time() - t0
    nl = sum(1 for _ in open(out))
    print("%d files, batch %d: forward alone %.2f s = %5.0f utt/s | produce_evaluation_file, SCL_EVAL_THREADS=%s: %.2f s = %5.0f utt/s (%.2f x), %d score lines"
          % (NEVAL, B, t_fwd, NEVAL / t_fwd, thr, dt, NEVAL / dt, dt / t_fwd, nl), flush=True)
outs = [open(os.path.join(tmp, "scores_%s.txt" % t)).read() for t in os.environ.get("PROBE_EVAL_THREADS", "1,8").split(",")]
print("score files of every loop identical:", all(o == outs[0] for o in outs))
shutil.rmtree(tmp, ignore_errors=True)
