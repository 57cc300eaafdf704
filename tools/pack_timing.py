"""Where does the time of one training pack go on the host?  Builds the conf-3 pack (11 views x 64000 samples) from synthetic wav
files repeatedly and prints the per-pack wall time and a cProfile of the hot functions."""
import cProfile, os, pstats, sys, tempfile, time, wave
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import importlib


def write_wav(path, x, sr=16000):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


def write_flac(path, x, sr=16000):
    """The corpus files are FLAC (ASVspoof 2019 LA): through tests/flac_writer.py, so that the pack builder's own decoder (csrc/flac.hip) is on the clock."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import flac_writer
    os.makedirs(os.path.dirname(path), exist_ok=True)
    pcm = (np.clip(x, -1, 1) * 32767).astype(np.int64)
    with open(path, "wb") as f:
        f.write(flac_writer.write_flac(pcm, sample_rate=sr, bps=16))


tmp = tempfile.mkdtemp()
root = os.path.join(tmp, "data") + "/"
rs = np.random.RandomState(0)
EXT = os.environ.get("PACK_EXT", "flac")
ids = ["u%d.%s" % (i, EXT) for i in range(8)]
wr = write_flac if EXT == "flac" else write_wav
for u in ids:
    wr(os.path.join(root, "bonafide", u), 0.1 * rs.randn(70000 + 3000 * int(u[1])))
    for v in ("hifigan", "hn-sinc-nsf-hifi", "waveglow"):
        wr(os.path.join(root, "vocoded", v + "_" + u), 0.1 * rs.randn(66000))
for i in range(4):
    write_wav(os.path.join(root, "musan", "n%d.wav" % i), 0.05 * rs.randn(160000))
    write_wav(os.path.join(root, "rirs", "r%d.wav" % i), np.exp(-np.arange(8000) / 1200.0) * rs.randn(8000) * 0.3)
D = importlib.import_module("datautils.asvspoof_2019_augall_3")
from scl_amd.datautils_common import default_rawboost_args
args = default_rawboost_args()
ds = D.Dataset_for(args, list_IDs=ids, labels=[], base_dir=root, algo=5, vocoders=["hifigan", "hn-sinc-nsf-hifi", "waveglow"],
                   augmentation_methods=["RawBoost12", "background_noise_wrapper", "reverb_wrapper"], num_additional_real=1, trim_length=64000,
                   wav_samp_rate=16000, online_aug=True, aug_dir=os.path.join(tmp, "aug"), noise_path=os.path.join(root, "musan"),
                   rir_path=os.path.join(root, "rirs"), repeat_pad=True)
for i in range(3):
    ds[i]
torch.cuda.synchronize()
t0 = time.time()
n = 24
for i in range(n):
    uid, x, y = ds[i % len(ids)]
torch.cuda.synchronize()
print("per pack: %.1f ms  (pack %s on %s)" % ((time.time() - t0) / n * 1e3, tuple(x.shape), x.device))
# ---- does the host keep up with the GPU?  one optimizer step per pack (what 02_train.sh runs), pack building inline vs prefetched
from torch.utils.data import DataLoader
from scl_amd.model_linear import Model
from scl_amd.optim import FusedAdamW
from scl_amd.prefetch import Prefetcher
dev = torch.device("cuda:0")
model = Model({"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1}, dev)
model.train()
opt = FusedAdamW(model, lr=1e-5, weight_decay=1e-4)
conf = {"model": {"contra_mode": "all", "loss_type": 1}}


class Rep(torch.utils.data.Dataset):
    def __len__(self):
        return 40

    def __getitem__(self, i):
        return ds[i % len(ids)]


def epoch(loader):
    torch.cuda.synchronize(); t0 = time.time(); n = 0
    for uid, bx, by in loader:
        x = bx.to(dev).squeeze(0).transpose(0, 1)
        y = by.view(-1).long().to(dev)
        out, feats, emb = model(x)
        loss = sum(model.loss(out, feats, emb, y, conf).values())
        opt.zero_grad(); loss.backward(); opt.step(); n += 1
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


ld = DataLoader(Rep(), batch_size=1, shuffle=False, num_workers=0)
epoch(ld)                                   # warm-up: launch plans recorded, banks resident
t_inline = epoch(ld)
t_pref = epoch(Prefetcher(ld, depth=2, device=dev))
xs = ds[0][1].t().contiguous()
ys = ds[0][2].long().to(dev)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(20):
    out, feats, emb = model(xs)
    loss = sum(model.loss(out, feats, emb, ys, conf).values())
    opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
t_gpu = (time.time() - t0) / 20 * 1e3
print("one optimizer step per 11-view pack: GPU step alone %.1f ms | pack building inline %.1f ms/iter | prefetch thread %.1f ms/iter" % (t_gpu, t_inline, t_pref))
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    ds[i % len(ids)]
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
