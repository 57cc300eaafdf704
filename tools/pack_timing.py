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


tmp = tempfile.mkdtemp()
root = os.path.join(tmp, "data") + "/"
rs = np.random.RandomState(0)
ids = ["u%d.wav" % i for i in range(8)]
for u in ids:
    write_wav(os.path.join(root, "bonafide", u), 0.1 * rs.randn(70000 + 3000 * int(u[1])))
    for v in ("hifigan", "hn-sinc-nsf-hifi", "waveglow"):
        write_wav(os.path.join(root, "vocoded", v + "_" + u), 0.1 * rs.randn(66000))
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
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    ds[i % len(ids)]
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
