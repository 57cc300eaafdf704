"""End-to-end run of main.py at PRODUCTION shapes (XLS-R-300M-size encoder, 64000-sample packs of 11 views, 64600-sample eval) on
synthetic wav files: train 1 epoch, score, predict.  Not a test (takes ~1-2 min on the GPU); a smoke check of the real sizes."""
import os, sys, tempfile, time, wave
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yaml
import main as M


def write_wav(path, x, sr=16000):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


tmp = tempfile.mkdtemp()
root = os.path.join(tmp, "data")
rs = np.random.RandomState(0)
ids = ["u%d.wav" % i for i in range(6)]
os.makedirs(os.path.join(root, "scp"))
for sub, names in (("scp/train_bonafide.lst", ids[:4]), ("scp/dev_bonafide.lst", ids[4:]), ("scp/test.lst", ids)):
    open(os.path.join(root, sub), "w").write("\n".join(names) + "\n")
open(os.path.join(root, "protocol.txt"), "w").write("")
for u in ids:
    write_wav(os.path.join(root, "bonafide", u), 0.1 * rs.randn(70000 + 3000 * int(u[1])))
    write_wav(os.path.join(root, "eval", u), 0.1 * rs.randn(50000 + 9000 * int(u[1])))
    for v in ("hifigan", "hn-sinc-nsf-hifi", "waveglow"):
        write_wav(os.path.join(root, "vocoded", v + "_" + u), 0.1 * rs.randn(66000))
for i in range(2):
    write_wav(os.path.join(root, "musan", "n%d.wav" % i), 0.05 * rs.randn(90000))
    write_wav(os.path.join(root, "rirs", "r%d.wav" % i), np.exp(-np.arange(4000) / 600.0) * rs.randn(4000) * 0.3)
name = sys.argv[1] if len(sys.argv) > 1 else "wav2vec2_linear_nll"
cfg = {"model": {"name": name, "flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1},
       "data": {"name": "asvspoof_2019_augall_3", "kwargs": {"vocoders": ["hifigan", "hn-sinc-nsf-hifi", "waveglow"], "augmentation_methods":
                ["RawBoost12", "background_noise_wrapper", "reverb_wrapper"], "num_additional_real": 1, "trim_length": 64000,
                "wav_samp_rate": 16000, "online_aug": True, "aug_dir": os.path.join(tmp, "aug"), "noise_path": os.path.join(root, "musan"),
                "rir_path": os.path.join(root, "rirs")}}}
cp = os.path.join(tmp, "conf.yaml")
open(cp, "w").write(yaml.safe_dump(cfg))
os.chdir(tmp)
t0 = time.time()
rc = M.main(["--seed", "1", "--config", cp, "--database_path", root, "--batch_size", "1", "--num_epochs", "2", "--padding_type", "repeat",
             "--comment", "full"])
print("train rc", rc, "%.1f s" % (time.time() - t0))
out = os.path.join(tmp, "scores.txt")
t0 = time.time()
M.main(["--config", cp, "--database_path", root, "--batch_size", "3", "--eval", "--eval_output", out])
print(open(out).read().strip().split("\n")[:3], "%.1f s" % (time.time() - t0))
