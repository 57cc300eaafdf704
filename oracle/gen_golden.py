"""Golden-vector generator (runs ONLY in the build container, where /root/reference exists).

Imports the reference's own Python — with empty stand-in modules for the third-party packages that
are not installed (fairseq, librosa, torchaudio, soundfile, pydub, tensorboardX); none of the code
paths exercised below calls into those stand-ins except `librosa.load`, which is replaced by an
in-memory array provider — runs it on seeded synthetic inputs and writes small .npz fixtures to
tests/golden/.  The fixtures are data (inputs + expected outputs); the reference never travels.

    python oracle/gen_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


class _Any(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return type(k, (), {})


def import_reference():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    # the reference's datautils/ has no __init__.py (it ships "__inti__.py"), so this repo's regular package of the same
    # name would win the import; pin the name to the reference's directory explicitly
    pkg = types.ModuleType("datautils")
    pkg.__path__ = [os.path.join(REF, "datautils")]
    sys.modules["datautils"] = pkg
    for name in [k for k in sys.modules if k.startswith("datautils.")]:
        del sys.modules[name]
    for name in ["fairseq", "librosa", "librosa.effects", "torchaudio", "torchaudio.functional", "torchaudio.io",
                 "torchaudio.transforms", "torchaudio.sox_effects", "soundfile", "pydub", "pydub.effects",
                 "tensorboardX"]:
        if name not in sys.modules:
            m = _Any(name)
            m.__path__ = []
            sys.modules[name] = m


class Args:
    algo = 5
    nBands, minF, maxF, minBW, maxBW = 5, 20, 8000, 100, 1000
    minCoeff, maxCoeff, minG, maxG = 10, 100, 0, 0
    minBiasLinNonLin, maxBiasLinNonLin, N_f = 5, 20, 5
    P, g_sd, SNRmin, SNRmax = 10, 2, 10, 40


def synth_clip(L, seed):
    return (0.1 * np.random.RandomState(seed).randn(L)).astype(np.float32)


def gen_rawboost():
    from datautils import RawBoost as R
    import datautils.asvspoof_2019_augall_3 as D
    out = {}
    L = 4000
    for seed in (0, 1):
        x = synth_clip(L, 100 + seed)
        out["x_s%d" % seed] = x
        for algo in range(0, 9):
            np.random.seed(1000 * algo + seed)
            y = D.process_Rawboost_feature(x, 16000, Args(), algo)
            out["algo%d_s%d" % (algo, seed)] = np.asarray(y)
    # one clip at a BASELINE config-1 length through LnL only, louder so that normWav triggers
    x = (0.9 * np.random.RandomState(7).randn(16000)).astype(np.float32)
    np.random.seed(77)
    a = Args()
    out["x_long"] = x
    out["lnl_long"] = R.LnL_convolutive_noise(x, a.N_f, a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff,
                                              a.maxCoeff, a.minG, a.maxG, a.minBiasLinNonLin, a.maxBiasLinNonLin, 16000)
    # notch design + centred FIR on their own
    for seed in (3, 4, 5):
        np.random.seed(seed)
        b = R.genNotchCoeffs(a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff, -5, -20, 16000)
        out["notch_b_s%d" % seed] = b
    out["fir_y"] = R.filterFIR(out["x_s0"], out["notch_b_s3"])
    # eval-side pad (augall_3:49-60)
    xs = synth_clip(700, 9)
    out["pad_in"] = xs
    out["pad_repeat"] = D.pad(xs, "repeat", 2000)
    out["pad_zero"] = D.pad(xs, "zero", 2000)
    out["pad_cut"] = D.pad(synth_clip(2500, 10), "zero", 2000)
    np.savez_compressed(os.path.join(OUT, "rawboost.npz"), **out)
    print("rawboost.npz", len(out), "arrays")


def gen_multiview():
    from core_scripts.data_io import wav_augmentation as A
    out = {}
    lens = {"longer": [5000, 4000, 6500, 5000], "shorter": [900, 1300, 500, 900], "exact": [2000, 2500, 100]}
    for name, ls in lens.items():
        views = [synth_clip(l, 20 + i)[:, None].astype(np.float64) for i, l in enumerate(ls)]
        for i, v in enumerate(views):
            out["%s_in%d" % (name, i)] = v
        for rp in (False, True):
            np.random.seed(5)
            res = A.batch_pad_for_multiview(views, 16000, 2000, random_trim_nosil=True, repeat_pad=rp)
            for i, v in enumerate(res):
                out["%s_rp%d_out%d" % (name, int(rp), i)] = v
    np.savez_compressed(os.path.join(OUT, "multiview.npz"), **out)
    print("multiview.npz", len(out), "arrays")


def gen_pack():
    """Dataset_for.__getitem__ (augall_3:103-146) on in-memory 'files', RawBoost12 online only."""
    import datautils.asvspoof_2019_augall_3 as D
    files = {}
    ids = ["a.flac", "b.flac", "c.flac", "d.flac"]
    vocoders = ["hifigan", "waveglow"]
    rs = np.random.RandomState(1)
    for i, u in enumerate(ids):
        files[os.path.join("/syn/", "bonafide", u)] = (0.1 * rs.randn(3000 + 400 * i)).astype(np.float32)
        for v in vocoders:
            files[os.path.join("/syn/", "vocoded", v + "_" + u)] = (0.1 * rs.randn(3000 + 400 * i - 37)).astype(np.float32)
    D.librosa.load = lambda path, sr=16000, mono=True: (files[path], sr)
    args = Args()
    ds = D.Dataset_for(args, list_IDs=ids, labels=[], base_dir="/syn/", algo=5, vocoders=vocoders,
                       augmentation_methods=["RawBoost12"], num_additional_real=1, trim_length=2000,
                       wav_samp_rate=16000, online_aug=True, aug_dir="/tmp/x", repeat_pad=True)
    out = {"ids": np.array(ids), "vocoders": np.array(vocoders)}
    for k, v in files.items():
        out["file:" + k] = v
    for idx in (0, 2):
        np.random.seed(40 + idx)
        uid, data, label = ds[idx]
        out["pack%d_data" % idx] = data.numpy()
        out["pack%d_label" % idx] = label.numpy()
        out["pack%d_id" % idx] = np.array(uid)
    np.savez_compressed(os.path.join(OUT, "pack.npz"), **out)
    print("pack.npz", len(out), "arrays")


def gen_pack_variants():
    """The sibling data plugins (aug_2, SCL_normal, augall_5): same procedure as gen_pack, RawBoost12 online only.
    os.listdir is made order-stable (sorted) because SCL_normal / augall_5 index the spoof directory by position."""
    import importlib
    import random
    import tempfile
    out = {}
    ids = ["a.flac", "b.flac", "c.flac", "d.flac", "e.flac"]
    vocoders = ["hifigan", "waveglow"]
    root = tempfile.mkdtemp()
    rs = np.random.RandomState(2)
    files = {}
    for sub in ("bonafide", "vocoded", "spoof_train", "spoof_dev", "spoof"):
        os.makedirs(os.path.join(root, sub))
    for i, u in enumerate(ids):
        files[os.path.join(root + "/", "bonafide", u)] = (0.1 * rs.randn(2600 + 300 * i)).astype(np.float32)
        for v in vocoders:
            files[os.path.join(root + "/", "vocoded", v + "_" + u)] = (0.1 * rs.randn(2500 + 300 * i)).astype(np.float32)
    for j in range(4):
        name = "s%d.wav" % j
        wav = (0.1 * rs.randn(2400 + 200 * j)).astype(np.float32)
        for sub in ("spoof_train", "spoof"):
            open(os.path.join(root, sub, name), "w").close()
            files[os.path.join(root + "/", sub, name)] = wav
    for k, v in files.items():
        out["file:" + os.path.relpath(k, root)] = v
    out["ids"], out["vocoders"] = np.array(ids), np.array(vocoders)
    real_listdir = os.listdir
    for modname, kw in (("asvspoof_2019_aug_2", dict(vocoders=vocoders, num_additional_real=2)),
                        ("SCL_normal", dict(vocoders=vocoders, num_additional_real=2, num_additional_spoof=2)),
                        ("asvspoof_2019_augall_5", dict(vocoders=vocoders, num_additional_real=1, num_additional_spoof=2))):
        D = importlib.import_module("datautils." + modname)
        D.librosa.load = lambda path, sr=16000, mono=True: (files[path], sr)
        D.os.listdir = lambda p: sorted(real_listdir(p))
        args = Args()
        args.is_train = True
        try:
            ds = D.Dataset_for(args, list_IDs=ids, labels=[], base_dir=root + "/", algo=5, augmentation_methods=["RawBoost12"],
                               trim_length=1500, wav_samp_rate=16000, online_aug=True, aug_dir="/tmp/x", repeat_pad=True, **kw)
        finally:
            D.os.listdir = real_listdir
        for idx in (1, 3):
            np.random.seed(60 + idx)
            random.seed(70 + idx)
            uid, data, label = ds[idx]
            out["%s:pack%d_data" % (modname, idx)] = data.numpy()
            out["%s:pack%d_label" % (modname, idx)] = label.numpy()
    np.savez_compressed(os.path.join(OUT, "pack_variants.npz"), **out)
    print("pack_variants.npz", len(out), "arrays")


def gen_head_loss():
    """Reference Model (linear head) with an injected encoder: BackEnd / _forward / loss / supcon."""
    import model.wav2vec2_linear_nll as M
    from model.loss_metrics import supcon_loss
    from oracle import head as OH
    E = 64

    class Enc(torch.nn.Module):
        out_dim = E

        def extract_feat(self, x, is_train=True):
            return x  # the test feeds encoder features directly

    M.SSLModel = lambda device: Enc()
    torch.manual_seed(3)
    m = M.Model({"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1}, "cpu")
    m.eval()
    out = {}
    hsd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for k, v in hsd.items():
        out["sd:" + k] = v.numpy()
    B, T = 6, 49
    x = torch.randn(B, T, E)
    y = torch.tensor([1, 1, 1, 0, 0, 0])
    x.requires_grad_(True)
    o, feats, emb = m(x)
    losses = m.loss(o, feats, emb, y, {"model": {"contra_mode": "all", "loss_type": 1}})
    total = sum(losses.values())
    total.backward()
    out.update(x=x.detach().numpy(), y=y.numpy(), out=o.detach().numpy(), feats=feats.detach().numpy(),
               emb=emb.detach().numpy(), grad_x=x.grad.numpy(), grad_LL_weight=m.LL.weight.grad.numpy(),
               grad_utt_weight=m.backend.m_utt_level.weight.grad.numpy(),
               grad_frame0_weight=m.backend.m_frame_level[0].weight.grad.numpy())
    for k, v in losses.items():
        out["loss:" + k] = np.array(v.item())
    # supcon on its own: seeds -> expected loss and a gradient fingerprint (inputs regenerated from the seed)
    rows = []
    for bz, T2, d, seed in ((4, 49, 128, 0), (11, 199, 128, 1), (64, 49, 128, 2), (11, 128, 1, 3), (32, 199, 128, 4)):
        torch.manual_seed(seed)
        f = torch.randn(bz, 1, T2, d, requires_grad=True)
        lab = torch.tensor(([1] * ((5 * bz + 10) // 11) + [0] * bz)[:bz])
        l = supcon_loss(f, labels=lab)
        l.backward()
        rows.append([bz, T2, d, seed, l.item(), f.grad.norm().item(), f.grad[0, 0, 0, 0].item(), f.grad[-1, 0, -1, -1].item()])
    out["supcon_cases"] = np.array(rows, dtype=np.float64)
    # the single-member-class NaN (loss_metrics.py:202)
    torch.manual_seed(9)
    f = torch.randn(4, 1, 10, 8)
    out["supcon_nan"] = np.array(supcon_loss(f, labels=torch.tensor([1, 0, 0, 0])).item())
    np.savez_compressed(os.path.join(OUT, "head_loss.npz"), **out)
    print("head_loss.npz", len(out), "arrays")


def gen_train_step():
    """One train_epoch iteration (main.py:53-80) of the reference Model whose encoder is the
    oracle's wav2vec2 restatement (tiny config), AdamW as main.py:339."""
    import model.wav2vec2_linear_nll as M
    from oracle import wav2vec2 as W
    cfg = W.W2VConfig.tiny()

    class Enc(torch.nn.Module):
        out_dim = cfg.embed

        def __init__(self):
            super().__init__()
            sd = W.init_state(cfg, seed=11)
            self.names = list(sd)
            self.model = torch.nn.ParameterDict({k.replace(".", "/"): torch.nn.Parameter(v) for k, v in sd.items()})

        def sd(self):
            return {k: self.model[k.replace(".", "/")] for k in self.names}

        def extract_feat(self, x, is_train=True):
            return W.forward(self.sd(), cfg, x)

    M.SSLModel = lambda device: Enc()
    torch.manual_seed(5)
    m = M.Model({"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1}, "cpu")
    m.eval()  # dropout off (its RNG cannot be shared); everything else as train_epoch
    out = {}
    for k, v in m.state_dict().items():
        if not k.startswith("ssl_model"):
            out["sd:" + k] = v.detach().clone().numpy()
    g = torch.Generator().manual_seed(1234)
    x = 0.1 * torch.randn(4, 4000, generator=g)
    y = torch.tensor([1, 1, 0, 0])
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)  # larger lr so the update is visible in fp32
    o, feats, emb = m(x)
    losses = m.loss(o, feats, emb, y, {"model": {"contra_mode": "all", "loss_type": 1}})
    train_loss = 0.0
    for k, v in losses.items():
        train_loss = train_loss + v
    opt.zero_grad()
    train_loss.backward()
    enc = m.ssl_model
    probe = ["feature_extractor.conv_layers.0.0.weight", "feature_extractor.conv_layers.3.0.weight",
             "feature_extractor.conv_layers.6.2.1.weight", "post_extract_proj.weight",
             "encoder.pos_conv.0.weight_g", "encoder.pos_conv.0.weight_v",
             "encoder.layers.0.self_attn.q_proj.weight", "encoder.layers.0.self_attn.k_proj.bias",
             "encoder.layers.1.fc2.weight", "encoder.layers.1.final_layer_norm.bias", "encoder.layer_norm.weight"]
    for p in probe:
        out["grad:ssl_model.model." + p] = enc.model[p.replace(".", "/")].grad.numpy().copy()
    out["grad:LL.weight"] = m.LL.weight.grad.numpy().copy()
    out["grad:backend.m_utt_level.bias"] = m.backend.m_utt_level.bias.grad.numpy().copy()
    opt.step()
    for p in probe:
        out["post:ssl_model.model." + p] = enc.model[p.replace(".", "/")].detach().numpy().copy()
    out["post:LL.weight"] = m.LL.weight.detach().numpy().copy()
    out.update(x=x.numpy(), y=y.numpy(), out=o.detach().numpy(), emb=emb.detach().numpy(),
               feats=feats.detach().numpy(), total_loss=np.array(train_loss.item()))
    for k, v in losses.items():
        out["loss:" + k] = np.array(v.item())
    np.savez_compressed(os.path.join(OUT, "train_step.npz"), **out)
    print("train_step.npz", len(out), "arrays")


def gen_trajectory():
    """SIX optimizer steps through the reference's OWN train_epoch (main.py:47-84, imported from the file) with AdamW as main.py:339 and the
    per-epoch CyclicLR of main.py:341 stepped as main.py:416 — two epochs of three packs, so steps 1-3 run at the scheduler's first rate and
    steps 4-6 at its second.  Model = the reference linear plugin with the oracle's tiny wav2vec2 restatement injected (as gen_train_step),
    train mode with every Dropout p = 0 (its RNG cannot be shared).  Input packs are the 3-D [1, L, V] tensors of --batch_size 1
    (main.py:59-61).  Stored per step: the three loss terms (captured at Model.loss, the values train_epoch sums), the learning rate the
    step ran at, and five weights after the step; per epoch: train_epoch's own running_loss / accuracy / detail."""
    import importlib.util
    import model.wav2vec2_linear_nll as M
    from oracle import wav2vec2 as W
    spec = importlib.util.spec_from_file_location("ref_main", os.path.join(REF, "main.py"))
    ref_main = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_main)
    cfg = W.W2VConfig.tiny()

    class Enc(torch.nn.Module):
        out_dim = cfg.embed

        def __init__(self):
            super().__init__()
            sd = W.init_state(cfg, seed=11)
            self.names = list(sd)
            self.model = torch.nn.ParameterDict({k.replace(".", "/"): torch.nn.Parameter(v) for k, v in sd.items()})

        def sd(self):
            return {k: self.model[k.replace(".", "/")] for k in self.names}

        def extract_feat(self, x, is_train=True):
            return W.forward(self.sd(), cfg, x)

    M.SSLModel = lambda device: Enc()
    torch.manual_seed(5)
    conf = {"model": {"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1}}
    m = M.Model(conf["model"], "cpu")
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    out = {}
    for k, v in m.state_dict().items():
        if not k.startswith("ssl_model"):
            out["sd:" + k] = v.detach().clone().numpy()
    max_lr, min_lr, wd = 4e-4, 1e-4, 1e-4          # CLI arguments of main.py (--max_lr / --min_lr / --weight_decay); large enough to move bf16 copies
    opt = torch.optim.AdamW(m.parameters(), lr=max_lr, weight_decay=wd)                                            # main.py:339
    sched = torch.optim.lr_scheduler.CyclicLR(opt, base_lr=min_lr, max_lr=max_lr, step_size_up=3, mode="exp_range", gamma=0.85,
                                              cycle_momentum=False)                                                # main.py:341
    g = torch.Generator().manual_seed(4321)
    V, L, n_epochs, n_steps = 4, 4000, 2, 3
    xs = 0.1 * torch.randn(n_epochs * n_steps, 1, L, V, generator=g)
    ys = torch.tensor([[1, 1, 0, 0], [1, 0, 0, 1], [0, 1, 1, 0], [1, 1, 0, 0], [0, 0, 1, 1], [1, 0, 1, 0]]).view(-1, 1, V)
    watch = {"LL.weight": m.LL.weight, "backend.m_utt_level.weight": m.backend.m_utt_level.weight,
             "ssl_model.model.encoder.layers.0.self_attn.q_proj.weight": m.ssl_model.model["encoder/layers/0/self_attn/q_proj/weight"],
             "ssl_model.model.encoder.layers.1.fc2.weight": m.ssl_model.model["encoder/layers/1/fc2/weight"],
             "ssl_model.model.feature_extractor.conv_layers.0.0.weight": m.ssl_model.model["feature_extractor/conv_layers/0/0/weight"]}
    rec = {"losses": [], "lr": [], "w": {k: [] for k in watch}}
    ref_loss = m.loss
    # the same six packs through the UNTRAINED weights (train mode, no step): what a trajectory that never moved would report — the
    # tests use the distance between this and the trained losses to show that their tolerance can tell the two apart
    m.train()
    frozen = []
    with torch.no_grad():
        for i in range(n_epochs * n_steps):
            o, f, e = m(xs[i].squeeze(0).transpose(0, 1))
            r = ref_loss(o, f, e, ys[i].view(-1), conf, "pack%d" % i)
            frozen.append([float(r[k]) for k in ("L_CE", "L_CF1", "L_CF2")])

    def spy_loss(*a, **kw):
        r = ref_loss(*a, **kw)
        rec["losses"].append([float(r[k]) for k in ("L_CE", "L_CF1", "L_CF2")])
        return r
    m.loss = spy_loss

    def post(optimizer, args, kwargs):
        rec["lr"].append(optimizer.param_groups[0]["lr"])
        for k, p in watch.items():
            rec["w"][k].append(p.detach().clone().numpy())
    opt.register_step_post_hook(post)
    epoch_rows = []
    for ep in range(n_epochs):
        loader = [("pack%d" % i, xs[i], ys[i]) for i in range(ep * n_steps, (ep + 1) * n_steps)]
        running, acc, detail = ref_main.train_epoch(loader, m, min_lr, opt, "cpu", conf)                            # main.py:408
        epoch_rows.append([running, acc] + [detail[k] for k in ("L_CE", "L_CF1", "L_CF2")])
        sched.step()                                                                                               # main.py:416
    assert len(rec["lr"]) == n_epochs * n_steps and m.training
    out.update(x=xs.numpy(), y=ys.numpy(), losses=np.array(rec["losses"]), losses_frozen=np.array(frozen), lr=np.array(rec["lr"], dtype=np.float64),
               epochs=np.array(epoch_rows), hyper=np.array([max_lr, min_lr, wd]))
    for k, lst in rec["w"].items():
        out["w:" + k] = np.stack(lst)
    np.savez_compressed(os.path.join(OUT, "trajectory.npz"), **out)
    print("trajectory.npz", len(out), "arrays; lr per step", rec["lr"], "loss totals", [round(sum(r), 5) for r in rec["losses"]])


def gen_w2v_hf():
    """Pin the wav2vec2 restatement against transformers.Wav2Vec2Model with copied weights."""
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    from oracle import wav2vec2 as W

    def hf_model(cfg):
        hc = Wav2Vec2Config(hidden_size=cfg.embed, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                            intermediate_size=cfg.ffn, conv_dim=[cfg.conv_dim] * 7, conv_kernel=list(cfg.conv_kernels),
                            conv_stride=list(cfg.conv_strides), feat_extract_norm="layer", do_stable_layer_norm=True,
                            conv_bias=True, num_conv_pos_embeddings=cfg.pos_k,
                            num_conv_pos_embedding_groups=cfg.pos_groups, hidden_dropout=0.0, attention_dropout=0.0,
                            activation_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0, final_dropout=0.0,
                            mask_time_prob=0.0, hidden_act="gelu", apply_spec_augment=False)
        return Wav2Vec2Model(hc).eval()

    cfg = W.W2VConfig.tiny()
    sd = W.init_state(cfg, seed=11)
    hf = hf_model(cfg)
    hsd = {k: v for k, v in W.to_hf_state(sd, cfg).items() if k in hf.state_dict()}
    missing, unexpected = hf.load_state_dict(hsd, strict=False)
    assert not unexpected and all("masked_spec" in k for k in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(1234)
    x = 0.1 * torch.randn(2, 4000, generator=g)
    with torch.no_grad():
        ho = hf(x, output_hidden_states=True)
        mine, inter = W.forward(sd, cfg, x, return_all=True)
    err = (mine - ho.last_hidden_state).abs().max().item()
    print("tiny wav2vec2 vs HF max abs err: %.3e" % err)
    assert err < 1e-4
    out = {"x": x.numpy(), "hf_last_hidden": ho.last_hidden_state.numpy(),
           "hf_extract_features": ho.extract_features.numpy(), "seed": np.array(11)}
    np.savez_compressed(os.path.join(OUT, "w2v_tiny_hf.npz"), **out)
    # XLS-R-300M shape (24 x 1024, 16 heads of 64, ffn 4096, pos-conv k = 128 in 16 groups — the shapes the fused attention and
    # positional-convolution kernels specialise on), 1 x 16000 samples, seeded random weights: a strided fingerprint of what
    # transformers.Wav2Vec2Model returns (final output, conv features, a few hidden states)
    cfg = W.W2VConfig()
    sd = W.init_state(cfg, seed=12)
    hf = hf_model(cfg)
    hf.load_state_dict({k: v for k, v in W.to_hf_state(sd, cfg).items() if k in hf.state_dict()}, strict=False)
    x = 0.1 * torch.randn(1, 16000, generator=torch.Generator().manual_seed(4321))
    with torch.no_grad():
        ho = hf(x, output_hidden_states=True)
        e = (W.forward(sd, cfg, x) - ho.last_hidden_state).abs().max().item()
    print("full-size wav2vec2 vs HF max abs err: %.3e (|out| max %.3f)" % (e, ho.last_hidden_state.abs().max().item()))
    assert e < 2e-4
    fp = {"x": x.numpy(), "seed": np.array(12), "last_hidden": ho.last_hidden_state[:, ::8, ::32].numpy(),
          "last_hidden_absmax": np.array(ho.last_hidden_state.abs().max().item()),
          "extract_features": ho.extract_features[:, ::8, ::16].numpy()}
    for li in (0, 1, 12, 24):      # hidden_states[0] = after the positional convolution, [n] = after layer n (pre final LayerNorm)
        fp["hidden%d" % li] = ho.hidden_states[li][:, ::8, ::32].numpy()
    np.savez_compressed(os.path.join(OUT, "w2v_xlsr_hf.npz"), **fp)
    print("w2v_xlsr_hf.npz", {k: v.shape for k, v in fp.items()})


def gen_eer():
    from evaluate_metrics import compute_eer
    out = {}
    # merges as Result.ipynb cells 4 and 13 do: LA19 protocol = "sid utt phy attack label" filtered to LA_E and
    # joined on the score file's utt with ".flac" stripped; In-the-Wild protocol = "path subset label"
    for tag, score_file, proto, keycol, strip in (
            ("la19_conf3", "docs/asvspoof2019_conf-3.txt", "DATA/asvspoof_2019_supcon/protocol.txt", 1, True),
            ("itw_conf3", "docs/inwild_conf-3.txt", "DATA/in_the_wild/protocol.txt", 0, False)):
        lab = {}
        with open(os.path.join(REF, proto)) as f:
            for line in f:
                p = line.split()
                if len(p) >= 3 and (not strip or "LA_E" in p[keycol]):
                    lab[p[keycol]] = p[-1]
        scores, labels = [], []
        with open(os.path.join(REF, score_file)) as f:
            for line in f:
                p = line.split()
                key = p[0].split(".")[0] if strip else p[0]
                if key in lab:
                    scores.append(float(p[1]))
                    labels.append(1 if lab[key] == "bonafide" else 0)
        s, l = np.array(scores), np.array(labels)
        full_eer, full_thr = compute_eer(s[l == 1], s[l == 0])
        print(tag, "full EER %.7f thr %.7f n=%d" % (full_eer, full_thr, len(s)))
        idx = np.arange(0, len(s), 20)
        ss, ll = s[idx], l[idx]
        eer, thr = compute_eer(ss[ll == 1], ss[ll == 0])
        out[tag + "_scores"] = ss
        out[tag + "_labels"] = ll.astype(np.int8)
        out[tag + "_eer_thr"] = np.array([eer, thr])
        out[tag + "_full_eer_thr_n"] = np.array([full_eer, full_thr, len(s)])
    np.savez_compressed(os.path.join(OUT, "eer.npz"), **out)
    print("eer.npz", len(out), "arrays")


def gen_aasist():
    """Reference wav2vec2_aasist.Model (graph-attention back-end) with an injected encoder; parameters filled by
    oracle/aasist.py::fill_state.  Case `eval`: model.eval().  Case `train`: model.train() with every Dropout p set to 0
    (BatchNorm batch statistics, running-stat updates and the discarded bn1 of Residual_block are exercised)."""
    import model.wav2vec2_aasist as M
    from oracle.aasist import fill_state
    E = 16

    class Enc(torch.nn.Module):
        out_dim = E

        def extract_feat(self, x):
            return x

    M.SSLModel = lambda device: Enc()
    cfg = {"filts": [128, [1, 32], [32, 32], [32, 64], [64, 64]], "gat_dims": [64, 32], "pool_ratios": [0.5, 0.5, 0.5, 0.5],
           "temperatures": [2.0, 2.0, 100.0, 100.0], "nclasses": 2}
    out = {}
    # size tags: "" = the small map the fixture has held since round 2 (T = 61 -> 42 x 20 map, 20 temporal nodes); "199:" = the BASELINE
    # size (T = 199 frames of a 64000-sample clip -> 42 x 66 map, 66 temporal nodes: top-k with 33 kept nodes, the fused kernels' 256-position
    # tiles and their kS + kT <= 64 LDS budget); "202:" = an odd map width (67 nodes).  The big cases store every parameter gradient:
    # whole when small, as (norm, sum, first 16 values) fingerprints otherwise.
    for tag, B, T, seed in (("", 4, 61, 11), ("199:", 4, 199, 12), ("202:", 2, 202, 13)):
        rs = np.random.RandomState(seed)
        x0 = rs.standard_normal((B, T, E)).astype(np.float32)
        wl = rs.standard_normal((B, 2)).astype(np.float32)
        wh = rs.standard_normal((B, 160)).astype(np.float32)
        out.update({tag + "x": x0, tag + "w_logits": wl, tag + "w_hidden": wh})
        for case in ("eval", "train"):
            m = M.Model({"aasist": cfg}, "cpu", is_train=True)
            sd = m.state_dict()
            filled = fill_state({k: tuple(v.shape) for k, v in sd.items()}, seed=5)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
            if case == "eval":
                m.eval()
            else:
                m.train()
                for mod in m.modules():
                    if isinstance(mod, torch.nn.Dropout):
                        mod.p = 0.0
            x = torch.from_numpy(x0).clone().requires_grad_(True)
            logits, hidden = m(x)
            (logits * torch.from_numpy(wl)).sum().add((hidden * torch.from_numpy(wh)).sum()).backward()
            pre = tag + case
            out[pre + ":logits"] = logits.detach().numpy()
            out[pre + ":hidden"] = hidden.detach().numpy()
            out[pre + ":grad_x"] = x.grad.numpy()
            if not tag:
                for k in ("LL.weight", "out_layer.weight", "pos_S", "master1", "encoder.0.0.conv1.weight", "encoder.3.0.conv2.weight",
                          "attention.0.weight", "GAT_layer_T.att_weight", "HtrgGAT_layer_ST11.att_weight12", "HtrgGAT_layer_ST22.proj_with_attM.weight",
                          "pool_hS1.proj.weight", "first_bn1.weight"):
                    out[pre + ":grad:" + k] = dict(m.named_parameters())[k].grad.numpy()
            else:
                for k, p_ in m.named_parameters():
                    if p_.grad is None:
                        continue
                    g = p_.grad.numpy()
                    if g.size <= 2048:
                        out[pre + ":grad:" + k] = g
                    else:
                        out[pre + ":gradfp:" + k] = np.concatenate([[np.sqrt((g.astype(np.float64) ** 2).sum()), g.astype(np.float64).sum()], g.flatten()[:16]]).astype(np.float64)
            if case == "train":
                for k in ("first_bn.running_mean", "encoder.1.0.bn1.running_mean", "encoder.1.0.bn1.running_var", "HtrgGAT_layer_ST12.bn.running_var"):
                    out[tag + "train:buf:" + k] = m.state_dict()[k].numpy()
    np.savez_compressed(os.path.join(OUT, "aasist.npz"), **out)
    print("aasist.npz", len(out), "arrays")


def gen_resnet():
    """Reference wav2vec2_resnet_nll.Model (pre-activation ResNet-18 back-end) with an injected encoder; parameters filled by
    oracle/aasist.py::fill_state.  Cases `eval` / `train` (BatchNorm batch statistics; the model has no dropout), with the
    reference's own Model.loss terms (CrossEntropy and SupCon without the linear plugin's 1/bz)."""
    import model.wav2vec2_resnet_nll as M
    from oracle.aasist import fill_state
    E = 16

    class Enc(torch.nn.Module):
        out_dim = E

        def extract_feat(self, x, is_train=True):
            return x

    M.SSLModel = lambda device: Enc()
    args = {"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1,
            "resnet": {"num_nodes": 3, "enc_dim": 256, "resnet_type": "18", "nclasses": 2}}
    out = {}
    B, T = 4, 70
    rs = np.random.RandomState(13)
    x0 = rs.standard_normal((B, T, E)).astype(np.float32)
    y = np.array([1, 1, 0, 0], dtype=np.int64)
    out.update(x=x0, y=y)
    for case in ("eval", "train"):
        m = M.Model(args, "cpu", is_train=True)
        sd = m.state_dict()
        filled = fill_state({k: tuple(v.shape) for k, v in sd.items()}, seed=7)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
        m.eval() if case == "eval" else m.train()
        x = torch.from_numpy(x0).clone().requires_grad_(True)
        logits, feats, emb = m(x)
        losses = m.loss(logits, feats, emb, torch.from_numpy(y), {"model": args})
        sum(losses.values()).backward()
        out[case + ":logits"] = logits.detach().numpy()
        out[case + ":emb"] = emb.detach().numpy()
        out[case + ":feats"] = feats.detach().numpy()
        out[case + ":grad_x"] = x.grad.numpy()
        for k, v in losses.items():
            out[case + ":loss:" + k] = np.array(v.item())
        pd = dict(m.named_parameters())
        for k in ("LL.weight", "first_bn.weight", "resnet.conv1.weight", "resnet.layer1.0.conv1.weight", "resnet.layer2.0.shortcut.0.weight",
                  "resnet.layer4.1.bn2.weight", "resnet.fc.weight"):
            out[case + ":grad:" + k] = pd[k].grad.numpy()
        g5 = pd["resnet.conv5.weight"].grad        # 393 k elements: keep a fingerprint (norm, sum, first 16 values)
        out[case + ":gradfp:resnet.conv5.weight"] = np.concatenate([[g5.norm().item(), g5.sum().item()], g5.flatten()[:16].numpy()]).astype(np.float64)
        if case == "train":
            for k in ("first_bn.running_mean", "resnet.bn1.running_var", "resnet.layer3.0.bn1.running_mean"):
                out["train:buf:" + k] = m.state_dict()[k].numpy()
    np.savez_compressed(os.path.join(OUT, "resnet.npz"), **out)
    print("resnet.npz", len(out), "arrays")


def gen_audio_int16():
    """A7 (and the int16 conversion both A7 and A8 rest on), executed FROM THE REFERENCE: ReverbAugmentor.transform
    (datautils/audio_augmentor/reverb.py:33-44) and librosa_to_pydub / pydub_to_librosa (utils.py:20-30).  pydub and librosa are not
    installed, so two inert stand-ins are injected: pydub.AudioSegment reduced to a sample container (raw bytes in,
    get_array_of_samples() out — no arithmetic of its own) and librosa.load returning the RIR array it is handed.  Every arithmetic
    line that runs — np.convolve on float32, the peak normalisation, `np.array(x * (1<<15), dtype=np.int16)` — is the reference's.
    Also stored: the float64 convolution of the same inputs, so that tests can tell a genuine deviation from a float32-rounding tie
    at an int16 truncation boundary."""
    import array
    import datautils.audio_augmentor.reverb as R
    import datautils.audio_augmentor.utils as U

    class Seg:
        def __init__(self, data, frame_rate, sample_width, channels):
            assert sample_width == 2 and channels == 1
            self.raw = bytes(data)

        def get_array_of_samples(self):
            return array.array("h", self.raw)

    U.AudioSegment = Seg
    out = {}
    rs = np.random.RandomState(11)
    cases = {"short": (5000, 900, 120.0), "clip16000": (16000, 4000, 400.0), "long_rir": (12000, 8000, 1500.0)}
    for name, (L, Rn, tau) in cases.items():
        sp = (0.1 * rs.randn(L)).astype(np.float32)
        rir = (np.exp(-np.arange(Rn) / tau) * rs.randn(Rn)).astype(np.float32)
        R.librosa.load = lambda path, sr, _r=rir: (_r, sr)
        aug = object.__new__(R.ReverbAugmentor)
        aug.sr, aug.data, aug.rir_file = 16000, sp.copy(), "rir.wav"
        aug.transform()
        res = U.pydub_to_librosa(aug.augmented_audio)
        assert res.dtype == np.int16 and res.shape == (L + Rn - 1,)
        out[name + ":speech"], out[name + ":rir"], out[name + ":out"] = sp, rir, res
        out[name + ":conv64"] = np.convolve(sp.astype(np.float64), rir.astype(np.float64))
    conv_in = np.array([1.0, -1.0, 0.99999, -0.00002, 0.5, 0.25, -0.75, 3.0517578125e-05, -3.0517578125e-05, 0.999969482421875], np.float32)
    out["conv:in"] = conv_in
    with np.errstate(all="ignore"):
        out["conv:out"] = U.pydub_to_librosa(U.librosa_to_pydub(conv_in))
    np.savez_compressed(os.path.join(OUT, "audio_int16.npz"), **out)
    print("audio_int16.npz", len(out), "arrays")


def gen_audioop():
    """A8 + the speed augmenter: pydub (absent) does its integer arithmetic in CPython's `audioop` (rms / mul / add), which IS in
    this container's standard library.  Vectors from the real module: the three primitives on random / saturating / silent /
    extreme inputs, the MUSAN-overlay chain of background_noise.py:40-56 built from them, and pydub's speedup() sequence
    (oracle.audio_speed_pitch.Seg) with its two arithmetic hooks swapped for the real audioop.mul / audioop.add."""
    import audioop
    import math
    from oracle import audio_speed_pitch as SP
    b = lambda a: np.asarray(a, dtype="<i2").tobytes()
    u = lambda raw: np.frombuffer(raw, dtype="<i2").copy()
    rs = np.random.RandomState(20261003)
    L = 16000
    t = np.arange(L)
    speech = np.round(5000 * np.sin(2 * np.pi * t / 61.0) * np.sin(2 * np.pi * t / 2300.0) + 300 * rs.randn(L)).astype(np.int16)
    inputs = {"speech": speech, "loud": rs.randint(-32768, 32768, L).astype(np.int16), "silent": np.zeros(L, np.int16),
              "tiny": rs.randint(-2, 3, 4001).astype(np.int16), "extremes": np.tile(np.array([-32768, 32767, -1, 0, 1, -32767], np.int16), 500),
              "one": np.array([-7], np.int16)}
    factors = np.array([0.0, 1e-6, 0.5, 0.999, 1.0, 1.7, 3.3, 10 ** (-7.3 / 20), 10 ** (9.1 / 20), 10 ** (-120 / 20), 100.0, -0.75], np.float64)
    out = {"factors": factors}
    for name, x in inputs.items():
        out["in:" + name] = x
        out["rms:" + name] = np.array(audioop.rms(b(x), 2), np.int64)
        out["mul:" + name] = np.stack([u(audioop.mul(b(x[:4000]), 2, float(f))) for f in factors])      # on the first 4000 samples
        other = np.roll(inputs["loud"], 17)[: len(x)] if len(x) <= L else None
        out["add:" + name] = u(audioop.add(b(x), b(other), 2))
    out["rms:empty"] = np.array(audioop.rms(b"", 2), np.int64)
    # background_noise.py:40-56 from the real primitives (dBFS as pydub.utils.ratio_to_db: 20 * log(rms / 32768, 10))
    for tag, sp, noise, snr in (("bn0", speech, (inputs["loud"] // 9).astype(np.int16), 7), ("bn1", (speech // 3).astype(np.int16), inputs["loud"][:9000], 15),
                                ("bn2", inputs["loud"], np.concatenate([speech, speech])[:24000], 5)):
        db = lambda x: 20 * math.log(audioop.rms(b(x), 2) / 32768.0, 10)
        gain = snr * db(noise) / db(sp)
        g = u(audioop.mul(b(sp), 2, 10 ** (gain / 20.0)))
        n = min(len(g), len(noise))
        res = g.copy()
        res[:n] = u(audioop.add(b(g[:n]), b(noise[:n]), 2))
        out[tag + ":speech"], out[tag + ":noise"], out[tag + ":snr"], out[tag + ":gain"], out[tag + ":out"] = sp, noise, np.array(snr), np.array(gain), res
    # pydub speedup() with audioop doing the arithmetic
    keep = SP._mul, SP._add
    SP._mul = lambda frames, factor: u(audioop.mul(b(frames), 2, float(factor)))
    SP._add = lambda a, c: u(audioop.add(b(a), b(c), 2))
    try:
        for i, (src, fac) in enumerate((("speech", 0.9), ("speech", 0.9713), ("speech", 1.0461), ("speech", 1.1), ("loud", 0.93), ("loud", 1.08))):
            x = inputs[src][:12000] if src == "loud" else inputs[src]
            out["speed%d:in" % i], out["speed%d:factor" % i] = x, np.array(fac)
            out["speed%d:out" % i] = SP.speedup(SP.Seg(x, 16000), fac).f
    finally:
        SP._mul, SP._add = keep
    np.savez_compressed(os.path.join(OUT, "audioop.npz"), **out)
    print("audioop.npz", len(out), "arrays")


def gen_conformer():
    """Reference model/conformer.py::ConformerBlock (imported from the file: the module needs einops only).  Parameters by
    oracle/aasist.py::fill_state, four cases: `a` train mode (gradients of the small tensors whole, fingerprints = norm, sum, first 16 values of the large ones); `b` eval mode with a key / query mask (one
    utterance fully masked: the uniform rows of masked_fill(-max)); `c` the causal form (no BatchNorm, even kernel, padding (k - 1, 0));
    `d` n = 600 > max_pos_emb + 1, so that the distances clamp at both ends of the relative-position table.  The tests rebuild the
    parameters with the same fill_state(shapes, seed) call.  Also the state dict a
    seeded construction gives (a fingerprint), for the "same seed, same initial weights" property of the HIP block."""
    import importlib.util
    from oracle.aasist import fill_state
    spec = importlib.util.spec_from_file_location("ref_conformer", os.path.join(REF, "model", "conformer.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    cases = {
        "a": dict(cfg=dict(dim=64, dim_head=16, heads=4), B=3, n=50, train=True, mask=False),
        "b": dict(cfg=dict(dim=64, dim_head=16, heads=4), B=3, n=37, train=False, mask=True),
        "c": dict(cfg=dict(dim=32, dim_head=32, heads=2, ff_mult=2, conv_kernel_size=8, conv_causal=True), B=2, n=21, train=True, mask=False),
        "d": dict(cfg=dict(dim=32, dim_head=16, heads=2, ff_mult=2, conv_expansion_factor=1, conv_kernel_size=5), B=1, n=600, train=True, mask=False),
    }
    out = {}
    for name, c in cases.items():
        blk = M.ConformerBlock(**c["cfg"])
        sd = blk.state_dict()
        filled = fill_state({k: tuple(v.shape) for k, v in sd.items()}, seed=31 + ord(name))
        blk.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
        blk.train() if c["train"] else blk.eval()
        rs = np.random.RandomState(100 + ord(name))
        x0 = rs.standard_normal((c["B"], c["n"], c["cfg"]["dim"])).astype(np.float32)
        wout = rs.standard_normal(x0.shape).astype(np.float32)
        mask = None
        if c["mask"]:
            mk = rs.rand(c["B"], c["n"]) > 0.3
            mk[1, :] = False
            mask = torch.from_numpy(mk)
            out[name + ":mask"] = mk
        x = torch.from_numpy(x0).clone().requires_grad_(True)
        y = blk(x, mask=mask)
        (y * torch.from_numpy(wout)).sum().backward()
        out[name + ":cfg"] = np.array(repr(sorted(c["cfg"].items())))
        out[name + ":x"], out[name + ":wout"], out[name + ":y"], out[name + ":grad_x"] = x0, wout, y.detach().numpy(), x.grad.numpy()
        out[name + ":seed"] = np.array(31 + ord(name))          # the parameters are fill_state(shapes, seed): not stored
        for k, p_ in blk.named_parameters():
            g = p_.grad.numpy()
            if g.size <= (8192 if name == "a" else 4096) or (k == "attn.fn.rel_pos_emb.weight" and name in "ad"):
                out[name + ":grad:" + k] = g
            else:
                out[name + ":gradfp:" + k] = np.concatenate([[np.sqrt((g.astype(np.float64) ** 2).sum()), g.astype(np.float64).sum()], g.flatten()[:16]]).astype(np.float64)
        if c["train"] and not c["cfg"].get("conv_causal", False):
            for k in ("conv.net.5.running_mean", "conv.net.5.running_var", "conv.net.5.num_batches_tracked"):
                out[name + ":buf:" + k] = blk.state_dict()[k].numpy()
    torch.manual_seed(1234)
    blk = M.ConformerBlock(dim=64, dim_head=16, heads=4)
    out["init:keys"] = np.array(list(blk.state_dict().keys()))
    out["init:fp"] = np.array([float(v.double().sum()) for v in blk.state_dict().values()])
    np.savez_compressed(os.path.join(OUT, "conformer.npz"), **out)
    print("conformer.npz", len(out), "arrays")


def gen_btse():
    """Reference model/wav2vec2_btse/model.py::Model (BASELINE configs[4]) — imported with a stand-in for the `biosegment` package (a
    dangling symlink in the reference: only the tokeniser class `Wav2bioCNN` lives there, and Model.forward takes the bio tokens as an
    argument) and the SSL encoder injected as identity, as for the other plugins.  Parameters by oracle/aasist.py::fill_state over the
    reference's own state-dict shapes.  model.train() with every Dropout p = 0 (the plugin has no BatchNorm: train == eval then).
    Cases: `cat` (conf-5-btse-trans64.yaml: concat, lengths 17 / 9 / 1 / 17 of 17 — full, padded tail, length 1), `add` (is_add with
    bio_out = 128, the only width at which the reference's fc2 accepts the sum), `short` (3 tokens < window + 1: the slice branch of
    _get_relative_embeddings), `one` (a single token)."""
    import importlib
    from oracle.aasist import fill_state
    st = types.ModuleType("model.wav2vec2_btse.biosegment")
    st.Wav2bioCNN = type("Wav2bioCNN", (), {"__init__": lambda self, device=None: None})
    sys.modules["model.wav2vec2_btse.biosegment"] = st
    M = importlib.import_module("model.wav2vec2_btse.model")          # `model.wav2vec2_btse` the attribute is the class (model/__init__.py:2)
    BK = importlib.import_module("model.wav2vec2_btse.backend")
    E = 16

    class Enc(torch.nn.Module):
        out_dim = E

        def extract_feat(self, x, is_train=True):
            return x

    BK.SSLModel = lambda device: Enc()
    base = dict(flag_fix_ssl=False, contra_mode="all", loss_type=1, bio_out=64, nb_classes=2, is_add=False, bio_hid=256, n_heads=4,
                pf_dim=128, n_layers=3, n_bios=3, bio_dim=32)
    cases = {"cat": dict(args={}, B=4, T=20, L=17, lens=[17, 9, 1, 17]),
             "add": dict(args=dict(is_add=True, bio_out=128), B=3, T=7, L=12, lens=[12, 12, 5]),
             "short": dict(args={}, B=2, T=5, L=3, lens=[3, 2]),
             "one": dict(args={}, B=2, T=6, L=1, lens=[1, 1])}
    out = {}
    for ci, (name, c) in enumerate(cases.items()):
        args = dict(base, **c["args"])
        m = M.Model(args, "cpu")
        sd = m.state_dict()
        filled = fill_state({k: tuple(v.shape) for k, v in sd.items()}, seed=41 + ci)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
        m.train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        rs = np.random.RandomState(200 + ci)
        x0 = rs.standard_normal((c["B"], c["T"], E)).astype(np.float32)
        bio = rs.randint(0, 3, size=(c["B"], c["L"])).astype(np.int32)
        lens = np.array(c["lens"], dtype=np.int32)
        x = torch.from_numpy(x0).clone().requires_grad_(True)
        logp, feats, b = m(x, torch.from_numpy(bio), torch.from_numpy(lens))
        wl, wf, wb = (rs.standard_normal(tuple(t.shape)).astype(np.float32) for t in (logp, feats, b))
        (logp * torch.from_numpy(wl)).sum().add((feats * torch.from_numpy(wf)).sum()).add((b * torch.from_numpy(wb)).sum()).backward()
        out[name + ":keys"] = np.array(list(sd.keys()))
        out[name + ":shapes"] = np.array([repr(tuple(v.shape)) for v in sd.values()])
        out[name + ":args"] = np.array(repr(sorted(c["args"].items())))
        out[name + ":seed"] = np.array(41 + ci)
        out.update({name + ":x": x0, name + ":bio": bio, name + ":lens": lens, name + ":w_logp": wl, name + ":w_feats": wf, name + ":w_b": wb,
                    name + ":logp": logp.detach().numpy(), name + ":feats": feats.detach().numpy(), name + ":b": b.detach().numpy(),
                    name + ":grad_x": x.grad.numpy()})
        for k, p_ in m.named_parameters():
            if p_.grad is None:
                assert "m_utt_level" in k, k          # its logits are discarded (model.py:324): no gradient, AdamW leaves it alone
                continue
            g = p_.grad.numpy()
            if g.size <= 4224:
                out[name + ":grad:" + k] = g
            else:
                out[name + ":gradfp:" + k] = np.concatenate([[np.sqrt((g.astype(np.float64) ** 2).sum()), g.astype(np.float64).sum()], g.flatten()[:16]]).astype(np.float64)
        m.is_train = False
        assert torch.equal(m(torch.from_numpy(x0), torch.from_numpy(bio), torch.from_numpy(lens)), logp.detach())   # eval contract: log-probs only
    np.savez_compressed(os.path.join(OUT, "btse.npz"), **out)
    print("btse.npz", len(out), "arrays")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])     # e.g. `python oracle/gen_golden.py aasist` regenerates one fixture
    want = lambda name: not only or name in only
    if want("w2v_hf"):
        gen_w2v_hf()          # before the stand-in modules exist: transformers probes for torchaudio
    import_reference()
    for name, fn in (("rawboost", gen_rawboost), ("multiview", gen_multiview), ("pack", gen_pack), ("pack_variants", gen_pack_variants),
                     ("head_loss", gen_head_loss), ("train_step", gen_train_step), ("trajectory", gen_trajectory), ("eer", gen_eer), ("aasist", gen_aasist), ("resnet", gen_resnet), ("audio_int16", gen_audio_int16), ("audioop", gen_audioop), ("conformer", gen_conformer), ("btse", gen_btse)):
        if want(name):
            fn()
