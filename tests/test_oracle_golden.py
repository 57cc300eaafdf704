"""Pins the oracle (oracle/*.py) to vectors produced by the reference's own Python
(oracle/gen_golden.py, run in the build container) — CPU only."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import audio_int16 as AI
from oracle import audio_speed_pitch as SP
from oracle import eer as OE
from oracle import head as OH
from oracle import multiview as OM
from oracle import rawboost as RB
from oracle import wav2vec2 as W

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_rawboost_all_algos_match_reference_bit_for_bit_in_draw_order():
    g = load("rawboost.npz")
    for seed in (0, 1):
        x = g["x_s%d" % seed]
        for algo in range(0, 9):
            np.random.seed(1000 * algo + seed)
            y = RB.process_rawboost_feature(x, 16000, RB.RawBoostArgs(), algo)
            ref = g["algo%d_s%d" % (algo, seed)]
            assert y.shape == ref.shape
            np.testing.assert_allclose(y, ref, rtol=0, atol=1e-12, err_msg="algo %d seed %d" % (algo, seed))


def test_rawboost_lnl_long_and_notch_and_fir():
    g = load("rawboost.npz")
    a = RB.RawBoostArgs()
    np.random.seed(77)
    y = RB.lnl(g["x_long"], a.N_f, a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff, a.minG, a.maxG,
               a.minBiasLinNonLin, a.maxBiasLinNonLin, 16000)
    np.testing.assert_allclose(y, g["lnl_long"], rtol=0, atol=1e-12)
    assert np.abs(y).max() <= 1.0 + 1e-12  # normWav(.,0) triggered on the loud clip
    for seed in (3, 4, 5):
        np.random.seed(seed)
        b = RB.gen_notch_coeffs(a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff, -5, -20, 16000)
        np.testing.assert_allclose(b, g["notch_b_s%d" % seed], rtol=0, atol=1e-15)
    np.testing.assert_allclose(RB.filter_fir(g["x_s0"], g["notch_b_s3"]), g["fir_y"], rtol=0, atol=1e-13)


def test_eval_pad():
    g = load("rawboost.npz")
    np.testing.assert_array_equal(OM.pad_eval(g["pad_in"], "repeat", 2000), g["pad_repeat"])
    np.testing.assert_array_equal(OM.pad_eval(g["pad_in"], "zero", 2000), g["pad_zero"])
    assert OM.pad_eval(np.zeros(2500), "zero", 2000).shape == g["pad_cut"].shape


def test_multiview_pad():
    g = load("multiview.npz")
    for name, n in (("longer", 4), ("shorter", 4), ("exact", 3)):
        views = [g["%s_in%d" % (name, i)] for i in range(n)]
        for rp in (False, True):
            np.random.seed(5)
            res = OM.batch_pad_for_multiview(views, 16000, 2000, random_trim_nosil=True, repeat_pad=rp)
            for i, v in enumerate(res):
                np.testing.assert_array_equal(v, g["%s_rp%d_out%d" % (name, int(rp), i)])


def _head_sd(g):
    return {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd:") and "first_bn" not in k}


def test_head_and_loss_match_reference_model():
    g = load("head_loss.npz")
    sd = _head_sd(g)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = torch.from_numpy(g["y"])
    for p in sd.values():
        p.requires_grad_(True)
    out, feats, emb = OH.head_forward(sd, x)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], atol=1e-6)
    np.testing.assert_allclose(feats.detach().numpy(), g["feats"], atol=1e-6)
    np.testing.assert_allclose(emb.detach().numpy(), g["emb"], atol=1e-6)
    losses = OH.model_loss(out, feats, emb, y, 1)
    for k in ("L_CE", "L_CF1", "L_CF2"):
        assert abs(losses[k].item() - float(g["loss:" + k])) <= 2e-6 * max(1.0, abs(float(g["loss:" + k])))
    sum(losses.values()).backward()
    np.testing.assert_allclose(x.grad.numpy(), g["grad_x"], atol=2e-6, rtol=1e-4)
    np.testing.assert_allclose(sd["LL.weight"].grad.numpy(), g["grad_LL_weight"], atol=2e-6, rtol=1e-4)
    np.testing.assert_allclose(sd["backend.m_utt_level.weight"].grad.numpy(), g["grad_utt_weight"], atol=2e-6, rtol=1e-4)
    np.testing.assert_allclose(sd["backend.m_frame_level.0.weight"].grad.numpy(), g["grad_frame0_weight"], atol=2e-6, rtol=1e-4)


def supcon_case_inputs(bz, T2, d, seed):
    torch.manual_seed(int(seed))
    f = torch.randn(int(bz), 1, int(T2), int(d))
    lab = torch.tensor(([1] * ((5 * int(bz) + 10) // 11) + [0] * int(bz))[:int(bz)])
    return f, lab


def test_supcon_known_answers():
    g = load("head_loss.npz")
    for bz, T2, d, seed, loss, gnorm, g00, gll in g["supcon_cases"]:
        f, lab = supcon_case_inputs(bz, T2, d, seed)
        f.requires_grad_(True)
        l = OH.supcon_loss(f, lab)
        l.backward()
        assert abs(l.item() - loss) <= 1e-5 * max(1.0, abs(loss))
        assert abs(f.grad.norm().item() - gnorm) <= 1e-4 * gnorm
        assert abs(f.grad[0, 0, 0, 0].item() - g00) <= 1e-4 * abs(g00) + 1e-7
    torch.manual_seed(9)
    f = torch.randn(4, 1, 10, 8)
    assert np.isnan(float(g["supcon_nan"])) and torch.isnan(OH.supcon_loss(f, torch.tensor([1, 0, 0, 0])))


def test_wav2vec2_restatement_matches_transformers():
    g = load("w2v_tiny_hf.npz")
    cfg = W.W2VConfig.tiny()
    sd = W.init_state(cfg, seed=int(g["seed"]))
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        y, inter = W.forward(sd, cfg, x, return_all=True)
    # HF's `extract_features` is the conv-stack output AFTER feature_projection.layer_norm
    ln = torch.nn.functional.layer_norm(inter["conv"][-1], (cfg.conv_dim,), sd["layer_norm.weight"], sd["layer_norm.bias"], 1e-5)
    np.testing.assert_allclose(ln.numpy(), g["hf_extract_features"], atol=2e-5)
    np.testing.assert_allclose(y.numpy(), g["hf_last_hidden"], atol=1e-4)
    assert y.shape[1] == cfg.out_len(x.shape[1])
    full = W.W2VConfig()
    assert [full.out_len(L) for L in (16000, 64000, 64600)] == [49, 199, 201]
    n_train = sum(int(np.prod(s)) for _, s, tr in W.param_shapes(full) if tr)
    assert abs(n_train - 315.4e6) < 0.3e6


def test_wav2vec2_restatement_matches_transformers_at_xlsr_300m_shape():
    """The shapes the fused attention / positional-convolution kernels specialise on (head dim 64, 16 groups, k = 128, 24 layers):
    tests/golden/w2v_xlsr_hf.npz is a strided fingerprint of transformers.Wav2Vec2Model's output for 1 x 16000 samples with the
    oracle's seeded weights copied in (oracle/gen_golden.py::gen_w2v_hf, build container)."""
    g = load("w2v_xlsr_hf.npz")
    cfg = W.W2VConfig()
    sd = W.init_state(cfg, seed=int(g["seed"]))
    with torch.no_grad():
        y, inter = W.forward(sd, cfg, torch.from_numpy(g["x"]), return_all=True)
    assert y.shape == (1, 49, 1024) and abs(y.abs().max().item() - float(g["last_hidden_absmax"])) < 1e-4
    np.testing.assert_allclose(y[:, ::8, ::32].numpy(), g["last_hidden"], atol=1e-4)
    ln = torch.nn.functional.layer_norm(inter["conv"][-1], (cfg.conv_dim,), sd["layer_norm.weight"], sd["layer_norm.bias"], 1e-5)
    np.testing.assert_allclose(ln[:, ::8, ::16].numpy(), g["extract_features"], atol=2e-5)
    # hidden_states[0] = after the positional convolution, [n] = after layer n; HF's last entry is already layer-normed
    np.testing.assert_allclose(inter["pos"][:, ::8, ::32].numpy(), g["hidden0"], atol=1e-4)
    for n in (1, 12):
        ref = g["hidden%d" % n]
        np.testing.assert_allclose(inter["layers"][n - 1][:, ::8, ::32].numpy(), ref, atol=1e-4 * max(1.0, float(np.abs(ref).max())))
    np.testing.assert_allclose(g["hidden24"], g["last_hidden"], atol=0)


def test_train_step_matches_reference_train_epoch_iteration():
    g = load("train_step.npz")
    cfg = W.W2VConfig.tiny()
    ssl = W.init_state(cfg, seed=11)
    head = {k[3:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("sd:") and "first_bn" not in k}
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    losses, grads, (out, feats, emb), _ = OH.train_step(ssl, head, cfg, x, y, loss_type=1, lr=1e-3, wd=1e-4)
    np.testing.assert_allclose(out.numpy(), g["out"], atol=1e-5)
    np.testing.assert_allclose(emb.numpy(), g["emb"], atol=1e-5)
    for k in ("L_CE", "L_CF1", "L_CF2"):
        assert abs(losses[k] - float(g["loss:" + k])) <= 1e-5 * max(1.0, abs(float(g["loss:" + k])))
    assert abs(sum(losses.values()) - float(g["total_loss"])) < 1e-4
    for k in g.files:
        if k.startswith("grad:"):
            ref = g[k]
            got = grads[k[5:]].numpy()
            scale = max(np.abs(ref).max(), 1e-8)
            assert np.abs(got - ref).max() / scale < 2e-4, k
        if k.startswith("post:"):
            name = k[5:]
            got = (ssl[name[len("ssl_model.model."):]] if name.startswith("ssl_model.model.") else head[name]).detach().numpy()
            np.testing.assert_allclose(got, g[k], atol=2e-6, err_msg=k)


def test_six_step_trajectory_matches_the_reference_train_epoch_and_scheduler():
    """tests/golden/trajectory.npz = six optimizer steps through the reference's own train_epoch + per-epoch CyclicLR (gen_trajectory).
    The oracle's train_step, carried from step to step with ONE AdamW and the scheduler stepped after every third pack, must land on
    the same losses, learning rates and weights — what pins steps 2..6 (bias correction at t > 1, the scheduler writing lr)."""
    g = load("trajectory.npz")
    cfg = W.W2VConfig.tiny()
    ssl = W.init_state(cfg, seed=11)
    head = {k[3:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("sd:") and "first_bn" not in k}
    max_lr, min_lr, wd = (float(v) for v in g["hyper"])
    opt = sched = None
    step = 0
    for ep in range(2):
        for i in range(3):
            x = torch.from_numpy(g["x"][step, 0]).transpose(0, 1).contiguous()          # [1, L, V] pack -> [V, L] (main.py:59-61)
            y = torch.from_numpy(g["y"][step]).view(-1)
            if opt is None:
                # the first call builds the optimizer over the oracle's tensors (AdamW(lr=max_lr), then CyclicLR writes base_lr: main.py:339-341);
                # it must exist BEFORE the first step runs, so do a zero-cost construction through train_step's own path
                params = [ssl[n] for n, _, tr in W.param_shapes(cfg) if tr] + list(head.values())
                opt = torch.optim.AdamW(params, lr=max_lr, weight_decay=wd)
                sched = torch.optim.lr_scheduler.CyclicLR(opt, base_lr=min_lr, max_lr=max_lr, step_size_up=3, mode="exp_range", gamma=0.85,
                                                          cycle_momentum=False)
            assert abs(opt.param_groups[0]["lr"] - float(g["lr"][step])) < 1e-15
            losses, _, _, _ = OH.train_step(ssl, head, cfg, x, y, loss_type=1, opt_state=opt)
            for j, k in enumerate(("L_CE", "L_CF1", "L_CF2")):
                ref = float(g["losses"][step, j])
                assert abs(losses[k] - ref) <= 2e-4 * max(1.0, abs(ref)), (step, k, losses[k], ref)
            for k in g.files:
                if k.startswith("w:"):
                    name = k[2:]
                    got = (ssl[name[len("ssl_model.model."):]] if name.startswith("ssl_model.model.") else head[name]).detach().numpy()
                    ref = g[k][step]
                    assert np.abs(got - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-3) + 1e-6, (step, name, np.abs(got - ref).max())
            step += 1
        sched.step()
    ep_rows = g["epochs"]          # train_epoch's own returns: running loss = the sum of the three steps' totals
    assert abs(ep_rows[0, 0] - g["losses"][:3].sum()) < 1e-4 and abs(ep_rows[1, 0] - g["losses"][3:].sum()) < 1e-4


def test_eer_known_answers():
    g = load("eer.npz")
    for tag, full in (("la19_conf3", 0.0288370), ("itw_conf3", 0.0450978)):
        s, l = g[tag + "_scores"], g[tag + "_labels"]
        eer, thr = OE.compute_eer(s[l == 1], s[l == 0])
        assert abs(eer - g[tag + "_eer_thr"][0]) < 1e-12 and abs(thr - g[tag + "_eer_thr"][1]) < 1e-12
        assert abs(g[tag + "_full_eer_thr_n"][0] - full) < 5e-8   # Result.ipynb: 2.8837 % / 4.5098 %


def test_int16_semantics_corner_cases():
    # +1.0 wraps to -32768, truncation toward zero, saturating add, floor-after-clip gain
    assert AI.librosa_to_int16(np.array([1.0, -1.0, 0.99999, -0.00002, 0.5])).tolist() == [-32768, -32768, 32767, 0, 16384]
    assert AI.overlay(np.array([32000, -32000, 5], np.int16), np.array([1000, -1000], np.int16)).tolist() == [32767, -32768, 5]
    # audioop.mul: floor AFTER the clip, on the double product (x * 1.5 = +-4.5 / 10.5: floor, not truncation; +-45000 clip first)
    assert AI.apply_gain(np.array([3, -3, 7, 30000, -30000], np.int16), 20 * np.log10(1.5)).tolist() == [4, -5, 10, 32767, -32768]
    assert AI.rms_int(np.array([3, 4], np.int16)) == 3
    sp = (0.1 * np.random.RandomState(0).randn(1000)).astype(np.float32)
    noise = (300 * np.random.RandomState(1).randn(800)).astype(np.int16)
    out, gain = AI.background_noise(sp, noise, 10)
    assert out.dtype == np.int16 and out.shape == (1000,) and np.isfinite(gain)
    rv = AI.reverb(sp, np.exp(-np.arange(200) / 30.0).astype(np.float32))
    assert rv.shape == (1199,) and rv.dtype == np.int16 and (np.abs(rv.astype(np.int32)).max() in (32767, 32768))


def test_reverb_and_int16_conversion_match_the_reference_run():
    """tests/golden/audio_int16.npz was produced by the REFERENCE's ReverbAugmentor.transform / librosa_to_pydub / pydub_to_librosa
    (oracle/gen_golden.py::gen_audio_int16; pydub.AudioSegment stood in by a bare sample container).  The oracle restatement must
    reproduce it bit for bit: same np.convolve on float32, same peak normalisation, same C cast with wrap-around."""
    g = np.load(os.path.join(G, "audio_int16.npz"))
    assert AI.librosa_to_int16(g["conv:in"]).tolist() == g["conv:out"].tolist()
    for name in ("short", "clip16000", "long_rir"):
        assert np.array_equal(AI.reverb(g[name + ":speech"], g[name + ":rir"]), g[name + ":out"]), name


def test_int16_oracle_matches_cpython_audioop():
    """tests/golden/audioop.npz holds outputs of CPython 3.10's own audioop.rms / mul / add (the C code pydub 0.25.1 calls from
    AudioSegment.rms / apply_gain / overlay) on random, saturating, silent and extreme inputs, and of the MUSAN-overlay chain of
    background_noise.py:40-56 built from them (oracle/gen_golden.py::gen_audioop, build container).  The restatement must agree
    bit for bit."""
    g = np.load(os.path.join(G, "audioop.npz"))
    names = [k[3:] for k in g.files if k.startswith("in:")]
    assert len(names) >= 6
    assert AI.rms_int(np.zeros(0, np.int16)) == int(g["rms:empty"]) == 0
    for name in names:
        x = g["in:" + name]
        assert AI.rms_int(x) == int(g["rms:" + name]), name
        for f, want in zip(g["factors"], g["mul:" + name]):
            got = SP._mul(x[:4000], f)
            assert np.array_equal(got, want), (name, f)
            db = 20 * math.log(f, 10) if f > 0 else None
            if db is not None and 10.0 ** (db / 20.0) == f:          # apply_gain goes through dB: where the round trip is exact
                assert np.array_equal(AI.apply_gain(x[:4000], db), want), (name, f)
        other = np.roll(g["in:loud"], 17)[: len(x)]
        assert np.array_equal(AI.overlay(x, other), g["add:" + name]), name
        assert np.array_equal(SP._add(x, other), g["add:" + name]), name
    for tag in ("bn0", "bn1", "bn2"):
        out, gain = AI.background_noise(g[tag + ":speech"].astype(np.float64) / 32768.0, g[tag + ":noise"], int(g[tag + ":snr"]))
        assert gain == float(g[tag + ":gain"]), tag
        assert np.array_equal(out, g[tag + ":out"]), tag


def test_speed_oracle_matches_the_sequence_run_on_cpython_audioop():
    """pydub's speedup() sequence (slicing, fades, looped overlay, append) with the real audioop.mul / audioop.add doing the
    arithmetic (gen_golden.py::gen_audioop) against the same sequence on the restated primitives: factors below 1 (negative
    cross-fade) and above 1, a speech-like and a full-scale clip."""
    g = np.load(os.path.join(G, "audioop.npz"))
    n = 0
    while "speed%d:in" % n in g.files:
        x, fac = g["speed%d:in" % n], float(g["speed%d:factor" % n])
        got = SP.speedup(SP.Seg(x, 16000), fac).f
        assert np.array_equal(got, g["speed%d:out" % n]), (n, fac)
        n += 1
    assert n >= 6
