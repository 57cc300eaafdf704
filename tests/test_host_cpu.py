"""Host-side logic that needs no GPU: EER routine vs the reference's known answers, file lists, eval padding,
WAV reader, flat-parameter layout and the reference's state-dict names, configs."""
import os
import wave

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_eer_known_answers_product_routine():
    from evaluate_metrics import compute_eer
    g = np.load(os.path.join(G, "eer.npz"))
    for tag in ("la19_conf3", "itw_conf3"):
        s, l = g[tag + "_scores"], g[tag + "_labels"]
        eer, thr = compute_eer(s[l == 1], s[l == 0])
        assert abs(eer - g[tag + "_eer_thr"][0]) < 1e-12 and abs(thr - g[tag + "_eer_thr"][1]) < 1e-12


def test_eval_pad_and_wav_reader_and_lists(tmp_path):
    from scl_amd import pack
    g = np.load(os.path.join(G, "rawboost.npz"))
    assert np.array_equal(pack.pad_eval(g["pad_in"], "repeat", 2000), g["pad_repeat"].astype(np.float32))
    assert np.array_equal(pack.pad_eval(g["pad_in"], "zero", 2000), g["pad_zero"].astype(np.float32))
    x = (0.3 * np.sin(np.arange(800) / 7.0)).astype(np.float32)
    p = str(tmp_path / "a.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((x * 32768).astype("<i2").tobytes())
    y = pack.load_audio(p, 16000)
    assert y.dtype == np.float32 and np.abs(y - x).max() < 1e-4
    y8 = pack.load_audio(p, 8000)
    assert abs(len(y8) - 400) <= 1
    os.makedirs(tmp_path / "scp")
    (tmp_path / "scp" / "train_bonafide.lst").write_text("a.wav\nb.wav\n")
    _, files = pack.gen_list_scp(str(tmp_path / "protocol.txt"), is_train=True)
    assert files == ["a.wav", "b.wav"]


def test_flat_params_layout_and_reference_names():
    from scl_amd.encoder import W2VConfig, param_specs
    from scl_amd.model_linear import head_specs
    from scl_amd.params import FlatParams
    from oracle import wav2vec2 as W
    cfg = W2VConfig.tiny()
    P = FlatParams(param_specs(cfg) + head_specs(cfg.embed), "cpu")
    ours = {n[len("ssl_model.model."):]: tuple(s) for n, (_, _, s, _) in P.index.items() if n.startswith("ssl_model.model.")}
    ref = {n: tuple(s) for n, s, _ in W.param_shapes(W.W2VConfig.tiny())}
    assert ours == ref
    # q, k, v weights (and biases) of a layer are adjacent: one [3E, E] GEMM operand
    E = cfg.embed
    oq, ok, ov = (P.off("ssl_model.model.encoder.layers.0.self_attn.%s_proj.weight" % p) for p in "qkv")
    assert ok - oq == E * E and ov - ok == E * E
    assert all(P.off(n) % 8 == 0 for n in P.index)
    assert all(P.index[n][0] < P.n_train for n in P.index if P.index[n][3]) and P.n_total > P.n_train
    full = FlatParams.__new__(FlatParams)  # count only
    n_train = sum(int(np.prod(s)) for _, s, t in param_specs(W2VConfig()) if t)
    assert abs(n_train - 315.4e6) < 0.3e6
    # gradient views alias the flat gradient buffer
    P.grad.fill_(3.0)
    assert float(P.params["LL.bias"].grad.sum()) == 3.0 * 128


def test_configs_keep_the_reference_keys():
    for fn, plugin in (("conf-1-linear.yaml", "SCL_normal"), ("conf-2-linear.yaml", "asvspoof_2019_aug_2"),
                       ("conf-3-linear.yaml", "asvspoof_2019_augall_3"), ("conf-5-linear.yaml", "asvspoof_2019_augall_5")):
        c = yaml.safe_load(open(os.path.join(ROOT, "configs", fn)))
        assert c["model"]["name"] == "wav2vec2_linear_nll" and c["model"]["contra_mode"] == "all" and c["model"]["loss_type"] == 1
        assert c["data"]["name"] == plugin and c["data"]["kwargs"]["trim_length"] == 64000
        assert os.path.exists(os.path.join(ROOT, "datautils", plugin + ".py"))
