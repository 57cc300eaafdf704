"""Host-side logic that needs no GPU: EER routine vs the reference's known answers, file lists, eval padding,
WAV reader, flat-parameter layout and the reference's state-dict names, configs."""
import os
import wave

import numpy as np
import pytest
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


@pytest.mark.parametrize("nb,cmax", [(5, 100), (12, 100), (9, 260)])
def test_batched_notch_design_matches_scipy_per_filter_design(nb, cmax):
    """The throughput-mode sampler designs all notch filters of a batch at once (closed-form firwin, FFT product,
    FFT freqz); given the same (fc, bw, c, G) it must reproduce genNotchCoeffs (RawBoost.py:28-48) to round-off — also for
    non-default --nBands / --maxCoeff whose convolved length exceeds 1024 taps (the FFT size grows with it)."""
    from scipy import signal
    from scl_amd import augment
    rs = np.random.RandomState(0)
    n, fs = (12 if nb > 5 else 150), 16000          # 150: more than one design chunk
    fc = rs.uniform(20, 8000, (n, nb)); bw = rs.uniform(100, 1000, (n, nb))
    c = rs.uniform(10, cmax, (n, nb)).astype(np.int64); c = np.where(c % 2 == 0, c + 1, c)
    if nb > 5:
        c[0] = cmax + 1 - cmax % 2          # one filter at the largest length of the configuration
    G = rs.uniform(-20, 0, n)
    got = augment.design_notch_filters(fc, bw, c, G, fs)
    for i in range(n):
        b = 1
        for j in range(nb):
            f1, f2 = fc[i, j] - bw[i, j] / 2, fc[i, j] + bw[i, j] / 2
            f1 = 1 / 1000 if f1 <= 0 else f1
            f2 = fs / 2 - 1 / 1000 if f2 >= fs / 2 else f2
            b = np.convolve(signal.firwin(int(c[i, j]), [f1, f2], window="hamming", fs=fs), b)
        _, h = signal.freqz(b, 1, fs=fs)
        ref = 10 ** (G[i] / 20) * b / np.amax(np.abs(h))
        assert got[i].shape == ref.shape
        np.testing.assert_allclose(got[i], ref, rtol=0, atol=1e-12)
    # the FFT route (any tap counts; what the zero-phase route replaced for the odd counts genNotchCoeffs draws) agrees as well
    f1 = np.where(fc - bw / 2 <= 0, 1 / 1000, fc - bw / 2) / (fs / 2)
    f2 = np.where(fc + bw / 2 >= fs / 2, fs / 2 - 1 / 1000, fc + bw / 2) / (fs / 2)
    for g, o in zip(got, augment._design_notch_filters_general(f1, f2, c, G)):
        np.testing.assert_allclose(o, g, rtol=0, atol=1e-12)


def test_reference_order_sampler_equals_the_oracle_draw_for_draw():
    """sampler="reference" (what the RawBoost goldens and main.py's SCL_PACK_SAMPLER=reference run): the host draws of scl_amd/augment.py
    consume the global np.random stream exactly as the oracle's restatement of RawBoost.py:28-97 does — same seed, same parameters, the
    batched closed-form filter design within 1e-12 of the per-filter scipy chain — for the LnL, ISD and SSI stages in sequence."""
    from oracle import rawboost as RB
    from scl_amd import augment
    a = RB.RawBoostArgs()
    for seed in (3, 4):
        np.random.seed(seed)
        ref_taps = RB.lnl_draw(a.N_f, a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff, a.minG, a.maxG, a.minBiasLinNonLin,
                               a.maxBiasLinNonLin, 16000)
        ref_p, ref_f = RB.isd_draw(64000, a.P)
        ref_noise, ref_b, ref_snr = RB.ssi_draw(64000, a.SNRmin, a.SNRmax, a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff,
                                                a.minG, a.maxG, 16000)
        np.random.seed(seed)
        taps = augment._draw_lnl(a, 16000)
        p, f = augment._draw_isd(a, 64000)
        noise, b, snr = augment._draw_ssi(a, 64000, 16000)
        assert len(taps) == len(ref_taps) == a.N_f
        for t, r in zip(taps, ref_taps):
            assert t.shape == r.shape
            np.testing.assert_allclose(t, r, rtol=0, atol=1e-12)
        assert np.array_equal(p, np.asarray(ref_p).astype(np.int32)) and np.allclose(f, np.asarray(ref_f, dtype=np.float32), rtol=0, atol=0)
        assert np.array_equal(noise, np.asarray(ref_noise, dtype=np.float32)) and abs(snr - float(ref_snr)) == 0.0
        np.testing.assert_allclose(b, ref_b, rtol=0, atol=1e-12)


def test_fast_sampler_statistics():
    from scl_amd import augment
    from scl_amd.datautils_common import default_rawboost_args
    a = default_rawboost_args()
    augment.seed_fast_sampler(7)
    taps = augment._fast_lnl(a, 40, 16000)
    assert all(len(t) == 5 for t in taps)
    lens = np.array([len(b) for t in taps for b in t])
    assert lens.min() >= 5 * 11 - 4 and lens.max() <= 501 and 200 < lens.mean() < 340      # reference: mean 268.5
    isd = augment._fast_isd(a, 50, 64000)
    frac = np.array([len(p) / 64000 for p, _ in isd])
    assert 0 <= frac.min() and frac.max() <= 0.10 and 0.03 < frac.mean() < 0.07
    assert all(len(np.unique(p)) == len(p) for p, _ in isd)


def test_fairseq_checkpoint_reader_survives_missing_packages(tmp_path):
    """pretrained/xlsr2_300m.pt is a fairseq checkpoint: {'model': tensors, 'cfg': {'model': <omegaconf / dataclass object>}}.
    The reader must work without fairseq / omegaconf installed: classes that cannot be imported become inert stubs, tensors and
    the dropout / layerdrop probabilities are recovered (model/xlsr.py:14-16)."""
    import pickle
    import sys
    import types
    import torch
    from scl_amd import checkpoint
    mod = types.ModuleType("fairseq_like_pkg")

    class FakeCfg:
        def __init__(self, **kw):
            self.__dict__.update(kw)
    FakeCfg.__module__ = "fairseq_like_pkg"
    FakeCfg.__qualname__ = "FakeCfg"
    mod.FakeCfg = FakeCfg
    sys.modules["fairseq_like_pkg"] = mod
    sd = {"encoder.layers.0.fc1.weight": torch.arange(6.0).view(2, 3), "mask_emb": torch.ones(4)}
    ck = {"model": sd, "cfg": {"model": FakeCfg(dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0,
                                                 dropout_features=0.0, encoder_layerdrop=0.05, _name="wav2vec2"), "task": FakeCfg(x=1)}}
    path = tmp_path / "fake_fairseq.pt"
    torch.save(ck, str(path), pickle_module=pickle)
    del sys.modules["fairseq_like_pkg"]                      # the class is gone when the file is read back
    tensors, probs = checkpoint.read_fairseq_checkpoint(str(path))
    assert set(tensors) == set(sd) and torch.equal(tensors["encoder.layers.0.fc1.weight"], sd["encoder.layers.0.fc1.weight"])
    assert probs["encoder_layerdrop"] == 0.05 and probs["dropout"] == 0.0 and len(probs) == 6
    # old-style checkpoints carry an argparse.Namespace under 'args'
    import argparse
    torch.save({"model": sd, "args": argparse.Namespace(dropout=0.1, encoder_layerdrop=0.0)}, str(path))
    _, probs = checkpoint.read_fairseq_checkpoint(str(path))
    assert probs == {"dropout": 0.1, "encoder_layerdrop": 0.0}


def test_merge_rank_outputs_restores_protocol_order(tmp_path):
    import main as M
    path = str(tmp_path / "scores.txt")
    lines = ["utt%d 0.%d 1" % (i, i) for i in range(7)]
    for r in range(3):
        with open("%s.rank%d" % (path, r), "w") as fh:
            fh.write("\n".join(lines[r::3]) + "\n")
    M.merge_rank_outputs(path, 3, 7)
    assert open(path).read().splitlines() == lines and not os.path.exists(path + ".rank0")


def test_prefetcher_preserves_order_and_propagates_errors():
    from scl_amd.prefetch import Prefetcher
    items = [(i, torch.full((3,), float(i))) for i in range(7)]
    got = list(Prefetcher(items, depth=2))
    assert [g[0] for g in got] == list(range(7)) and all(torch.equal(a[1], b[1]) for a, b in zip(got, items))

    def bad():
        yield 1
        raise ValueError("decode failed")
    import pytest as _pt
    with _pt.raises(ValueError, match="decode failed"):
        list(Prefetcher(bad()))
    it = iter(Prefetcher(range(1000), depth=1))        # early exit must not hang the producer thread
    assert next(it) == 0
    it.close()


def test_parallel_prefetcher_builds_on_several_threads_and_hands_over_in_order():
    """Round 6: `workers` builder threads take the DataLoader's own index batches one sequence number each; whatever order they finish in,
    the consumer sees the batches in the sampler's order, each exactly once; an exception surfaces at ITS batch; early exit stops the
    builders; a loader that is not a plain in-process DataLoader falls back to the single-thread path."""
    import threading
    import time as _time
    import pytest as _pt
    from torch.utils.data import DataLoader, Dataset, Subset
    from scl_amd.prefetch import Prefetcher
    seen_threads = set()

    class DS(Dataset):
        def __init__(self, n, fail_at=None):
            self.n, self.fail_at = n, fail_at

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            seen_threads.add(threading.current_thread().name)
            _time.sleep(0.002 * ((i * 7) % 5))          # uneven build times: later batches finish first
            if i == self.fail_at:
                raise ValueError("decode failed at %d" % i)
            return "u%d" % i, torch.full((4, 2), float(i)), torch.tensor([float(i)])

    ld = DataLoader(DS(41), batch_size=3, shuffle=False, drop_last=False, num_workers=0)
    got = list(Prefetcher(ld, depth=2, workers=4))
    ref = list(ld)
    assert len(got) == len(ref) == 14
    for a, b in zip(got, ref):
        assert list(a[0]) == list(b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert len([t for t in seen_threads if t.startswith("scl-prefetch-")]) >= 2
    # shuffle + drop_last + Subset: the sampler's own semantics — every index of the epoch exactly once
    ld2 = DataLoader(Subset(DS(50), list(range(5, 45))), batch_size=4, shuffle=True, drop_last=True, num_workers=0)
    ids = sorted(int(u[1:]) for b in Prefetcher(ld2, workers=3) for u in b[0])
    assert ids == list(range(5, 45))
    # an exception surfaces at its batch, after the batches before it
    it = iter(Prefetcher(DataLoader(DS(30, fail_at=13), batch_size=2, num_workers=0), workers=4))
    heads = []
    with _pt.raises(ValueError, match="decode failed at 13"):
        for b in it:
            heads.append(b[0][0])
    assert heads == ["u%d" % i for i in range(0, 12, 2)]
    # early exit does not hang; one worker / a non-DataLoader iterable use the serial path
    it = iter(Prefetcher(DataLoader(DS(1000), batch_size=1, num_workers=0), workers=4))
    assert next(it)[0] == ["u0"] or list(next(it)[0]) == ["u1"]
    it.close()
    assert [b[0][0] for b in Prefetcher(DataLoader(DS(5), batch_size=1, num_workers=0), workers=1)] == ["u%d" % i for i in range(5)]
    assert list(Prefetcher(range(5), workers=4)) == list(range(5))


def test_prefetcher_hints_the_dataset_which_items_come_next():
    """Round 6: a dataset with decode_ahead(indices) (scl_amd.pack.PackDataset: decode the next packs' files on decoder threads) is told the
    index batches ahead of their build — every item is hinted before it is built, through a Subset with the base dataset's indices."""
    from torch.utils.data import DataLoader, Dataset, Subset
    from scl_amd.prefetch import Prefetcher
    log = []

    class DS(Dataset):
        def __len__(self):
            return 40

        def decode_ahead(self, idxs):
            log.append(("hint", tuple(idxs)))

        def __getitem__(self, i):
            log.append(("build", i))
            return torch.tensor([float(i)])

    sub = Subset(DS(), list(range(39, 9, -1)))          # 30 items, base indices 39 .. 10
    got = [int(b[0, 0]) for b in Prefetcher(DataLoader(sub, batch_size=1, shuffle=False, num_workers=0), workers=1)]
    assert got == list(range(39, 9, -1))
    hinted = set()
    for kind, v in log:
        if kind == "hint":
            hinted.update(v)
        else:
            assert v in hinted, (v, sorted(hinted))     # hinted (with the BASE index) before it was built
    assert hinted == set(range(10, 40))
    first_build = next(i for i, (k, _) in enumerate(log) if k == "build")
    assert sum(1 for k, _ in log[:first_build] if k == "hint") >= 2          # at least two batches ahead when the first build starts


def test_threaded_scoring_loop_writes_the_same_lines_in_protocol_order(tmp_path):
    """main._score_loop (round 6): decoder threads + results read back one batch late must produce exactly the score file of the plain
    DataLoader loop — both through per-item __getitem__ futures and through EvalDataset-style load_into rows (a ragged last batch included)."""
    import time as _time
    import main as M
    from torch.utils.data import Dataset

    class Items(Dataset):
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            _time.sleep(0.001 * ((i * 5) % 4))
            return torch.full((16,), float(i)), "utt%04d" % i

    class Rows(Items):
        cut = 16

        def load_into(self, i, out):
            _time.sleep(0.001 * ((i * 3) % 4))
            out[:] = float(i)
            return "utt%04d" % i

    class Net(torch.nn.Module):
        def forward(self, x):
            s = x.mean(dim=1, keepdim=True)
            return torch.cat([s, -2.0 * s], dim=1), x, s          # (output, feats, emb) as the plugins' training-mode forward (is_train)

    net = Net()
    outs = {}
    for name, ds, thr in (("plain", Items(37), "1"), ("futures", Items(37), "4"), ("rows", Rows(37), "4")):
        os.environ["SCL_EVAL_THREADS"] = thr
        path = str(tmp_path / (name + ".txt"))
        M.produce_evaluation_file(ds, net, torch.device("cpu"), path, batch_size=5)
        outs[name] = open(path).read()
    os.environ.pop("SCL_EVAL_THREADS")
    lines = outs["plain"].splitlines()
    assert len(lines) == 37 and lines[0].split()[0] == "utt0000" and lines[-1] == "utt0036 36.0 -72.0"
    assert outs["futures"] == outs["plain"] and outs["rows"] == outs["plain"]


def test_non_wav_corpus_is_refused_at_start_up_without_a_decoder():
    from scl_amd import pack
    pack.set_audio_loader(None)
    pack.require_decoder_for(["a.wav", "b.WAV", "LA_T_1000137.flac"], "list")     # WAV: stdlib; FLAC: the library's decoder
    try:
        import soundfile  # noqa: F401
        has_sf = True
    except ImportError:
        has_sf = False
    if not has_sf:
        import pytest as _pt
        with _pt.raises(RuntimeError, match="soundfile"):
            pack.require_decoder_for(["music-fma-0001.mp3"], "noise corpus")
    pack.set_audio_loader(lambda p, sr: np.zeros(10, np.float32))
    pack.require_decoder_for(["music-fma-0001.mp3"], "list")               # a registered loader decodes anything
    pack.set_audio_loader(None)


@pytest.mark.parametrize("per_layer", [[64, 64, 16, 48], [16, 16, 4, 12], [100, 100, 30, 90], [300, 7]])
@pytest.mark.parametrize("carry", [True, False])
def test_carried_over_group_launches_cover_every_tile_once_and_never_hold_work_for_two_layers(per_layer, carry):
    """Host logic of the grouped weight-gradient launches (scl_amd/encoder.py::plan_group_launches): a backward over 24 layers, each
    contributing `per_layer` tile counts.  Every tile of every problem is launched exactly once, in order; no problem is still pending
    when the layer AFTER the next one starts (its operands live one extra layer); with carrying on, every launch but the last of a
    flush is a whole round of 256 tiles — 18 launches instead of 24 for the encoder's 64 + 64 + 16 + 48."""
    from scl_amd.encoder import plan_group_launches
    pend, seen, launches = [], {}, []
    for layer in range(24):
        for it in pend:
            it[3] += 1
        pend.extend(["L%d.%d" % (layer, j), c, 0, 0] for j, c in enumerate(per_layer))
        out = plan_group_launches(pend, final=(layer == 23), carry=carry)
        assert all(it[3] == 0 for it in pend), "work older than the current layer was held back"
        for parts in out:
            assert 1 <= len(parts) <= 8
            for name, first, count in parts:
                assert count >= 1 and seen.get(name, 0) == first, (name, first, seen.get(name))
                seen[name] = first + count
        launches += out
    assert not pend
    for layer in range(24):
        for j, c in enumerate(per_layer):
            assert seen["L%d.%d" % (layer, j)] == c
    sizes = [sum(c for _, _, c in parts) for parts in launches]
    if carry:
        assert all(s <= 256 for s in sizes)
        if per_layer == [64, 64, 16, 48]:
            assert sizes == [256] * 18
    elif sum(per_layer) <= 256 and len(per_layer) <= 8:
        assert sizes == [sum(per_layer)] * 24
