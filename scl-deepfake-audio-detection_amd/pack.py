"""Anchor-pack builder: host-side mirror of the reference's `Dataset_for.__getitem__`
(datautils/asvspoof_2019_augall_3.py:103-146 and its siblings aug_2 / SCL_normal / augall_5), with every
waveform transformation executed on the GPU (scl_amd.augment) instead of in DataLoader workers.

Contract kept (SURVEY.md §8b "Data plugin"):
  Dataset_for(args, list_IDs=, labels=, base_dir=, algo=, repeat_pad=, **yaml_kwargs)[i]
      -> (id: str, Tensor[L, V] fp32, Tensor[V] fp32 labels)         view order and labels as the reference
  Dataset_for_eval(list_IDs=, base_dir=, padding_type=)[i] -> (Tensor[64600], id)
  augmenters are looked up BY NAME with signature (x, args, sr, audio_path=...) -> waveform
The returned pack lives on the GPU as a transposed view of a contiguous [V, L] buffer, so that
main.py's `batch_x.squeeze(0).transpose(0, 1)` yields the contiguous [V, L] the model wants.

Randomness: `np.random` / `random` are consumed in the reference's order (one pack after the other),
so a seeded run draws the same files, crops and RawBoost parameters as the reference would.
File decoding is host IO (out of the kernel path): WAV through the stdlib, anything else through
`soundfile` when it is installed; `set_audio_loader()` lets callers (and the tests) supply arrays.
"""
import os
import threading
import random
import wave

import numpy as np
import torch
from torch.utils.data import Dataset

from . import augment

_LOADER = None
_FILE_INDEX = {}


def set_audio_loader(fn):
    """fn(path, sr) -> 1-D float32 numpy array at `sr` Hz (mono); None restores the default reader."""
    global _LOADER
    _LOADER = fn


def _read_wav(path):
    with wave.open(path, "rb") as w:
        n, ch, sw, sr = w.getnframes(), w.getnchannels(), w.getsampwidth(), w.getframerate()
        raw = w.readframes(n)
    if sw == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif sw == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError("unsupported WAV sample width %d in %s" % (sw, path))
    if ch > 1:
        x = x.reshape(-1, ch).mean(axis=1)
    return x, sr


_FLAC_MAX_SAMPLES = 1 << 26      # 70 minutes at 16 kHz: the bound for streams whose STREAMINFO carries no length


def _read_flac(path):
    """FLAC through the library's own decoder (csrc/flac.hip: frame CRCs and the STREAMINFO MD5 are verified); channels are averaged
    as librosa.load(mono=True) does."""
    import ctypes
    from . import lib as L
    with open(path, "rb") as f:
        raw = f.read()
    lib = L.load()
    buf = ctypes.create_string_buffer(raw, len(raw))
    fs, ch, bits, total = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int64(0)
    L.check(lib.scl_flac_info(buf, len(raw), ctypes.byref(fs), ctypes.byref(ch), ctypes.byref(bits), ctypes.byref(total)), "scl_flac_info(%s)" % path)
    # STREAMINFO without a length (streamed encoders): a CONSTANT / silence frame expands without bound per byte, so the output
    # grows until the decoder stops reporting "too small" (it writes nothing past the capacity it is given)
    got = ctypes.c_int64(0)
    if total.value > 0:      # the usual case: one pass straight to mono float32 (scl_flac_decode_mono_f32 = the arithmetic of the numpy lines below)
        x = np.empty(total.value, dtype=np.float32)
        L.check(lib.scl_flac_decode_mono_f32(buf, len(raw), x.ctypes.data_as(ctypes.c_void_p), total.value, ctypes.byref(got), 1), "scl_flac_decode_mono_f32(%s)" % path)
        return x[: got.value], fs.value
    cap = max(len(raw) * 16, 1 << 16)
    while True:
        out = np.empty((cap, ch.value), dtype=np.int32)
        rc = lib.scl_flac_decode_i32(buf, len(raw), out.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(got), 1)
        if rc != 0 and total.value <= 0 and b"output too small" in (lib.scl_last_error() or b""):
            if cap >= _FLAC_MAX_SAMPLES:      # a stream without a declared length that is still growing: refuse instead of exhausting the worker's memory
                raise ValueError("%s: FLAC stream without a STREAMINFO length exceeds %d samples per channel" % (path, _FLAC_MAX_SAMPLES))
            cap = min(cap * 8, _FLAC_MAX_SAMPLES)
            continue
        L.check(rc, "scl_flac_decode_i32(%s)" % path)
        break
    x = out[: got.value].astype(np.float32) / np.float32(1 << (bits.value - 1))
    return (x.mean(axis=1) if ch.value > 1 else x[:, 0]), fs.value


_DECODE_POOL = None


def _decode_pool():
    """SCL_DECODE_THREADS (default 3; 0 = off) threads that decode files ahead of the pack builder (PackDataset.decode_ahead)."""
    global _DECODE_POOL
    if _DECODE_POOL is None:
        n = int(os.environ.get("SCL_DECODE_THREADS", "3"))
        if n <= 0:
            _DECODE_POOL = False
        else:
            from concurrent.futures import ThreadPoolExecutor
            _DECODE_POOL = ThreadPoolExecutor(max_workers=n, thread_name_prefix="scl-decode")
    return _DECODE_POOL or None


def load_audio(path, sr=16000):
    """librosa.load(path, sr=sr, mono=True) stand-in for the formats this image can decode."""
    if _LOADER is not None:
        return np.asarray(_LOADER(path, sr), dtype=np.float32)
    with open(path, "rb") as f:          # by content, not by name: the offline augmentation cache keeps the source file's name
        magic = f.read(4)                # (augall_3:285-291 writes out_format='wav' bytes under it)
    if magic == b"RIFF":
        x, fs = _read_wav(path)
    elif magic == b"fLaC":
        x, fs = _read_flac(path)
    else:
        try:
            import soundfile as sf
        except ImportError as e:  # pragma: no cover
            raise RuntimeError("cannot decode %s: install `soundfile` or call scl_amd.pack.set_audio_loader()" % path) from e
        x, fs = sf.read(path, dtype="float32", always_2d=False)
        if x.ndim > 1:
            x = x.mean(axis=1)
    if fs != sr:
        from scipy.signal import resample_poly
        g = np.gcd(int(fs), int(sr))
        x = resample_poly(x, sr // g, fs // g).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


def require_decoder_for(paths, what):
    """Fail at start-up when the corpus holds formats this image cannot decode: WAV goes through the stdlib, FLAC (ASVspoof) through
    the library's decoder, everything else (MP3 in some MUSAN mirrors) needs `soundfile` — or a loader registered with
    set_audio_loader()."""
    if _LOADER is not None:
        return
    other = [p for p in paths if not str(p).lower().endswith((".wav", ".flac"))]
    if not other:
        return
    try:
        import soundfile  # noqa: F401
    except ImportError as e:
        raise RuntimeError("%s: %d files such as %s are neither WAV nor FLAC and the `soundfile` package is not installed — install it or register "
                           "a decoder with scl_amd.pack.set_audio_loader(fn)" % (what, len(other), other[0])) from e


def list_audio_files(root):
    """Index of a noise / RIR corpus, built ONCE per path (the reference os.walk()s it for every sample,
    audio_augmentor/background_noise.py:22, reverb.py:30)."""
    if root not in _FILE_INDEX:
        files = []
        for dp, _, fns in os.walk(root):
            for fn in fns:
                if fn.split(".")[-1] in ("wav", "mp3", "flac"):
                    files.append(os.path.join(dp, fn))
        _FILE_INDEX[root] = files
    return _FILE_INDEX[root]


def _dev(args):
    return torch.device(getattr(args, "device", "cuda"))


# Decoded noise / RIR files stay resident in HBM (the reference decodes a possibly minutes-long MUSAN file from disk for every
# view, background_noise.py:40-42): an LRU of device tensors bounded by SCL_AUDIO_BANK_GB (default 8 of the 288 GB).
_BANK, _BANK_BYTES = {}, 0
_BANK_LIMIT = int(float(os.environ.get("SCL_AUDIO_BANK_GB", "8")) * (1 << 30))
_BANK_SUMSQ = {}      # bank key -> sum of squares of the int16 image (host integer)
_BANK_LOCK = threading.Lock()      # several pack-builder threads (scl_amd/prefetch.py), each on its own HIP stream


def bank_tensor(path, sr, args, kind):
    """kind 'i16': int16 PCM as pydub's AudioSegment.from_file would hold it; 'f32': float waveform (RIRs).
    A tensor enters the bank only when the stream that made it has finished (once per file: the builder threads that find it there
    later run on other streams), and every user marks it on its own stream, so an LRU eviction cannot hand its memory out under a kernel
    that is still reading it."""
    global _BANK_BYTES
    key = (path, sr, kind, str(_dev(args)))
    with _BANK_LOCK:
        t = _BANK.pop(key, None)
        if t is not None:
            _BANK[key] = t        # most recently used last
    if t is None:
        xh = load_audio(path, sr)
        x = _to_dev(xh, args)                             # decode + upload outside the lock: other builders keep going
        t = augment.to_int16(x) if kind == "i16" else x
        if kind == "i16":
            _BANK_SUMSQ[key] = augment.host_i16_sumsq(xh)      # the noise file's integer power, once per file (background_noise needs it per view)
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
        with _BANK_LOCK:
            if key in _BANK:                              # another builder decoded the same file meanwhile: keep one copy
                t = _BANK.pop(key)
            else:
                _BANK_BYTES += t.numel() * t.element_size()
            _BANK[key] = t
            while _BANK_BYTES > _BANK_LIMIT and len(_BANK) > 1:
                old = _BANK.pop(next(iter(_BANK)))
                _BANK_BYTES -= old.numel() * old.element_size()
    if t.is_cuda:
        t.record_stream(torch.cuda.current_stream(t.device))
    return t


def _to_dev(x, args):
    """Host samples -> device.  Through pinned staging + an ASYNCHRONOUS copy (round 6): a plain `.to(device)` of pageable memory is a
    blocking copy that first waits for everything queued on the builder's stream — eleven times per conf-3 pack, each time behind the
    whole augmentation chain of the previous view, on a GPU the training step keeps saturated: the builder then runs at the pace of its
    kernels' queueing delays (the bimodal 1.05 x / 1.9 x steps of profiles/r6_pack_builder.txt).  The caching host allocator hands a
    pinned block out again only after the copy that reads it has completed."""
    dev = _dev(args)
    if torch.is_tensor(x):
        return x.to(dev, dtype=torch.float32)
    a = np.ascontiguousarray(x, dtype=np.float32)
    if dev.type != "cuda":
        return torch.from_numpy(a).to(dev)
    host = torch.empty(a.shape, dtype=torch.float32, pin_memory=True)
    host.numpy()[...] = a
    return augment.upload_async(host, dev)      # plain non_blocking copy in a builder thread, upload stream when the trainer's thread builds


# ---- augmenters, resolved by name (augall_3:271-374) ---------------------------------------------
def RawBoost12(x, args, sr=16000, audio_path=None):
    """RawBoost algo 5 (LnL then ISD), augall_3:359-374."""
    if not getattr(args, "online_aug", True):
        return _offline_cached("RawBoost12", x, args, sr, audio_path, lambda: RawBoost12_online(x, args, sr))
    return RawBoost12_online(x, args, sr)


def RawBoost12_online(x, args, sr=16000):
    """args.rawboost_sampler = "fast" (main.py's training default, SCL_PACK_SAMPLER): the draws come from the builder thread's own numpy
    Generator (no 64000-element permutation for the ISD positions, batched closed-form filter design) instead of the reference's
    draw-for-draw order on the global np.random stream ("reference": the default here, what the pack goldens pin)."""
    xd = _to_dev(x, args)
    return augment.rawboost_batch(xd[None], args, 5, sr, sampler=getattr(args, "rawboost_sampler", "reference"))[0]


def background_noise_wrapper(x, args, sr=16000, audio_path=None):
    """MUSAN overlay, augall_3:271-283 -> audio_augmentor/background_noise.py:40-56 (returns int16-scaled values)."""
    if not getattr(args, "online_aug", True) and getattr(args, "aug_dir", None) and audio_path:
        return _offline_cached("background_noise", x, args, sr, audio_path, lambda: _background_noise_online(x, args, sr), int16_values=True)
    return _background_noise_online(x, args, sr)


def _background_noise_online(x, args, sr):
    noise_list = list_audio_files(args.noise_path)
    noise_file = random.choice(noise_list)
    snr_db = random.randint(5, 15)
    noise_i16 = bank_tensor(noise_file, sr, args, "i16")   # AudioSegment.from_file decodes to int16 PCM; resident in HBM after first use
    nsq = _BANK_SUMSQ.get((noise_file, sr, "i16", str(_dev(args))))
    if nsq is not None and not torch.is_tensor(x):
        # both integer powers on the host (the speech clip is still a host array here): no host <- device round trip in the builder
        return augment.background_noise(_to_dev(x, args), noise_i16, snr_db, sumsq=(augment.host_i16_sumsq(x), nsq))
    return augment.background_noise(_to_dev(x, args), noise_i16, snr_db)


def reverb_wrapper(x, args, sr=16000, audio_path=None):
    """RIR convolution, augall_3:314-326 -> audio_augmentor/reverb.py:33-44 (returns int16-scaled values, length L+R-1)."""
    if not getattr(args, "online_aug", True) and getattr(args, "aug_dir", None) and audio_path:
        return _offline_cached("reverb", x, args, sr, audio_path, lambda: _reverb_online(x, args, sr), int16_values=True)
    return _reverb_online(x, args, sr)


def _reverb_online(x, args, sr):
    rir_file = random.choice(list_audio_files(args.rir_path))
    return augment.reverb(_to_dev(x, args), bank_tensor(rir_file, sr, args, "f32"))


UNSUPPORTED_AUGMENTERS = {}      # name -> what it needs; every augmenter the reference's configs name runs (checked at start-up)


def speed_wrapper(x, args, sr=16000, audio_path=None):
    """pydub speed change, augall_5:342-361 -> audio_augmentor/speed.py:9-33: factor ~ random.uniform(0.9, 1.1) (augall_5:246-247)."""
    if not getattr(args, "online_aug", True) and getattr(args, "aug_dir", None) and audio_path:
        return _offline_cached("speed", x, args, sr, audio_path, lambda: _speed_online(x, args, sr), int16_values=True)
    return _speed_online(x, args, sr)


def _speed_online(x, args, sr):
    return augment.speed(_to_dev(x, args), random.uniform(0.9, 1.1), sr)


def pitch_wrapper(x, args, sr=16000, audio_path=None):
    """Phase-vocoder pitch shift, augall_5:300-319 -> audio_augmentor/pitch.py:11-38: n_steps = random.randint(-1, 1) (augall_5:207-208)."""
    if not getattr(args, "online_aug", True) and getattr(args, "aug_dir", None) and audio_path:
        return _offline_cached("pitch", x, args, sr, audio_path, lambda: _pitch_online(x, args, sr), int16_values=True)
    return _pitch_online(x, args, sr)


def _pitch_online(x, args, sr):
    return augment.pitch_shift(_to_dev(x, args), random.randint(-1, 1), sr)


def _offline_cached(method, x, args, sr, audio_path, make, int16_values=False):
    """online_aug: false — reuse / create <aug_dir>/<method>/<utt> (PCM16), augall_3:285-291,366-374.  The noise / reverb augmenters
    hand back int16-VALUED samples (pydub_to_librosa keeps them unscaled) and the reference exports the AudioSegment itself, then
    re-loads it with librosa.load, i.e. scaled to [-1, 1): the cached file holds the int16 values, a cache hit returns them / 32768."""
    aug_path = os.path.join(args.aug_dir, method, os.path.basename(audio_path))
    if os.path.exists(aug_path):
        return _to_dev(load_audio(aug_path, sr), args)
    y = make()
    os.makedirs(os.path.dirname(aug_path), exist_ok=True)
    if int16_values:
        pcm = np.clip(y.detach().cpu().numpy(), -32768, 32767).astype("<i2")
        y = y / 32768.0                   # the reference re-loads the file it just wrote with librosa.load (augall_3:288-291)
    else:
        pcm = np.clip(np.round(y.detach().cpu().numpy() * 32767.0), -32768, 32767).astype("<i2")
    # PCM16 WAV bytes whatever the extension, as the reference's export(out_format='wav') does; load_audio() sniffs RIFF / fLaC
    with wave.open(aug_path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes(pcm.tobytes())
    return y


AUGMENTERS = {"RawBoost12": RawBoost12, "background_noise_wrapper": background_noise_wrapper, "reverb_wrapper": reverb_wrapper,
              "speed_wrapper": speed_wrapper, "pitch_wrapper": pitch_wrapper}


# ---- file lists (augall_3:14-47) -------------------------------------------------------------------
def gen_list_scp(dir_meta, is_train=False, is_eval=False, is_dev=False):
    base = os.path.dirname(dir_meta)
    if is_train:
        meta = os.path.join(base, "scp/train_bonafide.lst")
    elif is_dev:
        meta = os.path.join(base, "scp/dev_bonafide.lst")
    elif is_eval:
        meta = os.path.join(base, "scp/test.lst")
    else:
        raise ValueError("one of is_train / is_dev / is_eval must be set")
    with open(meta) as f:
        files = [ln.strip().split()[0] for ln in f if ln.strip()]
    return [], files


def pad_eval(x, padding_type, max_len=64600):
    """augall_3:49-60"""
    n = x.shape[0]
    if n >= max_len:
        return x[:max_len]
    if padding_type == "repeat":
        return np.tile(x, int(max_len / n) + 1)[:max_len]
    out = np.zeros(max_len, dtype=x.dtype)
    out[:n] = x
    return out


class PackDataset(Dataset):
    """recipe: 'augall_3' | 'aug_2' | 'scl_normal' | 'augall_5' — the view composition of the four reference plugins."""

    def __init__(self, recipe, args, list_IDs, labels, base_dir, algo=5, vocoders=(), augmentation_methods=(), num_additional_real=2,
                 num_additional_spoof=2, trim_length=64000, wav_samp_rate=16000, noise_path=None, rir_path=None, aug_dir=None,
                 online_aug=False, repeat_pad=True, is_train=True):
        self.recipe, self.args = recipe, args
        args.noise_path, args.rir_path, args.aug_dir, args.online_aug = noise_path, rir_path, aug_dir, online_aug
        self.list_IDs = list_IDs
        require_decoder_for(list_IDs, "training / validation list")
        self.bonafide_dir = os.path.join(base_dir, "bonafide")
        self.vocoded_dir = os.path.join(base_dir, "vocoded")
        self.trim_length, self.sample_rate, self.repeat_pad = trim_length, wav_samp_rate, repeat_pad
        self.vocoders = list(vocoders)
        self.num_additional_real, self.num_additional_spoof = num_additional_real, num_additional_spoof
        self.methods = list(augmentation_methods) if len(augmentation_methods) >= 1 else ["RawBoost12"]
        # fail at start-up, not on the first __getitem__ after the model has been built and the corpus listed
        unknown = [m for m in self.methods if m not in AUGMENTERS]
        missing = [m for m in self.methods if m in UNSUPPORTED_AUGMENTERS]
        if unknown or missing:
            raise NotImplementedError("augmentation_methods %s cannot run: %s" % (
                unknown + missing, "; ".join(["%s is not a known augmenter" % m for m in unknown] +
                                             ["%s needs %s" % (m, UNSUPPORTED_AUGMENTERS[m]) for m in missing])))
        if recipe in ("scl_normal", "augall_5"):
            # SCL_normal.py:69-71 switches on args.is_train; asvspoof_2019_augall_5.py:82 always reads <base>/spoof
            self.spoof_dir = os.path.join(base_dir, "spoof") if recipe == "augall_5" else \
                os.path.join(base_dir, "spoof_train" if getattr(args, "is_train", is_train) else "spoof_dev")
            self.spoof_list = [f for f in sorted(os.listdir(self.spoof_dir)) if f.endswith(".wav") or f.endswith(".flac")] \
                if os.path.isdir(self.spoof_dir) else []
        self._ahead = {}      # path -> Future of load_audio (decode_ahead)
        print("vocoders:", self.vocoders)

    def __len__(self):
        return len(self.list_IDs)

    def _aug(self, name, x, path):
        return AUGMENTERS[name](x, self.args, self.sample_rate, audio_path=path)

    def _load(self, path):
        fut = self._ahead.pop(path, None)          # decoded ahead by decode_ahead()?
        if fut is not None:
            return fut.result()
        return load_audio(path, self.sample_rate)

    def decode_ahead(self, indices):
        """Hint from the prefetcher: these items come next.  The files an item is KNOWN to read before any random draw — its bona fide
        utterance and that utterance's vocoded versions, 4 of the 5 files of a conf-3 pack — are decoded now on a small pool of threads (the
        FLAC decoder is C and releases the interpreter lock), so that the builder thread finds them ready instead of spending 0.5 - 0.9 ms
        per file itself: 2.5 - 3 of the 6.3 ms a pack costs it.  No RNG is touched; a file that is not found here is decoded inline."""
        pool = _decode_pool()
        if pool is None:
            return
        for idx in indices:
            uid = self.list_IDs[idx]
            paths = [os.path.join(self.bonafide_dir, uid)]
            if self.recipe in ("augall_3", "aug_2", "augall_5"):
                paths += [os.path.join(self.vocoded_dir, v + "_" + uid) for v in self.vocoders]
            for p in paths:
                if p not in self._ahead and len(self._ahead) < 512:
                    self._ahead[p] = pool.submit(load_audio, p, self.sample_rate)

    def __getitem__(self, idx):
        uid = self.list_IDs[idx]
        real_path = os.path.join(self.bonafide_dir, uid)
        real = self._load(real_path)
        m, r = self.methods, self.recipe
        voc, aug_voc = [], []
        if r in ("augall_3", "aug_2", "augall_5"):
            for vf in (os.path.join(self.vocoded_dir, v + "_" + uid) for v in self.vocoders):
                va = self._load(vf)
                voc.append(va)
                name = m[random.choice(range(len(m)))] if r == "aug_2" else m[0]
                aug_voc.append(self._aug(name, va, vf))
        aug_real = [self._aug(name, real, real_path) for name in m]
        idxs = list(range(len(self.list_IDs)))
        idxs.remove(idx)
        add_idx = np.random.choice(idxs, self.num_additional_real, replace=False)
        add_paths = [os.path.join(self.bonafide_dir, self.list_IDs[i]) for i in add_idx]
        add_real = [self._load(p) for p in add_paths]
        aug_add, add_spoof, aug_spoof = [], [], []
        if r in ("aug_2", "scl_normal"):
            for a, p in zip(add_real, add_paths):
                aug_add.append(self._aug(m[random.choice(range(len(m)))], a, p))
        if r in ("scl_normal", "augall_5"):
            sp_names = np.random.choice(self.spoof_list, self.num_additional_spoof, replace=False)
            sp_paths = [os.path.join(self.spoof_dir, s) for s in sp_names]
            add_spoof = [self._load(p) for p in sp_paths]
            if r == "scl_normal":
                for a, p in zip(add_spoof, sp_paths):
                    aug_spoof.append(self._aug(m[random.choice(range(len(m)))], a, p))
        if r == "augall_3":
            views, n_pos = [real] + aug_real + add_real + voc + aug_voc, 1 + len(aug_real) + len(add_real)
        elif r == "aug_2":
            views, n_pos = [real] + aug_real + add_real + aug_add + voc + aug_voc, 1 + len(aug_real) + len(add_real) + len(aug_add)
        elif r == "scl_normal":
            views, n_pos = [real] + aug_real + add_real + aug_add + add_spoof + aug_spoof, 1 + len(aug_real) + len(add_real) + len(aug_add)
        else:
            views, n_pos = [real] + aug_real + add_real + voc + aug_voc + add_spoof, 1 + len(aug_real) + len(add_real)
        dev_views = [_to_dev(v, self.args).reshape(-1) for v in views]
        pack = augment.multiview_crop(dev_views, self.trim_length, self.repeat_pad, random_trim=True)   # [V, L]
        lab = np.asarray([1.0] * n_pos + [0.0] * (len(views) - n_pos), dtype=np.float32)
        if pack.is_cuda:
            # the labels travel with the pack (pinned + asynchronous, on the builder's stream): as a host tensor they cost the TRAINING loop a
            # blocking `.to(device)` per step (main.run_epoch / the reference's train_epoch: batch_y.to(device)), i.e. one full
            # synchronisation of the launch thread with the GPU per optimizer step
            label = augment._h2d_pack([lab], pack.device)[0]
        else:
            label = torch.from_numpy(lab)
        return uid, pack.t(), label


class EvalDataset(Dataset):
    def __init__(self, list_IDs, base_dir, padding_type="zero", subdir="eval"):
        self.list_IDs = list_IDs
        require_decoder_for(list_IDs, "evaluation list")
        self.base_dir = os.path.join(base_dir, subdir) if subdir else base_dir
        self.cut = 64600
        self.padding_type = padding_type

    def __len__(self):
        return len(self.list_IDs)

    def __getitem__(self, index):
        utt_id = self.list_IDs[index]
        x = load_audio(os.path.join(self.base_dir, utt_id), 16000)
        return torch.from_numpy(np.ascontiguousarray(pad_eval(x, self.padding_type, self.cut), dtype=np.float32)), utt_id

    def load_into(self, index, out):
        """Decode + pad utterance `index` straight into `out` (a float32 numpy row of length self.cut — a row of the scoring loop's pinned
        batch buffer: no per-item tensor, no stacking copy); returns the utterance id.  Same samples as __getitem__."""
        utt_id = self.list_IDs[index]
        x = load_audio(os.path.join(self.base_dir, utt_id), 16000)
        n = x.shape[0]
        if n >= self.cut:
            out[:] = x[: self.cut]
        elif self.padding_type == "repeat":
            out[:] = pad_eval(x, "repeat", self.cut)
        else:
            out[:n] = x
            out[n:] = 0.0
        return utt_id
