"""ORACLE — test infrastructure only.

CPU restatements (numpy float64 / torch fp32) of the reference's algorithm for every row of
SURVEY.md §8(a).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package, and only as the checker: the product path (`scl_amd`, `model/`, `datautils/`,
`main.py`) never imports it and fails loudly when libscl_hip.so is missing.

Pinning: the restatements are checked against vectors produced by the reference's own Python
(imported from /root/reference in the build container by `oracle/gen_golden.py`; the vectors live in
`tests/golden/`).  Two parts have no reference-side run to pin against and say so:
  * wav2vec2 / XLS-R forward (fairseq @ a5402130 is not installed): cross-checked against
    `transformers.Wav2Vec2Model` with copied weights instead — "parity unpinned" w.r.t. fairseq;
  * pydub 0.25.1 / audioop integer semantics (MUSAN overlay, int16 conversions): restated from the
    published semantics — "parity unpinned";
  * pydub AudioSegment.speedup and librosa 0.10.0 effects.pitch_shift (conf-5's speed / pitch augmenters,
    oracle/audio_speed_pitch.py): restated from the packages' published algorithms, checked against hand-derived
    known answers only — "parity unpinned"; soxr's resampling filter is replaced by a Kaiser-windowed sinc.
"""
