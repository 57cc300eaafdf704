"""Host pipeline: build packs i+1, i+2, ... (decode, parameter sampling, GPU augmentation) while pack i trains.

The reference hides its augmentation behind 8 forked DataLoader workers (main.py:379,392).  Here the augmentation runs on the GPU, so
forked workers are out (the parent has initialised the GPU) — THREADS build the batches instead: file decoding (the FLAC decoder is C
behind ctypes), numpy filter design and the C-ABI launches all release the GIL, and every builder thread issues its kernels on its OWN
HIP stream, so they overlap the training step's kernels instead of queueing behind them.

Round 6: one builder thread made 98 packs/s = 1078 utterances/s against a GPU that trains 1500+.  What fixed it: the builder no longer
synchronises with the GPU (scl_amd/pack.py, augment.py: pinned asynchronous uploads, host-side integer powers), costs less (fast sampler, a
1.3 - 1.5 x faster one-pass FLAC decoder) and finds its files decoded AHEAD: a plain in-process DataLoader is driven from its batch_sampler
(shuffle / drop_last / Subset semantics unchanged), so the index batches are known two ahead and the dataset is hinted
(`decode_ahead(indices)`, decoder threads in C).  `workers` > 1 builder threads take one sequence number each and hand over IN ORDER —
measured, they never beat one builder inside a training run (they share the interpreter lock with the launch thread): opt-in
(SCL_PREFETCH_THREADS).  Any other iterable is simply iterated on one thread.  Hand-over is an event: the consumer's stream waits for it and the tensors are marked as used on that
stream (allocator safety).  Exceptions of a producer surface in the consumer.  Like the reference's workers, concurrent builders draw
from the process-wide `random` / `np.random` streams in whatever order they run: every draw has the right distribution, the assignment of
draws to packs is not reproducible (SCL_PREFETCH_THREADS=1 restores one sequential stream).
"""
import collections
import os
import queue
import sys
import threading

import torch


def _augment():
    from . import augment      # imported on first use: the prefetcher itself needs neither scipy nor the HIP library
    return augment


def default_workers():
    """ONE builder thread by default.  Measured on MI355X boxes (profiles/r6_pack_builder.txt): with no host <-> device synchronisation
    left in the builder, the fast RawBoost sampler, the round-6 FLAC decoder and the files decoded ahead on decoder threads, one thread
    builds 227 - 256 packs/s of 11 views = 2500 - 2800 utterances/s — the GPU trains 1580 - 1640 at 6 packs per step — and end to end
    main.run_epoch runs at 1.03 - 1.05 x the resident-batch step (profiles/r6_upload_stream_final.txt).  More builder threads share the interpreter lock with the launch thread
    (~760 C calls per step) and never beat one inside a training run (1.05 - 1.27 x).  SCL_PREFETCH_THREADS=<n> to try it anyway."""
    try:
        return max(1, int(os.environ.get("SCL_PREFETCH_THREADS", "1")))
    except ValueError:
        return 1


class Prefetcher:
    def __init__(self, loader, depth=2, device=None, workers=None):
        self.loader, self.depth = loader, max(1, int(depth))
        self.device = torch.device(device) if device is not None else None
        self.workers = default_workers() if workers is None else max(1, int(workers))

    def __len__(self):
        return len(self.loader)

    def _parallel_ok(self):
        # a plain in-process DataLoader is driven from its batch_sampler (also with ONE builder: the index batches are then known two
        # ahead, which is what lets the dataset decode the next items' files on its decoder threads: PackDataset.decode_ahead)
        ld = self.loader
        return (isinstance(ld, torch.utils.data.DataLoader) and ld.num_workers == 0 and ld.batch_sampler is not None
                and not isinstance(ld.dataset, torch.utils.data.IterableDataset))

    def __iter__(self):
        return self._iter_parallel() if self._parallel_ok() else self._iter_serial()

    # ---- several builders, ordered hand-over ------------------------------------------------------------------------------------
    def _iter_parallel(self):
        ld = self.loader
        dataset, collate = ld.dataset, ld.collate_fn
        use_gpu = self.device is not None and self.device.type == "cuda"
        batches = iter(ld.batch_sampler)
        lock, cv = threading.Lock(), threading.Condition()
        slots = threading.Semaphore(self.depth + self.workers)      # batches built or being built but not yet consumed
        stop = threading.Event()
        done = {}                    # sequence number -> (item | exception, event)
        state = {"next": 0, "end": None, "drained": False}
        ahead = collections.deque()
        hint = _decode_hint(dataset)

        def build(wid):
            side = torch.cuda.Stream(device=self.device) if use_gpu else None
            try:
                if use_gpu:
                    torch.cuda.set_device(self.device)
                if use_gpu:
                    _augment().use_upload_stream(False)      # this thread's stream is nearly empty: copies go straight onto it
                ctx = torch.cuda.stream(side) if use_gpu else _null()
                with ctx:
                    while not stop.is_set():
                        if not slots.acquire(timeout=0.1):
                            continue
                        with lock:                      # the sampler is not thread-safe; the sequence number fixes the hand-over order
                            while len(ahead) < self.workers + 2 and not state["drained"]:      # index batches drawn (and hinted) ahead of their build
                                try:
                                    nb = next(batches)
                                except StopIteration:
                                    state["drained"] = True
                                    break
                                ahead.append(nb)
                                if hint is not None:
                                    hint(nb)
                            if not ahead:
                                if state["end"] is None:
                                    state["end"] = state["next"]
                                slots.release()
                                with cv:
                                    cv.notify_all()
                                return
                            idxs = ahead.popleft()
                            seq = state["next"]
                            state["next"] += 1
                        try:
                            item = collate([dataset[i] for i in idxs])
                            ev = None
                            if use_gpu:
                                ev = torch.cuda.Event()
                                ev.record(side)
                            res = (item, ev)
                        except BaseException as e:      # noqa: BLE001 - re-raised in the consumer, in order
                            res = (e, None)
                        with cv:
                            done[seq] = res
                            cv.notify_all()
            except BaseException as e:                  # noqa: BLE001 - a failure outside a batch (stream creation, sampler)
                with cv:
                    done[-1] = (e, None)
                    cv.notify_all()

        threads = [threading.Thread(target=build, args=(w,), name="scl-prefetch-%d" % w, daemon=True) for w in range(self.workers)]
        # The launch thread (the consumer) replays ~760 short C calls per step and needs the interpreter lock back after each: with the
        # default 5 ms switch interval a builder in a pure-Python stretch keeps it waiting up to 5 ms per hand-over.
        old_switch = sys.getswitchinterval()
        sys.setswitchinterval(min(old_switch, 5e-4))
        for th in threads:
            th.start()
        try:
            seq = 0
            while True:
                with cv:
                    while seq not in done and -1 not in done and not (state["end"] is not None and seq >= state["end"]):
                        cv.wait(timeout=0.5)
                    if -1 in done:
                        raise done[-1][0]
                    if seq not in done:
                        break
                    item, ev = done.pop(seq)
                slots.release()
                if isinstance(item, BaseException):
                    raise item
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    for t in _tensors(item):
                        if t.is_cuda:
                            t.record_stream(cur)
                yield item
                seq += 1
        finally:
            stop.set()
            for th in threads:
                th.join(timeout=5.0)
            sys.setswitchinterval(old_switch)

    # ---- one builder iterating the loader (round 3 - 5) ---------------------------------------------------------------------------
    def _iter_serial(self):
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        use_gpu = self.device is not None and self.device.type == "cuda"
        side = torch.cuda.Stream(device=self.device) if use_gpu else None
        END = object()

        def produce():
            try:
                if use_gpu:
                    torch.cuda.set_device(self.device)
                if use_gpu:
                    _augment().use_upload_stream(False)      # this thread's stream is nearly empty: copies go straight onto it
                ctx = torch.cuda.stream(side) if use_gpu else _null()
                with ctx:
                    for item in self.loader:
                        ev = None
                        if use_gpu:
                            ev = torch.cuda.Event()
                            ev.record(side)
                        while not stop.is_set():
                            try:
                                q.put((item, ev), timeout=0.1)
                                break
                            except queue.Full:
                                continue
                        if stop.is_set():
                            return
                q.put((END, None))
            except BaseException as e:      # noqa: BLE001 - re-raised in the consumer
                q.put((e, None))

        th = threading.Thread(target=produce, name="scl-prefetch", daemon=True)
        old_switch = sys.getswitchinterval()
        sys.setswitchinterval(min(old_switch, 5e-4))      # see _iter_parallel: the launch thread must not wait 5 ms for the interpreter lock
        th.start()
        try:
            while True:
                item, ev = q.get()
                if item is END:
                    break
                if isinstance(item, BaseException):
                    raise item
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    for t in _tensors(item):
                        if t.is_cuda:
                            t.record_stream(cur)
                yield item
        finally:
            stop.set()
            th.join(timeout=5.0)
            sys.setswitchinterval(old_switch)


def _decode_hint(dataset):
    """dataset.decode_ahead(indices) if the dataset (or the dataset under a Subset) offers it: the prefetcher tells it which items come next."""
    idx_map = None
    while isinstance(dataset, torch.utils.data.Subset):
        inner = dataset.indices
        idx_map = inner if idx_map is None else [inner[i] for i in idx_map]
        dataset = dataset.dataset
    fn = getattr(dataset, "decode_ahead", None)
    if fn is None:
        return None
    if idx_map is None:
        return lambda idxs: fn(list(idxs))
    return lambda idxs: fn([idx_map[i] for i in idxs])


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors(o)
