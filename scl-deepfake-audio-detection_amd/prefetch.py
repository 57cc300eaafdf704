"""Host pipeline: build pack i+1 (decode, parameter sampling, GPU augmentation) while pack i trains.

The reference hides its augmentation behind 8 forked DataLoader workers (main.py:379).  Here the augmentation runs on the GPU, so
forked workers are out (the parent has initialised the GPU) — a THREAD iterates the DataLoader instead: file decoding, scipy / numpy
filter design and the C-ABI launches all release the GIL, and the thread issues its kernels on its own HIP stream, so they overlap the
training step's kernels instead of queueing behind them.  Hand-over is an event: the consumer's stream waits for it and the tensors
are marked as used on that stream (allocator safety).  Exceptions of the producer surface in the consumer.
"""
import queue
import threading

import torch


class Prefetcher:
    def __init__(self, loader, depth=2, device=None):
        self.loader, self.depth = loader, max(1, int(depth))
        self.device = torch.device(device) if device is not None else None

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        use_gpu = self.device is not None and self.device.type == "cuda"
        side = torch.cuda.Stream(device=self.device) if use_gpu else None
        END = object()

        def produce():
            try:
                if use_gpu:
                    torch.cuda.set_device(self.device)
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
