"""Host side of the two dataset walks: keep the device fed.

The reference's loops (``activation_based.py:341-358`` and ``:392-433``) are synchronous: fetch a batch from the
DataLoader (``torch.stack`` of 256 samples = 154 MB into fresh pageable memory), copy it to the device from pageable
memory (blocking), run.  With the collect and gather steps on the device that host work became the critical path of
the API (73 ms per batch of 256 against 45 ms of device work).  Here a background thread walks the DataLoader
``depth`` batches ahead:

* ``PinnedStack`` — collate function of the probed model's loader: stacks the samples straight into a pinned ring
  buffer (one copy instead of three, no page faults), labels are dropped like the reference's ``for images, _ in ...``;
* ``Prefetcher`` — iterates any iterable in a thread, applies ``stage`` to every item under its own HIP stream
  (the host->device copy, or ``fm.preprocess`` of a list of raw images) and hands ``(result, event)`` to the consumer,
  which makes its stream wait for the event.  Order is the loader's order; an exception in the thread is re-raised in
  the consumer at the position where it occurred.

Plumbing only: no arithmetic happens here, and nothing here is used without a HIP device.
"""
from __future__ import annotations

import queue
import threading

import torch
from torch.utils.data import default_collate


class PinnedStack:
    """``collate_fn``: ``[(image, label), ...] -> (B, ...) images`` stacked into a ring of pinned host buffers.

    A slot is rewritten ``slots`` batches later; ``release(slot, event)`` lets the consumer of a batch say when its
    upload has completed.  Samples that are not same-shaped tensors fall back to ``default_collate`` (pageable).
    """

    def __init__(self, slots: int = 4):
        self.slots = slots
        self._bufs: list[torch.Tensor | None] = [None] * slots
        self._events: list[torch.cuda.Event | None] = [None] * slots
        self._next = 0
        self.last_slot = -1

    def __call__(self, batch):
        first = batch[0]
        images = [b[0] for b in batch] if isinstance(first, (tuple, list)) else list(batch)
        img0 = images[0]
        same = isinstance(img0, torch.Tensor) and not img0.is_cuda and all(
            isinstance(t, torch.Tensor) and t.shape == img0.shape and t.dtype == img0.dtype for t in images)
        if not same or not torch.cuda.is_available():
            self.last_slot = -1
            out = default_collate(batch)
            return out[0] if isinstance(first, (tuple, list)) else out
        slot = self._next
        self._next = (slot + 1) % self.slots
        if self._events[slot] is not None:
            self._events[slot].synchronize()  # the upload that last read this slot
            self._events[slot] = None
        buf = self._bufs[slot]
        need = (len(images),) + tuple(img0.shape)
        if buf is None or buf.dtype != img0.dtype or buf.shape[1:] != need[1:] or buf.shape[0] < need[0]:
            buf = self._bufs[slot] = torch.empty(need, dtype=img0.dtype).pin_memory()
        out = buf[: need[0]]
        torch.stack(images, out=out)
        self.last_slot = slot
        return out

    def release(self, slot: int, event: torch.cuda.Event):
        if slot >= 0:
            self._events[slot] = event


_END = object()


class Prefetcher:
    """``for result in Prefetcher(iterable, stage, device)``: ``stage(item)`` runs in a background thread, on a HIP
    stream of its own, ``depth`` items ahead; the consumer's current stream waits for the item's event and the
    device tensors of ``result`` are marked as used by it (``record_stream``)."""

    def __init__(self, iterable, stage, device, depth: int = 2):
        self._iterable, self._stage = iterable, stage
        self._device = torch.device(device)
        self._q: queue.Queue = queue.Queue(maxsize=max(1, depth))
        self._stop = threading.Event()
        self._stream = torch.cuda.Stream(self._device)
        self._thread = threading.Thread(target=self._work, name="sl-prefetch", daemon=True)
        self._started = False

    def __len__(self):
        return len(self._iterable)

    def _put(self, item) -> bool:
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _work(self):
        try:
            torch.cuda.set_device(self._device)
            with torch.cuda.stream(self._stream), torch.no_grad():
                for item in self._iterable:
                    if self._stop.is_set():
                        return
                    result = self._stage(item)
                    event = torch.cuda.Event()
                    event.record(self._stream)
                    if not self._put((result, event, None)):
                        return
            self._put((_END, None, None))
        except BaseException as exc:  # noqa: BLE001 — handed to the consumer
            self._put((_END, None, exc))

    def __iter__(self):
        if self._started:
            raise RuntimeError("a Prefetcher can be iterated once")
        self._started = True
        self._thread.start()
        try:
            while True:
                result, event, exc = self._q.get()
                if result is _END:
                    if exc is not None:
                        raise exc
                    return
                cur = torch.cuda.current_stream(self._device)
                cur.wait_event(event)
                for t in _tensors(result):
                    if t.is_cuda:
                        t.record_stream(cur)
                yield result
        finally:
            self.close()

    def close(self):
        self._stop.set()
        while self._thread.is_alive():  # unblock a producer waiting on a full queue
            try:
                self._q.get_nowait()
            except queue.Empty:
                pass
            self._thread.join(timeout=0.05)


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            yield from _tensors(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors(o)


def upload_stage(device, stack: PinnedStack | None = None):
    """``stage`` of the probed model's loader: batch (pinned or pageable host tensor, possibly ``(images, labels)``)
    -> device tensor; tells ``stack`` when its slot may be rewritten."""
    device = torch.device(device)

    def stage(batch):
        images = batch[0] if isinstance(batch, (tuple, list)) else batch
        slot = stack.last_slot if stack is not None else -1
        out = images.to(device, non_blocking=True)
        if slot >= 0:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            stack.release(slot, ev)
        return out

    return stage
