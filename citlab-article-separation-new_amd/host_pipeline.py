"""Host side of one GPU-owning process: image decode and PAGE-XML writing run in worker processes AROUND it.

After the device stages moved into the single-digit-millisecond range a page costs ~110 ms of PNG / JPEG decode and
~10 ms of PAGE-XML work on the host against ~4-12 ms on the GPU, so a process that does decode ->
GPU -> XML one after the other leaves the GPU idle > 90 % of the time.  The reference fans whole sub-lists out over
``ProcessPoolExecutor(num_processes)`` workers that each own a TensorFlow session
(``run_net_post_processing.py:61-82``); here ONE process owns the GPU and ``num_processes`` host workers feed it:

    DecodePool   workers decode the next images of the list ahead of the GPU into shared-memory slots that the owner has
                 page-locked (``asep_host_register``): the upload is a DMA from the slot, nothing is pickled or copied.
    WritePool    the owner hands (page path, small results) to workers that parse / modify / write the PAGE-XML.

Both keep the order of the image list; with ``n_workers <= 1`` everything runs inline in the calling process (the
behaviour of round 1, and what the unit tests use).  Workers never touch the GPU and do not import torch.
"""
import multiprocessing as mp
import os
import queue
import traceback
from multiprocessing import shared_memory

import numpy as np

SLOT_BYTES = 64 << 20            # one decoded page: 3000 x 4500 x 3 uint8 = 40.5 MB


def _decode_worker(tasks, ready, loader_name):
    """worker process: (seq, path, slot name) -> decode -> pixels into the slot -> (seq, shape, dtype, error)"""
    from . import image_io
    loader = getattr(image_io, loader_name)
    slots = {}
    while True:
        item = tasks.get()
        if item is None:
            break
        seq, path, slot = item
        try:
            img = loader(path)
            img = np.ascontiguousarray(img)
            shm = slots.get(slot)
            if shm is None:
                shm = slots[slot] = shared_memory.SharedMemory(name=slot)
            if img.nbytes > shm.size:
                ready.put((seq, None, None, f"{path}: decoded image of {img.nbytes} bytes exceeds the {shm.size}-byte slot"))
                continue
            np.ndarray(img.shape, img.dtype, buffer=shm.buf)[...] = img
            ready.put((seq, img.shape, img.dtype.str, None))
        except Exception:                                   # surfaced in the owner, with the worker's traceback
            ready.put((seq, None, None, f"{path}: {traceback.format_exc()}"))
    for shm in slots.values():
        shm.close()


class DecodePool:
    """Iterate ``(path, image)`` over ``paths`` in order while ``n_workers`` processes decode ahead.

    ``image`` is a numpy view into a shared-memory slot: valid until the next item is requested.  ``register`` /
    ``unregister`` (optional callables ``(address, nbytes)``) page-lock the slots for DMA uploads."""

    def __init__(self, paths, n_workers=0, loader="load_image_bgr", n_slots=None, slot_bytes=SLOT_BYTES,
                 register=None, unregister=None):
        self.paths = list(paths)
        self.n_workers = max(0, int(n_workers)) if len(self.paths) > 1 else 0
        self.loader = loader
        self.n_slots = n_slots or max(2, self.n_workers + 2)
        self.slot_bytes = slot_bytes
        self._register, self._unregister = register, unregister

    def __iter__(self):
        if self.n_workers <= 1:
            from . import image_io
            load = getattr(image_io, self.loader)
            for p in self.paths:
                yield p, load(p)
            return
        ctx = mp.get_context("spawn")                       # the owner may have initialised HIP: never fork it
        tasks, ready = ctx.Queue(), ctx.Queue()
        slots = [shared_memory.SharedMemory(create=True, size=self.slot_bytes) for _ in range(self.n_slots)]
        registered = []
        procs = [ctx.Process(target=_decode_worker, args=(tasks, ready, self.loader), daemon=True)
                 for _ in range(self.n_workers)]
        try:
            if self._register:
                for s in slots:
                    addr = np.ndarray((1,), np.uint8, buffer=s.buf).ctypes.data
                    if self._register(addr, s.size):
                        registered.append(addr)
            for p in procs:
                p.start()
            free = list(range(self.n_slots))
            slot_of, done, next_task, next_out = {}, {}, 0, 0
            held = None
            n = len(self.paths)
            while next_out < n:
                while free and next_task < n:               # keep every free slot busy
                    k = free.pop()
                    slot_of[next_task] = k
                    tasks.put((next_task, self.paths[next_task], slots[k].name))
                    next_task += 1
                while next_out not in done:
                    try:
                        seq, shape, dtype, err = ready.get(timeout=1.0)
                    except queue.Empty:
                        if not any(p.is_alive() for p in procs):
                            raise RuntimeError("image decode workers died")
                        continue
                    done[seq] = (shape, dtype, err)
                shape, dtype, err = done.pop(next_out)
                if err:
                    raise IOError("image decode failed: " + err)
                if held is not None:                        # the previous page's slot is free again
                    free.append(held)
                held = slot_of.pop(next_out)
                img = np.ndarray(shape, np.dtype(dtype), buffer=slots[held].buf)
                yield self.paths[next_out], img
                del img
                next_out += 1
        finally:
            for _ in procs:
                tasks.put(None)
            for p in procs:
                p.join(timeout=5)
                if p.is_alive():
                    p.terminate()
            if self._unregister:
                for addr in registered:
                    self._unregister(addr)
            for s in slots:
                s.close()
                s.unlink()


def _run_task(fn_module, fn_name, args):
    import importlib
    return getattr(importlib.import_module(fn_module), fn_name)(*args)


class WritePool:
    """``submit(function, *args)`` runs a module-level function in a worker process (inline when n_workers <= 1);
    ``close()`` waits for everything and re-raises the first failure."""

    def __init__(self, n_workers=0):
        self.n_workers = max(0, int(n_workers))
        self._pool = None
        self._futures = []

    def submit(self, fn, *args):
        if self.n_workers <= 1:
            fn(*args)
            return
        if self._pool is None:
            from concurrent.futures import ProcessPoolExecutor
            self._pool = ProcessPoolExecutor(self.n_workers, mp_context=mp.get_context("spawn"))
        self._futures.append(self._pool.submit(_run_task, fn.__module__, fn.__name__, args))
        if len(self._futures) > 4 * self.n_workers:         # bounded backlog: surface errors early
            self._futures.pop(0).result()

    def close(self):
        try:
            for f in self._futures:
                f.result()
        finally:
            self._futures = []
            if self._pool is not None:
                self._pool.shutdown()
                self._pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def pin_callbacks(device=0):
    """(register, unregister) for DecodePool that page-lock a host range with the engine's C ABI"""
    from . import _lib
    lib = _lib.init_device(device)

    def register(addr, nbytes):
        return lib.asep_host_register(addr, nbytes) == 0

    def unregister(addr):
        lib.asep_host_unregister(addr)
    return register, unregister


def host_workers_default():
    """host workers per GPU owner when the CLI does not say: enough to hide a ~110 ms decode behind a ~10 ms GPU stage"""
    return max(1, min(12, (os.cpu_count() or 2) // 2))
