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
import contextlib
import errno
import multiprocessing as mp
import os
import queue
import sys
import threading
import time
import traceback
from multiprocessing import shared_memory

import numpy as np

SLOT_BYTES = 64 << 20            # default slot: one decoded page, 3000 x 4500 x 3 uint8 = 40.5 MB
SLOT_BYTES_MAX = 1 << 30         # larger scans are decoded inline by the owner instead of page-locking GBs per slot
_TOO_BIG = "__too_big__"


def needed_slot_bytes(paths, default=SLOT_BYTES, limit=SLOT_BYTES_MAX, probe=64):
    """Slot size from the image HEADERS (width x height x 3 channels, the widest thing a loader returns): the largest of
    the first ``probe`` files and of ``probe`` more spread over the list, rounded up to 1 MiB, never above ``limit``.  A
    scan beyond the slot is not an error: the worker reports it and the owner decodes that file inline."""
    from . import image_io
    paths = list(paths)
    pick = paths[:probe] + paths[probe::max(1, (len(paths) - probe) // probe)][:probe] if len(paths) > probe else paths
    need = 0
    for p in pick:
        try:
            w, h = image_io.get_image_dimensions(p)
            need = max(need, int(w) * int(h) * 3)
        except Exception:                                   # unreadable header: the decode itself reports the file
            pass
    if need == 0:
        need = default
    return min(limit, (need + (1 << 20) - 1) >> 20 << 20)


_THREAD_ENV = ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS")


@contextlib.contextmanager
def single_threaded_children():
    """Processes spawned inside inherit *_NUM_THREADS = 1: a decode / XML worker does no linear algebra, and numpy's BLAS
    otherwise creates one thread per core at import -- on a 256-CPU box that is most of a worker's start-up time when dozens
    of them start at once."""
    old = {k: os.environ.get(k) for k in _THREAD_ENV}
    os.environ.update({k: "1" for k in _THREAD_ENV})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _resolve_loader(name):
    """``name`` of a function in image_io, or ``"package.module:function"``"""
    if ":" in name:
        import importlib
        mod, fn = name.split(":", 1)
        return getattr(importlib.import_module(mod), fn)
    from . import image_io
    return getattr(image_io, name)


def _decode_worker(tasks, ready, loader_name):
    """worker process: (seq, path, slot name) -> decode -> pixels into the slot -> (seq, shape, dtype, error)"""
    loader = _resolve_loader(loader_name)
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
                ready.put((seq, None, None, _TOO_BIG))
                continue
            np.ndarray(img.shape, img.dtype, buffer=shm.buf)[...] = img
            ready.put((seq, img.shape, img.dtype.str, None))
        except Exception:                                   # surfaced in the owner, with the worker's traceback
            ready.put((seq, None, None, f"{path}: {traceback.format_exc()}"))
    for shm in slots.values():
        shm.close()


class DecodePool:
    """Iterate ``(path, image)`` over ``paths`` in order while ``n_workers`` processes decode ahead.

    ``image`` is a numpy view into a shared-memory slot: valid until the next item is requested (``hold`` = 2: until the one
    after the next is requested).  ``register`` /
    ``unregister`` (optional callables ``(address, nbytes)``) page-lock the slots for DMA uploads.  ``slot_bytes`` None:
    sized from the image headers (``needed_slot_bytes``).  An image that does not fit its slot is decoded inline by the
    owner; if the shared-memory slots cannot be created (a small /dev/shm) the whole list is decoded inline; if a worker
    process dies (OOM kill, SIGBUS, decoder crash) the iteration raises instead of waiting for its page forever."""

    def __init__(self, paths, n_workers=0, loader="load_image_bgr", n_slots=None, slot_bytes=None,
                 register=None, unregister=None, strict_slots=False, hold=1):
        self.paths = list(paths)
        self.n_workers = max(0, int(n_workers)) if len(self.paths) > 1 else 0
        self.loader = loader
        # hold = 2: an image stays valid while the NEXT one is in the consumer's hands too (a consumer that queues the
        # upload of page n+1 behind page n's kernels and only then waits for page n)
        self.hold = max(1, int(hold))
        self.n_slots = n_slots or max(2, self.n_workers + 1 + self.hold)
        self.slot_bytes = slot_bytes
        self.strict_slots = strict_slots                    # True: an image beyond the slot is an IOError (tests)
        self.inline_decodes = 0                             # pages the owner had to decode itself
        self._register, self._unregister = register, unregister

    def _inline(self, start=0):
        load = _resolve_loader(self.loader)
        for p in self.paths[start:]:
            yield p, load(p)

    def __iter__(self):
        if self.n_workers <= 1:
            yield from self._inline()
            return
        load = _resolve_loader(self.loader)
        slot_bytes = self.slot_bytes or needed_slot_bytes(self.paths)
        ctx = mp.get_context("spawn")                       # the owner may have initialised HIP: never fork it
        # one task queue per worker: the owner knows which worker holds which page, so a dead worker's pages can be named
        tasks, ready = [ctx.Queue() for _ in range(self.n_workers)], ctx.Queue()
        slots, registered = [], []
        procs = [ctx.Process(target=_decode_worker, args=(tasks[i], ready, self.loader), daemon=True)
                 for i in range(self.n_workers)]
        free = []

        def add_slot():
            """one more shared-memory slot, touched (a /dev/shm that is too small fails here, not as SIGBUS in a worker) and
            page-locked; False when the host has no room for it"""
            try:
                s = shared_memory.SharedMemory(create=True, size=slot_bytes)
            except (OSError, MemoryError, ValueError):
                return False
            try:
                fd = getattr(s, "_fd", -1)
                reserved = False
                if fd >= 0 and hasattr(os, "posix_fallocate"):
                    try:
                        os.posix_fallocate(fd, 0, s.size)   # reserves the pages (ENOSPC now) without faulting each one in
                        reserved = True
                    except OSError as e:
                        if e.errno in (errno.ENOSPC, errno.ENOMEM, errno.EDQUOT):
                            raise                           # (anything else: this file system cannot preallocate -- touch instead)
                if not reserved:
                    np.ndarray((s.size,), np.uint8, buffer=s.buf)[::4096] = 0
            except (OSError, MemoryError, ValueError):
                s.close()
                s.unlink()
                return False
            if self._register:
                addr = np.ndarray((1,), np.uint8, buffer=s.buf).ctypes.data
                if self._register(addr, s.size):
                    registered.append(addr)
            slots.append(s)
            free.append(len(slots) - 1)                     # (list.append is atomic: the preparer thread may be the caller)
            return True

        def prepare_rest():
            """preparer thread: the remaining slots, while the first pages are already being decoded and consumed (reserving and
            page-locking a slot takes tens of milliseconds; both calls release the interpreter lock)"""
            while len(slots) < self.n_slots and not stop_preparing:
                if not add_slot():                          # the host has no room for more: run with the slots there are
                    self.n_slots = len(slots)
                    break

        inline_instead = False
        preparer, stop_preparing = None, []
        trace = os.environ.get("ASEP_POOL_TRACE") == "1"     # start-up timeline on stderr (scripts/e2e_feed_bench.py)
        t_start = time.perf_counter()

        def mark(what):
            if trace:
                print(f"[DecodePool] {time.perf_counter() - t_start:7.3f} s  {what}", file=sys.stderr, flush=True)
        try:
            # the workers start first and import while the owner prepares slots; a slot is given out as soon as it exists, the
            # remaining ones are created while the first pages are being decoded
            with single_threaded_children():
                for p in procs:
                    p.start()
            mark(f"{len(procs)} workers started")
            while len(slots) < self.hold + 1:               # the fewest slots the consumer's contract needs
                if not add_slot():
                    inline_instead = True
                    break
            if not inline_instead and len(self.paths) > len(slots):
                preparer = threading.Thread(target=prepare_rest, daemon=True)
                preparer.start()
            slot_of, done, next_task, next_out = {}, {}, 0, 0
            outstanding = [set() for _ in procs]            # pages handed to each worker and not yet reported
            worker_of = {}
            held = []                                       # slots of the last `hold` pages handed out, oldest first
            n = len(self.paths)
            while next_out < n and not inline_instead:
                while free and next_task < n:               # keep every slot that exists busy
                    k = free.pop()
                    slot_of[next_task] = k
                    wi = min(range(len(procs)), key=lambda i: len(outstanding[i]))
                    outstanding[wi].add(next_task)
                    worker_of[next_task] = wi
                    tasks[wi].put((next_task, self.paths[next_task], slots[k].name))
                    next_task += 1
                while next_out not in done:
                    try:
                        # (a short wait while slots are still being prepared: new ones are handed out as they appear)
                        seq, shape, dtype, err = ready.get(timeout=0.005 if preparer and preparer.is_alive() else 1.0)
                    except queue.Empty:
                        if free and next_task < n:
                            break
                        dead = [i for i, p in enumerate(procs) if not p.is_alive() and outstanding[i]]
                        if dead:                            # a worker died holding pages: they would never arrive
                            lost = sorted(self.paths[q] for i in dead for q in outstanding[i])
                            raise RuntimeError(f"{len(dead)} of {len(procs)} image decode workers died (exit codes "
                                               f"{[procs[i].exitcode for i in dead]}) while holding {lost}")
                        continue
                    outstanding[worker_of.pop(seq)].discard(seq)
                    done[seq] = (shape, dtype, err)
                if next_out not in done:                    # left the wait to hand out a slot that has just appeared
                    continue
                shape, dtype, err = done.pop(next_out)
                if next_out < 3:
                    mark(f"page {next_out} decoded ({len(slots)} slots exist, {next_task} pages handed out)")
                if len(held) >= self.hold:                  # the oldest page still held gives its slot back
                    free.append(held.pop(0))
                held.append(slot_of.pop(next_out))
                if err == _TOO_BIG and not self.strict_slots:
                    self.inline_decodes += 1
                    img = load(self.paths[next_out])        # the owner decodes what no slot can hold
                elif err:
                    raise IOError("image decode failed: " + (f"{self.paths[next_out]}: decoded image exceeds the "
                                                             f"{slot_bytes}-byte slot" if err == _TOO_BIG else err))
                else:
                    img = np.ndarray(shape, np.dtype(dtype), buffer=slots[held[-1]].buf)
                yield self.paths[next_out], img
                del img
                next_out += 1
        finally:
            mark("last page handed out")
            stop_preparing.append(True)
            if preparer:
                preparer.join()
            for q in tasks:
                q.put(None)
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
            mark("workers joined, slots released")
        if inline_instead:                                  # no room for the shared-memory slots (a small /dev/shm)
            self.inline_decodes = len(self.paths)
            yield from self._inline()


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
        with single_threaded_children():                     # (the executor spawns its processes inside submit, on demand)
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
    seen = threading.local()

    def register(addr, nbytes):
        if not getattr(seen, "device_set", False):          # the slot preparer is a thread of its own: HIP's current device is
            lib.asep_init(device)                           # per thread, and an owner of GPU k must not touch GPU 0
            seen.device_set = True
        return lib.asep_host_register(addr, nbytes) == 0

    def unregister(addr):
        lib.asep_host_unregister(addr)
    return register, unregister


def host_workers_default():
    """host workers per GPU owner when the caller does not say: enough to hide a ~110 ms PNG decode (+ the PAGE-XML write) behind
    a ~9 ms GPU stage, few enough that eight owners fit a 256-CPU box, and sized by what the container may use
    (``effective_cpus``), not by the machine: on a box of this pool (256 logical CPUs under a cgroup quota of 16) 14 workers
    feed 83 pages/s and 24 feed 94 -- the workers also wait for files and queues, so 1.5 per CPU is the better fill -- and
    beyond that the quota throttles the whole group, the owner included"""
    from .host_util import effective_cpus
    cpus = effective_cpus()
    return max(1, min(24, cpus // 4 if cpus >= 96 else cpus * 3 // 2))
