"""The GPU owner's device stages without files around them: decoded pages in page-locked host memory -> enqueue_page / collect_page,
one page behind the chip, on one lane and on two.  Says what the chip needs per page for upload + resize + net + classical stages +
segments, i.e. the ceiling of the files-in / files-out rate.      python scripts/e2e_owner_probe.py [n_pages=96]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from citlab_article_separation_new_amd import net_post_processing_helper as helper, polygonize, synth
from citlab_article_separation_new_amd.config import AruConfig
from citlab_article_separation_new_amd.separator_net_post_processor import SeparatorNetPostProcessor
from citlab_article_separation_new_amd.weights import init_aru_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
W, H = 3000, 4500
cfg = AruConfig(compute_dtype=os.environ.get("ASEP_COMPUTE_DTYPE", "f32"))
graph = helper.AruGraph(init_aru_weights(cfg, 21, logit_scale=0.05), cfg)
pages = [torch.from_numpy(synth.synth_page(k, W, H)).pin_memory().numpy() for k in range(4)]
proc = SeparatorNetPostProcessor([], graph, H, 1.0, 0.5, "0")
from citlab_article_separation_new_amd import _lib
from multiprocessing import shared_memory
lib = _lib.init_device(0)
shms = []
if os.environ.get("PROBE_SRC", "pinned") == "shm":          # like DecodePool's slots: shared memory, hipHostRegister'ed
    for k in range(4):
        s = shared_memory.SharedMemory(create=True, size=W * H)
        a = np.ndarray((H, W), np.uint8, buffer=s.buf)
        a[...] = pages[k]
        assert lib.asep_host_register(a.ctypes.data, s.size) == 0
        shms.append(s)
        pages[k] = a
for lanes, chain in ((1, False), (2, False), (2, True), (3, True)):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pending = []
        for k in range(n):
            pending.append(proc.enqueue_page(pages[k % 4], edges_only=True, lane=k % lanes))
            if len(pending) >= lanes + (0 if lanes > 1 else 1):
                masks, sc, extras = proc.collect_page(pending.pop(0))
                if chain:
                    for starts, ends in masks.values():
                        polygonize.shapes_from_segments(starts, ends, H, W, connectivity=8)
        for t in pending:
            proc.collect_page(t)
        dt = time.perf_counter() - t0
    print(f"lanes {lanes}, ring chaining {chain!s:5s}: {dt / n * 1e3:6.2f} ms/page = {n / dt:6.1f} pages/s")

for k, s in enumerate(shms):
    lib.asep_host_unregister(pages[k].ctypes.data)
pages = None
for s in shms:
    s.close()
    s.unlink()
