"""Do two page lanes run better when they are out of phase?  Inside one asep_aru_forward_batch_dev call the lanes start together, so
level-0 blocks (vector ALU) meet level-0 blocks and deep Winograd layers (LDS / MFMA) meet Winograd layers.  Here: two model
instances on two streams, each a single lane, 8 pages per call, never joined -- once in phase, once with the second stream half a
pass behind -- against the engine's own two-lane call on 16 pages.        python scripts/lane_offset_probe.py [steps=12] [dtype=f32]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
mode = sys.argv[3] if len(sys.argv) > 3 else "two"          # "two": two single-lane instances; "engine": the engine's own lanes
os.environ["ASEP_LANES"] = "1" if mode == "two" else "2"
import torch

from citlab_article_separation_new_amd import _lib, net_post_processing_helper as helper
from citlab_article_separation_new_amd.config import AruConfig
from citlab_article_separation_new_amd.weights import init_aru_weights

H, W = 4500, 3000
cfg = AruConfig(compute_dtype=dtype)
graph = helper.AruGraph(init_aru_weights(cfg, 21, logit_scale=0.05), cfg)
lib = _lib.init_device(0)
pages = [torch.rand(H, W, device="cuda") for _ in range(16)]
outs = [torch.empty(H, W, 2, device="cuda") for _ in range(16)]


def arr(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def call(h, lo, hi, stream):
    _lib.check(lib.asep_aru_forward_batch_dev(h, hi - lo, arr(pages[lo:hi]), H, W, arr(outs[lo:hi]), None, None, 0.05,
                                              C.c_void_p(stream.cuda_stream)), "batch")


if mode == "engine":
    h = graph.handle(0)
    s = torch.cuda.Stream()
    for _ in range(2):
        call(h, 0, 16, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        call(h, 0, 16, s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{dtype} engine lanes (2), 16 pages per call: {16 * steps / dt:.1f} pages/s")
else:
    h0, h1 = graph.handle(0, 0), graph.handle(0, 1)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    for offset in (False, True, False, True):
        for _ in range(2):
            call(h0, 0, 8, s0); call(h1, 8, 16, s1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if offset:                                           # the second stream starts when the first is half a pass ahead
            call(h0, 0, 4, s0)
            ev = torch.cuda.Event()
            ev.record(s0)
            s1.wait_event(ev)
        for _ in range(steps):
            call(h0, 0, 8, s0); call(h1, 8, 16, s1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = 16 * steps + (4 if offset else 0)
        print(f"{dtype} two single-lane instances, 8 pages per call each, offset {offset!s:5s}: {n / dt:.1f} pages/s")
