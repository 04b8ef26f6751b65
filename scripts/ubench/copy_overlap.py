"""Does an upload on a side stream run beside kernels that are queued on the main stream?  For page-locked torch memory, for a
numpy buffer registered with hipHostRegister (what DecodePool's slots are) and for pageable memory; main stream = torch's default
stream and a non-default one.  Prints the wait for the copy with the chip idle and with ~20 ms of kernels queued."""
import ctypes as C
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from citlab_article_separation_new_amd import _lib

lib = _lib.init_device(0)
dev = torch.device("cuda", 0)
n = 3000 * 4500
a = torch.randn(8192, 8192, device=dev)
pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
reg = np.zeros(n, np.uint8)
assert lib.asep_host_register(reg.ctypes.data, reg.nbytes) == 0
pageable = np.zeros(n, np.uint8)
copy = torch.cuda.Stream(dev)


def busy(k=8):
    for _ in range(k):
        torch.mm(a, a)


for main_name, main in (("default stream", torch.cuda.default_stream(dev)), ("side stream", torch.cuda.Stream(dev))):
    for name, src in (("torch pinned", pinned), ("hipHostRegister", torch.from_numpy(reg)), ("pageable", torch.from_numpy(pageable))):
        for load in (False, True):
            torch.cuda.synchronize()
            with torch.cuda.stream(main):
                t0 = time.perf_counter()
                if load:
                    busy()
                t1 = time.perf_counter()
                with torch.cuda.stream(copy):
                    d = src.to(dev, non_blocking=True)
                copy.synchronize()
                t2 = time.perf_counter()
                main.synchronize()
                t3 = time.perf_counter()
            print(f"main = {main_name:14s} src = {name:16s} kernels queued: {load!s:5s}  queueing {1e3 * (t1 - t0):6.2f} ms, "
                  f"copy wait {1e3 * (t2 - t1):6.2f} ms, main done after another {1e3 * (t3 - t2):6.2f} ms")
