"""Development aid: phase stamps (s_memtime) of the 8-wave convb_kernel blocks (>= 64-channel layers of the bf16 path).  Needs a
library built with -DCVB_TRACE (ASEP_HIP_LIB=...): python scripts/gpu_cvb_trace.py [H W].  The buffer holds the LAST launch of that
instantiation in a forward."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from citlab_article_separation_new_amd.config import AruConfig
from citlab_article_separation_new_amd.weights import init_aru_weights
from citlab_article_separation_new_amd import net_post_processing_helper as helper, _lib
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4500, 3000)
cfg = AruConfig(compute_dtype='bf16')
g = helper.AruGraph(init_aru_weights(cfg, 1234), cfg)
lib = _lib.init_device(0); h = g.handle(0)
img = torch.rand(H, W, device='cuda'); out = torch.empty(H, W, 2, device='cuda')
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    _lib.check(lib.asep_aru_forward_dev(h, img.data_ptr(), H, W, out.data_ptr(), None, None, 0.05, s), "fwd")
torch.cuda.synchronize()
names = ["entry -> set-up done", "first stage: requests -> tile + weights in LDS (barrier passed)", "first stage MFMAs", "middle stages",
         "last stage MFMAs", "epilogue (stores issued)", "stores drained"]
n = 4096 * 8
buf = (C.c_ulonglong * n)()
rc = lib.asep_debug_cvb_trace(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
ok = (a[:, 0] > 0) & (a[:, 7] > a[:, 0]) & (a[:, 7] - a[:, 0] < 10_000_000)
a = a[ok]
print(f"rc {rc}, {len(a)} sampled blocks")
d = np.diff(a, axis=1)
for i, nm in enumerate(names):
    print(f"   {nm:70s} mean {d[:, i].mean():8.0f}  median {np.median(d[:, i]):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
life = a[:, 7] - a[:, 0]
print(f"   block life: mean {life.mean():.0f} median {np.median(life):.0f} ticks; launch span {a[:, 7].max() - a[:, 0].min()} ticks")
