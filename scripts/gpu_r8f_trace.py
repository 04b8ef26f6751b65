"""Development aid: phase time stamps (s_memtime) of sampled res8f_kernel blocks.  Needs a library built with -DR8F_TRACE
(make CXXFLAGS+="-DR8F_TRACE -DR8F_TRACE_TID=0" into a separate .so, ASEP_HIP_LIB=<that .so>): python scripts/gpu_r8f_trace.py [H W]
Ticks are shader clocks (~2 GHz under load).  Round-3 reading at 4500 x 3000 (DESIGN lesson 20): a block of res8f_kernel<true> lives
18.5 k ticks: 5.5 k waiting for its input window, 4.7 k in conv1, 2.0 - 2.2 k per tail stage."""
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
names = ["entry->located", "located->window in regs+LDS writes issued", "->barrier 1 passed", "conv1", "barrier 2", "stage 1", "barrier 3",
         "stage 2", "barrier 4", "stage 3 (stores issued)", "stores drained"]
for up in (0, 1):
    n = 4096 * 12
    buf = (C.c_ulonglong * n)()
    rc = lib.asep_debug_r8f_trace(up, buf, n)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 12).astype(np.int64)
    ok = (a[:, 0] > 0) & (a[:, 11] > a[:, 0])            # interior tiles that ran all marks
    a = a[ok]
    print(f"res8f_kernel<{bool(up)}>: rc {rc}, {len(a)} sampled interior blocks; span of the samples {a[:, 11].max() - a[:, 0].min()} ticks")
    d = np.diff(a, axis=1)
    for i, nm in enumerate(names):
        print(f"   {nm:45s} mean {d[:, i].mean():8.0f}  median {np.median(d[:, i]):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
    life = a[:, 11] - a[:, 0]
    print(f"   block life: mean {life.mean():.0f} median {np.median(life):.0f} ticks")
    st = np.sort(a[:, 0] - a[:, 0].min())
    print("   start times of the samples (ticks), every 256th:", st[::256][:20])
