"""Per-layer HIP-event timing of one page (development aid): python scripts/gpu_layer_profile.py [H W]"""
import os, sys, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from citlab_article_separation_new_amd.config import AruConfig
from citlab_article_separation_new_amd.weights import init_aru_weights
from citlab_article_separation_new_amd import net_post_processing_helper as helper, _lib
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4500, 3000)
dtype = sys.argv[3] if len(sys.argv) > 3 else 'f32'
cfg = AruConfig(compute_dtype=dtype)
g = helper.AruGraph(init_aru_weights(cfg, 1234), cfg)
lib = _lib.init_device(0); h = g.handle(0)
img = torch.rand(H, W, device='cuda'); out = torch.empty(H, W, 2, device='cuda')
s = torch.cuda.current_stream().cuda_stream
def step(): _lib.check(lib.asep_aru_forward_dev(h, img.data_ptr(), H, W, out.data_ptr(), None, None, 0.05, s), "fwd")
step(); step(); torch.cuda.synchronize()
lib.asep_aru_profile(h, 2)
for _ in range(3): step()
torch.cuda.synchronize()
buf = C.create_string_buffer(1 << 20)
_lib.check(lib.asep_aru_profile_report(h, buf, len(buf)), "report")
ks = json.loads(buf.value.decode())
tot = sum(k["total_ms"] for k in ks) / 3
print(f"total {tot:.3f} ms/page")
for k in ks:
    ms = k["total_ms"] / k["calls"]; tf = k["flops"] / k["calls"] / (ms * 1e-3) / 1e12 if ms > 0 else 0
    print(f"{ms*1e3:9.1f} us {tf:7.1f} TF  {k['kernel']}")
