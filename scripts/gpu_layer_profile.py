"""Per-layer HIP-event timing of one page (development aid): python scripts/gpu_layer_profile.py [H W [dtype [passes]]]
Every layer is timed in `passes` (default 5) separate forwards, each alone on the chip (asep_aru_profile mode 2); printed is the MEDIAN
per layer with the spread of the samples, and the page total as the sum of the medians -- one sample per layer let a 3x outlier of a
single launch into profiles/r3q/layers_one_page.log."""
import os, sys, json, ctypes as C, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from citlab_article_separation_new_amd.config import AruConfig
from citlab_article_separation_new_amd.weights import init_aru_weights
from citlab_article_separation_new_amd import net_post_processing_helper as helper, _lib
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4500, 3000)
dtype = sys.argv[3] if len(sys.argv) > 3 else 'f32'
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 5
kw = json.loads(os.environ.get("ASEP_LAYER_PROFILE_CFG", "{}"))          # e.g. '{"activation_name": "elu"}' for the graph variants
cfg = AruConfig(compute_dtype=dtype, **kw)
g = helper.AruGraph(init_aru_weights(cfg, 1234), cfg)
lib = _lib.init_device(0); h = g.handle(0)
img = torch.rand(H, W, device='cuda'); out = torch.empty(H, W, cfg.n_classes, device='cuda')
s = torch.cuda.current_stream().cuda_stream
NP = int(os.environ.get("ASEP_LAYER_PROFILE_PAGES", "1"))                # pages per call (the bench step runs 4 pages per launch)
outs = [torch.empty(H, W, cfg.n_classes, device='cuda') for _ in range(NP)]
Arr = C.c_void_p * NP
p_img, p_out = Arr(*[img.data_ptr()] * NP), Arr(*[o.data_ptr() for o in outs])
def step():
    if NP == 1: _lib.check(lib.asep_aru_forward_dev(h, img.data_ptr(), H, W, out.data_ptr(), None, None, 0.05, s), "fwd")
    else: _lib.check(lib.asep_aru_forward_batch_dev(h, NP, p_img, H, W, p_out, None, None, 0.05, s), "fwd")
step(); step(); torch.cuda.synchronize()
samples, order, flops = {}, [], {}
buf = C.create_string_buffer(1 << 20)
for _ in range(passes):
    lib.asep_aru_profile(h, 2)
    step()
    torch.cuda.synchronize()
    _lib.check(lib.asep_aru_profile_report(h, buf, len(buf)), "report")
    for k in json.loads(buf.value.decode()):
        if k["kernel"] not in samples:
            order.append(k["kernel"]); samples[k["kernel"]] = []
        samples[k["kernel"]].append(k["total_ms"] / k["calls"])
        flops[k["kernel"]] = (k["flops"] / k["calls"], k["calls"], k.get("bytes", 0.0) / k["calls"])
lib.asep_aru_profile(h, 0)
tot = sum(statistics.median(v) * flops[k][1] for k, v in samples.items())
tot /= NP
print(f"total {tot:.3f} ms/page ({NP} page(s) per call; sum of per-layer medians over {passes} passes, {dtype}" + (f", {kw}" if kw else "") + ")")
for k in order:
    v = samples[k]
    ms = statistics.median(v); tf = flops[k][0] / (ms * 1e-3) / 1e12 if ms > 0 else 0
    gbs = flops[k][2] / (ms * 1e-3) / 1e9 if ms > 0 else 0
    print(f"{ms*1e3:9.1f} us [{min(v)*1e3:7.1f} .. {max(v)*1e3:7.1f}] {tf:7.1f} TF {gbs:7.0f} GB/s  {k}")
