"""The heading owner's device stages without files around them (decoded pages + line boxes in memory -> enqueue_page /
collect_boxes on two lanes): what the chip needs per page.      python scripts/heading_owner_probe.py [n_pages=48]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from citlab_article_separation_new_amd import net_post_processing_helper as helper, synth
from citlab_article_separation_new_amd.config import AruConfig
from citlab_article_separation_new_amd.heading_net_post_processor import HeadingNetPostProcessor, read_line_geometry
from citlab_article_separation_new_amd.weights import init_aru_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
W, H = 3000, 4500
cfg = AruConfig(compute_dtype=os.environ.get("ASEP_COMPUTE_DTYPE", "f32"))
graph = helper.AruGraph(init_aru_weights(cfg, 21, logit_scale=0.05), cfg)
pages = [torch.from_numpy(synth.cached_synth_page(k, W, H)).pin_memory().numpy() for k in range(4)]
with tempfile.TemporaryDirectory() as tmp:
    synth.synth_page_xml(os.path.join(tmp, "p.xml"), W, H, 0)
    lines = read_line_geometry(os.path.join(tmp, "p.xml"))
hp = HeadingNetPostProcessor([], graph, H, 1.0, {'net': 0.8, 'stroke_width': 0.0, 'text_height': 0.2}, 0.4,
                             {'net_thresh': 1.0, 'stroke_width_thresh': 1.0, 'text_height_thresh': 0.9, 'sw_th_thresh': 0.9}, 0.8)
hp.gpu_devices = "0"
for lanes in (1, 2):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pending = []
        for k in range(n):
            pending.append(hp.enqueue_page(pages[k % 4], lane=k % lanes))
            if len(pending) > 2:
                hp.collect_boxes(pending.pop(0), *lines)
        for t in pending:
            hp.collect_boxes(t, *lines)
        dt = time.perf_counter() - t0
    print(f"heading owner, lanes {lanes}, {len(lines[0])} lines per page: {dt / n * 1e3:6.2f} ms/page = {n / dt:6.1f} pages/s")
