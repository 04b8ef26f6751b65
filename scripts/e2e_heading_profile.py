"""cProfile of the GPU owner of the heading command line (files in, files out) on synthetic scans with ~700 text lines per page:
where does the owner's time per page go?      python scripts/e2e_heading_profile.py [n_pages=96] [host_workers=24] [fixed_height=4500]"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image

n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 96
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 24
fixed_height = int(sys.argv[3]) if len(sys.argv) > 3 else 4500
W, H = 3000, 4500


def main():
    from citlab_article_separation_new_amd import net_post_processing_helper as helper, synth
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.heading_net_post_processor import HeadingNetPostProcessor
    from citlab_article_separation_new_amd.weights import init_aru_weights
    cfg = AruConfig()
    graph = helper.AruGraph(init_aru_weights(cfg, 21, logit_scale=0.05), cfg)
    with tempfile.TemporaryDirectory(prefix="asep_hprof_") as tmp:
        os.makedirs(os.path.join(tmp, "page"))
        paths = []
        for k in range(n_pages):
            p = os.path.join(tmp, f"p{k:03d}.png")
            if k < 4:
                Image.fromarray(synth.cached_synth_page(k, W, H)).save(p, compress_level=1)
            else:
                os.symlink(os.path.join(tmp, f"p{k % 4:03d}.png"), p)
            synth.synth_page_xml(os.path.join(tmp, "page", f"p{k:03d}.xml"), W, H, k % 4)
            paths.append(p)
        wd = {'net': 0.8, 'stroke_width': 0.0, 'text_height': 0.2}
        td = {'net_thresh': 1.0, 'stroke_width_thresh': 1.0, 'text_height_thresh': 0.9, 'sw_th_thresh': 0.9}
        hp = HeadingNetPostProcessor(paths[:2], graph, fixed_height, 1.0, wd, 0.4, td, 0.8)
        hp.host_workers = 0
        hp.run(gpu_device="0")
        hp = HeadingNetPostProcessor(paths, graph, fixed_height, 1.0, wd, 0.4, td, 0.8)
        hp.host_workers = workers
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        hp.run(gpu_device="0")
        pr.disable()
        dt = time.perf_counter() - t0
        print(f"{n_pages / dt:.1f} pages/s ({dt / n_pages * 1e3:.1f} ms/page under the profiler)")
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
