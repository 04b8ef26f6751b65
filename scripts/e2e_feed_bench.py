"""Files in, files out on ONE GPU: how fast does `run_net_post_processing --mode separator` turn scan files into PAGE-XML
when worker processes decode ahead of / write behind the GPU owner (host_pipeline.py), and how busy is the GPU?

    python scripts/e2e_feed_bench.py [n_pages=32] [host_workers=12] [fixed_height=4500]

Prints pages/s and the GPU-busy fraction (time inside the device stages / wall) for host_workers = 0 (everything
inline, the round-1 behaviour) and for the requested number of workers; also the seam cost of `get_net_output` on host
arrays (page-locked transfers) and the batched small-page rate (CLI default size 1000 x 1500)."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image

n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 192
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 12
fixed_height = int(sys.argv[3]) if len(sys.argv) > 3 else 4500
W, H = 3000, 4500


def main():
    import ctypes as C
    import torch
    from citlab_article_separation_new_amd import _lib, net_post_processing_helper as helper, synth
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.separator_net_post_processor import SeparatorNetPostProcessor
    from citlab_article_separation_new_amd.weights import init_aru_weights
    cfg = AruConfig()
    graph = helper.AruGraph(init_aru_weights(cfg, 21, logit_scale=0.05), cfg)
    with tempfile.TemporaryDirectory(prefix="asep_feed_") as tmp:
        os.makedirs(os.path.join(tmp, "page"))
        base = [synth.synth_page(k, W, H) for k in range(4)]
        paths = []
        for k in range(n_pages):                       # four real files, the rest are links to them (own PAGE-XML each)
            p = os.path.join(tmp, f"p{k:03d}.png")
            if k < 4:
                Image.fromarray(base[k]).save(p, compress_level=1)
            else:
                os.symlink(os.path.join(tmp, f"p{k % 4:03d}.png"), p)
            paths.append(p)
        t0 = time.perf_counter()
        from citlab_article_separation_new_amd import image_io
        for p in paths[:4]:
            image_io.load_image_bgr(p)
        print(f"PNG decode alone: {(time.perf_counter() - t0) / 4 * 1e3:.1f} ms/page (one process)")
        for hw in (0, workers):
            proc = SeparatorNetPostProcessor(paths[:2], graph, fixed_height, 1.0, 0.5, "0", host_workers=0)
            proc.run()                                   # warm-up: library, pools, first-touch allocations
            proc = SeparatorNetPostProcessor(paths if hw else paths[:24], graph, fixed_height, 1.0, 0.5, "0", host_workers=hw)
            t0 = time.perf_counter()
            proc.run()
            dt = time.perf_counter() - t0
            n_xml = len([f for f in os.listdir(os.path.join(tmp, "page")) if f.endswith(".xml.xml")])
            n_run = len(proc.image_paths)
            print(f"separator CLI path, fixed_height {fixed_height}, host_workers {hw:2d}: {n_run / dt:6.2f} pages/s "
                  f"({dt / n_run * 1e3:6.1f} ms/page incl. worker start-up); GPU owner: device stages {proc.device_seconds / dt:5.1%} "
                  f"({proc.device_seconds / n_run * 1e3:.1f} ms/page), waiting for decoded images {proc.wait_seconds / dt:5.1%}, "
                  f"rings + hand-over {proc.host_seconds / dt:5.1%}; {n_xml} PAGE-XML files written; per page: result wait {proc.result_seconds / n_run * 1e3:.2f} ms, first page after "
                  f"{proc.first_page_seconds:.2f} s")
        # ---- heading mode: needs PAGE-XML with text lines; parsing and writing happen in workers, the owner measures ----
        from citlab_article_separation_new_amd.heading_net_post_processor import HeadingNetPostProcessor
        rng = np.random.default_rng(0)
        colw = (W - 120 - 5 * 40) // 6
        for k in range(n_pages):
            regs, rid = [], 0
            for c in range(6):
                x0, y = 60 + c * (colw + 40), 60
                while y < H - 300:
                    nl, pitch = int(rng.integers(4, 13)), int(rng.integers(28, 37))
                    y1 = y + nl * pitch
                    lines = "".join(
                        f'<TextLine id="r{rid}l{i}"><Coords points="{x0},{y + i * pitch} {x0 + colw},{y + i * pitch} '
                        f'{x0 + colw},{y + (i + 1) * pitch - 4} {x0},{y + (i + 1) * pitch - 4}"/>'
                        f'<Baseline points="{x0},{y + (i + 1) * pitch - 8} {x0 + colw},{y + (i + 1) * pitch - 8}"/></TextLine>'
                        for i in range(nl))
                    regs.append(f'<TextRegion id="r{rid}"><Coords points="{x0},{y} {x0 + colw},{y} {x0 + colw},{y1} {x0},{y1}"/>'
                                + lines + '</TextRegion>')
                    rid += 1
                    y = y1 + int(rng.integers(20, 60))
            with open(os.path.join(tmp, "page", f"p{k:03d}.xml"), "w") as f:
                f.write('<?xml version="1.0" encoding="UTF-8"?>\n<PcGts xmlns="http://schema.primaresearch.org/PAGE/gts/'
                        'pagecontent/2013-07-15"><Metadata><Creator>t</Creator><Created>2020-01-01T00:00:00</Created>'
                        '<LastChange>2020-01-01T00:00:00</LastChange></Metadata>'
                        f'<Page imageFilename="x.png" imageWidth="{W}" imageHeight="{H}">' + "".join(regs) + '</Page></PcGts>')
        wd = {'net': 0.8, 'stroke_width': 0.0, 'text_height': 0.2}
        td = {'net_thresh': 1.0, 'stroke_width_thresh': 1.0, 'text_height_thresh': 0.9, 'sw_th_thresh': 0.9}
        for hw in (0, workers):
            proc = HeadingNetPostProcessor(paths if hw else paths[:16], graph, 900, 1.0, wd, 0.4, td, 0.8)
            proc.host_workers = hw
            t0 = time.perf_counter()
            proc.run(gpu_device="0")
            dt = time.perf_counter() - t0
            n_run = len(proc.image_paths)
            print(f"heading CLI path, fixed_height 900 (default), host_workers {hw:2d}: {n_run / dt:6.2f} pages/s "
                  f"({dt / n_run * 1e3:6.1f} ms/page incl. worker start-up)")
        # ---- seam: get_net_output on host arrays (float64 in, float32 out), page-locked staging ----
        img = (base[0] / 255.0)
        helper.get_net_output(img, graph, "0")
        t0 = time.perf_counter()
        for _ in range(5):
            out = helper.get_net_output(img, graph, "0")
        print(f"get_net_output(float64 {W}x{H}) -> float32 [{H},{W},2]: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms/page "
              f"(device-resident forward alone: see bench.py)")
        img32 = img.astype(np.float32)
        t0 = time.perf_counter()
        for _ in range(5):
            out = helper.get_net_output(img32, graph, "0")
        print(f"get_net_output(float32 {W}x{H}): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms/page")
        # ---- batched small pages: the CLI default net input 1000 x 1500 ----
        lib = _lib.init_device(0)
        h = graph.handle(0)
        for hh, ww in ((1500, 1000), (900, 600), (768, 512)):
            for B in (1, 16):
                ins = [torch.rand(hh, ww, device="cuda") for _ in range(B)]
                outs = [torch.empty(hh, ww, 2, device="cuda") for _ in range(B)]
                Arr = C.c_void_p * B
                pi, po = Arr(*[t.data_ptr() for t in ins]), Arr(*[t.data_ptr() for t in outs])
                s = torch.cuda.current_stream().cuda_stream
                step = lambda: _lib.check(lib.asep_aru_forward_batch_dev(h, B, pi, hh, ww, po, None, None, 0.05, s), "batch")
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    step()
                torch.cuda.synchronize()
                print(f"net input {ww}x{hh}, batch {B:2d}: {(time.perf_counter() - t0) / 20 / B * 1e3:.3f} ms/page")


if __name__ == "__main__":
    main()
