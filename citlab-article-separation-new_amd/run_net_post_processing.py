"""CLI mirror of ``article_separation/image_segmentation/net_post_processing/run_net_post_processing.py``.

Same flags and defaults (``--fixed_height`` defaults to 900 for headings and 1500 for separators, threshold 0.05).
The reference fans image sub-lists out over a ``ProcessPoolExecutor(num_processes)`` of TensorFlow-CPU workers
(``gpu_devices=''``).  Here ONE process per GPU owns the device and ``--num_processes`` is the number of HOST worker
processes in total: they decode images ahead of the GPU owners and write PAGE-XML behind them (``host_pipeline.py``),
because a page costs ~110 ms of decode against single-digit milliseconds on the GPU.  Sub-lists are built exactly like
the reference's (``:64-72``) and dealt round-robin to the GPU owners.
"""
import argparse
import multiprocessing as mp
import os

from .path_util import load_list_file

MAX_SUBLIST_SIZE = 50


def run_separator(image_list, path_to_pb, fixed_height, scaling_factor, threshold, gpu_devices='0', host_workers=0):
    from .separator_net_post_processor import SeparatorNetPostProcessor
    import time
    proc = SeparatorNetPostProcessor(image_list, path_to_pb, fixed_height, scaling_factor, threshold,
                                     gpu_devices=gpu_devices, host_workers=host_workers)
    t0 = time.perf_counter()
    proc.run()
    _owner_stats(gpu_devices, len(image_list), time.perf_counter() - t0, host_workers, proc)


def _owner_stats(gpu, n_pages, seconds, host_workers, proc):
    """ASEP_OWNER_STATS_DIR=<dir>: every GPU owner leaves <dir>/owner_<pid>.json -- where its wall time went (bench.py's files-in / files-out
    legs read them; the reference has no such output, so nothing is written unless asked for)."""
    d = os.environ.get("ASEP_OWNER_STATS_DIR")
    if not d:
        return
    import json
    with open(os.path.join(d, f"owner_{os.getpid()}.json"), "w") as f:
        json.dump({"device": str(gpu), "pages": n_pages, "seconds": seconds, "host_workers": host_workers,
                   "device_seconds": getattr(proc, "device_seconds", None), "wait_seconds": getattr(proc, "wait_seconds", None),
                   "host_seconds": getattr(proc, "host_seconds", None), "first_page_seconds": getattr(proc, "first_page_seconds", None)}, f)


def run_heading(image_list, path_to_pb, fixed_height=900, scaling_factor=1, is_heading_threshold=0.4,
                weight_dict=None, thresh_dict=None, text_line_percentage=0.8, gpu_devices='0', host_workers=0):
    from .heading_net_post_processor import HeadingNetPostProcessor
    if thresh_dict is None:
        thresh_dict = {'net_thresh': 1.0, 'stroke_width_thresh': 1.0, 'text_height_thresh': 0.9, 'sw_th_thresh': 0.9}
    if weight_dict is None:
        weight_dict = {'net': 0.8, 'stroke_width': 0.0, 'text_height': 0.2}
    proc = HeadingNetPostProcessor(image_list, path_to_pb, fixed_height, scaling_factor, weight_dict, is_heading_threshold,
                                   thresh_dict, text_line_percentage)
    proc.host_workers = host_workers
    proc.run(gpu_device=gpu_devices)


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--path_to_image_list", type=str, required=True,
                        help="Path to the list file holding the image paths.")
    parser.add_argument("--path_to_pb", type=str, required=True,
                        help="Path to the pixel labelling graph (TF1 frozen .pb or .asepw).")
    parser.add_argument("--num_processes", type=int, required=False, default=8,
                        help="Host worker processes in total (image decode ahead of / PAGE-XML behind the GPU owners; "
                             "one GPU-owning process per visible device is started besides them).")
    parser.add_argument("--fixed_height", type=int, required=False, help="Input image height")
    parser.add_argument("--scaling_factor", type=float, required=False, default=1.0, help="Scaling factor of images.")
    parser.add_argument("--mode", type=str, required=True, choices=['heading', 'separator'],
                        help="Which information should be processed, e.g. headings or separator.")
    parser.add_argument("--threshold", type=float, required=False, default=0.05,
                        help="Threshold for binarization of net output.")
    return parser


def build_sub_lists(image_path_list, num_processes):
    """:64-72."""
    size_sub_lists = len(image_path_list) // num_processes
    if size_sub_lists == 0:
        size_sub_lists = 1
        num_processes = len(image_path_list)
    size_sub_lists = min(MAX_SUBLIST_SIZE, size_sub_lists)
    return [image_path_list[i: i + size_sub_lists] for i in range(0, len(image_path_list), size_sub_lists)]


def _worker(mode, sub_lists, path_to_pb, fixed_height, scaling_factor, threshold, gpu, host_workers=0):
    # one model per GPU owner: its sub-lists are processed as one stream so that the decode workers never run dry at a
    # sub-list boundary (the 50-page cap of :70 only bounded the memory of the reference's per-run image lists)
    images = [p for sub in sub_lists for p in sub]
    if mode == 'separator':
        run_separator(images, path_to_pb, fixed_height, scaling_factor, threshold, gpu_devices=str(gpu),
                      host_workers=host_workers)
    else:
        run_heading(images, path_to_pb, fixed_height, scaling_factor, 0.4, None, None, 0.8, gpu_devices=str(gpu),
                    host_workers=host_workers)


def main(argv=None):
    args = build_parser().parse_args(argv)
    mode = args.mode
    image_path_list = load_list_file(args.path_to_image_list)
    if args.fixed_height is None:
        fixed_height = 900 if mode == 'heading' else 1500
    else:
        fixed_height = args.fixed_height
    if not image_path_list:
        return 0
    sub_lists = build_sub_lists(image_path_list, args.num_processes)
    from . import _lib
    import torch
    # torch.cuda.device_count() does not initialise the GPU, so worker processes can still be spawned afterwards
    n_gpus = torch.cuda.device_count()
    if n_gpus <= 0:
        raise _lib.AsepError("no HIP device visible: the MI355X (gfx950) engine has no CPU fallback")
    # one GPU-owning process per visible device; ASEP_GPU_OWNERS="0,0,1" names the device of every owner instead (more than one
    # owner per device, or a subset of the devices -- also how the multi-owner path is tested on a one-GPU box)
    devices = list(range(n_gpus))
    if os.environ.get("ASEP_GPU_OWNERS"):
        devices = [int(x) for x in os.environ["ASEP_GPU_OWNERS"].split(",") if x.strip() != ""]
        bad = [d for d in devices if not 0 <= d < n_gpus]
        if bad or not devices:
            raise _lib.AsepError(f"ASEP_GPU_OWNERS={os.environ['ASEP_GPU_OWNERS']!r}: device ids must be in 0..{n_gpus - 1}")
    n_workers = max(1, min(len(devices), len(sub_lists)))           # GPU owners
    host_workers = max(1, args.num_processes) // n_workers            # decode / XML workers per owner (<= 1: inline)
    per_worker = [sub_lists[i::n_workers] for i in range(n_workers)]
    if n_workers == 1:
        _worker(mode, per_worker[0], args.path_to_pb, fixed_height, args.scaling_factor, args.threshold, devices[0], host_workers)
        return 0
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(mode, per_worker[i], args.path_to_pb, fixed_height,
                                               args.scaling_factor, args.threshold, devices[i], host_workers))
             for i in range(n_workers)]
    for p in procs:
        p.start()
    rc = 0
    for p in procs:
        p.join()
        rc = rc or p.exitcode
    return rc


if __name__ == '__main__':
    raise SystemExit(main())
