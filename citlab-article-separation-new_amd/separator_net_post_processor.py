"""Separator detection pipeline: image -> ARU-Net -> classical post-processing -> polygons -> PAGE-XML.

Mirror of ``article_separation/image_segmentation/net_post_processing/separator_net_post_processor.py`` and its
base class ``region_net_post_processor_base.py`` (same class / method names, argument meaning and output files:
``<dir>/page/<name>.xml.xml``).  Between the decoded image and the two binary separator masks nothing leaves the
GPU: uint8 upload -> resize + gray (a1) -> ARU-Net with fused uint8 / threshold epilogue (a3-a8) -> CC filter and
rectangular openings on bit planes (a9); only the two masks come back for polygon extraction (a10).
"""
import ctypes as C
import time

import numpy as np

from . import _lib, image_ops, polygonize
from .host_util import rescale_points
from .image_io import load_image_bgr
from .net_post_processing_helper import (AruGraph, _device_of, apply_threshold, get_net_output, get_scaling_factor,
                                         load_graph)
from .path_util import get_page_path, load_list_file
from .region_to_page_writer import SEPARATOR_REGION, SeparatorRegionToPageWriter


class RegionNetPostProcessor:
    """region_net_post_processor_base.py:17-268 (the parts the separator pipeline uses)."""

    def __init__(self, image_list, path_to_pb, fixed_height, scaling_factor, threshold=None, gpu_devices='0',
                 host_workers=0):
        self.image_paths = load_list_file(image_list) if isinstance(image_list, str) else list(image_list)
        # worker processes that decode images ahead of the GPU and write PAGE-XML behind it (host_pipeline.py);
        # 0 / 1 = everything inline in this process
        self.host_workers = host_workers
        self.device_seconds = 0.0      # wall time this process spent inside the device stages (upload .. results back)
        self.wait_seconds = 0.0        # ... waiting for the next decoded image
        self.first_page_seconds = None # run() start -> first decoded image in hand (worker start-up, slot page-locking)
        self.host_seconds = 0.0        # ... chaining polygon rings / handing the page to the writers
        self.result_seconds = 0.0      # part of device_seconds: waiting for the previous page's results
        self.fixed_height = fixed_height
        self.scaling_factor = scaling_factor
        self.threshold = threshold
        self.pb_graph = load_graph(path_to_pb)
        self.gpu_devices = gpu_devices
        # the reference keeps every image / net output of a run in memory (base:33-35); here that is opt-in
        self.keep_outputs = False
        self.images = []
        self.net_outputs = []
        self.net_outputs_post = []

    @property
    def device(self):
        return _device_of(self.gpu_devices)

    def apply_cc_analysis(self, net_output, threshold):
        return image_ops.apply_cc_analysis(net_output, threshold, self.device)

    def apply_contour_detection2(self, binary_image):
        """base:186-197: polygons (exterior + interior rings) of the regions with value 255."""
        return polygonize.shapes(binary_image, value=255, connectivity=8)

    def apply_contour_detection(self, image, use_alpha_shape=False):
        """base:163-184 without the alpha-shape option (needs shapely / Delaunay alpha complexes)."""
        if use_alpha_shape:
            raise NotImplementedError("alpha shapes are not part of this build")
        return [p[0] for p in polygonize.shapes(image, value=255, connectivity=8)]

    def rescale_polygons(self, polygons_dict, scaling_factor):
        """base:253-268: every ring times ``scaling_factor`` with int() truncation."""
        for region_name, polygon_list in polygons_dict.items():
            polygons_dict[region_name] = [[rescale_points(ring, scaling_factor) for ring in polygon]
                                          for polygon in polygon_list]
        return polygons_dict


def write_separator_page(page_path, image_path, fixed_height, scaling_factor, polygons_dict):
    """:120-133 as a plain function (runs in a WritePool worker): regions into the PAGE-XML next to the image"""
    writer = SeparatorRegionToPageWriter(page_path, image_path, fixed_height, scaling_factor, polygons_dict)
    writer.remove_separator_regions_from_page()
    writer.merge_regions()
    writer.save_page_xml(page_path + ".xml")
    return writer.page_object


class SeparatorNetPostProcessor(RegionNetPostProcessor):
    def __init__(self, image_list, path_to_pb, fixed_height, scaling_factor, threshold, gpu_devices, host_workers=0):
        super().__init__(image_list, path_to_pb, fixed_height, scaling_factor, threshold, gpu_devices, host_workers)

    def post_process(self, net_output):
        """:26-97 on a thresholded uint8 HWC net output (host array) -> {"horizontal", "vertical"}."""
        return image_ops.separator_post_process(net_output, self.device)

    def to_polygons(self, net_output, separator_type=None):
        contours = self.apply_contour_detection2(net_output)
        if separator_type is None:
            return {SEPARATOR_REGION: contours}
        return {SEPARATOR_REGION + "_" + separator_type: contours}

    def to_page_xml(self, page_path, image_path=None, polygons_dict=None, *args, **kwargs):
        return write_separator_page(page_path, image_path, self.fixed_height, self.scaling_factor, polygons_dict)

    # -- the fused device path ---------------------------------------------------------------------------------
    SEGMENT_CAPACITY = 1 << 14         # boundary segments per mask copied back without asking (a page has a few hundred)
    PAGE_LANES = 2                     # pages in flight of the pipelined run() (= streams / model instances / scratch arenas)

    def _lane_stream(self, tdev, lane):
        """lane 0 works on the caller's current stream; further lanes own a stream (and, below, a model instance and a scratch
        arena), so that pages of different lanes share the chip"""
        import torch
        if lane == 0:
            return torch.cuda.current_stream(tdev)
        streams = self.__dict__.setdefault("_lane_streams", {})
        if (tdev.index, lane) not in streams:
            streams[(tdev.index, lane)] = torch.cuda.Stream(tdev)
        return streams[(tdev.index, lane)]

    PAGE_GROUP = 4                     # pages per batched net call of the pipelined run() (asep_aru_forward_batch_dev2: any sizes)

    def enqueue_page(self, image, edges_only=True, lane=0):
        """One page = a group of one (see :meth:`enqueue_group`)."""
        return self.enqueue_group([image], edges_only=edges_only, lane=lane)[0]

    def enqueue_group(self, images, edges_only=True, lane=0):
        """Queue the device stages of up to PAGE_GROUP decoded pages -- of ANY sizes -- and return their tickets for :meth:`collect_page`.
        Uploads and resize + gray per page, then ONE batched net call for the group (round 6, asep_aru_forward_batch_dev2: the pages share every
        layer's launches, so the deep levels see 12 problems per launch like the device-resident bench instead of 3), then the classical
        stages per page.  Same arithmetic as the single-page form, bit for bit.

        Everything is only enqueued -- upload, resize + gray, ARU-Net, CC filter / openings, boundary segments, the copy of
        the segment keys into page-locked host memory -- so ``image`` has to stay valid until the ticket is collected.  Same arithmetic as load_and_scale_image ->
        get_net_output -> uint8(x*255) -> apply_threshold -> post_process (:141-151), executed without leaving HBM."""
        import torch
        dev = self.device
        lib = _lib.init_device(dev)
        tdev = torch.device("cuda", dev)
        ncls = self.pb_graph.cfg.n_classes
        _, ws = image_ops._workspace(dev, lane)
        tickets = []
        with torch.cuda.device(tdev), torch.cuda.stream(self._lane_stream(tdev, lane)):
            stream = torch.cuda.current_stream(tdev)
            sp = C.c_void_p(stream.cuda_stream)
            for image in images:
                image = np.require(image, dtype=np.uint8, requirements=['C', 'W'])   # Pillow hands out read-only views
                if image.ndim == 2:
                    image = image[:, :, None]
                H, W, Cn = image.shape
                sc = get_scaling_factor(H, W, self.scaling_factor, fixed_height=self.fixed_height)
                h, w = image_ops.scaled_size(H, W, sc)
                t = {"sc": sc, "size": (h, w), "edges_only": edges_only, "device": dev}
                # the upload is queued like everything else (a page is 0.3 ms of PCIe; a copy on a second stream ended up behind the
                # engine's kernels in a shared hardware queue and made the host wait for them): ``image`` must stay valid until the
                # page's upload has run -- DecodePool(hold=...) guarantees that for its slots, pageable arrays are staged by the runtime
                # before the call returns
                d_img = torch.empty((H, W, Cn), dtype=torch.uint8, device=tdev)
                d_img.copy_(torch.from_numpy(image), non_blocking=True)
                t["uploaded"] = torch.cuda.Event()
                t["uploaded"].record(stream)
                d_gray = torch.empty((h, w), dtype=torch.float32, device=tdev)
                _lib.check(lib.asep_prep_scale_gray_dev(ws, d_img.data_ptr(), H, W, Cn, float(sc), None,
                                                        d_gray.data_ptr(), sp), "asep_prep_scale_gray_dev")
                d_out = torch.empty((h, w, ncls), dtype=torch.float32, device=tdev)
                d_u8 = torch.empty((h, w, ncls), dtype=torch.uint8, device=tdev)
                d_mask = torch.empty((h, w, ncls), dtype=torch.uint8, device=tdev)
                t["keep"] = (d_img, d_gray, d_out, d_u8, d_mask)          # alive until the page is collected
                tickets.append(t)
            n = len(tickets)
            handle = self.pb_graph.handle(dev, lane)
            if n == 1:
                d_img, d_gray, d_out, d_u8, d_mask = tickets[0]["keep"]
                h, w = tickets[0]["size"]
                _lib.check(lib.asep_aru_forward_dev(handle, d_gray.data_ptr(), h, w, d_out.data_ptr(), d_u8.data_ptr(), d_mask.data_ptr(),
                                                    float(self.threshold), sp), "asep_aru_forward_dev")
            else:
                Arr, Ints = C.c_void_p * n, C.c_int32 * n
                keep = [t["keep"] for t in tickets]
                _lib.check(lib.asep_aru_forward_batch_dev2(
                    handle, n, Arr(*[k[1].data_ptr() for k in keep]), Ints(*[t["size"][0] for t in tickets]), Ints(*[t["size"][1] for t in tickets]),
                    Arr(*[k[2].data_ptr() for k in keep]), Arr(*[k[3].data_ptr() for k in keep]), Arr(*[k[4].data_ptr() for k in keep]),
                    float(self.threshold), sp), "asep_aru_forward_batch_dev2")
            for t in tickets:
                d_img, d_gray, d_out, d_u8, d_mask = t["keep"]
                h, w = t["size"]
                size = h * w
                min_size = int(size * (1 / size * 100))
                k_h, k_v, k_c = image_ops.separator_kernel_sizes(h, w)
                d_hz = torch.empty((h, w), dtype=torch.uint8, device=tdev)
                d_vt = torch.empty((h, w), dtype=torch.uint8, device=tdev)
                _lib.check(lib.asep_post_separator_dev(ws, d_mask.data_ptr(), h, w, ncls, 0, min_size, k_h, k_v, k_c,
                                                       d_hz.data_ptr(), d_vt.data_ptr(), sp), "asep_post_separator_dev")
                t["masks"] = {"horizontal": d_hz, "vertical": d_vt}
                if edges_only:
                    # polygon extraction needs only the boundary segments: the masks stay in HBM
                    cap = self.SEGMENT_CAPACITY
                    d_keys = torch.empty((2, 2, cap), dtype=torch.int32, device=tdev)
                    d_tot = torch.empty((2, 2), dtype=torch.int64, device=tdev)
                    for i, d_m in enumerate((d_hz, d_vt)):
                        _lib.check(lib.asep_post_boundary_segments_enqueue_dev(
                            ws, d_m.data_ptr(), h, w, 255, d_keys[i, 0].data_ptr(), d_keys[i, 1].data_ptr(), cap,
                            d_tot[i].data_ptr(), sp), "asep_post_boundary_segments_enqueue_dev")
                    t["h_keys"] = torch.empty((2, 2, cap), dtype=torch.int32, pin_memory=True)
                    t["h_tot"] = torch.empty((2, 2), dtype=torch.int64, pin_memory=True)
                    t["h_tot"].copy_(d_tot, non_blocking=True)
                    t["h_keys"].copy_(d_keys, non_blocking=True)
                else:
                    t["h_masks"] = {k: torch.empty((h, w), dtype=torch.uint8, pin_memory=True) for k in t["masks"]}
                    for k, d_m in t["masks"].items():
                        t["h_masks"][k].copy_(d_m, non_blocking=True)
                if self.keep_outputs:
                    t["h_u8"] = torch.empty((h, w, ncls), dtype=torch.uint8, pin_memory=True)
                    t["h_u8"].copy_(d_u8, non_blocking=True)
                t["done"] = torch.cuda.Event()
                t["done"].record(stream)
        return tickets

    def collect_page(self, t):
        """wait for a ticket of :meth:`enqueue_page` -> ({"horizontal", "vertical"}, sc, extras): uint8 [h,w] masks, or with
        ``edges_only`` the (starts, ends) segment keys of ``asep_post_boundary_segments``"""
        import torch
        t_res = time.perf_counter()
        t["done"].synchronize()
        self.result_seconds += time.perf_counter() - t_res
        h, w = t["size"]
        if t["edges_only"]:
            masks = {}
            tot = t["h_tot"].numpy()
            for i, name in enumerate(("horizontal", "vertical")):
                n_s, n_e = int(tot[i, 0]), int(tot[i, 1])
                if n_s != n_e:
                    raise _lib.AsepError(f"boundary segments: {n_s} starts but {n_e} ends")
                if n_s > self.SEGMENT_CAPACITY:               # an unusually ragged mask: ask again with room for all of it
                    tdev = torch.device("cuda", t["device"])
                    with torch.cuda.device(tdev):
                        sp = C.c_void_p(torch.cuda.current_stream(tdev).cuda_stream)
                        masks[name] = image_ops.boundary_segments_dev(t["masks"][name].data_ptr(), h, w, 255, t["device"],
                                                                      sp, capacity=n_s)
                else:
                    keys = t["h_keys"][i, :, :n_s].numpy()
                    masks[name] = (keys[0].copy(), keys[1].copy())
        else:
            masks = {k: v.numpy().copy() for k, v in t["h_masks"].items()}
        extras = {"net_output_u8": t["h_u8"].numpy().copy()} if "h_u8" in t else {}
        extras["size"] = (h, w)
        sc = t["sc"]
        t.clear()
        return masks, sc, extras

    def separator_masks(self, image, edges_only=False):
        """decoded image (uint8 [H,W,3] BGR or [H,W]) -> ({"horizontal", "vertical"} uint8 [h,w], sc, extras)"""
        return self.collect_page(self.enqueue_page(image, edges_only=edges_only))

    def _finish_page(self, image_path, ticket, writers, pipelined, page_objects):
        t_dev = time.perf_counter()
        masks, sc, extras = self.collect_page(ticket)
        t_host = time.perf_counter()
        self.device_seconds += t_host - t_dev
        polygons_dict = {}
        if self.keep_outputs:
            self.net_outputs.append(extras["net_output_u8"])
            self.net_outputs_post.append(masks)
            for separator_type, net_output_post in masks.items():
                polygons_dict.update(self.to_polygons(net_output_post, separator_type))
        else:
            h, w = extras["size"]
            for separator_type, (starts, ends) in masks.items():
                polygons_dict[SEPARATOR_REGION + "_" + separator_type] = \
                    polygonize.shapes_from_segments(starts, ends, h, w, connectivity=8)
        polygons_dict = self.rescale_polygons(polygons_dict, scaling_factor=1 / sc)
        if pipelined:
            writers.submit(write_separator_page, get_page_path(image_path), image_path, self.fixed_height,
                           self.scaling_factor, polygons_dict)
        else:
            page_objects.append(self.to_page_xml(get_page_path(image_path), image_path=image_path,
                                                 polygons_dict=polygons_dict))
        self.host_seconds += time.perf_counter() - t_host

    def run(self):
        """:135-159.  With ``host_workers`` > 1 the images are decoded ahead of the GPU by worker processes (DMA-able
        shared-memory slots) and the PAGE-XML files are written behind it; the GPU-owning process only runs the device
        stages and chains the polygon rings -- PAGE_LANES - 1 pages behind the GPU: the next pages are uploaded and queued
        before a page's segments are waited for, so the chip does not idle while the host chains rings (``device_seconds`` =
        queueing + waiting for results, ``host_seconds`` = chaining and handing over)."""
        from .host_pipeline import DecodePool, WritePool, pin_callbacks
        page_objects = []
        pipelined = self.host_workers > 1 and not self.keep_outputs
        reg, unreg = pin_callbacks(self.device) if pipelined else (None, None)
        group = self.PAGE_GROUP if pipelined else 1
        decode = DecodePool(self.image_paths, self.host_workers if pipelined else 0, register=reg, unregister=unreg, hold=group + 1)
        with WritePool(self.host_workers if pipelined else 0) as writers:
            t_prev = t_run = time.perf_counter()
            pending, n_groups = [], 0
            lanes = self.PAGE_LANES if pipelined else 1
            batch = []

            def flush():
                """the decoded pages waiting in ``batch`` as ONE group on the next lane; a lane's previous group has been collected before"""
                nonlocal n_groups, t_prev
                t_dev = time.perf_counter()
                tickets = self.enqueue_group([img for _, img in batch], edges_only=not self.keep_outputs, lane=n_groups % lanes)
                n_groups += 1
                self.device_seconds += time.perf_counter() - t_dev
                pending.extend((path, t) for (path, _), t in zip(batch, tickets))
                while len(pending) > max(1, (lanes - 1) * group):    # (the newest group stays in flight while older pages' rings are chained)
                    self._finish_page(*pending.pop(0), writers, pipelined, page_objects)
                tickets[-1]["uploaded"].synchronize()        # (long done) the images' slots may be recycled from here on
                batch.clear()
                t_prev = time.perf_counter()

            n_paths = len(self.image_paths)
            for n_seen, (image_path, image) in enumerate(decode, 1):
                t_dev = time.perf_counter()
                if self.first_page_seconds is None:          # worker start-up + slot page-locking + the first decode
                    self.first_page_seconds = t_dev - t_run
                self.wait_seconds += t_dev - t_prev
                # pipelined: PAGE_GROUP consecutive pages per batched net call, consecutive groups on alternating lanes (streams, model
                # instances, scratch arenas): the chip works on the next group's nets while a group's classical stages -- small kernels that do
                # not fill it -- run, and while this process chains rings
                batch.append((image_path, image))
                if len(batch) >= group or n_seen == n_paths:   # (the last pages are uploaded HERE: the pool releases its slots when it ends)
                    flush()
                else:
                    t_prev = time.perf_counter()
            assert not batch
            for item in pending:
                self._finish_page(*item, writers, pipelined, page_objects)
        return page_objects


def build_parser():
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument('--image_list', type=str, required=False)
    parser.add_argument('--path_to_pb', type=str, required=False)
    parser.add_argument('--fixed_height', type=int, required=False, default=1500)
    parser.add_argument('--scaling_factor', type=float, required=False, default=1.0)
    parser.add_argument('--threshold', type=float, required=False, default=0.05)
    parser.add_argument('--gpu_devices', type=str, required=False, default='')
    return parser


if __name__ == '__main__':
    args = build_parser().parse_args()
    SeparatorNetPostProcessor(args.image_list, args.path_to_pb, args.fixed_height, args.scaling_factor,
                              args.threshold, args.gpu_devices).run()
