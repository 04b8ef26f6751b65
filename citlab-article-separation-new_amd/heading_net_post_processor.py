"""Heading detection pipeline: image -> ARU-Net heading probability + stroke-width transform -> PAGE-XML tags.

Mirror of ``article_separation/image_segmentation/net_post_processing/heading_net_post_processor.py`` (same class /
method names, fusion rule and output file ``<dir>/page/<name>.xml.xml``) and of
``python_util/image_processing/swt_dist_trafo.py``.  The net and the stroke-width distance transform (Gaussian ->
Otsu -> exact Euclidean distance transform on the full-resolution scan) run on the GPU; the per-text-line
statistics are host numpy on small crops, evaluated with the reference's own numpy expressions so that the
threshold comparisons see identical doubles.
"""
import ctypes as C
from collections import Counter

import numpy as np
from scipy import ndimage

from . import _lib, image_ops
from .host_util import rescale_points
from .image_io import load_image_bgr
from .net_post_processing_helper import get_scaling_factor
from .path_util import get_page_path
from .region_to_page_writer import RegionToPageWriter
from .separator_net_post_processor import RegionNetPostProcessor

HEADING = "heading"          # page_constants.py:58-59 TextRegionTypes
PARAGRAPH = "paragraph"
_EIGHT = np.ones((3, 3), dtype=bool)


def bgr_to_gray_u8(img):
    """cv2.imread(path, IMREAD_GRAYSCALE) of an already decoded BGR image (OpenCV 4.x fixed-point weights)."""
    if img.ndim == 2:
        return img
    b, g, r = (img[:, :, i].astype(np.int32) for i in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)


class StrokeWidthDistanceTransform:
    """swt_dist_trafo.py:5-66."""

    def __init__(self, dark_on_bright=True, clean_ccs=2, device=0):
        if not dark_on_bright:
            raise NotImplementedError("only dark text on bright paper (the reference's only use)")
        self._clean_ccs = clean_ccs
        self.device = device

    def distance_transform(self, img_or_path, on_device=False):
        """:18-24 -- accepts a file path like the reference, or an already decoded gray / BGR uint8 array.
        ``on_device`` leaves the result in HBM (``image_ops.DeviceImage``) for the batched per-line statistics."""
        if isinstance(img_or_path, str):
            img_or_path = load_image_bgr(img_or_path)
        gray = bgr_to_gray_u8(np.asarray(img_or_path))
        if on_device:
            return image_ops.swt_distance_transform_device(gray, self.device)
        return image_ops.swt_distance_transform(gray, self.device)

    def connected_components_cv(self, image, connectivity=8):
        """:31-41: (x, y, w, h) of every connected component of non-zero pixels."""
        assert connectivity in (4, 8), f"Connectivity has to be 4 or 8 (was {connectivity})."
        lab, _ = ndimage.label(np.asarray(image) != 0, structure=_EIGHT if connectivity == 8 else None)
        return [(sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start)
                for sl in ndimage.find_objects(lab)]

    def clean_connected_components(self, components):
        """:43-66."""
        out = []
        for component in components:
            width, height = component[2], component[3]
            if self._clean_ccs > 0 and (width < 3 or height < 3 or height > 500 or width > 500):
                continue
            if self._clean_ccs > 1 and (width / height > 8 or height / width > 8):
                continue
            out.append(component)
        return out


def _scale_to_new_interval(data, old_min, old_max, new_min=0, new_max=1):
    """:50-63."""
    if old_max - old_min == 0:
        return data
    return (new_max - new_min) / (old_max - old_min) * (data - old_min) + new_min


def apply_heading_values(writer, text_lines, values, weight_dict, threshold, thresh_dict, text_line_percentage, save_path):
    """:121-200: fusion rule on the per-line measurements, heading tags on lines and regions, PAGE-XML written."""
    stroke_width_dict, height_dict, net_prob_dict = (dict(v) for v in values)
    page_object = writer.page_object
    stroke_width_list = list(stroke_width_dict.values())
    use_swt_features = len(stroke_width_list) > 0
    if use_swt_features:
        stroke_width_mode = Counter(stroke_width_list).most_common(1)[0][0]
        height_mode = Counter(list(height_dict.values())).most_common(1)[0][0]
        for text_line in text_lines:
            stroke_width_dict[text_line.id] = stroke_width_dict[text_line.id] - stroke_width_mode
            height_dict[text_line.id] = height_dict[text_line.id] - height_mode
        stroke_width_list = list(stroke_width_dict.values())
        stroke_width_min, stroke_width_max = np.min(stroke_width_list), np.max(stroke_width_list)
        height_list = list(height_dict.values())
        height_min, height_max = np.min(height_list), np.max(height_list)
        net_weight = weight_dict["net"]
        stroke_width_weight = weight_dict["stroke_width"]
        height_weight = weight_dict["text_height"]
        net_thresh = thresh_dict["net_thresh"]
        stroke_width_thresh = thresh_dict["stroke_width_thresh"]
        height_thresh = thresh_dict["text_height_thresh"]
        sw_th_thresh = thresh_dict["sw_th_thresh"]

    for text_line in text_lines:
        net_conf = net_prob_dict[text_line.id]
        if use_swt_features:
            sw_conf = _scale_to_new_interval(stroke_width_dict[text_line.id], old_min=stroke_width_min,
                                             old_max=stroke_width_max)
            th_conf = _scale_to_new_interval(height_dict[text_line.id], old_min=height_min, old_max=height_max)
            if sw_conf >= stroke_width_thresh or th_conf >= height_thresh or \
                    (sw_conf + th_conf) / 2 >= sw_th_thresh or net_conf >= net_thresh:
                is_heading_confidence = 1.0
            else:
                is_heading_confidence = net_weight * net_conf + stroke_width_weight * sw_conf \
                    + height_weight * th_conf
        else:
            is_heading_confidence = net_conf
        if is_heading_confidence > threshold:
            text_line.set_structure_attribute("semantic_type", HEADING)
            text_line.flush()

    for text_region in page_object.get_text_regions():
        text_region.region_type = PARAGRAPH
        if text_region.text_lines:
            num_headings = sum(1 for tl in text_region.text_lines if tl.get_semantic_type() == HEADING)
            if num_headings / len(text_region.text_lines) >= text_line_percentage:
                text_region.region_type = HEADING
        text_region.node.set("type", text_region.region_type)

    writer.save_page_xml(save_path)
    return page_object


class LineGeometry:
    """what the measurements need of a text line (picklable: travels from a worker that parsed the PAGE-XML)"""
    __slots__ = ("id", "surr_p")

    def __init__(self, line_id, surr_p):
        self.id, self.surr_p = line_id, surr_p

    def get_bounding_box(self):
        xs = [p[0] for p in self.surr_p]
        ys = [p[1] for p in self.surr_p]
        return min(xs), min(ys), max(xs) - min(xs) + 1, max(ys) - min(ys) + 1


def line_boxes(text_lines):
    """(ids, bounding boxes int64 [L,4] = xmin, ymin, xmax, ymax of each outline, has-outline bool [L]) of text lines in order:
    all the per-line measurements need (outline coordinates are integers, page_xml.parse_points)"""
    ids = [tl.id for tl in text_lines]
    boxes = np.zeros((len(ids), 4), dtype=np.int64)
    has = np.zeros(len(ids), dtype=bool)
    for i, tl in enumerate(text_lines):
        if tl.surr_p:
            xs = [p[0] for p in tl.surr_p]
            ys = [p[1] for p in tl.surr_p]
            boxes[i] = (min(xs), min(ys), max(xs), max(ys))
            has[i] = True
    return ids, boxes, has


def read_line_geometry(page_path):
    """worker task: ``line_boxes`` of every text line of a PAGE-XML in document order, or None if the file is missing"""
    import os
    from .page_xml import Page
    if not os.path.exists(page_path):
        return None
    return line_boxes(Page(page_path).get_textlines())


def _slice_bounds(start, length, size):
    """what numpy makes of ``a[start:start + length]`` on an axis of ``size`` (negative indices count from the end): the bounds
    of ``slice(start, start + length).indices(size)`` for int64 arrays -> (first, last + 1 >= first)"""
    stop = start + length
    first = np.where(start < 0, np.maximum(start + size, 0), np.minimum(start, size))
    last = np.where(stop < 0, np.maximum(stop + size, 0), np.minimum(stop, size))
    return first, np.maximum(last, first)


def write_heading_page(page_path, image_path, fixed_height, scaling_factor, values, weight_dict, threshold, thresh_dict,
                       text_line_percentage):
    """worker task: fusion + tags + PAGE-XML of one page from the measurements the GPU owner made"""
    writer = RegionToPageWriter(page_path, path_to_image=image_path, fixed_height=fixed_height, scaling_factor=scaling_factor)
    text_lines = writer.page_object.get_textlines()
    apply_heading_values(writer, text_lines, values, weight_dict, threshold, thresh_dict, text_line_percentage, page_path + ".xml")


class HeadingNetPostProcessor(RegionNetPostProcessor):
    def __init__(self, image_list, path_to_pb, fixed_height, scaling_factor, weight_dict=None, threshold=0.5,
                 thresh_dict=None, text_line_percentage=None):
        super().__init__(image_list, path_to_pb, fixed_height, scaling_factor)
        self.SWT = StrokeWidthDistanceTransform(dark_on_bright=True)
        self.weight_dict = weight_dict if weight_dict is not None else {"net": 0.33, "stroke_width": 0.33,
                                                                        "text_height": 0.33}
        self.threshold = threshold
        self.thresh_dict = thresh_dict if thresh_dict is not None else {"net_thresh": 0.9, "stroke_width_thresh": 0.9,
                                                                        "text_height_thresh": 0.9,
                                                                        "sw_th_thresh": 0.8}
        self.text_line_percentage = text_line_percentage if text_line_percentage is not None else 1.0

    def scale_to_new_interval(self, data, old_min, old_max, new_min=0, new_max=1):
        """:50-63."""
        return _scale_to_new_interval(data, old_min, old_max, new_min, new_max)

    def post_process(self, net_output):
        """:202-208."""
        return net_output[:, :, 0] / 255

    def get_swt_features_image(self, image_path):
        return self.SWT.distance_transform(image_path)

    def get_swt_features_textline(self, swt_feature_image, text_line):
        """:218-245."""
        x, y, w, h = text_line.get_bounding_box()
        xa, xb = x, x + w
        ya, yb = y, y + h
        text_line_swt = swt_feature_image[ya:yb + 1, xa:xb + 1]
        text_line_ccs = self.SWT.clean_connected_components(self.SWT.connected_components_cv(text_line_swt))
        swt_cc_values = []
        text_line_height = 0
        for cc in text_line_ccs:
            swt_cc_values.append(np.max(text_line_swt[cc[1]: cc[1] + cc[3], cc[0]: cc[0] + cc[2]]))
            if cc[3] > text_line_height:
                text_line_height = cc[3]
        text_line_stroke_width = np.median(swt_cc_values) if swt_cc_values else 0.0
        return text_line_stroke_width, text_line_height

    def get_net_prob_for_text_line(self, net_output, text_line, scaling_factor):
        """:247-270: mean net confidence over the rescaled bounding box (divided by its nominal size)."""
        if not text_line.surr_p:
            return 0
        pts = rescale_points(text_line.surr_p, scaling_factor)
        xs = [p[0] for p in pts]
        ys = [p[1] for p in pts]
        xa, ya = min(xs), min(ys)
        width, height = max(xs) - xa + 1, max(ys) - ya + 1
        net_output_text_line = net_output[ya:ya + height, xa:xa + width]
        return np.sum(net_output_text_line) / (width * height)

    def line_values(self, text_lines, scaling_factor, net_output_post, swt_feature_image):
        """The three per-line measurements of :94-119 for ``text_lines`` (objects with ``id``, ``surr_p`` and
        ``get_bounding_box``): stroke width, text height, mean net confidence -> three dicts keyed by line id."""
        stroke_width_dict, height_dict, net_prob_dict = {}, {}, {}
        batched = {}
        if isinstance(swt_feature_image, image_ops.DeviceImage):
            # all lines of the page in one kernel launch on the device-resident distance transform
            with_coords = [tl for tl in text_lines if tl.surr_p]
            boxes = []
            for tl in with_coords:
                x, y, w, h = tl.get_bounding_box()
                boxes.append([x, y, x + w + 1, y + h + 1])         # the crop [ya:yb+1, xa:xb+1] of :232-236
            sws, hts = image_ops.swt_line_features(swt_feature_image, boxes, device=swt_feature_image.device)
            batched = {tl.id: (sws[i], int(hts[i])) for i, tl in enumerate(with_coords)}
        for text_line in text_lines:
            if not text_line.surr_p:
                stroke_width, height = 0, 0
            elif batched:
                stroke_width, height = batched[text_line.id]
            else:
                stroke_width, height = self.get_swt_features_textline(swt_feature_image, text_line)
            stroke_width_dict[text_line.id] = stroke_width
            height_dict[text_line.id] = height
            if self.weight_dict['net'] == 0 or net_output_post is None:
                net_prob_dict[text_line.id] = 0
            else:
                net_prob_dict[text_line.id] = self.get_net_prob_for_text_line(net_output_post, text_line, scaling_factor)
        return stroke_width_dict, height_dict, net_prob_dict

    def to_page_xml(self, page_path, image_path=None, net_output_post=None, swt_feature_image=None, *args, **kwargs):
        """:66-200."""
        writer = RegionToPageWriter(page_path, path_to_image=image_path, fixed_height=self.fixed_height,
                                    scaling_factor=self.scaling_factor)
        if swt_feature_image is None:
            swt_feature_image = self.get_swt_features_image(image_path)
        text_lines = writer.page_object.get_textlines()
        values = self.line_values(text_lines, writer.scaling_factor, net_output_post, swt_feature_image)
        return apply_heading_values(writer, text_lines, values, self.weight_dict, self.threshold, self.thresh_dict,
                                    self.text_line_percentage, page_path + ".xml")

    def heading_probability(self, image):
        """decoded image -> uint8 net output [h,w,n_cls] at the scaled size (:285-288), device resident in between."""
        import torch
        dev = self.device
        lib = _lib.init_device(dev)
        tdev = torch.device("cuda", dev)
        image = np.require(image, dtype=np.uint8, requirements=['C', 'W'])   # Pillow hands out read-only views
        if image.ndim == 2:
            image = image[:, :, None]
        H, W, Cn = image.shape
        sc = get_scaling_factor(H, W, self.scaling_factor, fixed_height=self.fixed_height)
        h, w = image_ops.scaled_size(H, W, sc)
        ncls = self.pb_graph.cfg.n_classes
        _, ws = image_ops._workspace(dev)
        with torch.cuda.device(tdev):
            sp = C.c_void_p(torch.cuda.current_stream(tdev).cuda_stream)
            d_img = torch.from_numpy(image).to(tdev)
            d_gray = torch.empty((h, w), dtype=torch.float32, device=tdev)
            _lib.check(lib.asep_prep_scale_gray_dev(ws, d_img.data_ptr(), H, W, Cn, float(sc), None,
                                                    d_gray.data_ptr(), sp), "asep_prep_scale_gray_dev")
            d_out = torch.empty((h, w, ncls), dtype=torch.float32, device=tdev)
            d_u8 = torch.empty((h, w, ncls), dtype=torch.uint8, device=tdev)
            _lib.check(lib.asep_aru_forward_dev(self.pb_graph.handle(dev), d_gray.data_ptr(), h, w, d_out.data_ptr(),
                                                d_u8.data_ptr(), None, 0.0, sp), "asep_aru_forward_dev")
            return d_u8.cpu().numpy()

    PAGE_LANES = 2                     # lanes of the pipelined run(): stream, model instance and scratch arena each

    def _lane_stream(self, tdev, lane):
        import torch
        if lane == 0:
            return torch.cuda.current_stream(tdev)
        streams = self.__dict__.setdefault("_lane_streams", {})
        if (tdev.index, lane) not in streams:
            streams[(tdev.index, lane)] = torch.cuda.Stream(tdev)
        return streams[(tdev.index, lane)]

    PAGE_GROUP = 4                     # pages per batched net call of the pipelined run() (asep_aru_forward_batch_dev2: any sizes)

    def enqueue_page(self, image, lane=0):
        """One page = a group of one (see :meth:`enqueue_group`)."""
        return self.enqueue_group([image], lane=lane)[0]

    def enqueue_group(self, images, lane=0):
        """Queue the device stages of up to PAGE_GROUP decoded pages of any sizes (``images`` stay valid until their uploads have run) -- uploads,
        resize + gray per page, ONE batched heading-net call for the group with uint8 epilogue (:285-288; round 6: the pages share every layer's
        launches), then per page full-size gray + stroke-width distance transform (swt_dist_trafo.py:18-29) -- and return the tickets for
        :meth:`collect_page`.  Neither the net output nor the distance transform leaves HBM."""
        import torch
        dev = self.device
        lib = _lib.init_device(dev)
        tdev = torch.device("cuda", dev)
        ncls = self.pb_graph.cfg.n_classes
        _, ws = image_ops._workspace(dev, 0 if lane == 0 else 10 + lane)     # (arena 1 belongs to collect_boxes' side stream)
        tickets = []
        with torch.cuda.device(tdev), torch.cuda.stream(self._lane_stream(tdev, lane)):
            stream = torch.cuda.current_stream(tdev)
            sp = C.c_void_p(stream.cuda_stream)
            if getattr(self, "_side_stream", None) is None or self._side_stream.device != tdev:
                self._side_stream = torch.cuda.Stream(tdev)
            use_net = self.weight_dict['net'] > 0
            for image in images:
                image = np.require(image, dtype=np.uint8, requirements=['C', 'W'])   # Pillow hands out read-only views
                if image.ndim == 2:
                    image = image[:, :, None]
                H, W, Cn = image.shape
                sc = get_scaling_factor(H, W, self.scaling_factor, fixed_height=self.fixed_height)
                h, w = image_ops.scaled_size(H, W, sc)
                t = {"sc": sc, "size": (h, w, ncls), "device": dev, "full": (H, W, Cn)}
                # the upload is queued like everything else (a page is 0.3 ms of PCIe; a copy on a second stream ended up behind the
                # engine's kernels in a shared hardware queue and made the host wait for them): ``image`` must stay valid until its
                # upload has run -- DecodePool(hold=...) guarantees that for its slots, pageable arrays are staged by the runtime
                # before the call returns
                d_img = torch.empty((H, W, Cn), dtype=torch.uint8, device=tdev)
                d_img.copy_(torch.from_numpy(image), non_blocking=True)
                t["uploaded"] = torch.cuda.Event()
                t["uploaded"].record(stream)
                t["d_img"] = d_img
                if use_net:
                    d_gray = torch.empty((h, w), dtype=torch.float32, device=tdev)
                    _lib.check(lib.asep_prep_scale_gray_dev(ws, d_img.data_ptr(), H, W, Cn, float(sc), None,
                                                            d_gray.data_ptr(), sp), "asep_prep_scale_gray_dev")
                    t["d_u8"] = torch.empty((h, w, ncls), dtype=torch.uint8, device=tdev)
                    t["keep"] = (d_gray, torch.empty((h, w, ncls), dtype=torch.float32, device=tdev))
                tickets.append(t)
            if use_net:
                n = len(tickets)
                handle = self.pb_graph.handle(dev, lane)
                if n == 1:
                    t = tickets[0]
                    _lib.check(lib.asep_aru_forward_dev(handle, t["keep"][0].data_ptr(), t["size"][0], t["size"][1], t["keep"][1].data_ptr(),
                                                        t["d_u8"].data_ptr(), None, 0.0, sp), "asep_aru_forward_dev")
                else:
                    Arr, Ints = C.c_void_p * n, C.c_int32 * n
                    _lib.check(lib.asep_aru_forward_batch_dev2(
                        handle, n, Arr(*[t["keep"][0].data_ptr() for t in tickets]), Ints(*[t["size"][0] for t in tickets]),
                        Ints(*[t["size"][1] for t in tickets]), Arr(*[t["keep"][1].data_ptr() for t in tickets]),
                        Arr(*[t["d_u8"].data_ptr() for t in tickets]), None, 0.0, sp), "asep_aru_forward_batch_dev2")
            for t in tickets:
                H, W, Cn = t.pop("full")
                d_img = t.pop("d_img")
                if Cn == 1:
                    d_g8 = d_img
                else:
                    d_g8 = torch.empty((H, W), dtype=torch.uint8, device=tdev)
                    _lib.check(lib.asep_prep_gray_u8_dev(ws, d_img.data_ptr(), H, W, d_g8.data_ptr(), sp), "asep_prep_gray_u8_dev")
                d_swt = torch.empty((H, W), dtype=torch.uint8, device=tdev)
                _lib.check(lib.asep_swt_distance_transform_dev(ws, d_g8.data_ptr(), H, W, d_swt.data_ptr(), sp),
                           "asep_swt_distance_transform_dev")
                t["swt"] = image_ops.DeviceImage(d_swt, dev)
                t["inputs"] = (d_img, d_g8)
                t["done"] = torch.cuda.Event()
                t["done"].record(stream)
        return tickets

    def collect_page(self, t, text_lines):
        """:meth:`collect_boxes` for text-line objects (``id``, ``surr_p``)"""
        return self.collect_boxes(t, *line_boxes(text_lines))

    def collect_boxes(self, t, ids, boxes, has):
        """The three per-line measurements of :94-119 for a ticket of :meth:`enqueue_page` and the lines' bounding boxes
        (``line_boxes``), taken on a side stream while the next page's kernels run: stroke width and text height from the
        device-resident distance transform (:218-245, the crop [ymin : ymax + 2, xmin : xmax + 2]), the mean net confidence
        (:247-270) from exact integer box sums of the uint8 net output: ``sum / 255 / (width * height)`` -- the reference sums
        ``uint8 / 255`` in float64, which differs by rounding in the last bits only.  The box of the rescaled outline is the
        rescaled box of the outline (``int(x * sc)`` is monotone), so only the four extremes of a line are needed; everything
        per line is array arithmetic.  -> (stroke widths, text heights, net confidences), dicts by line id"""
        import torch
        dev = t["device"]
        tdev = torch.device("cuda", dev)
        h, w, ncls = t["size"]
        sc = t["sc"]
        sel = np.flatnonzero(has)
        bx = boxes[sel]
        sw = np.zeros(len(ids))
        ht = np.zeros(len(ids), dtype=np.int64)
        prob = np.zeros(len(ids))
        with torch.cuda.device(tdev):
            side = self._side_stream
            side.wait_event(t["done"])
            sp = C.c_void_p(side.cuda_stream)
            crop = np.stack([bx[:, 0], bx[:, 1], bx[:, 2] + 2, bx[:, 3] + 2], axis=1) if len(sel) else np.zeros((0, 4), np.int64)
            sws, hts = image_ops.swt_line_features(t["swt"], crop, device=dev, stream=sp, lane=1)
            sw[sel], ht[sel] = sws, hts
            if "d_u8" in t and len(sel):
                lo = (bx[:, :2] * sc).astype(np.int64)                  # rescale_points: int(p * sc), truncation towards zero
                hi = (bx[:, 2:] * sc).astype(np.int64)
                size = hi - lo + 1                                      # nominal width, height
                x0, x1 = _slice_bounds(lo[:, 0], size[:, 0], w)
                y0, y1 = _slice_bounds(lo[:, 1], size[:, 1], h)
                sums = image_ops.box_sums_dev(t["d_u8"].data_ptr(), (h, w, ncls), np.stack([x0, y0, x1, y1], axis=1), channel=0,
                                              device=dev, stream=sp, lane=1)
                prob[sel] = sums / 255 / (size[:, 0] * size[:, 1]).astype(np.float64)
            else:
                side.synchronize()
        t.clear()
        return dict(zip(ids, sw.tolist())), dict(zip(ids, ht.tolist())), dict(zip(ids, prob.tolist()))

    def run(self, gpu_device='0'):
        """:272-303."""
        self.gpu_devices = gpu_device
        self.SWT.device = self.device
        new_page_objects = []
        # images are decoded ahead of the GPU by worker processes when host_workers > 1 (host_pipeline.py); the fusion
        # itself needs the device again (per-line statistics), so the PAGE-XML part stays in this process
        from .host_pipeline import DecodePool, WritePool, pin_callbacks, single_threaded_children
        from .net_post_processing_helper import get_scaling_factor
        pipelined = getattr(self, "host_workers", 0) > 1 and not self.keep_outputs
        reg, unreg = pin_callbacks(self.device) if pipelined else (None, None)
        n_workers = self.host_workers if pipelined else 0
        geometry = {}                                        # page path -> future of read_line_geometry, a few pages ahead
        with WritePool(n_workers) as writers:
            if pipelined:
                from concurrent.futures import ProcessPoolExecutor
                import multiprocessing as mp
                parsers = ProcessPoolExecutor(max(1, n_workers // 4), mp_context=mp.get_context("spawn"))
            try:
                ahead = iter(self.image_paths)

                def prefetch_geometry(k):
                    for _ in range(k):
                        nxt = next(ahead, None)
                        if nxt is not None:
                            with single_threaded_children():
                                geometry[nxt] = parsers.submit(read_line_geometry, get_page_path(nxt))
                if pipelined:
                    prefetch_geometry(2 * n_workers)
                pending, n_enqueued = [], 0

                def finish(image_path, ticket):
                    # the GPU owner only measures; parsing happened in a worker, fusion + tags + XML go to a worker
                    page_path = get_page_path(image_path)
                    lines = geometry.pop(image_path).result()
                    prefetch_geometry(1)
                    if lines is None:                       # no PAGE-XML yet: the writer creates an empty one
                        lines = ([], np.zeros((0, 4), np.int64), np.zeros(0, bool))
                    values = self.collect_boxes(ticket, *lines)
                    writers.submit(write_heading_page, page_path, image_path, self.fixed_height, self.scaling_factor,
                                   list(values), self.weight_dict, self.threshold, self.thresh_dict, self.text_line_percentage)

                group = self.PAGE_GROUP if pipelined else 1
                keep = max(2, (self.PAGE_LANES - 1) * group)  # pages queued behind the one whose lines are being measured
                batch, n_groups, n_paths = [], 0, len(self.image_paths)
                for n_seen, (image_path, image) in enumerate(DecodePool(self.image_paths, n_workers, register=reg, unregister=unreg,
                                                                         hold=group + 1 if pipelined else 3), 1):
                    if pipelined:
                        # behind the GPU: PAGE_GROUP decoded pages go through ONE batched net call, consecutive groups on alternating lanes; the next
                        # group is uploaded and queued before a page's lines are measured (the measuring calls wait for their small kernels; the chip
                        # has the next nets to work on meanwhile).  The last pages are uploaded HERE: the pool releases its slots when it ends.
                        batch.append((image_path, image))
                        if len(batch) >= group or n_seen == n_paths:
                            tickets = self.enqueue_group([img for _, img in batch], lane=n_groups % self.PAGE_LANES)
                            n_groups += 1
                            pending.extend((pth, t) for (pth, _), t in zip(batch, tickets))
                            while len(pending) > keep:
                                finish(*pending.pop(0))
                            tickets[-1]["uploaded"].synchronize()    # the images' slots may be recycled from here on
                            batch.clear()
                        continue
                    if self.weight_dict['net'] > 0:
                        net_output = self.heading_probability(image)
                        net_output_post = self.post_process(net_output)
                        if self.keep_outputs:
                            self.net_outputs.append(net_output)
                            self.net_outputs_post.append(net_output_post)
                    else:
                        net_output_post = None
                    swt_feature_image = self.SWT.distance_transform(image, on_device=True)
                    new_page_objects.append(self.to_page_xml(get_page_path(image_path), image_path, net_output_post,
                                                             swt_feature_image))
                for item in pending:
                    finish(*item)
            finally:
                if pipelined:
                    parsers.shutdown()
        return new_page_objects
