"""CLI mirror of ``article_separation/gnn/run_gnn_clustering.py`` on the MI355X engine.

    python -m citlab_article_separation_new_amd.run_gnn_clustering --model_dir net.pb --eval_list jsons.lst \\
        --input_params node_feature_dim=15 edge_feature_dim=2 node_input_feature_mask=[1,1,1,1,0,0,0,0,0,0,0,0,1,1,1] \\
        --clustering_method dbscan --out_dir out

Same flags (``run_gnn_clustering.py:19-73``), same per-page loop (``:237-300``): json -> feed dict -> engine ->
confidences [N,N] -> (optional json) -> TextblockClustering -> ``<out_dir>/.../clustering/<info>/<name>_clustering.xml``.
``--num_workers`` > 1: that many host worker processes prepare feeds ahead of / cluster and write behind one GPU-owning process
per entry of ``--gpu_devices`` (the reference forks one session-owning ``mp.Process`` per sub-list, ``:322-340``).  Worker errors
are surfaced (the reference drops them).
"""
import logging
import multiprocessing as mp
import queue
import os
import sys
import time

import numpy as np

from . import cli_flags
from .host_util import split_list
from .path_util import get_page_from_json_path, get_path_from_exportdir, load_list_file


def build_parser():
    p = cli_flags.LineArgumentParser(fromfile_prefix_chars="@")
    p.add_argument("--model_dir", type=str, default="")
    p.add_argument("--eval_list", type=str, default="")
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--num_classes", type=int, default=2)
    p.add_argument("--num_relation_components", type=int, default=2)
    p.add_argument("--sample_num_relations_to_consider", type=int, default=100)
    p.add_argument("--sample_relations", type=cli_flags.str2bool, default=False)
    p.add_argument("--image_input", type=cli_flags.str2bool, default=False)
    # extension: the backbone end points a graph exported with --image_input reads its visual node features from
    # (the reference fixed them at training time with --feature_map_generation_params from_layer=[...])
    p.add_argument("--visual_layers", type=str, nargs="*", default=None)
    p.add_argument("--assign_visual_features_to_nodes", type=cli_flags.str2bool, default=True)
    p.add_argument("--assign_visual_features_to_edges", type=cli_flags.str2bool, default=False)
    p.add_argument("--mvn", type=cli_flags.str2bool, default=True)
    cli_flags.define_dict(p, "input_params", {})
    p.add_argument("--clustering_method", type=str, default="dbscan", choices=["dbscan", "linkage", "greedy", "dbscan_std"])
    p.add_argument("--mask_horizontally_separated_confs", type=cli_flags.str2bool, default=False)
    p.add_argument("--mask_heading_separated_confs", type=cli_flags.str2bool, default=False)
    cli_flags.define_dict(p, "clustering_params", {})
    p.add_argument("--out_dir", type=str, default="")
    p.add_argument("--save_conf", type=str, default="no_conf", choices=["no_conf", "with_conf", "only_conf"])
    p.add_argument("--num_workers", type=int, default=1)
    p.add_argument("--gpu_devices", type=int, nargs="*", default=[])
    p.add_argument("--gpu_memory_fraction", type=float, default=0.95)
    p.add_argument("--batch_limiter", type=int, default=-1)
    p.add_argument("--try_gpu", type=cli_flags.str2bool, default=None)
    return p


def resolve_model_path(flags):
    """run_gnn_clustering.py:191-204 (also accepts the engine's .asepw container)."""
    if os.path.isfile(flags.model_dir):
        ext = os.path.splitext(os.path.basename(flags.model_dir))[1]
        if ext not in (".pb", ".asepw"):
            raise IOError(f"Given model path {flags.model_dir} is not a .pb")
        return flags.model_dir
    pb = None
    try_gpu = flags.try_gpu if flags.try_gpu is not None else bool(flags.gpu_devices)
    if try_gpu:
        try:
            pb = get_path_from_exportdir(flags.model_dir, "*_gpu.pb", "cpu")
        except IOError:
            logging.warning("Could not find gpu-model-pb-file, continue with cpu-model-pb-file")
    return pb or get_path_from_exportdir(flags.model_dir, "*best*.pb", "_gpu.pb")


def _prepare_page(input_fn, flags, json_path):
    """:237-262 for one page: json (+ scan for a graph exported with image_input) -> (PAGE-XML path, feed dict, number of nodes), or
    None if the json does not exist"""
    page_path = get_page_from_json_path(json_path)
    if not os.path.isfile(json_path):
        logging.warning(f"No json file found to given pageXML {page_path}. Skipping.")
        return None
    image = None
    if flags.image_input:
        from PIL import Image
        from . import image_io
        from .path_util import get_img_from_json_path
        img_path = get_img_from_json_path(json_path)
        # input_dataset.py:279-280: the scan as mode "L".  A plain 8-bit GRAY png is that already (image_io's fast decode);
        # colour files go through Pillow's own luma conversion, which is not OpenCV's
        image = image_io._load_png_plain(img_path) if img_path.lower().endswith(".png") else None
        if image is None or image.ndim != 2:
            with Image.open(img_path) as im:
                image = np.asarray(im.convert("L"))                      # uint8; widened after the resize's gathers
    feed = input_fn.feed_from_json(json_path, image)
    n = feed["node_features:0"].shape[1] if "node_features:0" in feed else int(feed["num_nodes:0"][0])
    return page_path, feed, n


def _finish_page(tb, flags, output, n, page_path):
    """:269-300 for one page: net output -> confidences [N,N] -> (masks, json) -> clustering -> PAGE-XML with article ids;
    -> path of the written file, or None with --save_conf only_conf"""
    from . import gnn_results
    confidences = gnn_results.confidences_from_output(output, n)
    if flags.mask_heading_separated_confs or flags.mask_horizontally_separated_confs:      # :279-281
        from .feature_generation import mask_horizontally_separated_confs
        confidences = mask_horizontally_separated_confs(confidences, page_path,
                                                        mask_heading=flags.mask_heading_separated_confs,
                                                        mask_horizontal=flags.mask_horizontally_separated_confs)
    if flags.save_conf != "no_conf":
        gnn_results.save_conf_to_json(confidences, page_path, flags.out_dir)
        if flags.save_conf == "only_conf":
            return None
    tb.set_confs(confidences)
    tb.calc(method=flags.clustering_method)
    return gnn_results.save_clustering_to_page(tb.tb_labels, page_path, flags.out_dir, info=tb.get_info(flags.clustering_method))


def _load_session(flags, device):
    from . import gnn_io
    graph = gnn_io.load_graph(resolve_model_path(flags), visual_layers=flags.visual_layers or None)
    if graph.cfg.visual_dims and not flags.image_input:
        raise ValueError("this model was exported with image_input: pass --image_input True")
    return gnn_io.GnnSession(graph, device)


def gnn_clustering(json_paths, flags, device="0"):
    from .clustering import TextblockClustering
    from .gnn_input import InputGNN
    sess = _load_session(flags, device)
    input_fn = InputGNN(flags)
    tb = TextblockClustering(flags)
    results = []
    t0 = time.time()
    for count, json_path in enumerate(list(json_paths)):
        if flags.batch_limiter != -1 and flags.batch_limiter <= count:
            break
        page = _prepare_page(input_fn, flags, json_path)
        if page is None:
            continue
        page_path, feed, n = page
        output = sess.run("output_belong_to_same_instance:0", feed_dict=feed)
        out = _finish_page(tb, flags, output, n, page_path)
        if out is not None:
            results.append(out)
    logging.info(f"Time: {time.time() - t0:.2f} seconds")
    return results


# ---- host workers around one GPU owner ------------------------------------------------------------------------------------------
# The forward of a page is 0.1 - 0.7 ms; everything else of the loop above is host work (json, scan decode and resize, clustering,
# PAGE-XML: 50 - 200 ms per page).  With --num_workers > 1 the pages of a GPU go through worker processes that prepare feeds ahead
# of the owner and cluster / write behind it; the owner only runs the sessions, in list order.
_task_state = {}


def _task_objects(argv):
    key = tuple(argv)
    if key not in _task_state:
        from .clustering import TextblockClustering
        from .gnn_input import InputGNN
        flags = build_parser().parse_known_args(list(argv))[0]
        _task_state[key] = (flags, InputGNN(flags), TextblockClustering(flags))
    return _task_state[key]


def _prepare_task(argv, json_path):
    flags, input_fn, _ = _task_objects(argv)
    return _prepare_page(input_fn, flags, json_path)


def _finish_task(argv, output, n, page_path):
    flags, _, tb = _task_objects(argv)
    return _finish_page(tb, flags, output, n, page_path)


def gnn_clustering_pipelined(json_paths, argv, device="0", host_workers=2):
    from concurrent.futures import ProcessPoolExecutor
    from .host_pipeline import single_threaded_children
    flags = build_parser().parse_known_args(list(argv))[0]
    json_paths = list(json_paths)
    if flags.batch_limiter != -1:
        json_paths = json_paths[:max(0, flags.batch_limiter)]
    t0 = time.time()
    results = []
    with ProcessPoolExecutor(max(1, host_workers), mp_context=mp.get_context("spawn")) as pool:
        def submit(fn, *args):
            with single_threaded_children():                # (the executor spawns its processes inside submit, on demand)
                return pool.submit(fn, list(argv), *args)
        ahead = 2 * max(1, host_workers)
        prepared = [submit(_prepare_task, p) for p in json_paths[:ahead]]   # the workers start on the first pages while the
        sess = _load_session(flags, device)                                  # owner imports the engine and loads the model
        finishing = []
        for k in range(len(json_paths)):
            page = prepared[k].result()
            prepared[k] = None
            if k + ahead < len(json_paths):
                prepared.append(submit(_prepare_task, json_paths[k + ahead]))
            if page is None:
                continue
            page_path, feed, n = page
            output = sess.run("output_belong_to_same_instance:0", feed_dict=feed)
            finishing.append(submit(_finish_task, output, n, page_path))
            while len(finishing) > 4 * max(1, host_workers):                 # bounded backlog: surface errors early
                out = finishing.pop(0).result()
                if out is not None:
                    results.append(out)
        for f in finishing:
            out = f.result()
            if out is not None:
                results.append(out)
    logging.info(f"Time: {time.time() - t0:.2f} seconds")
    return results


def _worker(json_paths, argv, device, q, index=0, host_workers=0):
    try:
        if host_workers > 1:
            q.put((index, "ok", gnn_clustering_pipelined(json_paths, argv, device, host_workers)))
        else:
            flags = build_parser().parse_known_args(argv)[0]
            q.put((index, "ok", gnn_clustering(json_paths, flags, device)))
    except Exception as e:  # surfaced to the parent instead of being dropped
        q.put((index, "err", repr(e)))


def _collect_results(procs, q):
    """One (worker index, "ok" | "err", payload) message per worker.  A worker that ends WITHOUT having posted its message is an
    error (native crash, OOM kill, HIP abort: fail the run, do not hang); a worker that posted its result and then exits
    non-zero (a crash at interpreter / HIP teardown) is not -- its pages are done."""
    out, errors = [], []
    reported = set()
    while len(reported) < len(procs):
        try:
            index, status, payload = q.get(timeout=1.0)
        except queue.Empty:
            dead = [k for k, pr in enumerate(procs) if k not in reported and not pr.is_alive() and pr.exitcode not in (0, None)]
            if dead and q.empty():
                errors.extend(f"worker {k} (pid {procs[k].pid}) ended with exit code {procs[k].exitcode} without a result" for k in dead)
                break
            continue
        reported.add(index)
        (out.extend if status == "ok" else errors.append)(payload)
    for pr in procs:
        if errors and pr.is_alive():
            pr.terminate()
        pr.join()
    return out, errors


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    flags = build_parser().parse_known_args(argv)[0]
    logging.getLogger().setLevel(logging.INFO)
    json_paths = [p for p in load_list_file(flags.eval_list) if p]
    devices = [str(d) for d in flags.gpu_devices] or ["0"]
    if flags.num_workers <= 1:
        return gnn_clustering(json_paths, flags, devices[0])
    # --num_workers = HOST workers (feeds ahead of / clustering and PAGE-XML behind the GPU owners), split evenly over one
    # GPU-owning process per device -- like --num_processes of run_net_post_processing; the reference forks one session-owning
    # process per sub-list (:322-340), which on one GPU means N HIP contexts and N model copies for 0.1 ms of work per page
    n_owners = max(1, min(len(devices), flags.num_workers, len(json_paths) or 1))
    host_workers = max(1, flags.num_workers // n_owners)
    if n_owners == 1:
        return gnn_clustering_pipelined(json_paths, argv, devices[0], host_workers)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = []
    for k, part in enumerate(split_list(json_paths, n_owners)):
        pr = ctx.Process(target=_worker, args=(part, argv, devices[k], q, k, host_workers))
        pr.start()
        procs.append(pr)
    out, errors = _collect_results(procs, q)
    if errors:
        raise RuntimeError("worker failure: " + "; ".join(errors))
    return out


if __name__ == "__main__":
    main()
