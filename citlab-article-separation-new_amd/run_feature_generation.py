"""CLI mirror of ``article_separation/gnn/run_feature_generation.py`` (same flags and worker fan-out).

    python -m citlab_article_separation_new_amd.run_feature_generation --pagexml_list pages.lst \\
        [--out_dir DIR] [--interaction delaunay|fully] [--visual_regions True] [--separators bb|line] \\
        [--external_jsons a.json b.json] [--num_workers N]

``--wv_language/--wv_path`` (word-vector similarities) are accepted for flag compatibility and rejected when set.
Workers are spawned processes, worker k computing its distance transforms on GPU ``k % n_gpus``.
"""
import logging
import multiprocessing as mp
import os
import sys

from . import cli_flags
from .host_util import split_list


def build_parser():
    p = cli_flags.LineArgumentParser(fromfile_prefix_chars="@")
    p.add_argument("--pagexml_list", type=str, default="")
    p.add_argument("--out_dir", type=str, default="")
    p.add_argument("--interaction", type=str, choices=["fully", "delaunay"], default="delaunay")
    p.add_argument("--visual_regions", type=cli_flags.str2bool, default=False)
    p.add_argument("--separators", type=str, choices=["line", "bb"], default="bb")
    p.add_argument("--external_jsons", type=str, nargs="*", default=[])
    p.add_argument("--wv_language", type=str, default=None)
    p.add_argument("--wv_path", type=str, default=None)
    p.add_argument("--num_workers", type=int, default=1)
    return p


def _worker(sublist, flags, device):
    from .feature_generation import generate_feature_jsons
    generate_feature_jsons(sublist, flags.out_dir, flags.interaction, flags.visual_regions, flags.external_jsons,
                           (flags.wv_language, flags.wv_path), flags.separators, device=device)


def main(argv=None):
    flags = build_parser().parse_known_args(sys.argv[1:] if argv is None else argv)[0]
    logging.getLogger().setLevel("INFO")
    if flags.external_jsons:
        logging.info("Forced num_workers to 1, since external jsons are used.")
        flags.num_workers = 1
    page_paths = [os.path.abspath(line.rstrip()) for line in open(flags.pagexml_list)]
    n = flags.num_workers
    if n > 1:
        import torch
        n_gpus = max(torch.cuda.device_count(), 1)          # does not initialise the GPU
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=_worker, args=(sub, flags, k % n_gpus))
                 for k, sub in enumerate(split_list(page_paths, n))]
        for pr in procs:
            pr.start()
        rc = 0
        for pr in procs:
            pr.join()
            rc = rc or pr.exitcode
        return rc
    _worker(page_paths, flags, 0)
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
