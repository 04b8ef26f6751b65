#!/usr/bin/env python3
"""bench.py -- pages/sec of the ARU-Net (+ GNN relation) hot path on synthetic 3000x4500 newspaper scans.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
  (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).  Started WITHOUT that environment
  (`python bench.py --gpus 8`), the process starts the N ranks itself -- torch.distributed.run as a CHILD process, before
  anything here has touched the GPU -- and relays the child's one JSON line and exit code (spawn_ranks).

A "step" = one pass of the hot path over one batch of `--pages-per-step` device-resident pages on every rank:
per page one ARU-Net forward (fp32, fused uint8/threshold epilogue) + one forward of the VISUAL relation net
BASELINE configs[3] names (mixed_gnn_vn7e2: RU backbone on the page image at 683 x 1024, ROI max + compression to
3 x 16 visual node features, graph with 7 + 48 = 55 node features, N=200 nodes, 20 000 directed edges after
correction, all 40 000 ordered pairs).  Pages are independent, so ranks shard the page list (weak scaling) and the only
collective is the weight broadcast at start-up.

Prints ONE JSON line on rank 0: BASELINE.json's metric + `roofline` (dominant kernel, timed live with HIP events on
the launch streams: in situ = in the real schedule beside the other streams, and isolated) + `cpu_baseline` (the CPU
oracle timed on this box's host cores on a bounded sample; rank 0, N == 1 only).  scripts/roofline_from_profiles.py
recomputes the roofline block from the rocprofv3 summaries under profiles/<tag>/.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# torch is imported in main(), not here: the files-in / files-out leg starts decode worker processes with the "spawn" method, and a
# spawned child re-imports THIS module as __mp_main__ -- with `import torch` at the top every one of the 24 workers paid for it
# (start-up, memory, a thread pool per worker) and the leg ran at 36 instead of 68 pages/s
torch = None

METRIC = "newspaper pages/sec (ARU-Net seg + GNN relation) at 3000x4500 px"
PEAK_F32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: dense f32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0      # same guide: dense bf16 matrix peak (not the 2:1-sparsity headline)
PEAK_HBM_GBS = 8000.0
# ASEP_* names the Python side / this script reads (not engine switches)
HOST_VARIABLES = {"ASEP_HIP_LIB", "ASEP_COMPUTE_DTYPE", "ASEP_GPU_OWNERS", "ASEP_OWNER_STATS_DIR", "ASEP_BENCH_DEVICE", "ASEP_BENCH_BACKEND",
                  "ASEP_BENCH_FORCE_DIST", "ASEP_BENCH_OWNERS", "ASEP_LAYER_PROFILE_CFG", "ASEP_LAYER_PROFILE_PAGES", "ASEP_POOL_TRACE"}
ACHIEVABLE_HBM_GBS = 6300.0         # same guide: float4 copy, 79 % of the 8 TB/s peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps; default 80 (x 16 pages = 1280 pages: a timed region of >= 10 s at the fp32 rate), 300 with "
                         "--dtype bf16 (>= 8 s up to 600 pages/s; 240 until the rate passed 480)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pages-per-step", type=int, default=16,
                    help="pages per rank and step (16 x 20 steps = 320 pages: a timed region of ~3 s, long enough for "
                         "sustained clocks and for an external smi sampler; ~6 GB of HBM per page in flight)")
    ap.add_argument("--height", type=int, default=4500)
    ap.add_argument("--width", type=int, default=3000)
    ap.add_argument("--no-gnn", action="store_true", help="ARU-Net only (diagnostic; not the headline metric)")
    ap.add_argument("--gnn", choices=["visual", "geometric"], default="visual",
                    help="visual = the net configs[3] names (mixed_gnn_vn7e2: backbone on 683 x 1024 + 55 node features; the "
                         "headline); geometric = the 7-feature net without image input (round-2 workload, diagnostic)")
    ap.add_argument("--kernel-timing", choices=["both", "in-situ", "isolated", "none"], default="both",
                    help="HIP-event per-kernel passes after the timed region (in-situ only under rocprofv3, so that every launch of "
                         "the traced process runs in the same schedule)")
    ap.add_argument("--dtype", choices=["f32", "bf16", "f32s"], default="f32s",
                    help="f32s = BASELINE configs[1] on the fp32 engine's default arithmetic (the headline since round 5): fp32 tensors, "
                         "fp32 accumulation, fp32 results, every product of the >= 12-channel convolutions as 6 bf16 MFMAs of the exact "
                         "3-way bfloat16 split of both factors -- held to the fp32 parity gates; f32 = the same step on the plain fp32 "
                         "MFMA / Winograd kernels (secondary.plain_f32_full_step); bf16 = configs[4] 'bf16 convs': bf16 tensors and MFMA "
                         "operands, fp32 accumulation, probability maps within 2e-2 (reported with dtype bf16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", metavar="OUT.json", default=None,
                    help="ONLY the CPU baseline by BASELINE.md section 3's protocol, no GPU work: whole pages from PNG files through every stage "
                         "(decode, scale + gray, ARU-Net oracle, uint8 / threshold, CC filter + openings, polygon rings, PAGE-XML) and graphs "
                         "through the relation-net oracle, clustering and PAGE-XML; 1 warm-up + 8 pages + 64 graphs, per-stage wall clock "
                         "(~10 minutes on 16 CPUs) -> OUT.json (committed as profiles/r5_cpu_baseline.json; the default line cites it)")
    ap.add_argument("--event-steps", type=int, default=3,
                    help="steps per HIP-event pass (scripts/profile_bench.sh uses many event-timed and few plain steps, so that "
                         "rocprofv3's per-kernel averages cover the launches the events bracket)")
    ap.add_argument("--dominant", default=None,
                    help="report the roofline block for THIS kernel (scripts/profile_round.sh passes the kernel the un-traced default "
                         "line named, so that the traced line and the rocprofv3 summaries describe the same kernel)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="same as --kernel-timing none")
    ap.add_argument("--e2e-pages", type=int, default=384,
                    help="scans of the files-in / files-out secondary figure (separator CLI path with host workers; 0 = skip)")
    ap.add_argument("--e2e-heading-pages", type=int, default=384,
                    help="scans of the heading command line inside the files-in / files-out leg (0 = skip)")
    ap.add_argument("--e2e-gnn-pages", type=int, default=384,
                    help="pages of the relation net's command line inside the files-in / files-out leg (0 = skip)")
    ap.add_argument("--e2e-n-pages-per-owner", type=int, default=96,
                    help="--gpus N > 1: scans per GPU owner of the N-owner files-in / files-out leg (secondary.e2e_files_n; 0 = skip)")
    ap.add_argument("--e2e-n-leg", type=int, default=0, help=argparse.SUPPRESS)    # internal: run only the N-owner leg with this many owners
    ap.add_argument("--e2e-leg", action="store_true", help=argparse.SUPPRESS)      # internal: run only the e2e_files leg, print its JSON
    ap.add_argument("--e2e-device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--bf16-steps", type=int, default=160,
                    help="timed steps of the secondary bf16 full step (configs[4]: bf16 ARU-Net + visual relation net with bf16 backbone; "
                         "160 x 16 pages >= 5 s at its rate; 0 = skip)")
    ap.add_argument("--plain-steps", type=int, default=50,
                    help="timed steps of the secondary step on the PLAIN fp32 kernels (compute_dtype f32; 50 x 16 pages >= 6 s at its rate; 0 = skip)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurements (bf16 variant, heading net + stroke-width fusion, visual GNN)")
    ap.add_argument("--cpu-sample-height", type=int, default=0,
                    help="rows of a page for the CPU baseline (0 = one whole page per worker on boxes with >= 64 cores, "
                         "a third of a page on smaller ones)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 300 if args.dtype == "bf16" else 80
    return args


def spawn_ranks(args):
    """`python bench.py --gpus N` (N > 1) without a torchrun environment: start the ranks as a child process (never an exec of
    this one; nothing in this process has initialised the GPU yet), relay its single JSON line and its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:                                 # a free rendezvous port (two runs on one box must not collide)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif r.returncode == 0:
        print(r.stdout, file=sys.stderr)
        return 1
    return r.returncode


def run_cpu_baseline(args):
    """Mirrors the reference's process fan-out (run_net_post_processing.py:61-82): P worker processes, each timing
    the torch-CPU ARU-Net oracle on a band of `rows` rows of the same synthetic scan the GPU path gets (+ the relation-net
    oracle on one graph: visual net incl. its backbone on 683 x 1024, or the geometric net).
    pages/s = P * (rows/H page) / slowest worker."""
    import subprocess
    from citlab_article_separation_new_amd.host_util import effective_cpus
    logical = os.cpu_count() or 1
    cores = effective_cpus()                       # affinity mask and cgroup quota: what this container may actually use
    threads = min(16, cores)                       # torch-CPU convs of this net stop scaling at ~16 threads
    workers = max(1, min(8, (cores // 2 or 1) // threads)) if cores > threads else 1
    rows = args.cpu_sample_height or (args.height if cores >= 64 else max(256, args.height // 3))
    cmd = [sys.executable, "-m", "oracle.cpu_worker", "--threads", str(threads), "--rows", str(rows),
           "--width", str(args.width), "--height", str(args.height)]
    if not args.no_gnn:
        cmd += ["--gnn"] + (["--visual"] if args.gnn == "visual" else [])
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd + ["--page", str(i)], cwd=ROOT, stdout=subprocess.PIPE, text=True) for i in range(workers)]
    res = []
    for p in procs:
        out, _ = p.communicate(timeout=900)
        if p.returncode == 0 and out.strip():
            res.append(json.loads(out.strip().splitlines()[-1]))
    wall = time.perf_counter() - t0
    if not res:
        return None
    frac = rows / args.height
    t_page = max(r["t_aru"] / frac + r["t_gnn"] for r in res)      # slowest worker (scaled to a whole page if a band)
    used = threads * len(res)
    short = (f"{len(res)}x{threads} threads: torch-CPU fp32 ARU-Net oracle on "
             + (f"one {args.width}x{args.height} scan" if rows >= args.height else f"a {args.width}x{rows} band (x{1 / frac:.2f})")
             + ("" if args.no_gnn else (" + visual relation-net oracle" if args.gnn == "visual" else " + numpy GNN oracle")))
    full = None
    try:                                           # the committed full-protocol run (bench.py --cpu-baseline-full), cited beside the band sample
        fq = json.load(open(os.path.join(ROOT, "profiles", "r5_cpu_baseline.json")))
        full = {"file": "profiles/r5_cpu_baseline.json", "pages_per_s_end_to_end": fq["pages_per_s_end_to_end"], "cores": fq["cores"],
                "seconds_per_page_by_stage": fq["seconds_per_page_by_stage"], "seconds_per_graph_by_stage": fq["seconds_per_graph_by_stage"]}
    except (OSError, KeyError, ValueError):
        pass
    return {
        "value": round(len(res) / t_page, 5), "unit": "pages/s", "cores": used, "cores_of": cores, "logical_cpus": logical, "kind": "port",
        "sample": short[:118], "full_protocol": full,
        "sample_detail": (f"{used} of the {cores} CPUs this container may use (the box has {logical} logical CPUs"
                   + (f", cgroup quota {cores}" if cores < logical else "") + f"): {len(res)} worker processes x {threads} threads, each: torch-CPU "
                   "fp32 ARU-Net oracle on "
                   + (f"one whole synthetic {args.width}x{args.height}px scan" if rows >= args.height else
                      f"a {args.width}x{rows}px band of a synthetic scan (scaled x{1 / frac:.2f} to a page)")
                   + ("" if args.no_gnn else (" + the visual relation-net oracle (torch-CPU backbone on 683x1024 + numpy graph)"
                                             if args.gnn == "visual" else " + numpy GNN oracle on one graph"))
                   + f"; no decode / post-processing; slowest worker {t_page:.2f}s/page (relation net "
                   f"{max(r['t_gnn'] for r in res):.2f}s); {wall:.1f}s wall incl. start-up and page generation"),
    }


def run_cpu_baseline_full(args):
    """the CPU restatement of the whole pipeline on whole pages and graphs, per stage (oracle/cpu_worker.py --full)."""
    import subprocess
    from citlab_article_separation_new_amd.host_util import effective_cpus
    cores = effective_cpus()
    threads = min(16, cores)
    cmd = [sys.executable, "-m", "oracle.cpu_worker", "--full", "--threads", str(threads), "--pages", "8", "--graphs", "64",
           "--width", str(args.width), "--height", str(args.height)]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, text=True, timeout=7200)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise SystemExit(f"oracle.cpu_worker --full exited with {r.returncode}")
    q = json.loads(lines[-1])
    q.update({"kind": "port", "unit": "pages/s", "value": q["pages_per_s_end_to_end"], "cores": threads, "cores_of": cores, "logical_cpus": os.cpu_count(),
              "wall_s": round(time.perf_counter() - t0, 1), "height": args.height, "width": args.width,
              "protocol": "BASELINE.md section 3: 1 warm-up + 8 whole pages from PNG files (decode .. PAGE-XML) + 64 graphs of 200 text blocks (json .. "
                          "PAGE-XML with article ids), one process, torch-CPU / numpy oracles, per-stage wall clock; pages/s = 1 / (s per page + s per graph)"})
    return q


# The fused level-0 residual blocks are ONE design instantiated for the two directions of the U (down: image in, pool out; up: the
# concatenation [skip, deconv] in): they are ranked and reported as one entry, "a+b" (under the isolated
# ranking the up block led, under rocprofv3's the down block, and the line printed the figure of whichever it had picked).
LEVEL0_FAMILIES = (("res8v_down_kernel", "res8v_up_kernel"), ("res8_down_kernel", "res8_up_kernel"), ("res8w_kernel<false>", "res8w_kernel<true>"),
                   ("res8f_kernel<false>", "res8f_kernel<true>"), ("res8b_kernel<false>", "res8b_kernel<true>"))


def family_members(kernel, names):
    """-> the kernels of `names` that are reported together with `kernel` (itself alone for everything but the level-0 blocks)"""
    for fam in LEVEL0_FAMILIES:
        if kernel.startswith(fam):
            tail = kernel[len(next(f for f in fam if kernel.startswith(f))):]          # e.g. "<0>": the activation of the res8v blocks
            return sorted(n for n in names if n.startswith(fam) and n[len(next(f for f in fam if n.startswith(f))):] == tail)
    return [kernel]


def merge_records(recs):
    """kernel records of one pass -> one record "a+b": summed launches, time, FLOPs and bytes; rates and the mean launch recomputed"""
    if len(recs) == 1:
        return dict(recs[0])
    m = {"kernel": "+".join(r["kernel"] for r in recs), "members": [r["kernel"] for r in recs]}
    for key in ("calls", "total_ms", "flops", "bytes", "executed_flops"):
        m[key] = sum(r.get(key, 0.0) for r in recs)
    sec = m["total_ms"] * 1e-3
    m["avg_us"] = 1e3 * m["total_ms"] / m["calls"]
    m["tflops"] = m["flops"] / sec / 1e12 if sec > 0 else 0.0
    m["executed_tflops"] = m["executed_flops"] / sec / 1e12 if sec > 0 else 0.0
    m["algo_gbs"] = m["bytes"] / sec / 1e9 if sec > 0 else 0.0
    for key in ("pipe", "pipe_peak", "bf16_tflops"):
        if key in recs[0]:
            m[key] = recs[0][key]
    return m


def rank_kernels(iso, situ):
    """kernel records ({name: {"kernel", "total_ms", ...}}) of the isolated and / or the in-situ pass -> list of ENTRIES (a kernel, or
    the two level-0 blocks as one "a+b" entry), largest summed launch time first.  Ranked IN SITU when that pass exists (round 6; the
    isolated figures ride beside it): rounds 3-5 ranked on the isolated totals and printed the in-situ figure of the kernel that led THERE."""
    base = situ or iso
    seen, entries = set(), []
    for name in base:
        if name in seen:
            continue
        mem = family_members(name, list(base))
        seen.update(mem)
        entries.append(merge_records([base[n] for n in mem]))
    return sorted(entries, key=lambda k: -k["total_ms"])


def entry_of(records, entry):
    """the same entry (kernel or "a+b") built from another pass's records, or None"""
    if not records:
        return None
    mem = entry.get("members") or [entry["kernel"]]
    return merge_records([records[n] for n in mem]) if all(n in records for n in mem) else None


def _timed(fn, iters, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


VISUAL_LAYERS = ["scale_0_unet_up_2_conv", "scale_0_unet_up_1_conv", "scale_0_unet_up_0_conv"]


def e2e_files(args, dev_index):
    """Files in, files out through the separator CLI path (SeparatorNetPostProcessor = run_net_post_processing.py --mode
    separator --fixed_height 4500): PNG scans on disk -> decode in host workers -> GPU stages -> PAGE-XML files.  Four real
    files, the rest links to them (each with its own PAGE-XML output)."""
    import tempfile
    from PIL import Image
    from citlab_article_separation_new_amd import net_post_processing_helper as helper, synth
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.host_pipeline import host_workers_default
    from citlab_article_separation_new_amd.host_util import effective_cpus
    from citlab_article_separation_new_amd.separator_net_post_processor import SeparatorNetPostProcessor
    from citlab_article_separation_new_amd.weights import init_aru_weights
    H, W, n = args.height, args.width, args.e2e_pages
    cfg = AruConfig(compute_dtype=args.dtype)
    graph = helper.AruGraph(init_aru_weights(cfg, 21, logit_scale=0.05), cfg)
    workers = host_workers_default()
    with tempfile.TemporaryDirectory(prefix="asep_e2e_") as tmp:
        os.makedirs(os.path.join(tmp, "page"))
        paths = []
        for k in range(n):
            q = os.path.join(tmp, f"p{k:03d}.png")
            if k < 4:
                Image.fromarray(synth.cached_synth_page(k, W, H)).save(q, compress_level=1)
            else:
                os.symlink(os.path.join(tmp, f"p{k % 4:03d}.png"), q)
            paths.append(q)
        SeparatorNetPostProcessor(paths[:2], graph, H, 1.0, 0.5, str(dev_index), host_workers=0).run()    # warm-up
        proc = SeparatorNetPostProcessor(paths, graph, H, 1.0, 0.5, str(dev_index), host_workers=workers)
        t0 = time.perf_counter()
        proc.run()
        dt = time.perf_counter() - t0
        n_xml = len([f for f in os.listdir(os.path.join(tmp, "page")) if f.endswith(".xml.xml")])
        # the heading command line on the same scans (BASELINE configs[2] as files): PAGE-XML with ~700 text lines per page in,
        # heading tags out; the net on the full page like the separator leg
        heading = None
        if args.e2e_heading_pages > 0:
            from citlab_article_separation_new_amd.heading_net_post_processor import HeadingNetPostProcessor
            nh = min(n, args.e2e_heading_pages)
            for f in os.listdir(os.path.join(tmp, "page")):
                os.remove(os.path.join(tmp, "page", f))
            n_lines = [synth.synth_page_xml(os.path.join(tmp, "page", f"p{k:03d}.xml"), W, H, k % 4) for k in range(nh)]
            wd = {'net': 0.8, 'stroke_width': 0.0, 'text_height': 0.2}           # run_net_post_processing.py:15-23
            td = {'net_thresh': 1.0, 'stroke_width_thresh': 1.0, 'text_height_thresh': 0.9, 'sw_th_thresh': 0.9}
            hp = HeadingNetPostProcessor(paths[:2], graph, H, 1.0, wd, 0.4, td, 0.8)
            hp.host_workers = 0
            hp.run(gpu_device=str(dev_index))                                    # warm-up
            hp = HeadingNetPostProcessor(paths[:nh], graph, H, 1.0, wd, 0.4, td, 0.8)
            hp.host_workers = workers
            t0 = time.perf_counter()
            hp.run(gpu_device=str(dev_index))
            dth = time.perf_counter() - t0
            n_hx = len([f for f in os.listdir(os.path.join(tmp, "page")) if f.endswith(".xml.xml")])
            heading = {"pages_per_s": round(nh / dth, 2), "ms_per_page": round(1e3 * dth / nh, 2), "scans": nh,
                       "text_lines_per_page": int(np.mean(n_lines)), "page_xml_written": n_hx,
                       "note": f"heading CLI path, --fixed_height {H}: PNG + PAGE-XML files -> decode / parse / write workers around "
                               "one GPU owner (net, gray conversion, stroke-width transform, per-line statistics and box sums on "
                               "the device) -> PAGE-XML with heading tags; worker start-up included"}
    graph.close()
    # the relation net's command line (BASELINE configs[3] as files): graph jsons + the scans + PAGE-XML in, article ids out
    gnn_cli = None
    if args.e2e_gnn_pages > 0 and not args.no_gnn:
        from citlab_article_separation_new_amd import run_gnn_clustering
        os.environ["ASEP_COMPUTE_DTYPE"] = args.dtype                        # conv backbone of a model loaded from a file
        ng = args.e2e_gnn_pages
        with tempfile.TemporaryDirectory(prefix="asep_e2e_gnn_") as tg:
            argv = synth.write_gnn_cli_inputs(tg, ng, visual=args.gnn == "visual", W=W, H=H)
            cwd = os.getcwd()
            os.chdir(tg)                                                     # outputs are placed relative to the cwd, like the reference
            try:
                t0 = time.perf_counter()
                outs = run_gnn_clustering.main(argv + ["--out_dir", "out", "--gpu_devices", str(dev_index), "--num_workers", str(workers)])
                dtg = time.perf_counter() - t0
            finally:
                os.chdir(cwd)
        gnn_cli = {"pages_per_s": round(len(outs) / dtg, 2), "ms_per_page": round(1e3 * dtg / max(len(outs), 1), 2), "pages": len(outs),
                   "relation_net": args.gnn,
                   "note": f"run_gnn_clustering, {workers} host workers around one GPU owner: graph json (200 text blocks, ~20k directed edges)"
                           + (" + PNG scan (decode, TF1 bilinear resize to 683x1024)" if args.gnn == "visual" else "")
                           + " + PAGE-XML -> relation net -> dbscan -> PAGE-XML with article ids; model load and worker start-up included"}
    first = proc.first_page_seconds or 0.0
    return {"pages_per_s": round(n / dt, 2), "ms_per_page": round(1e3 * dt / n, 2), "scans": n, "host_workers": workers,
            "first_page_s": round(first, 2), "steady_pages_per_s": round((n - 1) / max(dt - first, 1e-9), 2),
            "page_xml_written": n_xml, "gpu_owner_device_stage_share": round(proc.device_seconds / dt, 3),
            "gpu_owner_waiting_for_decode_share": round(proc.wait_seconds / dt, 3),
            "gpu_owner_ring_chaining_share": round(proc.host_seconds / dt, 3),
            "dtype": args.dtype, "heading": heading, "gnn_clustering": gnn_cli,
            "note": f"separator CLI path, --fixed_height {H} (net on the full {W}x{H} page): PNG files -> {workers} decode / XML "
                    f"worker processes around ONE GPU owner (one page behind the GPU: page n+1 is uploaded and queued before page n's segments are waited for; device_stage = upload + queueing + waiting for results) -> PAGE-XML files; worker start-up (first_page_s: process spawn, page-locking of the decode slots, first decode) inside pages_per_s, excluded from "
                    f"steady_pages_per_s; "
                    f"box has {os.cpu_count()} logical CPUs, this container may use {effective_cpus()} (affinity / cgroup quota): "
                    f"a PNG decode of one scan costs ~0.045 CPU-seconds (libdeflate / zlib + csrc/host_png.c; 0.11 through Pillow), the PAGE-XML ~0.03"}


def e2e_files_n(args, n_owners):
    """Files in, files out with N GPU OWNERS (what a --gpus N run would otherwise not show: the device-resident step shards perfectly, the
    files-in / files-out path is fed by host CPUs): `run_net_post_processing --mode separator` on n_owners x --e2e-n-pages-per-owner PNG
    scans, one GPU-owning process per device (ASEP_BENCH_OWNERS="0,0" puts several owners on one device: the two-owners-on-one-GPU test),
    the container's CPU quota split evenly into host workers -- the reference's fan-out (run_net_post_processing.py:61-82) with owners
    instead of sessions.  Runs as a child of rank 0 after the timed region; the command line spawns its owners itself."""
    import tempfile
    from PIL import Image
    from citlab_article_separation_new_amd import run_net_post_processing as cli, synth
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.host_util import effective_cpus
    from citlab_article_separation_new_amd.weights import init_aru_weights, save_weights
    H, W = args.height, args.width
    n = n_owners * args.e2e_n_pages_per_owner
    cpus = effective_cpus()
    workers = max(n_owners, cpus - n_owners)                  # host workers in total: the quota minus one CPU per owner process
    cfg = AruConfig(compute_dtype=args.dtype)
    with tempfile.TemporaryDirectory(prefix="asep_e2e_n_") as tmp:
        os.makedirs(os.path.join(tmp, "page"))
        os.makedirs(os.path.join(tmp, "stats"))
        model = os.path.join(tmp, "separator.asepw")
        save_weights(model, init_aru_weights(cfg, 21, logit_scale=0.05), {"aru_cfg": cfg.to_dict()})
        paths = []
        for k in range(n):
            q = os.path.join(tmp, f"p{k:04d}.png")
            if k < 4:
                Image.fromarray(synth.cached_synth_page(k, W, H)).save(q, compress_level=1)
            else:
                os.symlink(os.path.join(tmp, f"p{k % 4:04d}.png"), q)
            paths.append(q)
        lst = os.path.join(tmp, "images.lst")
        with open(lst, "w") as f:
            f.write("\n".join(paths) + "\n")
        os.environ["ASEP_OWNER_STATS_DIR"] = os.path.join(tmp, "stats")
        os.environ["ASEP_COMPUTE_DTYPE"] = args.dtype
        if os.environ.get("ASEP_BENCH_OWNERS"):
            os.environ["ASEP_GPU_OWNERS"] = os.environ["ASEP_BENCH_OWNERS"]
        t0 = time.perf_counter()
        rc = cli.main(["--path_to_image_list", lst, "--path_to_pb", model, "--mode", "separator", "--fixed_height", str(H),
                       "--threshold", "0.5", "--num_processes", str(workers)])
        dt = time.perf_counter() - t0
        n_xml = len([f for f in os.listdir(os.path.join(tmp, "page")) if f.endswith(".xml.xml")])
        owners = [json.load(open(os.path.join(tmp, "stats", f))) for f in sorted(os.listdir(os.path.join(tmp, "stats")))]
    share = lambda key: round(sum(o[key] or 0.0 for o in owners) / max(sum(o["seconds"] for o in owners), 1e-9), 3)
    return {"rc": rc, "pages_per_s": round(n / dt, 2), "ms_per_page": round(1e3 * dt / n, 2), "scans": n, "page_xml_written": n_xml,
            "gpu_owners": len(owners), "owner_devices": [o["device"] for o in owners], "host_workers_total": workers,
            "host_workers_per_owner": owners[0]["host_workers"] if owners else None, "cpus_this_container_may_use": cpus,
            "logical_cpus": os.cpu_count(), "pages_per_s_per_owner": [round(o["pages"] / o["seconds"], 2) for o in owners],
            "owner_device_stage_share": share("device_seconds"), "owner_waiting_for_decode_share": share("wait_seconds"),
            "owner_ring_chaining_share": share("host_seconds"), "dtype": args.dtype,
            "note": f"run_net_post_processing --mode separator --fixed_height {H} on {n} PNG scans: {len(owners)} GPU-owning processes x "
                    f"{owners[0]['host_workers'] if owners else 0} decode / PAGE-XML workers each (CPU quota {cpus} split evenly), model load and worker "
                    "start-up included; shares = owners' wall time inside the device stages / waiting for a decoded scan / chaining rings"}


def secondary_measurements(args, lib, dev, imgs, pages_u8, graphs, visual_pages):
    """Not the headline: the other BASELINE.json configs on the same box, each a short timed loop after the main
    measurement (device-resident inputs, same conventions).
      configs[4] precision: bf16 MFMA operands (fp32 accumulation / storage) at 3000 x 4500
      configs[2]: heading net + stroke-width distance transform + per-line statistics of ~700 text lines
      configs[3] alone: the visual relation net, one page per call and 16 pages per grouped call; the geometric 7-feature net
      files in / files out: the separator CLI path on PNG scans with host workers"""
    from citlab_article_separation_new_amd import _lib, gnn_io, image_ops
    from citlab_article_separation_new_amd.config import AruConfig, GnnConfig
    from citlab_article_separation_new_amd.net_post_processing_helper import AruGraph
    from citlab_article_separation_new_amd.weights import init_aru_weights, init_gnn_weights
    H, W = args.height, args.width
    out = {}
    s = torch.cuda.current_stream().cuda_stream
    B2 = min(4, len(imgs))
    Arr = C.c_void_p * B2
    prob = [torch.empty(H, W, 2, device=dev) for _ in range(B2)]
    u8 = [torch.empty(H, W, 2, device=dev, dtype=torch.uint8) for _ in range(B2)]
    p_img, p_out, p_u8 = Arr(*[t.data_ptr() for t in imgs[:B2]]), Arr(*[t.data_ptr() for t in prob]), Arr(*[t.data_ptr() for t in u8])
    try:
        # ---- the other precision ----
        other = "bf16" if args.dtype != "bf16" else "f32s"
        cfg16 = AruConfig(compute_dtype=other)
        g16 = AruGraph(init_aru_weights(cfg16, 1234), cfg16)
        h16 = g16.handle(dev.index or 0)
        dt = _timed(lambda: _lib.check(lib.asep_aru_forward_batch_dev(h16, B2, p_img, H, W, p_out, p_u8, None, 0.05, s), other), 8, warmup=2)
        out["aru_bf16_mfma" if other == "bf16" else "aru_f32s"] = {
            "pages_per_s": round(B2 / dt, 2), "ms_per_page": round(1e3 * dt / B2, 3), "dtype": other,
            "note": ("BASELINE configs[4] precision; probability maps within 2e-2 of the fp32 oracle (tests/test_full_frame_gpu.py); "
                     if other == "bf16" else "") + "ARU-Net alone, no relation net beside it"}
        g16.close()
        # ---- the UPSTREAM ARU-Net layout (ARU_v1.py:35-43: 6 levels, 5 attention scales; SURVEY 8d: the shipped .pb's true cfg is unknown),
        #      ARU-Net alone on the same pages, both arithmetics (parity at this size: tests/test_full_frame_gpu.py) ----
        up = {}
        for dt6 in (("f32s", "bf16") if args.dtype != "f32" else ("f32", "bf16")):
            cfg6 = AruConfig(scale_space_num=6, num_scales_att=5, compute_dtype=dt6)
            g6 = AruGraph(init_aru_weights(cfg6, 1234), cfg6)
            h6 = g6.handle(dev.index or 0)
            dt = _timed(lambda: _lib.check(lib.asep_aru_forward_batch_dev(h6, B2, p_img, H, W, p_out, p_u8, None, 0.05, s), "upstream layout"), 6, warmup=2)
            gf = lib.asep_aru_flops(h6, H, W) / 1e9
            up[dt6] = {"pages_per_s": round(B2 / dt, 2), "ms_per_page": round(1e3 * dt / B2, 3), "gflop_per_page": round(gf, 1),
                       "tflops": round(gf * B2 / dt / 1e3, 1)}
            g6.close()
        out["upstream_layout_6x5"] = dict(up, note="ARU-Net alone with scale_space_num 6 / num_scales_att 5 (the upstream paper layout), "
                                                   f"{B2} pages per call at {W}x{H}; the default line's net is 5 / 3")
        # ---- heading pipeline on one page: net + SWT + per-line features ----
        cfg = AruConfig()
        gh = AruGraph(init_aru_weights(cfg, 22), cfg)
        hh = gh.handle(dev.index or 0)
        _, ws = image_ops._workspace(dev.index or 0)
        d_gray = torch.from_numpy(pages_u8[0]).to(dev)
        d_swt = torch.empty(H, W, device=dev, dtype=torch.uint8)
        rng = np.random.default_rng(0)
        n_lines = 700
        x0 = rng.integers(40, W - 500, n_lines)
        y0 = rng.integers(40, H - 80, n_lines)
        boxes = np.stack([x0, y0, x0 + rng.integers(200, 450, n_lines), y0 + rng.integers(24, 40, n_lines)], axis=1).astype(np.int32)
        sw, hgt, flag = np.empty(n_lines, np.float32), np.empty(n_lines, np.int32), np.empty(n_lines, np.int32)
        sp = C.c_void_p(s)

        def heading_page():
            _lib.check(lib.asep_aru_forward_dev(hh, imgs[0].data_ptr(), H, W, prob[0].data_ptr(), u8[0].data_ptr(), None, 0.05, sp), "heading net")
            _lib.check(lib.asep_swt_distance_transform_dev(ws, d_gray.data_ptr(), H, W, d_swt.data_ptr(), sp), "swt")
            _lib.check(lib.asep_swt_line_features_dev(ws, d_swt.data_ptr(), H, W, n_lines, boxes.ctypes.data, sw.ctypes.data,
                                                      hgt.ctypes.data, flag.ctypes.data, sp), "line features")
        dt = _timed(heading_page, 5)
        out["heading_net_plus_swt_fusion"] = {"pages_per_s": round(1.0 / dt, 2), "ms_per_page": round(1e3 * dt, 3), "dtype": "f32/u8",
                                              "note": f"BASELINE configs[2]: heading ARU-Net at {W}x{H} + stroke-width distance "
                                                      f"transform + per-line statistics of {n_lines} text lines (host gets the statistics)"}
        gh.close()
        # ---- the relation nets alone ----
        g = graphs[0]
        N, E = g["num_nodes"], int(g["interacting_nodes"].shape[0])
        if visual_pages is not None:
            vg, vpages, vh, vw, P = visual_pages
            q = vpages[0]
            dt = _timed(lambda: gnn_io.gnn_forward_visual_dev(vg, q.N, q.E, q.d_edges, q.d_node_feat, q.d_edge_feat, q.d_image, vh, vw,
                                                              q.d_regions, P, q.d_num_points, q.R, None, q.d_probs_out, s,
                                                              dev.index or 0), 20, warmup=2)
            dtb = _timed(lambda: gnn_io.gnn_forward_visual_batch_dev(vg, vpages, vh, vw, P, s, dev.index or 0), 5, warmup=1)
            out["visual_gnn_vn7e2_shape"] = {
                "pages_per_s": round(1.0 / dt, 1), "us_per_page": round(1e6 * dt, 1), "dtype": "f32",
                "us_per_page_grouped": round(1e6 * dtb / len(vpages), 1), "pages_per_grouped_call": len(vpages),
                "step_kernel": gnn_io.step_mode(vg, dev.index or 0),
                "note": "BASELINE configs[3] as named (mixed_gnn_vn7e2 = visual net), alone on the chip: RU backbone on 683x1024 + "
                        "ROI max / compression + graph with 55 node features, 200 nodes / 20k edges / 40k pairs; one page per call "
                        "and (grouped) the step's pages in one asep_gnn_forward_visual_batch_dev call"}
        gcfg = GnnConfig()
        gg = gnn_io.GnnGraph(init_gnn_weights(gcfg, 1234), gcfg)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d_e, d_u, d_f = t(g["interacting_nodes"]), t(g["node_features"]), t(g["edge_features"])
        d_conf = torch.empty(N * N, 2, device=dev)
        hg = gg.handle(dev.index or 0)
        dt = _timed(lambda: _lib.check(lib.asep_gnn_forward_dev(hg, N, E, d_e.data_ptr(), d_u.data_ptr(), d_f.data_ptr(), N * N, None,
                                                                d_conf.data_ptr(), s), "gnn"), 50, warmup=3)
        out["geometric_gnn_7_features"] = {"us_per_page": round(1e6 * dt, 1), "dtype": "f32",
                                           "note": "the relation net without image input (round-2 headline workload), alone on the chip"}
        gg.close()
        # ---- BASELINE configs[4] as a whole step: bf16 ARU-Net + the visual relation net with its backbone on the bf16 kernels, the
        #      same step / timing code as the headline, in a child process of its own (>= 5 s timed at its rate), with its roofline ----
        if args.dtype == "f32s" and args.bf16_steps > 0:
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--dtype", "bf16", "--steps", str(args.bf16_steps), "--warmup", "2",
                   "--pages-per-step", str(args.pages_per_step), "--height", str(H), "--width", str(W), "--gnn", args.gnn,
                   "--no-cpu-baseline", "--no-secondary", "--event-steps", "2"]
            env = dict(os.environ, ASEP_BENCH_DEVICE=str(dev.index or 0))
            r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True, timeout=1200)
            lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            if r.returncode == 0 and lines:
                q = json.loads(lines[-1])
                rr, rd = q["roofline"] or {}, q.get("roofline_detail") or {}
                out["bf16_full_step"] = {
                    "pages_per_s": q["value"], "ms_per_step": q["ms_per_step"], "steps": q["steps"], "timed_region_s": q["config"]["timed_region_s"],
                    "dtype": "bf16", "workload": q["config"]["workload"],
                    "roofline": {k: rr.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes",
                                                        "hbm_frac", "mfma_frac", "whole_page_traffic_gb", "whole_page_hbm_frac",
                                                        "whole_page_executed_frac", "avg_launch_us", "timing", "traffic_source")},
                    "whole_page_algorithmic_gb": rd.get("whole_page_algorithmic_gb"),
                    "note": "same step as the headline with --dtype bf16 (child process; this process's fp32 buffers stay allocated beside it)"}
            else:
                out["bf16_full_step"] = {"error": f"bf16 child exited with {r.returncode}"}
        # ---- the headline step on the PLAIN fp32 kernels (compute_dtype "f32": v_mfma_f32_16x16x4_f32 / Winograd / vector ALU, what rounds 1-4
        #      reported as the headline): same step, same timing code, own process.  The headline itself runs the fp32 engine's default
        #      arithmetic since round 5 (compute_dtype "f32s", csrc/split_kernels.h: fp32 tensors / accumulation / results, every product of the
        #      >= 12-channel convolutions as six bf16 x bf16 partial products of the exact 3-way split; the same fp32 parity gates) ----
        if args.dtype == "f32s" and args.plain_steps > 0:
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--dtype", "f32", "--steps", str(args.plain_steps), "--warmup", "2",
                   "--pages-per-step", str(args.pages_per_step), "--height", str(H), "--width", str(W), "--gnn", args.gnn,
                   "--no-cpu-baseline", "--no-secondary", "--event-steps", "2"]
            env = dict(os.environ, ASEP_BENCH_DEVICE=str(dev.index or 0))
            r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True, timeout=1200)
            lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            if r.returncode == 0 and lines:
                q = json.loads(lines[-1])
                rr = q["roofline"] or {}
                out["plain_f32_full_step"] = {
                    "pages_per_s": q["value"], "ms_per_step": q["ms_per_step"], "steps": q["steps"], "timed_region_s": q["config"]["timed_region_s"],
                    "dtype": "f32", "workload": q["config"]["workload"],
                    "roofline": {k: rr.get(k) for k in ("bound", "kernel", "pipe", "achieved", "peak", "unit", "frac", "algorithmic_tflops",
                                                        "algorithmic_over_peak", "avg_launch_us", "timing", "share_of_gpu_time",
                                                        "whole_page_executed_frac")},
                    "note": "the headline step with compute_dtype f32: every product on the fp32 matrix / vector pipes (the rounds 1-4 headline); child process"}
            else:
                out["plain_f32_full_step"] = {"error": f"f32 child exited with {r.returncode}"}
        # ---- the same steps on ONE page lane (ASEP_LANES=1: rounds 1-5's default schedule).  Two lanes are the default since round 6 (the round-5
        #      driver run: f32s 135.8 -> 138.4, bf16 469.6 -> 487.5, outputs bit-identical); the roofline block's per-launch figures come from
        #      one-lane event passes either way (DESIGN_LESSONS 47, 49).  Throughput only, no event passes. ----
        if args.dtype == "f32s" and args.bf16_steps > 0 and "ASEP_LANES" not in os.environ:
            import subprocess
            lanes = {}
            for dt, st in (("f32s", max(10, args.steps // 3)), ("bf16", max(20, args.bf16_steps // 2))):
                cmd = [sys.executable, os.path.abspath(__file__), "--dtype", dt, "--steps", str(st), "--warmup", "2",
                       "--pages-per-step", str(args.pages_per_step), "--height", str(H), "--width", str(W), "--gnn", args.gnn,
                       "--no-cpu-baseline", "--no-secondary", "--kernel-timing", "none"]
                env = dict(os.environ, ASEP_BENCH_DEVICE=str(dev.index or 0), ASEP_LANES="1")
                r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True, timeout=1200)
                lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
                if r.returncode == 0 and lines:
                    q = json.loads(lines[-1])
                    lanes[dt] = {"pages_per_s": q["value"], "ms_per_step": q["ms_per_step"], "steps": q["steps"]}
                else:
                    lanes[dt] = {"error": f"child exited with {r.returncode}"}
            lanes["note"] = ("the headline step and the bf16 step with ASEP_LANES=1 (child processes; throughput only): the schedule of rounds 1-5. "
                             "The default is two page lanes for calls of >= 8 pages")
            out["one_page_lane_full_step"] = lanes
        # ---- files in, files out ----
        if args.e2e_pages > 0:
            # in a child process of its own, so that the leg's worker processes do not inherit this process's module state
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--e2e-leg", "--e2e-pages", str(args.e2e_pages), "--e2e-heading-pages", str(args.e2e_heading_pages), "--e2e-gnn-pages", str(args.e2e_gnn_pages), "--gnn", args.gnn, "--height", str(args.height),
                   "--width", str(args.width), "--e2e-device", str(dev.index or 0), "--dtype", args.dtype]
            r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, text=True, timeout=1800)
            lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            out["e2e_files"] = json.loads(lines[-1]) if r.returncode == 0 and lines else {"error": f"e2e leg exited with {r.returncode}"}
    except Exception as e:  # a secondary figure must never take the headline line down
        out["error"] = repr(e)
    return out


# every kernel of csrc/bf16_kernels.h that multiplies on v_mfma_f32_16x16x32_bf16 (tests/test_bench_host.py holds this list against the
# __global__ functions of that header: a new kernel that is missing here would be priced against the fp32 peak, 16 x too kind -- ADVICE r5)
BF16_MFMA_KERNELS = ("convb_kernel", "deconvb_kernel", "deconvb8_kernel", "res8f_kernel", "res8b_kernel", "res8w_kernel", "res8wb_kernel", "res16f_kernel", "res32_tail_kernel",
                     "resb_tail_kernel", "att_headb_kernel", "convr_kernel")
# `roofline.bound` names the pipe the dominant kernel is priced against
BOUND_OF_PIPE = {"bf16 MFMA": "mfma_bf16", "fp32 MFMA": "mfma_fp32"}


def bound_of(pipe, hbm_bound):
    if hbm_bound:
        return "hbm"
    if pipe.startswith("fp32 vector ALU"):
        return "valu_fp32"
    if pipe.startswith("bf16 MFMA, 6 split"):
        return "mfma_bf16_split6"
    return BOUND_OF_PIPE.get(pipe, "mfma")


def pipe_of(kernel, dtype):
    """-> (pipe, peak in TFLOP/s of fp32-equivalent products) of a kernel of the ARU-Net engine.
    fp32 MFMA and the fp32 vector ALU are ONE datapath on gfx950 (157.3 TFLOP/s either way, DESIGN lesson 15); a split-product kernel
    executes SIX bf16 MFMA products per fp32 product, so its fp32-equivalent peak is the dense bf16 peak / 6; the bf16 path's
    convolutions run at the dense bf16 peak."""
    if kernel.startswith(("convs_kernel", "convs16_kernel", "deconvs_kernel")):
        return "bf16 MFMA, 6 split products per fp32 product", PEAK_BF16_MFMA_TFLOPS / 6.0
    if dtype == "bf16" and kernel.startswith(BF16_MFMA_KERNELS):
        return "bf16 MFMA", PEAK_BF16_MFMA_TFLOPS
    if kernel.startswith(("res8v_", "deconv8v_kernel", "att_headv_kernel", "conv_c1out_kernel", "conv_c1_kernel", "combine_kernel")):
        return "fp32 vector ALU (v_pk_fma_f32: the fp32 MFMA's datapath and peak)", PEAK_F32_MFMA_TFLOPS
    return "fp32 MFMA", PEAK_F32_MFMA_TFLOPS


def build_roofline(args, dom, d_iso, d_situ, kernels, exec_flops_page, pages_per_s_gpu, peak_tf, pages_per_launch, n_prof, visual, B, H, W,
                   pipe_seconds_page=None):
    """-> (roofline, roofline_detail).  `roofline` holds at most 20 keys, the HBM ones before the event-timing details (a record that
    keeps the first keys of a nested object keeps the informative ones); everything else goes to `roofline_detail`.
      f32s / f32 (layout 5, round 5): bound "mfma": achieved = EXECUTED TFLOP/s of the dominant kernel (fp32-equivalent products it
            really multiplies / its average launch duration: a Winograd F(2x2,3x3) kernel executes 1/2.25 of its direct-convolution
            credit) against the peak of the pipe that kernel runs on (`pipe`, `peak`: 157.3 for the fp32 MFMA and the fp32 vector
            ALU, 2500 / 6 for a split-product kernel) -- never above 1.  The direct-convolution rate (SURVEY.md section 8d's
            2 H W k^2 Cin Cout / time) is carried beside it as `algorithmic_tflops` / `algorithmic_over_peak` (round 4 had it in `frac`).
      bf16: bound "hbm": achieved = ALGORITHMIC bytes per launch (inputs read once, outputs written once: the engine's shape
            arithmetic, asep_aru_profile_report "bytes") / average launch duration, against 8 TB/s.
    `traffic` = HBM bytes per launch by the PMC counters ((2 FETCH_SIZE + WRITE_SIZE) * 1024, separate rocprofv3 --pmc passes of the
    same workload, profiles/traffic_per_kernel*.json); traffic / algorithmic_bytes > 1 is re-fetched halo."""
    lead = d_situ or d_iso                      # `achieved` / `frac` are the in-situ figures when measured (the lower ones)
    hbm_bound = args.dtype == "bf16"
    calls = dom["calls"]
    algo_bytes = dom["bytes"] / calls
    pipe, pipe_peak = pipe_of((dom.get("members") or [dom["kernel"]])[0], args.dtype)
    rate = (lambda d: d["algo_gbs"]) if hbm_bound else (lambda d: d["executed_tflops"])
    peak = PEAK_HBM_GBS if hbm_bound else pipe_peak
    total_ms = sum(k["total_ms"] for k in kernels)
    if pipe_seconds_page is None:               # (callers without per-kernel pipes: everything priced against peak_tf)
        pipe_seconds_page = exec_flops_page / (peak_tf * 1e12)
    r = {
        "bound": bound_of(pipe, hbm_bound), "kernel": dom["kernel"],
        "achieved": round(rate(lead), 1 if hbm_bound else 3), "peak": round(peak, 2), "unit": "GB/s" if hbm_bound else "TFLOP/s",
        "frac": round(rate(lead) / peak, 4),
        "traffic": None, "algorithmic_bytes": round(algo_bytes), "hbm_frac": None,
        "whole_page_traffic_gb": None, "whole_page_hbm_frac": None,
        "timing": ("in situ, one page lane (event passes run on one lane)" if d_situ else "isolated"),
        "frac_in_situ": round(rate(d_situ) / peak, 4) if d_situ else None,
        "frac_isolated": round(rate(d_iso) / peak, 4) if d_iso else None,
        **({"mfma_frac": round(lead["executed_tflops"] / peak_tf, 4)} if hbm_bound else
           {"pipe": pipe[:118], "algorithmic_tflops": round(lead["tflops"], 3)}),
        "avg_launch_us": round(lead["avg_us"], 2),
        **({"launches_per_step": calls / n_prof} if hbm_bound else {}),
        # the whole page: the time its kernels' executed products would take with every pipe at its peak / the measured time per page
        "whole_page_executed_frac": round(pipe_seconds_page * pages_per_s_gpu, 4),
        "share_of_gpu_time": round(dom["total_ms"] / total_ms, 4),
        "traffic_source": None,
    }
    detail = {
        "layout": 6,
        "avg_launch_us_in_situ": round(d_situ["avg_us"], 2) if d_situ else None,
        "avg_launch_us_isolated": round(d_iso["avg_us"], 2) if d_iso else None,
        "launch_population": "all launches of this kernel in a step" + (": the page net's and the relation nets' backbone's" if visual else ""),
        "pipe": pipe, "pipe_peak_tflops": round(pipe_peak, 2),
        "flops_per_launch": dom["flops"] / calls, "executed_flops_per_launch": dom["executed_flops"] / calls,
        "algorithmic_tflops": round(lead["tflops"], 3), "executed_tflops": round(lead["executed_tflops"], 3),
        # the direct-convolution credit of the kernel over its pipe's peak: above 1 is what a Winograd transform buys, not a roofline figure
        "algorithmic_over_peak": round(lead["tflops"] / pipe_peak, 4),
        "algorithmic_gbs": round(lead["algo_gbs"], 1),
        "event_timed_steps": n_prof * ((d_iso is not None) + (d_situ is not None)),
        "launches_per_step": calls / n_prof,
        # a launch of the page net carries at most 12 problems = 4 pages x 3 scales; the level-0 block kernels are launched
        # exactly once per such group, so they count the groups
        "pages_per_launch": pages_per_launch,
        # executed FLOPs (fp32-equivalent products) of ALL ARU-Net kernels of a page (incl. the relation net's backbone)
        "whole_page_executed_gflop": round(exec_flops_page / 1e9, 2),
        "whole_page_executed_tflops": round(exec_flops_page * pages_per_s_gpu / 1e12, 3),
        "whole_page_pipe_seconds_at_peak": pipe_seconds_page,
        "whole_page_algorithmic_gb": round(sum(k["bytes"] for k in kernels) / (B * n_prof) / 1e9, 3),
        "mfma_peak": peak_tf, "hbm_peak_gbs": PEAK_HBM_GBS, "hbm_achievable_gbs": ACHIEVABLE_HBM_GBS,
    }
    if dom.get("members"):                            # the level-0 blocks as one entry: each member's own figures
        by_name = {k["kernel"]: k for k in kernels}
        detail["members"] = [{"kernel": n, "calls": by_name[n]["calls"], "avg_us": round(by_name[n]["avg_us"], 2),
                              "frac": round((by_name[n]["algo_gbs"] if hbm_bound else by_name[n]["executed_tflops"]) / peak, 4)}
                             for n in dom["members"] if n in by_name]
    tp = os.path.join(ROOT, "profiles", "traffic_per_kernel.json" if args.dtype == "f32" else f"traffic_per_kernel_{args.dtype}.json")
    # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/README.md, scripts/make_traffic_json.py).  The counters cannot
    # be read from inside this process: the figure comes from the committed summary of the SAME workload (same pages per step, same
    # relation net, same dtype and page size = the same launch population); anything else is reported as the reason, not as a number
    try:
        tj_all = json.load(open(tp))
        want = {"dtype": args.dtype, "pages_per_step": B, "relation_net": "none" if args.no_gnn else args.gnn, "height": H, "width": W}
        diff = {k: (tj_all.get(k), v) for k, v in want.items() if tj_all.get(k) != v}
        mem = dom.get("members") or [dom["kernel"]]
        tjs = [tj_all["kernels"].get(n) for n in mem]
        tj = None
        if all(tjs):                                  # an "a+b" entry: the mean launch of the family, weighted by the counted dispatches
            nd = sum(t["dispatches"] for t in tjs)
            tj = {"bytes_per_launch": sum(t["bytes_per_launch"] * t["dispatches"] for t in tjs) / nd}
        if diff:
            r["traffic_source"] = ("no counters for this workload: " + ", ".join(f"{k} {a} != {b}" for k, (a, b) in diff.items()))[:118]
        elif not tj:
            r["traffic_source"] = f"{os.path.basename(tp)} has no row for this kernel"
        else:
            r["traffic"] = round(tj["bytes_per_launch"])
            src = str(tj_all.get("source"))
            tracked = src.startswith("profiles/") and os.path.exists(os.path.join(ROOT, src))
            r["traffic_source"] = (f"offline PMC: {src} @ {tj_all.get('commit', 'n/a')}" + ("" if tracked else " (being collected)"))[:118]
            tbs = r["traffic"] / (lead["avg_us"] * 1e-6) / 1e12
            r["hbm_frac"] = round(tbs / (PEAK_HBM_GBS / 1e3), 4)
            r["whole_page_traffic_gb"] = round(tj_all["page_bytes"] / 1e9, 2)
            r["whole_page_hbm_frac"] = round(tj_all["page_bytes"] * pages_per_s_gpu / 1e9 / PEAK_HBM_GBS, 4)
            detail["hbm_tb_per_s"] = round(tbs, 3)
            detail["traffic_over_algorithmic"] = round(r["traffic"] / algo_bytes, 3) if algo_bytes else None
    except Exception as e:  # noqa: BLE001 -- the line must say why the counters are missing
        r["traffic_source"] = f"{os.path.basename(tp)}: {e!r}"[:118]
    return r, detail


def main():
    global torch
    args = parse_args()
    if args.e2e_leg:                                         # child process of the secondary files-in / files-out figure
        print(json.dumps(e2e_files(args, args.e2e_device)))
        return
    if args.cpu_baseline_full:                               # the optional full-protocol CPU baseline: no GPU work in this mode
        q = run_cpu_baseline_full(args)
        with open(args.cpu_baseline_full, "w") as f:
            json.dump(q, f, indent=1)
        print(json.dumps(q))
        return
    if args.e2e_n_leg:                                       # child process of the N-owner files-in / files-out figure (--gpus N)
        print(json.dumps(e2e_files_n(args, args.e2e_n_leg)))
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:     # no torchrun environment: start the ranks ourselves (a child process)
        raise SystemExit(spawn_ranks(args))
    import torch
    if args.no_kernel_timing:
        args.kernel_timing = "none"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU")
    # ASEP_BENCH_FORCE_DIST=1 runs the RCCL code path (init, weight broadcast, barrier, max-reduction) with one rank too
    distributed = world > 1 or os.environ.get("ASEP_BENCH_FORCE_DIST") == "1"

    # ---- CPU baseline first (rank 0, N == 1), BEFORE this process touches the GPU: the CPU oracle (a port of the
    #      reference graph, kind="port") in worker processes on this box's host cores, bounded sample -------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(args)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    # ASEP_BENCH_DEVICE pins every rank to one device (two-rank test of the N > 1 code path on a one-GPU box); a box with fewer
    # devices than ranks deals the ranks round-robin over what it has (the line then says so: config.devices)
    ndev = max(1, torch.cuda.device_count())
    dev_index = int(os.environ["ASEP_BENCH_DEVICE"]) if "ASEP_BENCH_DEVICE" in os.environ else local_rank % ndev
    ranks_share_devices = world > 1 and ("ASEP_BENCH_DEVICE" in os.environ or ndev < world)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    stdout_fd = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL writes a version banner to STDOUT when its first communicator is created; stdout carries exactly one JSON
        # line, so fd 1 points at stderr until the communicator exists (restored after the weight broadcast below)
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
        # two ranks on one device (ASEP_BENCH_DEVICE) cannot form an RCCL communicator: that test uses gloo for the
        # broadcast / barrier / max-reduction, everything else is the same code
        backend = os.environ.get("ASEP_BENCH_BACKEND", "gloo" if ranks_share_devices else "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from citlab_article_separation_new_amd import _lib, gnn_io, sharding, synth
    from citlab_article_separation_new_amd.config import AruConfig, GnnConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights, init_gnn_weights, pack_blob, unpack_blob
    from citlab_article_separation_new_amd.net_post_processing_helper import AruGraph
    from citlab_article_separation_new_amd.gnn_io import GnnGraph

    H, W, B = args.height, args.width, args.pages_per_step
    visual = args.gnn == "visual" and not args.no_gnn
    aru_cfg = AruConfig(compute_dtype=args.dtype)
    # --dtype bf16 = configs[4] "bf16 convs": the relation net's conv backbone runs on the bf16 kernels too (graph stays fp32)
    gnn_cfg = (GnnConfig(visual_dims=[16, 16, 16], mvn=True, visual_layers=VISUAL_LAYERS, backbone={"compute_dtype": args.dtype})
               if visual else GnnConfig())

    # ---- weights: rank 0 creates them, every other rank receives the blob over RCCL (the only collective) ----
    if rank == 0:
        blobs = [pack_blob(init_aru_weights(aru_cfg, 1234)), pack_blob(init_gnn_weights(gnn_cfg, 1234))]
    else:
        blobs = [None, None]
    if distributed:
        bdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        blobs = [sharding.broadcast_blob(b or b"", rank, bdev) for b in blobs]
        dist.barrier()
        torch.cuda.synchronize()
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        os.close(stdout_fd)
    aru = AruGraph(unpack_blob(blobs[0]), aru_cfg)
    gnn = GnnGraph(unpack_blob(blobs[1]), gnn_cfg)       # visual: holds the backbone's aru_net/... tensors too
    lib = _lib.init_device(dev_index)
    h_aru, h_gnn = aru.handle(dev_index), gnn.handle(dev_index)

    # ---- synthetic inputs, resident in HBM before the timed region ------------------------------------------
    # (the page generator costs ~3 s of numpy per page: four distinct pages per rank through a per-seed file cache that ranks
    # and runs on one box share; every page of the batch has its own buffers in HBM)
    distinct = [synth.cached_synth_page(rank * 4 + k, W, H) for k in range(min(B, 4))]
    pages_u8 = [distinct[k % len(distinct)] for k in range(B)]
    imgs = [torch.from_numpy(p).to(dev).float().div_(255.0).contiguous() for p in pages_u8]
    graphs = [synth.synth_graph(rank * B + k) for k in range(B)]
    N = graphs[0]["num_nodes"]
    g_edges = [torch.from_numpy(g["interacting_nodes"]).to(dev) for g in graphs]
    g_u = [torch.from_numpy(g["node_features"]).to(dev) for g in graphs]
    g_ef = [torch.from_numpy(g["edge_features"]).to(dev) for g in graphs]
    E = [int(g["interacting_nodes"].shape[0]) for g in graphs]
    ncls = aru_cfg.n_classes
    out_prob = [torch.empty(H, W, ncls, device=dev) for _ in range(B)]
    out_u8 = [torch.empty(H, W, ncls, device=dev, dtype=torch.uint8) for _ in range(B)]
    out_mask = [torch.empty(H, W, ncls, device=dev, dtype=torch.uint8) for _ in range(B)]
    out_conf = [torch.empty(N * N, gnn_cfg.num_classes, device=dev) for _ in range(B)]
    vpages, vkeep, vh, vw, VP = None, [], 0, 0, 4
    if visual:
        # image feeds of the visual net: the page after the input pipeline's resize (3000 x 4500 -> 683 x 1024, values 0..255)
        # and one rectangular region per text block
        small = [synth.visual_inputs(p, N, rank * 4 + k) for k, p in enumerate(distinct)]
        vh, vw = small[0][0].shape
        vpages = (_lib.GnnPage * B)()
        for k in range(B):
            im, reg, npts = small[k % len(small)]
            t = [torch.from_numpy(im).to(dev), torch.from_numpy(reg).to(dev), torch.from_numpy(npts).to(dev)]
            vkeep.append(t)
            q = vpages[k]
            q.N, q.E, q.R = N, E[k], N * N
            q.d_edges, q.d_node_feat, q.d_edge_feat = g_edges[k].data_ptr(), g_u[k].data_ptr(), g_ef[k].data_ptr()
            q.d_image, q.d_regions, q.d_num_points = t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr()
            q.d_relations, q.d_probs_out = None, out_conf[k].data_ptr()
    stream = torch.cuda.current_stream().cuda_stream
    # the relation graphs do not depend on the segmentation of the same step (different command lines in the pipeline): they
    # run on a second stream and fill the launch tails of the CNN
    gnn_torch_stream = torch.cuda.Stream()
    gnn_stream = gnn_torch_stream.cuda_stream

    PtrArr = C.c_void_p * B
    p_img = PtrArr(*[t.data_ptr() for t in imgs])
    p_out = PtrArr(*[t.data_ptr() for t in out_prob])
    p_u8 = PtrArr(*[t.data_ptr() for t in out_u8])
    p_mask = PtrArr(*[t.data_ptr() for t in out_mask])

    def step(with_gnn=True):
        # one batched ARU-Net call: every layer is launched once for all B pages x 3 scale-space levels
        _lib.check(lib.asep_aru_forward_batch_dev(h_aru, B, p_img, H, W, p_out, p_u8, p_mask, 0.05, stream),
                   "asep_aru_forward_batch_dev")
        if not with_gnn or args.no_gnn:
            return
        if visual:
            # the step's relation nets: ONE grouped backbone forward over the B resized page images, then ROI + graph per page
            _lib.check(lib.asep_gnn_forward_visual_batch_dev(h_gnn, B, vpages, vh, vw, VP, gnn_stream), "asep_gnn_forward_visual_batch_dev")
        else:
            for k in range(B):
                _lib.check(lib.asep_gnn_forward_dev(h_gnn, N, E[k], g_edges[k].data_ptr(), g_u[k].data_ptr(),
                                                    g_ef[k].data_ptr(), N * N, None, out_conf[k].data_ptr(), gnn_stream),
                           "asep_gnn_forward_dev")

    def sync_all():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    if distributed:
        dt = sharding.max_over_ranks(dt, dev if dist.get_backend() == "nccl" else torch.device("cpu"))
    pages = world * B * args.steps
    value = pages / dt

    # ---- per-kernel timing with HIP events on the launch streams (same workload, separate passes so that the events do not
    #      perturb `value`): "in situ" = the real schedule (attention branch on its side stream, relation nets on theirs: what
    #      rocprofv3 sees), "isolated" = everything serialised on one stream, nothing beside the bracketed kernel ----------
    peak_tf = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS   # (f32s: fp32-equivalent FLOPs against the fp32 matrix peak;
                                                                                         #  what its kernels execute on the bf16 pipe is bf16_mfma_frac)

    h_bb = gnn._backbones[dev_index].handle(dev_index) if visual else None

    def kernel_pass(mode):
        """One event-timed pass.  mode 3 (in situ): the real schedule, both handles recording.  mode 1 (isolated): the net on one
        stream, then the relation nets' grouped call, each alone on the chip.  A kernel's record sums ALL its launches of a step
        by name -- the page net's and the (much smaller) ones of the relation nets' backbone -- which is exactly the population
        rocprofv3 averages over, so the two can be compared without a name table."""
        handles = [h_aru] + ([h_bb] if h_bb else [])
        for h in handles:
            lib.asep_aru_profile(h, mode)
        n = max(1, args.event_steps)
        for _ in range(n):
            if mode == 3:
                step()
            else:
                step(with_gnn=False)
                torch.cuda.synchronize()
                if visual:
                    _lib.check(lib.asep_gnn_forward_visual_batch_dev(h_gnn, B, vpages, vh, vw, VP, gnn_stream), "visual batch")
            torch.cuda.synchronize()
        merged, main_calls = {}, {}
        for h in handles:
            buf = C.create_string_buffer(1 << 16)
            _lib.check(lib.asep_aru_profile_report(h, buf, len(buf)), "asep_aru_profile_report")
            lib.asep_aru_profile(h, 0)
            for k in json.loads(buf.value.decode()):
                if h == h_aru:
                    main_calls[k["kernel"]] = k["calls"]
                m = merged.setdefault(k["kernel"], {"kernel": k["kernel"], "calls": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0, "xflops": 0.0})
                m["calls"] += k["calls"]; m["total_ms"] += k["total_ms"]; m["flops"] += k["flops"]; m["bytes"] += k.get("bytes", 0.0)
                m["xflops"] += k.get("executed_flops", k["flops"])
        for k in merged.values():
            k["avg_us"] = 1e3 * k["total_ms"] / k["calls"]
            k["tflops"] = k["flops"] / (k["total_ms"] * 1e-3) / 1e12 if k["total_ms"] > 0 else 0.0
            # Winograd F(2x2,3x3) kernels are credited with the direct-convolution FLOPs of their layers (the algorithmic
            # work) but execute 2.25x fewer multiplications on the MFMA: report both
            # (other kernels that execute fewer products than their credit say so themselves: the engine's "executed_flops", e.g. the
            #  logits conv of combine_kernel on the difference filter of two classes behind a soft-max -- ADVICE r5)
            k["executed_flops"] = k["xflops"] / 2.25 if "wino" in k["kernel"] else k["xflops"]
            k["executed_tflops"] = k["executed_flops"] / (k["total_ms"] * 1e-3) / 1e12 if k["total_ms"] > 0 else 0.0
            # split-product kernels (f32s): every fp32 product is SIX bf16 products on the bf16 pipe; executed_* stays the fp32-equivalent
            # figure (1 x), bf16_tflops is what the bf16 matrix pipeline executes
            k["bf16_tflops"] = 6.0 * k["tflops"] if k["kernel"].startswith(("convs_kernel", "convs16_kernel", "deconvs_kernel")) else 0.0
            k["pipe"], k["pipe_peak"] = pipe_of(k["kernel"], args.dtype)
            # ALGORITHMIC bytes per launch (every input read once, every output written once: the engine's shape arithmetic) / time
            k["algo_gbs"] = k["bytes"] / (k["total_ms"] * 1e-3) / 1e9 if k["total_ms"] > 0 else 0.0
        return merged, n, main_calls

    roofline = roofline_detail = None
    kernels = []
    if rank == 0 and args.kernel_timing != "none":
        iso = situ = None
        if args.kernel_timing in ("both", "isolated"):
            iso, n_prof, main_calls = kernel_pass(1)
        if args.kernel_timing in ("both", "in-situ"):
            situ, n_prof, main_calls = kernel_pass(3)
        # the dominant entry = the largest summed launch time IN SITU (the real schedule on one page lane: attention branch on its side
        # stream, relation nets beside the page net), the two level-0 blocks counted as one entry; its isolated figures ride beside it.
        # (An event pair in situ brackets what a launch waits behind other streams' work as well; rocprofv3's tracer serialises part of
        # that overlap, so its per-kernel averages lie between the isolated and the in-situ ones: DESIGN.md section 5.)
        entries = rank_kernels(iso, situ)
        kernels = sorted((situ or iso).values(), key=lambda k: -k["total_ms"])          # the per-kernel table of the line
        dom = next((k for k in entries if k["kernel"] == args.dominant), entries[0])
        d_iso, d_situ = entry_of(iso, dom), entry_of(situ, dom)
        # (the level-0 up block of the PAGE net -- res8v_up_kernel / res8_up_kernel in fp32, res8f_kernel<true> / res8b in bf16 -- is
        # launched exactly once per group of pages: its call count in the page net's own profile counts the groups)
        groups = next((c for name, c in main_calls.items()
                       if name.startswith("res8") and ("_up_" in name or name.endswith("<true>"))), dom["calls"])
        exec_flops_page = sum(k["executed_flops"] for k in kernels) / (B * n_prof)
        pipe_seconds_page = sum(k["executed_flops"] / (k["pipe_peak"] * 1e12) for k in kernels) / (B * n_prof)
        roofline, roofline_detail = build_roofline(args, dom, d_iso, d_situ, kernels, exec_flops_page, value / world, peak_tf,
                                                   B * n_prof / groups, n_prof, visual, B, H, W, pipe_seconds_page)
        for k in kernels:
            o_s, o_i = (situ or {}).get(k["kernel"]), (iso or {}).get(k["kernel"])
            k["avg_us_in_situ"] = o_s["avg_us"] if o_s else None
            k["avg_us_isolated"] = o_i["avg_us"] if o_i else None

    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary and not args.no_gnn:
        # the secondary figures start GPU processes of this one's size (plain fp32 step: 96 GB of arenas): this process gives its arenas back first
        # (the model stays loaded; the agreement check below rebuilds what it needs)
        torch.cuda.synchronize()
        for h in [h_aru] + ([h_bb] if h_bb else []):
            _lib.check(lib.asep_aru_trim(h), "asep_aru_trim")
        secondary = secondary_measurements(args, lib, dev, imgs, pages_u8, graphs, (gnn, vpages, vh, vw, VP) if visual else None)

    # --gpus N: the files-in / files-out path with N GPU owners, as a child of rank 0 (the other ranks wait at the final barrier; their
    # buffers stay allocated -- 288 GB per device leave room for both)
    if rank == 0 and world > 1 and not args.no_secondary and args.e2e_n_pages_per_owner > 0:
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--e2e-n-leg", str(world), "--e2e-n-pages-per-owner", str(args.e2e_n_pages_per_owner),
               "--height", str(H), "--width", str(W), "--dtype", args.dtype]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE",
                                                                  "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
        if ranks_share_devices:                                  # (test boxes: the owners share the devices the ranks share)
            env["ASEP_BENCH_OWNERS"] = ",".join(str(int(os.environ["ASEP_BENCH_DEVICE"]) if "ASEP_BENCH_DEVICE" in os.environ else r % ndev)
                                                for r in range(world))
        try:
            r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True, timeout=1500)
            lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            leg = json.loads(lines[-1]) if r.returncode == 0 and lines else {"error": f"N-owner leg exited with {r.returncode}"}
        except Exception as e:  # noqa: BLE001 -- a secondary figure must never take the headline line down
            leg = {"error": repr(e)}
        secondary = dict(secondary or {}, e2e_files_n=leg)

    # --dtype f32s: the step's own outputs against the plain fp32 kernels of the same library on the same pages (in the measured process, after
    # the timed region): what "fp32 with split products" means for THIS run's numbers, not only for the test suite's
    agreement = None
    if rank == 0 and args.dtype == "f32s":
        step(with_gnn=False)
        torch.cuda.synchronize()
        g32 = AruGraph(aru.tensors, AruConfig(compute_dtype="f32"))
        ref32 = torch.empty(H, W, ncls, device=dev)
        worst, flips = 0.0, 0
        for k in range(min(B, 2)):
            _lib.check(lib.asep_aru_forward_dev(g32.handle(dev_index), imgs[k].data_ptr(), H, W, ref32.data_ptr(), None, None, 0.05, stream), "asep_aru_forward_dev")
            torch.cuda.synchronize()
            worst = max(worst, float((out_prob[k] - ref32).abs().max()))
            flips += int(((out_prob[k] * 255.0).to(torch.uint8) != (ref32 * 255.0).to(torch.uint8)).sum())
        agreement = {"pages": min(B, 2), "max_abs_dp_vs_plain_fp32_kernels": worst, "uint8_values_that_differ": flips,
                     "of": min(B, 2) * H * W * ncls, "gate_of_the_fp32_parity_tests": 1e-4}
        g32.close()
        del ref32

    if rank == 0:
        gnn_flops = 0.0
        if not args.no_gnn:
            gnn_flops = lib.asep_gnn_flops(h_gnn, N, 2 * E[0], N * N)
            if visual:
                gnn_flops += lib.asep_aru_flops(gnn._backbones[dev_index].handle(dev_index), vh, vw)
        flops_page = lib.asep_aru_flops(h_aru, H, W) + gnn_flops
        engine_reads = set(lib.asep_engine_switches().decode().split())
        if args.no_gnn:
            rel = "ARU-Net only (diagnostic)"
        elif visual:
            rel = ("+ per page the relation net BASELINE configs[3] names (mixed_gnn_vn7e2 = VISUAL net: RU backbone "
                   f"({'bf16 convs' if args.dtype == 'bf16' else 'fp32, like the page net'}) on the page at {vw}x{vh} + ROI max / compression to 3x16 visual features + graph with 55 node features, 200 nodes / 20k edges / "
                   "40k pairs), the step's relation nets as one grouped call on a second stream")
        else:
            rel = "+ geometric 7-feature relation graph per page (200 nodes / 20k edges / 40k pairs; diagnostic: not the net configs[3] names)"
        line = {
            "metric": METRIC, "value": round(value, 4), "unit": "pages/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {
                # (short on purpose: records that cut strings at 120 characters keep it whole; the long form is workload_detail)
                "workload": (({"f32": "configs[1] fp32", "bf16": "configs[4] bf16 convs", "f32s": "configs[1] fp32 (split products)"}[args.dtype]) + f": ARU-Net on {W}x{H} pages + "
                             + ("no relation net" if args.no_gnn else
                                ("configs[3] visual GNN (mixed_gnn_vn7e2)" if visual else "geometric GNN") + " per page")),
                # what the dtype label stands for
                "arithmetic": {"f32s": "fp32 tensors / fp32 accumulate / fp32 results; products of the >= 12-channel convolutions = 6 bf16 MFMAs of "
                                       "the exact 3-way bfloat16 split of both factors (dropped terms <= 2^-23 |x w|); level 0, first / last layers, "
                                       "deconvolutions: fp32 FMA / fp32 MFMA; held to the fp32 parity gates (tests/test_full_frame_gpu.py)",
                               "f32": "fp32 tensors; every product on the fp32 MFMA (v_mfma_f32_16x16x4_f32, Winograd F(2x2,3x3) from 32 channels) or "
                                      "the fp32 vector ALU (level 0)",
                               "bf16": "bf16 tensors and MFMA operands (v_mfma_f32_16x16x32_bf16), fp32 accumulate; image, attention maps, logits and "
                                       "probabilities fp32"}[args.dtype],
                "workload_detail": (("BASELINE configs[1]: ARU-Net separator detection on 3000x4500 px pages, fp32 (plain fp32 MFMA / vector-ALU kernels), "
                                     if args.dtype == "f32" else
                                     "BASELINE configs[1], fp32 engine default (fp32 tensors / accumulation / results; products of the wide convolutions "
                                     "as 6 bf16 MFMAs on the 3-way bfloat16 split of both factors): ARU-Net separator detection on 3000x4500 px pages, "
                                     if args.dtype == "f32s" else
                                     "BASELINE configs[4] precision (bf16 activations / MFMA convs, fp32 accumulate): ARU-Net separator "
                                     "detection on 3000x4500 px pages, ") + rel),
                "height": H, "width": W, "pages_per_step_per_gpu": B, "sharding": f"pages over {world} rank(s)",
                "devices": ndev if not ranks_share_devices else f"{min(ndev, world)} (ranks share devices: {world} ranks)",
                "relation_net": "none" if args.no_gnn else args.gnn,
                "timed_region_s": round(dt, 3),
                **({"agreement_with_plain_fp32": agreement} if agreement else {}),
                "aru_cfg": "ARU featRoot=8 levels=5 res_depth=3 att_scales=3 n_classes=2 (secondary.upstream_layout_6x5: levels=6 att_scales=5)",
                # engine / bench switches of the environment this line was measured under (none = the defaults the documents describe)
                # (only what this build of the library reads, asep_engine_switches(); a set-but-ignored ASEP_* name is listed apart)
                "engine_switches": {k: v for k, v in sorted(os.environ.items()) if k in engine_reads},
                "ignored_asep_variables": sorted(k for k in os.environ if k.startswith("ASEP_") and k not in engine_reads and k not in HOST_VARIABLES),
                "gflop_per_page": round(flops_page / 1e9, 2),
                "whole_page_tflops_per_gpu": round(flops_page * value / world / 1e12, 3),
            },
            "roofline": roofline, "cpu_baseline": cpu_baseline, "roofline_detail": roofline_detail,
            "kernels": [{"kernel": k["kernel"], "calls": k["calls"], "avg_us": round(k["avg_us"], 2),
                         "avg_us_in_situ": None if k.get("avg_us_in_situ") is None else round(k["avg_us_in_situ"], 2),
                         "avg_us_isolated": None if k.get("avg_us_isolated") is None else round(k["avg_us_isolated"], 2),
                         "flops": k["flops"], "executed_flops": k["executed_flops"], "tflops": round(k["tflops"], 2), "executed_tflops": round(k["executed_tflops"], 2),
                         "pipe_peak_tflops": round(k["pipe_peak"], 2), "executed_frac_of_pipe_peak": round(k["executed_tflops"] / k["pipe_peak"], 4),
                         "bytes": k["bytes"], "algorithmic_gbs": round(k["algo_gbs"], 1),
                         # split-product kernels: six bf16 products per fp32 product, against the bf16 matrix peak
                         **({"bf16_mfma_frac": round(k["bf16_tflops"] / PEAK_BF16_MFMA_TFLOPS, 4)} if k.get("bf16_tflops") else {})}
                        for k in kernels],
            "secondary": secondary,
        }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
