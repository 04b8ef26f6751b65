"""bench.py contract on a GPU box: ONE JSON line on stdout carrying the driver's keys, with and without the RCCL code path
(torch.distributed.run with one rank and ASEP_BENCH_FORCE_DIST=1: init, weight broadcast, barrier, max-reduction)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline"}


def _free_port():
    """a port nobody listens on right now (fixed ports collide between concurrent sessions / stale listeners)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must hold exactly one line, got {len(lines)}: {lines[:3]}"
    line = json.loads(lines[0])
    assert KEYS <= set(line), KEYS - set(line)
    assert line["unit"] == "pages/s" and line["value"] > 0 and line["n_gpus"] == 1 and line["scaling"] == "weak"
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    # what a record keeps when it cuts nested objects at 24 keys and strings at 120 characters
    assert len(line["roofline"]) <= 20 and list(line["roofline"]).index("whole_page_hbm_frac") < 12
    assert len(line["config"]["workload"]) < 120 and all(len(v) < 120 for v in line["roofline"].values() if isinstance(v, str))
    return line


def r_src(line):
    return line["roofline"]["traffic_source"]


def test_single_process_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--pages-per-step", "3", "--e2e-pages", "6", "--bf16-steps", "3", "--plain-steps", "2"], cwd=ROOT, capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    # round 5: the default line is the fp32 engine's default arithmetic (f32s), spelled out in config.arithmetic; the plain fp32 kernels' step
    # rides beside it as secondary.plain_f32_full_step
    assert line["steps"] == 2 and line["warmup"] == 1 and line["dtype"] == "f32s" and "6 bf16 MFMAs" in line["config"]["arithmetic"]
    ag = line["config"]["agreement_with_plain_fp32"]
    assert ag["max_abs_dp_vs_plain_fp32_kernels"] <= 1e-5 and ag["uint8_values_that_differ"] <= 1e-4 * ag["of"]
    assert line["config"]["pages_per_step_per_gpu"] == 3 and line["roofline_detail"]["pages_per_launch"] == 3.0
    # secondary figures (bf16 variant, heading net + stroke-width fusion, the visual relation net) ride on the same line
    sec = line["secondary"]
    assert "error" not in sec, sec
    assert sec["aru_bf16_mfma"]["pages_per_s"] > 0 and sec["heading_net_plus_swt_fusion"]["pages_per_s"] > 0
    assert sec["visual_gnn_vn7e2_shape"]["step_kernel"] == "factored" and sec["visual_gnn_vn7e2_shape"]["us_per_page"] > 0
    assert sec["visual_gnn_vn7e2_shape"]["us_per_page_grouped"] > 0 and sec["geometric_gnn_7_features"]["us_per_page"] > 0
    assert sec["e2e_files"]["page_xml_written"] == 6 and sec["e2e_files"]["pages_per_s"] > 0
    # BASELINE configs[4] as a whole step (bf16 ARU-Net + visual relation net with a bf16 backbone), priced against HBM by algorithmic bytes
    b16 = sec["bf16_full_step"]
    assert "error" not in b16, b16
    assert b16["pages_per_s"] > 0 and b16["steps"] == 3 and b16["roofline"]["bound"] == "hbm" and b16["roofline"]["timing"].startswith("in situ, one page lane") and b16["roofline"]["unit"] == "GB/s"
    assert b16["roofline"]["algorithmic_bytes"] > 0 and 0 < b16["roofline"]["frac"] < 1 and b16["whole_page_algorithmic_gb"] > 1
    p32 = sec["plain_f32_full_step"]
    assert "error" not in p32, p32
    assert p32["pages_per_s"] > 0 and p32["steps"] == 2 and p32["dtype"] == "f32" and 0 < p32["roofline"]["frac"] <= 1.0
    tl = sec["one_page_lane_full_step"]                   # round 6: two page lanes are the default; the one-lane schedule of rounds 1-5 rides beside it
    assert tl["f32s"]["pages_per_s"] > 0 and tl["bf16"]["pages_per_s"] > 0 and "ASEP_LANES=1" in tl["note"]
    up = sec["upstream_layout_6x5"]
    assert up["f32s"]["pages_per_s"] > 0 and up["bf16"]["pages_per_s"] > up["f32s"]["pages_per_s"] and 1000 < up["f32s"]["gflop_per_page"] < 1120
    assert all("executed_tflops" in k and 0 <= k["executed_frac_of_pipe_peak"] <= 1.0 for k in line["kernels"])
    # the headline step carries the VISUAL relation net (BASELINE configs[3]: mixed_gnn_vn7e2), and the roofline block both timings
    assert line["config"]["relation_net"] == "visual" and "mixed_gnn_vn7e2" in line["config"]["workload"]
    assert line["config"]["devices"] >= 1 and "mixed_gnn_vn7e2" in line["config"]["workload_detail"]
    # algorithmic bytes of every launch (the engine's shape arithmetic) ride on the kernel table
    assert all(k["bytes"] > 0 for k in line["kernels"] if k["kernel"].startswith(("conv", "res8", "deconv", "combine")))
    # three pages at pages-per-step 3 are not the workload the committed counters were taken on: the line says so instead of a number
    assert r_src(line) and line["roofline"]["traffic"] is None
    r = line["roofline"]
    assert r["frac_in_situ"] and r["frac_isolated"] and r["whole_page_executed_frac"] > 0
    assert r["frac"] == r["frac_in_situ"] and r["frac_in_situ"] <= r["frac_isolated"] * 1.05 and r["bound"] in ("valu_fp32", "mfma_bf16_split6", "mfma_fp32")
    # the dominant kernel leads the IN-SITU totals (rocprofv3's ordering of the same command)
    # (entries: single kernels, the two level-0 blocks as one "down+up" entry)
    tot = lambda names: sum(k["calls"] * k["avg_us_in_situ"] for k in line["kernels"] if k["kernel"] in names)
    assert tot(r["kernel"].split("+")) >= max(k["calls"] * k["avg_us_in_situ"] for k in line["kernels"])
    if "+" in r["kernel"]:
        assert [m["kernel"] for m in line["roofline_detail"]["members"]] == r["kernel"].split("+")
    assert set(line["config"]["engine_switches"]) <= {"ASEP_LANES"} and line["config"]["ignored_asep_variables"] == []
    assert 0 < r["frac"] <= 1.0 and r["frac_isolated"] <= 1.0 and r["peak"] in (157.3, 416.67) and r["pipe"]   # executed products over the kernel's own pipe
    assert r["algorithmic_tflops"] >= r["achieved"] - 1e-3
    assert set(r["kernel"].split("+")) <= {k["kernel"] for k in line["kernels"]} and "<" in "".join(k["kernel"] for k in line["kernels"])


def test_rccl_path_with_one_rank_keeps_stdout_clean():
    env = dict(os.environ, ASEP_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-secondary", "--pages-per-step", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    _one_json_line(r.stdout)


def test_two_ranks_on_the_one_gpu():
    """the N > 1 branch of bench.py (rank-dependent page seeds, weight broadcast from rank 0, barrier brackets,
    max-over-ranks, pages = world x B x steps) had only ever run with one rank.  Two ranks through torch.distributed.run, both
    pinned to device 0 (ASEP_BENCH_DEVICE) -- RCCL cannot form a communicator of two ranks on one device, so the collective
    backend of this test is gloo; the ranks are started by torchrun before anything touches the GPU."""
    env = dict(os.environ, ASEP_BENCH_DEVICE="0", ASEP_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-secondary", "--pages-per-step", "2", "--kernel-timing", "none"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["sharding"] == "pages over 2 rank(s)" and line["cpu_baseline"] is None
    # value = pages of BOTH ranks / slowest rank's time
    assert abs(line["value"] - 2 * 2 * 2 / (line["ms_per_step"] * 2 / 1e3)) < 1e-2 * line["value"]


def test_gpus_2_without_a_torchrun_environment_starts_its_own_ranks():
    """VERDICT r3 next #1a: `python bench.py --gpus N` with WORLD_SIZE unset used to exit with a message; the scaling run may be
    started exactly like that.  The parent spawns torch.distributed.run as a child before touching the GPU and relays the child's
    one line and exit code.  On this one-GPU box the two ranks share device 0 (dealt round-robin over the visible devices, gloo
    for the broadcast / barrier / max-reduction because RCCL cannot form a communicator of two ranks on one device)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-secondary", "--pages-per-step", "2", "--kernel-timing", "none"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:3]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["sharding"] == "pages over 2 rank(s)" and line["cpu_baseline"] is None
    assert "ranks share devices" in str(line["config"]["devices"])
    assert abs(line["value"] - 2 * 2 * 2 / (line["ms_per_step"] * 2 / 1e3)) < 1e-2 * line["value"]


def test_gpus_2_reports_the_files_leg_with_two_gpu_owners():
    """VERDICT r4 missing #3 / next #5: a --gpus N run times device-resident pages only, and what decides the 1 -> N curve files in / files
    out is the host.  With --gpus N rank 0 also runs the separator command line with N GPU owners x (CPU quota / N) host workers on
    N x --e2e-n-pages-per-owner scans and reports it as secondary.e2e_files_n.  Here: two ranks and two owners on the one GPU of the box."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ASEP_BENCH_DEVICE"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--pages-per-step", "2", "--kernel-timing", "none", "--e2e-n-pages-per-owner", "12"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines[:3]
    line = json.loads(lines[0])
    leg = line["secondary"]["e2e_files_n"]
    assert "error" not in leg, leg
    assert leg["rc"] == 0 and leg["gpu_owners"] == 2 and leg["owner_devices"] == ["0", "0"] and leg["scans"] == 24 and leg["page_xml_written"] == 24
    assert leg["pages_per_s"] > 0 and len(leg["pages_per_s_per_owner"]) == 2 and 0 < leg["owner_device_stage_share"] <= 1.0
    assert leg["host_workers_per_owner"] * 2 <= leg["cpus_this_container_may_use"]
