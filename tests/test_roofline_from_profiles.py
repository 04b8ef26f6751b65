"""The `roofline` block of the bench line must follow from the rocprofv3 summaries committed under profiles/ (VERDICT r2, next #1):
scripts/roofline_from_profiles.py recomputes it from kernel_stats.csv (launch durations), pmc_summary.json (FETCH_SIZE / WRITE_SIZE,
FETCH doubled: the gfx950 correction of MI355X_MICROARCH.md) and the FLOP totals of the line, and the two must agree to 3 %.
Runs on the committed summaries of the fp32 headline build and of the bf16 build (CPU only)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import make_traffic_json  # noqa: E402
import roofline_from_profiles as rfp  # noqa: E402

# every round-3 measurement point that carries the full set of summaries (profiles/r3*: fp32 headline builds and their bf16 twins)
TAGS = sorted(t for t in os.listdir(os.path.join(ROOT, "profiles"))
              if t.startswith(("r3", "r4", "r5", "r6")) and all(os.path.exists(os.path.join(ROOT, "profiles", t, f))
                                            for f in ("kernel_stats.csv", "pmc_summary.json", "bench_under_trace.json", "bench.json")))


def test_there_are_committed_summaries_to_check():
    assert any(t.endswith("_bf16") for t in TAGS) and any(not t.endswith("_bf16") for t in TAGS), TAGS


def test_kernel_keys_are_the_full_rocprof_names():
    key = make_traffic_json.kernel_key
    assert key("void asep::conv_mfma_kernel<3, 3, 1, false, 16, false, false, false, 4>(asep::ConvArgs)") == \
        "conv_mfma_kernel<3,3,1,false,16,false,false,false,4>"
    assert key("void asep::conv_mfma_kernel<3, 3, 1, false, 16, false, false, false, 2>(asep::ConvArgs)") != \
        key("void asep::conv_mfma_kernel<3, 3, 1, false, 16, false, false, false, 4>(asep::ConvArgs)")       # r2: these two collided
    assert key("asep::res8v_up_kernel(asep::Res8Args)") == "res8v_up_kernel"
    assert key("void asep::combine_kernel<8, 2, false>(asep::CombineArgs)") == "combine_kernel<8,2,false>"


@pytest.mark.parametrize("tag", TAGS)
def test_roofline_block_is_reproducible_from_the_committed_summaries(tag):
    tagdir = os.path.join(ROOT, "profiles", tag)
    line, rec, table = rfp.recompute(tagdir)
    pairs, dev, ok = rfp.compare(line, rec, 0.03)
    assert ok, {k: (pairs[k], rec[k], dev[k]) for k in pairs}
    r = rfp.block(line)
    # the compared set covers the launch time, the achieved rate and fraction, the PMC traffic and the whole-page figures
    assert {"avg_launch_us", "achieved", "frac", "traffic", "hbm_frac", "whole_page_executed_frac", "whole_page_traffic_gb"} <= set(pairs)
    mem = r["kernel"].split("+")                             # round 6: the two level-0 blocks as one entry "down+up"
    assert r["timing"].startswith("in situ") and set(mem) <= {t["kernel"] for t in table}
    # the table joins every engine-side kernel name with a rocprofv3 row (no name table in between)
    engine = {k["kernel"] for k in line["kernels"]}
    assert engine <= {t["kernel"] for t in table}, engine - {t["kernel"] for t in table}
    # sanity against the definitions: traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 of the dominant kernel's PMC rows
    pmc = json.load(open(os.path.join(tagdir, "pmc_summary.json")))
    fs = [v for k, v in pmc["FETCH_SIZE"].items() if make_traffic_json.kernel_key(k) in mem]
    ws = [v for k, v in pmc["WRITE_SIZE"].items() if make_traffic_json.kernel_key(k) in mem]
    assert len(fs) == len(ws) == len(mem)
    # (the traced line carries the table of the FIRST profile pass of scripts/profile_round.sh, the directory holds the second
    # pass's counters: the same build and command, equal to a few 1e-4 .. 1e-3)
    mean = sum((2 * f["sum"] + w["sum"]) * 1024 for f, w in zip(fs, ws)) / sum(f["dispatches"] for f in fs)
    assert abs(mean / r["traffic"] - 1) < 5e-3


@pytest.mark.parametrize("tag", TAGS)
def test_untraced_line_of_the_same_build_uses_the_same_traffic_table(tag):
    """profiles/<tag>/bench.json is the default (un-traced) run: its traffic fields come from the table of the traced build and its
    fractions are arithmetic on its own timings (launch time x traffic -> TB/s -> fraction of 8 TB/s)."""
    line = rfp.load_line(os.path.join(ROOT, "profiles", tag, "bench.json"))
    r = rfp.block(line)
    assert r["traffic"] and r["frac_in_situ"] and r["frac_isolated"] and r["whole_page_executed_frac"]
    tb_s = r["traffic"] / (r["avg_launch_us"] * 1e-6) / 1e12
    if r.get("layout", 3) >= 4:
        # round 4: `achieved` of a bf16 line = ALGORITHMIC bytes per launch / launch time; the counter bytes give hbm_frac in both dtypes
        assert len(line["roofline"]) <= 20 and abs(tb_s / 8.0 / r["hbm_frac"] - 1) < 2e-3
        assert abs(r["achieved"] / r["peak"] / r["frac"] - 1) < 2e-3
        if r["bound"] == "hbm":
            assert abs(r["algorithmic_bytes"] / (r["avg_launch_us"] * 1e-6) / 1e9 / r["achieved"] - 1) < 2e-3 and r["mfma_peak"] == 2500.0
            assert r["traffic"] >= 0.98 * r["algorithmic_bytes"]          # the counters cannot see less than what must move
    elif r["bound"] == "hbm":
        assert abs(tb_s * 1e3 / r["achieved"] - 1) < 2e-3 and abs(r["achieved"] / r["peak"] / r["frac"] - 1) < 2e-3
        assert r["mfma"]["peak"] == 2500.0
    else:
        assert abs(tb_s / r["hbm_tb_per_s"] - 1) < 2e-3 and abs(r["achieved"] / r["peak"] / r["frac"] - 1) < 2e-3
        assert r["peak"] == 157.3
    # default run: 80 steps x 16 pages = 1280 pages -- >= 8 s of timed region at the fp32 rate; bf16: 2.99 .. 3.3 s up to
    # profiles/r3n_bf16 (80 steps), >= 8 s from then on (240 steps; profiles/r5final_bf16 ran them in 7.98 s at 481 pages/s: 300 steps since)
    assert line["config"]["timed_region_s"] >= (7.9 if line["dtype"] == "f32" or line["steps"] >= 240 else 2.9)


def test_summaries_are_committed():
    assert TAGS, "profiles/r3* with kernel_stats.csv / pmc_summary.json / bench_under_trace.json / bench.json expected"


def test_dominant_kernel_rule():
    """bench.rank_kernels (round 6): entries are ranked on the IN-SITU totals when that pass exists, and the two level-0 blocks are one
    entry "down+up" -- with the fp32 numbers of profiles/r3k and the bf16 ones; the family leads under the isolated, the in-situ and
    rocprofv3's ordering in all three arithmetics, which no single kernel does"""
    import bench
    rec = lambda name, calls, avg_us: {"kernel": name, "calls": calls, "total_ms": calls * avg_us * 1e-3, "flops": 1.0, "bytes": 1.0,
                                       "executed_flops": 1.0}
    iso = {k["kernel"]: k for k in (rec("conv_wino_kernel<4,false>", 216, 319.13), rec("res8v_up_kernel<0>", 18, 3857.94),
                                    rec("res8v_down_kernel<0>", 18, 2700.41))}
    situ = {k["kernel"]: k for k in (rec("conv_wino_kernel<4,false>", 216, 527.24), rec("res8v_up_kernel<0>", 18, 4003.94),
                                     rec("res8v_down_kernel<0>", 18, 4750.26))}
    for a, b in ((iso, situ), (iso, None), (None, situ)):
        top = bench.rank_kernels(a, b)
        assert [k["kernel"] for k in top] == ["res8v_down_kernel<0>+res8v_up_kernel<0>", "conv_wino_kernel<4,false>"]
        assert top[0]["calls"] == 36 and top[0]["members"] == ["res8v_down_kernel<0>", "res8v_up_kernel<0>"]
    e = bench.rank_kernels(iso, situ)[0]
    assert abs(e["avg_us"] - (4003.94 + 4750.26) / 2) < 1e-6 and abs(bench.entry_of(iso, e)["avg_us"] - (3857.94 + 2700.41) / 2) < 1e-6
    iso_b = {k["kernel"]: k for k in (rec("res8f_kernel<true>", 18, 1079.5), rec("convb", 162, 84.35), rec("res8f_kernel<false>", 18, 683.6))}
    situ_b = {k["kernel"]: k for k in (rec("res8f_kernel<true>", 18, 1203.6), rec("convb", 162, 114.7), rec("res8f_kernel<false>", 18, 1340.4))}
    assert bench.rank_kernels(iso_b, situ_b)[0]["kernel"] == "res8f_kernel<false>+res8f_kernel<true>"
    # kernels of different activations are different families (elu blocks beside ReLU blocks: the relation net's backbone is always ReLU)
    mixed = {k["kernel"]: k for k in (rec("res8v_up_kernel<0>", 2, 10.0), rec("res8v_down_kernel<1>", 2, 50.0), rec("res8v_up_kernel<1>", 2, 60.0))}
    assert [k["kernel"] for k in bench.rank_kernels(None, mixed)] == ["res8v_down_kernel<1>+res8v_up_kernel<1>", "res8v_up_kernel<0>"]
