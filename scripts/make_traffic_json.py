"""profiles/traffic_per_kernel.json from a rocprofv3 PMC summary (scripts/summarize_pmc.py output):
    python scripts/make_traffic_json.py profiles/<tag> [commit]
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, averaged over the dispatches of a kernel (FETCH_SIZE doubled: the gfx950
correction of /opt/skills/guides/MI355X_MICROARCH.md, section HBM).  Keys are the kernels' rocprofv3 names without "void ", "asep::",
blanks and the argument list -- exactly what the engine's profiler (asep_aru_profile_report) and bench.py report, so no two
instantiations share a key (round 2's shortened names made conv_mfma_kernel<...,4> and <...,2> collide)."""
import json
import os
import re
import sys


def kernel_key(rocprof_name):
    """'void asep::conv_mfma_kernel<3, 3, 1, false, 16, false, false, false, 4>(asep::ConvArgs)' -> 'conv_mfma_kernel<3,3,1,false,16,false,false,false,4>'"""
    n = rocprof_name.strip()
    if n.startswith("void "):
        n = n[5:]
    depth, cut = 0, len(n)
    for i, ch in enumerate(n):                      # the argument list starts at the first '(' outside the template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    n = n[:cut].replace("asep::", "").replace(" ", "")
    return n


def traffic_table(pmc_summary, pages_in_run):
    """-> ({key: {...}}, bytes of all asep:: kernels per page)"""
    kernels, total = {}, 0.0
    for name, f in pmc_summary["FETCH_SIZE"].items():
        w = pmc_summary["WRITE_SIZE"].get(name)
        if w is None or "asep::" not in name:
            continue
        per_launch = (2.0 * f["avg_per_dispatch"] + w["avg_per_dispatch"]) * 1024.0
        kernels[kernel_key(name)] = {"bytes_per_launch": per_launch, "fetch_size_kb": f["avg_per_dispatch"],
                                     "write_size_kb": w["avg_per_dispatch"], "dispatches": f["dispatches"]}
        total += (2.0 * f["sum"] + w["sum"]) * 1024.0
    return kernels, total / pages_in_run


def main():
    tagdir = sys.argv[1].rstrip("/")
    commit = sys.argv[2] if len(sys.argv) > 2 else "n/a"          # the build the counters were collected on
    s = json.load(open(os.path.join(tagdir, "pmc_summary.json")))
    line = json.loads(open(os.path.join(tagdir, "bench_under_trace.json")).read().strip().splitlines()[-1])
    # pages of the profiled process: (warm-up + timed + event-timed steps) x pages per step
    rb = {**(line.get("roofline_detail") or {}), **line["roofline"]}          # round 4: detail keys live beside the slim block
    ppl = rb["pages_per_launch"]
    B = line["config"]["pages_per_step_per_gpu"]
    steps_total = line["warmup"] + line["steps"] + rb.get("event_timed_steps", 0)
    kernels, page_bytes = traffic_table(s, steps_total * B)
    # `source` names where the summary is COMMITTED (profiles/<tag>/, copied there from the box's gpurun_out/<tag>/): a reader of the repo can open it
    out = {"source": f"profiles/{os.path.basename(tagdir)}/pmc_summary.json", "commit": commit, "pages_per_launch": ppl, "dtype": line["dtype"],
           "pages_per_step": B, "relation_net": line["config"]["relation_net"], "height": line["config"]["height"],
           "width": line["config"]["width"],
           "unit": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024, mean over dispatches (scripts/profile_bench.sh)",
           "page_bytes": page_bytes, "pages_in_profiled_run": steps_total * B, "kernels": kernels}
    name = "traffic_per_kernel.json" if line["dtype"] == "f32" else f"traffic_per_kernel_{line['dtype']}.json"
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", name)
    json.dump(out, open(dst, "w"), indent=1)
    print(dst, len(kernels), "kernels;", f"{page_bytes / 1e9:.2f} GB per page")


if __name__ == "__main__":
    main()
