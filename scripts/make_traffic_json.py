"""profiles/traffic_per_kernel.json from a rocprofv3 PMC summary (scripts/summarize_pmc.py output):
    python scripts/make_traffic_json.py profiles/<tag>/pmc_summary.json <tag> [commit]
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, averaged over the dispatches of a kernel (FETCH_SIZE doubled: the gfx950
correction of /opt/skills/guides/MI355X_MICROARCH.md).  Keys are the kernel names bench.py reports."""
import json, os, re, sys


def bench_name(rocprof_name):
    n = rocprof_name.replace("void ", "").replace("asep::", "")
    n = n[:n.index("(")] if "(" in n else n
    m = re.match(r"(\w+)<(.*)>$", n)
    if not m:
        return n
    base, targs = m.group(1), [t.strip() for t in m.group(2).split(",")]
    if base in ("res8_up_kernel", "res8_down_kernel", "conv_winor_kernel"):
        return base
    if base in ("conv_wino_kernel", "deconv_mfma_kernel"):
        return f"{base}<{targs[0]}>"
    if base == "conv_mfma_kernel":
        head = ",".join(targs[:4])
        return f"{base}<{head},16,false>" if targs[4] == "16" else f"{base}<{head}>"
    return n


def main():
    src, tag = sys.argv[1], sys.argv[2]
    commit = sys.argv[3] if len(sys.argv) > 3 else "n/a"          # the build the counters were collected on
    s = json.load(open(src))
    out = {"source": f"profiles/{tag}/pmc_summary.json", "commit": commit, "pages_per_launch": 2,
           "unit": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024, mean over dispatches "
                   "(scripts/profile_bench.sh: bench.py --pages-per-step 2)",
           "kernels": {}}
    for name, f in s["FETCH_SIZE"].items():
        w = s["WRITE_SIZE"].get(name)
        if w is None or "asep::" not in name:
            continue
        out["kernels"][bench_name(name)] = {
            "bytes_per_launch": (2.0 * f["avg_per_dispatch"] + w["avg_per_dispatch"]) * 1024.0,
            "fetch_size_kb": f["avg_per_dispatch"], "write_size_kb": w["avg_per_dispatch"], "dispatches": f["dispatches"]}
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic_per_kernel.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(dst, len(out["kernels"]), "kernels")


if __name__ == "__main__":
    main()
