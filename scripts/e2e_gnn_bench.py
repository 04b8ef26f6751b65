"""Files in, files out for the relation net: graph jsons (+ scans for the visual net) + PAGE-XML -> run_gnn_clustering command
line -> PAGE-XML with article ids.  200 text blocks / ~20k directed edges / 40k pairs per page (BASELINE configs[3]).

    python scripts/e2e_gnn_bench.py [n_pages=64] [workers=8] [visual=1]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 64
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
visual = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True


def main():
    from citlab_article_separation_new_amd import run_gnn_clustering, synth
    with tempfile.TemporaryDirectory(prefix="asep_gnn_e2e_") as tmp:
        t0 = time.perf_counter()
        argv = synth.write_gnn_cli_inputs(tmp, n_pages, visual=visual)
        lst = argv[argv.index("--eval_list") + 1]
        jsons = [ln for ln in open(lst).read().split("\n") if ln]
        print(f"inputs written in {time.perf_counter() - t0:.1f} s; json {os.path.getsize(jsons[0]) / 1e6:.2f} MB per page")
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            for nw in sorted({1, workers}):
                part = jsons if nw > 1 else jsons[: max(8, n_pages // 8)]
                with open(lst, "w") as f:
                    f.write("\n".join(part) + "\n")
                t0 = time.perf_counter()
                outs = run_gnn_clustering.main(argv + ["--out_dir", f"out{nw}", "--gpu_devices", "0", "--num_workers", str(nw)])
                dt = time.perf_counter() - t0
                print(f"run_gnn_clustering, {'visual' if visual else 'geometric'} net, {nw:2d} worker(s): {len(outs)} pages in {dt:.2f} s = "
                      f"{len(outs) / dt:.1f} pages/s ({dt / len(outs) * 1e3:.1f} ms/page incl. start-up)")
        finally:
            os.chdir(cwd)


if __name__ == "__main__":
    main()
