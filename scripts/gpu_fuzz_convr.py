"""Development aid: random page sizes (single pages and heterogeneous batches) through the bf16 engine with ASEP_BF_CONVR=1 and =0; the probabilities must be
bit-identical (convr_kernel keeps convb_kernel's accumulation order).   python scripts/gpu_fuzz_convr.py [cases] [seed]"""
import os, sys, subprocess, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if len(sys.argv) > 3:                                        # child: run the cases in one mode, dump the outputs' hashes
    import hashlib, ctypes as C, torch
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights
    from citlab_article_separation_new_amd import net_post_processing_helper as helper, _lib
    cfg = AruConfig(compute_dtype="bf16")
    g = helper.AruGraph(init_aru_weights(cfg, 12), cfg)
    lib = _lib.init_device(0); h = g.handle(0)
    rng = np.random.default_rng(SEED)
    out = []
    s = torch.cuda.current_stream().cuda_stream
    for c in range(N):
        npg = int(rng.integers(1, 5))
        sizes = [(int(rng.integers(16, 1400)), int(rng.integers(16, 1400))) for _ in range(npg)]
        imgs = [torch.rand(H, W, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1000 * c + i)) for i, (H, W) in enumerate(sizes)]
        outs = [torch.empty(H, W, 2, device="cuda") for H, W in sizes]
        Arr = C.c_void_p * npg
        I32 = C.c_int32 * npg
        _lib.check(lib.asep_aru_forward_batch_dev2(h, npg, Arr(*[t.data_ptr() for t in imgs]), I32(*[H for H, _ in sizes]), I32(*[W for _, W in sizes]),
                                                    Arr(*[t.data_ptr() for t in outs]), None, None, 0.05, s), "fwd")
        torch.cuda.synchronize()
        out.append((sizes, [hashlib.sha1(o.cpu().numpy().tobytes()).hexdigest() for o in outs]))
    pickle.dump(out, open(sys.argv[3], "wb"))
    sys.exit(0)
res = {}
for v in ("1", "0"):
    subprocess.check_call([sys.executable, __file__, str(N), str(SEED), f"/tmp/convr_fuzz_{v}.pkl"], env={**os.environ, "ASEP_BF_CONVR": v})
    res[v] = pickle.load(open(f"/tmp/convr_fuzz_{v}.pkl", "rb"))
bad = [(a[0], i) for a, b in zip(res["1"], res["0"]) for i, (x, y) in enumerate(zip(a[1], b[1])) if x != y]
pages = sum(len(a[0]) for a in res["1"])
print(f"{N} calls, {pages} pages, sizes {min(min(s) for a in res['1'] for s in a[0])}..{max(max(s) for a in res['1'] for s in a[0])}: {len(bad)} pages differ", bad[:5])
sys.exit(1 if bad else 0)
