"""GPU parity of the GNN's visual branch: image -> ARU_v1 backbone end points -> per-node ROI
max -> compression -> concatenated node features -> GNN, through the C ABI and through the session mirror, against
the oracle.  Float path: node features / probabilities within 1e-4 / 1e-5 of the oracle (fp32)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(mvn=False, node_dim=7, layers=("scale_0_unet_up_2_conv", "scale_0_unet_up_1_conv", "scale_0_unet_up_0_conv"),
           dims=(16, 16, 16), seed=11, backbone=None, **cfg_kw):
    from citlab_article_separation_new_amd.config import GnnConfig
    from citlab_article_separation_new_amd.gnn_io import GnnGraph
    from citlab_article_separation_new_amd.weights import init_gnn_weights
    cfg = GnnConfig(node_feature_dim=node_dim, visual_dims=list(dims), visual_layers=list(layers), mvn=mvn,
                    backbone=dict(backbone or {}), **cfg_kw)
    w = init_gnn_weights(cfg, seed, bias_jitter=0.05)
    return cfg, w, GnnGraph(w, cfg)


def _page(rng, N, h, w, P=4):
    from citlab_article_separation_new_amd import synth
    img = synth.synth_page(5, W=w, H=h).astype(np.float32)          # 0..255 as fed (input_dataset.py:279-280)
    regions = np.zeros((N, 2, P), np.float32)
    npts = np.full(N, P, np.int32)
    for n in range(N):
        x0, y0 = rng.random() * 0.8, rng.random() * 0.8
        x1, y1 = x0 + 0.02 + rng.random() * 0.18, y0 + 0.01 + rng.random() * 0.1
        regions[n, 0] = [x0, x1, x1, x0]
        regions[n, 1] = [y0, y0, y1, y1]
    regions[0, 0, :] = [0.0, 1.0, 1.0, 0.0]                          # full page, touches the clamp at fw-1
    regions[0, 1, :] = [0.0, 0.0, 1.0, 1.0]
    npts[1] = 0                                                      # no points -> ROI is cell (0, 0)
    regions[2, :, :] = 0.5                                           # degenerate single point
    npts[3] = 2
    return img, regions, npts


@pytest.mark.parametrize("mvn", [False, True])
def test_visual_forward_matches_oracle(mvn):
    from citlab_article_separation_new_amd import gnn_io, synth
    from oracle import gnn_oracle
    cfg, w, graph = _setup(mvn=mvn)
    rng = np.random.default_rng(3)
    N = 30
    g = synth.synth_graph(1, N=N, n_pairs=80, node_dim=7)
    img, regions, npts = _page(rng, N, 200, 136)
    if not mvn:
        img = img / np.float32(255)                                  # keep un-normalised activations moderate
    probs = gnn_io.gnn_forward_visual(graph, N, g["interacting_nodes"], g["node_features"], g["edge_features"], img,
                                      regions, npts)
    u = gnn_io.gnn_node_features(graph, N)
    ref_probs, ref_u = gnn_oracle.forward_visual(N, g["interacting_nodes"], g["node_features"], g["edge_features"],
                                                 img, regions, npts, None, w, cfg)
    assert u.shape == (N, 55)
    assert np.array_equal(u[:, :7], ref_u[:, :7])
    print("max |du| =", np.abs(u - ref_u).max(), "max |u| =", np.abs(ref_u).max(),
          "max |dp| =", np.abs(probs - ref_probs).max())
    assert np.abs(u - ref_u).max() <= 1e-4 * max(1.0, np.abs(ref_u).max())
    assert (ref_u[:, 7:] > 0).any()                                  # the compression ReLU is not dead everywhere
    assert np.abs(probs - ref_probs).max() <= 1e-5                   # fp32 tolerance of the GNN tests
    assert probs.shape == (N * N, 2)


def test_visual_forward_with_a_bf16_backbone_stays_within_the_bf16_tolerance():
    """BASELINE configs[4] ("bf16 convs") for the relation net: ``backbone={"compute_dtype": "bf16"}`` runs the RU backbone
    on the bf16 kernels and the ROI kernel reads the bf16 end points (gnn_roi_compress_kernel<true>); compression, graph and
    classifier stay fp32.  Gates: the geometric columns are untouched, the visual node features within 2e-2 of the fp32
    oracle relative to the largest feature (the whole-frame bf16 gate of the page net), the relation
    probabilities within 2e-2 absolute."""
    from citlab_article_separation_new_amd import gnn_io, synth
    from oracle import gnn_oracle
    cfg, w, graph = _setup(mvn=True, backbone={"compute_dtype": "bf16"})
    rng = np.random.default_rng(3)
    N = 30
    g = synth.synth_graph(1, N=N, n_pairs=80, node_dim=7)
    img, regions, npts = _page(rng, N, 200, 136)
    probs = gnn_io.gnn_forward_visual(graph, N, g["interacting_nodes"], g["node_features"], g["edge_features"], img,
                                      regions, npts)
    u = gnn_io.gnn_node_features(graph, N)
    ref_probs, ref_u = gnn_oracle.forward_visual(N, g["interacting_nodes"], g["node_features"], g["edge_features"],
                                                 img, regions, npts, None, w, cfg)
    du, dp = float(np.abs(u - ref_u).max()), float(np.abs(probs - ref_probs).max())
    print("bf16 backbone: max |du| =", du, "max |u| =", float(np.abs(ref_u).max()), "max |dp| =", dp)
    assert np.array_equal(u[:, :7], ref_u[:, :7])
    assert du <= 2e-2 * max(1.0, float(np.abs(ref_u).max()))
    assert du > 0.0                                                   # the bf16 kernels did run (fp32 would agree to ~1e-6)
    assert dp <= 2e-2
    graph.close()


def test_visual_forward_with_a_split_product_backbone_holds_the_fp32_gates():
    """``backbone={"compute_dtype": "f32s"}``: the RU backbone's wide convolutions on the split-product kernels (fp32 tensors and results,
    csrc/split_kernels.h) -- the fp32 tolerances of test_visual_forward_matches_oracle apply unchanged"""
    from citlab_article_separation_new_amd import gnn_io, synth
    from oracle import gnn_oracle
    cfg, w, graph = _setup(mvn=True, backbone={"compute_dtype": "f32s"})
    _, _, graph32 = _setup(mvn=True, backbone={"compute_dtype": "f32"})            # the plain fp32 kernels
    rng = np.random.default_rng(3)
    N = 30
    g = synth.synth_graph(1, N=N, n_pairs=80, node_dim=7)
    img, regions, npts = _page(rng, N, 200, 136)
    probs = gnn_io.gnn_forward_visual(graph, N, g["interacting_nodes"], g["node_features"], g["edge_features"], img, regions, npts)
    u = gnn_io.gnn_node_features(graph, N)
    ref_probs, ref_u = gnn_oracle.forward_visual(N, g["interacting_nodes"], g["node_features"], g["edge_features"],
                                                 img, regions, npts, None, w, cfg)
    du, dp = float(np.abs(u - ref_u).max()), float(np.abs(probs - ref_probs).max())
    print("f32s backbone: max |du| =", du, "max |u| =", float(np.abs(ref_u).max()), "max |dp| =", dp)
    assert np.array_equal(u[:, :7], ref_u[:, :7])
    assert du <= 1e-4 * max(1.0, float(np.abs(ref_u).max())) and dp <= 1e-5
    gnn_io.gnn_forward_visual(graph32, N, g["interacting_nodes"], g["node_features"], g["edge_features"], img, regions, npts)
    assert not np.array_equal(gnn_io.gnn_node_features(graph32, N), u)              # the split kernels did run
    graph.close(); graph32.close()


def test_session_mirror_with_image_feeds_and_pb_roundtrip(tmp_path):
    from citlab_article_separation_new_amd import gnn_io, pb_import, synth
    from citlab_article_separation_new_amd.gnn_input import build_full_relations
    from oracle import gnn_oracle
    cfg, w, _ = _setup(mvn=True, layers=("scale_0_unet_up_1_conv", "scale_0_unet_down_2_conv"), dims=(16, 8), seed=5)
    extra = [{"name": "graph/map/per_image_standardization/Mean", "op": "Mean"}]
    pb = tmp_path / "gnn_visual.pb"
    pb.write_bytes(pb_import.weights_to_graphdef(w, "graph/", extra, meta={"num_transition_steps": cfg.num_transition_steps}))
    # the importer cannot know the from_layer names: defaults assume up-path outputs ...
    g_default = gnn_io.load_graph(str(pb))
    assert g_default.cfg.visual_layers == ["scale_0_unet_up_1_conv", "scale_0_unet_up_2_conv"]
    assert g_default.cfg.visual_dims == [16, 8] and g_default.cfg.node_feature_dim == 7 and g_default.cfg.mvn
    # ... and the caller can name them
    graph = gnn_io.load_graph(str(pb), visual_layers=list(cfg.visual_layers))
    assert graph.cfg.backbone_cfg().graph == "RU" and graph.cfg.backbone_cfg().mvn
    rng = np.random.default_rng(8)
    N = 12
    g = synth.synth_graph(2, N=N, n_pairs=30, node_dim=7)
    img, regions, npts = _page(rng, N, 120, 96)
    rel = build_full_relations(N)[0]
    E = g["interacting_nodes"].shape[0]
    feed = {"num_nodes:0": np.array([N], np.int32), "num_interacting_nodes:0": np.array([E], np.int32),
            "interacting_nodes:0": g["interacting_nodes"][None], "node_features:0": g["node_features"][None],
            "edge_features:0": g["edge_features"][None], "image:0": img[None, :, :, None],
            "image_shape:0": np.array([[120, 96, 1]], np.int32), "visual_regions_nodes:0": regions[None],
            "num_points_visual_regions_nodes:0": npts[None],
            "relations_to_consider_belong_to_same_instance:0": rel[None]}
    with gnn_io.GnnSession(graph) as sess:
        out = sess.run("output_belong_to_same_instance:0", feed)
    ref, _ = gnn_oracle.forward_visual(N, g["interacting_nodes"], g["node_features"], g["edge_features"], img,
                                       regions, npts, rel, w, cfg)
    assert out.shape == (1, N * N, 2)
    assert np.abs(out[0] - ref).max() <= 1e-5
    del feed["image:0"]
    with pytest.raises(KeyError):
        gnn_io.GnnSession(graph).run("output_belong_to_same_instance:0", feed)


def test_attach_rejects_unknown_end_points():
    from citlab_article_separation_new_amd import _lib
    from citlab_article_separation_new_amd.config import GnnConfig
    with pytest.raises(ValueError):
        GnnConfig(visual_dims=[16], visual_layers=["Mixed_5d"]).visual_channels()
    cfg, w, graph = _setup()
    graph.cfg.visual_layers = ["scale_0_unet_up_9_conv", "scale_0_unet_up_1_conv", "scale_0_unet_up_0_conv"]
    with pytest.raises(_lib.AsepError):
        graph.handle(0)


def test_visual_net_at_c4_size_uses_the_mfma_step_and_matches_oracle():
    """BASELINE configs[3] names mixed_gnn_vn7e2.pb -- the VISUAL net: 200 nodes, E' = 20 000, all 40 000 pairs, image
    683 x 1024 (a 3000 x 4500 scan after the input pipeline's resize), 7 + 3 x 16 = 55 node features (K = 350).  The
    wide-feature MFMA step kernel must serve it (not the scalar fallback), through the host entry and through the
    device-resident entry on a side stream, and both must match the oracle."""
    import torch
    from citlab_article_separation_new_amd import gnn_io, synth
    from oracle import gnn_oracle
    cfg, w, graph = _setup(mvn=True)
    assert gnn_io.step_mode(graph) == "factored"
    rng = np.random.default_rng(13)
    N = 200
    g = synth.synth_graph(0, N=N, n_pairs=10000, node_dim=7)
    img, regions, npts = _page(rng, N, 1024, 683)
    probs = gnn_io.gnn_forward_visual(graph, N, g["interacting_nodes"], g["node_features"], g["edge_features"], img,
                                      regions, npts)
    u = gnn_io.gnn_node_features(graph, N)
    ref_probs, ref_u = gnn_oracle.forward_visual(N, g["interacting_nodes"], g["node_features"], g["edge_features"],
                                                 img, regions, npts, None, w, cfg)
    du, dp = float(np.abs(u - ref_u).max()), float(np.abs(probs - ref_probs).max())
    print(f"\nvisual C4: max|du| = {du:.2e} (max|u| {np.abs(ref_u).max():.2f}), max|dp| = {dp:.2e}")
    assert du <= 1e-4 * max(1.0, float(np.abs(ref_u).max())) and dp <= 1e-5
    # device-resident entry, non-default stream, no host synchronisation inside
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    t_e, t_u, t_f = dev(g["interacting_nodes"]), dev(g["node_features"]), dev(g["edge_features"])
    t_img, t_reg, t_np = dev(img), dev(regions), dev(npts)
    t_out = torch.zeros(N * N, 2, device="cuda")
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        for _ in range(2):                                            # a second call reuses every buffer
            gnn_io.gnn_forward_visual_dev(graph, N, t_e.shape[0], t_e.data_ptr(), t_u.data_ptr(), t_f.data_ptr(),
                                          t_img.data_ptr(), 1024, 683, t_reg.data_ptr(), regions.shape[2], t_np.data_ptr(),
                                          N * N, None, t_out.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    assert np.array_equal(t_out.cpu().numpy(), probs)
    assert np.array_equal(gnn_io.gnn_node_features(graph, N), u)
    graph.close()


def test_reattach_and_feature_readback_are_stable():
    """a second attach frees the first one's uploads; the concatenated features stay readable after another
    call has recycled the forward's buffer pool."""
    import ctypes as C
    from citlab_article_separation_new_amd import _lib, gnn_io, synth
    cfg, w, graph = _setup()
    rng = np.random.default_rng(4)
    N = 20
    g = synth.synth_graph(3, N=N, n_pairs=40, node_dim=7)
    img, regions, npts = _page(rng, N, 96, 80)
    img = img / np.float32(255)
    p1 = gnn_io.gnn_forward_visual(graph, N, g["interacting_nodes"], g["node_features"], g["edge_features"], img, regions, npts)
    u1 = gnn_io.gnn_node_features(graph, N)
    gnn_io.correct_edges(graph, N, g["interacting_nodes"], g["edge_features"])       # recycles the pool
    assert np.array_equal(gnn_io.gnn_node_features(graph, N), u1)
    lib = _lib.init_device(0)
    names = (C.c_char_p * 3)(*[n.encode() for n in cfg.visual_layers])
    for _ in range(3):
        _lib.check(lib.asep_gnn_attach_backbone(graph.handle(0), graph._backbones[0].handle(0), 3, names), "attach")
    p2 = gnn_io.gnn_forward_visual(graph, N, g["interacting_nodes"], g["node_features"], g["edge_features"], img, regions, npts)
    assert np.array_equal(p1, p2)
    graph.close()


def test_batched_visual_forward_equals_the_single_page_calls():
    """asep_gnn_forward_visual_batch_dev (bench.py's step): the backbones of all pages as one grouped forward, then ROI
    kernels + graph per page.  Pages with different graphs / images / node counts must come out exactly as from
    asep_gnn_forward_visual (bit-identical: same kernels on the same values), and page 0 also matches the oracle."""
    import torch
    from citlab_article_separation_new_amd import gnn_io, synth
    from oracle import gnn_oracle
    cfg, w, graph = _setup(mvn=True)
    rng = np.random.default_rng(29)
    h, wd = 160, 112
    pages, keep, singles = [], [], []
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for b, N in enumerate((24, 17, 31)):
        g = synth.synth_graph(40 + b, N=N, n_pairs=3 * N, node_dim=7)
        img, regions, npts = _page(rng, N, h, wd)
        img = np.ascontiguousarray(np.roll(img, 9 * b, axis=1))       # a different image per page
        singles.append(gnn_io.gnn_forward_visual(graph, N, g["interacting_nodes"], g["node_features"], g["edge_features"], img,
                                                 regions, npts))
        if b == 0:
            ref, _ = gnn_oracle.forward_visual(N, g["interacting_nodes"], g["node_features"], g["edge_features"], img, regions,
                                               npts, None, w, cfg)
            assert float(np.abs(singles[0] - ref).max()) <= 1e-5
        t = [dev(g["interacting_nodes"]), dev(g["node_features"]), dev(g["edge_features"]), dev(img), dev(regions), dev(npts),
             torch.zeros(N * N, 2, device="cuda")]
        keep.append(t)
        pages.append(dict(N=N, E=int(t[0].shape[0]), R=N * N, d_edges=t[0].data_ptr(), d_node_feat=t[1].data_ptr(),
                          d_edge_feat=t[2].data_ptr(), d_image=t[3].data_ptr(), d_regions=t[4].data_ptr(),
                          d_num_points=t[5].data_ptr(), d_relations=None, d_probs_out=t[6].data_ptr()))
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    for _ in range(2):                                                # the second call reuses every buffer
        gnn_io.gnn_forward_visual_batch_dev(graph, pages, h, wd, 4, stream.cuda_stream)
    stream.synchronize()
    for t, single in zip(keep, singles):
        assert np.array_equal(t[6].cpu().numpy(), single)
    graph.close()


def _edge_regions(rng, regions, edges):
    """a region per interaction: the bounding box of its two nodes' regions (what feature_generation writes), some without points"""
    E, P = len(edges), regions.shape[2]
    er = np.zeros((E, 2, P), np.float32)
    for e, (a, b) in enumerate(edges):
        x0, x1 = min(regions[a, 0].min(), regions[b, 0].min()), max(regions[a, 0].max(), regions[b, 0].max())
        y0, y1 = min(regions[a, 1].min(), regions[b, 1].min()), max(regions[a, 1].max(), regions[b, 1].max())
        er[e, 0] = [x0, x1, x1, x0]
        er[e, 1] = [y0, y0, y1, y1]
    enp = np.full(E, P, np.int32)
    enp[::7] = 0                                                     # no points -> cell (0, 0)
    return er, enp


@pytest.mark.parametrize("edge_dim,undirected", [(2, True), (0, True), (2, False)])
def test_visual_edge_features_match_oracle(edge_dim, undirected):
    """graph_relation.py:141-172 assign_visual_features_to_edges (VERDICT r3 missing #2): ROI max over the backbone end points of every
    interaction's region -> visual_edge_feature_compression_fm_<i> -> concatenated behind the fed edge features, BEFORE the edge
    correction (duplicates and reversed duplicates keep the first occurrence's visual features like the geometric ones).  Through the C
    ABI, through the session mirror with the reference's feed names, and as a batch on the device."""
    from citlab_article_separation_new_amd import _lib, gnn_io, synth
    from oracle import gnn_oracle
    cfg, w, graph = _setup(dims=(8, 4, 4), visual_edges=True, edge_feature_dim=edge_dim, undirected_graph=undirected)
    assert cfg.edge_in_dim == edge_dim + 16 and "visual_edge_feature_compression_fm_2/dense/weights" in w
    rng = np.random.default_rng(8)
    N = 24
    g = synth.synth_graph(2, N=N, n_pairs=60, node_dim=7)
    edges = g["interacting_nodes"].copy()
    edges[1] = edges[0]                                              # duplicate and reversed duplicate with DIFFERENT regions
    edges[2] = edges[0][::-1]
    ef = g["edge_features"][:, :edge_dim] if edge_dim else None
    img, regions, npts = _page(rng, N, 200, 136)
    img = img / np.float32(255)
    er, enp = _edge_regions(rng, regions, edges)
    er[1] = er[5]
    er[2] = er[9]
    probs = gnn_io.gnn_forward_visual(graph, N, edges, g["node_features"], ef, img, regions, npts, edge_regions=er, edge_num_points=enp)
    ref, _ = gnn_oracle.forward_visual(N, edges, g["node_features"], ef, img, regions, npts, None, w, cfg, edge_regions=er, edge_num_points=enp)
    # the visual edge features matter: without them (zeros in their place) the oracle gives something else
    w0 = dict(w)
    for k in w:
        if k.startswith("visual_edge_feature_compression"):
            w0[k] = np.zeros_like(w[k])
    ref0, _ = gnn_oracle.forward_visual(N, edges, g["node_features"], ef, img, regions, npts, None, w0, cfg, edge_regions=er, edge_num_points=enp)
    print("visual edges: max |dp| =", np.abs(probs - ref).max(), "effect of the edge features:", np.abs(ref - ref0).max())
    assert np.abs(ref - ref0).max() > 1e-4 and np.abs(probs - ref).max() <= 1e-5
    # the session mirror with the reference's feed names (graph_relation.py:141-146)
    sess = gnn_io.GnnSession(graph)
    feed = {"num_nodes:0": [N], "num_interacting_nodes:0": [len(edges)], "interacting_nodes:0": edges[None], "node_features:0": g["node_features"][None],
            "image:0": img[None, :, :, None], "image_shape:0": [[img.shape[0], img.shape[1], 1]], "visual_regions_nodes:0": regions[None],
            "num_points_visual_regions_nodes:0": npts[None], "visual_regions_edges:0": er[None], "num_points_visual_regions_edges:0": enp[None],
            "relations_to_consider_belong_to_same_instance:0": gnn_oracle.build_full_relations(N)[None]}
    if edge_dim:
        feed["edge_features:0"] = ef[None]
    out = sess.run("output_belong_to_same_instance:0", feed)
    assert np.array_equal(out[0], probs)
    del feed["visual_regions_edges:0"]
    with pytest.raises(KeyError, match="visual_regions_edges"):
        sess.run("output_belong_to_same_instance:0", feed)
    # the plain entry points refuse such a graph instead of reading edge features of the wrong width
    with pytest.raises(_lib.AsepError, match="visual features to edges"):
        gnn_io.gnn_forward(graph, N, edges, np.zeros((N, cfg.u_in_dim), np.float32), np.zeros((len(edges), cfg.edge_in_dim), np.float32))
    graph.close()
