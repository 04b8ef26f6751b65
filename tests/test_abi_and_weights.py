"""CPU-side checks: the C-ABI library loads and exports every symbol declared in include/asep_hip.h
(no compute calls without a GPU), the ctypes signature table matches the header, weight containers round-trip."""
import os
import re

import numpy as np
import pytest


def _header_functions(repo_root):
    src = open(os.path.join(repo_root, "include", "asep_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(asep_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(repo_root):
    from citlab_article_separation_new_amd import _lib
    names = _header_functions(repo_root)
    assert len(names) >= 17
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load_library()
    for n in names:
        assert hasattr(lib, n), n
    assert lib.asep_version().decode().startswith("asep_hip")
    assert lib.asep_device_count() >= 0


def _header_struct_fields(repo_root, name):
    """field names of `typedef struct <name> { ... } <name>;` in include/asep_hip.h, in order (all are int32_t)"""
    src = open(os.path.join(repo_root, "include", "asep_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    body = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), src, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        typ, names = decl.split(None, 1)
        assert typ == "int32_t", f"{name}: the ctypes mirrors assume int32_t fields, found {decl!r}"
        fields += [n.strip() for n in names.split(",")]
    return fields


def test_cfg_structs_match_the_header_field_by_field(repo_root):
    """the header gained two fields and a hand-written binding kept the old struct -- the callee read 8 bytes of
    stack.  The ctypes mirrors must list the header's fields in the header's order, and every struct starts with its size."""
    import ctypes as C
    from citlab_article_separation_new_amd import _lib
    for cname, cls in (("asep_aru_cfg", _lib.AruCfg), ("asep_gnn_cfg", _lib.GnnCfg)):
        want = _header_struct_fields(repo_root, cname)
        assert [f[0] for f in cls._fields_] == want, cname
        assert want[0] == "struct_size" and all(f[1] is C.c_int32 for f in cls._fields_)
        assert cls().struct_size == C.sizeof(cls) == 4 * len(want)          # filled in by the constructor
    src = open(os.path.join(repo_root, "include", "asep_hip.h")).read()
    assert int(re.search(r"#define\s+ASEP_ABI_VERSION\s+(\d+)", src).group(1)) == _lib.ABI_VERSION
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    assert _lib.load_library().asep_abi_version() == _lib.ABI_VERSION


def integration_md_stub(repo_root):
    """the ctypes stub of INTEGRATION.md section 1, executed: -> its namespace (lib, AruCfg, load_graph, get_net_output)"""
    text = open(os.path.join(repo_root, "INTEGRATION.md")).read()
    sec = text[text.index("## 1. ARU-Net seam"):text.index("## 2. GNN seam")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    code = next(b for b in blocks if "C.CDLL(" in b)
    ns = {}
    cwd = os.getcwd()
    os.chdir(repo_root)                      # the document's library path is relative to the repository root
    try:
        exec(compile(code, "INTEGRATION.md#1", "exec"), ns)
    finally:
        os.chdir(cwd)
    return ns


def test_integration_md_stub_loads_and_matches_the_header(repo_root):
    """the document's hand-written binding is code: it must load the built library, pass its ABI check and declare the struct the
    header declares (load only: no compute call without a GPU)"""
    import ctypes as C
    from citlab_article_separation_new_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    ns = integration_md_stub(repo_root)
    assert [f[0] for f in ns["AruCfg"]._fields_] == _header_struct_fields(repo_root, "asep_aru_cfg")
    assert C.sizeof(ns["AruCfg"]) == C.sizeof(_lib.AruCfg)
    assert callable(ns["load_graph"]) and callable(ns["get_net_output"])


def test_host_library_exports_every_declared_symbol(repo_root):
    """include/asep_host.h (plain C helper of the scan decode, csrc/host_png.c -> libasep_host.so)"""
    import ctypes as C
    src = open(os.path.join(repo_root, "include", "asep_host.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(asep_[a-z0-9_]+)\s*\(", src))
    assert names == {"asep_png_unfilter", "asep_rgb_to_bgr"}
    path = os.path.join(repo_root, "citlab-article-separation-new_amd", "csrc", "libasep_host.so")
    if not os.path.exists(path):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(path)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/asep_host.h but not exported"


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product path must fail loudly instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from citlab_article_separation_new_amd import _lib, net_post_processing_helper as helper
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights
    cfg = AruConfig()
    g = helper.AruGraph(init_aru_weights(cfg, 1), cfg)
    with pytest.raises(_lib.AsepError):
        helper.get_net_output(np.zeros((16, 16), np.float32), g, "0")


def test_product_code_never_imports_the_oracle(repo_root):
    pkg = os.path.join(repo_root, "citlab-article-separation-new_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "oracle/" not in text.replace("the oracle under ``oracle/``", ""), f


def test_weight_inventory_and_blob_roundtrip(tmp_path):
    from citlab_article_separation_new_amd.config import AruConfig, GnnConfig
    from citlab_article_separation_new_amd import weights as W
    cfg = AruConfig()
    w = W.init_aru_weights(cfg, 1234)
    assert sum(v.size for v in w.values()) == 1043839           # 1 031 552 + 12 029 + 258
    assert w["aru_net/featMapG/unet_up_0/deconv/weights"].shape == (3, 3, 8, 16)     # [kh,kw,Cout,Cin]
    assert w["aru_net/featMapG/unet_up_0/conv1/weights"].shape == (3, 3, 16, 8)
    assert np.all(w["aru_net/logit/class/biases"] == np.float32(0.1))
    back = W.unpack_blob(W.pack_blob(w))
    assert list(back) == list(w) and all(np.array_equal(back[k], w[k]) for k in w)
    g = W.init_gnn_weights(GnnConfig(), 7)
    assert g["GraphLSTM1/message_fn_default/head_0/calculation_interaction_features/concat_u_and_h/"
             "interaction_features/fully_connected_layer_h1/weights"].shape == (158, 32)
    assert g["GraphLSTM1/update_function_LSTM/ingate_activation/dense/weights"].shape == (71, 32)
    assert g["Classification/logits/fully_connected_layer_h1/weights"].shape == (64, 64)
    p = str(tmp_path / "m.asepw")
    W.save_weights(p, w, {"aru_cfg": cfg.to_dict()})
    t, meta = W.load_weights(p)
    assert meta["aru_cfg"]["feat_root"] == 8 and np.array_equal(t["aru_net/logit/class/weights"], w["aru_net/logit/class/weights"])
    with pytest.raises(IOError):
        W.unpack_blob(b"garbage!" + b"\0" * 16)


def test_load_graph_accepts_container_and_rejects_pb(tmp_path):
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd import weights as W, net_post_processing_helper as helper
    cfg = AruConfig(n_classes=3)
    p = str(tmp_path / "sep.asepw")
    W.save_weights(p, W.init_aru_weights(cfg, 3), {"aru_cfg": cfg.to_dict()})
    g = helper.load_graph(p)
    assert g.cfg.n_classes == 3 and g.input_name == "inImg:0" and g.output_name == "output:0"
    with pytest.raises(IOError):
        helper.load_graph(str(tmp_path / "missing.pb"))
    (tmp_path / "x.pb").write_bytes(b"\x0a\x00")              # a GraphDef with one empty node: no ARU-Net constants
    with pytest.raises(IOError):
        helper.load_graph(str(tmp_path / "x.pb"))
    assert helper.get_scaling_factor(4500, 3000, 1.0, fixed_height=1500) == pytest.approx(1 / 3)


def test_every_environment_switch_the_sources_read_is_listed_and_documented():
    """asep_engine_switches() (ABI 6) and DESIGN.md section 4.5 name exactly what the product build reads: every getenv("ASEP_...") of csrc/*.hip
    outside an ASEP_ABLATION block must be in the function's list and in the table (bench.py records only listed names as engine switches)."""
    import glob
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(ROOT, "citlab-article-separation-new_amd", "csrc")
    read, ablation_only = set(), set()
    for path in glob.glob(os.path.join(csrc, "*.hip")):
        depth_abl = 0
        for line in open(path):
            s = line.strip()
            if s.startswith("#ifdef ASEP_ABLATION") or s.startswith("#if defined(ASEP_ABLATION)"):
                depth_abl += 1
            elif s.startswith("#endif") and depth_abl:
                depth_abl -= 1
            for name in re.findall(r'getenv\("(ASEP_[A-Z0-9_]+)"\)', line):
                (ablation_only if depth_abl else read).add(name)
    common = open(os.path.join(csrc, "asep_common.hip")).read()
    body = common[common.index("const char* asep_engine_switches(void)"):]
    body = body[:body.index("int asep_abi_version")]
    product = body[:body.index("#ifdef ASEP_ABLATION")] if "#ifdef ASEP_ABLATION" in body else body
    listed = set(re.findall(r"ASEP_[A-Z0-9_]+", product)) - {"ASEP_ABLATION"}
    assert read, "no switches found: the scan is broken"
    assert read == listed, (sorted(read - listed), sorted(listed - read))
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    sec = design[design.index("### 4.5 Engine switches"):design.index("## 5. Measurement")]
    missing = [n for n in sorted(read) if "`" + n not in sec]
    assert not missing, missing
    assert not (ablation_only & listed)
