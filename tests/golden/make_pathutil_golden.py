"""Generates tests/golden/pathutil_golden.json by IMPORTING the reference's folder-convention helpers and
``get_scaling_factor`` in the authoring container and recording what they return (or raise) on a fixed file tree.

    /root/reference/python_util/io/path_util.py, python_util/io/file_loader.py
    /root/reference/python_util/image_processing/image_stats.py  (imports cv2 / matplotlib at module level although
        get_scaling_factor / get_image_dimensions use neither: the two names are registered as empty modules for
        the import only, like `kneed` in make_clustering_golden.py)

Run:  python tests/golden/make_pathutil_golden.py
The fixture is data only: (function, arguments relative to the tree root) -> result / exception type.
"""
import json
import os
import sys
import tempfile
import types

for name in ("cv2", "matplotlib", "matplotlib.pyplot"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, "/root/reference")
from python_util.io import path_util as ref_path  # noqa: E402
from python_util.io import file_loader as ref_loader  # noqa: E402
from python_util.image_processing.image_stats import get_scaling_factor as ref_scaling  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# the tree every case runs on ("" = directory); kept in the fixture so the test can rebuild it
TREE = [
    "news/a.png", "news/b.tif", "news/c.jpg", "news/d.jpg", "news/d.png", "news/e.v2.png", "news/f-1_x.jpg",
    "news/g.bmp", "news/page/a.xml", "news/page/b.xml", "news/page/c.xml", "news/page/d.xml", "news/page/e.v2.xml",
    "news/page/f-1_x.xml", "news/page/g.xml", "news/page/a.png.xml", "news/page/orphan.xml",
    "news/json15d2vbb/a.json", "news/json15d2vbb/b.json", "news/json/c.json", "news/jsonX_1/d.json",
    "news/json15d2vbb/e.v2.json", "news/json15d2vbb/nopage.json", "news/confidences/a_confidences.json",
    "news/confidences/zz_confidences.json", "news/h.tif.png",
    "model1/export/net_best_2020.pb", "model1/export/net_best_2020_gpu.pb", "model1/export/readme.txt",
    "model2/export/x_2019.pb", "model2/export/y_2019.pb", "model3/export/",
]

CASES = [
    ("get_img_from_page_path", ["news/page/a.xml"]), ("get_img_from_page_path", ["news/page/b.xml"]),
    ("get_img_from_page_path", ["news/page/c.xml"]), ("get_img_from_page_path", ["news/page/d.xml"]),
    ("get_img_from_page_path", ["news/page/e.v2.xml"]), ("get_img_from_page_path", ["news/page/f-1_x.xml"]),
    ("get_img_from_page_path", ["news/page/g.xml"]), ("get_img_from_page_path", ["news/page/a.png.xml"]),
    ("get_img_from_page_path", ["news/page/orphan.xml"]), ("get_img_from_page_path", ["news/a.png"]),
    ("get_img_from_page_path", ["news/page/h.tif.xml"]),
    ("get_img_from_json_path", ["news/json15d2vbb/a.json"]), ("get_img_from_json_path", ["news/json15d2vbb/b.json"]),
    ("get_img_from_json_path", ["news/json/c.json"]), ("get_img_from_json_path", ["news/jsonX_1/d.json"]),
    ("get_img_from_json_path", ["news/json15d2vbb/e.v2.json"]), ("get_img_from_json_path", ["news/json15d2vbb/nopage.json"]),
    ("get_img_from_json_path", ["news/json-x/a.json"]), ("get_img_from_json_path", ["news/json15d2vbb/a.png.json"]),
    ("get_page_from_img_path", ["news/a.png"]), ("get_page_from_img_path", ["news/b.tif"]),
    ("get_page_from_img_path", ["news/e.v2.png"]), ("get_page_from_img_path", ["news/g.bmp"]),
    ("get_page_from_img_path", ["news/missing.png"]), ("get_page_from_img_path", ["news/f-1_x.jpg"]),
    ("get_page_from_img_path", ["news/h.tif.png"]),
    ("get_page_from_json_path", ["news/json15d2vbb/a.json"]), ("get_page_from_json_path", ["news/json/c.json"]),
    ("get_page_from_json_path", ["news/jsonX_1/d.json"]), ("get_page_from_json_path", ["news/json15d2vbb/e.v2.json"]),
    ("get_page_from_json_path", ["news/json15d2vbb/nopage.json"]), ("get_page_from_json_path", ["news/json15d2vbb/a.png"]),
    ("get_page_from_conf_path", ["news/confidences/a_confidences.json"]),
    ("get_page_from_conf_path", ["news/confidences/zz_confidences.json"]),
    ("get_path_from_exportdir", ["model1", "*best*.pb", "_gpu.pb"]), ("get_path_from_exportdir", ["model1", "*_gpu.pb", "cpu"]),
    ("get_path_from_exportdir", ["model1", "*.pb", "nothing"]), ("get_path_from_exportdir", ["model2", "*.pb", "_gpu.pb"]),
    ("get_path_from_exportdir", ["model2", "x*.pb", "_gpu.pb"]), ("get_path_from_exportdir", ["model3", "*.pb", "_gpu.pb"]),
    ("get_path_from_exportdir", ["model4", "*.pb", "_gpu.pb"]),
    ("prepend_folder_name", ["news/a.png"]), ("prepend_folder_name", ["news/page/a.xml"]),
    ("get_page_path", ["news/a.png"]), ("get_page_path", ["news/e.v2.png"]), ("get_page_path", ["other/dir/x.y.jpg"]),
    ("get_page_path", ["noext"]),
]

SCALING = [
    # image_height, image_width, scaling_factor, fixed_height, fixed_width
    [4500, 3000, 1.0, 1500, None], [4500, 3000, 1.0, 900, None], [768, 512, 1.0, 768, None], [4500, 3000, 0.5, 1500, None],
    [4500, 3000, 0.1, 1500, None], [4500, 3000, 0.05, 1500, None], [4500, 3000, None, 1500, None],
    [4500, 3000, 1.0, None, 1000], [4500, 3000, 0.1, None, 1000], [4500, 3000, None, None, 1000],
    [4500, 3000, 0.35, None, None], [4500, 3000, None, None, None], [4500, 3000, 0, 0, 0], [4500, 3000, 0.0, None, None],
    [4500, 3000, 1.0, 0, 1000], [4500, 3000, 0.05, 0, 1000], [4500, 3000, 1.0, 1500, 1000], [4500, 3000, 0.05, 1500, 1000],
    [4500, 3000, 2, 1500, None], [333, 777, 1.0, 1500, None], [1, 1, 0.11, 7, None],
]


def build_tree(root):
    for rel in TREE:
        path = os.path.join(root, rel)
        if rel.endswith("/"):
            os.makedirs(path, exist_ok=True)
            continue
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write("x")


def rel(root, value):
    return os.path.relpath(value, root) if isinstance(value, str) else value


def main():
    out = {"tree": TREE, "cases": [], "scaling": [], "list_file": None}
    with tempfile.TemporaryDirectory() as root:
        build_tree(root)
        for fn, args in CASES:
            f = getattr(ref_path, fn, None) or getattr(ref_loader, fn)
            if fn == "get_path_from_exportdir":
                call = [os.path.join(root, args[0])] + args[1:]
            else:
                call = [os.path.join(root, a) for a in args]
            try:
                res = {"result": rel(root, f(*call))}
            except Exception as e:  # the type is the contract; messages are not compared
                res = {"raises": type(e).__name__}
            out["cases"].append({"fn": fn, "args": args, **res})
        lst = os.path.join(root, "pages.lst")
        text = "news/a.png\nnews/b.tif  \n\n  news/c.jpg\t\nlast"
        with open(lst, "w") as f:
            f.write(text)
        out["list_file"] = {"text": text, "load_list_file": ref_loader.load_list_file(lst)}
    for h, w, sf, fh, fw in SCALING:
        out["scaling"].append({"args": [h, w, sf, fh, fw], "result": ref_scaling(h, w, sf, fh, fw)})
    with open(os.path.join(HERE, "pathutil_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote pathutil_golden.json:", len(out["cases"]), "path cases,", len(out["scaling"]), "scaling cases")


if __name__ == "__main__":
    main()
