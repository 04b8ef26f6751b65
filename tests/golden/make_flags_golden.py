"""Generates tests/golden/flags_golden.json by IMPORTING the reference's flag system and helpers
(/root/reference/python_util/basic/flags.py, python_util/parser/xml/page/page_util.py) in the authoring container.

Run:  python tests/golden/make_flags_golden.py
"""
import json
import os
import sys
import tempfile

sys.path.insert(0, "/root/reference")
import python_util.basic.flags as flags  # noqa: E402
from python_util.parser.xml.page.page_util import format_custom_attr  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

ARGVS = [
    ["--input_params", "node_feature_dim=15", "edge_feature_dim=2",
     "node_input_feature_mask=[1,1,1,1,0,0,0,0,0,0,0,0,1,1,1]"],
    ["--input_params", "a=true", "b=F", "c=1.5", "d=3.0", "e=hello", "f=[t, 2, 2.5, x,]", "g=[]", "h=-4"],
    ["--input_params", "k=v=w", "plain", "x=1e3"],
    ["--clustering_params", "confidence_threshold=0.6", "min_neighbors_for_cluster=2", "method=centroid"],
    ["--input_params", "a=1", "--input_params", "b=2"],
]


def main():
    flags.define_dict("input_params", {}, "")
    flags.define_dict("clustering_params", {}, "")
    out = {"dict_flags": [], "config_file": None, "custom_attr": []}
    for argv in ARGVS:
        ns, _ = flags.global_parser.parse_known_args(argv)
        out["dict_flags"].append({"argv": argv, "input_params": ns.input_params, "clustering_params": ns.clustering_params})
    with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
        f.write("--input_params node_feature_dim = 15  edge_feature_dim=2   # trailing comment\n")
        f.write("--clustering_params epsilon=0.25\n")
        cfg = f.name
    ns, _ = flags.global_parser.parse_known_args(["@" + cfg])
    out["config_file"] = {"lines": open(cfg).read().splitlines(), "input_params": ns.input_params,
                          "clustering_params": ns.clustering_params}
    os.unlink(cfg)
    for d in [{"readingOrder": {"index": "1"}, "structure": {"id": "a3", "type": "article"}},
              {"structure": {"semantic_type": "heading"}}, {"structure": {"orientation": "horizontal"}}, {}]:
        out["custom_attr"].append({"dict": d, "string": format_custom_attr(d)})
    with open(os.path.join(HERE, "flags_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote flags_golden.json")


if __name__ == "__main__":
    main()
