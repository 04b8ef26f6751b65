"""Folder-convention helpers and get_scaling_factor against vectors recorded from the reference
(tests/golden/make_pathutil_golden.py imports /root/reference/python_util/io/{path_util,file_loader}.py and
python_util/image_processing/image_stats.py)."""
import builtins
import json
import os

import pytest

from citlab_article_separation_new_amd import net_post_processing_helper as helper
from citlab_article_separation_new_amd import path_util

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pathutil_golden.json")))


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = tmp_path_factory.mktemp("tree")
    for rel in GOLDEN["tree"]:
        path = root / rel
        if rel.endswith("/"):
            path.mkdir(parents=True, exist_ok=True)
            continue
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text("x")
    return str(root)


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=lambda c: f"{c['fn']}:{c['args'][0]}")
def test_path_case(tree, case):
    fn = getattr(path_util, case["fn"])
    args = [os.path.join(tree, case["args"][0])] + case["args"][1:] if case["fn"] == "get_path_from_exportdir" \
        else [os.path.join(tree, a) for a in case["args"]]
    if "raises" in case:
        with pytest.raises(getattr(builtins, case["raises"])):
            fn(*args)
    else:
        assert os.path.relpath(fn(*args), tree) == case["result"]


def test_load_list_file(tmp_path):
    p = tmp_path / "pages.lst"
    p.write_text(GOLDEN["list_file"]["text"])
    assert path_util.load_list_file(str(p)) == GOLDEN["list_file"]["load_list_file"]


@pytest.mark.parametrize("case", GOLDEN["scaling"], ids=lambda c: str(c["args"]))
def test_scaling_factor(case):
    got = helper.get_scaling_factor(*case["args"])
    assert got == case["result"] and type(got) is type(case["result"])      # exact floats, None stays None
