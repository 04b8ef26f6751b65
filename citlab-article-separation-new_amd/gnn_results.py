"""What happens to the GNN output: confidences -> json, labels -> PAGE-XML.

Same file naming and content as ``article_separation/gnn/io.py:69-163``."""
import json
import logging
import os
import re

import numpy as np
from scipy.stats import gmean

from .page_xml import Page


def confidences_from_output(output, num_nodes):
    """run_gnn_clustering.py:272-279: class-1 probability of every ordered pair, reshaped [N, N]."""
    return np.reshape(np.asarray(output)[0, :, 1], [int(num_nodes), -1])


def save_conf_to_json(confidences, page_path, save_dir, symmetry_fn=gmean):
    page = Page(page_path)
    text_regions = page.get_regions().get("TextRegion", [])
    assert len(confidences) == len(text_regions), (
        f"Number of nodes in confidences ({len(confidences)}) does not match number of text regions "
        f"({len(text_regions)}) in {page_path}.")
    if symmetry_fn:
        confidences = symmetry_fn(np.stack([confidences, confidences.transpose()], axis=-1), axis=-1)
    conf = {ri.id: {rj.id: str(confidences[i, j]) for j, rj in enumerate(text_regions)}
            for i, ri in enumerate(text_regions)}
    save_name = os.path.splitext(os.path.basename(page_path))[0] + "_confidences.json"
    page_dir = re.sub(r'page$', 'confidences', os.path.dirname(page_path))
    out_dir = os.path.join(save_dir, page_dir) if save_dir else page_dir
    os.makedirs(out_dir, exist_ok=True)
    save_path = os.path.join(out_dir, save_name)
    with open(save_path, "w") as f:
        json.dump({"confidences": conf}, f)
    logging.info(f"Saved json with graph confidences '{save_path}'")
    return save_path


def save_clustering_to_page(clustering, page_path, save_dir, info=""):
    page = Page(page_path)
    text_regions = page.get_regions().get("TextRegion", [])
    assert len(clustering) == len(text_regions), (
        f"Number of nodes in clustering ({len(clustering)}) does not match number of text regions "
        f"({len(text_regions)}) in {page_path}.")
    for index, region in enumerate(text_regions):
        for line in region.text_lines:
            line.set_article_id(f"a{clustering[index]}")
    page.set_text_regions(text_regions, overwrite=True)
    rel = os.path.relpath(page_path)
    save_name = re.sub(r'\.xml$', '_clustering.xml', os.path.basename(rel))
    page_dir = re.sub(r'page$', 'clustering', os.path.dirname(rel))
    out_dir = os.path.join(save_dir, page_dir, info) if info else os.path.join(save_dir, page_dir)
    os.makedirs(out_dir, exist_ok=True)
    save_path = os.path.join(out_dir, save_name)
    page.write_page_xml(save_path)
    logging.info(f"Saved pageXML with graph clustering '{os.path.abspath(save_path)}'")
    return save_path
