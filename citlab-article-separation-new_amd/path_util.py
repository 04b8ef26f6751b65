"""Folder conventions of the pipelines: image <-> page/<name>.xml <-> json*/<name>.json.

Same behaviour as ``python_util/io/path_util.py`` and ``python_util/io/file_loader.py:23-42``."""
import glob
import os
import re


def load_list_file(path_to_list_file):
    with open(path_to_list_file) as f:
        return [line.rstrip() for line in f.readlines()]


def get_path_from_exportdir(model_dir, pattern, not_pattern):
    export_dir = os.path.join(model_dir, "export")
    names = [x for x in glob.glob1(export_dir, pattern) if not_pattern not in x]
    if len(names) == 1:
        return os.path.join(export_dir, names[0])
    raise IOError(f"Found {len(names)} '{pattern}' files in {export_dir}, there must be exact one.")


def _existing_image(stem_path):
    for ending in ("tif", "png", "jpg"):
        cand = f"{stem_path}.{ending}"
        if os.path.isfile(cand):
            return cand
    return None


def get_img_from_page_path(page_path):
    direct = re.sub(r'/page/([-\w.]+)\.xml$', r'/\1', page_path)
    if direct.endswith(("tif", "jpg", "png")) and os.path.isfile(direct):
        return direct
    img = _existing_image(direct)
    if img is None:
        raise IOError(f"No image file (tif, png, jpg) found to given pagexml {page_path}")
    return img


def get_img_from_json_path(json_path):
    direct = re.sub(r'/json\w*/([-\w.]+)\.json$', r'/\1', json_path)
    if direct.endswith(("tif", "jpg", "png")) and os.path.isfile(direct):
        return direct
    img = _existing_image(direct)
    if img is None:
        raise IOError("No image file (tif, png, jpg) found to given json ", json_path)
    return img


def get_page_from_img_path(img_path):
    page_path = re.sub(r'/([-\w.]+)$', r'/page/\1.xml', img_path)
    if os.path.isfile(page_path):
        return page_path
    page_path = re.sub(r'/([-\w.]+)\.\w+$', r'/page/\1.xml', img_path)
    if not os.path.isfile(page_path):
        raise IOError("No pagexml file found to given img file ", img_path)
    return page_path


def get_page_path(path_to_img):
    """file_loader.py:23-36: <dir>/<name>.<ext> -> <dir>/page/<name>.xml (existence not required)."""
    folder, name = os.path.split(path_to_img)
    return os.path.join(folder, "page", os.path.splitext(name)[0] + ".xml")


def get_page_from_json_path(json_path):
    page_path = re.sub(r'/json\w*/([-\w.]+)$', r'/page/\1.xml', json_path)
    if os.path.isfile(page_path):
        return page_path
    page_path = re.sub(r'/json\w*/([-\w.]+)\.json$', r'/page/\1.xml', json_path)
    if not os.path.isfile(page_path):
        raise IOError("No pagexml file found to given json file ", json_path)
    return page_path
