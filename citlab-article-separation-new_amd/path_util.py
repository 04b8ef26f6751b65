"""Folder conventions of the pipelines:

    <dir>/<name>.{tif,png,jpg}  <->  <dir>/page/<name>.xml  <->  <dir>/json<tag>/<name>.json
                                <->  <dir>/confidences/<name>_confidences.json

Restates the lookups of ``python_util/io/path_util.py:6-88`` and ``python_util/io/file_loader.py:23-42`` as one
"sibling" rule: strip a (sub-folder, suffix) pair from the tail of the path, add another pair, probe the file system.
Results and exception types are pinned by ``tests/golden/pathutil_golden.json`` (recorded from the reference)."""
import fnmatch
import os
import re

_STEM = r"([-\w.]+)"                    # file-name alphabet the reference's patterns accept
_IMAGE_SUFFIXES = ("tif", "png", "jpg")  # probe order when the image suffix is unknown


def load_list_file(path_to_list_file):
    """One path per line, right-stripped (empty lines are kept as '')."""
    with open(path_to_list_file) as f:
        return [line.rstrip() for line in f]


def _tail(path, folder, suffix):
    """(head, stem) if ``path`` ends in ``/<folder>/<stem><suffix>`` else None; ``folder`` / ``suffix`` are regex
    fragments, an empty ``folder`` means "directly in the directory"."""
    sub = f"/{folder}" if folder else ""
    m = re.search(f"{sub}/{_STEM}{suffix}$", path)
    return None if m is None else (path[:m.start()], m.group(1))


def _image_sibling(path, folder, suffix, error):
    """Image next to the sub-folder ``folder`` holding ``path``: a stem that already carries an image suffix is taken
    as it is when that file exists, otherwise .tif / .png / .jpg are probed in this order.  A path that does not follow
    the convention is returned unchanged if it is a file."""
    hit = _tail(path, folder, suffix)
    if hit is None:
        if os.path.isfile(path):
            return path
        raise error
    head, stem = hit
    plain = f"{head}/{stem}"
    if plain.endswith(_IMAGE_SUFFIXES) and os.path.isfile(plain):
        return plain
    for ext in _IMAGE_SUFFIXES:
        if os.path.isfile(f"{plain}.{ext}"):
            return f"{plain}.{ext}"
    raise error


def _page_sibling(path, folder, suffix, error):
    """page/<file name>.xml if it exists (suffix appended), else page/<stem>.xml with ``suffix`` replaced."""
    appended = _tail(path, folder, "")
    if appended is not None:
        cand = f"{appended[0]}/page/{appended[1]}.xml"
        if os.path.isfile(cand):
            return cand
    replaced = _tail(path, folder, suffix)
    cand = path if replaced is None else f"{replaced[0]}/page/{replaced[1]}.xml"
    if os.path.isfile(cand):
        return cand
    raise error


def get_img_from_page_path(page_path):
    return _image_sibling(page_path, "page", r"\.xml",
                          IOError(f"no tif/png/jpg image belongs to PAGE-XML {page_path}"))


def get_img_from_json_path(json_path):
    return _image_sibling(json_path, r"json\w*", r"\.json", IOError(f"no tif/png/jpg image belongs to json {json_path}"))


def get_page_from_img_path(img_path):
    return _page_sibling(img_path, "", r"\.\w+", IOError(f"no PAGE-XML belongs to image {img_path}"))


def get_page_from_json_path(json_path):
    return _page_sibling(json_path, r"json\w*", r"\.json", IOError(f"no PAGE-XML belongs to json {json_path}"))


def get_page_from_conf_path(json_path):
    hit = _tail(json_path, "confidences", r"_confidences\.json")
    cand = json_path if hit is None else f"{hit[0]}/page/{hit[1]}.xml"
    if not os.path.isfile(cand):
        raise IOError(f"no PAGE-XML belongs to confidence file {json_path}")
    return cand


def get_page_path(path_to_img, page_folder_name="page", append_extension=False):
    """<dir>/<name>.<ext> -> <dir>/page/<name>.xml (nothing is probed)."""
    folder, name = os.path.split(path_to_img)
    if not append_extension:
        name = os.path.splitext(name)[0]
    return os.path.join(folder, page_folder_name, name + ".xml")


def get_path_from_exportdir(model_dir, pattern, not_pattern):
    """The one file under <model_dir>/export matching the glob ``pattern`` whose name does not contain
    ``not_pattern``; anything but exactly one candidate is an IOError."""
    export_dir = os.path.join(model_dir, "export")
    try:
        entries = os.listdir(export_dir)
    except OSError:
        entries = []
    keep = [n for n in fnmatch.filter(entries, pattern)
            if not_pattern not in n and (pattern.startswith(".") or not n.startswith("."))]
    if len(keep) != 1:
        raise IOError(f"{export_dir}: {len(keep)} files match '{pattern}' (without '{not_pattern}'), need exactly one")
    return os.path.join(export_dir, keep[0])


def prepend_folder_name(file_path):
    folder = os.path.dirname(file_path)
    return os.path.join(folder, f"{os.path.basename(folder)}_{os.path.basename(file_path)}")
