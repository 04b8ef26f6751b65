"""Small host-side helpers of the two pipelines (same names and results as the reference).

    split_list       python_util/basic/misc.py:4-7     page list -> n contiguous chunks (process / GPU sharding)
    rescale_points   python_util/geometry/point.py:1-11  polygon rescaling with int() truncation
"""


def split_list(list_to_split, n):
    """n chunks whose sizes differ by at most one; the first ``len % n`` chunks are the longer ones."""
    base, extra = divmod(len(list_to_split), n)
    out, start = [], 0
    for i in range(n):
        stop = start + base + (1 if i < extra else 0)
        out.append(list_to_split[start:stop])
        start = stop
    return out


def rescale_points(points, scale):
    """(x, y) points times ``scale``, truncated towards zero."""
    return [(int(px * scale), int(py * scale)) for (px, py) in points]
