"""Small host-side helpers of the two pipelines (same names and results as the reference).

    split_list       python_util/basic/misc.py:4-7     page list -> n contiguous chunks (process / GPU sharding)
    rescale_points   python_util/geometry/point.py:1-11  polygon rescaling with int() truncation
    effective_cpus   the CPUs this process may actually use (affinity mask and cgroup quota), for sizing worker pools
"""
import math
import os


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


def _cgroup_quota_cpus(root="/sys/fs/cgroup"):
    """CPU quota of this container in CPUs (cgroup v2 ``cpu.max``, v1 ``cpu.cfs_quota_us`` / ``cpu.cfs_period_us``) or None"""
    try:
        with open(os.path.join(root, "cpu.max")) as f:
            quota, period = f.read().split()[:2]
        if quota != "max" and int(period) > 0:
            return int(quota) / int(period)
        return None
    except (OSError, ValueError):
        pass
    try:
        with open(os.path.join(root, "cpu", "cpu.cfs_quota_us")) as f:
            quota = int(f.read())
        with open(os.path.join(root, "cpu", "cpu.cfs_period_us")) as f:
            period = int(f.read())
        return quota / period if quota > 0 and period > 0 else None
    except (OSError, ValueError):
        return None


def effective_cpus(cgroup_root="/sys/fs/cgroup"):
    """``os.cpu_count()`` counts the machine; a container may be allowed far less (a GPU box of this pool shows 256 logical
    CPUs under a quota of 16).  More busy workers than that do not run faster -- the quota throttles the whole group in every
    scheduling period, the GPU-owning process included."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = _cgroup_quota_cpus(cgroup_root)
    if quota is not None:
        n = min(n, max(1, math.floor(quota + 1e-9)))
    return max(1, n)
