"""One process per GPU: page sharding and the start-up weight broadcast.

Pages have no cross-page dependency in either net (separator_net_post_processor.py:141-159,
run_gnn_clustering.py:237-300), so ranks own contiguous sub-lists of the page list exactly like the
reference's worker processes (python_util/basic/misc.py:4-7 split_list) and never exchange activations.
The only collective is a broadcast of the packed weight blobs from rank 0 (RCCL over xGMI on the GPU box,
gloo in the CPU tests).
"""
import numpy as np
import torch

from .host_util import split_list


def shard_pages(pages, world_size: int, rank: int):
    """Contiguous shard of `pages` owned by `rank` (sizes differ by at most one page)."""
    return split_list(list(pages), world_size)[rank]


def broadcast_blob(blob, rank: int, device=None, src: int = 0, group=None) -> bytes:
    """Every rank returns rank `src`'s bytes.  Two broadcasts: length (int64), then payload (uint8)."""
    import torch.distributed as dist
    device = device or torch.device("cpu")
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src, group=group)
    if rank == src:
        buf = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(device)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src, group=group)
    return buf.cpu().numpy().tobytes()


def max_over_ranks(value: float, device=None, group=None) -> float:
    """Timing reduction used by bench.py (the slowest rank defines the step time)."""
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device or torch.device("cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
