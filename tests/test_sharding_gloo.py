"""N > 1 path on CPU: two gloo ranks shard a page list, receive rank 0's weights and agree on max-over-ranks."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from citlab_article_separation_new_amd import sharding
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights, pack_blob, unpack_blob
    cfg = AruConfig(scale_space_num=2, num_scales_att=1, graph="RU")
    blob = pack_blob(init_aru_weights(cfg, 1234)) if rank == 0 else b""
    got = sharding.broadcast_blob(blob, rank)
    w = unpack_blob(got)
    pages = [f"page_{i:03d}.png" for i in range(11)]
    mine = sharding.shard_pages(pages, world, rank)
    tmax = sharding.max_over_ranks(1.0 + rank)
    q.put((rank, len(got), float(sum(float(v.sum()) for v in w.values())), mine, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_weight_broadcast():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, s0, p0, t0), (r1, n1, s1, p1, t1) = res
    assert n0 == n1 > 0 and s0 == s1                     # identical weights on both ranks
    assert p0 + p1 == [f"page_{i:03d}.png" for i in range(11)] and len(p0) == 6 and len(p1) == 5
    assert t0 == t1 == 2.0                                # max over ranks
