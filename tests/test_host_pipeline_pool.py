"""host_pipeline.py on the CPU: worker processes decode ahead into shared-memory slots (order of the list kept,
pixels identical to an inline decode, failures surfaced with the file name), write tasks run in worker processes."""
import os

import numpy as np
import pytest
from PIL import Image

from citlab_article_separation_new_amd import host_pipeline, image_io


def _touch(path, text):
    with open(path, "w") as f:
        f.write(text)


def test_decode_pool_keeps_order_and_pixels(tmp_path):
    rng = np.random.default_rng(0)
    paths, want = [], []
    for k in range(7):
        if k % 3 == 2:
            arr = rng.integers(0, 255, (40 + k, 30, 3), dtype=np.uint8)
        else:
            arr = rng.integers(0, 255, (33, 50 + k), dtype=np.uint8)
        p = tmp_path / f"img{k}.png"
        Image.fromarray(arr).save(p)
        paths.append(str(p))
        want.append(image_io.load_image_bgr(str(p)))
    pinned, released = [], []
    pool = host_pipeline.DecodePool(paths, n_workers=3, slot_bytes=1 << 16,
                                    register=lambda a, n: pinned.append((a, n)) or True, unregister=released.append)
    got = [(p, img.copy()) for p, img in pool]
    assert [p for p, _ in got] == paths
    for (_, img), ref in zip(got, want):
        assert img.dtype == np.uint8 and np.array_equal(img, ref)
    # (slots beyond the first two are prepared by a thread while pages are already consumed: a short list may end before all exist)
    assert 2 <= len(pinned) <= pool.n_slots and sorted(released) == sorted(a for a, _ in pinned)
    # inline mode gives the same
    inline = [img for _, img in host_pipeline.DecodePool(paths, n_workers=0)]
    assert all(np.array_equal(a, b) for a, b in zip(inline, want))


def test_hold_two_keeps_the_previous_image_valid(tmp_path):
    """hold = 2 (the GPU owner queues page n+1's upload before it waits for page n): an image handed out is still intact while
    the next one is in the consumer's hands -- with the fewest slots that allows (hold + 1) and more pages than slots, so every
    slot is recycled several times"""
    rng = np.random.default_rng(1)
    paths, want = [], []
    for k in range(9):
        arr = rng.integers(0, 255, (30, 40 + k), dtype=np.uint8)
        p = tmp_path / f"h{k}.png"
        Image.fromarray(arr).save(p)
        paths.append(str(p))
        want.append(arr)
    pool = host_pipeline.DecodePool(paths, n_workers=2, slot_bytes=1 << 14, n_slots=3, hold=2)
    assert host_pipeline.DecodePool(paths, n_workers=4, hold=2).n_slots == 7
    prev = None
    for k, (p, img) in enumerate(pool):
        assert np.array_equal(img, want[k])
        if prev is not None:
            assert np.array_equal(prev, want[k - 1])         # a view into its slot, not a copy
        prev = img
    assert k == 8


def test_decode_pool_reports_failures(tmp_path):
    ok = tmp_path / "a.png"
    Image.fromarray(np.zeros((4, 4), np.uint8)).save(ok)
    paths = [str(ok), str(tmp_path / "missing.png"), str(ok)]
    with pytest.raises(IOError, match="missing.png"):
        list(host_pipeline.DecodePool(paths, n_workers=2, slot_bytes=1 << 12))
    big = tmp_path / "big.png"
    Image.fromarray(np.zeros((100, 100), np.uint8)).save(big)
    with pytest.raises(IOError, match="exceeds"):
        list(host_pipeline.DecodePool([str(ok), str(big)], n_workers=2, slot_bytes=1 << 12, strict_slots=True))
    # default: an image beyond the slot is decoded inline by the owner (the reference processes such scans)
    pool = host_pipeline.DecodePool([str(ok), str(big), str(ok)], n_workers=2, slot_bytes=1 << 12)
    got = [img.copy() for _, img in pool]
    assert [g.shape for g in got] == [(4, 4), (100, 100), (4, 4)] and pool.inline_decodes == 1


def test_slots_are_sized_from_the_image_headers(tmp_path):
    small, large = tmp_path / "s.png", tmp_path / "l.png"
    Image.fromarray(np.zeros((10, 10), np.uint8)).save(small)
    Image.fromarray(np.zeros((700, 900, 3), np.uint8)).save(large)
    assert host_pipeline.needed_slot_bytes([str(small)] * 3) == 1 << 20
    assert host_pipeline.needed_slot_bytes([str(small), str(large)]) == 2 << 20          # 1.89 MB -> 2 MiB
    assert host_pipeline.needed_slot_bytes([str(large)], limit=1 << 20) == 1 << 20        # capped: that scan goes inline
    pool = host_pipeline.DecodePool([str(small), str(large), str(small)], n_workers=2)    # slot_bytes None
    assert [img.shape for _, img in pool] == [(10, 10), (700, 900, 3), (10, 10)] and pool.inline_decodes == 0


def _die_on_second(path):
    """loader that kills its own process on files named kill*: stands for an OOM kill / SIGBUS / decoder crash"""
    if os.path.basename(path).startswith("kill"):
        os.kill(os.getpid(), 9)
    return image_io.load_image_bgr(path)


def test_a_dead_decode_worker_raises_instead_of_hanging(tmp_path):
    """with one of two workers SIGKILLed while holding a page the owner used to spin forever."""
    import time
    paths = []
    for k in range(6):
        p = tmp_path / (f"kill{k}.png" if k == 2 else f"ok{k}.png")
        Image.fromarray(np.full((8, 8), k, np.uint8)).save(p)
        paths.append(str(p))
    t0 = time.time()
    with pytest.raises(RuntimeError, match="decode workers died.*kill2.png"):
        list(host_pipeline.DecodePool(paths, n_workers=2, loader=__name__ + ":_die_on_second", slot_bytes=1 << 12))
    assert time.time() - t0 < 60


def test_write_pool_runs_tasks_and_surfaces_errors(tmp_path):
    with host_pipeline.WritePool(2) as w:
        for k in range(5):
            w.submit(_touch, str(tmp_path / f"f{k}.txt"), f"page {k}")
    assert sorted(os.listdir(tmp_path)) == [f"f{k}.txt" for k in range(5)]
    assert (tmp_path / "f3.txt").read_text() == "page 3"
    with pytest.raises(FileNotFoundError):
        with host_pipeline.WritePool(2) as w:
            w.submit(_touch, str(tmp_path / "no_such_dir" / "x.txt"), "x")
    with host_pipeline.WritePool(0) as w:                    # inline
        w.submit(_touch, str(tmp_path / "inline.txt"), "i")
    assert (tmp_path / "inline.txt").read_text() == "i"


def test_effective_cpus_honours_the_cgroup_quota(tmp_path):
    """os.cpu_count() counts the machine; the pools are sized by what the container may use (cgroup v2 cpu.max, v1 cfs quota)"""
    from citlab_article_separation_new_amd import host_util
    allowed = len(os.sched_getaffinity(0))
    v2 = tmp_path / "v2"
    v2.mkdir()
    (v2 / "cpu.max").write_text("150000 100000\n")
    assert host_util.effective_cpus(str(v2)) == min(allowed, 1)
    (v2 / "cpu.max").write_text("1600000 100000\n")
    assert host_util.effective_cpus(str(v2)) == min(allowed, 16)
    (v2 / "cpu.max").write_text("max 100000\n")
    assert host_util.effective_cpus(str(v2)) == allowed
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("300000\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert host_util.effective_cpus(str(v1)) == min(allowed, 3)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert host_util.effective_cpus(str(v1)) == allowed
    assert host_util.effective_cpus(str(tmp_path / "none")) == allowed
    assert 1 <= host_pipeline.host_workers_default() <= 24


def test_workers_are_spawned_with_single_threaded_blas_and_the_environment_is_restored(monkeypatch):
    """numpy's BLAS creates a thread per core at import; decode / XML workers do no linear algebra and start with
    *_NUM_THREADS = 1 (on a 256-CPU box that was most of their start-up time) -- the parent's own settings come back afterwards"""
    monkeypatch.setenv("OMP_NUM_THREADS", "7")
    monkeypatch.delenv("OPENBLAS_NUM_THREADS", raising=False)
    with host_pipeline.single_threaded_children():
        assert os.environ["OMP_NUM_THREADS"] == os.environ["OPENBLAS_NUM_THREADS"] == os.environ["MKL_NUM_THREADS"] == "1"
        import multiprocessing as mp
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        p = ctx.Process(target=_report_env, args=(q,))
        p.start()
        assert q.get(timeout=60) == ("1", "1")
        p.join()
    assert os.environ["OMP_NUM_THREADS"] == "7" and "OPENBLAS_NUM_THREADS" not in os.environ


def _report_env(q):
    q.put((os.environ.get("OMP_NUM_THREADS"), os.environ.get("OPENBLAS_NUM_THREADS")))
