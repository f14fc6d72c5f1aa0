"""world_size-2 gloo tests of the batch-sharding path (runs on CPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from implicit_depth_amd.dist import all_gather_metrics, nanmean_rows, shard_batch, shard_range


def test_shard_range_is_a_partition():
    for total in (0, 1, 5, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = {"x": torch.arange(B * 3, dtype=torch.float32).view(B, 3), "min_depth": torch.tensor(0.25).view(1, 1, 1, 1)}
        mine = shard_batch(full, world, rank, B)
        assert mine["min_depth"].shape == (1, 1, 1, 1)
        # "forward": per-frame metric rows = f(frame), one NaN to exercise the nan-mean
        rows = torch.stack([mine["x"].sum(1), mine["x"][:, 0] * 2], 1)
        if rank == 0 and rows.shape[0] > 0:
            rows[0, 1] = float("nan")
        g = all_gather_metrics(rows)
        # numpy payloads: torch tensors travel by file descriptor, which races with the worker's exit
        q.put((rank, g.numpy().copy(), nanmean_rows(g).numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5])
def test_two_rank_gather_matches_single_process(B):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = torch.arange(B * 3, dtype=torch.float32).view(B, 3)
    ref = torch.stack([x.sum(1), x[:, 0] * 2], 1)
    ref[0, 1] = float("nan")
    for rank, g, m in got:
        g, m = torch.from_numpy(g), torch.from_numpy(m)
        assert g.shape == ref.shape
        assert torch.equal(torch.nan_to_num(g, nan=-1), torch.nan_to_num(ref, nan=-1))
        assert torch.allclose(m, torch.nanmean(ref, 0))
