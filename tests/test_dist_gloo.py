"""N > 1 path on CPU: world_size-2 gloo processes shard images round-robin, all_gather their padded detections with the
same DetectionGatherer bench.py uses over RCCL, and every rank must end up with the single-process result."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from detectorch_amd.dist import DetectionGatherer, shard_images, unshard_order


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_dets(i, max_out):
    rs = np.random.RandomState(100 + i)
    n = int(rs.randint(0, max_out + 1))
    d = np.zeros((max_out, 6), np.float32)
    d[:n] = rs.rand(n, 6)
    return d, n


def _worker(rank, world, port, n_images, max_out, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_images(n_images, rank, world)
    per = (n_images + world - 1) // world
    dets = torch.zeros((per, max_out, 6))
    cnt = torch.zeros((per,), dtype=torch.int32)
    for j, i in enumerate(mine):
        d, n = _fake_dets(i, max_out)
        dets[j] = torch.from_numpy(d)
        cnt[j] = n
    g = DetectionGatherer(per, max_out, torch.device("cpu"), world)
    all_d, all_c = g.all_gather(dets, cnt)
    # the overlapped single-collective form bench.py uses must give the same answer, also after several steps in flight
    for step in range(3):
        g.all_gather_async(dets * float(step == 2), cnt * int(step == 2))
    a_d, a_c = g.finish()
    assert torch.equal(a_d, all_d) and torch.equal(a_c, all_c)
    order = unshard_order(n_images, world)
    flat_d = all_d.reshape(world * per, max_out, 6)[order]
    flat_c = all_c.reshape(world * per)[order]
    out_q.put((rank, flat_d.numpy().copy(), flat_c.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_helpers():
    assert shard_images(7, 0, 2) == [0, 2, 4, 6] and shard_images(7, 1, 2) == [1, 3, 5]
    o = unshard_order(8, 2)
    assert sorted(o) == list(range(8)) and o[:4] == [0, 4, 1, 5]


def test_all_gather_detections_world_size_2():
    world, n_images, max_out = 2, 6, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_images, max_out, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp_d = np.stack([_fake_dets(i, max_out)[0] for i in range(n_images)])
    exp_c = np.array([_fake_dets(i, max_out)[1] for i in range(n_images)], np.int32)
    for rank, d, c in res:
        assert np.array_equal(d, exp_d), rank
        assert np.array_equal(c, exp_c), rank
