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


# ---- everything COCO scoring needs (detections + device RLE strings) in one collective, then all_boxes / all_segms -------
def _fake_image(i, max_out, stride):
    rs = np.random.RandomState(500 + i)
    n = int(rs.randint(0, max_out + 1))
    d = np.zeros((max_out, 6), np.float32)
    d[:n, :4] = rs.rand(n, 4) * 300
    d[:n, 4] = rs.rand(n)
    d[:n, 5] = np.sort(rs.randint(1, 6, n))                     # class-major like dtc_postprocess_detections
    ln = np.zeros(max_out, np.int32)
    st = np.zeros((max_out, stride), np.uint8)
    for k in range(n):
        ln[k] = rs.randint(1, stride + 1)
        st[k, :ln[k]] = rs.randint(48, 112, ln[k])
    return d, n, ln, st, np.array([200 + i, 300 + i], np.float32)


def _worker_results(rank, world, port, n_images, max_out, stride, out_q):
    from detectorch_amd.dist import ResultGatherer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_images(n_images, rank, world)
    per = (n_images + world - 1) // world
    dets, cnt = torch.zeros((per, max_out, 6)), torch.zeros((per,), dtype=torch.int32)
    ln, st, sz = torch.zeros((per, max_out), dtype=torch.int32), torch.zeros((per, max_out, stride), dtype=torch.uint8), torch.zeros((per, 2))
    for j, i in enumerate(mine):
        d, n, l, s, z = _fake_image(i, max_out, stride)
        dets[j], cnt[j], ln[j], st[j], sz[j] = torch.from_numpy(d), n, torch.from_numpy(l), torch.from_numpy(s), torch.from_numpy(z)
    g = ResultGatherer(per, max_out, torch.device("cpu"), world, str_stride=stride)
    res = g.all_gather(dets, cnt, sz, st, ln)
    order = unshard_order(n_images, world)
    out_q.put((rank, {k: (v[order].numpy().copy() if torch.is_tensor(v) else v) for k, v in res.items()}))
    # counts above 2^24 survive the packed DetectionGatherer (int32 bit-cast, not a float conversion)
    dg = DetectionGatherer(per, max_out, torch.device("cpu"), world)
    big = torch.full((per,), 16777217 + rank, dtype=torch.int32)
    dg.all_gather_async(dets, big)
    _, c = dg.finish()
    assert c[rank].tolist() == [16777217 + rank] * per
    dist.barrier()
    dist.destroy_process_group()


def test_gather_results_and_assemble_world_size_2():
    from detectorch_amd.utils import result_utils
    world, n_images, max_out, stride = 2, 5, 12, 40
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_results, args=(r, world, port, n_images, max_out, stride, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fake = [_fake_image(i, max_out, stride) for i in range(n_images)]
    # single-process expectation: the reference's per-image bookkeeping (empty_results + extend_results per image)
    exp_boxes, exp_segms, _ = result_utils.empty_results(6, n_images)
    for i, (d, n, ln, st, sz) in enumerate(fake):
        cls_boxes = [[] for _ in range(6)]
        cls_segms = [[] for _ in range(6)]
        for j in range(1, 6):
            sel = np.flatnonzero(d[:n, 5] == j)
            cls_boxes[j] = d[sel, :5]
            cls_segms[j] = [{'size': [int(sz[0]), int(sz[1])], 'counts': st[k, :ln[k]].tobytes().decode('ascii')} for k in sel]
        result_utils.extend_results(i, exp_boxes, cls_boxes)
        result_utils.extend_results(i, exp_segms, cls_segms)
    for rank, r in res:
        assert not r["truncated"]
        boxes, segms = result_utils.assemble_results(r["dets"][:n_images], r["det_count"][:n_images], r["im_size"][:n_images],
                                                     r["rle_str"][:n_images], r["rle_len"][:n_images], num_classes=6)
        for j in range(1, 6):
            for i in range(n_images):
                assert np.array_equal(boxes[j][i], exp_boxes[j][i]) and segms[j][i] == exp_segms[j][i], (rank, j, i)
        recs = result_utils.coco_segm_results(boxes, segms, list(range(100, 100 + n_images)), {j: 10 * j for j in range(6)})
        assert len(recs) == sum(f[1] for f in fake) and all(set(x) == {'image_id', 'category_id', 'segmentation', 'score'} for x in recs)
        brecs = result_utils.coco_bbox_results(boxes, list(range(100, 100 + n_images)), {j: 10 * j for j in range(6)})
        assert len(brecs) == len(recs) and all(len(x['bbox']) == 4 for x in brecs)
        k0 = next(x for x in brecs if x['image_id'] == 100)
        d0 = exp_boxes[k0['category_id'] // 10][0][0].astype(np.float64)
        assert k0['bbox'] == [d0[0], d0[1], d0[2] - d0[0] + 1, d0[3] - d0[1] + 1]
