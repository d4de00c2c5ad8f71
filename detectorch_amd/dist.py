"""Multi-GPU: one process per GPU, images sharded across ranks, ONE collective per step.

The reference is single-process, batch 1 (eval_mask_FPN.ipynb:93) with no cross-image state, so the path shards over
independent images: image i -> rank i mod W.  Every rank runs the full hot path on its own HBM;
the only exchange is an all_gather of the fixed-size padded detections (float32 [imgs_per_rank, max_out, 6] + int32
counts = ~3 KB/image) so that every rank (or rank 0) holds all detections for COCO scoring.  Over RCCL (backend "nccl" on
ROCm) on xGMI this is latency-bound; no all-reduce, no ring bucket tuning applies.  Works with gloo on CPU tensors too
(world_size-2 tests).
"""
import torch
import torch.distributed as dist


def shard_images(n_images, rank, world):
    """Indices of the images this rank owns (round-robin, image i -> rank i mod W)."""
    return list(range(rank, n_images, world))


def unshard_order(n_images, world):
    """Position in the gathered [world, imgs_per_rank] layout of global image i (for reassembling in dataset order)."""
    per = (n_images + world - 1) // world
    return [(i % world) * per + (i // world) for i in range(n_images)]


class DetectionGatherer:
    def __init__(self, imgs_per_rank, max_out, device, world, group=None):
        self.world, self.group = world, group
        self.all_dets = torch.zeros((world, imgs_per_rank, max_out, 6), dtype=torch.float32, device=device)
        self.all_counts = torch.zeros((world, imgs_per_rank), dtype=torch.int32, device=device)

    def all_gather(self, dets, counts):
        """dets [imgs_per_rank, max_out, 6], counts int32 [imgs_per_rank] -> (all_dets [W,...], all_counts [W,...])."""
        dets, counts = dets.contiguous(), counts.contiguous()
        if dist.get_backend(self.group) == "gloo":
            dl = list(self.all_dets.unbind(0))
            cl = list(self.all_counts.unbind(0))
            dist.all_gather(dl, dets, group=self.group)
            dist.all_gather(cl, counts, group=self.group)
            self.all_dets, self.all_counts = torch.stack(dl), torch.stack(cl)
        else:
            dist.all_gather_into_tensor(self.all_dets, dets, group=self.group)
            dist.all_gather_into_tensor(self.all_counts, counts, group=self.group)
        return self.all_dets, self.all_counts

    # ---- overlapped form: ONE collective per step, off the compute stream's critical path --------------------------------
    # all_gather() above makes the compute stream wait for two latency-bound collectives before the next step can start.
    # Here detections and counts are packed into one row per image ([max_out*6 + 1] floats; a count <= 2^24 is exact in
    # float32), copied into one of two staging buffers on the compute stream, and gathered with async_op=True: the collective
    # runs on the process group's own stream while the next step's kernels run, and a staging buffer is only waited for when
    # it comes round again two steps later (or in finish()).
    def all_gather_async(self, dets, counts):
        if not hasattr(self, "_pack"):
            per, width = dets.shape[0], dets.shape[1] * dets.shape[2] + 1     # last column: the count, int32 bits
            self._pack = [torch.zeros((per, width), dtype=torch.float32, device=dets.device) for _ in range(2)]
            self._all = [torch.zeros((self.world, per, width), dtype=torch.float32, device=dets.device) for _ in range(2)]
            self._work = [None, None]
            self._turn = 0
        s = self._turn
        if self._work[s] is not None:
            self._work[s].wait()                      # the gather that last used this staging buffer (two steps ago)
        self._pack[s][:, :-1].copy_(dets.reshape(dets.shape[0], -1))
        self._pack[s][:, -1:].view(torch.int32).copy_(counts.reshape(-1, 1))     # bit-cast, not a float conversion
        if dist.get_backend(self.group) == "gloo":
            self._work[s] = dist.all_gather(list(self._all[s].unbind(0)), self._pack[s], group=self.group, async_op=True)
        else:
            self._work[s] = dist.all_gather_into_tensor(self._all[s], self._pack[s], group=self.group, async_op=True)
        self._last = s
        self._turn = 1 - s

    def finish(self):
        """Wait for the outstanding gathers; returns (all_dets [W,imgs,max_out,6], all_counts int32 [W,imgs]) of the last step.
        all_counts are the TRUE per-image counts (dtc_postprocess_detections: they exceed max_out when scores tie at the image
        threshold or max_det <= 0); the rows that hold data are `self.n_valid` = min(count, max_out)."""
        for w in self._work:
            if w is not None:
                w.wait()
        self._work = [None, None]
        a = self._all[self._last]
        self.all_dets = a[:, :, :-1].reshape(self.all_dets.shape)
        self.all_counts = a[:, :, -1:].view(torch.int32).reshape(a.shape[0], a.shape[1]).clone()
        self.n_valid = self.all_counts.clamp(max=self.all_dets.shape[2])
        return self.all_dets, self.all_counts


class ResultGatherer:
    """Everything COCO scoring needs from every rank, in ONE collective: per image a byte record
        [ dets float32 max_out x 6 | det_count int32 | im_size float32 x 2 | rle_len int32 max_out | rle_str uint8 max_out x str_stride ]
    (the masks travel as the COCO RLE count strings dtc_mask_rle produced on the device: ~100 bytes per mask instead of a
    bitmap), gathered with all_gather_into_tensor over RCCL (gloo: all_gather of the same byte tensors).  `str_stride` caps
    the string length that is shipped; a longer string (rle_len > str_stride) is reported by finish()."""

    def __init__(self, imgs_per_rank, max_out, device, world, str_stride=512, group=None):
        self.per, self.D, self.world, self.group, self.stride = imgs_per_rank, max_out, world, group, int(str_stride)
        assert (max_out * self.stride) % 4 == 0, "max_out * str_stride must be a multiple of 4 (the record is viewed as int32 / float32)"
        self.o_cnt = max_out * 24
        self.o_sz = self.o_cnt + 4
        self.o_len = self.o_sz + 8
        self.o_str = self.o_len + 4 * max_out
        self.width = self.o_str + max_out * self.stride
        self.pack = torch.zeros((imgs_per_rank, self.width), dtype=torch.uint8, device=device)
        self.all = torch.zeros((world, imgs_per_rank, self.width), dtype=torch.uint8, device=device)

    def all_gather(self, dets, counts, im_size, rle_str=None, rle_len=None):
        p, D = self.pack, self.D
        p[:, :self.o_cnt].view(torch.float32).copy_(dets.reshape(self.per, D * 6))
        p[:, self.o_cnt:self.o_sz].view(torch.int32).copy_(counts.reshape(self.per, 1))
        p[:, self.o_sz:self.o_len].view(torch.float32).copy_(im_size.reshape(self.per, 2).to(torch.float32))
        if rle_str is not None:
            p[:, self.o_len:self.o_str].view(torch.int32).copy_(rle_len.reshape(self.per, D))
            k = min(self.stride, rle_str.shape[2])
            p[:, self.o_str:].view(self.per, D, self.stride)[:, :, :k].copy_(rle_str[:, :, :k])
        else:
            p[:, self.o_len:self.o_str].zero_()
        if dist.get_backend(self.group) == "gloo":
            parts = list(self.all.unbind(0))
            dist.all_gather(parts, p, group=self.group)
            self.all = torch.stack(parts)
        else:
            dist.all_gather_into_tensor(self.all, p, group=self.group)
        return self.finish()

    def finish(self):
        """-> dict(dets [W*per, D, 6], det_count [W*per], n_valid [W*per] = min(det_count, D), im_size [W*per, 2], rle_len [W*per, D], rle_str [W*per, D, stride]) in
        (rank, local image) order; use unshard_order() to put the images back in dataset order."""
        a, D, n = self.all.reshape(-1, self.width), self.D, self.world * self.per
        out = dict(dets=a[:, :self.o_cnt].contiguous().view(torch.float32).reshape(n, D, 6),
                   det_count=a[:, self.o_cnt:self.o_sz].contiguous().view(torch.int32).reshape(n),
                   im_size=a[:, self.o_sz:self.o_len].contiguous().view(torch.float32).reshape(n, 2),
                   rle_len=a[:, self.o_len:self.o_str].contiguous().view(torch.int32).reshape(n, D),
                   rle_str=a[:, self.o_str:].reshape(n, D, self.stride))
        out["n_valid"] = out["det_count"].clamp(max=D)     # det_count is the true count; only max_out rows carry data
        out["truncated"] = bool((out["rle_len"] > self.stride).any())
        return out
