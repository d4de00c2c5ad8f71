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
