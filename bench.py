#!/usr/bin/env python3
"""Contract benchmark: images/sec of the Mask R-CNN R-50-FPN region-proposal hot path on 1333x800 COCO-shaped synthetic
input (BASELINE.json metric, config[2]: "Mask R-CNN R-50-FPN, 1 MI355X, 4-level RoIAlign + 14x14 mask head").

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one pass of the whole hot path (detectorch_amd.pipeline.FpnRegionPath: GenerateProposals x5 levels, NMS,
collect/distribute, 4-level RoIAlign 7x7, detection post-processing, mask-branch RoIAlign 14x14, mask resize/binarise)
over one batch of --batch images per GPU, inputs already resident in HBM.  The ResNet/FPN convs and the box/mask-head
GEMMs are not part of the path (they stay on MIOpen/hipBLASLt); their outputs are synthetic tensors of the right shape.
Images shard across ranks with no data-path collective ("weak" scaling: fixed images per GPU); the only exchange is one
RCCL all_gather of the padded detections per step (detectorch_amd.dist), which IS inside the timed region.

Prints ONE JSON line on rank 0 (fields: see the task contract + `roofline` + `cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step (BASELINE cfg4: 8 images/GPU)")
    ap.add_argument("--eager", action="store_true", help="launch kernels eagerly instead of replaying the hipGraph")
    ap.add_argument("--channels-last", action="store_true", help="NHWC feature maps (same logical shape)")
    ap.add_argument("--fp16", action="store_true", help="fp16 feature maps / pooled features (BASELINE cfg5 flavour)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=8, help="images of the same workload timed on the CPU oracle")
    ap.add_argument("--kernel-iters", type=int, default=20)
    ap.add_argument("--split", type=int, default=1, help="sub-batches run on separate HIP streams inside one hipGraph")
    return ap.parse_args()


def _cpu_all_cores(jobs, budget_s=60.0):
    """All-cores figure: P = min(host cores, 32) worker processes (oracle/cpu_worker.py, one image each, the
    n distinct images reused round-robin), inputs handed over as memory-mapped .npy files, a file barrier, wall clock from
    the common start to the last finish.  Returns images/sec over all workers."""
    import shutil
    import subprocess
    import tempfile
    n = len(jobs)
    procs = max(1, min(os.cpu_count() or 1, 32))
    need = sum(sum(a.nbytes for a in (j[0] + j[1] + j[2])) + j[3].nbytes + j[4].nbytes + j[5].nbytes for j in jobs) + (1 << 20)
    base = None       # default temp dir unless /dev/shm has room for the inputs (containers often cap it at 64 MB)
    if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 2 * need:
        base = "/dev/shm"
    d = tempfile.mkdtemp(prefix="dtc_cpu_", dir=base)
    ws = []
    try:
        for k, (rc, rb, ft, score, pred, masks, sf, imsz, ph, pw) in enumerate(jobs):
            for l in range(5):
                np.save(os.path.join(d, "img%d_cls%d.npy" % (k, l)), rc[l]); np.save(os.path.join(d, "img%d_bbox%d.npy" % (k, l)), rb[l])
            for l in range(4):
                np.save(os.path.join(d, "img%d_feat%d.npy" % (k, l)), ft[l])
            for name, arr in (("score", score), ("pred", pred), ("masks", masks), ("imsize", imsz)):
                np.save(os.path.join(d, "img%d_%s.npy" % (k, name)), arr)
            np.save(os.path.join(d, "img%d_meta.npy" % k), np.array([sf, ph, pw], np.float64))
        env = dict(os.environ, OMP_NUM_THREADS="1")
        for w in range(procs):
            ws.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), d, str(w % n), str(w)],
                                       env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        t_end = time.time() + budget_s
        while sum(os.path.exists(os.path.join(d, "ready%d" % w)) for w in range(procs)) < procs:
            if time.time() > t_end or any(p.poll() not in (None, 0) for p in ws):
                raise RuntimeError("workers did not become ready")
            time.sleep(0.01)
        open(os.path.join(d, "go"), "w").close()
        while sum(os.path.exists(os.path.join(d, "done%d" % w)) for w in range(procs)) < procs:
            if time.time() > t_end:
                raise RuntimeError("workers did not finish")
            time.sleep(0.005)
        spans = [tuple(float(v) for v in open(os.path.join(d, "done%d" % w)).read().split()) for w in range(procs)]
        wall = max(e for _, e in spans) - min(s for s, _ in spans)
        return {"value": round(procs / wall, 4), "unit": "images/sec", "processes": procs,
                "note": "%d worker processes x 1 image each (the %d sample images round-robin), common start, wall clock to the "
                        "last finish; single-threaded oracle per process" % (procs, n)}
    finally:
        for p in ws:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(d, ignore_errors=True)


def cpu_baseline(inputs, path, n_images):
    """Time the oracle (a plain-C port of the reference's CPU path, oracle/oracle.c) on the first n_images images of the
    same workload, single thread -- the reference itself is single-threaded (OpenMP pragma commented out at
    lib/cppcuda/roi_align_cpu.cpp:136-137; Cython loops are serial).  Also reported: the same images run
    image-parallel, one process per image on all host cores.  The oracle is the CHECKER, timed here as a reported baseline
    only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import chain  # noqa: E402  (test infrastructure: allowed in the cpu_baseline leg only)
    rpn_cls, rpn_bbox, feats, cls_score, bbox_pred, masks, sf, im_size = inputs
    n = min(n_images, path.B)
    host = lambda t: t.float().cpu().numpy()
    jobs = [([host(c[b]) for c in rpn_cls], [host(d[b]) for d in rpn_bbox], [host(f[b:b + 1]) for f in feats],
             host(cls_score[b]), host(bbox_pred[b]), host(masks[b * path.max_out:(b + 1) * path.max_out]),
             float(sf[b]), host(im_size[b]), path.pad_h, path.pad_w) for b in range(n)]
    T = {}
    t0 = time.perf_counter()
    for job in jobs:
        chain.fpn_hot_path(*job, timings=T)
    dt = time.perf_counter() - t0
    conv = sum(T.values())
    out = {"value": round(n / conv, 4), "unit": "images/sec", "cores": 1, "kind": "port",
           "sample": "%d images of the same synthetic cfg3 workload (R=1000, C=256), oracle/oracle.c via ctypes, "
                     "%.1f s CPU; per-stage s/img: %s" % (n, dt, {k: round(v / n, 4) for k, v in T.items()}),
           "host_cpus": os.cpu_count()}
    try:   # informational; the single-core figure above is the contract
        out["all_cores"] = _cpu_all_cores(jobs)
    except Exception as e:
        out["all_cores"] = {"error": repr(e)}
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from detectorch_amd import hip
    from detectorch_amd.pipeline import FpnRegionPath, OverlappedRegionPath, synthetic_batch
    hip.lib()   # fails loudly if the native library is missing
    fdt = torch.float16 if a.fp16 else torch.float32
    if a.split > 1 and a.batch % a.split == 0:
        path = OverlappedRegionPath(a.batch, dev, n_split=a.split, feat_dtype=fdt)
    else:
        path = FpnRegionPath(a.batch, dev, feat_dtype=fdt)
    inputs = synthetic_batch(a.batch, dev, seed=3000 + rank, feat_dtype=fdt, channels_last=a.channels_last)
    path.bind(*inputs)
    gather = None
    if world > 1:
        from detectorch_amd.dist import DetectionGatherer
        gather = DetectionGatherer(path.B, path.max_out, dev, world)

    def one_step():
        path.step(use_graph=not a.eager)
        if gather is not None:
            gather.all_gather_async(path.dets, path.det_count)      # one packed collective per step, overlapped with the next

    for _ in range(a.warmup):
        one_step()
    if gather is not None:
        gather.finish()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    if gather is not None:
        gather.finish()                  # the last steps' collectives complete INSIDE the timed region
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- roofline of the dominant kernel: multi-level RoIAlign 7x7 (box head), HIP events on the launch stream -----
    iters = a.kernel_iters
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    path._roi_align_box()
    torch.cuda.synchronize(dev)
    for i in range(iters):
        e0[i].record()
        path._roi_align_box()
        e1[i].record()
    torch.cuda.synchronize(dev)
    k_ms = float(np.mean([e0[i].elapsed_time(e1[i]) for i in range(iters)]))
    alg_bytes = path.box_roialign_bytes()
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9

    traffic = None
    try:   # HBM bytes per launch of the same kernel/config from the committed rocprofv3 --pmc passes (profiles/README.md)
        tj = json.load(open(os.path.join(ROOT, "profiles", "roialign_traffic.json")))
        key = "b%d_%s_%s" % (a.batch, "nhwc" if a.channels_last else "nchw", "f16" if a.fp16 else "f32")
        if key in tj:
            traffic = tj[key]
    except Exception:
        traffic = None

    if rank == 0:
        n_img = a.batch * a.steps * world
        out = {
            "metric": "images/sec Mask R-CNN R-50-FPN 1333x800 (region-proposal hot path)",
            "value": round(n_img / dt, 2), "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if a.fp16 else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: Mask R-CNN R-50-FPN, 1x3x800x1333 (padded 800x1344), 5-level RPN "
                                   "(268569 anchors) -> 1000 rois, 4-level RoIAlign 7x7 sr2 C256, 81-class postprocess, "
                                   "RoIAlign 14x14 mask branch, 28x28 mask paste",
                       "images_per_gpu_per_step": a.batch, "global_batch": a.batch * world, "rois_per_image": 1000,
                       "feature_layout": "NHWC" if a.channels_last else "NCHW", "launch": ("eager" if a.eager else "hipGraph") + (", %d sub-batches on %d streams" % (a.split, a.split) if isinstance(path, OverlappedRegionPath) else ""),
                       "parallelism": "images sharded over %d GPU(s); all_gather of detections" % world,
                       "not_in_path": "ResNet-50/FPN convs and box/mask-head GEMMs (MIOpen/hipBLASLt), outputs synthetic"},
            "roofline": {"bound": "hbm", "kernel": "roi_align_fwd (box head, 4 levels, %d rois)" % (a.batch * 1000),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(k_ms, 4)},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(inputs, path, a.cpu_images)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
