#!/usr/bin/env python3
"""Contract benchmark: images/sec of the Mask R-CNN R-50-FPN region-proposal hot path on 1333x800 COCO-shaped synthetic
input (BASELINE.json metric; default workload = configs[2]: "Mask R-CNN R-50-FPN, 1 MI355X, 4-level RoIAlign + 14x14 mask
head").

    python bench.py --gpus N --steps K --warmup W [--workload cfg3|cfg5|cfg2]

With N > 1 and no launcher environment the script re-executes itself under `python -m torch.distributed.run` with N ranks
(one per GPU, RCCL); launched by the driver's own torchrun line it just uses the environment it is given.  `n_gpus` in the
output is the size of the process group that actually ran.

One "step" = one pass of the whole hot path (detectorch_amd.pipeline.FpnRegionPath: GenerateProposals x5 levels, NMS,
collect/distribute, 4-level RoIAlign 7x7, detection post-processing, mask-branch RoIAlign 14x14, mask resize/binarise)
over one batch of --batch images per GPU, inputs already resident in HBM; consecutive steps alternate between TWO bound
input sets (different seeds), so no step re-reads what the previous one left in the caches.  The ResNet/FPN convs and the
box/mask-head GEMMs are not part of the path (they stay on MIOpen/hipBLASLt); their outputs are synthetic tensors of the
right shape.  Images shard across ranks with no data-path collective ("weak" scaling: fixed images per GPU); the only
exchange is one RCCL all_gather of the padded detections per step (detectorch_amd.dist), INSIDE the timed region.

Workloads (BASELINE.json configs):
  cfg3 (default)  configs[2]  Mask R-CNN R-50-FPN, 1000 rois / image, fp32 NCHW features
  cfg5            configs[4]  same path, 2000 proposals / image (collect top-N 2000) and fp16 feature maps / pooled features,
                              channels_last maps by default (--nchw: NCHW fp16 maps)
  cfg2            configs[1]  Faster R-CNN R-50-C4: 63 000 anchors -> 6000 -> NMS -> 1000 proposals, RoIAlign 7x7
                              (adaptive sampling) on res4 [B,1024,50,84], per-class NMS (detectorch_amd.pipeline.C4RegionPath)

Prints ONE JSON line on rank 0 (fields: the task contract + `roofline` + `cpu_baseline` + `consistency`).
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
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
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["cfg3", "cfg5", "cfg2"], default="cfg3")
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step (BASELINE cfg4: 8 images/GPU)")
    ap.add_argument("--eager", action="store_true", help="launch kernels eagerly instead of replaying the hipGraph")
    ap.add_argument("--channels-last", action="store_true", help="NHWC feature maps (same logical shape); the default for cfg5")
    ap.add_argument("--nchw", action="store_true", help="cfg5 only: NCHW fp16 feature maps instead of channels_last")
    ap.add_argument("--fp16", action="store_true", help="fp16 feature maps / pooled features (cfg5 sets this itself)")
    ap.add_argument("--max-out", type=int, default=104, help="fixed detection rows per image through the mask branch: max_detections_per_img = 100 plus room for ties at the image threshold (the reference keeps them, result_utils.py:159-163; more ties than rows raise); 128 until round 3")
    ap.add_argument("--c4-pooled", type=int, default=7, help="cfg2: pooled size (7 as BASELINE names it; 14 = the reference's C4 default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-modes", action="store_true", help="skip the extra launches of the other RoIAlign modes (cfg2 fast / bf16 output, cfg5 exact) and of the harder RoI set: counter and kernel-trace runs want ONE kernel and ONE RoI population per grid")
    ap.add_argument("--gather-always", action="store_true", help="run the per-step RCCL all-gather of the detections even at world size 1 (tests: exercises the N > 1 code path on one GPU; needs a launcher environment)")
    ap.add_argument("--cpu-images", type=int, default=8, help="images of the same workload run on the CPU oracle (timed + compared with the GPU); 2 when --gpus > 1")
    ap.add_argument("--cpu-procs", type=int, default=64, help="worker processes of the image-parallel CPU figure (capped by the host's cores)")
    ap.add_argument("--kernel-iters", type=int, default=20)
    ap.add_argument("--sustain-seconds", type=float, default=1.0, help="extra untimed-by-contract run of at least this long, reported under `consistency`")
    ap.add_argument("--inflight", type=int, default=2, help="steps in flight: consecutive steps (independent batches) are issued round-robin on this many HIP streams, so the latency-bound kernels of one step overlap the RoIAlign launches of another")
    ap.add_argument("--side-steps", type=int, default=100, help="timed steps of each short leg of the OTHER BASELINE configurations (cfg5 channels_last, cfg5 NCHW, cfg2) reported under `other_workloads` of a default cfg3 line at one GPU; 0 = none")
    ap.add_argument("--split", type=int, default=1, help="sub-batches run on separate HIP streams inside one hipGraph (cfg3/cfg5)")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, RCCL over xGMI)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


# ---- CPU baseline (+ the parity check of the benchmarked configuration) -------------------------------------------------
def _cpu_image_parallel(jobs, budget_s=90.0, max_procs=None):
    """Image-parallel figure ("one image per process"): P = min(host cores, --cpu-procs) worker processes
    (oracle/cpu_worker.py, one image each, the n distinct images reused round-robin), inputs handed over as memory-mapped
    .npy files, a file barrier, wall clock from the common start to the last finish.  Returns images/sec over all workers;
    `processes` says how many ran (NOT necessarily every core: the field used to be called all_cores)."""
    import shutil
    import tempfile
    n = len(jobs)
    procs = max(1, min(os.cpu_count() or 1, max_procs or 64))
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
        return {"value": round(procs / wall, 4), "unit": "images/sec", "processes": procs, "host_cpus": os.cpu_count(),
                "note": "%d worker processes x 1 image each (the %d sample images round-robin), common start, wall clock to the "
                        "last finish; single-threaded oracle port per process" % (procs, n)}
    finally:
        for p in ws:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(d, ignore_errors=True)


def cpu_baseline(workload, inputs, path, n_images, c4_pooled, max_procs=None, image_parallel=True):
    """Run the CPU checker on the first n_images images of the SAME inputs the timed GPU steps used, (1) time it -- single
    thread: the reference is single-threaded (OpenMP pragma commented out at lib/cppcuda/roi_align_cpu.cpp:136-137; Cython
    loops are serial) -- and (2) compare every intermediate with what the GPU path produced for those images
    (`parity_checked`; a mismatch fails the run).  RoIAlign, ~90 % of the CPU time, runs the reference's own
    roi_align_cpu_loop.cpp compiled unmodified (oracle/_ref/libref_roialign.so, built by `make -C oracle ref`) when that
    library has travelled to this host (`kind: "reference"`); the remaining stages, and everything when it has not
    (`kind: "port"`), run oracle/oracle.c, the plain-C restatement pinned to the reference in the CPU test-suite."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import chain  # noqa: E402  (test infrastructure: allowed in the cpu_baseline leg only)
    import ref_harness  # noqa: E402
    ref_ra = None
    if os.path.exists(os.path.join(ref_harness.REF_BUILD, "libref_roialign.so")):
        try:
            ref_harness.load_ref_roialign()
            ref_ra = ref_harness.ref_roi_align
        except OSError:
            ref_ra = None
    n = min(n_images, path.B)
    host = lambda t: t.float().cpu().numpy()
    T = {}
    refs = []
    if workload == "cfg2":
        rpn_cls, rpn_bbox, feat, cls_score, bbox_pred, sf, im_size = inputs
        t0 = time.perf_counter()
        for b in range(n):
            refs.append(chain.c4_hot_path(host(rpn_cls[b]), host(rpn_bbox[b]), host(feat[b:b + 1]), host(cls_score[b]), host(bbox_pred[b]),
                                          float(sf[b]), host(im_size[b]), path.im_h, path.im_w, pooled=c4_pooled, timings=T, roi_align=ref_ra))
        dt = time.perf_counter() - t0
        for b in range(n):
            chain.compare_c4_with_gpu(path, b, refs[b])
        jobs = None
    else:
        rpn_cls, rpn_bbox, feats, cls_score, bbox_pred, masks, sf, im_size = inputs
        jobs = [([host(c[b]) for c in rpn_cls], [host(d[b]) for d in rpn_bbox], [host(f[b:b + 1]) for f in feats],
                 host(cls_score[b]), host(bbox_pred[b]), host(masks[b * path.max_out:(b + 1) * path.max_out]),
                 float(sf[b]), host(im_size[b]), path.pad_h, path.pad_w) for b in range(n)]
        t0 = time.perf_counter()
        for job in jobs:
            refs.append(chain.fpn_hot_path(*job, top_n=path.top_n, timings=T, roi_align=ref_ra))
        dt = time.perf_counter() - t0
        if path.feat_dtype == torch.float32:
            for b in range(n):
                chain.compare_with_gpu(path, b, refs[b], int(im_size[b, 0]), int(im_size[b, 1]))
        else:   # fp16 pooled features: indices / boxes / detections exact, features to fp16 rounding (rel 1e-3)
            for b in range(n):
                r, k = refs[b], int(path.n_rois[b])
                assert k == r["rois"].shape[0] and np.array_equal(path.rois5[b, :k, 1:].cpu().numpy(), r["rois"])
                assert np.array_equal(path.roi_levels[b, :k].cpu().numpy(), r["roi_levels"])
                got = path.box_feats[b * path.top_n:b * path.top_n + k].float().cpu().numpy()
                assert np.allclose(got, r["box_feats"], rtol=1e-3, atol=1e-3)
                D = min(int(path.det_count[b]), path.max_out)
                assert np.array_equal(path.dets[b, :D].cpu().numpy(), r["dets"][:D])
    conv = sum(T.values())
    out = {"value": round(n / conv, 4), "unit": "images/sec", "cores": 1, "kind": "reference" if ref_ra else "port",
           "sample": "%d images of the same synthetic %s inputs the GPU steps ran, one thread, %.1f s CPU; RoIAlign = %s; "
                     "other stages = oracle/oracle.c (plain-C port); per-stage s/img: %s"
                     % (n, workload, dt, "the reference's roi_align_cpu_loop.cpp compiled unmodified (oracle/_ref)" if ref_ra
                        else "oracle/oracle.c (plain-C port)", {k: round(v / n, 4) for k, v in T.items()}),
           "host_cpus": os.cpu_count(),
           "parity_checked": {"images": n, "ok": True,
                              "what": "every intermediate of the benchmarked configuration (proposals, NMS survivors, rois, "
                                      "level ids, pooled features, detections%s) == CPU checker, bit-exact%s"
                                      % (", mask-branch features, binarised crops" if workload != "cfg2" else "",
                                         "" if workload != "cfg5" else " (fp16 pooled features: rel 1e-3)")}}
    if jobs is not None and image_parallel:
        try:   # informational; the single-core figure above is the contract
            out["image_parallel"] = _cpu_image_parallel(jobs, max_procs=max_procs)
        except Exception as e:
            out["image_parallel"] = {"error": repr(e)}
    return out


def recorded_traffic(wl, batch, channels_last, fp16, k_ms):
    """HBM-side bytes per launch of the box-head RoIAlign of this workload: the committed rocprofv3 --pmc TCC_EA0_* passes
    (profiles/roialign_traffic.json, profiles/README.md), returned only while the hash of the running kernel source matches the one
    the counters were collected on.  -> (traffic bytes | None, source string | None, l1 fill record | None)."""
    traffic, traffic_src, l1_fills = None, None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "roialign_traffic.json")))
        key = "%s_b%d_%s_%s" % (wl, batch, "nhwc" if channels_last else "nchw", "f16" if fp16 else "f32")
        if key in tj:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from kernel_hash import kernel_sha16
            stamped = tj.get(key + "_detail", {}).get("kernel_sha16")
            running = kernel_sha16(wl, channels_last=channels_last)
            if stamped == running and "l1_fill_requests" in tj.get(key + "_detail", {}):
                # the vector L1s' line fills of this launch (TCP -> TCC read requests x 128 B, same counter run)
                nreq = int(tj[key + "_detail"]["l1_fill_requests"])
                l1_fills = {"requests_per_launch": nreq, "bytes_per_launch": nreq * 128,
                            "rate_TBps": round(nreq * 128 / (k_ms * 1e-3) / 1e12, 2),
                            "l2_read_hit_fraction": tj[key + "_detail"].get("l2_read_hit_fraction"),
                            "fill_latency_cycles": tj[key + "_detail"].get("l1_fill_latency_cycles"),
                            # RECORDED microbenchmark results (a pure-load kernel, tools/micro/l1_fill_ceiling.hip), NOT measured in this run
                            "recorded_pure_load_TBps": tj.get("pure_load_ceilings_recorded"),
                            "source": tj[key + "_detail"].get("l1_fill_source")}
            if stamped == running:               # the counters were collected on THIS kernel source
                traffic, traffic_src = tj[key], "profiles/roialign_traffic.json[%s] (rocprofv3 --pmc TCC_EA0_* passes, tools/collect_profiles.sh; not measured in this run; kernel source hash %s matches)" % (key, stamped)
            else:
                traffic_src = "profiles/roialign_traffic.json[%s] is stale: collected on kernel source %s, running %s" % (key, stamped, running)
    except Exception:
        pass
    return traffic, traffic_src, l1_fills


def launch_ms(fn, iters, warm=3):
    """Mean duration of `fn` (one kernel launch on the current stream) over `iters` launches, HIP events on that stream."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def c4_modes(paths, iters, alg):
    """cfg2's RoIAlign launch beyond the exact float32 mode the steps ran in: the contract-legal FAST mode
    (dtc_roi_align_set_exact(0): merged taps, <= 1e-5 from exact, north_star allows 1e-4) and the 16-bit OUTPUT form (bf16 pooled
    features straight into the res5 head) with its own algorithmic bytes, exact and fast."""
    from detectorch_amd import hip
    p0 = paths[0]
    L = hip.lib()
    box = lambda k: (lambda: paths[k % len(paths)]._roi_align_box())
    k = [0]

    def rot():
        paths[k[0] % len(paths)]._roi_align_box()
        k[0] += 1
    p0._roi_align_box()
    torch.cuda.synchronize()
    exact_feats = p0.box_feats.clone()
    out16 = torch.empty(p0.box_feats.shape, dtype=torch.bfloat16, device=p0.dev)

    def to16():
        hip.check(L.dtc_roi_align_forward_packed_ws(p0.feat_lv, 1, p0.C, p0.feat_code, p0.roi_desc.data_ptr(), p0.B * p0.top_n,
                                                    p0.pooled, p0.pooled, p0.sr, out16.data_ptr(), hip.DTC_BF16,
                                                    p0.ra_ws.data_ptr(), p0.ra_ws.numel(), hip.stream_ptr(p0.dev)), "roi_align(c4, bf16 out)")
    alg16 = alg - exact_feats.numel() * exact_feats.element_size() + out16.numel() * 2
    frac = lambda bytes_, ms: round(bytes_ / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    res = {}
    e16 = launch_ms(to16, iters)
    torch.cuda.synchronize()
    ok16 = bool(torch.equal(out16, exact_feats.to(torch.bfloat16)))
    if not ok16:
        raise SystemExit("cfg2: bf16 output != float32 output rounded once")
    hip.roi_align_set_exact(False)
    try:
        f_ms = launch_ms(rot, iters)
        p0._roi_align_box()
        torch.cuda.synchronize()
        dev_fast = float((p0.box_feats.float() - exact_feats.float()).abs().max())
        f16 = launch_ms(to16, iters)
        torch.cuda.synchronize()
        dev16 = float((out16.float() - exact_feats).abs().max())
    finally:
        hip.roi_align_set_exact(True)
        for pth in paths:                 # leave exact results behind
            pth._roi_align_box()
        torch.cuda.synchronize()
    res["fast_mode"] = {"launch_ms": round(f_ms, 4), "frac": frac(alg, f_ms), "max_abs_diff_vs_exact": dev_fast,
                        "what": "dtc_roi_align_set_exact(0): (gh+1)x(gw+1) merged taps with separable weight sums instead of gh x gw x 4 "
                                "(contract: <= 1e-4 on pooled features); the timed steps and the parity check ran in exact mode"}
    res["bf16_output"] = {"algorithmic_bytes_per_launch": int(alg16),
                          "exact": {"launch_ms": round(e16, 4), "frac": frac(alg16, e16), "equals_float32_output_rounded_once": ok16},
                          "fast_mode": {"launch_ms": round(f16, 4), "frac": frac(alg16, f16), "max_abs_diff_vs_exact_float32": dev16},
                          "what": "the same launch writing bf16 pooled features (v_cvt_pk_bf16_f32 in the store path; SURVEY 8f-2: the res5 "
                                  "head's GEMM input type) -- half the output bytes of a launch that is 95 % writes"}
    return res


def cfg5_modes(paths, iters, alg, dev):
    """cfg5 runs in CONTRACT mode (dtc_roi_align_set_exact(0)); this times the same box-head launch in EXACT mode and measures how far
    the two are apart: on the fp16 output (ulps) and on a float32 output of the same 16 000 descriptors (the quantity north_star's
    1e-4 is about).  Outside the tolerance aborts the run.  Leaves contract-mode results behind."""
    from detectorch_amd import hip
    extra = {}
    p0 = paths[0]
    rot_i = [0]

    def rot():
        paths[rot_i[0] % len(paths)]._roi_align_box()
        rot_i[0] += 1
    o32 = torch.empty(p0.box_feats.shape, dtype=torch.float32, device=dev)
    to32 = lambda: hip.check(hip.lib().dtc_roi_align_forward_packed(p0.feat_lv, 4, p0.C, p0.feat_code, p0.roi_desc.data_ptr(), p0.B * p0.top_n,
                                                                  p0.box_p, p0.box_p, p0.sr, o32.data_ptr(), hip.DTC_F32, hip.stream_ptr(dev)), "roi_align f32 out")
    p0._roi_align_box(); to32()
    torch.cuda.synchronize(dev)
    c16, c32 = p0.box_feats.clone(), o32.clone()
    hip.roi_align_set_exact(True)
    try:
        x_ms = launch_ms(rot, iters)
        p0._roi_align_box(); to32()
        torch.cuda.synchronize(dev)
        ulp = (c16.view(torch.int16).int() - p0.box_feats.view(torch.int16).int()).abs()
        extra["exact_mode"] = {"launch_ms": round(x_ms, 4), "frac": round(alg / (x_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "what": "dtc_roi_align_set_exact(1): the reference's separate multiply and add on the up-cast maps, bit-equal to the CPU checker (tests)"}
        extra["contract_vs_exact"] = {"max_abs_diff_float32_output": float((c32 - o32).abs().max()), "tolerance": 1e-4,
                                      "fp16_output_max_ulp": int(ulp.max()), "fp16_output_fraction_differing": float((ulp != 0).float().mean())}
        if extra["contract_vs_exact"]["max_abs_diff_float32_output"] > 1e-4 or extra["contract_vs_exact"]["fp16_output_max_ulp"] > 1:
            raise SystemExit("cfg5 contract mode outside its tolerance")
    finally:
        hip.roi_align_set_exact(False)
        for pth in paths:
            pth._roi_align_box()
        torch.cuda.synchronize(dev)
    return extra


def side_leg(wl, channels_last, a, dev, steps, cpu_images):
    """A SHORT leg of another BASELINE configuration in the same process, after the contract line's timed region (VERDICT r04 item 2:
    cfg5 / cfg2 figures were builder-run only).  Same construction as the headline: two bound input sets, hipGraph replay, two steps in
    flight, `steps` timed steps between synchronisations; then the one-stream step, HIP-event samples of the box-head RoIAlign launch,
    and the parity check of the first `cpu_images` images of the timed tensors against the CPU checker (a mismatch aborts the run)."""
    from detectorch_amd import hip
    from detectorch_amd.pipeline import C4RegionPath, FpnRegionPath, StepPipeline, synthetic_batch, synthetic_c4_batch
    fp16 = wl == "cfg5"
    fdt = torch.float16 if fp16 else torch.float32
    top_n = 2000 if wl == "cfg5" else 1000
    # cfg5 (16-bit maps): the steps run in CONTRACT mode -- dtc_roi_align_set_exact(0), fused convert-multiply-accumulate pooling.  The
    # reference is float-only (roi_align_forward_cuda.cu:199-208): on fp16 maps there are no reference bits, the contract is <= 1e-4 on
    # the float32-accumulated result.  The exact-mode launch (the reference's unfused order on the up-cast maps) is timed beside it.
    contract = wl == "cfg5"
    hip.roi_align_set_exact(not contract)          # read at launch / capture time
    paths, inputs = [], []
    for s in range(2):
        seed = {"cfg3": 3000, "cfg5": 5000, "cfg2": 2000}[wl] + 500 * s
        if wl == "cfg2":
            p = C4RegionPath(a.batch, dev, pooled=a.c4_pooled, feat_dtype=fdt)
            inp = synthetic_c4_batch(a.batch, dev, seed=seed, feat_dtype=fdt)
        else:
            p = FpnRegionPath(a.batch, dev, feat_dtype=fdt, collect_top_n=top_n, max_out=a.max_out)
            inp = synthetic_batch(a.batch, dev, seed=seed, top_n=top_n, feat_dtype=fdt, channels_last=channels_last, max_out=a.max_out)
        p.bind(*inp)
        paths.append(p)
        inputs.append(inp)

    def run(n_inflight, n):
        pipe = StepPipeline(paths, dev, n_inflight=n_inflight)
        for _ in range(max(4, a.warmup)):                      # graph capture + clocks settle, as for the headline
            pipe.step(use_graph=not a.eager)
        pipe.synchronize()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            pipe.step(use_graph=not a.eager)
        pipe.synchronize()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n

    t2 = run(2, steps)
    t1 = run(1, max(20, steps // 2))
    iters = a.kernel_iters
    for _ in range(5):
        for p in paths:
            p._roi_align_box()
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        e0[i].record()
        paths[i % 2]._roi_align_box()
        e1[i].record()
    torch.cuda.synchronize(dev)
    k_all = [e0[i].elapsed_time(e1[i]) for i in range(iters)]
    k_ms = float(np.mean(k_all))
    alg = paths[0].box_roialign_bytes()
    traffic, traffic_src, _ = recorded_traffic(wl, a.batch, channels_last, fp16, k_ms)
    extra = {}
    if contract:
        extra.update(cfg5_modes(paths, iters, alg, dev))
    if wl == "cfg2":
        extra.update(c4_modes(paths, iters, alg))
    paths[0].step(use_graph=not a.eager)
    torch.cuda.synchronize(dev)
    cb = cpu_baseline(wl, inputs[0], paths[0], cpu_images, a.c4_pooled, image_parallel=False)
    hip.roi_align_set_exact(True)
    out = {"workload_id": wl, "feature_layout": "NHWC" if channels_last else "NCHW", "dtype": "f16" if fp16 else "f32",
           "rois_per_image": top_n, "images_per_gpu_per_step": a.batch, "steps": steps,
           "roi_align_mode": "contract (dtc_roi_align_set_exact(0): fused fp32 accumulate on the 16-bit maps)" if contract else "exact",
           "value": round(a.batch / t2, 2), "unit": "images/sec", "ms_per_step": round(t2 * 1e3, 4),
           "one_stream_ms_per_step": round(t1 * 1e3, 4),
           "roofline": {"bound": "hbm", "kernel": "roi_align (box head)", "achieved": round(alg / (k_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "algorithmic_bytes_per_launch": int(alg), "avg_launch_ms": round(k_ms, 4),
                        "launch_ms_min_median_max": [round(float(np.min(k_all)), 4), round(float(np.median(k_all)), 4), round(float(np.max(k_all)), 4)]},
           "parity_checked": cb["parity_checked"], "cpu_baseline_images_per_sec": cb["value"]}
    out["roofline"].update(extra)
    del paths, inputs
    torch.cuda.empty_cache()
    return out


def harder_set_launch(path, feats, top_n, dev, iters):
    """The box-head RoIAlign launch on a HARDER RoI population than the bench's own (RPN-on-noise proposals: 94 % on P2, median
    bin 1.3 px): the generator of tools/bench_roialign.py --sort -- log-uniform sides 16-600 px, FPN level by area, visited in
    (image, level, row band, x) order -- on the feature maps of the timed steps.  Same kernel, same launch shape."""
    from detectorch_amd import hip, synth
    rois, lvn, order_np = synth.harder_roi_set(path.B, top_n)
    order = torch.from_numpy(order_np).to(dev)
    lv, rois_t = torch.from_numpy(lvn).to(dev), torch.from_numpy(rois).to(dev)
    out = torch.empty((rois.shape[0], feats[0].shape[1], path.box_p, path.box_p), dtype=path.box_feats.dtype, device=dev)
    run = lambda: hip.roi_align_forward(feats, synth.FPN_ROI_SCALES, rois_t, path.box_p, path.box_p, 2, roi_levels=lv, out=out, roi_order=order)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / iters


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    rank = int(os.environ.get("RANK", "0"))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world_env != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world_env))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    world = 1
    if world_env > 1 or (a.gather_always and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world_env, device_id=dev)
        world = dist.get_world_size()
        assert world == a.gpus, (world, a.gpus)

    from detectorch_amd import hip
    from detectorch_amd.pipeline import (C4RegionPath, FpnRegionPath, OverlappedRegionPath, StepPipeline, synthetic_batch,
                                         synthetic_c4_batch)
    hip.lib()   # fails loudly if the native library is missing
    wl = a.workload
    # cfg5 ("fp16 feature maps"): the layout a 16-bit backbone emits on MI355X is channels_last (MIOpen's preferred layout for 16-bit
    # convolutions; SURVEY 7 hard-part 3 allows either) -- 15 % more images/s than NCHW fp16 maps; `--nchw` gives the other line
    if wl == "cfg5" and not a.nchw:
        a.channels_last = True
    fp16 = a.fp16 or wl == "cfg5"
    fdt = torch.float16 if fp16 else torch.float32
    top_n = 2000 if wl == "cfg5" else 1000
    NSETS = max(2, a.inflight)
    contract = wl == "cfg5" and fp16       # 16-bit maps: contract mode (see side_leg); read at launch / graph-capture time
    hip.roi_align_set_exact(not contract)
    paths, inputs = [], []
    for s in range(NSETS):
        seed = {"cfg3": 3000, "cfg5": 5000, "cfg2": 2000}[wl] + 500 * s + rank
        if wl == "cfg2":
            p = C4RegionPath(a.batch, dev, pooled=a.c4_pooled, feat_dtype=fdt)
            inp = synthetic_c4_batch(a.batch, dev, seed=seed, feat_dtype=fdt)
        else:
            kw = dict(feat_dtype=fdt, collect_top_n=top_n, max_out=a.max_out)
            if a.split > 1 and a.batch % a.split == 0:
                p = OverlappedRegionPath(a.batch, dev, n_split=a.split, **kw)
            else:
                p = FpnRegionPath(a.batch, dev, **kw)
            inp = synthetic_batch(a.batch, dev, seed=seed, top_n=top_n, feat_dtype=fdt, channels_last=a.channels_last, max_out=a.max_out)
        p.bind(*inp)
        paths.append(p)
        inputs.append(inp)
    gather = None
    if world > 1 or (a.gather_always and dist is not None):
        from detectorch_amd.dist import DetectionGatherer
        gather = DetectionGatherer(paths[0].B, paths[0].max_out, dev, world)

    pipe = StepPipeline(paths, dev, n_inflight=a.inflight)

    def one_step():
        p, st = pipe.step(use_graph=not a.eager)                  # step k+1 is issued while step k runs (StepPipeline)
        if gather is not None:
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                gather.all_gather_async(p.dets, p.det_count)      # one packed collective per step, overlapped with the next

    def timed(n_steps):
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n_steps):
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
        return dt

    for _ in range(NSETS):                   # eager pass + hipGraph capture of every path BEFORE the first collective is
        pipe.step(use_graph=not a.eager)     # issued: no stream capture ever overlaps RCCL work in flight
    pipe.synchronize()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    pipe.count = 0
    for _ in range(max(a.warmup, NSETS)):
        one_step()
    pipe.synchronize()
    if gather is not None:
        gather.finish()
    pipe.count = 0
    dt = timed(a.steps)                      # the contract: exactly K steps, barrier + synchronize on both sides, max over ranks
    gathered_ok, gathered_copy = None, None
    s_last = (a.steps - 1) % NSETS
    if gather is not None:                   # what every rank received for THIS rank's last step == what the path holds
        gd, gc = gather.finish()
        pl = paths[s_last]
        gathered_ok = bool(torch.equal(gd[rank], pl.dets) and torch.equal(gc[rank], pl.det_count))
        if not gathered_ok:
            raise SystemExit("all-gathered detections differ from the local result")
        if rank == 0:                        # kept for the cross-rank check below (the staging buffers are reused by later steps)
            gathered_copy = (gd.clone(), gc.clone())
    # a longer run of the same loop (>= --sustain-seconds): the contract region is only K steps long
    n_sus = int(max(a.steps, np.ceil(a.sustain_seconds / max(dt / a.steps, 1e-6)))) if a.sustain_seconds > 0 else 0
    if dist is not None and n_sus:
        tn = torch.tensor([n_sus], dtype=torch.int64, device=dev)
        dist.all_reduce(tn, op=dist.ReduceOp.MAX)
        n_sus = int(tn.item())
    dt_sus = timed(n_sus) if n_sus else None

    # ---- roofline of the dominant kernel: the box-head RoIAlign launch, HIP events on the launch stream -------------------
    iters = a.kernel_iters
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for _ in range(5):                       # untimed: clocks / caches settle after the step loop above
        for p in paths:                      # every path holds the descriptors of its own inputs from its last step
            p._roi_align_box()
    # NO synchronisation here: the timed launches queue up behind the warm-up ones, so the first of them does not start on a GPU
    # that has just gone idle (a launch that follows a device synchronisation was seen to take 1.2-1.4 ms instead of 0.37 in
    # three of five runs: one such sample moves the mean of 20 by 14 %; min / median / max and every sample are reported too)
    for i in range(iters):
        p = paths[i % NSETS]
        e0[i].record()
        p._roi_align_box()
        e1[i].record()
    torch.cuda.synchronize(dev)
    k_all = [e0[i].elapsed_time(e1[i]) for i in range(iters)]
    k_ms = float(np.mean(k_all))
    alg_bytes = paths[0].box_roialign_bytes()
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    c4_extra = {} if a.no_modes else c4_modes(paths, iters, alg_bytes) if wl == "cfg2" else cfg5_modes(paths, iters, alg_bytes, dev) if contract else {}
    harder = None
    if wl != "cfg2" and not a.no_modes:       # the same launch on the harder RoI population (a trained RPN looks like it);
                                              # not in counter / kernel-trace runs: they average per grid, and this leg launches the same grid
        h_ms = harder_set_launch(paths[0], inputs[0][2], top_n, dev, max(5, iters // 2))
        harder = {"launch_ms": round(h_ms, 4), "frac": round(alg_bytes / (h_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                  "rois": "log-uniform sides 16-600 px, FPN level by area, sorted (image, level, 32-row band, x): tools/bench_roialign.py --sort"}
    # ---- the step on ONE stream (no second step in flight): what a latency-sensitive caller sees ------------------------------
    one_stream_ms = None
    if a.inflight > 1 and a.split == 1:
        pipe1 = StepPipeline(paths, dev, n_inflight=1)
        n1 = max(20, min(a.steps, 200))
        for _ in range(NSETS):
            pipe1.step(use_graph=not a.eager)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(n1):
            pipe1.step(use_graph=not a.eager)
        torch.cuda.synchronize(dev)
        one_stream_ms = (time.perf_counter() - t1) / n1 * 1e3

    # ---- N > 1: the rows rank 0 RECEIVED from every other rank == what those ranks' inputs give when recomputed here ---------
    # ("verify gathered detections are bit-identical to the W=1 run".  Rank r's input set s is the deterministic
    # synthetic batch of seed base + 500 s + r, so rank 0 regenerates it, runs the path eagerly on its own GPU and compares
    # with the gathered rows of rank r's last timed step.  No collective is involved; the other ranks wait at the barrier.)
    recomputed_ok = None
    if gathered_copy is not None and world > 1:
        gd0, gc0 = gathered_copy
        base = {"cfg3": 3000, "cfg5": 5000, "cfg2": 2000}[wl]
        pv = paths[s_last]
        recomputed_ok = True
        for r in range(1, world):
            seed_r = base + 500 * s_last + r
            inp_r = (synthetic_c4_batch(a.batch, dev, seed=seed_r, feat_dtype=fdt) if wl == "cfg2" else
                     synthetic_batch(a.batch, dev, seed=seed_r, top_n=top_n, feat_dtype=fdt, channels_last=a.channels_last, max_out=a.max_out))
            pv.bind(*inp_r)
            pv.step(use_graph=False)         # eager: the captured graph holds the pointers of the original input set
            torch.cuda.synchronize(dev)
            ok_r = bool(torch.equal(gd0[r], pv.dets) and torch.equal(gc0[r], pv.det_count))
            recomputed_ok = recomputed_ok and ok_r
            del inp_r
        pv.bind(*inputs[s_last])             # restore (the roofline launches and the CPU check below use the original inputs)
        pv.step(use_graph=False)
        torch.cuda.synchronize(dev)
        if not recomputed_ok:
            raise SystemExit("detections gathered from another rank differ from their recomputation on rank 0")

    traffic, traffic_src, l1_fills = recorded_traffic(wl, a.batch, a.channels_last, fp16, k_ms)

    if rank == 0:
        n_img = a.batch * a.steps * world
        p0 = paths[0]
        if wl == "cfg2":
            desc = ("BASELINE configs[1]: Faster R-CNN R-50-C4, 1x3x800x1333, RPN 63000 anchors -> 6000 -> NMS 0.7 -> 1000 proposals, "
                    "RoIAlign %dx%d sampling_ratio 0 on res4 [B,1024,50,84], 81-class postprocess" % (a.c4_pooled, a.c4_pooled))
            kern = "roi_align (res4, adaptive sampling, %d rois x 1024 ch)" % (a.batch * 1000)
            not_in = "ResNet-50 convs, RPN head, res5 head GEMMs/convs (MIOpen/hipBLASLt), outputs synthetic"
        else:
            desc = ("BASELINE configs[%d]: Mask R-CNN R-50-FPN, 1x3x800x1333 (padded 800x1344), 5-level RPN (268569 anchors) -> %d rois, "
                    "4-level RoIAlign 7x7 sr2 C256%s, 81-class postprocess, RoIAlign 14x14 mask branch, 28x28 mask paste"
                    % (4 if wl == "cfg5" else 2, top_n, " fp16 features" if fp16 else ""))
            if wl == "cfg3" and world * a.batch == 64 and world == 8:
                # 8 GPUs x 8 images = BASELINE configs[3] (Mask R-CNN R-101-FPN, batch 64 over 8 MI355X): the hot-path shapes are those of
                # configs[2] -- R-101 only deepens the backbone, which is not in the path
                desc = desc.replace("BASELINE configs[2]: Mask R-CNN R-50-FPN", "BASELINE configs[3]: Mask R-CNN R-101-FPN, batch 64 sharded over 8 GPUs "
                                    "(hot-path shapes == configs[2]; the deeper backbone is not in the path)")
            kern = "roi_align (box head, 4 levels, %d rois)" % (a.batch * top_n)
            not_in = "ResNet-50/FPN convs and box/mask-head GEMMs (MIOpen/hipBLASLt), outputs synthetic"
        out = {
            "metric": "images/sec Mask R-CNN R-50-FPN 1333x800 (region-proposal hot path)" if wl != "cfg2" else
                      "images/sec Faster R-CNN R-50-C4 1333x800 (region-proposal hot path)",
            "value": round(n_img / dt, 2), "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if fp16 else "f32", "data": "synthetic",
            "config": {"workload": desc, "workload_id": wl,
                       "images_per_gpu_per_step": a.batch, "global_batch": a.batch * world, "rois_per_image": top_n,
                       "detection_rows_per_image": None if wl == "cfg2" else a.max_out,
                       "input_sets_rotated": NSETS,
                       "feature_layout": "NHWC" if a.channels_last else "NCHW",
                       "roi_align_mode": "contract (dtc_roi_align_set_exact(0): fused fp32 accumulate on the 16-bit maps)" if contract else "exact",
                       "launch": ("eager" if a.eager else "hipGraph") + (", %d sub-batches on %d streams" % (a.split, a.split) if isinstance(p0, OverlappedRegionPath) else "") +
                                 (", %d steps in flight on %d HIP streams (StepPipeline)" % (a.inflight, a.inflight) if a.inflight > 1 else ""),
                       "steps_in_flight": a.inflight,
                       "parallelism": "images sharded over %d GPU(s); all_gather of detections" % world,
                       "not_in_path": not_in},
            "roofline": {"bound": "hbm", "kernel": kern,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(k_ms, 4),
                         "launch_ms_min_median_max": [round(float(np.min(k_all)), 4), round(float(np.median(k_all)), 4), round(float(np.max(k_all)), 4)],
                         "launch_ms_samples": [round(float(v), 4) for v in k_all],
                         "l1_fills": l1_fills,
                         "note": ("HBM traffic of this launch == its algorithmic bytes (every feature byte is staged once: map-stationary "
                                  "kernel); what bounds it is the adaptive-grid gather from LDS -- ~5 samples x 4 taps per bin and channel, "
                                  "formed with the reference's unfused multiply-adds -- i.e. LDS reads and VALU issue, not HBM (DESIGN 3.6)") if wl == "cfg2" else
                                 ("direct-gather kernel (roi_align_nhwc16.hip, DESIGN 3.1): every tap is a 16-byte load per lane straight from L1 / L2, "
                                  "fabric traffic ~ the compulsory bytes; in contract mode (16-bit maps) the taps of a bin that share a pixel are "
                                  "requested once and pooled with one fused multiply-accumulate per element: recorded counters of the 16 000-RoI "
                                  "launch (profiles/r06_z_cfg5_counters.json, not measured in this run): vector loads 6.8 M -> 4.2 M instructions, "
                                  "VALU 66 -> 59 % of SIMD time, waves parked 46 -> 58 % -- a latency mix (3.4 waves per SIMD wait for ~9 loads, then "
                                  "compute), no saturated unit; exact mode: texture data path + the reference's unfused multiply-adds") if a.channels_last else
                                 "memory side of THIS formulation, not a CU pipe (DESIGN 3.1, tools/r06/README.md 1, profiles/r06_a_boxhead_three_way_replay.txt): "
                                 "the launch replayed with only its staging loads left in (same workgroups, clusters, passes, addresses) takes 62 % of the "
                                 "launch -- 2.67 GB of 128-byte L1 fills of which 1.22 GB come from the fabric at 5.3 TB/s; stores only 30 %, pooling from "
                                 "LDS only 44 %; the shipped launch puts 1.63 GB (1.44 x algorithmic) on the fabric = 0.31 ms at that rate and runs at "
                                 "1.1-1.2 x that; more loads in flight and an image-major order (Infinity-Cache-resident maps) were measured worse: what "
                                 "is left is the 0.49 GB of fabric re-reads by neighbouring workgroups.  LDS array 55 % busy, VALU 43 % (recorded counters)"},
            "consistency": {"timed_region_s": round(dt, 4), "one_stream_ms_per_step": None if one_stream_ms is None else round(one_stream_ms, 4),
                            "gathered_equals_local": gathered_ok,
                            "gathered_equals_recomputed": recomputed_ok,
                            "sustained": None if dt_sus is None else {"steps": n_sus, "seconds": round(dt_sus, 3),
                                                                      "ms_per_step": round(dt_sus / n_sus * 1e3, 4),
                                                                      "images_per_sec": round(a.batch * n_sus * world / dt_sus, 2)}},
        }
        out["roofline"]["harder_set"] = harder
        # first-class beside `frac`: the same launch on the RoI population a trained RPN produces
        out["roofline"]["frac_harder"] = None if harder is None else harder["frac"]
        out["roofline"].update(c4_extra)
        if not a.no_cpu_baseline and not isinstance(p0, OverlappedRegionPath):
            p0.step(use_graph=not a.eager)          # the configuration that was timed, on input set 0
            torch.cuda.synchronize(dev)
            # rank 0 only; at N > 1 a shorter sample (the other ranks wait at the final barrier meanwhile)
            out["cpu_baseline"] = cpu_baseline(wl, inputs[0], p0, a.cpu_images if world == 1 else min(a.cpu_images, 2), a.c4_pooled,
                                               max_procs=a.cpu_procs)
        if (world == 1 and a.side_steps > 0 and wl == "cfg3" and not a.channels_last and not a.fp16 and not a.no_cpu_baseline
                and not isinstance(p0, OverlappedRegionPath)):
            # BASELINE configs[4] (both layouts) and configs[1], short legs, AFTER everything the headline reports was measured
            del pipe
            paths.clear(); inputs.clear()
            torch.cuda.empty_cache()
            out["other_workloads"] = {}
            for name, (wl_s, cl_s) in (("cfg5_nhwc", ("cfg5", True)), ("cfg5_nchw", ("cfg5", False)), ("cfg2", ("cfg2", False))):
                try:      # a failing side leg (OOM, a parity abort) must not discard the headline line measured above
                    out["other_workloads"][name] = side_leg(wl_s, cl_s, a, dev, a.side_steps, 2)
                except BaseException as e:   # SystemExit of a parity abort included: recorded, not swallowed silently
                    out["other_workloads"][name] = {"error": repr(e)}
                    hip.roi_align_set_exact(True)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
