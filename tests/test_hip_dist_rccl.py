"""The RCCL leg of the multi-GPU path on the one GPU a test box has: a world-size-1 "nccl" (= RCCL on ROCm) process group
running the same DetectionGatherer calls bench.py makes at N > 1 (blocking two-collective form and the overlapped packed
form), so that the collective API usage is exercised on the real backend; the N = 2 semantics are covered on CPU by
tests/test_dist_gloo.py.  -m gpu."""
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_detection_gatherer_over_rccl_world1():
    from detectorch_amd.dist import DetectionGatherer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        g = DetectionGatherer(4, 16, dev, 1)
        gen = torch.Generator(device="cuda"); gen.manual_seed(3)
        side = torch.cuda.Stream()
        for step in range(5):
            dets = torch.rand((4, 16, 6), generator=gen, device=dev)
            cnt = torch.randint(0, 17, (4,), generator=gen, device=dev, dtype=torch.int32)
            a, c = g.all_gather(dets, cnt)
            assert torch.equal(a[0], dets) and torch.equal(c[0], cnt)
            g.all_gather_async(dets, cnt)
            dets2 = dets.clone()
            dets.zero_()                                   # the next step overwrites the source: staging must have its copy
        a, c = g.finish()
        torch.cuda.synchronize()
        assert torch.equal(a[0], dets2) and torch.equal(c[0], cnt)
    finally:
        dist.destroy_process_group()


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` launches two ranks itself and reports n_gpus 2; every rank's gathered detections equal what
    the owning rank computed.  Needs two visible GPUs (skipped on the one-GPU test box)."""
    import json
    import os
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--sustain-seconds", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2 * out["config"]["images_per_gpu_per_step"]
    # the self-verification of the N > 1 line: what rank 0 gathered from rank 1 == rank 1's inputs recomputed on rank 0
    assert out["consistency"]["gathered_equals_local"] is True and out["consistency"]["gathered_equals_recomputed"] is True


def test_bench_refuses_world_size_mismatch():
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_bench_gather_path_with_steps_in_flight_world1():
    """The N > 1 bench loop on one GPU: world-size-1 RCCL group, the per-step packed all-gather issued from the two streams
    the steps in flight run on; every gathered row must equal what the path produced (bench.py asserts that itself)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "4",
                        "--sustain-seconds", "0", "--no-cpu-baseline", "--gather-always", "--inflight", "2"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["steps_in_flight"] == 2 and out["consistency"]["gathered_equals_local"] is True
