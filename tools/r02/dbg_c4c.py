import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detectorch_amd.pipeline import C4RegionPath, synthetic_c4_batch
dev = torch.device("cuda", 0)
B = 8
paths = []
for s in range(2):
    p = C4RegionPath(B, dev, pooled=7)
    p.bind(*synthetic_c4_batch(B, dev, seed=2000 + 500 * s)); paths.append(p)
for p in paths:
    p.step(use_graph=True); p.step(use_graph=True)
torch.cuda.synchronize()
ref = [(p.pre_boxes.clone(), p.pre_scores.clone(), p.keep.clone(), p.keep_cnt.clone(), p.rois5.clone(), p.dets.clone()) for p in paths]
bad = {"pre_boxes": 0, "pre_scores": 0, "keep": 0, "keep_cnt": 0, "rois5": 0, "dets": 0}
mode = sys.argv[1] if len(sys.argv) > 1 else "sync"
N = 3000
for i in range(N):
    p = paths[i % 2]
    p.step(use_graph=True)
    if mode == "sync" or i % 50 == 49:
        torch.cuda.synchronize()
        r = ref[i % 2]
        for k, (name, t) in enumerate(zip(bad, (p.pre_boxes, p.pre_scores, p.keep, p.keep_cnt, p.rois5, p.dets))):
            if not torch.equal(t, r[k]):
                bad[name] += 1
                if bad[name] <= 3:
                    print("step", i, name, "differs; keep_cnt", p.keep_cnt.tolist(), "pre", p.pre_counts.tolist())
print(mode, "anomalies over", N, "steps:", bad)
