import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch
dev = torch.device("cuda", 0)
path = FpnRegionPath(8, dev)
path.bind(*synthetic_batch(8, dev, seed=3000))
path.step(use_graph=False)
torch.cuda.synchronize()
os.makedirs("gpurun_out/r02_desc", exist_ok=True)
np.save("gpurun_out/r02_desc/roi_desc_band%s.npy" % os.environ.get("DTC_FPN_BAND_LOG2", "5"), path.roi_desc.cpu().numpy())
np.save("gpurun_out/r02_desc/m_desc.npy", path.m_desc.cpu().numpy())
print("saved")
