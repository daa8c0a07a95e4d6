"""Lab: phase timeline of k_node_update's workgroups (needs a -DVSN_LAB_STAMPS build of layer_fwd.hip, VSN_LIB=...).
Stamps are the 100 MHz real-time counter (10 ns ticks), one row per wave: 0 entry, 1 rowptr there, 2/3/4 after the
wave's 1st/2nd/3rd edge, 5 edge loop done, 6 reduce-scatter done, 7 own component updated, 8/9 the two barriers of the
second exchange, 10 x stored, 11 LayerNorm stored (wave 0)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ai2bmd_amd import capi  # noqa: E402
from ai2bmd_amd.synthetic import default_hparams, make_state_dict  # noqa: E402
from ai2bmd_amd.visnet_calculator import ViSNetEngine  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "visnet_prot_chig.npz"))
hp = default_hparams()
eng = ViSNetEngine(hp, make_state_dict(hp, seed=2024), "cuda:0")
z = torch.as_tensor(g["z"], dtype=torch.int64).cuda()
p = torch.as_tensor(g["pos_relaxed"], dtype=torch.float32).cuda()
e = torch.empty(len(g["start"]), device="cuda")
f = torch.empty(len(g["z"]), 3, device="cuda")
for _ in range(5):
    eng.forces_device(z, p, g["start"], g["end"], e, f)
torch.cuda.synchronize()
N = len(g["z"])
buf = torch.zeros(N * 8 * 16, dtype=torch.int64, device="cuda")
L = capi.lib()
L.vsn_lab_set_stamps.argtypes = [C.c_void_p]
assert L.vsn_lab_set_stamps(C.c_void_p(buf.data_ptr())) == 0
eng.forces_device(z, p, g["start"], g["end"], e, f)
torch.cuda.synchronize()
L.vsn_lab_set_stamps(C.c_void_p(0))
s = buf.cpu().numpy().reshape(N, 8, 16).astype(np.float64) * 0.01  # microseconds
t0 = s[:, :, 0].min()
ent = s[:, :, 0] - t0
print(f"workgroup entry (first wave) spread: min {ent.min(1).min():.2f} max {ent.min(1).max():.2f} us; "
      f"last stamp of the launch at {np.nanmax(np.where(s > 0, s, np.nan)) - t0:.2f} us")
names = {1: "rowptr", 2: "edge 1", 3: "edge 2", 4: "edge 3", 5: "loop done", 6: "reduce-scatter", 7: "component",
         8: "barrier A", 9: "barrier B", 10: "x stored", 11: "LN stored"}
for k in range(1, 12):
    v = s[:, :, k]
    m = v > 0
    if m.any():
        d = (v - s[:, :, 0])[m]
        print(f"  {k:2d} {names[k]:15s} since wave entry: mean {d.mean():6.2f}  p10 {np.percentile(d, 10):6.2f}  "
              f"p90 {np.percentile(d, 90):6.2f} us   ({m.sum()} waves)")
life = np.where(s > 0, s, 0).max(2) - s[:, :, 0]
print(f"wave life: mean {life.mean():.2f} us, workgroup life (max over waves) mean {life.max(1).mean():.2f} us")
