"""Lab: stage timeline of hk_fused's workgroups (needs a -DHF_LAB_STAMPS build of head_fused.hip:
python tools/build_variant.py hfst head_fused.hip -DHF_LAB_STAMPS ; VSN_LIB=ai2bmd_amd/_ab/libvsn_hfst.so).
Stamps are the 100 MHz real-time counter (10 ns ticks), thread 0 of every workgroup, one after every barrier."""
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
WG = (N + 7) // 8
buf = torch.zeros(WG * 32, dtype=torch.int64, device="cuda")
L = capi.lib()
L.vsn_lab_set_hf_stamps.argtypes = [C.c_void_p]
assert L.vsn_lab_set_hf_stamps(C.c_void_p(buf.data_ptr())) == 0
eng.forces_device(z, p, g["start"], g["end"], e, f)
torch.cuda.synchronize()
L.vsn_lab_set_hf_stamps(C.c_void_p(0))
s = buf.cpu().numpy().reshape(WG, 32).astype(np.float64) * 0.01  # microseconds
names = ["entry", "S1 load cat0 + norms", "S2 a0 = cat0.Wa0 (512x256)", "act", "S3 u0 = ta.Wb0 (256x256)", "S4 gate",
         "S5 p1 = vec1o.W11 (64 rows 128x128)", "S6 norms", "S7 a1 = cat1.Wa1 (256x128)", "S8 energy", "S10 g_cat1 (128x256)",
         "S11 scale", "S12 g_vec1o (64 rows 128x128)", "S13 gate adjoint + g_pv0 store", "S14 g_h0 (256x256)", "dact",
         "S15 g_cat0 (256x512)", "S16 stores"]
t0 = s[:, 0].min()
print(f"{WG} workgroups; entry spread {s[:, 0].min() - t0:.2f}..{s[:, 0].max() - t0:.2f} us; last exit {s[:, 17].max() - t0:.2f} us")
for k in range(1, 18):
    d = s[:, k] - s[:, k - 1]
    print(f"  {k:2d} {names[k]:42s} mean {d.mean():6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us")
print(f"workgroup life mean {(s[:, 17] - s[:, 0]).mean():.2f} us")
