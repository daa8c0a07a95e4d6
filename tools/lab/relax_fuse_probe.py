"""lab: vsn_md_half1_build + vsn_hopt_run against vsn_md_half1_build_relax on identical state (which array differs?)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import ctypes as C
import numpy as np, torch
from ai2bmd_amd import capi
from ai2bmd_amd.amber import default_tables
from ai2bmd_amd.bonded import ShardedFragmentForces
from ai2bmd_amd.fragmentation import ProteinAtoms, build_plan
from ai2bmd_amd.hydrogen import build_hydrogen_plan
from ai2bmd_amd.md import LangevinHIP
from ai2bmd_amd.synthetic import default_hparams, make_state_dict
from ai2bmd_amd.visnet_calculator import ViSNetEngine

d = np.load("tests/golden/protein_chig.npz")
prot = ProteinAtoms(d["names"], d["resnames"], d["resnums"], d["numbers"], d["positions"].astype(np.float64))
plan = build_plan(prot)
hplan = build_hydrogen_plan(prot, plan, default_tables())
hp = default_hparams(embedding_dimension=64, num_layers=2)
eng = ViSNetEngine(hp, make_state_dict(hp, seed=1), "cuda:0")
L = capi.lib()
out = {}
for mode in (0, 1):
    ff = ShardedFragmentForces.for_engine(eng, plan, hydrogen=hplan)
    md = LangevinHIP(prot.numbers, prot.positions, ff.step, "cuda:0", seed=11, tether_k=2.0)
    fp, cp, frag_pos, F_prot, E_tot = ff.fused_tail
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (md._h, C.c_void_p(md.x.data_ptr()), C.c_void_p(md.v.data_ptr()), C.c_void_p(md.F.data_ptr()), fp,
            C.c_void_p(frag_pos.data_ptr()))
    if mode == 0:
        assert L.vsn_md_half1_build(*args, st) == 0
        torch.cuda.synchronize()
        placed = frag_pos.clone()
        ff.relaxer.run(frag_pos, torch.cuda.current_stream())
    else:
        assert L.vsn_md_half1_build_relax(*args, ff.relaxer._h, st) == 0
        placed = None
    torch.cuda.synchronize()
    out[mode] = (md.x.clone(), md.v.clone(), frag_pos.clone(), ff.relaxer.stats())
for k, name in enumerate(("x", "v", "frag_pos")):
    a, b = out[0][k], out[1][k]
    print(name, "equal" if torch.equal(a, b) else f"DIFFER max {float((a - b).abs().max()):.3e} n={int((a != b).sum())}")
print(out[0][3], out[1][3])
