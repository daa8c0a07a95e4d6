"""lab: the stand-alone GEMM tap (vsn_gemm) called again and again on the same operands - do the results change from
call to call when a second process does the same on the same GPU?   python tools/lab/gemm_determinism.py [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import ctypes as C
import torch
from ai2bmd_amd import capi
from ai2bmd_amd.synthetic import default_hparams, make_state_dict
from ai2bmd_amd.visnet_calculator import ViSNetEngine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
hp = default_hparams()
eng = ViSNetEngine(hp, make_state_dict(hp, seed=2024), "cuda:0")
L = capi.lib()
g = torch.Generator(device="cuda").manual_seed(3)
shapes = ((1_300_000, 768, 256), (81_000, 1280, 256), (1_300_000, 256, 768))
if os.environ.get("GD_ONLY"):
    shapes = shapes[:1]
for (M, Nc, K) in shapes:
    A = torch.randn(M, K, device="cuda", generator=g)
    B = torch.randn(Nc, K, device="cuda", generator=g)
    if os.environ.get("GD_ZERO"):  # (lab: an aggressor that moves nothing but zeros)
        A.zero_(); B.zero_()
    for mode in ((1,) if os.environ.get("GD_ONLY") else (0, 1)):
        eng.set_option("gemm_split3", mode)
        outs = None
        bad = 0
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for r in range(reps):
            Cm = torch.empty(M, Nc, device="cuda")
            rc = L.vsn_gemm(eng._h, C.c_void_p(A.data_ptr()), K, C.c_void_p(B.data_ptr()), K, C.c_void_p(Cm.data_ptr()), Nc,
                            None, M, Nc, K, 0, st)
            assert rc == 0, rc
            torch.cuda.synchronize()
            if outs is None:
                outs = Cm
            elif not torch.equal(outs, Cm):
                bad += 1
                if bad == 1:
                    d = (outs - Cm).abs()
                    first = (int((d > 0).sum()), float(d.max()), (d > 0).nonzero()[:3].tolist())
        print(f"pid {os.getpid()} M={M} Nc={Nc} K={K} split3={mode}: {bad} of {reps - 1} repeats differ" +
              (f" first: {first}" if bad else ""), flush=True)
        del outs
