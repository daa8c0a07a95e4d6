"""GPU debugging aid (not a pytest file): runs one golden case through the HIP
library with debug taps on and prints, tap by tap, the error against the fp64
oracle's intermediates (oracle/visnet_oracle.py::energy_forces_analytic).

    python tests/stage_report.py h64_l2
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import load_golden  # noqa: E402
from oracle.visnet_oracle import ViSNetOracle, dcosine_cutoff  # noqa: E402
from oracle.weights import make_state_dict  # noqa: E402


def main(name):
    from ai2bmd_amd.visnet_calculator import ViSNetEngine

    g = load_golden(name)
    hp = g["hparams"]
    sd = make_state_dict(hp, seed=g["weight_seed"])
    o = ViSNetOracle(hp, sd, torch.float64)
    E, F, c, b = o.energy_forces_analytic(g["z"], g["pos"], g["start"], g["end"])
    eng = ViSNetEngine(hp, sd, "cuda:0")
    eng.set_option("debug", 1)
    dev = torch.device("cuda:0")
    z = torch.as_tensor(g["z"]).to(dev)
    pos = torch.as_tensor(g["pos"]).to(dev)
    B, N = len(g["start"]), len(g["z"])
    e_out = torch.zeros(B, device=dev)
    f_out = torch.zeros(N, 3, device=dev)
    eng.forces_device(z, pos, g["start"], g["end"], e_out, f_out)
    torch.cuda.synchronize()
    H, S, L, R = o.H, o.S, o.L, o.R
    Rp = (R + 31) // 32 * 32
    Eg = eng.last_num_edges()
    gr = c["graph"]
    print(f"case {name}: N={N} B={B} E_oracle={len(gr['src'])} E_gpu={Eg}")
    bad = []

    def cmp(tag, got, ref, scale=None):
        got = np.asarray(got, dtype=np.float64).ravel()
        ref = np.asarray(ref, dtype=np.float64).ravel()
        if got.shape != ref.shape:
            print(f"  {tag:28s} SHAPE {got.shape} vs {ref.shape}")
            bad.append(tag)
            return
        err = np.abs(got - ref).max() if ref.size else 0.0
        mag = max(np.abs(ref).max() if ref.size else 0.0, 1e-30)
        flag = "" if err <= 2e-4 * max(mag, 1e-3) and np.isfinite(got).all() else "   <<<<<<"
        if flag:
            bad.append(tag)
        print(f"  {tag:28s} max|err|={err:10.3e}  max|ref|={mag:10.3e}  rel={err / mag:9.2e}{flag}")

    def t(x):
        return x.detach().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)

    rd = eng.debug_read
    for nm in ("rowptr", "colptr", "src", "tgt", "perm"):
        got = rd(nm, 0, dtype=np.int32)
        ref = gr[nm]
        ok = got.shape == ref.shape and (got == ref).all()
        print(f"  {nm:28s} {'exact' if ok else 'MISMATCH   <<<<<<'}")
        if not ok:
            bad.append(nm)
    if bad:
        print("graph differs; stopping")
        return 1
    E_ = Eg
    geo = rd("geo").reshape(E_, 8)
    cmp("geo.r", geo[:, 0], t(c["r"]))
    cmp("geo.C", geo[:, 1], t(c["C"]))
    cmp("geo.dC", geo[:, 2], t(dcosine_cutoff(c["r"], o.rc)))
    cmp("geo.u", geo[:, 3:6], t(c["u"]))
    cmp("d", rd("d").reshape(E_, 8)[:, :S], t(c["d"]))
    cmp("rbf", rd("rbf").reshape(E_, Rp)[:, :R], t(c["rbf"]))
    pp = rd("pp").reshape(E_, 2 * H)
    cmp("pp.phi", pp[:, :H], t(c["phi"]))
    cmp("pp.psi", pp[:, H:], t(c["psi"]))
    cat = rd("cat").reshape(N, 2 * H)
    cmp("cat.x0", cat[:, :H], t(c["x0"]))
    cmp("cat.nagg", cat[:, H:], t(c["nagg"]))
    cmp("x_emb", rd("x_emb"), t(c["x_emb"]))
    for l in range(L):
        lc = c["layers"][l]
        last = l == L - 1
        cmp(f"L{l}.x_in", rd("x_in", l), t(lc["x_in"]))
        cmp(f"L{l}.vec_in", rd("vec_in", l), t(lc["vec_in"]))
        cmp(f"L{l}.f_in", rd("f_in", l), t(lc["f_in"]))
        cmp(f"L{l}.xn", rd("xn", l), t(lc["xn"]))
        cmp(f"L{l}.vh", rd("vh", l), t(lc["vh"]))
        cmp(f"L{l}.qkv", rd("qkv", l), np.concatenate([t(lc["q"]), t(lc["k"]), t(lc["v"])], 1))
        vp = rd("vp", l).reshape(N, S, 5 * H)
        cmp(f"L{l}.vp", vp[..., :3 * H], t(lc["vp"]))
        pe = rd("pe", l).reshape(E_, 3 * H)
        cmp(f"L{l}.pk", pe[:, :H], t(lc["pk"]))
        cmp(f"L{l}.pv", pe[:, H:2 * H], t(lc["pv"]))
        if not last and l > 0:  # layer 0: vec == 0, these are identically zero / unused and never computed
            cmp(f"L{l}.wt", vp[..., 3 * H:4 * H], t(lc["wt"]))
            cmp(f"L{l}.ws", vp[..., 4 * H:], t(lc["ws"]))
            cmp(f"L{l}.pf", pe[:, 2 * H:], t(lc["pf"]))
        cmp(f"L{l}.m", rd("m", l), t(lc["m"]))
        cmp(f"L{l}.A", rd("A", l), t(lc["A"]))
        cmp(f"L{l}.tpre", rd("tpre", l), t(lc["tpre"]))
        cmp(f"L{l}.o", rd("o", l), t(lc["o"]))
    cmp("x_L", rd("x_in", L), t(c["x_L"]))
    cmp("vec_L", rd("vec_in", L), t(c["vec_L"]))
    cat0 = rd("cat0").reshape(N, 2 * H)
    cmp("head.xo", cat0[:, :H], t(c["xo"]))
    cmp("head.v1", cat0[:, H:], t(c["v1"]))
    cmp("head.vo", rd("vo"), t(c["vo"]))
    cmp("head.a0", rd("a0"), t(c["a0"]))
    cmp("head.u0", rd("u0"), t(c["u0"]))
    cmp("head.p1", rd("p1"), t(c["p1"]))
    cmp("head.a1b", rd("a1b"), t(c["a1b"]))
    cmp("head.y", rd("y"), t(c["y"]))
    e_gpu = e_out.cpu().numpy()
    nonempty = (g["end"] - g["start"]) > 0
    cmp("E", e_gpu[nonempty], E[:, 0])
    # reverse
    cmp("g_cat0", rd("g_cat0"), t(b["g_cat0"]))
    cmp("g_vo", rd("g_vo"), t(b["g_vo"]))
    cmp("g_x_L", rd("g_x_in", L), t(b["g_x_L"]))
    cmp("g_vec_L", rd("g_vec_in", L), t(b["g_vec_L"]))
    for l in reversed(range(L)):
        lb = b["layers"][l]
        last = l == L - 1
        cmp(f"L{l}.g_A", rd("g_A", l), t(lb["g_A"]))
        gvp = rd("g_vp", l).reshape(N, S, 5 * H)
        cmp(f"L{l}.g_vp", gvp[..., :3 * H], t(lb["g_vp"]))
        if not last and l > 0:
            cmp(f"L{l}.g_wt", gvp[..., 3 * H:4 * H], t(lb["g_wt"]))
            cmp(f"L{l}.g_ws", gvp[..., 4 * H:], t(lb["g_ws"]))
        cmp(f"L{l}.g_t", rd("g_t", l), t(lb["g_t"]))
        cmp(f"L{l}.g_m", rd("g_m", l), t(lb["g_m"]))
        gpe = rd("g_pe", l).reshape(E_, 3 * H)
        cmp(f"L{l}.g_pk", gpe[:, :H], t(lb["g_pk"]))
        cmp(f"L{l}.g_pv", gpe[:, H:2 * H], t(lb["g_pv"]))
        if not last and l > 0:
            cmp(f"L{l}.g_pf", gpe[:, 2 * H:], t(lb["g_pf"]))
        cmp(f"L{l}.g_qkv", rd("g_qkv", l), np.concatenate([t(lb["g_q"]), t(lb["g_k"]), t(lb["g_v"])], 1))
        if l > 0:  # the adjoint of vec_in(0) == 0 is not needed
            cmp(f"L{l}.g_vh", rd("g_vh", l), t(lb["g_vh"]))
            cmp(f"L{l}.g_vec_in", rd("g_vec_in", l), t(lb["g_vec_in"]))
        cmp(f"L{l}.g_xh", rd("g_xh", l), t(lb["g_xh"]))
        cmp(f"L{l}.g_x_in", rd("g_x_in", l), t(lb["g_x_in"]))
        cmp(f"L{l}.g_f_in", rd("g_f_in", l), t(lb["g_f_in"]))
    cmp("g_x_emb", rd("g_x"), t(b["g_x_emb"]))
    cmp("g_n", rd("g_n"), t(b["g_n"]))
    gpp = rd("g_pp").reshape(E_, 2 * H)
    cmp("g_phi", gpp[:, :H], t(b["g_phi"]))
    cmp("g_psi", gpp[:, H:], t(b["g_psi"]))
    cmp("g_rbf", rd("g_rbf").reshape(E_, Rp)[:, :R], t(b["g_rbf"]))
    gg = rd("g_geo").reshape(E_, 32)
    cmp("g_d", gg[:, :S] + gg[:, 16:16 + S] + gg[:, 24:24 + S], t(b["g_d"]))
    cmp("g_C", gg[:, 8], t(b["g_C"]))
    cmp("g_ev", rd("g_ev").reshape(E_, 4)[:, :3], t(b["g_ev"]))
    cmp("F", f_out.cpu().numpy(), F)
    cmp("F_vs_ref64", f_out.cpu().numpy(), g["F_ref64"])
    print("FIRST BAD TAPS:", bad[:8] if bad else "none - all stages agree")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "h64_l2"))
