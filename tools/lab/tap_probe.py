"""lab: which intermediate buffer of a fragment-batch evaluation changes from call to call (next to a neighbour process)?
    python tools/lab/tap_probe.py [reps] [nfrag]"""
import os, sys, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from ai2bmd_amd.synthetic import default_hparams, make_state_dict
from ai2bmd_amd.visnet_calculator import ViSNetEngine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nfrag = int(sys.argv[2]) if len(sys.argv) > 2 else 3600
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")
pool = []
for name in ("chig", "trpcage", "ww", "abd"):
    g = np.load(os.path.join(G, f"visnet_prot_{name}.npz"))
    for a, b in zip(g["start"], g["end"]):
        pool.append((g["z"][a:b], g["pos_relaxed"][a:b]))
rng = np.random.default_rng(7)
zs, ps, sizes = [], [], []
for i in range(nfrag):
    z, p = pool[i % len(pool)]
    zs.append(z); sizes.append(len(z)); ps.append(p if i < len(pool) else p + rng.normal(0, 0.05, size=p.shape))
end = np.cumsum(sizes); start = end - np.asarray(sizes)
z = torch.as_tensor(np.concatenate(zs), dtype=torch.int64).cuda()
pos = torch.as_tensor(np.concatenate(ps).astype(np.float32)).cuda()
hp = default_hparams()
eng = ViSNetEngine(hp, make_state_dict(hp, seed=2024), "cuda:0")
L = hp["num_layers"]
DEBUG = bool(os.environ.get("TAP_DEBUG"))
if DEBUG:
    eng.set_option("debug", 1)  # + copies of the running buffers per layer (x_in, vec_in, f_in, m, A) right behind their producers
GLOBAL = ["geo", "d", "rbf", "pp", "cat", "x_emb", "x", "vec", "f", "cat0", "vo", "pv0", "y", "g_cat0", "g_vo", "g_x", "g_n",
          "g_pp", "g_rbf", "g_geo"]
LAYER = ["xn", "rstd", "vh", "qkv", "vp", "pe", "tpre", "o"] + (["x_in", "vec_in", "f_in", "m", "A"] if DEBUG else [])

def snapshot():
    out = {}
    for nm in ("rowptr", "src", "perm"):
        out[nm] = zlib.crc32(eng.debug_read(nm, 0, dtype=np.int32, max_elems=1 << 24).tobytes())
    for nm in GLOBAL:
        out[nm] = zlib.crc32(eng.debug_read(nm, 0, max_elems=1 << 29).tobytes())
    for l in range(L):
        for nm in LAYER:
            out[f"{nm}[{l}]"] = zlib.crc32(eng.debug_read(nm, l, max_elems=1 << 29).tobytes())
    return out

ref = None
for r in range(reps):
    e = torch.empty(len(start), device="cuda:0"); f = torch.empty(len(z), 3, device="cuda:0")
    eng.forces_device(z, pos, start, end, e, f)
    torch.cuda.synchronize()
    s = snapshot()
    s["E"] = zlib.crc32(e.cpu().numpy().tobytes()); s["F"] = zlib.crc32(f.cpu().numpy().tobytes())
    if ref is None:
        ref = s
        print(f"pid {os.getpid()} atoms {len(z)} frags {nfrag}: reference taken", flush=True)
    else:
        bad = [k for k in ref if s[k] != ref[k]]
        print(f"rep {r}: {len(bad)} taps differ: {bad[:40]}", flush=True)
