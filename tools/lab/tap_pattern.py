"""lab: WHAT differs in the first differing output of an attention launch (`m`, `A` of a layer; LAB_NOTES section 15)?
Keeps the per-layer debug copies of run 0 and, for every later run whose copies differ, prints for the first such layer:
which rows (edges / nodes) differ, how they cluster (runs, workgroups of 4 nodes, XCD = workgroup % 8), which columns,
and whether the differing rows carry the values ANOTHER layer's launch left in the shared buffer (a lost write / stale
read) or new values (the launch computed something else).
    python tools/lab/tap_pattern.py [reps] [nfrag]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from ai2bmd_amd.synthetic import default_hparams, make_state_dict
from ai2bmd_amd.visnet_calculator import ViSNetEngine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nfrag = int(sys.argv[2]) if len(sys.argv) > 2 else 400
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
H, L = hp["embedding_dimension"], hp["num_layers"]
eng = ViSNetEngine(hp, make_state_dict(hp, seed=2024), "cuda:0")
eng.set_option("debug", 1)


def take():
    e = torch.empty(len(start), device="cuda:0"); f = torch.empty(len(z), 3, device="cuda:0")
    eng.forces_device(z, pos, start, end, e, f)
    torch.cuda.synchronize()
    rowptr = eng.debug_read("rowptr", 0, dtype=np.int32, max_elems=1 << 24)
    out = {"rowptr": rowptr}
    for l in range(L):
        for nm in ("m", "A", "qkv", "pe"):
            out[nm, l] = eng.debug_read(nm, l, max_elems=1 << 29)
    return out


def runs(idx):
    if len(idx) == 0:
        return []
    cut = np.flatnonzero(np.diff(idx) != 1)
    a = np.concatenate([[0], cut + 1]); b = np.concatenate([cut, [len(idx) - 1]])
    return [(int(idx[i]), int(idx[j])) for i, j in zip(a, b)]


ref = take()
N = len(ref["rowptr"]) - 1
E = int(ref["rowptr"][-1])
print(f"pid {os.getpid()} nodes {N} edges {E} H {H} layers {L}: reference taken", flush=True)
for r in range(1, reps):
    cur = take()
    inputs_same = all(np.array_equal(cur[k], ref[k]) for k in ref if k != "rowptr" and k[0] in ("qkv", "pe"))
    first = next((l for l in range(L) if not (np.array_equal(cur["m", l], ref["m", l]) and np.array_equal(cur["A", l], ref["A", l]))), None)
    if first is None:
        print(f"rep {r}: m / A of every layer identical (qkv / pe identical: {inputs_same})", flush=True)
        continue
    l = first
    mr, mc = ref["m", l][: E * H].reshape(E, H), cur["m", l][: E * H].reshape(E, H)
    Ar, Ac = ref["A", l][: N * H].reshape(N, H), cur["A", l][: N * H].reshape(N, H)
    drow = np.flatnonzero((mr != mc).any(1)); dnode = np.flatnonzero((Ar != Ac).any(1))
    tgt = np.searchsorted(ref["rowptr"], drow, side="right") - 1  # the target node of each differing edge
    print(f"rep {r}: first differing layer {l} (qkv / pe of all layers identical: {inputs_same}); "
          f"m: {len(drow)} of {E} rows, A: {len(dnode)} of {N} rows", flush=True)
    rr = runs(drow)
    print(f"   m rows: {len(rr)} runs, first {rr[:6]}, lengths min/median/max "
          f"{min(b - a + 1 for a, b in rr) if rr else 0}/{int(np.median([b - a + 1 for a, b in rr])) if rr else 0}/{max(b - a + 1 for a, b in rr) if rr else 0}")
    tn = np.unique(tgt)
    full = sum(1 for i in tn if np.isin(np.arange(ref['rowptr'][i], ref['rowptr'][i + 1]), drow).all())
    print(f"   target nodes of those rows: {len(tn)} ({full} with ALL their edges differing), runs {runs(tn)[:6]}; "
          f"workgroups (node // 4) {len(np.unique(tn // 4))}, XCD histogram (wg % 8) {np.bincount((tn // 4) % 8, minlength=8).tolist()}")
    print(f"   A rows: runs {runs(dnode)[:6]}; A rows that are targets of differing m rows: {int(np.isin(dnode, tn).sum())} of {len(dnode)}")
    if len(drow):
        cols = (mr[drow] != mc[drow])
        print(f"   columns differing per m row: min {int(cols.sum(1).min())} max {int(cols.sum(1).max())}; per column block of 64: "
              f"{[int(cols[:, k * 64:(k + 1) * 64].any(1).sum()) for k in range(H // 64)]}")
        # byte addresses: 4 KB / 64 KB page alignment of the runs
        a0 = [a * H * 4 for a, _ in rr[:6]]
        print(f"   byte offsets of the first runs {a0} (mod 4096: {[x % 4096 for x in a0]}, mod 65536: {[x % 65536 for x in a0]})")
        # do the differing rows carry what another layer's launch left in the shared buffer?
        for name, other in [(f"ref m[{k}]", ref["m", k]) for k in range(L) if k != l] + [(f"cur m[{k}]", cur["m", k]) for k in range(L) if k != l]:
            o = other[: E * H].reshape(E, H)
            same = int((o[drow] == mc[drow]).all(1).sum())
            if same:
                print(f"   {same} of {len(drow)} differing rows EQUAL {name} (a value left in the shared buffer)")
        for e_ in drow[:8]:
            cc = np.flatnonzero(mr[e_] != mc[e_])
            t_ = int(np.searchsorted(ref["rowptr"], e_, side="right") - 1)
            print(f"      m row {int(e_)} (target {t_}, edge {int(e_ - ref['rowptr'][t_])} of {int(ref['rowptr'][t_ + 1] - ref['rowptr'][t_])}): columns {cc.tolist()}; "
                  f"zeros in ref {int((mr[e_, cc] == 0).sum())} / cur {int((mc[e_, cc] == 0).sum())}")
        for i_ in dnode[:8]:
            cc = np.flatnonzero(Ar[i_] != Ac[i_])
            with np.errstate(all="ignore"):
                print(f"      A row {int(i_)} (degree {int(ref['rowptr'][i_ + 1] - ref['rowptr'][i_])}): columns {cc.tolist()}; ref {Ar[i_, cc[:3]]} cur {Ac[i_, cc[:3]]}")
        qr, qc = ref["qkv", l], cur["qkv", l]; pr, pc = ref["pe", l], cur["pe", l]
        print(f"   inputs of layer {l}: qkv identical {np.array_equal(qr, qc)}, pe identical {np.array_equal(pr, pc)}")
        e0 = int(drow[0]); c0 = np.flatnonzero(mr[e0] != mc[e0])[:4]
        with np.errstate(all="ignore"):
            print(f"   sample row {e0}: ref {mr[e0, c0]} cur {mc[e0, c0]} ratio {mc[e0, c0] / mr[e0, c0]}")
            ratio = mc[e0] / mr[e0]
            print(f"   ratio over the row: min {np.nanmin(ratio):.6g} max {np.nanmax(ratio):.6g}; zeros in cur {int((mc[e0] == 0).sum())}, nan {int(np.isnan(mc[e0]).sum())}")
