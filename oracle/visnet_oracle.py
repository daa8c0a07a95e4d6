"""TEST INFRASTRUCTURE ONLY - CPU restatement ("oracle") of the reference's
ViSNet energy+force hot path.  Nothing in the product package imports this
module; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.

It restates, in plain torch (no torch_geometric / torch_scatter / torch_cluster),
the algorithm of

  /root/reference/src/ViSNet/model/visnet.py:135-166        (ViSNet.forward)
  /root/reference/src/ViSNet/model/visnet_block.py:103-142  (ViSNetBlock.forward)
  /root/reference/src/ViSNet/model/visnet_block.py:237-312  (ViS_MP)
  /root/reference/src/ViSNet/model/utils.py:10-57,119-341   (cutoff, rbf, SH, norms, embeddings)
  /root/reference/src/ViSNet/model/output_modules.py:52-62,136-140
  /root/reference/src/ViSNet/model/priors.py:86-87

Parity status: the reference has NO tests / golden vectors for this path
(SURVEY.md section 4), so this restatement is pinned against outputs of the
reference's own source run in the build container (oracle/ref_import.py +
oracle/make_golden.py -> tests/golden/*.npz; tests/test_oracle_vs_reference.py
re-runs the comparison live whenever /root/reference is present).

Two independent force paths are provided:
  * energy_forces()           - torch.autograd through the restated forward,
                                 exactly like visnet.py:153-165;
  * energy_forces_analytic()  - hand-derived reverse pass, stage by stage in
                                 the same decomposition the HIP kernels use
                                 (its intermediates are what the -m gpu tests
                                 compare the kernels' debug dumps against).
"""
from __future__ import annotations

import math

import numpy as np
import torch

SQRT3 = math.sqrt(3.0)


# --------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------
def silu(x):
    return x * torch.sigmoid(x)


def dsilu(x):
    s = torch.sigmoid(x)
    return s * (1.0 + x * (1.0 - s))


def _softplus_shift(x):
    return torch.nn.functional.softplus(x) - math.log(2.0)


# the reference's activation table (utils.py:93-116): name -> (f, f')
ACTIVATIONS = {
    "silu": (silu, dsilu),
    "swish": (silu, dsilu),
    "ssp": (_softplus_shift, torch.sigmoid),
    "tanh": (torch.tanh, lambda x: 1.0 - torch.tanh(x) ** 2),
    "sigmoid": (torch.sigmoid, lambda x: torch.sigmoid(x) * (1.0 - torch.sigmoid(x))),
}


def cosine_cutoff(r, rc):
    """utils.py:16-19"""
    return 0.5 * (torch.cos(r * (math.pi / rc)) + 1.0) * (r < rc).to(r.dtype)


def dcosine_cutoff(r, rc):
    return -0.5 * (math.pi / rc) * torch.sin(r * (math.pi / rc)) * (r < rc).to(r.dtype)


def sphere(u, lmax):
    """utils.py:131-160 (real spherical harmonics of the unit vector)."""
    x, y, z = u[..., 0], u[..., 1], u[..., 2]
    if lmax == 1:
        return torch.stack([x, y, z], dim=-1)
    return torch.stack(
        [
            x,
            y,
            z,
            SQRT3 * x * z,
            SQRT3 * x * y,
            y * y - 0.5 * (x * x + z * z),
            SQRT3 * y * z,
            SQRT3 / 2.0 * (z * z - x * x),
        ],
        dim=-1,
    )


def sphere_vjp(u, g, lmax):
    """J^T g for d = sphere(u): returns dE/du [E,3] from dE/dd [E,S]."""
    x, y, z = u[..., 0], u[..., 1], u[..., 2]
    if lmax == 1:
        return g[..., 0:3]
    gx = g[..., 0] + SQRT3 * z * g[..., 3] + SQRT3 * y * g[..., 4] - x * g[..., 5] - SQRT3 * x * g[..., 7]
    gy = g[..., 1] + SQRT3 * x * g[..., 4] + 2.0 * y * g[..., 5] + SQRT3 * z * g[..., 6]
    gz = g[..., 2] + SQRT3 * x * g[..., 3] - z * g[..., 5] + SQRT3 * y * g[..., 6] + SQRT3 * z * g[..., 7]
    return torch.stack([gx, gy, gz], dim=-1)


def vec_layer_norm(vec, weight, norm_type):
    """utils.py:186-249.  The `(dist == 0).all()` early-outs return the same
    values as the formulas below (0 / eps = 0), so they are not restated."""

    def one(v):
        if norm_type == "none":
            return v
        if norm_type == "rms":
            dist = torch.sqrt((v * v).sum(dim=1))  # [N,H]
            dist = dist.clamp(min=1e-12)
            dist = torch.sqrt(torch.mean(dist ** 2, dim=-1))  # [N]
            return v / torch.relu(dist)[:, None, None]
        if norm_type == "max_min":
            dist = torch.sqrt((v * v).sum(dim=1, keepdim=True))  # [N,1,H]
            dist = dist.clamp(min=1e-12)
            direct = v / dist
            max_val, _ = torch.max(dist, dim=-1)
            min_val, _ = torch.min(dist, dim=-1)
            delta = (max_val - min_val).view(-1)
            delta = torch.where(delta == 0, torch.ones_like(delta), delta)
            dist = (dist - min_val.view(-1, 1, 1)) / delta.view(-1, 1, 1)
            return torch.relu(dist) * direct
        raise ValueError(norm_type)

    S = vec.shape[1]
    if S == 3:
        out = one(vec)
    elif S == 8:
        out = torch.cat([one(vec[:, :3]), one(vec[:, 3:])], dim=1)
    else:
        raise ValueError("VecLayerNorm only support 3 or 8 channels")
    return out * weight[None, None, :]


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    n = (x - mu) * rstd
    return n * w + b, n, rstd


def layer_norm_vjp(g, n, rstd, w):
    gn = g * w
    return rstd * (gn - gn.mean(dim=-1, keepdim=True) - n * (gn * n).mean(dim=-1, keepdim=True))


def build_graph(pos, start, end, cutoff, max_nb):
    """radius graph of utils.py:259-266 (torch_cluster.radius_graph, loop=True):
    strict d^2 < rc^2, per-fragment, self loops kept, at most max_nb sources per
    target - the lowest source indices when truncated.  Returns numpy int64
    arrays: src, tgt (edges sorted by target, sources ascending), rowptr[N+1],
    and the source-sorted view: colptr[N+1], perm (edge ids grouped by source,
    ascending edge id inside a group)."""
    p = np.asarray(pos, dtype=np.float64)
    N = p.shape[0]
    src_l, tgt_l = [], []
    rc2 = float(cutoff) ** 2
    for s, e in zip(np.asarray(start).tolist(), np.asarray(end).tolist()):
        if e <= s:
            continue
        q = np.asarray(pos[s:e])
        diff = q[:, None, :] - q[None, :, :]
        d2 = (diff * diff).sum(-1)
        adj = d2 < np.asarray(rc2, dtype=d2.dtype)
        rank = np.cumsum(adj, axis=1)
        adj &= rank <= max_nb
        ti, sj = np.nonzero(adj)
        src_l.append(sj + s)
        tgt_l.append(ti + s)
    if src_l:
        src = np.concatenate(src_l).astype(np.int64)
        tgt = np.concatenate(tgt_l).astype(np.int64)
    else:
        src = np.zeros(0, np.int64)
        tgt = np.zeros(0, np.int64)
    rowptr = np.zeros(N + 1, np.int64)
    np.add.at(rowptr, tgt + 1, 1)
    rowptr = np.cumsum(rowptr)
    perm = np.argsort(src, kind="stable").astype(np.int64)
    colptr = np.zeros(N + 1, np.int64)
    np.add.at(colptr, src + 1, 1)
    colptr = np.cumsum(colptr)
    return dict(src=src, tgt=tgt, rowptr=rowptr, perm=perm, colptr=colptr)


# --------------------------------------------------------------------------
class ViSNetOracle:
    def __init__(self, hp, sd, dtype=torch.float64):
        self.hp = dict(hp)
        self.dtype = dtype
        if hp["rbf_type"] not in ("expnorm", "gauss"):
            raise NotImplementedError(hp["rbf_type"])
        self.H = hp["embedding_dimension"]
        self.L = hp["num_layers"]
        self.R = hp["num_rbf"]
        self.nh = hp["num_heads"]
        self.hd = self.H // self.nh
        self.lmax = hp["lmax"]
        self.S = (self.lmax + 1) ** 2 - 1
        self.rc = float(hp["cutoff"])
        self.max_nb = hp["max_num_neighbors"]
        self.vn = hp["vecnorm_type"]
        self.alpha = 5.0 / self.rc
        self.act, self.dact = ACTIVATIONS[hp["activation"]]
        self.aact, self.daact = ACTIVATIONS[hp["attn_activation"]]
        self.w = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in sd.items()}
        self.has_atomref = "prior_model.atomref.weight" in self.w

    # -- helpers ---------------------------------------------------------
    def W(self, name):
        return self.w[name]

    def _lin(self, x, name, bias=True):
        y = x @ self.w[name + ".weight"].T
        if bias:
            y = y + self.w[name + ".bias"]
        return y

    # -- forward ---------------------------------------------------------
    def forward(self, z, pos, start, end, graph=None):
        """Restated forward. `pos` is a torch tensor (may require grad).
        Returns (E[B] over NON-EMPTY fragments in order, cache)."""
        dt = self.dtype
        z = torch.as_tensor(np.asarray(z), dtype=torch.long)
        start = np.asarray(start, dtype=np.int64)
        end = np.asarray(end, dtype=np.int64)
        N = pos.shape[0]
        if graph is None:
            graph = build_graph(pos.detach().to(torch.float32 if dt == torch.float32 else dt).numpy(), start, end,
                                self.rc, self.max_nb)
        src = torch.from_numpy(graph["src"])
        tgt = torch.from_numpy(graph["tgt"])
        E = src.numel()
        H, S, nh, hd, rc = self.H, self.S, self.nh, self.hd, self.rc
        c = {}
        rm = "representation_model."

        # geometry (utils.py:267-274, visnet_block.py:113-117)
        loop = src == tgt
        ev = pos[src] - pos[tgt]
        r2 = (ev * ev).sum(-1)
        r_safe = torch.sqrt(torch.where(loop, torch.ones_like(r2), r2))
        r = torch.where(loop, torch.zeros_like(r2), r_safe)
        u = torch.where(loop[:, None], torch.zeros_like(ev), ev / r_safe[:, None])
        d = sphere(u, self.lmax)  # [E,S]
        C = cosine_cutoff(r, rc)  # [E]
        if self.hp["rbf_type"] == "gauss":  # utils.py:84-87: no cutoff factor
            offset = self.w[rm + "distance_expansion.offset"]
            coeff = self.w[rm + "distance_expansion.coeff"]
            rbf = torch.exp(coeff * (r[:, None] - offset) ** 2)
        else:
            means = self.w[rm + "distance_expansion.means"]
            betas = self.w[rm + "distance_expansion.betas"]
            t = torch.exp(-self.alpha * r)[:, None]
            ek = torch.exp(-betas * (t - means) ** 2)
            rbf = C[:, None] * ek  # [E,R]  utils.py:53-57
        c.update(src=src, tgt=tgt, loop=loop, r=r, u=u, d=d, C=C, rbf=rbf, ev=ev)

        # embeddings (visnet_block.py:110,118-122; utils.py:296-317,331-337)
        x0 = self.w[rm + "embedding.weight"][z]
        nl = (~loop).to(dt)
        phi = self._lin(rbf, rm + "neighbor_embedding.distance_proj")  # [E,H]
        Wn = phi * (C * nl)[:, None]
        emb2 = self.w[rm + "neighbor_embedding.embedding.weight"][z]
        nmsg = emb2[src] * Wn
        nagg = torch.zeros(N, H, dtype=dt).index_add(0, tgt, nmsg)
        x = self._lin(torch.cat([x0, nagg], dim=1), rm + "neighbor_embedding.combine")
        psi = self._lin(rbf, rm + "edge_embedding.edge_proj")
        f = (x[tgt] + x[src]) * psi
        vec = torch.zeros(N, S, H, dtype=dt)
        c.update(x0=x0, phi=phi, emb2=emb2, nagg=nagg, x_emb=x, psi=psi, f_emb=f)

        layers = []
        for l in range(self.L):
            last = l == self.L - 1
            p = f"{rm}vis_mp_layers.{l}."
            lc = {}
            lc["x_in"], lc["vec_in"], lc["f_in"] = x, vec, f
            xh, xn, rstd = layer_norm(x, self.w[p + "layernorm.weight"], self.w[p + "layernorm.bias"])
            vh = vec_layer_norm(vec, self.w[p + "vec_layernorm.weight"], self.vn)
            q = self._lin(xh, p + "q_proj")
            k = self._lin(xh, p + "k_proj")
            v = self._lin(xh, p + "v_proj")
            pk = self._lin(f, p + "dk_proj")
            pv = self._lin(f, p + "dv_proj")
            dk, dv = self.act(pk), self.act(pv)
            vp = vh @ self.w[p + "vec_proj.weight"].T  # [N,S,3H]
            vec1, vec2, vec3 = vp[..., :H], vp[..., H:2 * H], vp[..., 2 * H:]
            vec_dot = (vec1 * vec2).sum(dim=1)
            # message (visnet_block.py:276-288)
            sat = (q[tgt] * k[src] * dk).view(E, nh, hd).sum(-1)  # [E,nh]
            a = self.aact(sat) * C[:, None]
            m = (v[src] * dv).view(E, nh, hd) * a[:, :, None]
            m = m.reshape(E, H)
            tpre = self._lin(m, p + "s_proj")  # [E,2H]
            st = self.act(tpre)
            s1, s2 = st[:, :H], st[:, H:]
            mv = vh[src] * s1[:, None, :] + d[:, :, None] * s2[:, None, :]
            A = torch.zeros(N, H, dtype=dt).index_add(0, tgt, m)
            V = torch.zeros(N, S, H, dtype=dt).index_add(0, tgt, mv)
            lc.update(xn=xn, rstd=rstd, xh=xh, vh=vh, q=q, k=k, v=v, pk=pk, pv=pv, vp=vp, vec_dot=vec_dot,
                      sat=sat, a=a, m=m, tpre=tpre, A=A, V=V)
            if not last:
                wt = vh @ self.w[p + "w_trg_proj.weight"].T  # [N,S,H]
                ws = vh @ self.w[p + "w_src_proj.weight"].T
                pf = self._lin(f, p + "f_proj")
                u1, u2 = wt[tgt], ws[src]
                a1 = (u1 * d[:, :, None]).sum(1)
                a2 = (u2 * d[:, :, None]).sum(1)
                # vector_rejection twice (visnet_block.py:206-209,290-295)
                w1 = u1 - a1[:, None, :] * d[:, :, None]
                w2 = u2 - a2[:, None, :] * d[:, :, None]
                wd = (w1 * w2).sum(1)
                df = self.act(pf) * wd
                lc.update(wt=wt, ws=ws, pf=pf, wd=wd, df=df)
            o = self._lin(A, p + "o_proj")
            o1, o2, o3 = o[:, :H], o[:, H:2 * H], o[:, 2 * H:]
            dx = vec_dot * o2 + o3
            dvec = vec3 * o1[:, None, :] + V
            lc.update(o=o, dx=dx, dvec=dvec)
            x = x + dx
            vec = vec + dvec
            if not last:
                f = f + df
            layers.append(lc)
        c["layers"] = layers
        c["x_L"], c["vec_L"] = x, vec

        # read-out (visnet_block.py:139-140, output_modules.py:52-62,136-140)
        xo, xon, xorstd = layer_norm(x, self.w[rm + "out_norm.weight"], self.w[rm + "out_norm.bias"])
        vo = vec_layer_norm(vec, self.w[rm + "vec_out_norm.weight"], self.vn)
        on = "output_model.output_network."
        p0 = vo @ self.w[on + "0.vec1_proj.weight"].T  # [N,S,H]
        v1 = torch.sqrt((p0 * p0).sum(dim=1))  # torch.norm(dim=-2)
        v2 = vo @ self.w[on + "0.vec2_proj.weight"].T  # [N,S,H/2]
        a0 = self._lin(torch.cat([xo, v1], dim=-1), on + "0.update_net.0")
        u0 = self._lin(self.act(a0), on + "0.update_net.2")  # [N,H]
        h2 = H // 2
        xs, gate = u0[:, :h2], u0[:, h2:]
        vec1o = gate[:, None, :] * v2
        x1 = self.act(xs)
        p1 = vec1o @ self.w[on + "1.vec1_proj.weight"].T  # [N,S,H/2]
        v1b = torch.sqrt((p1 * p1).sum(dim=1))
        a1b = self._lin(torch.cat([x1, v1b], dim=-1), on + "1.update_net.0")
        u1b = self._lin(self.act(a1b), on + "1.update_net.2")  # [N,2]
        y = u1b[:, 0:1]
        y = y * self.w["std"]
        if self.has_atomref:
            y = y + self.w["prior_model.atomref.weight"][z]
        c.update(xo=xo, xon=xon, xorstd=xorstd, vo=vo, p0=p0, v1=v1, v2=v2, a0=a0, u0=u0, p1=p1, v1b=v1b,
                 a1b=a1b, y=y)
        # per-fragment sum over NON-EMPTY fragments (visnet.py:146: scatter over batch ids)
        valid = np.flatnonzero(end - start)
        batch = np.zeros(N, np.int64)
        for bi, fi in enumerate(valid):
            batch[start[fi]:end[fi]] = bi
        Eb = torch.zeros(len(valid), 1, dtype=dt).index_add(0, torch.from_numpy(batch), y)
        if self.hp.get("reduce_op", "add") == "mean":  # visnet.py:146 scatter(..., reduce=self.reduce_op)
            Eb = Eb / torch.as_tensor((end - start)[valid], dtype=dt)[:, None]
        Eb = Eb + self.w["mean"]
        c["graph"] = graph
        c["z"] = z
        return Eb[:, 0], c

    # -- autograd forces (visnet.py:153-165) -------------------------------
    def energy_forces(self, z, pos, start, end):
        p = torch.as_tensor(np.asarray(pos)).to(self.dtype).clone().requires_grad_(True)
        Eb, c = self.forward(z, p, start, end)
        (g,) = torch.autograd.grad(Eb.sum(), p)
        return Eb.detach().numpy().reshape(-1, 1), (-g).numpy(), c

    # -- hand-derived reverse pass ---------------------------------------
    @torch.no_grad()
    def energy_forces_analytic(self, z, pos, start, end):
        if self.vn != "none":
            raise NotImplementedError("analytic reverse pass restated for vecnorm_type='none'")
        dt = self.dtype
        p = torch.as_tensor(np.asarray(pos)).to(dt)
        Eb, c = self.forward(z, p, start, end)
        src, tgt, loop = c["src"], c["tgt"], c["loop"]
        d, C, r, u = c["d"], c["C"], c["r"], c["u"]
        N, E = p.shape[0], src.numel()
        H, S, nh, hd, rc = self.H, self.S, self.nh, self.hd, self.rc
        h2 = H // 2
        rm = "representation_model."
        on = "output_model.output_network."
        w = self.w
        b = {}

        # ---- read-out ----
        g_h1 = w["std"] * w[on + "1.update_net.2.weight"][0][None, :].expand(N, h2)
        if self.hp.get("reduce_op", "add") == "mean":  # dE_b / dy_i = 1 / n_b
            st_, en_ = np.asarray(start), np.asarray(end)
            g_h1 = g_h1 / torch.as_tensor(np.repeat((en_ - st_), (en_ - st_)), dtype=dt)[:, None]
        g_a1 = g_h1 * self.dact(c["a1b"])
        g_cat1 = g_a1 @ w[on + "1.update_net.0.weight"]  # [N,H]
        g_x1, g_v1b = g_cat1[:, :h2], g_cat1[:, h2:]
        inv = torch.where(c["v1b"] > 0, 1.0 / c["v1b"].clamp(min=1e-300), torch.zeros_like(c["v1b"]))
        g_p1 = (g_v1b * inv)[:, None, :] * c["p1"]
        g_vec1o = g_p1 @ w[on + "1.vec1_proj.weight"]  # [N,S,h2]
        gate = c["u0"][:, h2:]
        xs = c["u0"][:, :h2]
        g_gate = (g_vec1o * c["v2"]).sum(1)
        g_v2 = g_vec1o * gate[:, None, :]
        g_xs = g_x1 * self.dact(xs)
        g_u0 = torch.cat([g_xs, g_gate], dim=1)
        g_h0 = g_u0 @ w[on + "0.update_net.2.weight"]
        g_a0 = g_h0 * self.dact(c["a0"])
        g_cat0 = g_a0 @ w[on + "0.update_net.0.weight"]  # [N,2H]
        g_xo, g_v1 = g_cat0[:, :H], g_cat0[:, H:]
        inv = torch.where(c["v1"] > 0, 1.0 / c["v1"].clamp(min=1e-300), torch.zeros_like(c["v1"]))
        g_p0 = (g_v1 * inv)[:, None, :] * c["p0"]
        g_vo = g_p0 @ w[on + "0.vec1_proj.weight"] + g_v2 @ w[on + "0.vec2_proj.weight"]
        g_x = layer_norm_vjp(g_xo, c["xon"], c["xorstd"], w[rm + "out_norm.weight"])
        g_vec = g_vo * w[rm + "vec_out_norm.weight"][None, None, :]
        b["g_x_L"], b["g_vec_L"] = g_x, g_vec
        b["g_cat0"], b["g_vo"] = g_cat0, g_vo

        g_f = torch.zeros(E, H, dtype=dt)
        g_d = torch.zeros(E, S, dtype=dt)
        g_C = torch.zeros(E, dtype=dt)
        bl = [None] * self.L
        for l in reversed(range(self.L)):
            last = l == self.L - 1
            pfx = f"{rm}vis_mp_layers.{l}."
            lc = c["layers"][l]
            lb = {}
            vp = lc["vp"]
            vec1, vec2, vec3 = vp[..., :H], vp[..., H:2 * H], vp[..., 2 * H:]
            o = lc["o"]
            o1, o2 = o[:, :H], o[:, H:2 * H]
            vh = lc["vh"]
            # node update (visnet_block.py:271-274)
            g_o = torch.cat([(g_vec * vec3).sum(1), g_x * lc["vec_dot"], g_x], dim=1)
            g_vdot = g_x * o2
            g_vp = torch.cat([g_vdot[:, None, :] * vec2, g_vdot[:, None, :] * vec1, g_vec * o1[:, None, :]], dim=-1)
            g_A = g_o @ w[pfx + "o_proj.weight"]
            g_vh = g_vp @ w[pfx + "vec_proj.weight"]
            lb.update(g_o=g_o, g_vp=g_vp, g_A=g_A)
            # edge update (not in the last layer)
            g_pe_f = None
            if not last:
                wt, ws, pf = lc["wt"], lc["ws"], lc["pf"]
                u1, u2 = wt[tgt], ws[src]
                a1 = (u1 * d[:, :, None]).sum(1)
                a2 = (u2 * d[:, :, None]).sum(1)
                cc = (d * d).sum(1) - 2.0  # [E]
                wd = (u1 * u2).sum(1) + a1 * a2 * cc[:, None]
                g_pf = g_f * wd * self.dact(pf)
                g_wd = g_f * self.act(pf)
                g_u1 = g_wd[:, None, :] * (u2 + (a2 * cc[:, None])[:, None, :] * d[:, :, None])
                g_u2 = g_wd[:, None, :] * (u1 + (a1 * cc[:, None])[:, None, :] * d[:, :, None])
                g_wt = torch.zeros(N, S, H, dtype=dt).index_add(0, tgt, g_u1)
                g_ws = torch.zeros(N, S, H, dtype=dt).index_add(0, src, g_u2)
                g_d += (g_wd[:, None, :] * (cc[:, None, None] * (a2[:, None, :] * u1 + a1[:, None, :] * u2)
                                            + 2.0 * (a1 * a2)[:, None, :] * d[:, :, None])).sum(-1)
                g_vh = g_vh + g_wt @ w[pfx + "w_trg_proj.weight"] + g_ws @ w[pfx + "w_src_proj.weight"]
                g_pe_f = g_pf
                lb.update(g_wt=g_wt, g_ws=g_ws, g_pf=g_pf)
            # vector messages
            st = self.act(lc["tpre"])
            s1, s2 = st[:, :H], st[:, H:]
            gV = g_vec[tgt]  # [E,S,H]
            g_s1 = (gV * vh[src]).sum(1)
            g_s2 = (gV * d[:, :, None]).sum(1)
            g_d += (gV * s2[:, None, :]).sum(-1)
            g_vh = g_vh.index_add(0, src, gV * s1[:, None, :])
            g_t = torch.cat([g_s1, g_s2], dim=1) * self.dact(lc["tpre"])
            g_m = g_t @ w[pfx + "s_proj.weight"] + g_A[tgt]
            lb.update(g_t=g_t, g_m=g_m)
            # attention
            q, k, v = lc["q"], lc["k"], lc["v"]
            dk, dv = self.act(lc["pk"]), self.act(lc["pv"])
            a = lc["a"]
            aH = a[:, :, None].expand(E, nh, hd).reshape(E, H)
            g_v_e = g_m * dv * aH
            g_dv = g_m * v[src] * aH
            g_a = (g_m * v[src] * dv).view(E, nh, hd).sum(-1)
            sat = lc["sat"]
            g_sat = g_a * self.daact(sat) * C[:, None]
            g_C += (g_a * self.aact(sat)).sum(-1)
            gsH = g_sat[:, :, None].expand(E, nh, hd).reshape(E, H)
            g_q = torch.zeros(N, H, dtype=dt).index_add(0, tgt, gsH * k[src] * dk)
            g_k = torch.zeros(N, H, dtype=dt).index_add(0, src, gsH * q[tgt] * dk)
            g_vn = torch.zeros(N, H, dtype=dt).index_add(0, src, g_v_e)
            g_pk = gsH * q[tgt] * k[src] * self.dact(lc["pk"])
            g_pv = g_dv * self.dact(lc["pv"])
            lb.update(g_sat=g_sat, g_q=g_q, g_k=g_k, g_v=g_vn, g_pk=g_pk, g_pv=g_pv)
            g_f = g_f + g_pk @ w[pfx + "dk_proj.weight"] + g_pv @ w[pfx + "dv_proj.weight"]
            if not last:
                g_f = g_f + g_pe_f @ w[pfx + "f_proj.weight"]
            g_xh = g_q @ w[pfx + "q_proj.weight"] + g_k @ w[pfx + "k_proj.weight"] + g_vn @ w[pfx + "v_proj.weight"]
            g_x = g_x + layer_norm_vjp(g_xh, lc["xn"], lc["rstd"], w[pfx + "layernorm.weight"])
            g_vec = g_vec + g_vh * w[pfx + "vec_layernorm.weight"][None, None, :]
            lb.update(g_vh=g_vh, g_xh=g_xh, g_x_in=g_x, g_vec_in=g_vec, g_f_in=g_f)
            bl[l] = lb
        b["layers"] = bl

        # ---- embeddings ----
        x = c["x_emb"]
        psi = c["psi"]
        g_psi = g_f * (x[tgt] + x[src])
        g_xe = g_x + torch.zeros(N, H, dtype=dt).index_add(0, tgt, g_f * psi).index_add(0, src, g_f * psi)
        g_rbf = g_psi @ w[rm + "edge_embedding.edge_proj.weight"]  # [E,R]
        Wc = w[rm + "neighbor_embedding.combine.weight"]
        g_n = g_xe @ Wc[:, H:]
        nl = (~loop).to(dt)
        g_Wn = g_n[tgt] * c["emb2"][src] * nl[:, None]
        g_phi = g_Wn * C[:, None]
        g_C += (g_Wn * c["phi"]).sum(-1)
        g_rbf = g_rbf + g_phi @ w[rm + "neighbor_embedding.distance_proj.weight"]
        b.update(g_x_emb=g_xe, g_rbf=g_rbf, g_d=g_d, g_C=g_C, g_n=g_n, g_phi=g_phi, g_psi=g_psi)

        # ---- geometry ----
        dC = dcosine_cutoff(r, rc)
        if self.hp["rbf_type"] == "gauss":
            offset, coeff = w[rm + "distance_expansion.offset"], w[rm + "distance_expansion.coeff"]
            dr = r[:, None] - offset
            drbf = 2.0 * coeff * dr * torch.exp(coeff * dr ** 2)
        else:
            means = w[rm + "distance_expansion.means"]
            betas = w[rm + "distance_expansion.betas"]
            tt = torch.exp(-self.alpha * r)[:, None]
            ek = torch.exp(-betas * (tt - means) ** 2)
            dek = 2.0 * self.alpha * betas * tt * (tt - means) * ek
            drbf = dC[:, None] * ek + C[:, None] * dek
        g_r = (g_rbf * drbf).sum(-1) + g_C * dC
        g_u = sphere_vjp(u, g_d, self.lmax)
        rinv = torch.where(loop, torch.zeros_like(r), 1.0 / torch.where(loop, torch.ones_like(r), r))
        g_ev = g_r[:, None] * u + (g_u - (g_u * u).sum(-1, keepdim=True) * u) * rinv[:, None]
        g_ev = g_ev * (~loop).to(dt)[:, None]
        g_pos = torch.zeros(N, 3, dtype=dt).index_add(0, src, g_ev).index_add(0, tgt, -g_ev)
        b.update(g_r=g_r, g_ev=g_ev)
        return Eb.numpy().reshape(-1, 1), (-g_pos).numpy(), c, b
