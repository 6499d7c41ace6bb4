"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- stage-by-stage restatement of the DualVGR
unit with a HAND-DERIVED backward (no autograd), graph_layers = 1.

This is the algebra the HIP kernels implement: the factorised attention
logits, the packed 8-head weight (common GAT heads 0-3, specific GAT heads 4-7 per modality) and
the same stage boundaries S0..S6 as `csrc/`.  `forward` returns every intermediate the kernels
save, `backward` every intermediate gradient, so GPU tests can localise a mismatch to one stage.
tests/test_oracle_manual.py checks it against autograd through oracle/unit_torch.py in float64.

Reference lines restated: model/utils.py:66-84 (S0 QueryAttn), :92-105 (S1 QueryPunish),
model/GraphNN.py:95-113,174-178 (S2/S3 punish-GAT), model/Attention.py:20-23 +
model/models.py:163-169 (S4/S5 SFGCN attention + residual), model/fusions/fusions.py:419-453 (S6).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

DROP_P = 0.15
SLOPE = 0.01
MODS = ("app", "mot")
FAM = {"app": ("acGCN", "appearance_GCN", "queryPunish_appear", "attention_appearance"),
       "mot": ("mcGCN", "motion_GCN", "queryPunish_motion", "attention_motion")}


def _keep(masks, site, like):
    """Scaled keep factor mask/(1-p) (ones in eval mode)."""
    if masks is None:
        return torch.ones_like(like)
    return masks[site].to(like.dtype) / (1.0 - DROP_P)


def packed_gat(sd, mod: str, l: int):
    """8-head packed weights of one modality: rows [g*4F + h*F, ...) of Wcat are head h of GAT g
    (g = 0 common = acGCN/mcGCN, g = 1 specific = appearance_GCN/motion_GCN)."""
    Ws, bs, al, ar, cs = [], [], [], [], []
    for fam in FAM[mod][:2]:
        for h in range(4):
            p = f"{fam}.{l}.attention_{h}"
            W = sd[p + ".W.weight"]
            Fh = W.shape[0]
            Ws.append(W)
            bs.append(sd[p + ".W.bias"])
            al.append(sd[p + ".a.weight"][0, :Fh])
            ar.append(sd[p + ".a.weight"][0, Fh:])
            cs.append(sd[p + ".a.bias"][0])
    return torch.cat(Ws, 0), torch.cat(bs, 0), torch.stack(al, 0), torch.stack(ar, 0), torch.stack(cs, 0)


def elu(x):
    return torch.where(x > 0, x, torch.expm1(x))


def forward(sd: Dict[str, torch.Tensor], A, M, dq, words, qlen, L: int, masks: Optional[dict] = None):
    B, N, D = A.shape
    T = dq.shape[1]
    Fh = D // 4
    valid = (torch.arange(T)[None, :] < qlen[:, None]).to(A.dtype)
    cache = {"B": B, "N": N, "D": D, "T": T, "L": L, "valid": valid, "layers": []}
    X = {"app": A, "mot": M}
    lists = {"com_app": [], "com_mot": [], "aq": [], "mq": []}
    o_last = {}
    for l in range(L):
        c = {}
        # ---- S0: query side (independent of A/M) --------------------------------------------
        W1, b1 = sd[f"queryAttn.{l}.feat_enhance.weight"], sd[f"queryAttn.{l}.feat_enhance.bias"]
        w2, b2 = sd[f"queryAttn.{l}.fc.weight"][0], sd[f"queryAttn.{l}.fc.bias"][0]
        y = dq @ W1.T + b1
        nrm = y.norm(dim=-1).clamp_min(1e-12)
        d = y / nrm[..., None]
        r = d @ w2 + b2
        a = torch.softmax(r, dim=1)
        at = a * valid
        S = at.sum(1, keepdim=True) + 1e-5
        ah = at / S
        q = torch.einsum("bt,btw->bw", ah, words)
        c.update(y=y, nrm=nrm, d=d, a=a, S=S, ah=ah, q=q)
        for mod in MODS:
            x = X[mod]
            cm = {"x": x}
            Wq = sd[f"{FAM[mod][2]}.{l}.query_weight.weight"]
            bq = sd[f"{FAM[mod][2]}.{l}.query_weight.bias"]
            p = q @ Wq.T + bq                                           # S0 tail
            s = torch.sigmoid(torch.einsum("bnd,bd->bn", x, p))         # S1
            Wcat, bcat, al, ar, cc = packed_gat(sd, mod, l)
            outs = []
            Wh_all, u_all, v_all, P_all, alpha_all = [], [], [], [], []
            for g in range(2):
                site = f"L{l}.G0.{FAM[mod][g]}"
                kx = _keep(masks, site + ".x", x)
                xd = x * kx
                Wg = Wcat[g * 4 * Fh:(g + 1) * 4 * Fh]
                Wh = (xd @ Wg.T + bcat[g * 4 * Fh:(g + 1) * 4 * Fh]).view(B, N, 4, Fh)       # S2
                u = torch.einsum("bnhf,hf->bhn", Wh, al[g * 4:(g + 1) * 4])                  # S3
                v = torch.einsum("bnhf,hf->bhn", Wh, ar[g * 4:(g + 1) * 4])
                pre = u[:, :, :, None] + v[:, :, None, :] + cc[g * 4:(g + 1) * 4][None, :, None, None]
                e = torch.where(pre > 0, pre, SLOPE * pre)
                alpha = torch.softmax(e, dim=-1)
                ka = torch.stack([_keep(masks, f"{site}.h{h}.alpha", alpha[:, h]) for h in range(4)], 1)
                alpha_d = alpha * ka
                Z = Wh * s[:, :, None, None]
                P = torch.einsum("bhij,bjhf->bihf", alpha_d, Z)
                Oh = elu(P).reshape(B, N, 4 * Fh)
                ko = _keep(masks, site + ".out", Oh)
                outs.append(Oh * ko)
                Wh_all.append(Wh); u_all.append(u); v_all.append(v); P_all.append(P); alpha_all.append(alpha)
                cm[f"kx{g}"], cm[f"ka{g}"], cm[f"ko{g}"], cm[f"Oh{g}"] = kx, ka, ko, Oh
            z0, z1 = outs                                               # common, specific
            P1 = sd[f"{FAM[mod][3]}.{l}.project.0.weight"]
            bP = sd[f"{FAM[mod][3]}.{l}.project.0.bias"]
            p2 = sd[f"{FAM[mod][3]}.{l}.project.2.weight"][0]
            zz = torch.stack([z0, z1], 1)                               # (B,2,N,D)
            tn = torch.tanh(zz @ P1.T + bP)                             # S4
            w = tn @ p2                                                 # (B,2,N)
            beta = torch.softmax(w, dim=1)                              # S5
            o = (beta[..., None] * zz).sum(1)
            X[mod] = x + o
            o_last[mod] = o
            cm.update(p=p, s=s, Wh=torch.stack(Wh_all, 2).reshape(B, N, 8 * Fh),
                      u=torch.cat(u_all, 1), v=torch.cat(v_all, 1), alpha=torch.cat(alpha_all, 1),
                      z0=z0, z1=z1, tn=tn, beta=beta, o=o)
            c[mod] = cm
            if mod == "app":
                lists["com_app"].append(z0); lists["aq"].append(z1)
            else:
                lists["com_mot"].append(z0); lists["mq"].append(z1)
        cache["layers"].append(c)
    # ---- S6: MFB ------------------------------------------------------------------------------
    W0, b0 = sd["visualfusion.linear0.weight"], sd["visualfusion.linear0.bias"]
    W1m, b1m = sd["visualfusion.linear1.weight"], sd["visualfusion.linear1.bias"]
    Wo, bo = sd["visualfusion.linear_out.weight"], sd["visualfusion.linear_out.bias"]
    x0 = elu(X["app"] @ W0.T + b0)
    x1 = elu(X["mot"] @ W1m.T + b1m)
    zf = (x0 * x1).view(B, N, 256, 2).sum(-1)
    vis = elu(zf @ Wo.T + bo)
    cache.update(Af=X["app"], Mf=X["mot"], x0=x0, x1=x1, zf=zf, vis=vis)
    outs = (vis, o_last["app"], o_last["mot"], lists["com_app"], lists["com_mot"], lists["aq"], lists["mq"])
    return outs, cache


def backward(sd, cache, words, dq, g_vis, g_aqe, g_mqe, g_ca, g_cm, g_aq, g_mq, pre_pos=None):
    """Upstream gradients for all 3 + 4L outputs -> gradients of the 4 float inputs and of every
    parameter (keys of `sd`).  Returns (gin, gw, dbg) where dbg holds per-stage intermediates.

    pre_pos: optional {(layer, mod): bool (B, 8, N, N)} -- the branch of LeakyReLU'(pre) to take for every attention
    logit instead of this restatement's own `pre > 0`.  The derivative jumps at 0, so an implementation whose logits
    differ by rounding can legitimately sit on the other side of the kink for a few entries; the parity tests count
    those entries, check that they really lie at the kink, and compare gradients with the branches aligned."""
    B, N, D, T, L = cache["B"], cache["N"], cache["D"], cache["T"], cache["L"]
    Fh = D // 4
    gw = {k: torch.zeros_like(v) for k, v in sd.items()}
    dbg = {"layers": [dict() for _ in range(L)]}
    # ---- S6 backward ----------------------------------------------------------------------------
    W0, W1m, Wo = sd["visualfusion.linear0.weight"], sd["visualfusion.linear1.weight"], sd["visualfusion.linear_out.weight"]
    vis, x0, x1, zf = cache["vis"], cache["x0"], cache["x1"], cache["zf"]
    dpo = g_vis * torch.where(vis > 0, torch.ones_like(vis), vis + 1.0)
    gw["visualfusion.linear_out.weight"] = torch.einsum("bnd,bnk->dk", dpo, zf)
    gw["visualfusion.linear_out.bias"] = dpo.sum((0, 1))
    dz = (dpo @ Wo).repeat_interleave(2, dim=-1)                        # pair (2k, 2k+1) share dz_k
    dp0 = dz * x1 * torch.where(x0 > 0, torch.ones_like(x0), x0 + 1.0)
    dp1 = dz * x0 * torch.where(x1 > 0, torch.ones_like(x1), x1 + 1.0)
    gw["visualfusion.linear0.weight"] = torch.einsum("bnk,bnd->kd", dp0, cache["Af"])
    gw["visualfusion.linear0.bias"] = dp0.sum((0, 1))
    gw["visualfusion.linear1.weight"] = torch.einsum("bnk,bnd->kd", dp1, cache["Mf"])
    gw["visualfusion.linear1.bias"] = dp1.sum((0, 1))
    dX = {"app": dp0 @ W0, "mot": dp1 @ W1m}
    dbg.update(dpo=dpo, dp0=dp0, dp1=dp1)
    g_dq = torch.zeros_like(dq)
    g_words = torch.zeros_like(words)
    for l in reversed(range(L)):
        c = cache["layers"][l]
        dl = dbg["layers"][l]
        dq_bar = torch.zeros_like(c["q"])                               # grad wrt q (B,Wd)
        for mod in MODS:
            cm = c[mod]
            x, s, p = cm["x"], cm["s"], cm["p"]
            z0, z1, tn, beta = cm["z0"], cm["z1"], cm["tn"], cm["beta"]
            zz = torch.stack([z0, z1], 1)
            P1 = sd[f"{FAM[mod][3]}.{l}.project.0.weight"]
            p2 = sd[f"{FAM[mod][3]}.{l}.project.2.weight"][0]
            # ---- S5/S4 backward (SFGCN attention + residual) ------------------------------------
            do = dX[mod].clone()
            if l == L - 1:
                do = do + (g_aqe if mod == "app" else g_mqe)
            dbeta = torch.einsum("bnd,bvnd->bvn", do, zz)
            dw = beta * (dbeta - (beta * dbeta).sum(1, keepdim=True))
            dpre = dw[..., None] * p2 * (1.0 - tn * tn)                 # (B,2,N,D)
            gw[f"{FAM[mod][3]}.{l}.project.2.weight"] = torch.einsum("bvn,bvnd->d", dw, tn)[None]
            gw[f"{FAM[mod][3]}.{l}.project.0.weight"] = torch.einsum("bvnk,bvnd->kd", dpre, zz)
            gw[f"{FAM[mod][3]}.{l}.project.0.bias"] = dpre.sum((0, 1, 2))
            dzz = beta[..., None] * do[:, None] + dpre @ P1             # (B,2,N,D)
            gl_c = (g_ca if mod == "app" else g_cm)[l]
            gl_s = (g_aq if mod == "app" else g_mq)[l]
            dz = [dzz[:, 0] + gl_c, dzz[:, 1] + gl_s]
            # ---- S3/S2 backward (punish-GAT, 8 heads) -------------------------------------------
            Wcat, bcat, al, ar, cc = packed_gat(sd, mod, l)
            Wh8 = cm["Wh"].view(B, N, 8, Fh)
            dWh8 = torch.zeros_like(Wh8)
            ds = torch.zeros_like(s)
            dxd = []
            for g in range(2):
                fam = FAM[mod][g]
                ko, ka, kx = cm[f"ko{g}"], cm[f"ka{g}"], cm[f"kx{g}"]
                Wh = Wh8[:, :, g * 4:(g + 1) * 4]
                alpha = cm["alpha"][:, g * 4:(g + 1) * 4]
                u, v = cm["u"][:, g * 4:(g + 1) * 4], cm["v"][:, g * 4:(g + 1) * 4]
                dO = (dz[g] * ko).view(B, N, 4, Fh)
                # ELU'(P) = (O > 0 ? 1 : O + 1).  The kernels recover O from the saved, dropped output
                # (O = out*(1-p) where kept; dO = 0 elsewhere so the value there is irrelevant).
                Oh = cm[f"Oh{g}"].view(B, N, 4, Fh)
                dP = dO * torch.where(Oh > 0, torch.ones_like(Oh), Oh + 1.0)
                Z = Wh * s[:, :, None, None]
                alpha_d = alpha * ka
                dalpha = torch.einsum("bihf,bjhf->bhij", dP, Z) * ka
                dZ = torch.einsum("bhij,bihf->bjhf", alpha_d, dP)
                de = alpha * (dalpha - (alpha * dalpha).sum(-1, keepdim=True))
                pre = u[:, :, :, None] + v[:, :, None, :] + cc[g * 4:(g + 1) * 4][None, :, None, None]
                pos = pre > 0 if pre_pos is None else pre_pos[(l, mod)][:, g * 4:(g + 1) * 4]
                dpre_a = de * torch.where(pos, torch.ones_like(pre), torch.full_like(pre, SLOPE))
                du, dv = dpre_a.sum(-1), dpre_a.sum(-2)                 # (B,4,N)
                dWh = dZ * s[:, :, None, None] \
                    + torch.einsum("bhn,hf->bnhf", du, al[g * 4:(g + 1) * 4]) \
                    + torch.einsum("bhn,hf->bnhf", dv, ar[g * 4:(g + 1) * 4])
                ds = ds + (dZ * Wh).sum((2, 3))
                dal = torch.einsum("bhn,bnhf->hf", du, Wh)
                dar = torch.einsum("bhn,bnhf->hf", dv, Wh)
                dc = dpre_a.sum((0, 2, 3))
                dl.setdefault(mod + ".dc_abs", []).append(dpre_a.abs().sum((0, 2, 3)))   # cancellation scale of a.bias' gradient
                dWh8[:, :, g * 4:(g + 1) * 4] = dWh
                xd = x * kx
                dWh_flat = dWh.reshape(B, N, 4 * Fh)
                Wg = Wcat[g * 4 * Fh:(g + 1) * 4 * Fh]
                dxd.append((dWh_flat @ Wg) * kx)
                dWg = torch.einsum("bnk,bnd->kd", dWh_flat, xd)
                dbg_ = dWh_flat.sum((0, 1))
                for h in range(4):
                    pfx = f"{fam}.{l}.attention_{h}"
                    gw[pfx + ".W.weight"] = dWg[h * Fh:(h + 1) * Fh]
                    gw[pfx + ".W.bias"] = dbg_[h * Fh:(h + 1) * Fh]
                    gw[pfx + ".a.weight"] = torch.cat([dal[h], dar[h]])[None]
                    gw[pfx + ".a.bias"] = dc[h][None]
            # ---- S1 backward (QueryPunish) ------------------------------------------------------
            dtau = ds * s * (1.0 - s)
            dx = dX[mod] + dxd[0] + dxd[1] + dtau[..., None] * p[:, None, :]
            dp = torch.einsum("bn,bnd->bd", dtau, x)
            Wq = sd[f"{FAM[mod][2]}.{l}.query_weight.weight"]
            gw[f"{FAM[mod][2]}.{l}.query_weight.weight"] = dp.T @ c["q"]
            gw[f"{FAM[mod][2]}.{l}.query_weight.bias"] = dp.sum(0)
            dq_bar = dq_bar + dp @ Wq
            dl[mod] = dict(do=do, dw=dw, dpre=dpre, dz0=dz[0], dz1=dz[1], dWh=dWh8.reshape(B, N, 8 * Fh),
                           ds=ds, dtau=dtau, dp=dp, dx=dx)
            # cancellation scales of the bias gradients (column sums over rows): sum of the absolute terms, per column
            dl[mod + ".bias_abs"] = {"project.0.bias": dpre.abs().sum((0, 1, 2)), "W.bias": dWh8.reshape(B, N, 8 * Fh).abs().sum((0, 1)),
                                     "query_weight.bias": torch.einsum("bn,bnd->d", dtau.abs(), x.abs())}
            dX[mod] = dx
        # ---- S0 backward (QueryAttn) ----------------------------------------------------------------
        W1 = sd[f"queryAttn.{l}.feat_enhance.weight"]
        w2 = sd[f"queryAttn.{l}.fc.weight"][0]
        a, ah, S, d, nrm = c["a"], c["ah"], c["S"], c["d"], c["nrm"]
        dah = torch.einsum("btw,bw->bt", words, dq_bar)
        g_words = g_words + ah[..., None] * dq_bar[:, None, :]
        dat = (dah - (dah * ah).sum(1, keepdim=True)) / S
        da = dat * cache["valid"]
        dr = a * (da - (a * da).sum(1, keepdim=True))
        dd = dr[..., None] * w2
        gw[f"queryAttn.{l}.fc.weight"] = torch.einsum("bt,btd->d", dr, d)[None]
        gw[f"queryAttn.{l}.fc.bias"] = dr.sum()[None]
        dy = (dd - d * (d * dd).sum(-1, keepdim=True)) / nrm[..., None]
        # rows clamped by eps (||y|| < 1e-12) have zero gradient through the norm; not reachable here
        gw[f"queryAttn.{l}.feat_enhance.weight"] = torch.einsum("btk,btd->kd", dy, dq)
        gw[f"queryAttn.{l}.feat_enhance.bias"] = dy.sum((0, 1))
        g_dq = g_dq + dy @ W1
        dl.update(dq_bar=dq_bar, dr=dr, dy=dy)
        dl["feat_enhance.bias_abs"] = dy.abs().sum((0, 1))
    gin = {"A": dX["app"], "M": dX["mot"], "dq": g_dq, "words": g_words}
    return gin, gw, dbg
