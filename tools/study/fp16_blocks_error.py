"""Error study for storing the backward-only activation blocks of the SDF MLP as ONE fp16 piece (DESIGN.md section 4,
"Known inefficiencies (2)"): float64 restatement of the hand-derived double backward (csrc/svs_mlp_bwd.hip header) with
fp16 rounding inserted where the fp16x2 kernels would store / re-load a block, compared with the exact float64 result.

    python tools/study/fp16_blocks_error.py [n_points] [seed]

Rounding model: a stored value v of point p becomes fp16(v * s_p) / s_p with s_p a power of two that puts the point's
largest element of that block in [2^4, 2^5) (the per-point scale the sweeps already carry); h is stored as hi + mid
(22 bits) and the sweeps that only need softplus' read the hi piece alone.  CPU only; uses oracle/ and tests/golden/
(test infrastructure), never imported by the product.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("oracle", os.path.join("tests", "golden")):
    sys.path.insert(0, os.path.join(ROOT, p))
import synth          # noqa: E402
import torch_ref as tref   # noqa: E402

torch.set_default_dtype(torch.float64)


def q16(v, per_point_scale=True, headroom=4):
    """fp16 rounding of a (P, F) block under a per-point power-of-two scale."""
    if per_point_scale:
        m = v.abs().amax(1, keepdim=True).clamp_min(1e-300)
        s = torch.exp2(headroom - torch.floor(torch.log2(m)))
    else:
        s = torch.ones_like(v[:, :1])
    return (v * s).to(torch.float16).to(torch.float64) / s


def hi_piece(v):
    return v.to(torch.float16).to(torch.float64)


def manual_backward(W, b, x, nbar, sbar, fbar, quant):
    """-> [dW_0..dW_8], [db_0..db_8] by the algebra of svs_mlp_bwd.hip; quant: set of block names to round."""
    Q = lambda name, v, **kw: q16(v, **kw) if name in quant else v
    x = x.clone().requires_grad_(True)
    pe = tref.posenc(x, 6)
    J = torch.stack([torch.autograd.grad(pe[:, q].sum(), x, retain_graph=True)[0] for q in range(39)], 1)   # (P,39,3)
    pe = pe.detach()
    h = [pe]
    for l in range(8):
        inp = h[l] if l != 4 else torch.cat([h[4], pe], 1) / np.sqrt(2)
        h.append(torch.nn.functional.softplus(inp @ W[l].T + b[l], beta=100))
    # layer inputs as the kernels see them (the splice is part of lin4's input block)
    hin = [h[l] if l != 4 else torch.cat([h[4], pe], 1) / np.sqrt(2) for l in range(8)] + [h[8]]
    sp = lambda l, hh: 1 - torch.exp(-100 * hh)                       # s'(a_l) from h_{l+1}
    s1 = [sp(l, h[l + 1]) for l in range(8)]
    s1_sweeps = [sp(l, hi_piece(h[l + 1])) if "h_hi" in quant else s1[l] for l in range(8)]
    # gradient pass (inside sdf_full: exact values; the STORED copy ghat is what A / wgrad read)
    g = [None] * 9
    g[8] = W[8][0:1].expand(x.shape[0], -1)
    ghat = [None] * 8
    for l in range(7, -1, -1):
        ghat[l] = g[l + 1] * s1[l]
        gl = ghat[l] @ W[l]
        if l == 4:
            gl = gl[:, :217] / 1.0            # rows of h_4 only (1/sqrt2 is inside the product below)
            gl = (ghat[l] @ W[l])[:, :217] / np.sqrt(2)
        g[l] = gl
    ghat_st = [Q("ghat", v, per_point_scale="ghat_unscaled" not in quant) for v in ghat]
    # pass A
    u = [torch.einsum("pqc,pc->pq", J, nbar)]
    a2 = []
    u_st = [Q("u", u[0])]
    for l in range(8):
        uin = u[l] if l != 4 else torch.cat([u[4], u[0]], 1) / np.sqrt(2)
        v = uin @ W[l].T
        a2.append(v * ghat_st[l] * 100 * (1 - s1_sweeps[l]))
        u.append(v * s1_sweeps[l])
        u_st.append(Q("u", u[-1]))
    a2_st = [Q("a2", v) for v in a2]
    uin_st = [u_st[l] if l != 4 else torch.cat([u_st[4], u_st[0]], 1) / np.sqrt(2) for l in range(8)]
    # pass B
    hbar = sbar * W[8][0:1] + fbar @ W[8][1:]
    abar = [None] * 8
    for l in range(7, -1, -1):
        abar[l] = hbar * s1_sweeps[l] + a2_st[l]
        hb = abar[l] @ W[l]
        hbar = hb[:, :217] / np.sqrt(2) if l == 4 else hb
    abar_st = [Q("abar", v) for v in abar]
    hin_w = [hi_piece(v) if "h_wgrad_hi" in quant else v for v in hin]       # the weight gradient's B operand
    dW = [abar_st[l].T @ hin_w[l] + ghat_st[l].T @ uin_st[l] for l in range(8)]
    db = [abar_st[l].sum(0) for l in range(8)]
    a8bar = torch.cat([sbar, fbar], 1)
    dW8 = a8bar.T @ hin_w[8]
    dW8[0] += (g[8][:1] * 0).sum()            # (row 0 also receives u_8 through n: added below)
    dW8[0] += u_st[8].sum(0)
    dW.append(dW8); db.append(a8bar.sum(0))
    return dW, db


def autograd_reference(W, b, x, nbar, sbar, fbar):
    Wp = [w.clone().requires_grad_(True) for w in W]
    bp = [v.clone().requires_grad_(True) for v in b]
    x = x.clone().requires_grad_(True)
    pe = tref.posenc(x, 6)
    hcur = pe
    for l in range(9):
        if l == 4:
            hcur = torch.cat([hcur, pe], 1) / np.sqrt(2)
        hcur = hcur @ Wp[l].T + bp[l]
        if l < 8:
            hcur = torch.nn.functional.softplus(hcur, beta=100)
    sdf, feat = hcur[:, :1], hcur[:, 1:]
    n = torch.autograd.grad(sdf.sum(), x, create_graph=True)[0]
    loss = (n * nbar).sum() + (sdf * sbar).sum() + (feat * fbar).sum()
    loss.backward()
    return [w.grad for w in Wp], [v.grad for v in bp]


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    for label, params in (("init", synth.make_params(0)),) + ((("trained", synth.make_params(1, scale="trained")),)
                                                              if "scale" in synth.make_params.__code__.co_varnames else ()):
        p = tref.to_torch(params, torch.float64, False)
        W = [tref.weightnorm(p, "implicit_network", l) for l in range(9)]
        b = [p[f"implicit_network.lin{l}.bias"] for l in range(9)]
        K, pose = synth.make_camera()
        R = P // 64
        import svs_oracle as orc
        dirs, cam, _ = orc.rays_from_uv(synth.make_uv(R, seed=3), pose, K)
        z = np.sort(rng.uniform(0.3, 4.5, (R, 64)), -1)
        x = torch.tensor((cam[None, None] + z[:, :, None] * dirs[:, None, :]).reshape(-1, 3))
        # gradients as the loss produces them: a few surface samples per ray carry the weight, eikonal-like normals terms
        wgt = torch.tensor(rng.exponential(1.0, (x.shape[0], 1)) ** 4)
        wgt = wgt / wgt.sum() * 50
        nbar = torch.tensor(rng.normal(0, 1, (x.shape[0], 3))) * wgt
        sbar = torch.tensor(rng.normal(0, 1, (x.shape[0], 1))) * wgt
        fbar = torch.tensor(rng.normal(0, 1, (x.shape[0], 256))) * wgt * 0.1
        ref_W, ref_b = autograd_reference(W, b, x, nbar, sbar, fbar)
        exact_W, exact_b = manual_backward(W, b, x, nbar, sbar, fbar, set())
        chk = max(float((a - r).abs().max() / r.abs().max()) for a, r in zip(exact_W + exact_b, ref_W + ref_b))
        print(f"[{label}] {x.shape[0]} points; manual algebra vs float64 autograd: worst rel {chk:.2e}")
        for quant in (("abar",), ("u",), ("a2",), ("ghat",), ("ghat", "ghat_unscaled"), ("h_hi",), ("abar", "u", "a2", "ghat"),
                      ("abar", "u", "a2", "ghat", "h_hi"), ("h_wgrad_hi",), ("abar", "u", "a2", "ghat", "h_hi", "h_wgrad_hi")):
            qW, qb = manual_backward(W, b, x, nbar, sbar, fbar, set(quant))
            relmax = [float((a - r).abs().max() / r.abs().max()) for a, r in zip(qW + qb, ref_W + ref_b)]
            rel2 = [float((a - r).norm() / r.norm()) for a, r in zip(qW + qb, ref_W + ref_b)]
            print(f"  fp16 {'+'.join(quant):28s} worst max-norm rel {max(relmax):.2e} (dW {max(relmax[:9]):.2e}, db {max(relmax[9:]):.2e})"
                  f"   worst L2 rel {max(rel2):.2e}")


if __name__ == "__main__":
    main()
