"""Chunked (WY / UT-transform) form of the gated delta rule, forward and hand-written backward, in plain torch fp64 -- the algorithm
a chunked kernel pair would implement (round 3 built the forward on fp32 MFMA: 409 us against the sequential kernel's 363 at B = 8 -- it left the tree in round 6;
on split-bf16 operands its products would cost a fifth), checked here against the sequential recurrence the reference runs
(qwen3_next_attention.py:103-159) and its autograd gradients.  CPU only, no package import:  python tools/experimental/gdr_chunk_proto.py

Per (batch row, value head), chunk of C tokens with incoming state S0 [Dv, Dk] (rows of K, Q, V are tokens):
    g = cumsum(log alpha), gam = exp(g), D[i, j] = exp(g_i - g_j) (i >= j), Gp_j = exp(g_C - g_j)
    L = tril(K K^T * D, -1), N = I + diag(beta) L, R = N^-1, T = R diag(beta), P = tril(Q K^T * D)
    Uv = T V, Kw = T (gam * K)                                  -- independent of the state: chunk-parallel
    U = Uv - Kw S0^T,  O = scale (gam * (Q S0^T) + P U),  S_C = gam_C S0 + (Gp * U)^T K      -- the scan over chunks
"""
import math
import torch

torch.manual_seed(0)
F = torch.float64


def sequential(q, k, v, beta, alpha, S0):
    """q, k [S, Dk], v [S, Dv], beta, alpha [S]; returns o [S, Dv], final state [Dv, Dk]."""
    scale = q.shape[-1] ** -0.5
    st = S0
    out = []
    for t in range(q.shape[0]):
        st = alpha[t] * st
        u = beta[t] * (v[t] - st @ k[t])
        st = st + torch.outer(u, k[t])
        out.append(st @ (q[t] * scale))
    return torch.stack(out), st


def chunk_quantities(q, k, v, beta, alpha):
    C = q.shape[0]
    g = torch.cumsum(torch.log(alpha), 0)
    gam = torch.exp(g)
    D = torch.tril(torch.exp(g[:, None] - g[None, :]))
    Gp = torch.exp(g[-1] - g)
    KK, QK = k @ k.T, q @ k.T
    L = torch.tril(KK * D, -1)
    N = torch.eye(C, dtype=F) + beta[:, None] * L
    R = torch.linalg.inv(N)
    T = R * beta[None, :]
    P = torch.tril(QK * D)
    return dict(g=g, gam=gam, D=D, Gp=Gp, KK=KK, QK=QK, L=L, R=R, T=T, P=P, Uv=T @ v, Kw=T @ (gam[:, None] * k))


def chunk_forward(q, k, v, beta, alpha, S0):
    c = chunk_quantities(q, k, v, beta, alpha)
    scale = q.shape[-1] ** -0.5
    U = c["Uv"] - c["Kw"] @ S0.T
    O = scale * (c["gam"][:, None] * (q @ S0.T) + c["P"] @ U)
    SC = c["gam"][-1] * S0 + (c["Gp"][:, None] * U).T @ k
    return O, SC, c, U


def chunk_backward(q, k, v, beta, alpha, S0, dO, dSC):
    """Hand-written adjoint of chunk_forward.  Returns dq, dk, dv, dbeta, dalpha, dS0."""
    O, SC, c, U = chunk_forward(q, k, v, beta, alpha, S0)
    g, gam, D, Gp, KK, QK, L, R, T, P, Kw = (c[n] for n in ("g", "gam", "D", "Gp", "KK", "QK", "L", "R", "T", "P", "Kw"))
    scale = q.shape[-1] ** -0.5
    dOs = scale * dO
    # ---- the scan part (needs the state): O = gam * (Q S0^T) + P U;  S_C = gam_C S0 + (Gp * U)^T K;  U = Uv - Kw S0^T
    QS = q @ S0.T
    dQS = gam[:, None] * dOs
    dq = dQS @ S0
    dS0 = dQS.T @ q + gam[-1] * dSC
    dgam = (dOs * QS).sum(1)
    dgam[-1] = dgam[-1] + (dSC * S0).sum()
    dP = torch.tril(dOs @ U.T)
    dUg = k @ dSC.T
    dU = P.T @ dOs + Gp[:, None] * dUg
    dGp = (U * dUg).sum(1)
    dk = (Gp[:, None] * U) @ dSC
    dKw = -dU @ S0
    dS0 = dS0 - dU.T @ Kw
    # ---- the chunk-parallel part: Uv = T V, Kw = T (gam * K), T = R diag(beta), R = (I + diag(beta) L)^-1, L / P from K K^T, Q K^T and the decays
    gk = gam[:, None] * k
    dT = torch.tril(dU @ v.T + dKw @ gk.T)
    dv = T.T @ dU
    dGK = T.T @ dKw
    dk = dk + gam[:, None] * dGK
    dgam = dgam + (dGK * k).sum(1)
    dbeta = (dT * R).sum(0)
    dR = dT * beta[None, :]
    dN = -R.T @ dR @ R.T
    dbeta = dbeta + (torch.tril(dN, -1) * L).sum(1)
    dL = torch.tril(beta[:, None] * dN, -1)
    dKK, dQK = dL * D, dP * D
    E = (dL * KK + dP * QK) * D
    dk = dk + (dKK + dKK.T) @ k + dQK.T @ q
    dq = dq + dQK @ k
    dg = E.sum(1) - E.sum(0) + dgam * gam - dGp * Gp
    dg[-1] = dg[-1] + (dGp * Gp).sum()
    dlog = torch.flip(torch.cumsum(torch.flip(dg, [0]), 0), [0])
    return dq, dk, dU * 0 + dv, dbeta, dlog / alpha, dS0


def chunked(q, k, v, beta, alpha, S0, C):
    S, st, outs, states = q.shape[0], S0, [], []
    for c0 in range(0, S, C):
        sl = slice(c0, min(S, c0 + C))
        states.append(st)
        O, st, _, _ = chunk_forward(q[sl], k[sl], v[sl], beta[sl], alpha[sl], st)
        outs.append(O)
    return torch.cat(outs), st, states


def chunked_backward(q, k, v, beta, alpha, states, dO, dSfinal, C):
    S = q.shape[0]
    dq, dk, dv, db, da = (torch.zeros_like(t) for t in (q, k, v, beta, alpha))
    dS = dSfinal
    starts = list(range(0, S, C))
    for ci in reversed(range(len(starts))):
        sl = slice(starts[ci], min(S, starts[ci] + C))
        dq[sl], dk[sl], dv[sl], db[sl], da[sl], dS = chunk_backward(q[sl], k[sl], v[sl], beta[sl], alpha[sl], states[ci], dO[sl], dS)
    return dq, dk, dv, db, da, dS


if __name__ == "__main__":
    S, Dk, Dv, C = 150, 16, 24, 64
    q = torch.nn.functional.normalize(torch.randn(S, Dk, dtype=F), dim=-1).requires_grad_()
    k = torch.nn.functional.normalize(torch.randn(S, Dk, dtype=F), dim=-1).requires_grad_()
    v = torch.randn(S, Dv, dtype=F, requires_grad=True)
    beta = torch.sigmoid(torch.randn(S, dtype=F)).requires_grad_()
    alpha = torch.exp(-torch.exp(torch.randn(S, dtype=F) * 0.5) * torch.nn.functional.softplus(torch.randn(S, dtype=F))).requires_grad_()
    S0 = (torch.randn(Dv, Dk, dtype=F) * 0.3).requires_grad_()
    o_ref, s_ref = sequential(q, k, v, beta, alpha, S0)
    dO, dSf = torch.randn_like(o_ref), torch.randn_like(s_ref)
    grads = torch.autograd.grad((o_ref * dO).sum() + (s_ref * dSf).sum(), (q, k, v, beta, alpha, S0))
    with torch.no_grad():
        o, sf, states = chunked(q, k, v, beta, alpha, S0, C)
        print("forward  max |o - o_ref|", float((o - o_ref).abs().max()), " state", float((sf - s_ref).abs().max()))
        mine = chunked_backward(q, k, v, beta, alpha, states, dO, dSf, C)
        for n, a, b in zip(("dq", "dk", "dv", "dbeta", "dalpha", "dS0"), mine, grads):
            print(f"{n:7s} max abs err {float((a - b).abs().max()):.3e}   (max |ref| {float(b.abs().max()):.3e})")
