"""NumPy restatement of the HAND-DERIVED backward that csrc/cgan.hip launches.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  This file mirrors the
kernel sequence of ``agx_cgan_disc_grads`` / ``agx_cgan_gen_grads`` step by step
(stacked [real; fake; inter] batch, first-order delta chain, in-place
second-order chain of the gradient penalty) so that the derivation can be
checked against the autograd oracle (``cgan_oracle``) on the CPU, without a GPU,
and so that data-parallel sharding (1/B_global scaling + gradient sum) can be
exercised with gloo.  It is never imported by the product.
"""
import numpy as np


def _cov(p, cc, nc):
    return np.concatenate([nc] + [p['emb'][i][cc[:, i]] for i in range(len(p['emb']))], axis=1)


def _act_grad(a, slope):
    return np.where(a > 0, 1.0, slope).astype(a.dtype)


def gen_forward(p, z, cc, nc, slope=0.0, keep=False):
    h = np.concatenate([z, _cov(p, cc, nc)], axis=1)
    acts = [h]
    n = len(p['W'])
    for l in range(n):
        h = h @ p['W'][l] + p['b'][l]
        if l < n - 1:
            h = np.where(h > 0, h, slope * h)
        acts.append(h)
    return (h, acts) if keep else h


def _scatter_emb(p, cc, dcov, col0):
    out, off = [], col0
    for i, tbl in enumerate(p['emb']):
        g = np.zeros_like(tbl)
        e = tbl.shape[1]
        np.add.at(g, cc[:, i], dcov[:, off:off + e])
        out.append(g)
        off += e
    return out


def disc_grads(x, z, cc, nc, gen_p, disc_p, alpha, gp_weight=10.0, B_global=None, slope=0.0,
               aug_mask=None, aug_noise_fake=None, aug_noise_real=None):
    """Returns (losses [wgan, gp, total], grads dict like the params)."""
    B, G = x.shape
    Bg = B if B_global is None else B_global
    invb = 1.0 / Bg
    W, b = disc_p['W'], disc_p['b']
    L = len(W) - 1
    n_num = nc.shape[1]
    xf = gen_forward(gen_p, z, cc, nc, slope)
    xr = x.copy()
    if aug_mask is not None:
        xf = xf + aug_mask[:, None] * aug_noise_fake[None, :]
        xr = xr + aug_mask[:, None] * aug_noise_real
    xi = xr + alpha.reshape(-1, 1) * (xf - xr)
    cov = _cov(disc_p, cc, nc)
    X = np.concatenate([np.concatenate([xr, cov], 1), np.concatenate([xf, cov], 1),
                        np.concatenate([xi, cov], 1)], 0)
    A = [X]
    for l in range(L):
        pre = A[l] @ W[l] + b[l]
        A.append(np.where(pre > 0, pre, slope * pre))
    out = A[L] @ W[L][:, 0] + b[L][0]
    c = np.concatenate([np.full(B, -invb), np.full(B, invb), np.ones(B)]).astype(x.dtype)
    D = [None] * (L + 1)
    D[L] = _act_grad(A[L], slope) * (c[:, None] * W[L][:, 0][None, :])
    for l in range(L, 1, -1):
        D[l - 1] = _act_grad(A[l - 1], slope) * (D[l] @ W[l - 1].T)
    g = D[1][2 * B:] @ W[0][:G].T
    s = np.sqrt(np.sum(g * g, axis=1))
    with np.errstate(all='ignore'):
        dg = (gp_weight * 2.0 * invb) * (s - 1.0) / s
        dg = g * dg[:, None]
    wgan = -np.sum(out[:B]) * invb + np.sum(out[B:2 * B]) * invb
    gp = np.sum((s - 1.0) ** 2) * invb
    # second-order chain in place over the inter rows
    X2 = X.copy()
    X2[2 * B:, :G] = dg
    X2[2 * B:, G:] = 0.0
    A2 = [X2] + [a.copy() for a in A[1:]]
    e = dg
    for l in range(L):
        Wl = W[l][:G] if l == 0 else W[l]
        e = _act_grad(A[l + 1][2 * B:], slope) * (e @ Wl)
        A2[l + 1][2 * B:] = e
    gW, gb = [None] * (L + 1), [None] * (L + 1)
    gW[L] = (c[:, None] * A2[L]).sum(0)[:, None]
    gb[L] = np.array([B * (-invb) + B * invb], dtype=x.dtype)
    for l in range(L - 1, -1, -1):
        gW[l] = A2[l].T @ D[l + 1]
        gb[l] = D[l + 1][:2 * B].sum(0)
    dcov = D[1][:2 * B] @ W[0][G:].T
    gemb = _scatter_emb(disc_p, np.concatenate([cc, cc], 0), dcov, n_num)
    return np.array([wgan, gp, wgan + gp_weight * gp]), {'emb': gemb, 'W': gW, 'b': gb}


def gen_grads(z, cc, nc, gen_p, disc_p, B_global=None, slope=0.0, aug_mask=None, aug_noise=None):
    B = z.shape[0]
    Bg = B if B_global is None else B_global
    invb = 1.0 / Bg
    W, b = disc_p['W'], disc_p['b']
    L = len(W) - 1
    G = gen_p['W'][-1].shape[1]
    xg, gacts = gen_forward(gen_p, z, cc, nc, slope, keep=True)
    if aug_mask is not None:
        xg = xg + aug_mask[:, None] * aug_noise
    A = [np.concatenate([xg, _cov(disc_p, cc, nc)], 1)]
    for l in range(L):
        pre = A[l] @ W[l] + b[l]
        A.append(np.where(pre > 0, pre, slope * pre))
    out = A[L] @ W[L][:, 0] + b[L][0]
    loss = -np.sum(out) * invb
    D = _act_grad(A[L], slope) * (-invb * W[L][:, 0][None, :])
    for l in range(L, 1, -1):
        D = _act_grad(A[l - 1], slope) * (D @ W[l - 1].T)
    dout = D @ W[0][:G].T
    GW, Gb = gen_p['W'], gen_p['b']
    Lg = len(GW) - 1
    gW, gb = [None] * (Lg + 1), [None] * (Lg + 1)
    din0 = None
    for l in range(Lg, -1, -1):
        gW[l] = gacts[l].T @ dout
        gb[l] = dout.sum(0)
        if l > 0:
            dout = _act_grad(gacts[l], slope) * (dout @ GW[l].T)
        else:
            din0 = dout @ GW[0].T
    gemb = _scatter_emb(gen_p, cc, din0, z.shape[1] + nc.shape[1])
    return loss, {'emb': gemb, 'W': gW, 'b': gb}


def flat(grads):
    out = list(grads['emb'])
    for w, b in zip(grads['W'], grads['b']):
        out += [w, b]
    return out
