"""Shared test helpers: oracle <-> product weight plumbing, seeded synthetic inputs."""
import numpy as np
import torch

from oracle import cgan_oracle as O


def keras_list(p, dtype=np.float32):
    """Oracle param dict -> Keras-order list [emb.., W0, b0, W1, b1, ...]."""
    out = [np.asarray(a, dtype) for a in p['emb']]
    for w, b in zip(p['W'], p['b']):
        out += [np.asarray(w, dtype), np.asarray(b, dtype)]
    return out


def torch_params(p, dtype=torch.float64):
    return O.to_torch(p, dtype=dtype)


def tlist_to_np(tensors):
    return [t.detach().numpy() for t in tensors]


def make_problem(G=37, V=(15, 3), H=(16, 8), Z=6, B=5, n_num=1, seed=0, num_zero=True):
    rs = np.random.RandomState(seed)
    gp = O.init_params('gen', G, list(V), n_num, list(H), Z, seed=seed + 1)
    dp = O.init_params('disc', G, list(V), n_num, list(H), seed=seed + 2)
    # non-zero biases and larger embeddings so that every gradient path is exercised
    for p in (gp, dp):
        p['b'] = [rs.uniform(-0.1, 0.1, size=b.shape) for b in p['b']]
        p['emb'] = [rs.uniform(-0.5, 0.5, size=e.shape) for e in p['emb']]
    x = rs.randn(B, G)
    z = rs.randn(B, Z)
    cc = np.stack([rs.randint(0, v, B) for v in V], 1).astype(np.int32)
    nc = np.zeros((B, n_num)) if num_zero else rs.randn(B, n_num)
    alpha = rs.rand(B, 1)
    return dict(G=G, V=list(V), H=list(H), Z=Z, B=B, n_num=n_num, gp=gp, dp=dp, x=x, z=z, cc=cc, nc=nc, alpha=alpha)


def t64(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


def tl(a):
    return torch.tensor(np.asarray(a), dtype=torch.int64)


def lowrank(rs, n, g, k=4, noise=0.5):
    return (rs.randn(n, k) @ rs.randn(k, g) + noise * rs.randn(n, g)).astype(np.float32)


def max_rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-30, np.max(np.abs(b))))
