"""Torch-CPU autograd oracle of the OPTIONAL batch-norm generator (BASELINE.json north_star: "dense layers fused
with bias, LeakyReLU and batch-norm").

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  NO REFERENCE COUNTERPART: the reference's generator is
Dense + ReLU (``/root/reference/tf_utils.py:85-88``), so this option is off on every parity path and this oracle
pins nothing against the reference - it restates Keras ``BatchNormalization()`` (epsilon 1e-3, momentum 0.99,
gamma 1 / beta 0 / moving mean 0 / moving variance 1) placed after every hidden Dense and before the activation,
with the ``training`` flags the reference passes to its models:

* ``train_gen`` calls ``gen(..., training=True)`` (tf_utils.py:244): batch mean and BIASED batch variance over the
  batch rows normalise the layer, both are differentiated through (autograd here, a hand-derived kernel in the
  product), and the moving statistics take ``m = momentum * m + (1 - momentum) * batch`` (the non-fused path Keras
  uses for 2-D inputs: no Bessel correction);
* ``train_disc`` calls ``gen(..., training=False)`` (tf_utils.py:206) and ``predict`` defaults to it (:359-375):
  the moving statistics normalise, nothing is updated.

The critic has no batch-norm (its gradient penalty is a per-sample quantity, tf_utils.py:154-171).
"""
import numpy as np
import torch

from . import cgan_oracle as O


def init_bn(h_dims, seed=None):
    """Keras initial values; with a seed: random ones, so that every term of the backward is exercised."""
    rs = np.random.RandomState(seed) if seed is not None else None
    out = []
    for n in h_dims:
        if rs is None:
            out.append(dict(gamma=np.ones(n), beta=np.zeros(n), mean=np.zeros(n), var=np.ones(n)))
        else:
            out.append(dict(gamma=rs.uniform(0.5, 1.5, n), beta=rs.uniform(-0.3, 0.3, n), mean=rs.uniform(-0.2, 0.2, n),
                            var=rs.uniform(0.5, 2.0, n)))
    return out


def bn_to_torch(bn, dtype=torch.float64):
    return [{k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in layer.items()} for layer in bn]


def gen_forward(p, bn, z, cc, nc, training, eps=1e-3, momentum=0.99, update=True):
    h = torch.cat([z, O._cov_features(p, cc, nc)], dim=-1)
    n = len(p['W'])
    for i in range(n):
        h = h @ p['W'][i] + p['b'][i]
        if i < n - 1:
            L = bn[i]
            if training:
                mu = h.mean(0)
                var = ((h - mu) ** 2).mean(0)           # biased
                if update:
                    with torch.no_grad():
                        L['mean'].mul_(momentum).add_((1 - momentum) * mu.detach())
                        L['var'].mul_(momentum).add_((1 - momentum) * var.detach())
            else:
                mu, var = L['mean'], L['var']
            h = L['gamma'] * (h - mu) / torch.sqrt(var + eps) + L['beta']
            h = torch.relu(h)
    return h


def gen_param_list(p, bn):
    """Keras trainable_variables order: embeddings, then per layer kernel, bias (, gamma, beta)."""
    out = list(p['emb'])
    for i, (w, b) in enumerate(zip(p['W'], p['b'])):
        out += [w, b]
        if i < len(bn):
            out += [bn[i]['gamma'], bn[i]['beta']]
    return out


def disc_loss_and_grads(x, z, cc, nc, gen_p, bn, disc_p, alpha, grad_penalty_weight=10.0):
    """tf_utils.py:186-220 with the batch-norm generator in inference mode (:206)."""
    for t in O.param_list(disc_p):
        t.requires_grad_(True)
    with torch.no_grad():
        x_gen = gen_forward(gen_p, bn, z, cc, nc, training=False)
    disc_out = O.disc_forward(disc_p, x_gen, cc, nc)
    disc_real = O.disc_forward(disc_p, x, cc, nc)
    gp = O.gradient_penalty(lambda xi: O.disc_forward(disc_p, xi, cc, nc), x, x_gen, alpha)
    loss = O.discriminator_loss(disc_real, disc_out) + grad_penalty_weight * gp
    grads = torch.autograd.grad(loss, O.param_list(disc_p), allow_unused=True)
    grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, O.param_list(disc_p))]
    return loss.detach(), grads


def gen_loss_and_grads(z, cc, nc, gen_p, bn, disc_p, update=True):
    """tf_utils.py:226-255 with the generator in training mode (:244)."""
    params = gen_param_list(gen_p, bn)
    for t in params:
        t.requires_grad_(True)
    x_gen = gen_forward(gen_p, bn, z, cc, nc, training=True, update=update)
    loss = O.generator_loss(O.disc_forward(disc_p, x_gen, cc, nc))
    grads = torch.autograd.grad(loss, params, allow_unused=True)
    grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, params)]
    return loss.detach(), grads


def train_disc(x, z, cc, nc, gen_p, bn, disc_p, disc_opt, alpha):
    loss, grads = disc_loss_and_grads(x, z, cc, nc, gen_p, bn, disc_p, alpha)
    disc_opt.apply(O.param_list(disc_p), grads)
    return loss


def train_gen(z, cc, nc, gen_p, bn, disc_p, gen_opt):
    loss, grads = gen_loss_and_grads(z, cc, nc, gen_p, bn, disc_p)
    gen_opt.apply(gen_param_list(gen_p, bn), grads)
    return loss
