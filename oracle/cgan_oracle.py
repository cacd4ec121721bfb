"""PyTorch-CPU autograd oracle for the WGAN-GP conditional GAN train step.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

PARITY UNPINNED BY THE REFERENCE: the algorithm of this path executes inside
TensorFlow/Keras 2.x (un-vendored, un-pinned dependency, absent offline; the
reference ships no tests, goldens or loadable checkpoints for it).  This file
restates ``/root/reference/tf_utils.py`` line by line on top of torch-CPU
autograd (an engine independent of the product's hand-derived backward) and
encodes the published Keras semantics the reference relies on:

* ``Dense``: ``y = x @ kernel + bias``, kernel ``[in, out]``, glorot-uniform
  ``U(+-sqrt(6/(in+out)))``, zero bias.  ``Embedding``: table ``[V, e]``,
  ``U(-0.05, 0.05)``.  ``ReLU`` gradient at 0 is 0.
* ``tape.gradient(non-scalar)`` differentiates the sum of the target;
  ``reduce_mean`` averages over all elements.
* Keras-v2 ``RMSprop(lr, rho=0.9, momentum=0, epsilon=1e-7, centered=False)``:
  ``rms = rho*rms + (1-rho)*g*g ; w -= lr * g / (sqrt(rms) + eps)``.
  Embedding gradients arrive as IndexedSlices: optimizer_v2 first sums
  duplicate slices, then the no-momentum sparse path decays the WHOLE ``rms``
  slot by rho (``assign(rms, rms*rho)``), scatter-adds ``(1-rho)*g*g`` on the
  looked-up rows and scatter-adds the update on those rows only.  Rows that
  were not looked up have g = 0, so this is exactly the dense formula applied
  to the scatter-added dense gradient - which is what autograd returns here.
* Keras-v2 ``Adam(lr, 0.9, 0.999, epsilon=1e-7)``:
  ``lr_t = lr*sqrt(1-b2^t)/(1-b1^t) ; w -= lr_t * m / (sqrt(v) + eps)``;
  its sparse path decays m and v densely and updates the variable densely,
  i.e. again the dense formula on the scatter-added gradient.

Functions cite the reference lines they follow.
"""
import math

import numpy as np
import torch


# ---------------------------------------------------------------------------
# parameters
# ---------------------------------------------------------------------------

def emb_dims(vocab_sizes):
    """Rule of thumb ``int(vs ** 0.5) + 1``.  tf_utils.py:65, :121."""
    return [int(vs ** 0.5) + 1 for vs in vocab_sizes]


def init_params(kind, x_dim, vocab_sizes, nb_numeric, h_dims=None, z_dim=10, seed=0):
    """Return ``{'emb': [tables], 'W': [kernels], 'b': [biases]}`` as float64
    NumPy arrays with Keras-default initialisers.

    kind='gen':  layer-1 input is ``[z | num | emb_0 | emb_1 ...]`` and the last
    layer maps to ``x_dim``  (tf_utils.py:56-96).
    kind='disc': layer-1 input is ``[x | num | emb_0 | ...]`` and the last layer
    maps to 1  (tf_utils.py:113-136).
    """
    if h_dims is None:
        h_dims = [256, 256]  # tf_utils.py:79-80, :110-111
    rs = np.random.RandomState(seed)
    e = emb_dims(vocab_sizes)
    emb = [rs.uniform(-0.05, 0.05, size=(v, d)) for v, d in zip(vocab_sizes, e)]
    d_in = (z_dim if kind == 'gen' else x_dim) + nb_numeric + sum(e)
    d_out = x_dim if kind == 'gen' else 1
    dims = [d_in] + list(h_dims) + [d_out]
    W, b = [], []
    for i in range(len(dims) - 1):
        lim = math.sqrt(6.0 / (dims[i] + dims[i + 1]))
        W.append(rs.uniform(-lim, lim, size=(dims[i], dims[i + 1])))
        b.append(np.zeros(dims[i + 1]))
    return {'emb': emb, 'W': W, 'b': b}


def to_torch(params, dtype=torch.float64, requires_grad=False):
    out = {}
    for k in ('emb', 'W', 'b'):
        out[k] = [torch.tensor(np.asarray(a), dtype=dtype).requires_grad_(requires_grad) for a in params[k]]
    return out


def to_numpy(tparams):
    return {k: [a.detach().cpu().numpy().copy() for a in tparams[k]] for k in ('emb', 'W', 'b')}


def param_list(tparams):
    """Keras ``trainable_variables`` order: embeddings first (layers created at
    tf_utils.py:64-70 / :120-125), then kernel, bias per Dense."""
    out = list(tparams['emb'])
    for w, b in zip(tparams['W'], tparams['b']):
        out += [w, b]
    return out


# ---------------------------------------------------------------------------
# models
# ---------------------------------------------------------------------------

def _cov_features(p, cc, nc):
    """concat([num, emb_0(cat[:,0]), emb_1(cat[:,1]) ...]).  tf_utils.py:64-76, :120-131."""
    embs = [p['emb'][i][cc[:, i].long()] for i in range(len(p['emb']))]
    return torch.cat([nc] + embs, dim=-1)


def _mlp(p, h):
    n = len(p['W'])
    for i in range(n):
        h = h @ p['W'][i] + p['b'][i]
        if i < n - 1:
            h = torch.relu(h)  # tf_utils.py:85-88, :132-135 (Dense + ReLU; last Dense linear)
    return h


def gen_forward(p, z, cc, nc):
    """tf_utils.py:45-98: h = concat([z, concat([num, embeddings])])."""
    return _mlp(p, torch.cat([z, _cov_features(p, cc, nc)], dim=-1))


def disc_forward(p, x, cc, nc):
    """tf_utils.py:101-137: h = concat([x, num, embeddings])."""
    return _mlp(p, torch.cat([x, _cov_features(p, cc, nc)], dim=-1))


# ---------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------

def wasserstein_loss(y_true, y_pred):
    """tf_utils.py:140-144."""
    return torch.mean(y_true * y_pred)


def generator_loss(fake_output):
    """tf_utils.py:147-151."""
    return wasserstein_loss(-torch.ones_like(fake_output), fake_output)


def discriminator_loss(real_output, fake_output):
    """tf_utils.py:174-181."""
    return (wasserstein_loss(-torch.ones_like(real_output), real_output)
            + wasserstein_loss(torch.ones_like(fake_output), fake_output))


def gradient_penalty(f, real, fake, alpha):
    """tf_utils.py:154-171 with ``alpha`` injected (TF draws it from Philox,
    which cannot be replayed offline).  ``slopes`` has no epsilon."""
    diff = fake - real
    inter = real + alpha * diff
    if not inter.requires_grad:
        inter = inter.requires_grad_(True)
    pred = f(inter)
    grad = torch.autograd.grad(pred.sum(), inter, create_graph=True)[0]
    slopes = torch.sqrt(torch.sum(grad * grad, dim=1))
    return torch.mean((slopes - 1.0) ** 2)


# ---------------------------------------------------------------------------
# optimizers (Keras-v2 semantics, dense + sparse apply)
# ---------------------------------------------------------------------------

class RMSprop:
    def __init__(self, lr=0.001, rho=0.9, epsilon=1e-7):
        self.lr, self.rho, self.eps = lr, rho, epsilon
        self.state = {}

    def apply(self, params, grads):
        with torch.no_grad():
            for i, (w, g) in enumerate(zip(params, grads)):
                rms = self.state.setdefault(i, torch.zeros_like(w))
                rms.mul_(self.rho).add_((1 - self.rho) * g * g)
                w -= self.lr * g / (torch.sqrt(rms) + self.eps)


class Adam:
    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, epsilon
        self.state = {}
        self.t = 0

    def apply(self, params, grads):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        with torch.no_grad():
            for i, (w, g) in enumerate(zip(params, grads)):
                m, v = self.state.setdefault(i, (torch.zeros_like(w), torch.zeros_like(w)))
                m.mul_(self.b1).add_((1 - self.b1) * g)
                v.mul_(self.b2).add_((1 - self.b2) * g * g)
                w -= lr_t * m / (torch.sqrt(v) + self.eps)


# ---------------------------------------------------------------------------
# train steps
# ---------------------------------------------------------------------------

def disc_loss_and_grads(x, z, cc, nc, gen_p, disc_p, alpha, grad_penalty_weight=10.0,
                        aug_mask=None, aug_noise_fake=None, aug_noise_real=None):
    """Loss + gradients of one critic step, tf_utils.py:186-220.

    aug_*: the reference draws these with NumPy inside ``@tf.function`` so they
    are trace-time constants (tf_utils.py:202, :209-210); inject them here.
    ``aug_noise_fake`` is ONE [G] row broadcast over the batch (:209),
    ``aug_noise_real`` is [B, G] (:210).
    Returns (loss, gp, grads in ``param_list`` order).
    """
    for t in param_list(disc_p):
        t.requires_grad_(True)
    with torch.no_grad():
        x_gen = gen_forward(gen_p, z, cc, nc)  # generator is not differentiated (:206, :220)
    if aug_mask is not None:
        x_gen = x_gen + aug_mask[:, None] * aug_noise_fake[None, :]
        x = x + aug_mask[:, None] * aug_noise_real
    disc_out = disc_forward(disc_p, x_gen, cc, nc)
    disc_real = disc_forward(disc_p, x, cc, nc)
    gp = gradient_penalty(lambda xi: disc_forward(disc_p, xi, cc, nc), x, x_gen, alpha)
    loss = discriminator_loss(disc_real, disc_out) + grad_penalty_weight * gp
    grads = torch.autograd.grad(loss, param_list(disc_p), allow_unused=True)
    grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, param_list(disc_p))]
    return loss.detach(), gp.detach(), grads


def train_disc(x, z, cc, nc, gen_p, disc_p, disc_opt, alpha, grad_penalty_weight=10.0, **aug):
    """tf_utils.py:186-223."""
    loss, gp, grads = disc_loss_and_grads(x, z, cc, nc, gen_p, disc_p, alpha, grad_penalty_weight, **aug)
    disc_opt.apply(param_list(disc_p), grads)
    return loss, gp, grads


def gen_loss_and_grads(z, cc, nc, gen_p, disc_p, aug_mask=None, aug_noise=None):
    """tf_utils.py:226-255; ``aug_noise`` is [B, G] (:247)."""
    for t in param_list(gen_p):
        t.requires_grad_(True)
    x_gen = gen_forward(gen_p, z, cc, nc)
    if aug_mask is not None:
        x_gen = x_gen + aug_mask[:, None] * aug_noise
    disc_out = disc_forward(disc_p, x_gen, cc, nc)
    loss = generator_loss(disc_out)
    grads = torch.autograd.grad(loss, param_list(gen_p), allow_unused=True)
    grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, param_list(gen_p))]
    return loss.detach(), grads


def train_gen(z, cc, nc, gen_p, disc_p, gen_opt, **aug):
    """tf_utils.py:226-258."""
    loss, grads = gen_loss_and_grads(z, cc, nc, gen_p, disc_p, **aug)
    gen_opt.apply(param_list(gen_p), grads)
    return loss, grads


def predict(cc, nc, gen_p, z):
    """tf_utils.py:359-376 with z injected."""
    with torch.no_grad():
        return gen_forward(gen_p, z, cc, nc).numpy()


def train(dataset, cat_covs, num_covs, z_dim, epochs, batch_size, gen_p, disc_p, score_fn, save_fn,
          gen_opt, disc_opt, noise, nb_critic=5, patience=10, log=None):
    """Loop of tf_utils.py:261-356 (logging / checkpoints dropped).  ``noise`` is
    a callable ``noise(kind, shape)`` (kind in 'z', 'alpha') replacing TF's RNG so
    the product can be fed the very same draws.  Returns the trace."""
    best_score = -np.inf
    initial_patience = patience
    trace = {'gen_loss': [], 'disc_loss': [], 'score': [], 'saved': [], 'patience': []}
    dt = gen_p['W'][0].dtype
    for epoch in range(epochs):
        gl, dl = [], []
        for i in range(0, len(dataset), batch_size):
            x = torch.tensor(dataset[i:i + batch_size], dtype=dt)
            cc = torch.tensor(cat_covs[i:i + batch_size])
            nc = torch.tensor(num_covs[i:i + batch_size], dtype=dt)
            bs = x.shape[0]
            disc_loss = None
            for _ in range(nb_critic):
                z = torch.tensor(noise('z', (bs, z_dim)), dtype=dt)
                alpha = torch.tensor(noise('alpha', (bs, 1)), dtype=dt)
                disc_loss, _, _ = train_disc(x, z, cc, nc, gen_p, disc_p, disc_opt, alpha)
            dl.append(float(disc_loss))          # only the LAST critic loss is averaged (:319)
            z = torch.tensor(noise('z', (bs, z_dim)), dtype=dt)
            gen_loss, _ = train_gen(z, cc, nc, gen_p, disc_p, gen_opt)
            gl.append(float(gen_loss))
        if epoch % 5 == 0:                      # :333
            score = score_fn(gen_p)
            if score > best_score:
                save_fn()
                best_score = score
                patience = initial_patience
                trace['saved'].append(epoch)
            else:
                patience -= 1
            trace['score'].append(float(score))
        trace['gen_loss'].append(float(np.mean(np.float32(gl))))
        trace['disc_loss'].append(float(np.mean(np.float32(dl))))
        trace['patience'].append(patience)
        if patience == 0:
            break
    return trace
