"""Drop-in mirror of the reference ``tf_utils.py`` (WGAN-GP conditional GAN).

Same public names, argument order, defaults and return conventions as
``/root/reference/tf_utils.py``; every numeric operation runs in hand-written
HIP kernels of libagx.so (include/agx.h) on an MI355X.  There is no TensorFlow,
no autodiff engine and no CPU fallback: the backward pass (including the
gradient penalty's double-backward) is derived by hand in csrc/cgan.hip.

Differences a caller can see (all additive):
  * optimizers are ``RMSprop`` / ``Adam`` from this module (Keras-v2 formulas)
    instead of ``tfk.optimizers.*``;
  * ``train_disc`` / ``train_gen`` / ``train`` / ``predict`` take optional
    keyword-only ``alpha=``, ``aug=`` (and ``z=`` where the reference already
    has it) to inject the random draws TF would make, which is how parity
    tests feed the oracle and this path identical inputs;
  * TensorBoard summaries and ``tf.train.Checkpoint`` files are not written
    (``log_dir`` / ``checkpoint_dir`` are accepted and ignored); ``save_fn`` and
    ``gen.save`` work (own ``.npz`` format);
  * under ``torch.distributed`` (one process per GPU) the same functions run
    batch data-parallel: ``batch_size`` stays the GLOBAL batch, every rank
    takes its shard and gradients are summed with one RCCL all-reduce per
    optimizer step.
"""
import ast
import math
import os
import weakref
from types import SimpleNamespace

import numpy as np
import torch

from . import _dev, _lib, ops
from ._dev import DeviceArray, ptr

tfk = SimpleNamespace()  # populated below with .optimizers.{RMSprop, Adam}

CHECKPOINTS_DIR = '../checkpoints/'
_LOSS_SLOTS = 64  # floats appended to every flat gradient buffer (loss scalars ride the gradient exchange)
_EXCH_SLACK = 1024  # further floats so that [0, n_flat + slots) rounds up to a multiple of 4 * world (world <= 256)


def limit_gpu(gpu_idx=0, mem=2 * 1024):
    """Reference ``tf_utils.py:17-38`` pins one GPU and caps TF's allocator.
    Here it only selects the HIP device (no allocator to cap)."""
    if torch.cuda.is_available() and gpu_idx < torch.cuda.device_count():
        torch.cuda.set_device(gpu_idx)
        print(torch.cuda.device_count(), "Physical GPUs,", 1, "Logical GPUs")


# ---------------------------------------------------------------------------
# random draws (replace tf.random.normal / tf.random.uniform)
# ---------------------------------------------------------------------------

class _Philox:
    def __init__(self, seed=0):
        self.seed, self.offset = int(seed), 0

    def draw(self, shape, normal):
        lib = _lib.load()
        n = int(np.prod(shape))
        out = _dev.empty(shape)
        fn = lib.agx_rng_normal if normal else lib.agx_rng_uniform
        _lib.check(fn(_dev.stream(), ptr(out), n, self.seed, self.offset), 'agx_rng')
        self.offset += (n + 3) & ~3
        return out


_rng = _Philox(0)


def set_seed(seed):
    """Seed the device Philox stream used for z and alpha."""
    global _rng
    _rng = _Philox(seed)


def random_normal(shape):
    return _rng.draw(tuple(shape), True)


def random_uniform(shape):
    return _rng.draw(tuple(shape), False)


# ---------------------------------------------------------------------------
# models
# ---------------------------------------------------------------------------

def _emb_dims(vocab_sizes):
    return [int(vs ** 0.5) + 1 for vs in vocab_sizes]  # tf_utils.py:65, :121


class _Net:
    """MLP with categorical-covariate embeddings whose parameters live in ONE
    flat HBM buffer (Keras ``trainable_variables`` order: embedding tables,
    then kernel/bias per Dense; kernels ``[in, out]`` with a 16-B aligned
    leading dimension)."""

    kind = None

    def __init__(self, x_dim, vocab_sizes, nb_numeric, h_dims, z_dim, slope=0.0, weights=None, seed=None,
                 batch_norm=False, bn_epsilon=1e-3, bn_momentum=0.99):
        if h_dims is None:
            h_dims = [256, 256]
        # optional BatchNormalization after every hidden Dense, before the activation (BASELINE.json north_star; the
        # reference has none, tf_utils.py:85-88, so this is OFF on every parity path).  Generator only - the critic's
        # gradient penalty is a per-sample quantity that batch statistics would couple (tf_utils.py:154-171) - and
        # the eager step functions only (train() falls back to them).  Keras semantics: train_gen normalises with
        # batch statistics and updates the moving ones (training=True, :244), train_disc and predict use the moving
        # statistics (training=False, :206, :372-375).
        self.batch_norm = bool(batch_norm)
        self.bn_epsilon, self.bn_momentum = float(bn_epsilon), float(bn_momentum)
        if self.batch_norm and self.kind != 'gen':
            raise NotImplementedError('batch_norm is available for the generator only: the WGAN-GP critic is penalised per '
                                      'sample (tf_utils.py:154-171), batch statistics would couple the samples')
        if len(h_dims) < 1 or len(h_dims) + 1 > _lib.AGX_MAX_LAYERS:
            raise ValueError('h_dims must have between 1 and %d entries' % (_lib.AGX_MAX_LAYERS - 1))
        if len(vocab_sizes) > _lib.AGX_MAX_CAT:
            raise ValueError('at most %d categorical covariates' % _lib.AGX_MAX_CAT)
        self.x_dim, self.z_dim = int(x_dim), int(z_dim)
        self.vocab_sizes = [int(v) for v in vocab_sizes]
        self.nb_numeric = int(nb_numeric)
        self.h_dims = [int(h) for h in h_dims]
        self.slope = float(slope)
        self.emb_dims = _emb_dims(self.vocab_sizes)
        lead = self.z_dim if self.kind == 'gen' else self.x_dim
        out = self.x_dim if self.kind == 'gen' else 1
        self.dims = [lead + self.nb_numeric + sum(self.emb_dims)] + self.h_dims + [out]
        self._layout()
        # flat buffers; the tails (loss slots + slack) exist for the data-parallel exchange (_dp_update)
        self._pbuf = _dev.zeros(self.n_flat + _LOSS_SLOTS + _EXCH_SLACK)
        self._gbuf = _dev.zeros(self.n_flat + _LOSS_SLOTS + _EXCH_SLACK)
        self.params = self._pbuf[:self.n_flat]
        self.grads = self._gbuf[:self.n_flat + _LOSS_SLOTS]
        self._shards = {}
        self.set_weights(weights if weights is not None else self._initial_weights(seed))
        self._pair = {}
        # Keras-like introspection used by the reference (tf_utils.py:201, :371)
        ins = [SimpleNamespace(shape=(None, lead)), SimpleNamespace(shape=(None, len(self.vocab_sizes))),
               SimpleNamespace(shape=(None, self.nb_numeric))]
        self.input = ins
        self.output = SimpleNamespace(shape=(None, out))

    # -- flat layout ---------------------------------------------------------
    def _layout(self):
        off = 0
        a64 = lambda v: (v + 63) & ~63
        self.emb_off = []
        for v, e in zip(self.vocab_sizes, self.emb_dims):
            self.emb_off.append(off)
            off = a64(off + v * e)
        self.ldw, self.w_off, self.b_off = [], [], []
        self.bn_off = []      # per hidden layer: flat offsets of (gamma, beta, moving_mean, moving_variance)
        for l in range(len(self.dims) - 1):
            n = self.dims[l + 1]
            ld = 1 if n == 1 else _dev.r4(n)
            self.ldw.append(ld)
            self.w_off.append(off)
            off = a64(off + self.dims[l] * ld)
            self.b_off.append(off)
            off = a64(off + n)
            if self.batch_norm and l < len(self.dims) - 2:
                four = []
                for _ in range(4):
                    four.append(off)
                    off = a64(off + n)
                self.bn_off.append(tuple(four))
        self.n_flat = off
        d = _lib.MlpDesc()
        d.n_layers = len(self.dims) - 1
        for i, v in enumerate(self.dims):
            d.dims[i] = v
        for l in range(d.n_layers):
            d.ldw[l], d.w_off[l], d.b_off[l] = self.ldw[l], self.w_off[l], self.b_off[l]
        d.emb.n_cat = len(self.vocab_sizes)
        for i in range(d.emb.n_cat):
            d.emb.vocab[i], d.emb.edim[i], d.emb.table_off[i] = self.vocab_sizes[i], self.emb_dims[i], self.emb_off[i]
        d.n_num = self.nb_numeric
        d.n_flat = self.n_flat
        d.bn = int(self.batch_norm)
        d.bn_eps, d.bn_momentum = self.bn_epsilon, self.bn_momentum
        for l, four in enumerate(self.bn_off):
            d.bn_gamma_off[l], d.bn_beta_off[l], d.bn_mean_off[l], d.bn_var_off[l] = four
        self.desc = d

    def _initial_weights(self, seed):
        """Keras defaults: Embedding U(-0.05, 0.05); Dense glorot-uniform, zero bias."""
        rs = np.random if seed is None else np.random.RandomState(seed)
        ws = [rs.uniform(-0.05, 0.05, size=(v, e)).astype(np.float32)
              for v, e in zip(self.vocab_sizes, self.emb_dims)]
        for l in range(len(self.dims) - 1):
            lim = math.sqrt(6.0 / (self.dims[l] + self.dims[l + 1]))
            ws.append(rs.uniform(-lim, lim, size=(self.dims[l], self.dims[l + 1])).astype(np.float32))
            ws.append(np.zeros(self.dims[l + 1], np.float32))
            if self.batch_norm and l < len(self.dims) - 2:     # Keras BatchNormalization: gamma 1, beta 0, moving 0 / 1
                n = self.dims[l + 1]
                ws += [np.ones(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32), np.ones(n, np.float32)]
        return ws

    def _segments(self):
        """(flat offset, rows, cols, ld) per variable in Keras `get_weights()` order (a batch-norm layer contributes
        gamma, beta, moving_mean, moving_variance behind its Dense)."""
        segs = [(o, v, e, e) for o, v, e in zip(self.emb_off, self.vocab_sizes, self.emb_dims)]
        self._vector_seg, self._frozen_seg = set(), set()
        for l in range(len(self.dims) - 1):
            segs.append((self.w_off[l], self.dims[l], self.dims[l + 1], self.ldw[l]))
            self._vector_seg.add(len(segs))
            segs.append((self.b_off[l], 1, self.dims[l + 1], self.dims[l + 1]))
            if self.batch_norm and l < len(self.bn_off):
                for q, o in enumerate(self.bn_off[l]):
                    self._vector_seg.add(len(segs))
                    if q >= 2:
                        self._frozen_seg.add(len(segs))      # moving statistics: not trainable
                    segs.append((o, 1, self.dims[l + 1], self.dims[l + 1]))
        return segs

    def _unflatten(self, flat_host):
        out = []
        for k, (o, r, c, ld) in enumerate(self._segments()):
            a = flat_host[o:o + r * ld].reshape(r, ld)[:, :c]
            out.append(a.reshape(-1).copy() if k in self._vector_seg else a.copy())
        return out

    def _flatten(self, arrays):
        segs = self._segments()
        if len(arrays) != len(segs):
            raise ValueError('expected %d weight arrays, got %d' % (len(segs), len(arrays)))
        flat = np.zeros(self.n_flat, np.float32)
        for a, (o, r, c, ld) in zip(arrays, segs):
            a = np.asarray(a, np.float32).reshape(r, c)
            flat[o:o + r * ld].reshape(r, ld)[:, :c] = a
        return flat

    def get_weights(self):
        return self._unflatten(self.params.cpu().numpy())

    def set_weights(self, arrays):
        self.params.copy_(torch.from_numpy(self._flatten(arrays)))

    def get_gradients(self):
        """Last computed gradients, Keras variable order (host copies)."""
        return self._unflatten(self.grads[:self.n_flat].cpu().numpy())

    @property
    def trainable_variables(self):
        segs = self._segments()
        return [DeviceArray(self.params[o:o + r * ld].view(r, ld)[:, :c]) for k, (o, r, c, ld) in enumerate(segs)
                if k not in self._frozen_seg]

    def count_params(self):
        return sum(r * c for _, r, c, _ in self._segments())

    def summary(self):
        print('Model: "%s"' % ('generator' if self.kind == 'gen' else 'critic'))
        names = ['embedding_%d' % i for i in range(len(self.vocab_sizes))]
        for l in range(len(self.dims) - 1):
            names += ['dense_%d/kernel' % l, 'dense_%d/bias' % l]
            if self.batch_norm and l < len(self.bn_off):
                names += ['batch_normalization_%d/%s' % (l, q) for q in ('gamma', 'beta', 'moving_mean', 'moving_variance')]
        for n, (_, r, c, _) in zip(names, self._segments()):
            print('  %-22s %-16s %d' % (n, (r, c) if r > 1 or 'kernel' in n or 'emb' in n else (c,), r * c))
        print('Total params: {:,}'.format(self.count_params()))

    def save(self, path):
        """Own ``.npz`` format (the reference writes Keras H5, gtex_tcga_gan.py:112)."""
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        cfg = dict(kind=self.kind, x_dim=self.x_dim, z_dim=self.z_dim, vocab_sizes=self.vocab_sizes,
                   nb_numeric=self.nb_numeric, h_dims=self.h_dims, slope=self.slope)
        if self.batch_norm:
            cfg.update(batch_norm=True, bn_epsilon=self.bn_epsilon, bn_momentum=self.bn_momentum)
        np.savez(path, __config__=np.array(repr(cfg)), **{'w%03d' % i: w for i, w in enumerate(self.get_weights())})

    # -- forward (op-level ABI) ---------------------------------------------
    def _embed_concat(self, lead, cc, nc, rows):
        lib = _lib.load()
        width = self.dims[0]
        ld = _dev.r4(width)
        out = _dev.zeros((rows, ld))
        n_lead = lead.shape[1]
        _lib.check(lib.agx_embed_concat_fwd(_dev.stream(), self.desc.emb, ptr(self.params), ptr(lead), n_lead,
                                            lead.stride(0), ptr(nc), self.nb_numeric, self.nb_numeric, ptr(cc),
                                            len(self.vocab_sizes), rows, 0, ptr(out), ld), 'agx_embed_concat_fwd')
        return out, ld

    def _forward(self, lead, cc, nc, keep=False):
        rows = lead.shape[0]
        h, _ = self._embed_concat(lead, cc, nc, rows)
        acts = [h]
        nl = len(self.dims) - 1
        for l in range(nl):
            n, k = self.dims[l + 1], self.dims[l]
            w = self.params[self.w_off[l]:self.w_off[l] + k * self.ldw[l]].view(k, self.ldw[l])
            b = self.params[self.b_off[l]:self.b_off[l] + n]
            if self.batch_norm and l < nl - 1:       # moving statistics (training=False)
                ga, be, mm, mv = (self.params[o:o + n] for o in self.bn_off[l])
                h = ops.dense_bn(h, w, b, rows, n, k, gamma=ga, beta=be, eps=self.bn_epsilon, slope=self.slope,
                                 running=(mm, mv))[0]
            else:
                h = ops.gemm(h, w, rows, n, k, bias=b, act=1 if l < nl - 1 else 0, slope=self.slope)
            acts.append(h)
        out = h[:, :self.dims[-1]]
        return (out, acts) if keep else out

    def __call__(self, inputs, training=False):
        lead, cc, nc = inputs
        if isinstance(lead, _CriticProbe):
            return _ProbeResult(self, cc, nc)
        lead = _dev.to_dev(lead)
        cc = _dev.to_dev(cc, torch.int32)
        nc = _dev.to_dev(nc)
        if lead.dim() != 2 or lead.shape[1] != self.dims[0] - self.nb_numeric - sum(self.emb_dims):
            raise ValueError('first input must be [batch, %d]' % (self.dims[0] - self.nb_numeric - sum(self.emb_dims)))
        return DeviceArray(self._forward(lead, cc, nc))


class Generator(_Net):
    kind = 'gen'


class Critic(_Net):
    kind = 'disc'


def make_generator(x_dim, vocab_sizes, nb_numeric, h_dims=None, z_dim=10, **kw):
    """Reference ``tf_utils.py:45-98`` (layer-1 input ``[z | num | emb_0 | emb_1 ...]``)."""
    model = Generator(x_dim, vocab_sizes, nb_numeric, h_dims, z_dim, **kw)
    model.summary()  # tf_utils.py:97
    return model


def make_discriminator(x_dim, vocab_sizes, nb_numeric, h_dims=None, **kw):
    """Reference ``tf_utils.py:101-137`` (layer-1 input ``[x | num | emb_0 | ...]``; linear Dense(1))."""
    return Critic(x_dim, vocab_sizes, nb_numeric, h_dims, z_dim=1, **kw)


def load_model(path):
    if not os.path.exists(path) and os.path.exists(path + '.npz'):
        path = path + '.npz'
    with np.load(path, allow_pickle=False) as f:
        cfg = ast.literal_eval(str(f['__config__']))   # literal dict written by save(); never evaluated as code
        ws = [f['w%03d' % i] for i in range(len(f.files) - 1)]
    cfg = _check_model_config(cfg)
    cls = Generator if cfg['kind'] == 'gen' else Critic
    return cls(cfg['x_dim'], cfg['vocab_sizes'], cfg['nb_numeric'], cfg['h_dims'], cfg['z_dim'],
               slope=cfg['slope'], weights=ws, **{k: cfg[k] for k in ('batch_norm', 'bn_epsilon', 'bn_momentum') if k in cfg})


def _check_model_config(cfg):
    """Keys and types of a saved model's config (a file is untrusted input)."""
    if not isinstance(cfg, dict):
        raise ValueError('model file: __config__ is not a dict')
    isint = lambda v: isinstance(v, int) and not isinstance(v, bool)
    intlist = lambda v: isinstance(v, (list, tuple)) and all(isint(a) and a > 0 for a in v)
    ok = (cfg.get('kind') in ('gen', 'disc') and isint(cfg.get('x_dim')) and cfg['x_dim'] > 0
          and isint(cfg.get('z_dim')) and cfg['z_dim'] > 0 and isint(cfg.get('nb_numeric')) and cfg['nb_numeric'] >= 0
          and intlist(cfg.get('vocab_sizes')) and intlist(cfg.get('h_dims')) and len(cfg['h_dims']) >= 1
          and isinstance(cfg.get('slope', 0.0), (int, float)))
    if not ok:
        raise ValueError('model file: malformed __config__ %r' % (cfg,))
    bn = {}
    if cfg.get('batch_norm', False):
        if cfg['batch_norm'] is not True or not all(isinstance(cfg.get(k), float) and cfg[k] > 0 for k in ('bn_epsilon', 'bn_momentum')):
            raise ValueError('model file: malformed batch-norm entries in __config__ %r' % (cfg,))
        bn = dict(batch_norm=True, bn_epsilon=cfg['bn_epsilon'], bn_momentum=cfg['bn_momentum'])
    return dict(kind=cfg['kind'], x_dim=cfg['x_dim'], z_dim=cfg['z_dim'], nb_numeric=cfg['nb_numeric'],
                vocab_sizes=list(cfg['vocab_sizes']), h_dims=list(cfg['h_dims']), slope=float(cfg.get('slope', 0.0)), **bn)


# ---------------------------------------------------------------------------
# losses (reference surface; tf_utils.py:140-181)
# ---------------------------------------------------------------------------

def _flat_dev(a):
    return _dev.to_dev(a).reshape(-1)


def wasserstein_loss(y_true, y_pred):
    """``mean(y_true * y_pred)``  (tf_utils.py:140-144)."""
    lib = _lib.load()
    a, b = _flat_dev(y_true), _flat_dev(y_pred)
    sums = _dev.zeros(8, dtype=torch.float64)
    ws = _dev.empty(int(lib.agx_sums_ws_doubles()), dtype=torch.float64)
    _lib.check(lib.agx_pearson_sums(_dev.stream(), ptr(a), ptr(b), a.numel(), 0, ptr(sums), ptr(ws)),
               'agx_pearson_sums')
    return np.float32(sums[4].item() / a.numel())


def _loss_kernel(out, n0, n1, slopes=None, gp_weight=0.0):
    lib = _lib.load()
    losses = _dev.empty(4)
    n = max(n0, n1, 1)
    _lib.check(lib.agx_wgan_loss(_dev.stream(), ptr(out), n0, n1, ptr(slopes), 0 if slopes is None else slopes.numel(),
                                 1.0 / n, gp_weight, 1.0, ptr(losses)), 'agx_wgan_loss')
    return losses


def generator_loss(fake_output):
    """``mean(-fake_output)``  (tf_utils.py:147-151)."""
    f = _flat_dev(fake_output)
    return DeviceArray(_loss_kernel(f, f.numel(), 0)[0:1])


def discriminator_loss(real_output, fake_output):
    """``mean(-real) + mean(fake)``  (tf_utils.py:174-181)."""
    r, f = _flat_dev(real_output), _flat_dev(fake_output)
    if r.numel() != f.numel():
        return np.float32(float(generator_loss(r)) - float(generator_loss(f)))
    return DeviceArray(_loss_kernel(torch.cat([r, f]), r.numel(), f.numel())[0:1])


class _CriticProbe:
    """Stand-in for ``inter`` handed to the user's ``f`` so that the reference call
    pattern ``gradient_penalty(lambda x: disc([x, cc, nc], training=True), x, x_gen)``
    (tf_utils.py:218) can be resolved to (critic, cc, nc) without tracing Python."""


class _ProbeResult:
    def __init__(self, net, cc, nc):
        self.net, self.cc, self.nc = net, cc, nc


def _critic_input_gradient(disc, x, cc, nc):
    """d sum(D(x)) / d x for a ReLU critic, hand-derived chain on the op-level ABI."""
    lib = _lib.load()
    rows = x.shape[0]
    _, acts = disc._forward(x, cc, nc, keep=True)
    L = len(disc.dims) - 2
    a = acts[L]
    d = _dev.empty(a.shape)
    w_out = disc.params.data_ptr() + 4 * disc.w_off[L]
    _lib.check(lib.agx_delta_init(_dev.stream(), ptr(a), a.stride(0), w_out, ptr(d), d.stride(0), rows,
                                  disc.dims[L], 0, 0, 0.0, 0.0, 1.0, disc.slope), 'agx_delta_init')
    for l in range(L, 0, -1):
        n = disc.dims[l - 1] if l > 1 else disc.x_dim
        k = disc.dims[l]
        w = disc.params[disc.w_off[l - 1]:disc.w_off[l - 1] + disc.dims[l - 1] * disc.ldw[l - 1]].view(
            disc.dims[l - 1], disc.ldw[l - 1])
        d = ops.gemm(d, w, rows, n, k, tb=1, slope=disc.slope, mask=acts[l - 1] if l > 1 else None)
    return d  # [rows, r4(x_dim)]


def gradient_penalty(f, real_output, fake_output, alpha=None):
    """Gradient penalty of WGAN-GP (tf_utils.py:154-171).  ``f`` must be a closure over
    a critic built by ``make_discriminator`` as in tf_utils.py:218."""
    lib = _lib.load()
    res = f(_CriticProbe())
    if not isinstance(res, _ProbeResult):
        raise TypeError('gradient_penalty: f must call a critic from make_discriminator on its argument')
    disc = res.net
    real, fake = _dev.to_dev(real_output), _dev.to_dev(fake_output)
    B, G = real.shape
    a = random_uniform((B,)) if alpha is None else _dev.to_dev(alpha).reshape(-1)
    inter = _dev.empty((B, G))
    _lib.check(lib.agx_gp_interp(_dev.stream(), ptr(real), real.stride(0), ptr(fake), fake.stride(0), ptr(a),
                                 ptr(inter), G, B, G), 'agx_gp_interp')
    g = _critic_input_gradient(disc, inter, _dev.to_dev(res.cc, torch.int32), _dev.to_dev(res.nc))
    slopes = _dev.empty(B)
    _lib.check(lib.agx_gp_rownorm_scale(_dev.stream(), ptr(g), g.stride(0), ptr(slopes), B, G, 0.0),
               'agx_gp_rownorm_scale')
    out = _dev.zeros(1)
    losses = _dev.empty(4)
    _lib.check(lib.agx_wgan_loss(_dev.stream(), ptr(out), 0, 0, ptr(slopes), B, 1.0 / B, 1.0, 1.0, ptr(losses)),
               'agx_wgan_loss')
    return DeviceArray(losses[1:2])


# ---------------------------------------------------------------------------
# optimizers (Keras-v2 semantics; tf_utils.py:285-288)
# ---------------------------------------------------------------------------

class RMSprop:
    """``rms = rho*rms + (1-rho) g^2 ; w -= lr * g / (sqrt(rms) + eps)`` (epsilon outside the root)."""

    def __init__(self, learning_rate=0.001, rho=0.9, momentum=0.0, epsilon=1e-7, centered=False, **kw):
        if momentum != 0.0 or centered:
            raise NotImplementedError('the reference path uses plain RMSprop (momentum=0, centered=False)')
        self.learning_rate, self.rho, self.epsilon = float(kw.get('lr', learning_rate)), float(rho), float(epsilon)
        self._state = {}
        self.iterations = 0

    def slot(self, net):
        ent = self._state.get(id(net))
        if ent is None or ent[0]() is not net:     # id() of a dead net may be reused: never hand its slot to another net
            ent = self._state[id(net)] = (weakref.ref(net), _dev.zeros(net.n_flat))
        return ent[1]

    def apply(self, net, grad_scale=1.0):
        lib = _lib.load()
        rms = self.slot(net)
        _lib.check(lib.agx_opt_rmsprop(_dev.stream(), ptr(net.params), ptr(net.grads), ptr(rms), net.n_flat,
                                       self.learning_rate, self.rho, self.epsilon, grad_scale), 'agx_opt_rmsprop')
        self.iterations += 1

    def apply_range(self, net, lo, n, g, advance=True):
        """Update parameters [lo, lo + n) from the gradient slice g (a rank's shard of a data-parallel step;
        advance=False: a further piece of a step that has already been counted)."""
        lib = _lib.load()
        rms = self.slot(net)
        _lib.check(lib.agx_opt_rmsprop(_dev.stream(), net.params.data_ptr() + 4 * lo, ptr(g), rms.data_ptr() + 4 * lo, n,
                                       self.learning_rate, self.rho, self.epsilon, 1.0), 'agx_opt_rmsprop')
        self.iterations += int(bool(advance))

    def state_arrays(self, net):
        """Optimizer slots in Keras variable order.  Under torch.distributed this is a COLLECTIVE (every rank
        holds only its own shares of the slots): call it on every rank, e.g. from an all-rank save hook."""
        return {'rms': net._unflatten(_gathered_slot(net, self.slot(net)).cpu().numpy())}


class Adam:
    """Keras-v2 Adam: ``lr_t = lr*sqrt(1-b2^t)/(1-b1^t) ; w -= lr_t * m / (sqrt(v) + eps)``."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False, **kw):
        if amsgrad:
            raise NotImplementedError('amsgrad')
        self.learning_rate = float(kw.get('lr', learning_rate))
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self._state = {}
        self.iterations = 0

    def slot(self, net):
        ent = self._state.get(id(net))
        if ent is None or ent[0]() is not net:
            ent = self._state[id(net)] = (weakref.ref(net), [_dev.zeros(net.n_flat), _dev.zeros(net.n_flat), 0])
        return ent[1]

    def lr_t(self, t):
        return self.learning_rate * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)

    def apply(self, net, grad_scale=1.0):
        lib = _lib.load()
        st = self.slot(net)
        st[2] += 1
        lr_t = self.lr_t(st[2])
        _lib.check(lib.agx_opt_adam(_dev.stream(), ptr(net.params), ptr(net.grads), ptr(st[0]), ptr(st[1]), net.n_flat,
                                    lr_t, self.beta_1, self.beta_2, self.epsilon, grad_scale), 'agx_opt_adam')
        self.iterations += 1

    def apply_range(self, net, lo, n, g, advance=True):
        lib = _lib.load()
        st = self.slot(net)
        st[2] += int(bool(advance))
        _lib.check(lib.agx_opt_adam(_dev.stream(), net.params.data_ptr() + 4 * lo, ptr(g), st[0].data_ptr() + 4 * lo,
                                    st[1].data_ptr() + 4 * lo, n, self.lr_t(st[2]), self.beta_1, self.beta_2, self.epsilon,
                                    1.0), 'agx_opt_adam')
        self.iterations += int(bool(advance))

    def state_arrays(self, net):
        """As RMSprop.state_arrays: a COLLECTIVE under torch.distributed."""
        st = self.slot(net)
        return {'m': net._unflatten(_gathered_slot(net, st[0]).cpu().numpy()),
                'v': net._unflatten(_gathered_slot(net, st[1]).cpu().numpy())}


tfk.optimizers = SimpleNamespace(RMSprop=RMSprop, Adam=Adam)


# ---------------------------------------------------------------------------
# train steps (tf_utils.py:186-258)
# ---------------------------------------------------------------------------

class _Pair:
    """C-side step context of one (generator, critic) pair: descriptor + workspace."""
    MAX_GRAPHS = 16

    def __init__(self, gen, disc, max_batch, max_steps=6):
        lib = _lib.load()
        if gen.x_dim != disc.x_dim or gen.vocab_sizes != disc.vocab_sizes or gen.nb_numeric != disc.nb_numeric:
            raise ValueError('generator and critic were built for different data / covariates')
        if gen.slope != disc.slope:
            raise ValueError('generator and critic must use the same activation slope')
        cfg = _lib.CganConfig()
        cfg.x_dim, cfg.z_dim = gen.x_dim, gen.z_dim
        cfg.gen, cfg.disc = gen.desc, disc.desc
        cfg.slope = gen.slope
        cfg.max_batch = max_batch
        cfg.max_steps = max_steps
        self.max_batch, self.max_steps = max_batch, max_steps
        self.handle = _lib.C.c_void_p()
        _lib.check(lib.agx_cgan_create(cfg, _lib.C.byref(self.handle)), 'agx_cgan_create')
        nbytes = int(lib.agx_cgan_workspace_bytes(self.handle))
        self.workspace = _dev.zeros(nbytes // 4 + 64)
        _lib.check(lib.agx_cgan_bind(self.handle, ptr(gen.params), ptr(gen.grads), ptr(disc.params), ptr(disc.grads),
                                     ptr(self.workspace), self.workspace.numel() * 4), 'agx_cgan_bind')

        self._graphs = {}
        self._keep = {}

    @staticmethod
    def _iter_args(gen, disc, gen_opt, disc_opt, B, nb_critic, gp_weight, seed, aug_d, aug_g, Bg, row_lo, split):
        a = _lib.IterArgs()
        a.B, a.nb_critic, a.gp_weight = B, nb_critic, gp_weight
        a.lr_disc, a.lr_gen = disc_opt.learning_rate, gen_opt.learning_rate
        a.eps = disc_opt.epsilon
        if int(split) >= 2:
            pass     # no optimizer launch inside the segments: the slots stay with the caller (_dp_update / the C driver)
        elif type(disc_opt) is Adam:
            a.opt_kind, a.beta1, a.beta2 = 1, disc_opt.beta_1, disc_opt.beta_2
            sd, sg = disc_opt.slot(disc), gen_opt.slot(gen)
            a.disc_m, a.disc_v, a.gen_m, a.gen_v = ptr(sd[0]), ptr(sd[1]), ptr(sg[0]), ptr(sg[1])
        else:
            a.rho = disc_opt.rho
            a.disc_rms, a.gen_rms = ptr(disc_opt.slot(disc)), ptr(gen_opt.slot(gen))
        a.seed = int(seed)
        if aug_d is not None:
            a.aug_mask_disc, a.aug_noise_fake, a.aug_noise_real = (ptr(aug_d['mask']), ptr(aug_d['noise_fake']),
                                                                   ptr(aug_d['noise_real']))
        if aug_g is not None:
            a.aug_mask_gen, a.aug_noise_gen = ptr(aug_g['mask']), ptr(aug_g['noise'])
        if split or Bg != B:
            a.B_global, a.row_lo, a.split = Bg, int(row_lo), int(split)
            # loss scalars go to the slots behind the gradients, so they ride the all-reduce
            a.disc_losses = disc.grads.data_ptr() + 4 * disc.n_flat
            a.gen_losses = gen.grads.data_ptr() + 4 * gen.n_flat
        return a

    def iteration_eager(self, x, cc, nc, gen, disc, gen_opt, disc_opt, nb_critic, gp_weight, seed, rng_offset,
                        losses=None):
        """The captured one-GPU iteration's launch sequence queued directly on the stream (what the in-stream
        event brackets of agx_prof_* can time: events cannot be recorded inside a graph).  Returns the
        number of RNG stream positions consumed."""
        lib = _lib.load()
        B = x.shape[0]
        a = self._iter_args(gen, disc, gen_opt, disc_opt, B, nb_critic, gp_weight, seed, None, None, B, 0, False)
        if type(disc_opt) is Adam:
            lrs = _adam_step_sizes(gen, disc, gen_opt, disc_opt, nb_critic)
            a.adam_lr_t_host = _lib.C.cast(lrs, _lib.C.c_void_p)
        draws = _lib.C.c_int64(0)
        _lib.check(lib.agx_cgan_iter_eager(self.handle, _dev.stream(), a, ptr(x), x.stride(0), ptr(cc), ptr(nc),
                                           int(rng_offset), ptr(losses) if losses is not None else None,
                                           _lib.C.byref(draws)), 'agx_cgan_iter_eager')
        return int(draws.value)

    def iteration_graph(self, gen, disc, gen_opt, disc_opt, B, nb_critic, gp_weight, seed, aug_d=None, aug_g=None,
                        batch_global=None, row_lo=0, split=False):
        """Capture (once per key) the whole loop iteration of tf_utils.py:316-323 as a hipGraph
        (split=True: one graph per segment between the gradient all-reduces of a data-parallel run)."""
        Bg = int(batch_global) if batch_global else B
        hyper = ((gen_opt.rho, disc_opt.rho) if type(disc_opt) is RMSprop else
                 (gen_opt.beta_1, gen_opt.beta_2, disc_opt.beta_1, disc_opt.beta_2))
        key = (B, nb_critic, float(gp_weight), int(seed), id(gen_opt), id(disc_opt), type(disc_opt).__name__,
               gen_opt.learning_rate, disc_opt.learning_rate, gen_opt.epsilon, disc_opt.epsilon, hyper,
               id(aug_d), id(aug_g), Bg, int(row_lo), int(split))
        g = self._graphs.get(key)
        if g is None:
            lib = _lib.load()
            a = self._iter_args(gen, disc, gen_opt, disc_opt, B, nb_critic, gp_weight, seed, aug_d, aug_g, Bg, row_lo, split)
            # the graph bakes in the device addresses of the optimizer slots and the augmentation constants:
            # hold them (and the optimizer objects whose id() is in the key) for as long as the graph lives
            self._keep[key] = (aug_d, aug_g, gen_opt, disc_opt, gen_opt.slot(gen), disc_opt.slot(disc))
            while len(self._graphs) >= self.MAX_GRAPHS:     # train() builds fresh optimizers per call: bound the cache
                old = next(iter(self._graphs))
                torch.cuda.synchronize()
                lib.agx_cgan_iter_destroy(self._graphs.pop(old))
                self._keep.pop(old, None)
            g = _lib.C.c_void_p()
            side = torch.cuda.Stream()          # capture needs a non-default stream
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _lib.check(lib.agx_cgan_iter_capture(self.handle, _dev.stream(), a, _lib.C.byref(g)),
                           'agx_cgan_iter_capture')
            torch.cuda.current_stream().wait_stream(side)
            self._graphs[key] = g
        return g

    def drop_graphs(self):
        """Forget captured iterations (e.g. after toggling agx_prof_enable, whose event brackets
        are baked into a graph at capture time)."""
        torch.cuda.synchronize()
        lib = _lib.load()
        for g in self._graphs.values():
            lib.agx_cgan_iter_destroy(g)
        self._graphs.clear()
        self._keep.clear()

    def __del__(self):
        try:
            lib = _lib.load()
            for g in self._graphs.values():
                lib.agx_cgan_iter_destroy(g)
            if self.handle:
                lib.agx_cgan_destroy(self.handle)
        except Exception:
            pass


def _pair(gen, disc, batch, steps=6):
    ent = gen._pair.get(id(disc))
    p = ent[1] if ent is not None and ent[0]() is disc else None   # a dead critic's id() may have been reused
    if p is None or p.max_batch < batch or p.max_steps < steps:
        _dev.device()
        if p is not None:
            batch, steps = max(batch, p.max_batch), max(steps, p.max_steps)
        p = _Pair(gen, disc, max(batch, 32), max(steps, 6))
        for k in [k for k, (ref, _) in gen._pair.items() if ref() is None]:
            del gen._pair[k]
        gen._pair[id(disc)] = (weakref.ref(disc), p)
    return p


def _world():
    return _dev.world()


def _allreduce_grads(net):
    _dev.allreduce_sum(net.grads)  # RCCL sum over xGMI; the loss slots behind the gradients ride along


def _exchange_layout(net, world, rank):
    """(total, shard, lo): the flat buffers' first `total` floats (gradients + loss slots, rounded up to a
    multiple of 4 * world inside the slack) are dealt to the ranks in equal contiguous shards."""
    q = 4 * world
    total = (net.n_flat + _LOSS_SLOTS + q - 1) // q * q
    if total > net._gbuf.numel():
        raise ValueError('world size %d too large for the exchange slack' % world)
    shard = total // world
    return total, shard, rank * shard


def _dp_chunks(net=None, B=None):
    """Pieces of the data-parallel exchange of `net` (0 or 1 = one piece: the flat one-shard-per-rank layout).
    AGX_DP_CHUNKS if set; otherwise 4 from 128 rows per GPU on - the row chunks of the largest weight-gradient product
    (164 us at 512 rows) then cover most of the earlier pieces' exchange - and ONE piece below: at 32 rows that product
    takes 22 us, nothing can hide behind it, and every further piece costs two more collectives per step.
    The choice fixes which rank owns which optimizer slots, so it is made at the network's first data-parallel step
    and kept on the network for the rest of the run."""
    kept = getattr(net, '_dp_pieces', None) if net is not None else None
    if kept is not None:
        return kept
    env = os.environ.get('AGX_DP_CHUNKS')
    n = int(env) if env is not None else (4 if (B is not None and B >= 128) else 1)
    if net is not None and B is not None:
        net._dp_pieces = n
    return n


def _big_kernel(net):
    """(layer, flat offset, rows, ld) of the network's largest Dense kernel (critic: W_1, generator: W_out)."""
    l = max(range(len(net.dims) - 1), key=lambda i: net.dims[i] * net.ldw[i])
    return l, net.w_off[l], net.dims[l], net.ldw[l]


def _exchange_rows(net, chunks):
    """Row boundaries [r_0 = 0, r_1, ..., r_C = rows] of the largest kernel's `chunks` row blocks (multiples of
    64 rows: whole tiles of its weight-gradient product, 16-B aligned operand offsets)."""
    _, _, rows, _ = _big_kernel(net)
    if chunks <= 1:
        return [0, rows]
    cuts = sorted({min(rows, int(round(c * rows / chunks / 64.0)) * 64) for c in range(1, chunks)} - {0, rows})
    return [0] + cuts + [rows]


def _exchange_pieces(net, world, chunks=0):
    """[(start, per)]: contiguous pieces [start, start + per * world) of the flat gradient / parameter buffers
    that tile [0, total); rank r owns [start + r * per, start + (r + 1) * per) of each piece (reduce-scatter
    target, optimizer slots, all-gather source).  chunks <= 1: ONE piece (the flat layout of
    `_exchange_layout`).  Otherwise piece c ends where row block c of the largest kernel ends (rounded down to
    a multiple of 4 * world floats; `_exchange_rows`), so that it is complete - and can travel - as soon as
    that row block of the weight-gradient product is; the last piece also holds everything behind the kernel
    (biases, small kernels, loss slots).  Every piece and every rank's share of it is 16-B aligned."""
    total, shard, _ = _exchange_layout(net, world, 0)
    if chunks <= 1:
        return [(0, shard)]
    q = 4 * world
    _, off, rows, ld = _big_kernel(net)
    bounds = [0] + [(off + r * ld) // q * q for r in _exchange_rows(net, chunks)[1:-1]] + [total]
    bounds = sorted(set(bounds))
    return [(a, (b - a) // world) for a, b in zip(bounds[:-1], bounds[1:])]


def _exchange_plan(net, world, chunks):
    """(rows, pieces) of the overlapped exchange: row chunk c of the largest weight-gradient product and the piece of
    the flat buffers that is complete - and travels - once that chunk is (same cuts as `_exchange_pieces`)."""
    rows, pieces = _exchange_rows(net, chunks), _exchange_pieces(net, world, chunks)
    if len(pieces) != len(rows) - 1:        # degenerate cuts: one chunk, one piece
        rows, pieces = [0, rows[-1]], _exchange_pieces(net, world, 0)
    return rows, pieces


def _dp_native_iteration(x, cc, nc, gen, disc, gen_opt, disc_opt, nb_critic, gp_weight, aug_d, aug_g, Bg, row_lo, comm):
    """One data-parallel loop iteration driven by the library (agx_cgan_dp_iteration): per optimizer step the
    gradient segment, then the largest weight-gradient product in row chunks on the compute stream while a
    second stream reduce-scatters, updates and all-gathers the pieces that are complete - no Python between
    the steps.  Same exchange layout and results as `_dp_update`."""
    lib = _lib.load()
    world, rank = _world()
    B = x.shape[0]
    pair = _pair(gen, disc, B, nb_critic + 1)
    g = pair.iteration_graph(gen, disc, gen_opt, disc_opt, B, nb_critic, gp_weight, _rng.seed, aug_d, aug_g,
                             batch_global=Bg, row_lo=row_lo, split=3)
    a = _lib.DpArgs()
    keep = []
    for net, opt, dst in ((disc, disc_opt, a.disc), (gen, gen_opt, a.gen)):
        rows, pieces = _exchange_plan(net, world, _dp_chunks(net, B))
        if len(pieces) > _lib.AGX_MAX_DP_PIECES:
            raise ValueError('AGX_DP_CHUNKS: at most %d pieces' % _lib.AGX_MAX_DP_PIECES)
        dst.n_pieces = len(pieces)
        for i, r in enumerate(rows):
            dst.rows[i] = r
        for i, (st, per) in enumerate(pieces):
            dst.start[i], dst.per[i] = st, per
        if type(opt) is Adam:
            sl = opt.slot(net)
            dst.slot0, dst.slot1 = ptr(sl[0]), ptr(sl[1])
        else:
            dst.slot0, dst.lr = ptr(opt.slot(net)), opt.learning_rate
    if type(disc_opt) is Adam:
        a.opt_kind, a.beta1, a.beta2 = 1, disc_opt.beta_1, disc_opt.beta_2
        lrs = _adam_step_sizes(gen, disc, gen_opt, disc_opt, nb_critic)
        keep.append(lrs)
        a.lr_t_host = _lib.C.cast(lrs, _lib.C.c_void_p)
    else:
        a.rho = disc_opt.rho
    a.eps = disc_opt.epsilon
    cs = _dev.comm_stream()
    cs.wait_stream(torch.cuda.current_stream())      # (first use: everything queued so far precedes the exchange)
    _lib.check(lib.agx_cgan_dp_iteration(g, _dev.stream(), cs.cuda_stream, comm, a, ptr(x), x.stride(0), ptr(cc), ptr(nc),
                                         _rng.offset), 'agx_cgan_dp_iteration')
    _rng.offset += int(lib.agx_cgan_iter_rng_draws(g))
    disc_opt.iterations += nb_critic
    gen_opt.iterations += 1
    d_loss = _reduced_loss(disc, 2) if nb_critic > 0 else None
    return d_loss, _reduced_loss(gen, 0)


def _dp_update(net, opt):
    """Data-parallel optimizer step over xGMI, driven from Python (the rehearsal / fall-back form; under RCCL the
    library's own overlapped exchange `agx_cgan_dp_*` does the same on two streams): per piece of the exchange
    layout reduce-scatter the flat gradient buffer (each rank receives the sum of its 1/R share), apply the
    optimizer to that share only (the 94 MB RMSprop pass of the GTEx critic becomes 94/R MB per GPU; the
    optimizer slots of the other shares are never touched), all-gather the parameters.  Same bytes on the
    links as one all-reduce of the gradients.  The loss scalars sit in the slots behind the gradients: their
    sums travel back in the parameter buffer's tail (net._pbuf[n_flat:])."""
    world, rank = _world()
    pieces = _exchange_pieces(net, world, _dp_chunks(net, 0))   # (no overlap on this path: one piece unless AGX_DP_CHUNKS asks)
    key = (world, rank, len(pieces))
    bufs = net._shards.get(key)
    if bufs is None:
        m = max(per for _, per in pieces)
        bufs = net._shards[key] = (torch.empty(m, dtype=net._gbuf.dtype, device=net._gbuf.device),
                                   torch.empty(m, dtype=net._pbuf.dtype, device=net._pbuf.device))
    first = True
    for start, per in pieces:
        g_sh, p_sh = bufs[0][:per], bufs[1][:per]
        lo = start + rank * per
        _dev.reduce_scatter_sum(g_sh, net._gbuf[start:start + per * world])
        n_upd = max(0, min(lo + per, net.n_flat) - lo)
        if n_upd:
            opt.apply_range(net, lo, n_upd, g_sh, advance=first)
            first = False
        if lo + per > net.n_flat:           # loss slots (+ padding): the reduced values themselves are what travels
            a = max(lo, net.n_flat)
            net._pbuf[a:lo + per].copy_(g_sh[a - lo:])
        p_sh.copy_(net._pbuf[lo:lo + per])
        _dev.all_gather(net._pbuf[start:start + per * world], p_sh)
    if first:                               # this rank owns no parameter at all: the step still counts
        opt.iterations += 1
        if type(opt) is Adam:
            opt.slot(net)[2] += 1


def _gathered_slot(net, slot):
    """Optimizer slot with every rank's share filled in (each rank only ever updates its own shares).
    A COLLECTIVE under torch.distributed: every rank must call it (hence `state_arrays` too)."""
    world, rank = _world()
    if world == 1:
        return slot
    total = _exchange_layout(net, world, rank)[0]
    full = torch.zeros(total, dtype=slot.dtype, device=slot.device)
    for start, per in _exchange_pieces(net, world, _dp_chunks(net)):
        lo = start + rank * per
        mine = torch.zeros(per, dtype=slot.dtype, device=slot.device)
        n = max(0, min(lo + per, net.n_flat) - lo)
        mine[:n] = slot[lo:lo + n]
        _dev.all_gather(full[start:start + per * world], mine)
    return full[:net.n_flat]


def _reduced_loss(net, idx):
    """Loss scalar `idx` of the last data-parallel step (summed over ranks)."""
    return DeviceArray(net._pbuf[net.n_flat + idx:net.n_flat + idx + 1].clone())


_TRACE_CONSTS = {}


def _aug_constants(tag, key, p_aug, norm_scale, bs, nb_genes, rows=None):
    """The reference draws augmentation masks/noise with NumPy INSIDE ``@tf.function``
    (tf_utils.py:202, :209-210, :240, :247), so they are constants of each trace
    (one per input shape).  Same here: drawn once per (function, models, batch shape)
    with the same NumPy calls in the same order, then reused.  `key` holds weak references to the
    two models (never their id(): a freed model's id can be handed to a new one) and the shape."""
    k = (tag, p_aug, norm_scale, nb_genes) + key
    c = _TRACE_CONSTS.get(k)
    if c is None:
        while len(_TRACE_CONSTS) >= 64:            # bounded: notebooks rebuild models again and again
            _TRACE_CONSTS.pop(next(iter(_TRACE_CONSTS)))
        augs = np.random.binomial(1, p_aug, bs)
        if tag == 'disc':
            nf = np.random.normal(0, norm_scale, nb_genes)
            nr = np.random.normal(0, norm_scale, (bs, nb_genes))
            c = {'mask': augs, 'noise_fake': nf, 'noise_real': nr}
        else:
            c = {'mask': augs, 'noise': np.random.normal(0, norm_scale, (bs, nb_genes))}
        if rows is not None:   # data parallel: constants of the GLOBAL batch (same NumPy seed on every rank), own rows
            lo, hi = rows
            c = {n: (v if n == 'noise_fake' else v[lo:hi]) for n, v in c.items()}
        c = {n: _dev.to_dev(v) for n, v in c.items()}
        _TRACE_CONSTS[k] = c
    return c


def _step_args(pair, z, cc, nc, B, B_global, losses):
    a = _lib.StepArgs()
    a.z, a.cat, a.num = ptr(z), ptr(cc), ptr(nc)
    a.B, a.B_global = B, B_global
    a.losses = losses
    return a


def train_disc(x, z, cc, nc, gen, disc, disc_opt, grad_penalty_weight=10, p_aug=0, norm_scale=0.5, *,
               alpha=None, aug=None, batch_global=None, apply=True):
    """Train critic (reference ``tf_utils.py:186-223``).  Returns the critic loss.

    alpha: optional [B] / [B,1] GP interpolation coefficients (TF draws U[0,1), :162).
    aug:   optional dict(mask=[B], noise_fake=[G], noise_real=[B,G]) overriding the
           per-trace NumPy constants of :202, :209-210.
    """
    lib = _lib.load()
    z = _dev.to_dev(z)
    B = z.shape[0]
    x = _dev.to_dev(x)
    cc = _dev.to_dev(cc, torch.int32)
    nc = _dev.to_dev(nc)
    if x.shape != (B, gen.x_dim):
        raise ValueError('x must be [%d, %d], got %s' % (B, gen.x_dim, tuple(x.shape)))
    pair = _pair(gen, disc, B)
    world, _ = _world()
    if world > 1 and gen.batch_norm:
        raise NotImplementedError('batch-norm generators are not available under data parallelism (batch statistics '
                                  'would need their own exchange)')
    Bg = int(batch_global) if batch_global else B * world
    alpha = random_uniform((B,)) if alpha is None else _dev.to_dev(alpha).reshape(-1)
    losses = disc.grads.data_ptr() + 4 * disc.n_flat
    a = _step_args(pair, z, cc, nc, B, Bg, losses)
    a.x, a.ldx = ptr(x), x.stride(0)
    a.alpha = ptr(alpha)
    a.gp_weight = float(grad_penalty_weight)
    keep = None
    if aug is None and p_aug:
        aug = _aug_constants('disc', (weakref.ref(gen), weakref.ref(disc), B), p_aug, norm_scale, B, gen.x_dim)
    if aug is not None:
        keep = (_dev.to_dev(aug['mask']).reshape(-1), _dev.to_dev(aug['noise_fake']).reshape(-1),
                _dev.to_dev(aug['noise_real']))
        a.aug_mask, a.aug_noise_fake, a.aug_noise_real = ptr(keep[0]), ptr(keep[1]), ptr(keep[2])
    _lib.check(lib.agx_cgan_disc_grads(pair.handle, _dev.stream(), a), 'agx_cgan_disc_grads')
    if world > 1 and apply:
        _dp_update(disc, disc_opt)
        return _reduced_loss(disc, 2)
    if world > 1:
        _allreduce_grads(disc)
    if apply:
        disc_opt.apply(disc)
    return DeviceArray(disc.grads[disc.n_flat + 2:disc.n_flat + 3].clone())


def train_gen(z, cc, nc, gen, disc, gen_opt, p_aug=0, norm_scale=1, *, aug=None, batch_global=None, apply=True):
    """Train generator (reference ``tf_utils.py:226-258``).  Returns the generator loss.

    aug: optional dict(mask=[B], noise=[B,G]) overriding the per-trace constants of :240, :247."""
    lib = _lib.load()
    z = _dev.to_dev(z)
    B = z.shape[0]
    cc = _dev.to_dev(cc, torch.int32)
    nc = _dev.to_dev(nc)
    pair = _pair(gen, disc, B)
    world, _ = _world()
    if world > 1 and gen.batch_norm:
        raise NotImplementedError('batch-norm generators are not available under data parallelism')
    Bg = int(batch_global) if batch_global else B * world
    losses = gen.grads.data_ptr() + 4 * gen.n_flat
    a = _step_args(pair, z, cc, nc, B, Bg, losses)
    keep = None
    if aug is None and p_aug:
        aug = _aug_constants('gen', (weakref.ref(gen), weakref.ref(disc), B), p_aug, norm_scale, B, gen.x_dim)
    if aug is not None:
        keep = (_dev.to_dev(aug['mask']).reshape(-1), _dev.to_dev(aug['noise']))
        a.aug_mask, a.aug_noise_fake = ptr(keep[0]), ptr(keep[1])
    _lib.check(lib.agx_cgan_gen_grads(pair.handle, _dev.stream(), a), 'agx_cgan_gen_grads')
    if world > 1 and apply:
        _dp_update(gen, gen_opt)
        return _reduced_loss(gen, 0)
    if world > 1:
        _allreduce_grads(gen)
    if apply:
        gen_opt.apply(gen)
    return DeviceArray(gen.grads[gen.n_flat:gen.n_flat + 1].clone())


def _graph_ok(gen_opt, disc_opt, world=1, gen=None):
    if gen is not None and getattr(gen, 'batch_norm', False):
        return False      # batch-norm generators: eager step functions (training-mode forward in train_gen only)
    if type(gen_opt) is RMSprop and type(disc_opt) is RMSprop:
        return gen_opt.rho == disc_opt.rho and gen_opt.epsilon == disc_opt.epsilon
    if type(gen_opt) is Adam and type(disc_opt) is Adam:
        return (gen_opt.beta_1 == disc_opt.beta_1 and gen_opt.beta_2 == disc_opt.beta_2
                and gen_opt.epsilon == disc_opt.epsilon)
    return False


def _adam_step_sizes(gen, disc, gen_opt, disc_opt, nb_critic):
    """Bias-corrected step sizes of the coming iteration (nb_critic critic steps, then the generator
    step), computed on the host exactly as Adam.apply does; advances both step counters."""
    sd, sg = disc_opt.slot(disc), gen_opt.slot(gen)
    lrs = (_lib.C.c_float * (nb_critic + 1))()
    for k in range(nb_critic):
        lrs[k] = disc_opt.lr_t(sd[2] + 1 + k)
    lrs[nb_critic] = gen_opt.lr_t(sg[2] + 1)
    sd[2] += nb_critic
    sg[2] += 1
    return lrs


def train_iteration(x, cc, nc, gen, disc, gen_opt, disc_opt, nb_critic=5, grad_penalty_weight=10, p_aug=0,
                    norm_scale=0.5, *, batch_global=None, row_lo=0):
    """One iteration of the reference loop body (tf_utils.py:314-324) - ``nb_critic`` critic steps
    with fresh z / alpha on the same real batch, then one generator step - replayed as ONE hipGraph
    (the generator runs once on all nb_critic+1 z batches: its weights are constant inside an
    iteration).  Draws come from the module Philox stream at the same positions the eager
    ``train_disc`` / ``train_gen`` calls would use.  Returns (last critic loss, generator loss).

    Under torch.distributed (one process per GPU) ``x`` holds this rank's rows
    ``[row_lo, row_lo + len(x))`` of a global batch of ``batch_global`` rows: the iteration is
    captured as nb_critic+1 gradient segments; between them the flat gradient buffer is reduce-scattered
    over RCCL, each rank applies the optimizer to its 1/R shard and the parameters are all-gathered
    (``_dp_update``); draws are the global batch's, so R ranks reproduce the one-GPU big-batch run.
    RMSprop (the reference default) or Adam on both nets; ``train`` falls back to the eager step
    functions otherwise."""
    lib = _lib.load()
    x = _dev.to_dev(x)
    cc = _dev.to_dev(cc, torch.int32)
    nc = _dev.to_dev(nc)
    B = x.shape[0]
    world, _ = _world()
    if not _graph_ok(gen_opt, disc_opt, gen=gen):
        raise ValueError('train_iteration needs the same optimizer class (RMSprop or Adam) with equal decay rates / '
                         'epsilon on both networks, and a generator without batch-norm')
    Bg = int(batch_global) if batch_global else B * world
    if world == 1 and Bg != B:
        raise ValueError('batch_global differs from the local batch on a single GPU')
    pair = _pair(gen, disc, B, nb_critic + 1)
    aug_d = aug_g = None
    if p_aug:   # same per-trace constants, drawn in the order the reference's first calls would (:202-210, :240-247)
        rows = (row_lo, row_lo + B) if world > 1 else None
        aug_d = _aug_constants('disc', (weakref.ref(gen), weakref.ref(disc), Bg, row_lo, B), p_aug, norm_scale, Bg, gen.x_dim, rows)
        aug_g = _aug_constants('gen', (weakref.ref(gen), weakref.ref(disc), Bg, row_lo, B), p_aug, norm_scale, Bg, gen.x_dim, rows)
    # data parallel over RCCL (or AGX_DP_FORCE=1 on one GPU): the library drives the steps and overlaps the exchange
    comm = _dev.native_comm() if os.environ.get('AGX_DP_OVERLAP', '1') != '0' else None
    if comm is not None and nb_critic > 0:
        return _dp_native_iteration(x, cc, nc, gen, disc, gen_opt, disc_opt, nb_critic, grad_penalty_weight, aug_d, aug_g,
                                    Bg, row_lo, comm)
    g = pair.iteration_graph(gen, disc, gen_opt, disc_opt, B, nb_critic, grad_penalty_weight, _rng.seed, aug_d, aug_g,
                             batch_global=Bg, row_lo=row_lo, split=2 if world > 1 else 0)
    if type(disc_opt) is Adam and world == 1:
        lrs = _adam_step_sizes(gen, disc, gen_opt, disc_opt, nb_critic)
        _lib.check(lib.agx_cgan_iter_set_lr(g, lrs, nb_critic + 1), 'agx_cgan_iter_set_lr')
    if world == 1:
        losses = _dev.empty((nb_critic + 1, 4))
        _lib.check(lib.agx_cgan_iter_launch(g, _dev.stream(), ptr(x), x.stride(0), ptr(cc), ptr(nc), _rng.offset,
                                            ptr(losses)), 'agx_cgan_iter_launch')
        d_loss = DeviceArray(losses[nb_critic - 1, 2:3]) if nb_critic > 0 else None
        g_loss = DeviceArray(losses[nb_critic, 0:1])
    else:
        _lib.check(lib.agx_cgan_iter_stage(g, _dev.stream(), ptr(x), x.stride(0), ptr(cc), ptr(nc), _rng.offset),
                   'agx_cgan_iter_stage')
        # nb_critic + 1 gradient segments; between them the exchange + sharded optimizer step (_dp_update)
        d_loss = None
        for s in range(nb_critic + 1):
            _lib.check(lib.agx_cgan_iter_launch_seg(g, _dev.stream(), s), 'agx_cgan_iter_launch_seg')
            if s < nb_critic:
                _dp_update(disc, disc_opt)
                if s == nb_critic - 1:
                    d_loss = _reduced_loss(disc, 2)
            else:
                _dp_update(gen, gen_opt)
        g_loss = _reduced_loss(gen, 0)
        _rng.offset += int(lib.agx_cgan_iter_rng_draws(g))
        return d_loss, g_loss
    _rng.offset += int(lib.agx_cgan_iter_rng_draws(g))
    disc_opt.iterations += nb_critic
    gen_opt.iterations += 1
    return d_loss, g_loss


def _shard(lo, hi, world, rank):
    """Rows [lo, hi) of the global batch owned by `rank` (sizes differ by at most 1)."""
    n = hi - lo
    base, rem = divmod(n, world)
    start = lo + rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def train(dataset, cat_covs, num_covs, z_dim, epochs, batch_size, gen, disc, score_fn, save_fn,
          gen_opt=None, disc_opt=None, nb_critic=5, verbose=True, checkpoint_dir='./checkpoints/cpkt',
          log_dir='./logs/', patience=10, p_aug=0, norm_scale=0.5, *, noise_fn=None, use_graph=True):
    """Train model (reference ``tf_utils.py:261-356``): sequential un-shuffled mini-batches
    including the ragged tail, ``nb_critic`` critic steps (fresh z, same real batch) then one
    generator step, evaluation / ``save_fn`` / patience every 5 epochs.

    The training matrix is uploaded to HBM once and batches are device slices.
    use_graph: replay each iteration as one hipGraph (single GPU + RMSprop; bit-identical to the
    eager step functions for equal seeds).
    noise_fn(kind, shape): optional injection of the TF draws (kind in {'z', 'alpha'}) for the
    GLOBAL batch; default is the device Philox stream.  Returns a trace dict (additive)."""
    if gen_opt is None:
        gen_opt = RMSprop(5e-4)   # tf_utils.py:285-288
    if disc_opt is None:
        disc_opt = RMSprop(5e-4)
    world, rank = _world()
    data = _dev.to_dev(dataset)
    ccs = _dev.to_dev(cat_covs, torch.int32)
    ncs = _dev.to_dev(num_covs)
    n = data.shape[0]

    def draw(kind, rows, cols, lo, hi):
        if noise_fn is not None:
            full = _dev.to_dev(noise_fn(kind, (rows, cols)))
        else:
            full = random_normal((rows, cols)) if kind == 'z' else random_uniform((rows, cols))
        return full[lo:hi].contiguous()

    best_score = -np.inf
    initial_patience = patience
    trace = {'gen_loss': [], 'disc_loss': [], 'score': [], 'saved': [], 'patience': []}
    for epoch in range(epochs):
        gl, dl = [], []
        for i in range(0, n, batch_size):
            hi = min(n, i + batch_size)
            bg = hi - i
            lo_r, hi_r = _shard(i, hi, world, rank)
            if hi_r <= lo_r:
                raise ValueError('global batch of %d rows cannot be split over %d ranks' % (bg, world))
            x, cc, nc = data[lo_r:hi_r], ccs[lo_r:hi_r], ncs[lo_r:hi_r]
            if use_graph and noise_fn is None and nb_critic > 0 and _graph_ok(gen_opt, disc_opt, gen=gen):
                d_l, g_l = train_iteration(x, cc, nc, gen, disc, gen_opt, disc_opt, nb_critic, 10, p_aug, norm_scale,
                                           batch_global=bg, row_lo=lo_r - i)
                dl.append(d_l)
                gl.append(g_l)
                continue
            disc_loss = None
            for _ in range(nb_critic):   # tf_utils.py:316-318
                z = draw('z', bg, z_dim, lo_r - i, hi_r - i)
                al = draw('alpha', bg, 1, lo_r - i, hi_r - i)
                disc_loss = train_disc(x, z, cc, nc, gen, disc, disc_opt, p_aug=p_aug, norm_scale=norm_scale,
                                       alpha=al, batch_global=bg)
            dl.append(disc_loss)         # only the last critic loss is averaged (:319)
            z = draw('z', bg, z_dim, lo_r - i, hi_r - i)
            gl.append(train_gen(z, cc, nc, gen, disc, gen_opt, p_aug=p_aug, norm_scale=norm_scale, batch_global=bg))

        gen_mean = _mean32(gl)
        disc_mean = _mean32(dl)
        if epoch % 5 == 0:               # tf_utils.py:333-343 (checkpoint.save dropped)
            score = score_fn(gen)
            if score > best_score:
                if rank == 0:
                    print('Saving model ...')
                    save_fn()
                best_score = score
                patience = initial_patience
                trace['saved'].append(epoch)
            else:
                patience -= 1
            trace['score'].append(float(score))
            if verbose and rank == 0:
                print('Score: {:.3f}'.format(score))
        if verbose and rank == 0:
            print('Epoch {}. Gen loss: {:.2f}. Disc loss: {:.2f}'.format(epoch + 1, gen_mean, disc_mean))
        trace['gen_loss'].append(float(gen_mean))
        trace['disc_loss'].append(float(disc_mean))
        trace['patience'].append(patience)
        if patience == 0:
            break
    return trace


def _mean32(device_scalars):
    """Keras ``metrics.Mean(dtype=float32)``: fp32 running total / count; one D2H per epoch."""
    if not device_scalars:
        return np.float32(0)
    vals = torch.cat([d.tensor.reshape(1) for d in device_scalars]).cpu().numpy()
    total = np.float32(0)
    for v in vals:
        total = np.float32(total + v)
    return np.float32(total / np.float32(len(vals)))


def make_score_fn(x_test, cat_covs_test, num_covs_test):
    """Device-resident form of the reference scripts' ``score_fn`` closure (gtex_tcga_gan.py:97-108):
    ``score_fn(x_test, cc, nc)(gen) == gamma_coef(x_test, predict(cc, nc, gen))`` with x_test's
    correlation matrix cached in HBM and the generated matrix never copied to the host."""
    from .utils import GammaScorer
    scorer = GammaScorer(x_test)
    cc = _dev.to_dev(cat_covs_test, torch.int32)
    nc = _dev.to_dev(num_covs_test)

    def _score(gen):
        return scorer(predict_sharded(cc, nc, gen))
    return _score


def predict_sharded(cc, nc, gen, z=None):
    """``predict`` with the test set's rows dealt across the ranks of a torch.distributed job (independent
    samples: reference ``tf_utils.py:359-376`` is one full-batch forward): every rank runs the generator on
    its contiguous row shard - the rows of the GLOBAL z draw, so the result equals the one-GPU ``predict`` -
    and one all-gather assembles the [N, G] device matrix on every rank (what the tiled Gamma evaluation
    consumes).  Outside a distributed job this is ``predict(..., training=True)``."""
    world, rank = _world()
    cc = _dev.to_dev(cc, torch.int32)
    nc = _dev.to_dev(nc)
    n = cc.shape[0]
    if z is None:
        z = random_normal([n, gen.input[0].shape[-1]])     # same stream position on every rank
    z = _dev.to_dev(z)
    if world == 1 or n < world:
        return gen([z, cc, nc], training=True).tensor
    per = (n + world - 1) // world
    lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
    g = gen.x_dim
    mine = torch.zeros((per, g), dtype=torch.float32, device=z.device)
    if hi > lo:
        mine[:hi - lo] = gen([z[lo:hi], cc[lo:hi], nc[lo:hi]], training=True).tensor
    full = torch.empty((per * world, g), dtype=torch.float32, device=z.device)
    _dev.all_gather(full.view(-1), mine.view(-1))
    return full[:n]


def predict(cc, nc, gen, z=None, training=False):
    """Make predictions (reference ``tf_utils.py:359-376``): NumPy out unless ``training``."""
    nb_samples = cc.shape[0]
    if z is None:
        z_dim = gen.input[0].shape[-1]
        z = random_normal([nb_samples, z_dim])
    out = gen([z, cc, nc], training=training)
    if not training:
        return out.numpy()
    return out
