#!/usr/bin/env python3
"""Counterpart of the reference's E. coli entry script on this package (reference ``ecoli_gan.py:9-96``):
float32 cast -> split -> z-score with train statistics -> 128x128 generator / critic, latent 10, batch 32,
default RMSprop(5e-4), p_aug = 0.25 -> Gamma(Dx, Dz).  The M3D compendium and its condition parser are
outside this package: a synthetic matrix with the same shape (907 chips, 1076 or 4297 genes) and the seven
categorical condition variables of ``ecoli_discrete_conditions`` (6, 8, 5, 11, 5, 3, 5 levels) stands in.

    python examples/train_ecoli_shaped.py [--genes 1076] [--epochs 20]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adversarial_gene_expression_amd.tf_utils import make_discriminator, make_generator, predict, train  # noqa: E402
from adversarial_gene_expression_amd.utils import gamma_coef, standardize  # noqa: E402

EPOCHS, BATCH_SIZE, LATENT_DIM = 500, 32, 10      # ecoli_gan.py:4-7
VOCAB = [6, 8, 5, 11, 5, 3, 5]


def synthetic_compendium(n_chips=907, n_genes=1076, seed=0):
    rs = np.random.RandomState(seed)
    cond = np.stack([rs.randint(0, v, n_chips) for v in VOCAB], axis=1)
    load = rs.randn(24, n_genes)
    factors = rs.randn(n_chips, 24)
    for i, v in enumerate(VOCAB):
        factors += 0.4 * rs.randn(v, 24)[cond[:, i]]
    expr = 8.0 + (factors @ load) / np.sqrt(24.0) + 0.4 * rs.randn(n_chips, n_genes)
    return expr, cond


def run(expr, conditions, epochs=EPOCHS, train_rate=0.75, models_dir=None):
    expr = np.float32(expr)                                       # :38
    conditions = np.int32(conditions)                             # :31
    n_train = int(train_rate * expr.shape[0])                     # stands in for the replicate-aware split (:37, 680/227)
    perm = np.random.RandomState(0).permutation(expr.shape[0])
    tr, te = perm[:n_train], perm[n_train:]
    expr_train, expr_test = expr[tr], expr[te]
    cond_train, cond_test = conditions[tr], conditions[te]
    mean, std = np.mean(expr_train, axis=0), np.std(expr_train, axis=0)   # :45-48
    expr_train = standardize(expr_train, mean=mean, std=std)
    expr_test = standardize(expr_test, mean=mean, std=std)
    nc_train = np.zeros([cond_train.shape[0], 1])                 # :53-54 (float64 zeros, cast on entry like Keras)
    nc_test = np.zeros([cond_test.shape[0], 1])
    gen = make_generator(expr.shape[-1], VOCAB, 1, z_dim=LATENT_DIM, h_dims=[128, 128])     # :57-58
    disc = make_discriminator(expr.shape[-1], VOCAB, 1, h_dims=[128, 128])

    def score_fn(x_test, cat_covs_test, num_covs_test):           # :62-74, verbatim call pattern
        def _score(gen):
            x_gen = predict(cc=cat_covs_test, nc=num_covs_test, gen=gen)
            return gamma_coef(x_test, x_gen)
        return _score

    def save_fn():                                                # :77-78
        if models_dir:
            gen.save(os.path.join(models_dir, 'gen_ecoli.h5'))

    train(dataset=expr_train, cat_covs=cond_train, num_covs=nc_train, z_dim=LATENT_DIM, batch_size=BATCH_SIZE,
          epochs=epochs, gen=gen, disc=disc, p_aug=0.25, score_fn=score_fn(expr_test, cond_test, nc_test),
          save_fn=save_fn)                                        # :82-92
    score = score_fn(expr_test, cond_test, nc_test)(gen)
    print('Gamma(Dx, Dz): {:.2f}'.format(score))                  # :95-96
    return gen, float(score)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--genes', type=int, default=1076)
    ap.add_argument('--epochs', type=int, default=EPOCHS)
    ap.add_argument('--models-dir', default=None)
    a = ap.parse_args()
    e, c = synthetic_compendium(n_genes=a.genes)
    run(e, c, a.epochs, models_dir=a.models_dir)
