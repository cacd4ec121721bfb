#!/usr/bin/env python3
"""Counterpart of the reference's GTEx+TCGA entry script on this package (reference
``gtex_tcga_gan.py:21-140``): same preprocessing order, model sizes, optimizers, closures and final
print - with a synthetic expression matrix where the reference reads RNASeqDB TSVs (the loaders are
outside this package; pass your own arrays to ``run``).

    python examples/train_gtex_shaped.py [--samples 9147] [--genes 18154] [--epochs 10]
    python -m torch.distributed.run --nproc-per-node N examples/train_gtex_shaped.py   (data parallel)
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adversarial_gene_expression_amd.tf_utils import (make_discriminator, make_generator, make_score_fn,  # noqa: E402
                                                      tfk, train)
from adversarial_gene_expression_amd.utils import split_train_test, standardize  # noqa: E402

# gtex_tcga_gan.py:11-18 (the 'gpu' entry selects a TF device there; HIP_VISIBLE_DEVICES does it here)
CONFIG = {'epochs': 2000, 'latent_dim': 64, 'batch_size': 32, 'nb_layers': 2, 'hdim': 256, 'lr': 5e-4, 'nb_critic': 5}


def synthetic_rnaseqdb(n_samples, n_genes, seed=0):
    """Counts-like matrix with low-rank structure + covariates with the reference's cardinalities:
    15 tissues, 3 datasets in the proportions of gtex_tcga_analysis_2.ipynb (2322 / 683 / 6142 of 9147)."""
    rs = np.random.RandomState(seed)
    tissue = rs.randint(0, 15, n_samples)
    dataset = rs.choice(3, n_samples, p=np.array([2322, 683, 6142]) / 9147.0)
    load = rs.randn(32, n_genes)
    factors = rs.randn(n_samples, 32) + 0.5 * rs.randn(15, 32)[tissue] + 0.3 * rs.randn(3, 32)[dataset]
    log_expr = 2.0 + 0.35 * (factors @ load) / np.sqrt(32.0) + 0.3 * rs.randn(n_samples, n_genes)
    counts = np.expm1(np.clip(log_expr, 0.0, 12.0))
    return counts, np.stack([tissue, dataset], axis=1)


def run(x, cat_covs, vocab_sizes, epochs, models_dir=None, config=CONFIG):
    x = np.float32(np.log(1 + x))                                  # :31-32
    cat_covs = np.int32(cat_covs)                                  # :45
    num_covs = np.zeros((x.shape[0], 1), dtype=np.float32)         # :53 (placeholder numeric covariate)

    np.random.seed(0)                                              # :59-65 one shuffle, then head/tail split
    idx = np.arange(x.shape[0])
    np.random.shuffle(idx)
    x, num_covs, cat_covs = x[idx], num_covs[idx], cat_covs[idx]
    x_train, x_test = split_train_test(x)
    nc_train, nc_test = split_train_test(num_covs)
    cc_train, cc_test = split_train_test(cat_covs)

    mean, std = np.mean(x_train, axis=0), np.std(x_train, axis=0)  # :71-74 train statistics for both splits
    x_train = standardize(x_train, mean=mean, std=std)
    x_test = standardize(x_test, mean=mean, std=std)

    h_dims = [config['hdim']] * config['nb_layers']                # :87-94
    gen = make_generator(x.shape[-1], vocab_sizes, num_covs.shape[-1], h_dims=h_dims, z_dim=config['latent_dim'])
    disc = make_discriminator(x.shape[-1], vocab_sizes, num_covs.shape[-1], h_dims=h_dims)

    # :97-108 - gamma_coef(x_test, predict(...)) with the test-set matrix cached in HBM
    score_fn = make_score_fn(x_test, cc_test, nc_test)

    def save_fn():                                                 # :111-112
        if models_dir:
            gen.save(os.path.join(models_dir, 'gen_rnaseqdb.h5'))

    train(dataset=x_train, cat_covs=cc_train, num_covs=nc_train, z_dim=config['latent_dim'],
          batch_size=config['batch_size'], epochs=epochs, nb_critic=config['nb_critic'], gen=gen, disc=disc,
          gen_opt=tfk.optimizers.RMSprop(config['lr']), disc_opt=tfk.optimizers.RMSprop(config['lr']),
          score_fn=score_fn, save_fn=save_fn)                      # :116-136
    score = score_fn(gen)
    print('Gamma(Dx, Dz): {:.2f}'.format(score))                   # :139-140
    return gen, float(score)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples', type=int, default=9147)
    ap.add_argument('--genes', type=int, default=18154)
    ap.add_argument('--epochs', type=int, default=CONFIG['epochs'])
    ap.add_argument('--models-dir', default=None)
    a = ap.parse_args()
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
    counts, cat = synthetic_rnaseqdb(a.samples, a.genes)
    run(counts, cat, [15, 3], a.epochs, a.models_dir)


if __name__ == '__main__':
    main()
