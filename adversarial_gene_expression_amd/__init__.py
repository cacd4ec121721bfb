"""agx: MI355X-native cGAN train step + gene x gene correlation evaluation.

Drop-in for the hot path of rvinas/adversarial-gene-expression:

    from adversarial_gene_expression_amd.utils import *      # pearson_correlation, gamma_coef, ...
    from adversarial_gene_expression_amd.tf_utils import *   # make_generator, train, predict, ...

Importing the package does not touch the GPU; the first call does, and raises
``AgxError`` if libagx.so or an MI355X is missing (no CPU fallback).
"""
from ._lib import AgxError, LIB_PATH  # noqa: F401

__all__ = ['AgxError', 'LIB_PATH', 'tf_utils', 'utils']
