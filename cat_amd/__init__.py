"""cat_amd — MI355X-native (gfx950) implementation of snap-research/CAT's generator-distillation training step.

Public surface mirrors the reference for that path only:
  cat_amd.distillers.create_distiller / InceptionDistiller   (distillers/__init__.py, inception_distiller.py)
  cat_amd.networks.define_G / define_D                        (models/networks.py)
  cat_amd.prune.shrink / shrink_model / get_bn_to_prune       (utils/common.py, utils/prune.py)
  cat_amd.loss.KA / GANLoss                                   (utils/common.py, models/modules/loss.py)
All arithmetic runs in libcat_hip.so (include/cat_hip.h); there is no CPU or eager-PyTorch fallback."""
__version__ = '0.1.0'
