"""Measurement knobs of the executors, read ONCE when this module is imported.

None of them changes a result beyond the order of floating-point additions; the defaults are what the
benchmarks were measured with.  They exist for A/B runs and for the counter passes of the profiling scripts
(which need every branch of a step folded into one stream); product code reads the constants below and never
looks at the environment itself.  (The kernel library has its own selection overrides, STMI_NO_* / STMI_*_CFG,
documented where they are read in csrc/: they force a slower kernel family and are used by the tests that
compare the families with each other.)

  STMI_SERIAL        letters g / w / c: keep the Gram terms / weight gradients / content-target pass on the main
                     stream instead of a side stream; r: 3-channel layers on row patches    (default: '')
  STMI_BWD_CUTS      data parallel only: first block of every backward range but the last, e.g. '3', '11,7', ''
                     (one bucket); one all-reduce bucket per range                           (default: '3')
  STMI_GRAD_BUFS     gradient buffers the stylizer's backward rotates through                (default: 6)
  STMI_TILE_STATS    0: convolutions leave no per-tile sums, every instance norm runs its own statistics pass
  STMI_TILE_STATS_MAX  longest tile row per sample the norm's apply blocks fold themselves   (default: 32)
  STMI_IN_SUMS       0: data-gradient kernels leave no sums for the norm below them
  STMI_ABSORB_NORM   0: at inference the second convolution of a residual block reads the normalised tensor an
                     apply pass wrote, instead of absorbing that norm + ReLU itself
  STMI_FUSED_VGG     0: a training step runs the content pictures through a VGG pass of their own (beside the
                     stylizer forward) instead of in ONE pass of 2 x batch samples with the stylized pictures
"""
import os


def _flag(name, default='1'):
    return os.environ.get(name, default) == '1'


SERIAL = os.environ.get('STMI_SERIAL', '')
BWD_CUTS = [c for c in os.environ.get('STMI_BWD_CUTS', '3').split(',') if c.strip()]
GRAD_BUFS = int(os.environ.get('STMI_GRAD_BUFS', '6'))
TILE_STATS = _flag('STMI_TILE_STATS')
TILE_STATS_MAX = int(os.environ.get('STMI_TILE_STATS_MAX', '32'))
IN_SUMS = _flag('STMI_IN_SUMS')
FUSED_VGG = _flag('STMI_FUSED_VGG')
ABSORB_NORM = _flag('STMI_ABSORB_NORM')
