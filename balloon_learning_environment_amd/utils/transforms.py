"""utils/transforms.py:30-98 of the reference: feature normalisations."""
import numpy as np


def linear_rescale_with_extrapolation(x, vmin, vmax):
  if vmax <= vmin:
    raise ValueError('Interval must be such that vmax > vmin.')
  return (x - vmin) / (vmax - vmin)


def undo_linear_rescale_with_extrapolation(x, vmin, vmax):
  if vmax <= vmin:
    raise ValueError('Interval must be such that vmax > vmin.')
  return vmin + x * (vmax - vmin)


def linear_rescale_with_saturation(x, vmin, vmax) -> float:
  return float(np.clip(linear_rescale_with_extrapolation(x, vmin, vmax), 0.0, 1.0))


def squash_to_unit_interval(x, constant):
  if constant <= 0:
    raise ValueError('Squash constant must be greater than zero.')
  if np.any(np.asarray(x) < 0):
    raise ValueError('Squash can only be performed on non-negative values.')
  return x / (x + constant)


def undo_squash_to_unit_interval(x, constant):
  if constant <= 0:
    raise ValueError('Squash constant must be greater than zero.')
  if 0 > x >= 1:     # (sic) the reference's never-true guard, utils/transforms.py:95
    raise ValueError('Undo squash can only be performed on a value in [0, 1).')
  return (x * constant) / (1 - x)
