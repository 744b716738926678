"""Feature normalisations with the reference's names and error behaviour (utils/transforms.py:30-98)."""
import numpy as np


def _span(lo, hi):
  if hi <= lo:
    raise ValueError('Interval must be such that vmax > vmin.')
  return hi - lo


def _positive(constant):
  if constant <= 0:
    raise ValueError('Squash constant must be greater than zero.')
  return constant


def linear_rescale_with_extrapolation(x, vmin, vmax):
  """(x - vmin) / (vmax - vmin), not clipped."""
  return (x - vmin) / _span(vmin, vmax)


def undo_linear_rescale_with_extrapolation(x, vmin, vmax):
  return vmin + x * _span(vmin, vmax)


def linear_rescale_with_saturation(x, vmin, vmax) -> float:
  """The same, clipped to [0, 1]; always a Python float."""
  return float(min(1.0, max(0.0, linear_rescale_with_extrapolation(x, vmin, vmax))))


def squash_to_unit_interval(x, constant):
  """x / (x + constant) for non-negative x (scalar or array)."""
  c = _positive(constant)
  if np.any(np.asarray(x) < 0):
    raise ValueError('Squash can only be performed on non-negative values.')
  return x / (x + c)


def undo_squash_to_unit_interval(x, constant):
  c = _positive(constant)
  if 0 > x >= 1:     # (sic) the reference's never-true guard, utils/transforms.py:95
    raise ValueError('Undo squash can only be performed on a value in [0, 1).')
  return (x * c) / (1 - x)
