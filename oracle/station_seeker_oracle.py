"""CPU restatement of the reference's StationSeeker controller (TEST INFRASTRUCTURE ONLY).

Follows agents/station_seeker_agent.py of the reference:
  pick_action               :72-86    level < centre -> UP (2), level > centre -> DOWN (0), else STAY (1)
  find_best_pressure_level  :88-115   argmax of altitude_score over the valid levels (first maximum wins,
                                      best_score starts at 0 with a strict '>')
  altitude_score            :117-150  (1 - u + eps) wind_score + u g_unknown + k2 exp(-k3 |level - centre|)
  wind_score                :152-186  de-normalised bearing / magnitude, distance-dependent bearing weight
and env/features.py:146-266 (PerciatelliWindFeature.is_valid_wind, NamedPerciatelliFeatures,
convert_wind_feature_to_real_wind -- including its field order: it builds
PerciatelliWindFeature(uncertainty, un-rescaled BEARING, un-squashed MAGNITUDE)).
Vectorised over the 361 relative levels in float64 on the float32 feature vector, as NumPy promotes there.
Pinned by tests/test_oracle_golden.py against fixture F13 (every action and chosen level of a
960-step reference episode).  Imported by tests/ only; the product package has no agents.
"""
import numpy as np

HALF_RADIUS = 35.0
MAGNITUDE_WEIGHT = 0.07
CLOSE_BEARING_WEIGHT = 0.6
FAR_BEARING_WEIGHT = 0.45
CLOSE_BEARING = 250.0
FAR_BEARING = 500.0
DEFAULT_SCORE = 0.5
HYSTERESIS_K2 = 0.05
HYSTERESIS_K3 = 0.001
CONFIDENCE_EPSILON = 0.01


def scores(features: np.ndarray) -> np.ndarray:
  """altitude_score of every level (0 where the level is not valid)."""
  f = np.asarray(features)
  assert f.shape == (1099,)
  winds = f[16:].reshape(361, 3)                     # the centred column: 2 x 181 - 1 relative levels
  unc32, bear32, mag32 = winds[:, 0], winds[:, 1], winds[:, 2]
  valid = (mag32 != 1.0) | (bear32 != 1.0) | (unc32 != 0.0)            # features.py:153-159
  unc = unc32.astype(np.float64)
  with np.errstate(divide='ignore', invalid='ignore'):
    bearing = 0.0 + bear32.astype(np.float64) * (np.pi - 0.0)           # undo_linear_rescale_with_extrapolation(., 0, pi)
    magnitude = (mag32.astype(np.float64) * 30.0) / (1 - mag32.astype(np.float64))   # undo_squash_to_unit_interval(., 30)
    d = np.float64(f[7])
    distance = (d * 250.0) / (1 - d)                                    # station_seeker_agent.py:164-165
    coeff = np.clip((distance - CLOSE_BEARING) / (FAR_BEARING - CLOSE_BEARING), 0.0, 1.0)
    bearing_weight = CLOSE_BEARING_WEIGHT + coeff * (FAR_BEARING_WEIGHT - CLOSE_BEARING_WEIGHT)
    alpha_delta = np.exp(-distance / HALF_RADIUS)
    wind_score = (1 - alpha_delta) * np.exp(-bearing_weight * bearing) + alpha_delta * np.exp(-MAGNITUDE_WEIGHT * magnitude)
  level_distance = np.abs(np.arange(361) - 180)                           # wind_column_center() = 361 // 2
  hysteresis = HYSTERESIS_K2 * np.exp(-HYSTERESIS_K3 * level_distance)
  s = (1.0 - unc + CONFIDENCE_EPSILON) * wind_score + unc * DEFAULT_SCORE + hysteresis
  return np.where(valid, s, 0.0)


def best_level(features: np.ndarray) -> int:
  s = scores(features)
  best, best_score = None, 0.0
  for l in range(361):                                                  # first strict maximum, as the reference loop
    if s[l] > best_score:
      best_score, best = s[l], l
  assert best is not None, 'At least one pressure level should be valid.'
  return best


def pick_action(features: np.ndarray) -> int:
  level = best_level(features)
  return 2 if level < 180 else (0 if level > 180 else 1)
