"""Bit-identity of the five-wave form (experiment build, -DBLE_WITH_SPLIT5) against the one-lane kernel: the assertions of
tests/test_gpu_parity.py::test_split_kernel_equals_one_lane_kernel and ::test_fused_rollout_in_ground_truth_wind_... with waves = '5'.
  BLE_HIP_LIB=build_ab/libble_split5.so python profiles/split5_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
  sys.path.insert(0, p)
import test_gpu_parity as t  # noqa: E402
from balloon_learning_environment_amd import vec_state  # noqa: E402

for wide in (False, True):
  t.test_split_kernel_equals_one_lane_kernel.__wrapped__(vec_state, wide, '5') if hasattr(t.test_split_kernel_equals_one_lane_kernel, '__wrapped__') else t.test_split_kernel_equals_one_lane_kernel(vec_state, wide, '5')
  print(f'five waves == one lane (wide={wide}): bit for bit')
for cache in (True, False):
  t.test_fused_rollout_in_ground_truth_wind_equals_noise_plus_single_steps(vec_state, cache, '5')
  print(f'five waves, in-kernel noise == noise kernel + single steps (cache={cache}): bit for bit')
