"""The committed fixtures ARE what the reference produces today: tests/golden/make_golden.py is re-run against
/root/reference into a scratch directory and every array of every fixture must be EQUAL (keys, shapes, dtypes, bits) to
the committed file.

Why this is a test and not a claim in DESIGN.md: F14's expected values pass through oracle/noise_oracle.py (the stand-in for
the absent opensimplex primitive), which this repository may redefine; round 5 re-rounded it and left the committed F14
3e-15 behind its generator, unnoticed.  A fixture whose generator moved is caught here, mechanically.

Runs only where /root/reference exists (the build container); skipped on the GPU box.  CPU only.
"""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

REFERENCE = '/root/reference/balloon_learning_environment'
pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason='needs the reference checkout (build container only)')

FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'f*.npz')))


@pytest.fixture(scope='module')
def regenerated(tmp_path_factory):
  out = tmp_path_factory.mktemp('golden_regen')
  env = dict(os.environ, BLE_GOLDEN_OUT=str(out))
  proc = subprocess.run([sys.executable, os.path.join(GOLDEN, 'make_golden.py')], env=env, cwd=ROOT, capture_output=True,
                        text=True, timeout=900)
  assert proc.returncode == 0, proc.stderr[-2000:]
  return str(out)


def test_generator_writes_exactly_the_committed_fixture_set(regenerated):
  made = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(regenerated, '*.npz')))
  assert made == FIXTURES


@pytest.mark.parametrize('name', FIXTURES)
def test_fixture_regenerates_bit_for_bit(regenerated, name):
  new = np.load(os.path.join(regenerated, name + '.npz'))
  old = np.load(os.path.join(GOLDEN, name + '.npz'))
  assert sorted(new.files) == sorted(old.files)
  for k in old.files:
    a, b = old[k], new[k]
    assert a.dtype == b.dtype and a.shape == b.shape, (name, k, a.dtype, b.dtype, a.shape, b.shape)
    # bytes, not values: -0.0 vs 0.0 and NaN payloads count
    assert a.tobytes() == b.tobytes(), (name, k, int((a != b).sum()) if a.dtype.kind in 'fiub' else 'differs')
