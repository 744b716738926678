"""pytest configuration: registers the `gpu` marker, puts the repo root on sys.path and bounds the host thread pools."""
import os
import sys

# The oracle's small OpenMP regions (20-element cold starts inside every feature vector) and NumPy / SciPy's BLAS (120 x 120 factorisations)
# start one thread per HARDWARE thread they can see -- 256 on the GPU boxes, under a 16-CPU cgroup quota -- and spend their time waking
# them: the oracle-heavy tests ran 4 x slower for it (59 s -> 15 s for tests/test_oracle_golden.py on 8 cores).  Before NumPy and the
# oracle library are loaded; an explicit setting of the caller's wins.  (Calls that pass a thread count -- oracle.step(threads=...) --
# are not affected: num_threads clauses override the default.)
for _var in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
  os.environ.setdefault(_var, '4')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
  if p not in sys.path:
    sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')
