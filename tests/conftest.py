import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


class Golden(dict):
    """a fixture file read ONCE into memory (an NpzFile inflates the member again on every `g[key]`: a per-entry loop
    over a large fixture took minutes)"""
    @property
    def files(self):
        return list(self)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return Golden({k: z[k] for k in z.files})


@pytest.fixture(scope='session')
def golden_mcl():
    return load_golden('mcl_cases.npz')


@pytest.fixture(scope='session', params=['ingest_ctgs.npz', 'ingest_bins.npz'])
def golden_ingest(request):
    return load_golden(request.param)


@pytest.fixture(scope='session')
def golden_pipeline():
    return load_golden('pipeline_toy.npz')


# ---- wall-clock bookkeeping of the long GPU tests: the driver gives the whole `-m gpu` run 1200 s, so the heaviest oracle legs
# report where the time goes (gpurun_out/test_timings.jsonl when that directory exists) and the most expensive OPTIONAL legs
# skip themselves — loudly — when the session is already late (HHX_TEST_BUDGET_S, default 1000 s)
import contextlib
import json
import time

_T0 = time.time()


def elapsed():
    return time.time() - _T0


def late(margin=0.0):
    """True when the session has used its budget minus `margin` seconds"""
    return elapsed() + margin > float(os.environ.get('HHX_TEST_BUDGET_S', '1000'))


@contextlib.contextmanager
def tick(label):
    t0 = time.time()
    try:
        yield
    finally:
        d = os.path.join(ROOT, 'gpurun_out')
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, 'test_timings.jsonl'), 'a') as fh:
                    fh.write(json.dumps({'label': label, 'seconds': round(time.time() - t0, 2), 'at': round(elapsed(), 1)}) + '\n')
            except OSError:
                pass
