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
