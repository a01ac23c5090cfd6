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


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def golden_mcl():
    return load_golden('mcl_cases.npz')


@pytest.fixture(scope='session', params=['ingest_ctgs.npz', 'ingest_bins.npz'])
def golden_ingest(request):
    return load_golden(request.param)


@pytest.fixture(scope='session')
def golden_pipeline():
    return load_golden('pipeline_toy.npz')
