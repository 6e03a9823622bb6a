import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


class Golden(dict):
    def t(self, k):
        return torch.from_numpy(np.asarray(self[k]))


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return Golden({k: z[k] for k in z.files})


@pytest.fixture
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _fresh_f16_census(request):
    """The range census of the fp16 planes (point_teacher_amd.planes.CENSUS) is process-wide state: a group demoted by one GPU test
    (on purpose, or by a diverging toy model) must not decide the operand format of the next test."""
    if 'gpu' in request.keywords and torch.cuda.is_available():
        from point_teacher_amd import planes
        planes.CENSUS.reset()
    yield
