import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / 'tests' / 'golden'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden_loss():
    import numpy as np
    return np.load(GOLDEN / 'loss_reference.npz')


@pytest.fixture(scope='session')
def fixtures():
    import numpy as np
    return np.load(GOLDEN / 'fixtures.npz')
