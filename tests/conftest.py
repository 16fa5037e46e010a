import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu via gpurun)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(autouse=True)
def _literal_batches():
    """Tests pass small b_size values to drive the multi-batch logic (short last batch, plans per batch, graph segments):
    the evaluator's internal batch coalescing (evaluation.COALESCE_BATCH) is switched off for them;
    test_batch_coalescing_gives_identical_ranks switches it back on."""
    import torchkge_amd.evaluation as ev
    old = ev.COALESCE_BATCH
    ev.COALESCE_BATCH = 0
    yield
    ev.COALESCE_BATCH = old
