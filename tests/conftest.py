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
def _literal_batches(request):
    """Tests pass small b_size values to drive the multi-batch logic (short last batch, plans per batch, graph segments):
    the evaluator's internal batch coalescing (evaluation.COALESCE_BATCH) is switched off for them -- unless the test is
    parametrised over ``coalesce_mode``: 'literal' (b_size taken literally, as above) or 'default' (the SHIPPED setting:
    the facts are processed max(b_size, 32768) at a time).  The evaluator tests that compare ranks with the reference
    (parity, query columns, skewed graphs) run in both modes; the sharded workers and the full-split tests run the
    shipped default."""
    import torchkge_amd.evaluation as ev
    old = ev.COALESCE_BATCH
    mode = 'literal'
    if hasattr(request.node, 'callspec'):
        mode = request.node.callspec.params.get('coalesce_mode', 'literal')
    if mode == 'literal':
        ev.COALESCE_BATCH = 0
    yield
    ev.COALESCE_BATCH = old
