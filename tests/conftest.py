import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The parity tests run batches of 1-2 images and are meant to exercise the fused (throughput) plan; the small-batch
# plan has its own tests (tests/test_gpu_graph.py::test_small_batch_plan, test_host_logic.py::test_plan_variants).
os.environ.setdefault('YOLORET_SMALL_BATCH', '0')
# ... and (round 6) on the throughput plan WITH its weight-streaming block form, which a float32 model otherwise runs from 24 images on
# ('mid' below: tests/test_gpu_graph.py::test_small_batch_plan_of_float32_models)
os.environ.setdefault('YOLORET_MBK_BATCH', '0')
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu through gpurun)')


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
