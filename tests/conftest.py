import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Tests that FORK (DataLoader workers, torch.distributed.run / bench.py subprocesses) run first: a fork copies the page tables of the whole
# test process, and behind the full-depth parity test (27 GB of fp32 oracle weights + autograd) each worker start took ~40 s on the GPU box
# (the stage-1 driver test: 176 s at the end of the suite, 27-35 s on its own).
_FORKING_FIRST = ("test_surface_gpu.py", "test_trainer_gpu.py", "test_bench_gpu.py", "test_dp_gpu.py")


def pytest_collection_modifyitems(config, items):
    import torch

    items.sort(key=lambda it: 0 if os.path.basename(str(it.fspath)) in _FORKING_FIRST else 1)   # stable: the order inside each class stays
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _restore_default_conversation():
    """`lhrs.Dataset.conversation.default_conversation` is a MODULE GLOBAL that the datasets re-bind to their prompt template (the reference does
    the same: cap_dataset.py:343, 397, 664); a test that builds a "plain" stage-1 dataset must not change what a later test's evaluation prompt
    looks like."""
    from lhrs_bot_amd import conversation as conversation_lib
    saved = conversation_lib.default_conversation
    yield
    conversation_lib.default_conversation = saved
