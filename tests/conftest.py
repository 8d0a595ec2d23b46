import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


# Order of the GPU suite (the driver runs `pytest -m gpu -x`): kernel-level oracle tests and the reference-golden step first, then the
# configuration steps, then full-size / RCCL-launcher / script tests -- so that whatever stops the run, the tests that pin each kernel
# to the oracle have already reported.  Files not listed keep their alphabetical place after the listed ones.
_GPU_FILE_ORDER = ['test_ops_gpu', 'test_halo_gpu', 'test_grad_slots_gpu', 'test_split_norm_dx_gpu', 'test_wgrad_halo_gpu', 'test_graph_program_gpu', 'test_step_gpu', 'test_gradcheck_gpu',
                   'test_distributed_gpu', 'test_inference_gpu', 'test_extras_gpu', 'test_configs_gpu', 'test_fullsize_gpu', 'test_bench_gpu']


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        stem = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _GPU_FILE_ORDER.index(stem) if stem in _GPU_FILE_ORDER else len(_GPU_FILE_ORDER)
    items.sort(key=rank)            # stable: the order inside a file is kept
