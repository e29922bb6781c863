import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


# Collection order = importance for the hot path, not the alphabet: the driver runs `pytest -x`, so whatever fails first hides everything
# behind it.  The stages north_star demands bit-exact (voxel indices, rulebooks) and the headline kernels come first, the newest / widest-scope files last.
ORDER = ["test_abi", "test_voxelize", "test_spconv", "test_vcn", "test_chamfer", "test_vcn_train", "test_config1", "test_configs", "test_pointnet2", "test_boxes",
         "test_boundary", "test_head", "test_center_head", "test_pillars", "test_pvrcnn", "test_second_iou", "test_postprocess", "test_isolation", "test_io_nuscenes",
         "test_collate", "test_dense_ops", "test_pipeline", "test_dist"]


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(ORDER)}
    items.sort(key=lambda it: rank.get(os.path.splitext(os.path.basename(str(it.fspath)))[0], len(ORDER)))       # stable: order inside a file is kept


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip_lib():
    """The C-ABI library, built on demand (cross-compiles without a GPU)."""
    import seevcn_amd._lib as L
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return L.load()


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    return torch.device("cuda:0")
