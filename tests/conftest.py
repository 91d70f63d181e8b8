import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _graphs_for_short_runs_too(monkeypatch):
    """htm_run launches calls of fewer than 64 steps eagerly whatever use_graph says (BITHTM_EAGER_BELOW, read when a
    handle is created).  The tests ask for graphs to exercise every capture plan with short runs: switch the policy off --
    by default.  The batched-run tests set the variable themselves and run under the library's own policy as well
    (test_batched_run_equals_step_by_step, test_pipelined_schedules_and_select_paths_equal_step_by_step,
    test_random_call_patterns_equal_step_by_step: `EAGER_BELOW=64`), and the full-size tests make calls on both sides of it."""
    monkeypatch.setenv("BITHTM_EAGER_BELOW", "0")
