import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "multistream: opt-in experiment (slices of a batch on separate HIP streams); runs only with "
                                       "ARREAU_TEST_MULTISTREAM=1 or when the -m expression names it, and then a mismatch FAILS")


def pytest_collection_modifyitems(config, items):
    """The multi-stream mode is refused by the library by default (results were seen to differ in rare runs, DESIGN.md
    section 8), so its test is not part of the product's suite: it is skipped -- visibly -- unless asked for, and when it
    runs a mismatch is a failure, never an expected failure."""
    wanted = os.environ.get("ARREAU_TEST_MULTISTREAM", "0") == "1" or "multistream" in (config.getoption("-m") or "")
    if wanted:
        return
    skip = pytest.mark.skip(reason="opt-in experiment: ARREAU_TEST_MULTISTREAM=1 (or -m multistream) runs it; a mismatch then fails")
    for item in items:
        if "multistream" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
