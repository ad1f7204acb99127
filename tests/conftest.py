import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "enhancing-transformers_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def lpips_random_init(monkeypatch):
    """explicit opt-in to a randomly initialised LPIPS trunk (no lpips weights can be obtained offline; enhancing/losses/lpips.py raises without it)"""
    monkeypatch.setenv("ENH_LPIPS_RANDOM_INIT", "1")
