import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fetch_models():
    from gymnasium_robotics_amd.envs.fetch import load_fetch_model

    return {t: load_fetch_model(t) for t in ("FetchReach", "FetchPush", "FetchSlide", "FetchPickAndPlace")}
