"""pytest configuration: marker registration + shared fixtures.

`-m "not gpu"` : oracle vs goldens, host logic, C-ABI symbol/ABI checks (no GPU needed).
`-m gpu`       : parity tests proper -- HIP path (through the C-ABI) vs the oracle.
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "node-speex-resampler_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    import oracle
    if not os.path.exists(oracle.ORACLE_SO):
        oracle.build()
