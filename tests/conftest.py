import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def port():
  from oracle import oracle
  return oracle.port()


@pytest.fixture(scope="session")
def ref():
  """The reference compiled in place (oracle/_ref); None where it was never built."""
  from oracle import oracle
  return oracle.ref()


@pytest.fixture(scope="session")
def checker():
  from oracle import oracle
  return oracle.best()
