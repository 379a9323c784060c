import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


class _EqdEnv:
    """pytest's monkeypatch with one addition: after an EQD_* environment variable changes, the loaded kernel library is
    told to forget its snapshot of those switches (it reads them once per process, include/equidock_hip.h:
    eqd_tunables_reload)."""

    def __init__(self, mp):
        self._mp = mp

    def __getattr__(self, name):
        return getattr(self._mp, name)

    @staticmethod
    def _reload():
        from equidock_public_amd import _lib
        _lib.reload_tunables()

    def setenv(self, name, value, prepend=None):
        self._mp.setenv(name, value, prepend)
        if name.startswith('EQD_'):
            self._reload()

    def delenv(self, name, raising=True):
        self._mp.delenv(name, raising)
        if name.startswith('EQD_'):
            self._reload()


@pytest.fixture
def monkeypatch(monkeypatch):
    env = _EqdEnv(monkeypatch)
    yield env
    monkeypatch.undo()      # (restores the environment now, so that the reload below sees the restored values)
    env._reload()
