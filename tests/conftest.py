"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch, with one addition: the library reads its RSX_* switches once per process, so a test that sets
    or deletes one has the library read them again (rsx_reload_env) -- at once, and when the patch is undone."""
    real_setenv, real_delenv = monkeypatch.setenv, monkeypatch.delenv

    def reload():
        import radix_sorting_amd as rsa
        try:
            rsa.reload_env()
        except Exception:      # (no library / no GPU: nothing has cached anything)
            pass

    def setenv(name, value, prepend=None):
        real_setenv(name, value, prepend)
        if name.startswith("RSX_"):
            reload()

    def delenv(name, raising=True):
        real_delenv(name, raising)
        if name.startswith("RSX_"):
            reload()

    monkeypatch.setenv = setenv
    monkeypatch.delenv = delenv
    yield monkeypatch
    monkeypatch.undo()
    reload()
