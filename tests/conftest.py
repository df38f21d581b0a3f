import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


class _EnvAndOptions:
    """monkeypatch whose IOPX_* environment changes also reach libraries that are already loaded: the library asks the environment for an option
    ONCE (at the name's first lookup) and keeps the answer in its option table, so a test that switches a schedule between two proofs of one
    process does it through iopx_set_option / iopx_clear_option — which this wrapper calls beside the environment change (child processes a test
    spawns still see the environment)."""

    def __init__(self, mp):
        self._mp = mp
        self._touched = set()

    def __getattr__(self, name):
        return getattr(self._mp, name)

    @staticmethod
    def _libs():
        import libiop_amd
        libs = [getattr(libiop_amd, "_default", None)]
        emu_mod = sys.modules.get("emu_lib")
        if emu_mod is not None:
            libs.append(getattr(emu_mod, "_emu", None))
        return [l for l in libs if l is not None]

    def setenv(self, name, value, prepend=None):
        self._mp.setenv(name, value, prepend)
        if name.startswith("IOPX_") and str(value).lstrip("-").isdigit():
            self._touched.add(name)
            for lib in self._libs():
                lib.set_option(name, int(value))

    def delenv(self, name, raising=True):
        self._mp.delenv(name, raising)
        if name.startswith("IOPX_"):
            self._touched.add(name)
            for lib in self._libs():
                lib.clear_option(name)

    def _undo_options(self):
        for name in self._touched:
            for lib in self._libs():
                lib.clear_option(name)


@pytest.fixture
def monkeypatch(monkeypatch):
    wrapped = _EnvAndOptions(monkeypatch)
    yield wrapped
    monkeypatch.undo()                      # the environment first, then the option tables forget what the test set
    wrapped._undo_options()
