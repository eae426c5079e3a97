"""CPU: the C-ABI library loads and exports every symbol include/ibloc.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ibloc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ibl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ibloc_amd import _lib
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in include/ibloc.h but not exported by libibloc_hip.so"


def test_python_binding_covers_the_header():
    from ibloc_amd import _lib
    assert sorted(_lib.declared_symbols()) == declared_functions()


def test_error_channel_without_gpu():
    from ibloc_amd import _lib
    st = _lib.lib.ibl_assign_batch(None, None, 0, 0, 0, 0, None, None, None, 0, 0)
    assert st < 0 and b"null" in _lib.lib.ibl_last_error()


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import importlib
    from ibloc_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib._load()
        assert False, "expected IblError"
    except _lib.IblError as e:
        assert "no CPU fallback" in str(e)
