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


def test_struct_layouts_match_the_header(tmp_path):
    """the ctypes mirrors of the header's structs (ibloc_amd/vit.py, dator.py, preprocess.py, registration.py) have the sizes and the field
    offsets gcc gives the declarations of include/ibloc.h -- a drifted mirror would hand the library garbage pointers"""
    import subprocess
    from ibloc_amd import dator as D
    from ibloc_amd import preprocess as pp
    from ibloc_amd import registration as R
    from ibloc_amd import vit as V
    pairs = [("ibl_vit_desc", V.VitDesc), ("ibl_vit_layer", V.VitLayer), ("ibl_vit_weights", V.VitWeights),
             ("ibl_dator_head_weights", D.DatorHeadWeights), ("ibl_crop_desc", pp.CropDesc), ("ibl_instance_features", R._FeatStruct)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "ibloc.h"', 'int main(void) {']
    for cname, ct in pairs:
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'  printf(" %zu", offsetof({cname}, {fname}));')
        lines.append('  printf("\\n");')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    for (cname, ct), line in zip(pairs, out):
        parts = line.split()
        assert parts[0] == cname
        assert int(parts[1]) == ctypes.sizeof(ct), f"sizeof({cname}) = {parts[1]}, ctypes mirror {ctypes.sizeof(ct)}"
        for (fname, _), off in zip(ct._fields_, parts[2:]):
            assert int(off) == getattr(ct, fname).offset, f"{cname}.{fname}: offset {off} in the header, {getattr(ct, fname).offset} in the mirror"
