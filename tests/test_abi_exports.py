"""The C-ABI shared library loads without a GPU and exports every function include/*.h declares (no compute calls)."""
import ctypes
import glob
import os
import re

import __graft_entry__ as ge
from maro_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for h in glob.glob(os.path.join(REPO, "include", "*.h")):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(mrx_\w+)\s*\(", src))
    return names


def test_library_exports_every_declared_entry_point():
    ge.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 22
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    assert set(_lib.EXPORTS) == names


def test_errors_are_reported_not_thrown():
    L = _lib.load()
    assert L.mrx_cim_workspace_bytes(None, None) < 0 and b"null" in L.mrx_last_error()
    assert L.mrx_cb_workspace_bytes(None, None) < 0 and b"null" in L.mrx_last_error()
    assert L.mrx_cim_attr_id(0, b"empty") == 1 and L.mrx_cim_attr_id(0, b"nope") == -1
    assert L.mrx_cb_attr_id(0, b"bikes") == 0 and L.mrx_cb_attr_id(1, b"trips_adj") == 0 and L.mrx_cb_attr_id(0, b"nope") == -1
    assert b"gfx950" in L.mrx_version()
