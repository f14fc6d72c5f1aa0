"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/idh.h
declares (no compute calls — there is no GPU here), and argument validation paths that do
not touch the device behave."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _header_symbols():
    txt = "".join(open(os.path.join(ROOT, "include", f)).read() for f in sorted(os.listdir(os.path.join(ROOT, "include"))) if f.endswith(".h"))
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(idh_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from implicit_depth_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    h = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 5
    for s in syms:
        assert hasattr(h, s), f"{s} declared in include/idh.h but not exported by libidh.so"
    # the ctypes table binds exactly the header's symbols
    assert sorted(_lib.declared_symbols()) == syms


def test_error_strings_and_host_only_entry_points():
    from implicit_depth_amd import _lib

    L = _lib.lib()
    assert L.idh_version() >= 100
    assert L.idh_error_string(0) == b"ok"
    assert b"workspace" in L.idh_error_string(-4)
    # argument validation happens before any launch: safe without a GPU
    assert L.idh_cost_volume_dot_fwd(None, None, None, None, None, 0.25, 5.0, 1, 2, 16, 8, 8, 4, None, 0, None, None, None) == -1
    assert L.idh_cost_volume_dot_fwd(None, None, None, None, None, 0.25, 5.0, 1, 2, 8, 8, 8, 4, None, 0, None, None, None) == -2


def test_product_path_refuses_cpu_tensors():
    import torch

    import implicit_depth_amd.synthetic as syn
    from implicit_depth_amd import _lib
    from implicit_depth_amd.cost_volume import CostVolumeManager

    inp = syn.cost_volume_inputs(1, 2, 16, 8, 8, 0)
    with pytest.raises(_lib.IdhError):
        CostVolumeManager(8, 8, 4)(**inp)


def test_op_descriptor_layout_matches_library():
    import ctypes

    from implicit_depth_amd import _lib, nhwc

    assert ctypes.sizeof(nhwc.Op) == _lib.lib().idh_sizeof_op()
    assert _lib.lib().idh_packed_weight_floats(40, 24, 3) == 9 * 32 * 48
    assert _lib.lib().idh_run_ops(None, 0, None) == 0
