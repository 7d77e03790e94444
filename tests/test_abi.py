"""C-ABI checks that need no GPU: libbdrt.so loads and exports every symbol include/bdrt.h declares."""
import ctypes as C
import os
import re

from bayes_drt_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'bdrt.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(bdrt_[a-z_A-Z0-9]+)\s*\(', txt)))


def test_header_and_binding_agree():
    assert sorted(_lib.SYMBOLS) == _declared()


def test_library_exports_every_declared_symbol():
    lib = _lib.load_library()
    for name in _declared():
        assert hasattr(lib, name), name


def test_struct_sizes_match_header():
    # bdrt_dat: 2 int + 3*3 int + 3 double + 4*3 ptr + ptr + int + ptr + 4 double + int + 3 double + int + double
    assert C.sizeof(_lib.Dat) % 8 == 0
    lib = _lib.load_library()
    o = _lib.OptOptions(); lib.bdrt_opt_defaults(C.byref(o))
    assert (o.max_iter, o.history) == (50000, 5) and o.tol_param == 1e-8
    n = _lib.NutsControl(); lib.bdrt_nuts_defaults(C.byref(n))
    assert n.adapt_delta == 0.9 and n.adapt_t0 == 10 and n.max_treedepth == 10 and n.stepsize0 == 1


def test_version_string():
    assert b'gfx950' in _lib.load_library().bdrt_version()
