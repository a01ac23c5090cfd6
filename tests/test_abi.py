"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU
and exports exactly the symbols include/haphic_hip.h declares (no compute calls here)."""
import os
import re

from haphic_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'haphic_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return set(re.findall(r'\b(hhx_[a-z0-9_]+)\s*\(', txt))


def test_library_builds_and_exports_header():
    build.build()
    lib = _lib.load()
    decl = declared_symbols()
    assert decl, 'no declarations parsed'
    assert decl == set(_lib.SIGNATURES), decl ^ set(_lib.SIGNATURES)
    for name in decl:
        assert hasattr(lib, name), name
    assert lib.hhx_version() >= 100


def test_header_cites_reference_lines():
    txt = open(os.path.join(ROOT, 'include', 'haphic_hip.h')).read()
    for cite in (':2017-2023', ':2026-2062', ':1987-2014', ':310-373', ':1596-1655', ':1658-1752', ':2065-2095'):
        assert cite in txt, cite


def test_no_oracle_in_product():
    """the product package must never import or link the oracle (parity would be void)"""
    pkg = os.path.join(ROOT, 'haphic_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f), errors='replace').read()
                assert not re.search(r'^\s*(from|import)\s+oracle|hhx_oracle|\borc_', src, flags=re.M), (dirpath, f)


def test_new_bindings_marshal_and_reject_null_handles():
    """argument marshalling of the round-2 entry points (a ctypes ArgumentError only shows at call time); a null
    handle must come back as an error code, not a crash"""
    import numpy as np
    from haphic_amd import _lib
    L = _lib.load()
    out = np.zeros(4, np.int64)
    assert L.hhx_row_products(None, None, _lib.ptr(out)) != 0 and b'null' in L.hhx_last_error()
    o, f, z = _lib.C.c_void_p(), _lib.C.c_int64(0), _lib.C.c_int64(0)
    assert L.hhx_expand_links(None, 0, 0, 52, 2.0, 1e-4, _lib.C.byref(o), _lib.C.byref(f), _lib.C.byref(z)) != 0
    d = _lib.C.c_void_p()
    assert L.hhx_expand_links_dense(None, 0, 0, 52, 0, _lib.C.byref(d), _lib.C.byref(f), _lib.C.byref(z)) != 0
    assert L.hhx_dense_inflate_prune(None, 2.0, 1e-4, _lib.C.byref(o)) != 0 and L.hhx_dense_free(None) == 0
    assert L.hhx_csr_pack_block(None, None, 0) != 0
    assert L.hhx_tune(b'cls', 1) == 0


def test_tune_knobs_documented_and_refused_when_unknown():
    """every knob hhx_tune accepts (k_tune_names in csrc/hhx_runtime.hip) is named in the header's description of hhx_tune, every knob a kernel
    reads (tune_get in csrc/) is one hhx_tune accepts, and an unknown name is refused (hhx_tune does not touch the GPU)"""
    rt = open(os.path.join(ROOT, 'haphic_amd', 'csrc', 'hhx_runtime.hip')).read()
    block = re.search(r'k_tune_names\[\]\s*=\s*\{(.*?)nullptr\};', rt, flags=re.S).group(1)
    block = re.sub(r'#ifdef HHX_PROBE_BUILD.*?#endif', '', block, flags=re.S)            # the measurement build's own knob
    names = re.findall(r'"([a-z0-9_]+)"', block)
    assert len(names) >= 10 and 'block_tiles' in names
    header = open(os.path.join(ROOT, 'include', 'haphic_hip.h')).read()
    for n in names:
        assert '"%s"' % n in header, 'hhx_tune knob %r is not described in include/haphic_hip.h' % n
    read = set()
    for f in os.listdir(os.path.join(ROOT, 'haphic_amd', 'csrc')):
        if f.endswith(('.hip', '.h')):
            read |= set(re.findall(r'tune_get\("([a-z0-9_]+)"', open(os.path.join(ROOT, 'haphic_amd', 'csrc', f)).read()))
    # resources of the host side, read once through the same helper and settable by their environment variable alone (HHX_FILE_LANES: writer threads,
    # HHX_BED_HBM_GB: the size of the alignments.bed ring in HBM) — not kernel classes, hence not hhx_tune names
    env_only = {'file_lanes', 'bed_hbm_gb'}
    assert read - {'probe'} - env_only <= set(names), 'read by a kernel, refused by hhx_tune: %s' % sorted(read - {'probe'} - env_only - set(names))
    lib = _lib.load()
    assert lib.hhx_tune(b'no_such_knob', 1) != 0
    assert b'unknown knob' in lib.hhx_last_error()
    assert lib.hhx_tune(b'block_tiles', 0) == 0 and lib.hhx_tune(b'block_tiles', -2 ** 63) == 0
