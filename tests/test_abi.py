"""CPU: the C-ABI shared library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports exactly the
entry points include/cat_hip.h declares (no compute calls here)."""
import os
import re

from cat_amd import _build, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _headers_text():
    """cat_hip.h (the default path) + the opt-in headers beside it"""
    inc = os.path.join(ROOT, 'include')
    return '\n'.join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith('.h'))


def _declared():
    text = _headers_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(cat_[a-z0-9_]+)\s*\(', text)))


def test_library_builds_and_exports_header_symbols():
    path = _build.build(verbose=False)
    assert os.path.exists(path)
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/cat_hip.h but not exported by libcat_hip.so'
    assert lib.cat_hip_version() >= 2
    assert lib.cat_hip_last_error() is not None


def test_ctypes_signatures_cover_the_header():
    assert sorted(_lib.SIGNATURES) == _declared()


def test_workspace_queries_are_pure_host_functions():
    import ctypes as C
    lib = _lib.load()
    g = _lib.ConvGeom(16, 64, 64, 77, 80, 64, 64, 18, 20, 5, 5, 1, 2, 1, 0, 0.0, 20)
    assert lib.cat_conv2d_wgrad_ws_bytes(C.byref(g)) > 0
    n = _lib.NormGeom(16, 4096, 77, 80, _lib.NORM_BATCH, 1e-5, 0.1, 1, 0.0)
    assert lib.cat_norm_ws_bytes(C.byref(n)) > 0
    assert lib.cat_ka_ws_bytes(16) > 0


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under cat_amd/ may import it (or torch's CPU ops as a fallback path)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'cat_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in re.sub(r'""".*?"""', '', src, flags=re.S), f


def _header_prototypes():
    text = _headers_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    text = re.sub(r'//[^\n]*', '', text)
    protos = {}
    for ret, name, args in re.findall(r'([A-Za-z_][A-Za-z0-9_ \*]*?)\s*\b(cat_[a-z0-9_]+)\s*\(([^)]*)\)\s*;', text):
        args = [a.strip() for a in args.split(',')] if args.strip() not in ('', 'void') else []
        protos[name] = (ret.strip(), args)
    return protos


def _kind(ctype_decl):
    """'p' pointer-sized handle, 'i' int, 'f' float, 'l' 64-bit int, 'z' size_t, 'd' double"""
    d = ctype_decl.replace('const', ' ').strip()
    if '*' in d or d.startswith('cat_stream_t'):
        return 'p'
    base = d.split()[0] if len(d.split()) == 1 else ' '.join(d.split()[:-1])
    return {'int': 'i', 'float': 'f', 'int64_t': 'l', 'long long': 'l', 'size_t': 'z', 'double': 'd', 'char': 'p'}[base]


def test_ctypes_argument_lists_match_the_header():
    """Same number of arguments, same kinds, in the same order as the prototypes in include/cat_hip.h: a mismatch would only show as a
    crash on the GPU box."""
    import ctypes as C
    kinds = {C.c_int: 'i', C.c_float: 'f', C.c_int64: 'l', C.c_size_t: 'z', C.c_double: 'd', C.c_void_p: 'p', C.c_char_p: 'p'}
    protos = _header_prototypes()
    assert sorted(protos) == sorted(_lib.SIGNATURES)
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        ret, args = protos[name]
        want = [_kind(a) for a in args]
        got = [kinds.get(t, 'p') for t in argtypes]      # POINTER(struct) and friends are pointers
        assert got == want, f'{name}: ctypes {got} vs header {want}'
        rk = 'p' if '*' in ret else {'int': 'i', 'size_t': 'z', 'void': 'v', 'double': 'd'}[ret.replace('const', '').strip()]
        assert kinds.get(restype, 'v' if restype is None else 'p') == rk, f'{name}: return type'


def test_ctypes_structs_match_the_header():
    """Field names, order and types of cat_conv_t / cat_norm_t against their ctypes mirrors."""
    import ctypes as C
    text = open(os.path.join(ROOT, 'include', 'cat_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    for cname, cls in (('cat_conv_t', _lib.ConvGeom), ('cat_norm_t', _lib.NormGeom)):
        body = re.search(r'typedef struct\s*(?:\w+\s*)?\{([^}]*)\}\s*' + cname + r'\s*;', text).group(1)
        fields = []
        for decl in body.split(';'):
            decl = decl.strip()
            if decl:
                ctype, names = decl.split(None, 1)
                fields += [(n.strip(), {'int': C.c_int, 'float': C.c_float}[ctype]) for n in names.split(',')]
        assert fields == list(cls._fields_), cname


def test_qconv_structs_match_their_ctypes_mirrors(tmp_path):
    """sizeof / offsetof of cat_qseg_t, cat_qconv_t, cat_qplan_t as the C compiler lays them out (gcc on include/cat_hip.h) against the
    ctypes mirrors in cat_amd/_lib.py: a drifted field would only show as garbage geometry on the GPU box."""
    import ctypes as C
    import subprocess
    structs = {'cat_qseg_t': _lib.QSeg, 'cat_qconv_t': _lib.QConv, 'cat_qplan_t': _lib.QPlan, 'cat_tseg_t': _lib.TSeg, 'cat_tconv_t': _lib.TConv,
               'cat_nslice_t': _lib.NSlice, 'cat_ksum_seg_t': _lib.KSeg, 'cat_ksum_t': _lib.KSum, 'cat_dwmulti_t': _lib.DwMulti, 'cat_tstage1w_t': _lib.Stage1WGeom, 'cat_wgrad_item_t': _lib.WgradItem}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "cat_hip.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / 'probe.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'probe'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    got = {}
    for ln in out.splitlines():
        c, f, v = ln.split()
        got[(c, f)] = int(v)
    for cname, cls in structs.items():
        assert got[(cname, 'size')] == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)


def test_production_library_carries_no_diagnostic_switches():
    """The ablation switches that make results wrong by design and the per-phase clock probes exist only in the diagnostic build
    (common.h kDiag, `python -m cat_amd._build --diag`): in the production library their environment reads are compiled out."""
    from cat_amd import _build
    blob = open(_build.LIB, 'rb').read()
    for name in (b'CAT_PK_ABLATE', b'CAT_Q_ABLATE', b'CAT_DBG', b'CAT_SCHED', b'CAT_LDS_PAD', b'CAT_ABLATE'):
        assert name not in blob, name
    assert not os.path.basename(_build.DIAG_LIB) == os.path.basename(_build.LIB)
