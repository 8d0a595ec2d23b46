"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports every symbol include/ag2v.h declares."""
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, 'include', 'ag2v.h')).read()
    return sorted(set(re.findall(r'\b(ag2v_[a-z0-9_]+)\s*\(', src)))


def test_header_matches_binding():
    from ag2video_amd import _lib
    assert _declared() == list(_lib.EXPORTS)


def _struct_fields(src, pattern):
    """[(C type, name)] of the struct whose body `pattern` captures (comments and blank lines dropped)."""
    body = re.search(pattern, src, re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    body = re.sub(r'//[^\n]*', '', body)
    out = []
    for decl in body.split(';'):
        decl = ' '.join(decl.split())
        if decl:
            m = re.match(r'(.*?[\s\*])([A-Za-z_][A-Za-z0-9_]*)$', decl)
            out.append((m.group(1).replace(' ', ''), m.group(2)))
    return out


def test_conv_opts_struct_has_one_layout():
    """ag2v_conv_opts is declared three times -- include/ag2v.h (the contract), its mirror in csrc/conv_igemm.hip (the library is built
    without the public header) and the ctypes binding: same fields, same order, same C types everywhere."""
    import ctypes
    from ag2video_amd import _lib
    header = _struct_fields(open(os.path.join(REPO, 'include', 'ag2v.h')).read(), r'typedef struct ag2v_conv_opts \{(.*?)\} ag2v_conv_opts;')
    mirror = _struct_fields(open(os.path.join(REPO, 'ag2video_amd', 'csrc', 'conv_igemm.hip')).read(), r'\nstruct ag2v_conv_opts \{(.*?)\n\};')
    assert header == mirror
    ctype_of = {'int': ctypes.c_int, 'constfloat*': ctypes.c_void_p, 'float*': ctypes.c_void_p, 'constchar*': ctypes.c_char_p,
                'size_t': ctypes.c_size_t, 'longlong': ctypes.c_longlong}
    assert [(ctype_of[t], n) for t, n in header] == [(t, n) for n, t in _lib.ConvOpts._fields_]


def test_library_exports_every_symbol():
    from ag2video_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r' T (ag2v_[a-z0-9_]+)', out))
    missing = [s for s in _declared() if s not in exported]
    assert not missing, "libag2v.so lacks %s" % missing
    lib = _lib.load()          # dlopen + signature binding (no kernel is launched without a GPU)
    assert lib is not None
    assert _lib.query('ag2v_reduce_workspace_floats') == 1024
    assert _lib.query('ag2v_bn_workspace_floats', 2, 3, 5000) == 3 * 2 * 2 * 4 + 6


def test_ops_refuse_cpu_tensors():
    """The product path has no CPU fallback: operators raise on CPU tensors."""
    import torch
    from ag2video_amd import ops
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.randn(1, 3, 8, 8), torch.randn(4, 3, 3, 3))
    with pytest.raises(RuntimeError):
        ops.leaky_relu(torch.randn(8))
