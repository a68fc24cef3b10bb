"""ctypes binding of libha2g_hip.so -- the C-ABI library of hand-written gfx950 kernels.

The library is the product path: there is no fallback.  If it is missing or a symbol is absent the import
fails loudly (build with `python -c "import __graft_entry__ as g; g.build()"` or `make -C ha2g_amd/csrc`).
Prototypes mirror include/ha2g_hip.h ('p' = device pointer, 'i' = int, 'l' = long, 'f' = float).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libha2g_hip.so')

_T = {'p': ctypes.c_void_p, 'i': ctypes.c_int, 'l': ctypes.c_long, 'f': ctypes.c_float}

# name -> argument kinds (return type is int unless listed in _RET)
SIGNATURES = {
    'ha2g_abi_version': '',
    'ha2g_gemm_f32': 'iiiiifplplfplpipl' + 'p',
    'ha2g_colsum_f32': 'pllipf' + 'p',
    'ha2g_conv2d_fwd_f32': 'pppp' + 'iiiiiiiii' + 'i' + 'p',
    'ha2g_conv2d_dgrad_f32': 'ppp' + 'iiiiiiiii' + 'f' + 'p',
    'ha2g_conv2d_wgrad_workspace_bytes': 'iiiiiiiii',
    'ha2g_conv2d_wgrad_f32': 'ppp' + 'iiiiiiiii' + 'f' + 'pl' + 'p',
    'ha2g_gru_packed_floats': 'i',
    'ha2g_gru_supported_hidden': 'i',
    'ha2g_gru_pack_whh': 'pppi' + 'p',
    'ha2g_gru_layer_fwd': 'pppppp' + 'iii' + 'p',
    'ha2g_gru_layer_bwd': 'ppppp' + 'iii' + 'p',
}
_RET = {'ha2g_conv2d_wgrad_workspace_bytes': ctypes.c_long, 'ha2g_gru_packed_floats': ctypes.c_long}


class Ha2gError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError('ha2g_amd: %s not found -- the HIP extension is required (no CPU/torch fallback). '
                          'Build it with `make -C ha2g_amd/csrc` or __graft_entry__.build().' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.ha2g_last_error.restype = ctypes.c_char_p
    lib.ha2g_last_error.argtypes = []
    for name, kinds in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise ImportError('ha2g_amd: symbol %s missing from %s (stale build?)' % (name, LIB_PATH))
        fn.argtypes = [_T[k] for k in kinds]
        fn.restype = _RET.get(name, ctypes.c_int)
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise Ha2gError('ha2g kernel call failed (%d): %s' % (rc, lib.ha2g_last_error().decode()))
