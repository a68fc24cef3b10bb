"""ctypes binding of libha2g_hip.so -- the C-ABI library of hand-written gfx950 kernels.

The library is the product path: there is no fallback.  If it is missing, or any symbol declared in
include/ha2g_hip.h is absent, the import fails loudly (build with __graft_entry__.build() or
`make -C ha2g_amd/csrc`).  Prototypes are parsed from the header so Python and C cannot drift apart.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libha2g_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'ha2g_hip.h')

_SCALARS = {'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float, 'double': ctypes.c_double, 'unsigned': ctypes.c_uint}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    protos = {}
    for m in re.finditer(r'\b(const\s+char\s*\*|int|long|void)\s+(ha2g_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if '*' in ret else (None if ret == 'void' else _SCALARS[ret])
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_SCALARS[a.replace('const ', '').split()[0]])
        protos[name] = (restype, argtypes)
    return protos


class Ha2gError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError('ha2g_amd: %s not found -- the HIP extension is required (no CPU/torch fallback). '
                          'Build it with `make -C ha2g_amd/csrc` or __graft_entry__.build().' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in parse_header().items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise ImportError('ha2g_amd: symbol %s (declared in include/ha2g_hip.h) missing from %s -- stale build?'
                              % (name, LIB_PATH))
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def header_abi_version(path=HEADER_PATH):
    m = re.search(r'#define\s+HA2G_ABI_VERSION\s+(\d+)', open(path).read())
    return int(m.group(1))


ABI_VERSION = header_abi_version()
_clib = _load()                      # the ctypes view: symbol check against the header, and the fallback binding
if _clib.ha2g_abi_version() != ABI_VERSION:
    raise ImportError('ha2g_amd: %s reports ABI %d, include/ha2g_hip.h declares %d -- stale build (make -C ha2g_amd/csrc)'
                      % (LIB_PATH, _clib.ha2g_abi_version(), ABI_VERSION))


class _FastLib:
    """`lib.<entry point>` through the METH_FASTCALL extension built next to the library (csrc/gen_fastcall.py: ~0.3 us per call instead
    of ctypes' ~0.3 us per ARGUMENT); entry points it lacks (stale build) and the whole binding when it is missing fall back to ctypes --
    either way every call lands in libha2g_hip.so."""

    def __init__(self, clib):
        self._clib = clib
        try:
            from . import _ha2g_fastcall as fc
        except ImportError:
            fc = None
        self.fastcall = fc is not None and os.environ.get('HA2G_FASTCALL', '1') != '0'
        if self.fastcall:
            for name in parse_header():
                if hasattr(fc, name):
                    setattr(self, name, getattr(fc, name))

    def __getattr__(self, name):         # only reached for names not bound above
        return getattr(self._clib, name)


lib = _FastLib(_clib)


def check(rc):
    if rc != 0:
        raise Ha2gError('ha2g kernel call failed (%d): %s' % (rc, lib.ha2g_last_error().decode()))

# matrix-core mode (see ha2g_gemm_set_mode in include/ha2g_hip.h); the HA2G_GEMM_MODE environment variable overrides the default (tests restore
# THIS value after toggling modes).  The A/B switches of closed experiments are no longer read from the environment: tools and tests call the
# library's debug entry points (ha2g_conv_planes_tile3 / _debug / _korder, ha2g_conv_c32_prefetch, ha2g_gemm_debug_tile, ...) directly.
# 70 = 64 + 6 (round 4): every split product -- the trunk's forward AND backward convolutions, the dense products >= 4 GFLOP, both GRU chains, every
# backward GEMM -- on THREE bf16 pieces per operand (all 24 mantissa bits, six MFMAs: fp32-class, the reference's arithmetic class); the small forward
# GEMMs, the tap convolutions and the stem stay on the fp32 MFMA.  6 = the round-3 default (two-piece backward: 16-bit operand mantissa, fp32-MFMA
# forward), 0 = exact fp32 MFMA everywhere.
DEFAULT_GEMM_MODE = int(os.environ.get("HA2G_GEMM_MODE", "70"))
DEFAULT_DIRECT_C32 = 1
lib.ha2g_gemm_set_mode(DEFAULT_GEMM_MODE)
if os.environ.get("HA2G_TILE3") is not None:        # step-level A/B of the plane kernels' form (ha2g_conv_planes_tile3: 8 = the q kernel instead of the patch-resident one)
    lib.ha2g_conv_planes_tile3(int(os.environ["HA2G_TILE3"]))
lib.ha2g_conv_debug_direct_c32(DEFAULT_DIRECT_C32)
