"""ctypes binding of liblhrs_hip.so (the C ABI declared in include/lhrs_hip.h).

The prototypes are parsed from the header itself, so the header is the single source of truth and the
`-m "not gpu"` test-suite can check that the library exports every declared symbol.  There is NO fallback:
if the library is missing or a call is rejected, a RuntimeError is raised (the product path must fail loudly
when the HIP extension is absent - it never routes through oracle/ or a CPU path).
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(REPO_ROOT, "include", "lhrs_hip.h")
LIB_PATH = os.environ.get("LHRS_HIP_LIB") or os.path.join(_HERE, "csrc", "liblhrs_hip.so")  # override: kernel experiments only

_PROTO = re.compile(r"^\s*(const\s+char\s*\*|int|long|void)\s+(lhrs_\w+)\s*\(([^;{]*)\)\s*;", re.M | re.S)


def _strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _ctype_of(param: str):
    p = param.strip()
    if "*" in p:
        return ctypes.c_void_p
    base = p.rsplit(" ", 1)[0] if " " in p else p
    if "float" in base:
        return ctypes.c_float
    if "long" in base:
        return ctypes.c_long
    if base.strip() == "unsigned":
        return ctypes.c_uint
    if "int" in base or "uint8_t" in base:
        return ctypes.c_int
    raise ValueError(f"unparsed C parameter: {param!r}")


def parse_header(path: str = HEADER) -> Dict[str, Tuple[object, List[object]]]:
    """-> {symbol: (restype, [argtypes])} for every prototype in the header."""
    text = _strip_comments(open(path).read())
    out: Dict[str, Tuple[object, List[object]]] = {}
    for ret, name, params in _PROTO.findall(text):
        params = " ".join(params.split())
        args = [] if params in ("", "void") else [_ctype_of(p) for p in params.split(",")]
        restype = ctypes.c_char_p if "char" in ret else (None if ret.strip() == "void" else (ctypes.c_long if ret.strip() == "long" else ctypes.c_int))
        out[name] = (restype, args)
    return out


_lib = None


def load() -> ctypes.CDLL:
    """Load the HIP library; raise (never fall back) if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  torch's bundled HIP runtime must be the one in the process before the .so binds to it
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the LHRS hot path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in parse_header().items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    if os.environ.get("LHRS_DEBUG_POISON_LDS") == "1":  # test aid: see lhrs_debug_poison_lds in include/lhrs_hip.h
        lib = _PoisonProxy(lib)
    _lib = lib
    return lib


class _PoisonProxy:
    """LHRS_DEBUG_POISON_LDS=1: every entry point that takes a stream is preceded by a launch that fills all LDS with NaN bit patterns on
    that stream - an operator that reads LDS it did not write then returns NaN instead of silently using a previous kernel's leftovers."""

    def __init__(self, lib):
        self._lib, self._wrapped = lib, {}
        self._streamed = {n for n, (_, a) in parse_header().items() if a and a[-1] is ctypes.c_void_p and n != "lhrs_debug_poison_lds"}
        self._pattern = int(os.environ.get("LHRS_DEBUG_POISON_PATTERN", "0xFFFFFFFF"), 0)

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name not in self._streamed:
            return fn
        if name not in self._wrapped:
            poison, pattern = self._lib.lhrs_debug_poison_lds, self._pattern

            def call(*args, _fn=fn):
                poison(pattern, args[-1])
                return _fn(*args)
            self._wrapped[name] = call
        return self._wrapped[name]


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().lhrs_last_error()
        raise RuntimeError(f"{what} rejected by liblhrs_hip: {msg.decode() if msg else 'unknown error'}")
