"""ctypes binding of libmagpo_hip.so, generated from include/magpo.h.

The header is the single source of truth for the C ABI: this module parses its prototypes and
derives ``argtypes`` / ``restype`` from them, so binding and header cannot drift.  There is no
fallback: if the HIP library is missing the import of any product op raises immediately.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "magpo.h")
LIB_PATH = os.path.join(_HERE, "libmagpo_hip.so")

_SCALARS = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "uint32_t": ctypes.c_uint32,
    "magpo_stream_t": ctypes.c_void_p,
}
_PROTO = re.compile(r"^\s*(const char\*|int|long)\s+(magpo_\w+)\s*\(([^;]*?)\)\s*;", re.M | re.S)


def parse_header(path: str = HEADER) -> Dict[str, Tuple[str, List[Tuple[str, str]]]]:
    """name -> (return type, [(ctype, argname)])"""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for ret, name, args in _PROTO.findall(text):
        params = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                m = re.match(r"(.+?)\s*(\w+)$", a)
                params.append((m.group(1).strip(), m.group(2)))
        out[name] = (ret, params)
    return out


def _ctype(t: str):
    if "*" in t:
        return ctypes.c_void_p
    t = t.replace("const", "").strip()
    return _SCALARS[t]


class MagpoError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the MAGPO HIP library is required (no CPU fallback). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
        self._dll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self._dll.magpo_last_error.restype = ctypes.c_char_p
        for name, (ret, params) in self.protos.items():
            fn = getattr(self._dll, name)
            fn.argtypes = [_ctype(t) for t, _ in params]
            fn.restype = {"int": ctypes.c_int, "long": ctypes.c_long, "const char*": ctypes.c_char_p}[ret]

    def last_error(self) -> str:
        return (self._dll.magpo_last_error() or b"").decode()

    def raw(self, name):
        return getattr(self._dll, name)

    timer = None  # optional object with .begin(name, args) -> token and .end(token) (bench.py instrumentation)

    def call(self, name: str, *args):
        """Call an int-status entry point; tensors are passed as their data_ptr()."""
        if self.timer is not None:
            tok = self.timer.begin(name, args)
            if tok is not None:
                try:
                    return self._call(name, *args)
                finally:
                    self.timer.end(tok)
        return self._call(name, *args)

    def _call(self, name: str, *args):
        fn = getattr(self._dll, name)
        conv = []
        for a in args:
            if a is None:
                conv.append(None)
            elif hasattr(a, "data_ptr"):
                conv.append(a.data_ptr())
            else:
                conv.append(a)
        rc = fn(*conv)
        if self.protos[name][0] == "int" and name not in ("magpo_abi_version", "magpo_row_grid", "magpo_retention_num_chunks", "magpo_seg_bwd_grid",
                                                       "magpo_class_sum_slots", "magpo_obsnorm_grid", "magpo_sable_act_envs_per_wave") and rc != 0:
            msg = self.last_error()
            if rc == -1:
                raise ValueError(f"{name}: {msg}")
            raise MagpoError(f"{name} failed ({rc}): {msg}")
        return rc


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
