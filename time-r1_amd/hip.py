"""ctypes binding of libtimer1_hip.so. The C header include/timer1_hip.h is the single source of truth for signatures.

There is NO fallback: if the library is missing or a call fails, this raises. The product path never routes through oracle/.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "timer1_hip.h")
LIB_PATH = os.environ.get("TR1_HIP_LIB") or os.path.join(_HERE, "lib", "libtimer1_hip.so")   # TR1_HIP_LIB: A/B builds of the same ABI

_CT = {
    "void*": ctypes.c_void_p, "const void*": ctypes.c_void_p, "char*": ctypes.c_char_p, "const char*": ctypes.c_char_p,
    "int64_t*": ctypes.POINTER(ctypes.c_int64), "const int64_t*": ctypes.POINTER(ctypes.c_int64), "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64, "int": ctypes.c_int,
    "float": ctypes.c_float, "double*": ctypes.POINTER(ctypes.c_double),
}


def parse_header(path=HEADER):
    """-> {name: (restype_str, [(type_str, arg_name), ...])} for every `tr1_*` declaration."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    decls = {}
    for m in re.finditer(r"^\s*(const char\*|int64_t|int)\s+(tr1_\w+)\s*\(([^)]*)\)\s*;", txt, flags=re.M):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(.*?)(\w+)$", a)
                typ = mm.group(1).strip().replace(" *", "*")
                parsed.append((typ, mm.group(2)))
        decls[name] = (ret, parsed)
    return decls


class HipError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise HipError(
                "libtimer1_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "- there is no CPU fallback for the HIP path." % LIB_PATH)
        # torch first: it ships its own libamdhip64 (same soname as /opt/rocm's).  Whichever copy is loaded first serves the whole process,
        # and device tensors only interoperate with the kernels when both sides share torch's runtime ("no ROCm-capable device" otherwise).
        import torch  # noqa: F401
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.decls = parse_header()
        for name, (ret, args) in self.decls.items():
            fn = getattr(self.cdll, name)  # AttributeError if the .so does not export a declared symbol
            fn.argtypes = [_CT[t] for t, _ in args]
            fn.restype = _CT[ret]
        self.cdll.tr1_last_error.restype = ctypes.c_char_p

    def call(self, name, *args):
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            raise HipError("%s failed (code %d): %s" % (name, rc, (self.cdll.tr1_last_error() or b"").decode()))

    def raw(self, name):
        return getattr(self.cdll, name)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
