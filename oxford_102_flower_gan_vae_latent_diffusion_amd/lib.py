"""ctypes binding of libflowers_hip.so.

The C header (include/flowers_hip.h) is the single source of truth: it is parsed here to
build the ctypes signatures, so a symbol declared there but missing from the .so (or the
reverse) fails at load time. There is NO fallback path: if the library is missing the
package raises, loudly (the product never routes through the CPU oracle).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "flowers_hip.h")
LIB_PATH = os.environ.get("FL_LIB_PATH", os.path.join(_HERE, "libflowers_hip.so"))  # override: timing-only ablation builds

_CTYPES = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "size_t": ctypes.c_size_t,
    "fl_stream_t": ctypes.c_void_p,
    "void": None,
}


def _ctype(tok):
    tok = tok.strip()
    if tok.endswith("*"):
        if tok.replace("const", "").strip() == "char*":
            return ctypes.c_char_p
        return ctypes.c_void_p
    tok = tok.replace("const", "").strip()
    return _CTYPES[tok]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every function prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"^[ \t]*#[^\n]*$", "", src, flags=re.M)  # preprocessor lines are not part of any prototype
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(fl_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or "enum" in ret:
            continue
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                t, n = mm.group(1).strip(), mm.group(2)
                argtypes.append(_ctype(t))
                argnames.append(n)
        protos[name] = (_ctype(ret) if ret != "void" else None, argtypes, argnames)
    return protos


class FlowersLibError(RuntimeError):
    pass


_lib = None
_protos = None


def load():
    """dlopen the in-tree library and bind every prototype of the header. Needs no GPU."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    # PyTorch ships its own HIP runtime: it must be the first one the process maps. If this library is dlopen'ed before
    # `import torch` (build() followed by smoke() in one process) the system runtime comes in first and the second runtime
    # then finds no device ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise FlowersLibError(
            "libflowers_hip.so not found at %s — build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH
        )
    lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    missing = []
    for name, (res, argtypes, _) in _protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = argtypes
    if missing:
        raise FlowersLibError("libflowers_hip.so lacks symbols declared in flowers_hip.h: %s" % missing)
    _lib = lib
    return lib


def prototypes():
    load()
    return _protos


def last_error():
    return load().fl_last_error_string().decode()


def call(name, *args):
    """Call an int-returning entry point; raise RuntimeError with the library's message on failure."""
    fn = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError("%s failed (status %d): %s" % (name, rc, last_error()))
    return rc
