"""ctypes binding of libmmgclip_hip.so (the gfx950 kernels) — prototypes are parsed from include/mmgclip_hip.h.

There is NO fallback: if the library is missing or a call is rejected, this module raises.  The product path
never routes through the CPU oracle (oracle/ is test infrastructure only).
"""
import ctypes
import os
import re
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_ROOT = os.path.dirname(_HERE)                      # mmg-clip_amd/
_REPO_ROOT = os.path.dirname(_PKG_ROOT)
LIB_PATH = os.environ.get("MMGCLIP_HIP_LIB", os.path.join(_PKG_ROOT, "csrc", "libmmgclip_hip.so"))
HEADER_PATH = os.path.join(_REPO_ROOT, "include", "mmgclip_hip.h")
ABI_VERSION = 5      # 5: mmg_dwconv7_nhwc_mfma (round 4); 2: mmg_cnblock_mlp_fwd gained the optional xln output; the dropout entry points; 3: kernel-name notes; 4: cnblock_mlp_fwd gact, NT epilogue 5

_CTYPES = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "long long": ctypes.c_longlong,
    "unsigned long long": ctypes.c_ulonglong,
    "unsigned": ctypes.c_uint,
    "size_t": ctypes.c_size_t,
    "mmg_stream_t": ctypes.c_void_p,
}


class HipLibraryError(RuntimeError):
    pass


def parse_header(path=HEADER_PATH):
    """Return {name: (restype, [argtypes], [argnames])} for every function declared in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"(?:^|\n)\s*((?:const\s+)?[\w ]+?\**)\s*\b(mmg_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.+?)\s*(\w+)$", a)
                typ, nm = mm.group(1).strip(), mm.group(2)
                argnames.append(nm)
                if "*" in typ:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPES[typ.replace("const ", "").strip()])
        if ret.replace(" ", "") == "constchar*":
            restype = ctypes.c_char_p
        else:
            restype = _CTYPES[ret]
        protos[name] = (restype, argtypes, argnames)
    return protos


_lock = threading.Lock()
_lib = None


def load():
    """Load the shared library (once) and attach prototypes.  Raises HipLibraryError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.isfile(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path.")
        # PyTorch bundles its own libamdhip64.so.7; it must be the ONE HIP runtime of the process (streams and
        # device pointers are shared with it), so make sure it is loaded before our library resolves that SONAME.
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes, _) in parse_header().items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise HipLibraryError(f"{LIB_PATH} does not export {name} declared in {HEADER_PATH}") from e
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.mmg_abi_version() != ABI_VERSION:
            raise HipLibraryError(f"ABI version mismatch: library reports {lib.mmg_abi_version()}, host expects {ABI_VERSION}")
        _lib = lib
    return _lib


def last_error():
    return load().mmg_last_error().decode()


def call(name, *args):
    """Call an int-returning entry point; raise RuntimeError with mmg_last_error() on a non-zero return."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib.mmg_last_error().decode()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mmgclip HIP kernels need tensors on the MI355X (got a CPU tensor); "
                               "the hot path has no CPU fallback")
