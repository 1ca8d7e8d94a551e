"""ctypes binding of libembnet_hip.so (the C ABI in include/embnet.h).

The library is the product: if it is missing or does not load this module
raises — there is no CPU or eager-PyTorch fallback anywhere in the package.
Argument and return types are taken from the prototypes in include/embnet.h,
so the header is the single source of truth for the boundary.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EMBNET_LIB") or os.path.join(_HERE, "libembnet_hip.so")   # EMBNET_LIB: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "embnet.h")

_lib = None

_CTYPES = {"int": ctypes.c_int, "long": ctypes.c_long, "size_t": ctypes.c_size_t, "float": ctypes.c_float,
           "uint64_t": ctypes.c_uint64, "double": ctypes.c_double, "void": None}


class EmbnetError(RuntimeError):
    pass


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every prototype the header declares."""
    text = re.sub(r"/\*.*?\*/", " ", open(path).read(), flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r"\b(int|size_t|const char\*)\s+(embnet_\w+)\s*\(([^)]*)\)\s*;", text):
        argtypes = []
        for a in [x.strip() for x in args.split(",")]:
            if a in ("void", ""):
                continue
            if "*" in a:
                argtypes.append(ctypes.c_void_p)
            else:
                argtypes.append(_CTYPES[a.replace("const ", "").split()[0]])
        res = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p}[ret]
        protos[name] = (res, argtypes)
    return protos


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EmbnetError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C embeddingnet_amd/csrc`).  embeddingnet_amd has no fallback path.")
        l = ctypes.CDLL(LIB_PATH)
        # EMBNET_LIB_LAX=1 (A/B of an OLDER build through EMBNET_LIB, tools/exp): symbols the older library lacks are left
        # unbound — calling one raises AttributeError — and the ABI number is not compared
        lax = bool(os.environ.get("EMBNET_LIB")) and os.environ.get("EMBNET_LIB_LAX") == "1"
        for name, (res, argtypes) in parse_header().items():
            if lax and not hasattr(l, name):
                continue
            fn = getattr(l, name)          # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, argtypes
        if l.embnet_abi_version() != 22 and not lax:
            raise EmbnetError("libembnet_hip.so ABI version mismatch")
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise EmbnetError(f"embnet error {rc}: {lib().embnet_last_error().decode()}")


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise EmbnetError("embeddingnet_amd ops run on the GPU only (got a CPU tensor)")
    if not t.is_contiguous():
        raise EmbnetError("tensor must be contiguous")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """Raw handle of torch's current stream on the current device.  (torch.cuda.current_stream() builds a Stream object:
    ~8 us per call, 90 calls per ResNet18 step.)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def f32(x):
    return float(x)


# ---- per-kernel timing (embnet_trace_*: HIP events inside the library around every kernel launch) ----------------
_trace_on = False


def trace_enable(on):
    global _trace_on
    _trace_on = bool(on)
    lib().embnet_trace_enable(int(bool(on)))


def trace_is_enabled():
    return _trace_on


def trace_reset():
    lib().embnet_trace_reset()


def trace_records():
    """-> [(kernel name, ms, algorithmic work, unit, algorithmic bytes)] — unit 0: FLOP, 1: bytes.  Waits for the
    recorded events."""
    l = lib()
    name = ctypes.create_string_buffer(160)
    ms, work, unit, nbytes = ctypes.c_float(), ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
    out = []
    for i in range(l.embnet_trace_count()):
        check(l.embnet_trace_get(i, ctypes.addressof(name), 160, ctypes.addressof(ms), ctypes.addressof(work),
                                 ctypes.addressof(unit), ctypes.addressof(nbytes)))
        out.append((name.value.decode(), ms.value, work.value, unit.value, nbytes.value))
    return out
