"""ctypes binding of libuz_hip.so (C ABI declared in include/uz_api.h).

The library is loaded lazily; ``lib()`` raises a clear error when it has not been built
(``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C unet-zoo_amd/csrc``).
There is deliberately no fallback implementation.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# UZ_LIB: an experiment build of the same library (csrc/Makefile VARIANT=...); never a different implementation
LIB_PATH = os.environ.get("UZ_LIB") or os.path.join(_HERE, "libuz_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "uz_api.h")

_lib = None


class UzError(RuntimeError):
    pass


UZ_MAX_LANES = 8


class uz_sched(C.Structure):
    _fields_ = [("lane", C.c_int32), ("signal", C.c_int32), ("n_wait", C.c_int32), ("wait", C.c_int32 * UZ_MAX_LANES)]


class uz_op(C.Structure):
    _fields_ = [("code", C.c_int32), ("i", C.c_int32 * 15), ("f", C.c_float * 4), ("n", C.c_int64), ("p", C.c_void_p * 12)]


class uz_chain_op(C.Structure):
    _fields_ = [("code", C.c_int32), ("tile0", C.c_int32), ("ntiles", C.c_int32), ("rsv", C.c_int32),
                ("i", C.c_int32 * 16), ("f", C.c_float * 4), ("p", C.c_void_p * 12)]


def chain_codes():
    """The UZ_CH_* enum of the header (sub-ops of uz_chain_run)."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    body = next(b for b in re.findall(r"enum\s*\{(.*?)\};", text, flags=re.S) if "UZ_CH_" in b)
    codes, nxt = {}, 0
    for tok in body.split(","):
        tok = tok.strip()
        if not tok:
            continue
        if "=" in tok:
            name, val = [t.strip() for t in tok.split("=")]
            nxt = int(val)
        else:
            name = tok
        codes[name] = nxt
        nxt += 1
    return codes


def header_symbols():
    """Every function name declared in include/uz_api.h."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(uz_[a-z0-9_]+)\s*\(", text)))


def op_codes():
    """Parse the UZ_OP_* enum from the header (single source of truth for the tape encoding)."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    body = re.search(r"enum\s*\{(.*?)\};", text, flags=re.S).group(1)
    codes, nxt = {}, 0
    for tok in body.split(","):
        tok = tok.strip()
        if not tok:
            continue
        if "=" in tok:
            name, val = [t.strip() for t in tok.split("=")]
            nxt = int(val)
        else:
            name = tok
        codes[name] = nxt
        nxt += 1
    return codes


def prototypes():
    """Parse every function declaration of include/uz_api.h into (restype, [argtypes]) so that the
    header stays the single source of truth for the binding."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    text = re.sub(r"typedef struct (\w+) \{.*?\} \1;", "", text, flags=re.S)
    text = re.sub(r"enum\s*\{.*?\};", "", text, flags=re.S)
    text = "\n".join(ln for ln in text.splitlines() if not ln.lstrip().startswith("#") and 'extern "C"' not in ln)
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(uz_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()

        def ctype(t):
            t = t.strip()
            if "*" in t:
                return C.c_char_p if t.replace(" ", "") == "constchar*" else C.c_void_p
            base = t.replace("const", "").split()[0] if t.split() else "void"
            return {"int": C.c_int, "size_t": C.c_size_t, "int64_t": C.c_int64, "float": C.c_float, "void": None, "uint8_t": C.c_uint8}[base]

        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                # drop the parameter name (last identifier) unless the declaration is a bare type
                mm = re.match(r"(.*?[\*\s])([A-Za-z_]\w*)$", a)
                argtypes.append(ctype(mm.group(1) if mm else a))
        out[name] = (ctype(ret), argtypes)
    return out


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # libuz_hip.so resolves libamdhip64 by soname: when torch is already imported that is the HIP runtime torch ships and has mapped
    # (ONE runtime per process); loaded FIRST it would pull in the system's copy, and the process would then hold two HIP runtimes -
    # the second one finds no device ("no ROCm-capable device is detected").  So: torch first, always.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise UzError(f"{LIB_PATH} not found: build the HIP library first (make -C unet-zoo_amd/csrc). "
                      "There is no CPU fallback for the product path.")
    L = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in prototypes().items():
        fn = getattr(L, name)
        fn.restype = restype
        fn.argtypes = argtypes
    L.uz_run_tape.argtypes = [C.POINTER(uz_op), C.c_int, C.c_void_p]
    L.uz_graph_create.argtypes = [C.POINTER(uz_op), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        raise UzError(f"{what}: {lib().uz_last_error().decode(errors='replace')}")
