"""The C-ABI library loads, exports every symbol include/hrseg.h declares, and the ctypes
prototypes in _lib.py agree with the header argument by argument (no compute calls: CPU only)."""
import ctypes
import os
import re

from tests.helpers import ROOT

HEADER = os.path.join(ROOT, "include", "hrseg.h")


def parse_header():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|long|size_t|const char\*)\s+(hrseg_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(3).replace("\n", " ").split(",")]
        decls[m.group(2)] = [] if args == ["void"] else args
    return decls


def ctype_of(arg):
    if "*" in arg or "hrseg_stream_t" in arg:
        return ctypes.c_void_p
    base = arg.rsplit(" ", 1)[0].strip()
    return {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "size_t": ctypes.c_size_t}[base]


def test_header_symbols_exported_and_prototypes_match():
    from hrseg_amd import _lib
    decls = parse_header()
    assert len(decls) >= 35
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name, args in decls.items():
        assert hasattr(lib, name), f"{name} declared in hrseg.h but not exported"
        if name in ("hrseg_last_error_string", "hrseg_abi_version", "hrseg_tune", "hrseg_conv_wgrad_workspace_bytes",
                    "hrseg_launch_count", "hrseg_conv_x_split_ok"):
            continue
        protos = _lib.RAW_PROTOTYPES if name in _lib.RAW_PROTOTYPES else _lib.PROTOTYPES
        assert name in protos, f"{name} has no ctypes prototype"
        want = [ctype_of(a) for a in args]
        got = list(protos[name])
        got = [ctypes.c_void_p if (isinstance(t, type) and issubclass(t, ctypes._Pointer)) else t for t in got]
        assert got == want, f"{name}: ctypes {got} != header {want}"
    for name in list(_lib.PROTOTYPES) + list(_lib.RAW_PROTOTYPES):
        assert name in decls, f"{name} bound in _lib.py but missing from hrseg.h"
    assert _lib.abi_version() == _lib.ABI_VERSION == 15
    assert not any(n.startswith("hrseg_debug_") for n in decls), "experimental switches do not belong in the public header"


def test_invalid_arguments_are_reported_without_a_gpu():
    from hrseg_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.hrseg_last_error_string.restype = ctypes.c_char_p
    shape = _lib.ConvShape(B=1, Hi=8, Wi=8, Cin=16, ldx=16, Ho=7, Wo=8, Cout=16, ldy=16, ksize=3, stride=1)
    rc = lib.hrseg_conv_fwd(None, None, None, None, ctypes.byref(shape), None)
    assert rc == -1
    assert b"does not match" in lib.hrseg_last_error_string()
    lib.hrseg_set_scratch.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert lib.hrseg_set_scratch(None, 0) == 0                      # detach: always fine
    assert lib.hrseg_set_scratch(ctypes.c_void_p(4096), 1024) == -1 and b"at least 1 MiB" in lib.hrseg_last_error_string()
    assert lib.hrseg_set_scratch(ctypes.c_void_p(4097), 1 << 20) == -1 and b"aligned" in lib.hrseg_last_error_string()
    lib.hrseg_tune.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.hrseg_tune(b"igemm_wtm", 0) == 0
    assert lib.hrseg_tune(b"no_such_knob", 1) == -1 and b"unknown key" in lib.hrseg_last_error_string()
