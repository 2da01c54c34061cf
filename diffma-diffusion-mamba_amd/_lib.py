"""ctypes binding of libdiffma_hip.so (the C ABI declared in include/diffma_hip.h).

The ctypes Structures are generated from the header text itself, so the Python view of every args
struct cannot drift from the C one.  There is deliberately NO fallback: if the shared library is
missing or a call fails, the caller gets an exception (the product path never routes through the
CPU oracle).
"""
from __future__ import annotations

import ctypes
import os
import re
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "diffma_hip.h")
# DIFFMA_HIP_LIB: developer override used to A/B two builds of the kernels in one GPU session (tools/ab.sh)
LIB_PATH = os.environ.get("DIFFMA_HIP_LIB") or os.path.join(_HERE, "csrc", "libdiffma_hip.so")

DM_F32, DM_BF16, DM_F16 = 0, 1, 2
DM_FLAG_DELTA_SOFTPLUS = 1
DM_FLAG_SILU = 2
DM_FLAG_DOUT_PER_SEQ = 4
DM_FLAG_A_SHARED = 8
DM_FLAG_SCAN_SEQUENTIAL = 16
DM_FLAG_SCAN_CHUNKED = 32
DM_FLAG_OUT_ACCUMULATE = 64
DM_FLAG_DELTA_ACTIVATED = 128
DM_FLAG_DX_MERGED = 256
DM_FLAG_PARTIAL_COMPACT = 512

_SCALARS = {"int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double}


def _parse_structs(text: str):
    """Return {struct_name: [(field, ctype), ...]} for every `typedef struct { ... } name;`."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(const\s+)?(\w+)\s+(.*)$", decl)
            if not m:
                raise RuntimeError(f"cannot parse field declaration {decl!r} in {name}")
            base, rest = m.group(2), m.group(3)
            for item in rest.split(","):
                item = item.strip()
                is_ptr = item.startswith("*")
                fname = item.lstrip("* ").strip()
                if is_ptr:
                    fields.append((fname, ctypes.c_void_p))
                else:
                    if base not in _SCALARS:
                        raise RuntimeError(f"unknown scalar type {base!r} in {name}.{fname}")
                    fields.append((fname, _SCALARS[base]))
        out[name] = fields
    return out


def _parse_functions(text: str):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = re.findall(r"^\s*((?:const\s+)?\w+\s*\*?)\s*(dm_\w+)\s*\(([^)]*)\)\s*;", text, flags=re.M)
    return [(ret.strip(), name, args.strip()) for ret, name, args in protos]


with open(HEADER) as _f:
    _HEADER_TEXT = _f.read()

STRUCT_FIELDS = _parse_structs(_HEADER_TEXT)
FUNCTIONS = _parse_functions(_HEADER_TEXT)
EXPORTED_SYMBOLS = [name for _, name, _ in FUNCTIONS]


def _make_struct(name):
    return type(name, (ctypes.Structure,), {"_fields_": STRUCT_FIELDS[name]})


dm_scan_fwd_args = _make_struct("dm_scan_fwd_args")
dm_scan_bwd_args = _make_struct("dm_scan_bwd_args")
dm_conv_fwd_args = _make_struct("dm_conv_fwd_args")
dm_conv_bwd_args = _make_struct("dm_conv_bwd_args")
dm_conv_xproj_fwd_args = _make_struct("dm_conv_xproj_fwd_args")
dm_conv_xproj_bwd_args = _make_struct("dm_conv_xproj_bwd_args")
dm_merge_args = _make_struct("dm_merge_args")
dm_gate_bwd_args = _make_struct("dm_gate_bwd_args")
dm_dtproj_args = _make_struct("dm_dtproj_args")
dm_dtproj_bwd_args = _make_struct("dm_dtproj_bwd_args")
dm_ln_mod_args = _make_struct("dm_ln_mod_args")
dm_blend_args = _make_struct("dm_blend_args")
dm_gate_head_args = _make_struct("dm_gate_head_args")
dm_rmsnorm_merge_args = _make_struct("dm_rmsnorm_merge_args")
dm_colsum_args = _make_struct("dm_colsum_args")
dm_sum_partials_args = _make_struct("dm_sum_partials_args")
dm_diffusion_step_args = _make_struct("dm_diffusion_step_args")
dm_ssd_fwd_args = _make_struct("dm_ssd_fwd_args")
dm_ssd_bwd_args = _make_struct("dm_ssd_bwd_args")
dm_gemm_args = _make_struct("dm_gemm_args")
dm_repack_args = _make_struct("dm_repack_args")
dm_adamw_tensor = _make_struct("dm_adamw_tensor")
dm_adamw_args = _make_struct("dm_adamw_args")
dm_training_loss_args = _make_struct("dm_training_loss_args")

_lib = None
_lock = threading.Lock()


class DiffmaHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.isfile(LIB_PATH):
            raise DiffmaHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C diffma-diffusion-mamba_amd/csrc`.  There is no CPU fallback."
            )
        # torch must own the HIP runtime of the process: importing it first makes libamdhip64.so.7
        # resolve to the copy torch already mapped, so streams/pointers are interchangeable.
        import torch  # noqa: F401

        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for ret, name, args in FUNCTIONS:
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = ctypes.c_char_p if "char" in ret else ctypes.c_int
            if args in ("void", ""):
                fn.argtypes = []
            elif name in ("dm_conv_nchunk", "dm_scan_bwd_group_channels", "dm_gather_conv1d_xproj_width_supported"):   # int -> int helpers
                fn.argtypes = [ctypes.c_int]
            elif name in ("dm_ssd_fwd_supported", "dm_ssd_bwd_supported"):
                fn.argtypes = [ctypes.c_int] * 4
            elif name == "dm_scan_bwd_launch_group_channels":
                fn.argtypes = [ctypes.c_int] * 5
            elif name in ("dm_gather_conv1d_xproj_supported", "dm_gather_conv1d_xproj_bwd_supported", "dm_dtproj_softplus_supported", "dm_dtproj_bwd_supported"):
                fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
            elif name in ("dm_gemm_supported", "dm_gemm_large_supported"):
                fn.argtypes = [ctypes.c_int] * 7
            elif name.endswith("_n"):                    # an array of n argument structs (several congruent launches in one)
                fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
            else:
                fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        got = lib.dm_abi_version()
        want = int(re.search(r"#define\s+DM_ABI_VERSION\s+(\d+)", _HEADER_TEXT).group(1))
        if got != want:
            raise DiffmaHipError(f"libdiffma_hip.so ABI {got} != header ABI {want}; rebuild the library")
        _lib = lib
    return _lib


def check(status: int, what: str):
    if status != 0:
        msg = load().dm_last_error().decode(errors="replace")
        raise DiffmaHipError(f"{what} failed with status {status}: {msg}")


def call(name: str, args_struct, stream_handle: int):
    lib = load()
    status = getattr(lib, name)(ctypes.byref(args_struct), ctypes.c_void_p(stream_handle))
    check(status, name)


def call_n(name: str, args_structs, stream_handle: int):
    """`name`_n(args[0..n), n, stream): the structs must be of one ctypes type; congruent neighbours share a launch."""
    lib = load()
    arr = (type(args_structs[0]) * len(args_structs))(*args_structs)
    status = getattr(lib, name + "_n")(ctypes.cast(arr, ctypes.c_void_p), len(args_structs), ctypes.c_void_p(stream_handle))
    check(status, name + "_n")


HAS_N = None


def has_n(name: str) -> bool:
    global HAS_N
    if HAS_N is None:
        HAS_N = {n[:-2] for n in EXPORTED_SYMBOLS if n.endswith("_n")}
    return name in HAS_N


def build_info() -> str:
    return load().dm_build_info().decode()
