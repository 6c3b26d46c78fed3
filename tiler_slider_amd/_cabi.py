"""ctypes binding of lib/libtiler_slider_hip.so — the C-ABI declared in include/tiler_slider.h.

There is no CPU fallback: if the shared library is missing or does not load, every entry
point raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`tiler_slider_amd.build_library()` (hipcc --offload-arch=gfx950).
"""
import ctypes as C
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
SRC = os.path.join(_PKG, "csrc", "ts_kernels.hip")
HEADERS = [os.path.join(_PKG, "csrc", "ts_core.h"), os.path.join(ROOT, "include", "tiler_slider.h")]
LIB_PATH = os.path.join(_PKG, "lib", "libtiler_slider_hip.so")

ABI_VERSION = 3
OK, ERR_NULL, ERR_DIMS, ERR_LIMIT, ERR_HIP, ERR_ARG = 0, -1, -2, -3, -4, -5
FLAG_IS_WON, FLAG_INVALID_MOVE, FLAG_SUCCESS, FLAG_TIMEOUT = 0x01, 0x02, 0x04, 0x08
FLAG_STEPPED_DONE, FLAG_AUTORESET, FLAG_BAD_ACTION = 0x10, 0x20, 0x40
MODE_STRICT, MODE_AUTORESET = 0, 1
TUNE_MULTI_MIN_BOARDS, TUNE_NT_THRESHOLD_BYTES = 0, 1

EXPORTS = ("ts_abi_version", "ts_limits", "ts_status_string", "ts_last_hip_error", "ts_blk_words", "ts_cell_bytes",
           "ts_onehot_channels", "ts_check_dims", "ts_reset", "ts_step", "ts_valid_moves", "ts_is_won", "ts_encode",
           "ts_encode_u8", "ts_expand_u8", "ts_encode_onehot", "ts_reward", "ts_generate", "ts_fill_actions",
           "ts_lines_words", "ts_prepare", "ts_generate_mt19937", "ts_tuning")


class Dims(C.Structure):
    _fields_ = [("n_boards", C.c_int64), ("size", C.c_int32), ("n_tiles", C.c_int32), ("n_targets", C.c_int32),
                ("multi_color", C.c_int32), ("max_steps", C.c_int32), ("launch_hint", C.c_int32)]


class State(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("init", C.c_void_p), ("tgt", C.c_void_p), ("blk", C.c_void_p),
                ("step_count", C.c_void_p), ("done", C.c_void_p), ("lines", C.c_void_p)]


class StepOut(C.Structure):
    _fields_ = [("flags", C.c_void_p), ("obs", C.c_void_p), ("reward", C.c_void_p), ("onehot", C.c_void_p),
                ("valid", C.c_void_p), ("obs_u8", C.c_void_p)]


class TilerSliderLibraryError(RuntimeError):
    pass


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > built for p in [SRC] + HEADERS)


def build_library(force=False, verbose=False):
    """Compile the HIP kernels for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise TilerSliderLibraryError("hipcc not found: cannot build libtiler_slider_hip.so")
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    tmp = LIB_PATH + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall", "-o", tmp, SRC]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise TilerSliderLibraryError(f"hipcc failed ({res.returncode}): {' '.join(cmd)}")
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


_lib = None


def lib():
    """The loaded shared library; raises (never falls back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TilerSliderLibraryError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback.")
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise TilerSliderLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    missing = [s for s in EXPORTS if not hasattr(L, s)]
    if missing:
        raise TilerSliderLibraryError(f"{LIB_PATH} lacks symbols {missing}; rebuild it")
    P, DP, SP = C.c_void_p, C.POINTER(Dims), C.POINTER(State)
    L.ts_abi_version.restype = C.c_int32
    L.ts_limits.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.ts_limits.restype = None
    L.ts_status_string.argtypes = [C.c_int32]
    L.ts_status_string.restype = C.c_char_p
    L.ts_last_hip_error.restype = C.c_int32
    L.ts_blk_words.argtypes = [C.c_int32]
    L.ts_blk_words.restype = C.c_int32
    L.ts_cell_bytes.argtypes = [C.c_int32]
    L.ts_cell_bytes.restype = C.c_int32
    L.ts_lines_words.argtypes = [C.c_int32]
    L.ts_lines_words.restype = C.c_int32
    L.ts_tuning.argtypes = [C.c_int32, C.c_int64]
    L.ts_tuning.restype = C.c_int64
    L.ts_onehot_channels.argtypes = [DP]
    L.ts_onehot_channels.restype = C.c_int32
    L.ts_check_dims.argtypes = [DP]
    L.ts_check_dims.restype = C.c_int32
    for name, args in (("ts_reset", [DP, SP, P, P]),
                       ("ts_step", [DP, SP, P, C.c_uint32, C.POINTER(StepOut), P]),
                       ("ts_valid_moves", [DP, SP, P, P]), ("ts_is_won", [DP, SP, P, P]), ("ts_prepare", [DP, SP, P, P]),
                       ("ts_encode", [DP, SP, P, P]), ("ts_encode_u8", [DP, SP, P, P]),
                       ("ts_encode_onehot", [DP, SP, P, P]), ("ts_reward", [DP, SP, P, P]),
                       ("ts_generate", [DP, SP, C.c_uint64, C.c_int64, C.c_int32, P]),
                       ("ts_generate_mt19937", [DP, SP, P, C.c_int32, P]),
                       ("ts_fill_actions", [C.c_int64, C.c_uint64, C.c_int64, C.c_int64, P, P]),
                       ("ts_expand_u8", [P, P, C.c_int64, P])):
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int32
    if L.ts_abi_version() != ABI_VERSION:
        raise TilerSliderLibraryError(f"ABI version {L.ts_abi_version()} != {ABI_VERSION}; rebuild the library")
    _lib = L
    return L


def check(rc, what):
    if rc != OK:
        L = lib()
        msg = L.ts_status_string(rc).decode()
        extra = f" (hipError {L.ts_last_hip_error()})" if rc == ERR_HIP else ""
        raise TilerSliderLibraryError(f"{what}: {msg}{extra}")


def limits():
    a, b = C.c_int32(), C.c_int32()
    lib().ts_limits(C.byref(a), C.byref(b))
    return a.value, b.value
