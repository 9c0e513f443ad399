"""ctypes binding of libmgcmt_hip.so (the C-ABI declared in include/mgcmt_hip.h).

The product path has exactly one implementation: the HIP library built for gfx950.  If it is not
there, or it cannot see a GPU, every compute entry point raises — there is no CPU fallback.
(``use_library`` exists so the test-suite can point the binding at the host-only emulation build of
the SAME kernel sources, tests/hip_cpu_mock; nothing in this package calls it.)
"""
import ctypes
import os
import sys
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIBRARY = os.path.join(_HERE, "libmgcmt_hip.so")

WJACOBI, GS_LEX, SOR_LEX, GS_MC = 0, 1, 2, 3
SLOT_V, SLOT_F, SLOT_T, SLOT_W = 0, 1, 2, 3
OP_A, OP_M = 0, 1
HALO_ROWS = 16         # 2-D levels (a 1-D level keeps one halo "row"); exchanged rows per level: Plan.level_halo
RQ_HISTORY = 4096      # MGCMT_RQ_HISTORY: Rayleigh quotients Plan.rq_line_step can record on the device
MAX_TERMS = 4
MAX_VEC = 32
ABI_VERSION = 6
OPT_FUSED = 0
OPT_FUSED_ROWS = 1
OPT_TAIL = 4
OPT_LEX_WAVE = 5
OPT_LEX_CHAIN = 6
OPT_MGS_BLOCK = 7
OPT_GRAPH = 2
OPT_RECOMPUTE = 3


class MgcmtError(RuntimeError):
    pass


class PlanDesc(ctypes.Structure):
    _fields_ = [
        ("dim", c_int32), ("nterms", c_int32), ("g", c_int64), ("lowest", c_int64),
        ("xfac", POINTER(c_double)), ("yfac", POINTER(c_double)),
        ("m_nterms", c_int32), ("m_xfac", POINTER(c_double)), ("m_yfac", POINTER(c_double)),
        ("nvec", c_int32), ("device", c_int32), ("row_begin", c_int64), ("row_end", c_int64),
        ("strip_levels", c_int32), ("reserved", c_int32),
    ]


_dp = POINTER(c_double)


class P2POp(ctypes.Structure):
    _fields_ = [("ptr", c_void_p), ("count", c_int64), ("peer", c_int32), ("is_send", c_int32)]


P2P_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, POINTER(P2POp))
ALLGATHER_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_int64)
ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, _dp, c_int)
UNIQUE_ID_BYTES = 128
COMM_OPT_OVERLAP, COMM_OPT_SPLIT, COMM_OPT_SELF_RING, COMM_OPT_EMULATE_OF = 0, 1, 2, 3
SHARDED_V_HALO_VALID, SHARDED_F_HALO_VALID, SHARDED_GRAM_SCHMIDT = 1, 2, 4
CYCLE_GRAM_SCHMIDT, CYCLE_ZERO_START = 1, 2    # cycle_flags of mgcmt_vcycle
OPK_GENERAL, OPK_FIVE_POINT, OPK_FIVE_DIAG, OPK_NINE_CONST, OPK_NINE_VAR = 0, 1, 2, 3, 4    # mgcmt_level_operator_kind
HALO_RING = 0x100

_SIGNATURES = {
    "mgcmt_last_error": (c_char_p, []),
    "mgcmt_abi_version": (c_int, []),
    "mgcmt_device_count": (c_int, [POINTER(c_int)]),
    "mgcmt_device_name": (c_int, [c_int, ctypes.c_char_p, c_int]),
    "mgcmt_plan_create": (c_int, [POINTER(PlanDesc), POINTER(c_void_p)]),
    "mgcmt_plan_destroy": (c_int, [c_void_p]),
    "mgcmt_plan_num_levels": (c_int, [c_void_p, POINTER(c_int)]),
    "mgcmt_plan_level_shape": (c_int, [c_void_p, c_int, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "mgcmt_plan_get_factors": (c_int, [c_void_p, c_int, c_int, c_int, _dp, c_int64]),
    "mgcmt_vec_ptr": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_void_p)]),
    "mgcmt_plan_level_halo": (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int)]),
    "mgcmt_set_shifts": (c_int, [c_void_p, _dp, c_int, c_void_p]),
    "mgcmt_upload": (c_int, [c_void_p, c_int, c_int, c_int, _dp, c_int64, c_void_p]),
    "mgcmt_download": (c_int, [c_void_p, c_int, c_int, c_int, _dp, c_int64, c_void_p]),
    "mgcmt_fill": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_void_p]),
    "mgcmt_zero": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "mgcmt_copy": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mgcmt_sync": (c_int, [c_void_p]),
    "mgcmt_host_alloc": (c_int, [c_int64, POINTER(c_void_p)]),
    "mgcmt_host_free": (c_int, [c_void_p]),
    "mgcmt_smooth": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_int, c_void_p]),
    "mgcmt_residual_restrict": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "mgcmt_prolong_correct": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "mgcmt_coarse_solve": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "mgcmt_vcycle": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_double, c_int, c_int, c_void_p]),
    "mgcmt_twogrid": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_double, c_int, c_void_p]),
    "mgcmt_apply": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mgcmt_restrict": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mgcmt_prolong": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mgcmt_dot": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, _dp, c_void_p]),
    "mgcmt_axpy": (c_int, [c_void_p, c_int, c_double, c_int, c_int, c_int, c_int, c_void_p]),
    "mgcmt_gram": (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), _dp, c_void_p]),
    "mgcmt_rayleigh_residual": (c_int, [c_void_p, c_int, c_int, c_int, _dp, _dp, c_void_p]),
    "mgcmt_ritz_pair": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _dp, c_void_p]),
    "mgcmt_rqmin": (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_int), c_int, c_int, _dp, c_void_p]),
    "mgcmt_vcycle_rqmg": (c_int, [c_void_p, c_int, ctypes.POINTER(c_int), c_int, c_int, c_int, _dp, c_void_p]),
    "mgcmt_rq_line_step": (c_int, [c_void_p, c_int] + [ctypes.POINTER(c_int)] * 5 + [c_int, c_int, c_void_p]),
    "mgcmt_rq_history": (c_int, [c_void_p, c_int, c_int, _dp, c_void_p]),
    "mgcmt_lincomb": (c_int, [c_void_p, c_int, c_int, _dp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int, c_int, c_void_p]),
    "mgcmt_scale": (c_int, [c_void_p, c_int, c_double, c_int, c_int, c_void_p]),
    "mgcmt_block_gram": (c_int, [c_void_p, c_int, c_int, POINTER(c_int), POINTER(c_int), c_int, POINTER(c_int), POINTER(c_int),
                                 POINTER(c_double), c_void_p]),
    "mgcmt_block_combine": (c_int, [c_void_p, c_int, c_int, POINTER(c_int), POINTER(c_int), c_int, POINTER(c_int), POINTER(c_int),
                                    POINTER(c_double), c_void_p]),
    "mgcmt_gramschmidt": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mgcmt_normalize": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "mgcmt_level_operator_kind": (c_int, [c_void_p, c_int, ctypes.POINTER(c_int)]),
    "mgcmt_fused_pass": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_int, c_int, c_void_p]),
    "mgcmt_fused_max_sweeps": (c_int, [c_void_p, c_int, c_int, POINTER(c_int)]),
    "mgcmt_fused_max_recompute": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_int)]),
    "mgcmt_comm_unique_id": (c_int, [c_void_p]),
    "mgcmt_comm_init": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "mgcmt_comm_init_external": (c_int, [c_void_p, c_int, c_int, P2P_FN, ALLGATHER_FN, ALLREDUCE_FN, c_void_p]),
    "mgcmt_comm_destroy": (c_int, [c_void_p]),
    "mgcmt_comm_set_option": (c_int, [c_void_p, c_int, c_int]),
    "mgcmt_halo_exchange": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "mgcmt_gather_coarse": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "mgcmt_allreduce_sum": (c_int, [c_void_p, _dp, c_int, c_void_p]),
    "mgcmt_sharded_vcycle": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_int, c_int, c_void_p]),
    "mgcmt_csr_plan_create": (c_int, [c_int, c_int64, c_int64, POINTER(c_int64), POINTER(c_int32), _dp, POINTER(c_void_p)]),
    "mgcmt_csr_plan_destroy": (c_int, [c_void_p]),
    "mgcmt_csr_num_levels": (c_int, [c_void_p, POINTER(c_int)]),
    "mgcmt_csr_level_info": (c_int, [c_void_p, c_int, POINTER(c_int64), POINTER(c_int64), POINTER(c_int32)]),
    "mgcmt_csr_get_matrix": (c_int, [c_void_p, c_int, POINTER(c_int64), POINTER(c_int32), _dp]),
    "mgcmt_csr_upload": (c_int, [c_void_p, c_int, c_int, _dp, c_int64, c_void_p]),
    "mgcmt_csr_download": (c_int, [c_void_p, c_int, c_int, _dp, c_int64, c_void_p]),
    "mgcmt_csr_apply": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_void_p]),
    "mgcmt_csr_smooth": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_double, c_void_p]),
    "mgcmt_csr_vcycle": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_void_p]),
    "mgcmt_plan_set_option": (c_int, [c_void_p, c_int, c_int]),
    "mgcmt_bandwidth_probe": (c_int, [c_void_p, c_int, c_int, c_int, c_int, _dp, c_void_p]),
    "mgcmt_lex_wave_stats": (c_int, [c_void_p, POINTER(ctypes.c_uint32), c_int64]),
    "mgcmt_time_smoother": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_int, _dp, c_void_p]),
    "mgcmt_time_fused_pass": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_int, c_int, _dp, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))

_lib = None
_lib_path = None
_torch_first = None        # was torch already imported when the library was bound?
_torch_checked = False


def _check_torch_order():
    """Each of libmgcmt_hip.so and PyTorch brings a HIP runtime and only the first one loaded can open the device: a
    process that imports torch AFTER its first call into this library is left with a torch that sees no GPU
    ("Ownership") — silently.  Detected here on the first library use after such an import and raised
    once, with the remedy; MGCMT_ALLOW_TORCH_AFTER=1 silences it (a process that only wants CPU tensors)."""
    global _torch_checked
    if _torch_checked or _torch_first is not False or "torch" not in sys.modules:
        return
    if _lib_path != DEFAULT_LIBRARY:       # (a test binding — the host-only emulation build has no HIP runtime to conflict with)
        return
    _torch_checked = True
    if os.environ.get("MGCMT_ALLOW_TORCH_AFTER", "0") == "1":
        return
    torch = sys.modules["torch"]
    try:
        torch_sees_gpu = bool(torch.cuda.is_available())
        n = c_int(0)
        ours = _lib.mgcmt_device_count(ctypes.byref(n)) == 0 and n.value >= 1
    except Exception:
        return
    if ours and not torch_sees_gpu:
        raise MgcmtError(
            "torch was imported after the first call into libmgcmt_hip.so and now sees no GPU (torch.cuda.is_available() is "
            "False while this library sees %d device(s)): each brings a HIP runtime and only the first one loaded can open "
            "the device.  Import torch BEFORE the first use of multigridcmt_amd in this process, or set "
            "MGCMT_ALLOW_TORCH_AFTER=1 if torch is only used on the CPU." % n.value)


def _bind(path):
    global _torch_first, _torch_checked
    if _torch_first is None:
        _torch_first = "torch" in sys.modules
        _torch_checked = _torch_first
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.mgcmt_abi_version() != ABI_VERSION:
        raise MgcmtError("ABI version mismatch: library %d, binding %d" % (lib.mgcmt_abi_version(), ABI_VERSION))
    return lib


def use_library(path):
    """Bind a specific build of the C-ABI (tests only)."""
    global _lib, _lib_path
    _lib = _bind(path)
    _lib_path = path
    return _lib


def library_path():
    return _lib_path or DEFAULT_LIBRARY


def lib():
    """The bound library; raises MgcmtError when libmgcmt_hip.so has not been built."""
    global _lib, _lib_path
    if _lib is None:
        if not os.path.exists(DEFAULT_LIBRARY):
            raise MgcmtError(
                "libmgcmt_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % DEFAULT_LIBRARY)
        _lib = _bind(DEFAULT_LIBRARY)
        _lib_path = DEFAULT_LIBRARY
    if not _torch_checked:
        _check_torch_order()
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().mgcmt_last_error()
        raise MgcmtError("mgcmt error %d: %s" % (rc, msg.decode() if msg else "?"))


def device_count():
    n = c_int(0)
    check(lib().mgcmt_device_count(ctypes.byref(n)))
    return n.value


def device_name(device=0):
    buf = ctypes.create_string_buffer(256)
    check(lib().mgcmt_device_name(device, buf, 256))
    return buf.value.decode()


def as_dp(array):
    return array.ctypes.data_as(_dp)
