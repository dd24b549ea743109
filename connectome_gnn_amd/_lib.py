"""ctypes binding of libcgnn_hip.so (the C ABI declared in include/cgnn.h).

This is the binding a maintainer of the reference would add: the reference
reaches its hot path only through ATen calls in connectome_gnn/models.py:40-152, so the
"FFI" is a ctypes stub that hands raw device pointers, sizes and the current HIP stream to
the library.  There is NO fallback: if the library is missing or fails to load, importing the
compute path raises (a CPU/eager substitute would void every parity claim).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# CGNN_LIB selects another build of the same ABI (diagnostic builds only, e.g. the stamped one)
LIB_PATH = os.environ.get("CGNN_LIB") or os.path.join(_HERE, "libcgnn_hip.so")
ABI_VERSION = 2

CGNN_OK, CGNN_EINVAL, CGNN_ELAUNCH, CGNN_EUNSUPPORTED = 0, -1, -2, -3
_ERR = {CGNN_EINVAL: "CGNN_EINVAL (bad argument)", CGNN_ELAUNCH: "CGNN_ELAUNCH (kernel launch failed)",
        CGNN_EUNSUPPORTED: "CGNN_EUNSUPPORTED (shape not covered by this build)"}

P, I32, I64, U64, F32, F64 = c_void_p, c_int32, c_int64, c_uint64, c_float, c_double


class CgnnTiles(ctypes.Structure):
    """Mirror of `struct cgnn_tiles` (include/cgnn.h): host block of device pointers."""
    _fields_ = [("num_nodes", c_int64), ("num_tiles", c_int32), ("max_tile_rows", c_int32),
                ("tile_ptr", c_void_p), ("tile_blk", c_void_p),
                ("blk_off_dst", c_void_p), ("ent_dst", c_void_p),
                ("blk_off_src", c_void_p), ("ent_src", c_void_p),
                ("dis", c_void_p)]


TP = ctypes.POINTER(CgnnTiles)


class CgnnL0Src(ctypes.Structure):
    """Mirror of `struct cgnn_l0src` (include/cgnn.h): layer 0's output in factored form."""
    _fields_ = [("P0", c_void_p), ("W0", c_void_p), ("b0", c_void_p), ("F0", c_int32)]


LP = ctypes.POINTER(CgnnL0Src)

BN_ACC_BYTES = 16448


class CgnnBnTail(ctypes.Structure):
    """Mirror of `struct cgnn_bn_tail` (include/cgnn.h): a layer's BatchNorm finalisation run by the last
    workgroup of the kernel that produces its sums (csrc/bn_tail.h) instead of by a launch of its own."""
    _fields_ = [("acc", c_void_p), ("count", ctypes.c_double), ("mode", c_int32), ("zero_coef", c_int32),
                ("gamma", c_void_p), ("beta", c_void_p), ("running_mean", c_void_p), ("running_var", c_void_p),
                ("momentum", ctypes.c_float), ("eps", ctypes.c_float), ("num_batches_tracked", c_void_p),
                ("bn_out", c_void_p), ("rng_state", c_void_p), ("rng_n", c_int32), ("reserved", c_int32),
                ("dgamma", c_void_p), ("dbeta", c_void_p), ("bwc", c_void_p)]


BP = ctypes.POINTER(CgnnBnTail)

DW_MAX_JOBS = 8


class CgnnDwJobs(ctypes.Structure):
    """Mirror of `struct cgnn_dw_jobs` (include/cgnn.h): several slab -> dW/db reductions, one launch."""
    _fields_ = [("n", c_int32), ("dw_slab", c_void_p * DW_MAX_JOBS), ("db_slab", c_void_p * DW_MAX_JOBS),
                ("rows", c_int32 * DW_MAX_JOBS), ("out_cols", c_int32 * DW_MAX_JOBS),
                ("take_cols", c_int32 * DW_MAX_JOBS), ("dW", c_void_p * DW_MAX_JOBS),
                ("db", c_void_p * DW_MAX_JOBS)]

ADAM_MAX_JOBS = 32


class CgnnAdamJobs(ctypes.Structure):
    """Mirror of `struct cgnn_adam_jobs` (include/cgnn.h): the tensors of one optimizer launch."""
    _fields_ = [("n", c_int32), ("numel", c_int64 * ADAM_MAX_JOBS), ("param", c_void_p * ADAM_MAX_JOBS),
                ("grad", c_void_p * ADAM_MAX_JOBS), ("exp_avg", c_void_p * ADAM_MAX_JOBS),
                ("exp_avg_sq", c_void_p * ADAM_MAX_JOBS)]


GATHER_MAX_JOBS = 8


class CgnnGatherJobs(ctypes.Structure):
    """Mirror of `struct cgnn_gather_jobs` (include/cgnn.h): row gathers sharing one id list."""
    _fields_ = [("n", c_int32), ("src", c_void_p * GATHER_MAX_JOBS), ("dst", c_void_p * GATHER_MAX_JOBS),
                ("row_bytes", c_int64 * GATHER_MAX_JOBS)]


REDUCE_MAX_JOBS = 8


class CgnnReduceJobs(ctypes.Structure):
    """Mirror of `struct cgnn_reduce_jobs` (include/cgnn.h): f64 slab -> f32 vector reductions, one launch."""
    _fields_ = [("n", c_int32), ("slab", c_void_p * REDUCE_MAX_JOBS), ("rows", c_int32 * REDUCE_MAX_JOBS),
                ("width", c_int32 * REDUCE_MAX_JOBS), ("out", c_void_p * REDUCE_MAX_JOBS)]


class DeferredReduce:
    """Collects (f64 slab [rows][width] -> f32 out [width]) reductions and issues them as one launch
    (cgnn_slab_reduce_f64_multi) -- e.g. the bias gradients of all layers at the end of a backward pass."""

    def __init__(self):
        self.items = []

    def add(self, slab, rows: int, width: int, out) -> None:
        self.items.append((slab, int(rows), int(width), out))

    def flush(self, stream_ptr) -> None:
        lib = load()
        for lo in range(0, len(self.items), REDUCE_MAX_JOBS):
            chunk = self.items[lo:lo + REDUCE_MAX_JOBS]
            jobs = CgnnReduceJobs()
            jobs.n = len(chunk)
            for i, (slab, rows, width, out) in enumerate(chunk):
                jobs.slab[i], jobs.rows[i], jobs.width[i], jobs.out[i] = slab.data_ptr(), rows, width, out.data_ptr()
            check(lib.cgnn_slab_reduce_f64_multi(jobs, stream_ptr), "cgnn_slab_reduce_f64_multi")
        self.items = []


# name -> (restype, argtypes).  Order and meaning follow include/cgnn.h exactly.
PROTOTYPES = {
    "cgnn_abi_version": (c_int, []),
    "cgnn_build_target": (c_char_p, []),
    "cgnn_csr_workspace_bytes": (I64, [I64, I64]),
    "cgnn_csr_build": (c_int, [P, P, I64, I64, P, P, P, P, P, P, P, P, I64, P]),
    "cgnn_csr_build_grouped": (c_int, [P, P, P, I32, I64, I64, I32, I32, P, P, P, P, P, P, P, P]),
    "cgnn_gcn_norm": (c_int, [P, P, I64, I64, P, P, P, P, P, P, P, P, P]),
    "cgnn_sage_norm": (c_int, [P, P, I64, I64, P, P, P, P, P, P, P, P]),
    "cgnn_aggregate_f32": (c_int, [P, P, P, P, P, P, P, I64, P, I64, I64, I32, P]),
    "cgnn_aggregate_acc_f32": (c_int, [P, P, P, P, P, P, P, I64, P, I64, I64, I32, P]),
    "cgnn_aggregate_tiled_f16": (c_int, [TP, I32, P, I64, I32, P, P, P, P, I64, P]),
    "cgnn_head_supported": (c_int, [I32, I32, I32]),
    "cgnn_head_grid": (c_int, [I32, I32, I32, I32]),
    "cgnn_head_fwd_f32": (c_int, [P, I32, I32, I32, I32, P, P, P, P, F32, U64, P, P, P, P, P]),
    "cgnn_head_bwd_f32": (c_int, [P, P, P, P, I32, I32, I32, I32, P, P, P, P, I64, P]),
    "cgnn_head_loss_grid": (c_int, [I32, I32, I32, I32]),
    "cgnn_head_loss_f32": (c_int, [P, I32, I32, I32, I32, P, P, P, P, P, F32, U64, P, P, P, P, P, P, I64, P]),
    "cgnn_cross_entropy_f32": (c_int, [P, P, I32, I32, P, P, P]),
    "cgnn_dense_adj_f16": (c_int, [P, P, P, P, P, I32, I32, P, P]),
    "cgnn_dense_aggregate_f16": (c_int, [P, I32, P, I32, P, I64, I32, P, P, I64, P, I64, P]),
    "cgnn_dense_pack_count": (c_int, [P, P, P, P, P, I32, I32, P, P]),
    "cgnn_dense_pack_fill": (c_int, [P, P, P, P, P, I32, I32, P, P, P, P, P, P]),
    "cgnn_dense_aggregate_c16": (c_int, [P, P, P, P, P, P, I32, P, I32, P, I64, I32, P, P, I64, P, I64, P]),
    "cgnn_band_pack_f32": (c_int, [P, P, P, P, I32, I32, P, P, P, P]),
    "cgnn_band_aggregate_f32": (c_int, [P, P, P, I32, P, I32, P, I64, I32, P, P, I64, P, I64, P]),
    "cgnn_dense_aggregate_c16_bnbwd": (c_int, [P, P, P, P, P, P, I32, P, I32, P, I64, P, P, I64, P, P, P, I32, F32, I32, P, I64, P, I64, P]),
    "cgnn_linear_fwd_f32": (c_int, [P, I64, I32, P, I64, I32, P, P, I32, P, I64, I64, I32, P]),
    "cgnn_linear_fwd_stats_f32": (c_int, [P, I64, I32, P, I64, I32, P, P, I32, P, I64, I64, I32, P, I64, P]),
    "cgnn_linear_bwd_input_f32": (c_int, [P, I64, P, I32, I32, P, I64, I64, I32, I32, P]),
    "cgnn_linear_bwd_weight_workspace_bytes": (I64, [I64, I32, I32]),
    "cgnn_linear_bwd_weight2_workspace_bytes": (I64, [I64, I32, I32, I32]),
    "cgnn_linear_bwd_weight_f32": (c_int, [P, I64, P, I64, P, I32, I32, I64, I32, I32, P, I64, P]),
    "cgnn_linear_bwd_weight2_f32": (c_int, [P, I64, P, I64, I32, P, I64, I32, P, I32, I64, I32, P, I64, P]),
    "cgnn_linear_fwd_f16": (c_int, [P, I64, I32, P, I32, I32, P, P, I64, I64, I32, P]),
    "cgnn_linear_bwd_input_f16": (c_int, [P, I64, P, I32, P, I64, I64, I32, I32, P]),
    "cgnn_linear_fwd_stats_f16": (c_int, [P, I64, I32, P, I32, I32, P, P, I64, I64, I32, P, I64, P]),
    "cgnn_linear_bwd_weight_f16_workspace_bytes": (I64, [I64, I32, I32]),
    "cgnn_linear_bwd_weight_f16": (c_int, [P, I64, P, I64, P, I32, I32, I64, I32, I32, P, I64, P]),
    "cgnn_pad_cast_f16": (c_int, [P, I64, I32, P, I32, I64, P]),
    "cgnn_colsum_workspace_bytes": (I64, [I64, I32]),
    "cgnn_colsum_f32": (c_int, [P, I64, P, I64, I32, P, I64, P]),
    "cgnn_pool_mean_fwd_f32": (c_int, [P, I64, P, P, I32, I32, P]),
    "cgnn_pool_mean_bwd_f32": (c_int, [P, P, P, I64, I32, I32, P]),
    # BatchNorm + activation + dropout (layered path)
    "cgnn_bn_act_width_ok": (c_int, [I32]),
    "cgnn_bn_act_slab_rows": (I64, [I64]),
    "cgnn_bn_act_fwd_stats": (c_int, [P, I64, I32, P, I64, P]),
    "cgnn_bn_act_finalize": (c_int, [P, I32, I32, F64, P, I32, P, P, P, P, F32, F32, P, P, P]),
    "cgnn_bn_act_fwd_apply": (c_int, [P, P, I32, F32, U64, P, P, P, I64, I32, P]),
    "cgnn_bn_act_pool_fwd": (c_int, [P, P, I32, F32, U64, P, P, P, I32, P, I32, P, P]),
    "cgnn_bn_act_pool_bwd_finalize": (c_int, [P, P, P, I32, I32, F64, I32, P, P, P, P]),
    "cgnn_bn_act_bwd_stats": (c_int, [P, P, P, P, I32, F32, I64, I32, P, I64, P, P, P, P]),
    "cgnn_bn_act_bwd_finalize": (c_int, [P, I32, I32, F64, P, I32, P, P, P, P]),
    "cgnn_bn_act_apply_blocks": (I64, [I64, I32]),
    "cgnn_bn_act_bwd_apply": (c_int, [P, P, P, P, P, I32, F32, I32, P, I64, P, I64, I32, P, P, P, P]),
    "cgnn_bn_act_fwd_stats_f16": (c_int, [P, I64, I32, P, I64, P]),
    "cgnn_bn_act_fwd_apply_f16": (c_int, [P, P, I32, F32, U64, P, P, P, I64, I32, P]),
    "cgnn_bn_act_pool_fwd_f16": (c_int, [P, P, I32, F32, U64, P, P, P, I32, P, I32, P, P]),
    "cgnn_bn_act_bwd_stats_f16": (c_int, [P, P, P, P, I32, F32, I64, I32, P, I64, P, P, P, P]),
    "cgnn_bn_act_bwd_apply_f16": (c_int, [P, P, P, P, P, I32, F32, I32, P, I64, P, I64, I32, P, P, P, P]),
    # fused per-tile GCN path
    "cgnn_bell_plan": (c_int, [P, P, I32, I32, P, P, P, I64, P]),
    "cgnn_bell_fill": (c_int, [P, P, I32, P, P, P, P, F32, P, P, P]),
    "cgnn_aggregate_tiled_f32": (c_int, [TP, I32, P, I64, I32, P, P, P, P, I64, P, I64, P]),
    "cgnn_aggregate_tiled_bn_f32": (c_int, [TP, I32, P, I64, I32, P, P, P, P, I64, P, I32, F32, U64, P, P, P, I64, P]),
    "cgnn_slab_reduce_f64_multi": (c_int, [ctypes.POINTER(CgnnReduceJobs), P]),
    "cgnn_gather_f32": (c_int, [P, P, I64, P, P]),
    "cgnn_gather_rows": (c_int, [ctypes.POINTER(CgnnGatherJobs), P, I32, P, P]),
    "cgnn_epoch_advance": (c_int, [P, I64, P, F32, P, P]),
    "cgnn_gcn_dis": (c_int, [P, P, I64, P, P]),
    "cgnn_fused_grid": (c_int, []),
    "cgnn_set_fused_grid": (c_int, [I32]),
    "cgnn_gcn_fused_fwd_first": (c_int, [TP, P, I32, P, P, P, P, I64, P]),
    "cgnn_rng_advance": (c_int, [P, I32, P]),
    "cgnn_gcn_fused_fwd": (c_int, [TP, P, LP, P, F32, U64, P, P, P, P, P, P, I64, BP, P]),
    "cgnn_bn_reduce": (c_int, [P, I32, I32, P, P]),
    "cgnn_bn_finalize": (c_int, [P, F64, P, P, P, P, P, F32, F32, I32, P, P, P]),
    "cgnn_gcn_fused_pool_fwd": (c_int, [P, P, F32, U64, P, P, P, I32, P, P, P, P]),
    "cgnn_gcn_fused_pool_bwd_sums": (c_int, [P, P, P, P, I32, P, I64, P]),
    "cgnn_gcn_fused_pool_bwd": (c_int, [P, P, P, F32, P, P, I32, P, P, I64, P]),
    "cgnn_bn_bwd_finalize": (c_int, [P, F64, P, I32, P, P, P, P]),
    "cgnn_gcn_fused_bwd": (c_int, [TP, P, P, P, P, P, LP, P, F32, P, P, P, P, I64, P, I64, P, I64, P, P, P, P, BP, P]),
    "cgnn_gcn_fused_bwd_first": (c_int, [TP, P, P, P, P, P, I32, P, I64, P, I64, F32, P, P, P, P, P]),
    "cgnn_bn_stats_finalize": (c_int, [P, I32, F64, P, P, P, P, F32, F32, P, P, P]),
    "cgnn_bn_stats_finalize_rng": (c_int, [P, I32, F64, P, P, P, P, F32, F32, P, P, P, I32, P, P]),
    "cgnn_gcn_fused_pool_bwd_finalize": (c_int, [P, P, P, P, I32, F64, I32, P, P, P, P]),
    "cgnn_bn_bwd_stats_finalize": (c_int, [P, I32, F64, I32, P, P, P, P]),
    "cgnn_dw_db_reduce": (c_int, [P, P, I32, I32, I32, P, I32, P, P]),
    "cgnn_dw_db_reduce_multi": (c_int, [ctypes.POINTER(CgnnDwJobs), P]),
    "cgnn_adam_step": (c_int, [ctypes.POINTER(CgnnAdamJobs), P, P, I32, F64, F64, F64, F64, F64, P]),
    "cgnn_l0_grid": (c_int, [c_int64]),
    "cgnn_gcn_l0_center": (c_int, [TP, P, I32, P, P]),
    "cgnn_gcn_l0_fwd": (c_int, [TP, P, I32, P, P, P, P, P, I64, P, P, P, BP, P]),
    "cgnn_gcn_l0_bwd": (c_int, [P, P, LP, P, P, P, I64, P, I64, P, I64, P, P]),
    "cgnn_slab_reduce_f32": (c_int, [P, I32, I32, I32, I32, P, I32, P]),
    "cgnn_slab_reduce_f64": (c_int, [P, I32, I32, P, P]),
    "cgnn_slab_reduce_f32_split": (c_int, [P, I32, I32, I32, P, P, P]),
}


class CgnnError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """Load libcgnn_hip.so once; raise loudly if it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CgnnError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C connectome_gnn_amd/csrc`.  connectome_gnn_amd has no "
            "CPU/eager fallback for the message-passing path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise CgnnError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    v = lib.cgnn_abi_version()
    if v != ABI_VERSION:
        raise CgnnError(f"libcgnn_hip.so ABI {v} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != CGNN_OK:
        raise CgnnError(f"{what} failed: {_ERR.get(rc, rc)}")


def ptr(t) -> int:
    """Device pointer of a tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def nbytes(t) -> int:
    """Byte count of a tensor handed to the library as a scratch / partial-sum buffer (None -> 0):
    ABI 2 passes it right after the pointer and the library refuses a buffer that is too short."""
    return 0 if t is None else t.numel() * t.element_size()


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL = _NullCtx()


def device_guard(device):
    """``with device_guard(dev):`` == ``with torch.cuda.device(dev):`` but free when `dev` already
    is the current device (the context manager costs ~10 us per use on the launch path)."""
    import torch
    idx = device.index if getattr(device, "index", None) is not None else torch.cuda.current_device()
    if torch.cuda.current_device() == idx:
        return _NULL
    return torch.cuda.device(idx)


def stream_ptr(device=None) -> int:
    """The HIP stream torch is currently enqueuing on for `device` (default: the current device);
    kernels join torch's ordering."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


# ---------------------------------------------------------------------------------------------
# Dropout seeds.  The mask kernels are counter-based hashes keyed by a 62-bit seed per launch.
# A seed is (seed, offset) of the device's default torch CUDA generator, whose offset is then
# advanced -- what torch's own CUDA dropout consumes.  So ``torch.manual_seed(s)`` makes the
# dropout draw reproducible, and the global CPU generator, from which the loaders' shuffles draw
# (graph.py, reference graph.py:192-194), is never touched: it stays in lockstep across
# data-parallel ranks whatever execution path each rank's shard takes.  While a stream is being
# captured the generator may not be queried; the seed (frozen into the graph anyway, fresh masks
# per replay come from the device key word, graphed.py) then comes from a private CPU generator.
# ---------------------------------------------------------------------------------------------
_capture_gen = None


def _mix64(x: int) -> int:
    x &= 0xFFFFFFFFFFFFFFFF
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


def next_seed(device=None) -> int:
    import torch
    global _capture_gen
    if torch.cuda.is_current_stream_capturing():
        if _capture_gen is None:
            _capture_gen = torch.Generator()
            _capture_gen.manual_seed(_mix64(torch.initial_seed()) >> 1)
        return int(torch.randint(0, 2 ** 62, (1,), generator=_capture_gen).item())
    idx = getattr(device, "index", None)
    if idx is None:
        idx = torch.cuda.current_device()
    gen = torch.cuda.default_generators[idx]
    off = gen.get_offset()
    gen.set_offset(off + 4)                       # philox offsets advance in multiples of 4
    return _mix64(_mix64(gen.initial_seed()) + off) >> 2


# ---------------------------------------------------------------------------------------------
# Optional per-kernel timing for bench.py's roofline line: HIP events recorded on the stream
# the kernel is launched on, around selected C-ABI calls.  Off (zero overhead) by default.
# ---------------------------------------------------------------------------------------------
class KernelTimer:
    def __init__(self, names):
        self.names = set(names)
        self.events = {n: [] for n in self.names}

    def ms(self, name):
        """Per-launch durations in ms (call after a device synchronise)."""
        return [a.elapsed_time(b) for a, b in self.events[name]]


TIMER = None


class timed:
    """``with timed("cgnn_x", tag):`` brackets one C-ABI call with HIP events when a
    KernelTimer that names it is installed in ``_lib.TIMER``."""
    __slots__ = ("key", "pair")

    def __init__(self, name, tag=None):
        self.key = name if tag is None else f"{name}[{tag}]"
        self.pair = None

    def __enter__(self):
        t = TIMER
        if t is not None and self.key in t.names:
            import torch
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(torch.cuda.current_stream())
            self.pair = (a, b)
        return self

    def __exit__(self, *exc):
        if self.pair is not None:
            import torch
            self.pair[1].record(torch.cuda.current_stream())
            TIMER.events[self.key].append(self.pair)
        return False
