"""acvm_amd -- MI355X-native batched ACIR witness solver (drop-in for acvm::pwg::ACVM::solve()).

The compute path is libacvm_amd.so (hand-written gfx950 HIP kernels behind the C ABI of
include/acvm_amd.h). This module is only the ctypes view of that ABI for tests and the bench, shaped
like the reference API (acvm/src/pwg/mod.rs:145-304): Circuit.read -> Batch(...).solve() ->
results()/witness_map(). There is no CPU fallback: importing the solver without the built library, or
creating a batch without a gfx950 device, raises.
"""
import ctypes as C
import os

from . import acir  # noqa: F401  (data model + wire format)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (ACVM_AMD_LIB: another build of the same ABI, for A/B measurements of two library versions on one box: tools/gpu_ab_lib.sh)
LIB_PATH = os.environ.get("ACVM_AMD_LIB") or os.path.join(_HERE, "libacvm_amd.so")

STATUS_SOLVED, STATUS_IN_PROGRESS, STATUS_FAILURE, STATUS_REQUIRES_FOREIGN_CALL = 0, 1, 2, 3
(ERR_NONE, ERR_MISSING_ASSIGNMENT, ERR_TOO_MANY_UNKNOWNS, ERR_UNSUPPORTED_BLACKBOX, ERR_UNSATISFIED, ERR_INDEX_OOB,
 ERR_BLACKBOX_FAILED, ERR_BRILLIG_FAILED, ERR_PANIC, ERR_DEVICE_LIMIT) = range(10)
LIMIT_BRILLIG_STEPS, LIMIT_BRILLIG_CALL_DEPTH, LIMIT_BRILLIG_MEMORY, LIMIT_DEVICE_MEMORY = 1, 2, 3, 4

# every symbol include/acvm_amd.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "acvm_last_error", "acvm_abi_version", "acvm_device_count", "acvm_set_device", "acvm_device_synchronize",
    "acvm_device_arch", "acvm_selftest", "acvm_debug_grumpkin", "acvm_debug_secp", "acvm_circuit_from_bytes", "acvm_circuit_free", "acvm_circuit_num_opcodes",
    "acvm_circuit_num_witnesses", "acvm_circuit_plan_stats", "acvm_batch_new", "acvm_batch_free", "acvm_batch_set_initial_witness",
    "acvm_batch_set_initial_witness_device", "acvm_batch_solve", "acvm_batch_solve_then_import", "acvm_batch_reset", "acvm_batch_set_instances", "acvm_batch_set_force_slow_path",
    "acvm_batch_results", "acvm_batch_witness", "acvm_batch_witness_map", "acvm_batch_stats",
    "acvm_batch_set_profiling", "acvm_batch_pending_foreign_call", "acvm_batch_pending_foreign_call_inputs",
    "acvm_batch_resolve_foreign_call", "acvm_circuit_assert_message", "acvm_circuit_witness_set", "acvm_batch_error_string",
    "acvm_batch_extract_witnesses", "acvm_batch_digest", "acvm_witness_map_decode", "acvm_witness_map_encode", "acvm_batch_witness_map_bytes",
    "acvm_device_malloc", "acvm_device_free", "acvm_device_upload", "acvm_batch_solve_opcode", "acvm_bb_stubbed", "acvm_bb_dummy",
    "acvm_new", "acvm_free", "acvm_solve", "acvm_solve_opcode", "acvm_get_status", "acvm_instruction_pointer", "acvm_witness_map", "acvm_finalize",
    "acvm_get_pending_foreign_call", "acvm_pending_foreign_call_inputs", "acvm_resolve_pending_foreign_call",
    "acvm_multi_new", "acvm_multi_free", "acvm_multi_num_groups", "acvm_multi_solve", "acvm_multi_results", "acvm_multi_num_witnesses",
    "acvm_multi_witness_map", "acvm_multi_locate", "acvm_debug_modmul_rate", "acvm_debug_secp_rate", "acvm_batch_new_ex", "acvm_circuit_plan_stats_ex",
    "acvm_tuning_set", "acvm_tuning_get", "acvm_tuning_key",
    "acvm_device_release_tables", "acvm_circuit_opcode_kinds", "acvm_batch_error_expression", "acvm_debug_stream_rate", "acvm_node_new", "acvm_node_free", "acvm_node_tile_instances", "acvm_node_num_devices", "acvm_node_solve", "acvm_node_stats",
    "acvm_debug_cpulist", "acvm_debug_device_locality", "acvm_debug_plan_fingerprint", "acvm_circuit_plans_built", "acvm_circuit_check_schedule", "acvm_batch_digest_blake2s",
]


class AcvmError(RuntimeError):
    pass


class ExpressionHead(C.Structure):
    _fields_ = [("n_mul", C.c_uint32), ("n_lin", C.c_uint32), ("opcode_index", C.c_uint32), ("q_c", C.c_uint8 * 32)]


class Result(C.Structure):
    _fields_ = [("status", C.c_uint32), ("err", C.c_uint32), ("opcode_index", C.c_uint32), ("aux0", C.c_uint32),
                ("aux1", C.c_uint32), ("n_call_stack", C.c_uint32), ("call_stack", C.c_uint32 * 16),
                ("message", C.c_char * 200)]

    def as_tuple(self):
        return (self.status, self.err, self.opcode_index, self.aux0, self.aux1)


_SCHNORR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint8),
                          C.c_size_t, C.POINTER(C.c_uint8), C.c_void_p, C.c_size_t)
_PEDERSEN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.c_uint32, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8),
                           C.c_void_p, C.c_size_t)
_FIXED_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_void_p,
                        C.c_size_t)


_SCHNORR_BATCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t,
                                C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_void_p, C.c_size_t)
_PEDERSEN_BATCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t, C.c_uint32, C.POINTER(C.c_uint8),
                                 C.POINTER(C.c_uint8), C.c_void_p, C.c_size_t)
_FIXED_BATCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_void_p, C.c_size_t)


class BbSolver(C.Structure):
    """acvm_bb_solver_t: the BlackBoxFunctionSolver trait (blackbox_solver/src/lib.rs:27-45) as a vtable; the *_batch members
    are optional (NULL: the per-instance functions are called once per instance)."""
    _fields_ = [("ctx", C.c_void_p), ("schnorr_verify", _SCHNORR_FN), ("pedersen", _PEDERSEN_FN), ("fixed_base_scalar_mul", _FIXED_FN),
                ("schnorr_verify_batch", _SCHNORR_BATCH_FN), ("pedersen_batch", _PEDERSEN_BATCH_FN), ("fixed_base_scalar_mul_batch", _FIXED_BATCH_FN)]


def make_solver(schnorr_verify, pedersen, fixed_base_scalar_mul):
    """Wrap three Python callables as an acvm_bb_solver_t.
    schnorr_verify(pkx: int, pky: int, sig: bytes, msg: bytes) -> bool
    pedersen(inputs: list[int], domain_separator: int) -> (x, y)
    fixed_base_scalar_mul(low: int, high: int) -> (x, y)
    Raising BlackBoxFailed(msg) reports BlackBoxResolutionError::Failed, BlackBoxUnsupported reports Unsupported."""
    def be(ptr, n=32):
        return int.from_bytes(bytes(ptr[:n]), "big")

    def put(ptr, v):
        for i, byte in enumerate(int(v).to_bytes(32, "big")):
            ptr[i] = byte

    def guard(fn, err, err_len):
        try:
            fn()
            return 0
        except BlackBoxFailed as e:
            msg = str(e).encode()[: max(err_len - 1, 0)]
            if err:
                C.memmove(err, msg + b"\0", len(msg) + 1)
            return 1
        except BlackBoxUnsupported:
            return 2

    def c_schnorr(ctx, pkx, pky, sig, sig_len, msg, msg_len, ok, err, err_len):
        def run():
            ok[0] = 1 if schnorr_verify(be(pkx), be(pky), bytes(sig[:sig_len]), bytes(msg[:msg_len])) else 0
        return guard(run, err, err_len)

    def c_pedersen(ctx, inputs, n, ds, x, y, err, err_len):
        def run():
            vals = [int.from_bytes(bytes(inputs[32 * i:32 * i + 32]), "big") for i in range(n)]
            rx, ry = pedersen(vals, ds)
            put(x, rx)
            put(y, ry)
        return guard(run, err, err_len)

    def c_fixed(ctx, low, high, x, y, err, err_len):
        def run():
            rx, ry = fixed_base_scalar_mul(be(low), be(high))
            put(x, rx)
            put(y, ry)
        return guard(run, err, err_len)

    s = BbSolver(None, _SCHNORR_FN(c_schnorr), _PEDERSEN_FN(c_pedersen), _FIXED_FN(c_fixed))
    return s


def make_batched_solver(pedersen_batch=None, fixed_base_batch=None, schnorr_batch=None, fallback: "BbSolver" = None):
    """An acvm_bb_solver_t whose *_batch members are Python callables over the whole batch (one call per opcode):
    pedersen_batch(inputs: list[list[int]], domain_separator) -> list[(x, y)]
    fixed_base_batch(pairs: list[(low, high)]) -> list[(x, y)]
    schnorr_batch(items: list[(pkx, pky, sig: bytes, msg: bytes)]) -> list[bool]
    The per-instance members come from `fallback` (default: the stubbed vtable, so a missing batch member panics)."""
    fb = fallback if fallback is not None else bb_stubbed()

    def put(ptr, off, v):
        for i, byte in enumerate(int(v).to_bytes(32, "big")):
            ptr[off + i] = byte

    def c_ped(ctx, n, inputs, k, ds, xy, rc, err, stride):
        rows = [[int.from_bytes(bytes(inputs[(q * k + i) * 32:(q * k + i + 1) * 32]), "big") for i in range(k)] for q in range(n)]
        for q, (x, y) in enumerate(pedersen_batch(rows, ds)):
            put(xy, 64 * q, x)
            put(xy, 64 * q + 32, y)
            rc[q] = 0
        return 0

    def c_fixed(ctx, n, lh, xy, rc, err, stride):
        pairs = [(int.from_bytes(bytes(lh[64 * q:64 * q + 32]), "big"), int.from_bytes(bytes(lh[64 * q + 32:64 * q + 64]), "big")) for q in range(n)]
        for q, (x, y) in enumerate(fixed_base_batch(pairs)):
            put(xy, 64 * q, x)
            put(xy, 64 * q + 32, y)
            rc[q] = 0
        return 0

    def c_schnorr(ctx, n, pk, sig, sig_len, msg, msg_len, ok, rc, err, stride):
        items = [(int.from_bytes(bytes(pk[64 * q:64 * q + 32]), "big"), int.from_bytes(bytes(pk[64 * q + 32:64 * q + 64]), "big"),
                  bytes(sig[q * sig_len:(q + 1) * sig_len]), bytes(msg[q * msg_len:(q + 1) * msg_len])) for q in range(n)]
        for q, v in enumerate(schnorr_batch(items)):
            ok[q] = 1 if v else 0
            rc[q] = 0
        return 0

    s = BbSolver(None, fb.schnorr_verify, fb.pedersen, fb.fixed_base_scalar_mul,
                 _SCHNORR_BATCH_FN(c_schnorr) if schnorr_batch else _SCHNORR_BATCH_FN(), _PEDERSEN_BATCH_FN(c_ped) if pedersen_batch else _PEDERSEN_BATCH_FN(),
                 _FIXED_BATCH_FN(c_fixed) if fixed_base_batch else _FIXED_BATCH_FN())
    s._keep = fb
    return s


def bb_stubbed() -> "BbSolver":
    """StubbedBackend (acvm/tests/solver.rs:20-46) as shipped by the library."""
    return C.cast(lib().acvm_bb_stubbed(), C.POINTER(BbSolver)).contents


def bb_dummy() -> "BbSolver":
    """DummyBlackBoxSolver (brillig_vm/src/lib.rs:392-420) as shipped by the library."""
    return C.cast(lib().acvm_bb_dummy(), C.POINTER(BbSolver)).contents


class BlackBoxFailed(Exception):
    pass


class BlackBoxUnsupported(Exception):
    pass


class ForeignCallInfo(C.Structure):
    _fields_ = [("opcode_index", C.c_uint32), ("brillig_index", C.c_uint32), ("n_inputs", C.c_uint32), ("n_values", C.c_uint32),
                ("function", C.c_char * 64)]


class Stats(C.Structure):
    _fields_ = [("n_opcodes", C.c_uint32), ("n_witnesses", C.c_uint32), ("n_levels", C.c_uint32),
                ("n_fast_gates", C.c_uint32), ("n_dyn_gates", C.c_uint32), ("max_level_width", C.c_uint32),
                ("n_kernel_launches", C.c_uint32), ("n_slow_instances", C.c_uint32),
                ("algorithmic_bytes_per_instance", C.c_uint64), ("arith_algorithmic_bytes_per_instance", C.c_uint64),
                ("plan_ms", C.c_double), ("solve_device_ms", C.c_double), ("arith_kernel_ms", C.c_double),
                ("slow_path_ms", C.c_double), ("dyn_kernel_ms", C.c_double),
                ("dyn_algorithmic_bytes_per_instance", C.c_uint64), ("n_other_records", C.c_uint32), ("truncated_at", C.c_uint32),
                ("class_algorithmic_bytes_per_instance", C.c_uint64 * 4), ("class_kernel_ms", C.c_double * 4),
                ("n_gate_pairs", C.c_uint32), ("n_inverse_slots", C.c_uint32),
                ("n_scaled_witnesses", C.c_uint32), ("n_arith_launches", C.c_uint32),
                ("n_table_rows", C.c_uint32), ("n_digest_segments", C.c_uint32), ("n_brillig_inlined", C.c_uint32), ("n_brillig_retries", C.c_uint32),
                ("n_hash_chained", C.c_uint32),
                ("n_gate_out_asis", C.c_uint32), ("n_gate_out_weak", C.c_uint32), ("n_gate_out_canon", C.c_uint32), ("max_gate_bound", C.c_uint32),
                ("n_byte_planes", C.c_uint32), ("n_byte_plane_reads", C.c_uint32),
                ("n_stream_launches", C.c_uint32 * 6), ("n_stream_waits", C.c_uint32)]

    def as_dict(self):
        return {f: (list(getattr(self, f)) if f.startswith("class_") or f == "n_stream_launches" else getattr(self, f)) for f, _ in self._fields_}


_lib = None


def lib():
    """Load libacvm_amd.so. Raises if it has not been built (python -m acvm_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AcvmError(f"{LIB_PATH} is missing: build it with `python -m acvm_amd.build` (no CPU fallback exists)")
    L = C.CDLL(LIB_PATH)
    L.acvm_last_error.restype = C.c_char_p
    L.acvm_device_arch.argtypes = [C.c_char_p, C.c_size_t]
    L.acvm_selftest.argtypes = [C.c_uint32, C.c_uint64]
    L.acvm_debug_grumpkin.argtypes = [C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p]
    if hasattr(L, "acvm_debug_secp"):  # (an older build loaded through ACVM_AMD_LIB for an A/B run has no such probe)
        L.acvm_debug_secp.argtypes = [C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p]
    L.acvm_circuit_from_bytes.restype = C.c_void_p
    L.acvm_circuit_from_bytes.argtypes = [C.c_char_p, C.c_size_t]
    L.acvm_circuit_free.argtypes = [C.c_void_p]
    L.acvm_circuit_num_opcodes.restype = C.c_uint32
    L.acvm_circuit_num_opcodes.argtypes = [C.c_void_p]
    L.acvm_circuit_num_witnesses.restype = C.c_uint32
    L.acvm_circuit_num_witnesses.argtypes = [C.c_void_p]
    L.acvm_circuit_plan_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(Stats)]
    L.acvm_circuit_plan_stats_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(Stats)]
    L.acvm_debug_plan_fingerprint.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint32]
    L.acvm_circuit_plans_built.restype = C.c_uint64
    L.acvm_circuit_plans_built.argtypes = [C.c_void_p]
    L.acvm_circuit_check_schedule.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
    L.acvm_batch_new.restype = C.c_void_p
    L.acvm_batch_new.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.acvm_batch_new_ex.restype = C.c_void_p
    L.acvm_batch_new_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
    L.acvm_batch_free.argtypes = [C.c_void_p]
    L.acvm_batch_set_initial_witness.argtypes = [C.c_void_p, C.c_void_p]
    L.acvm_batch_set_initial_witness_device.argtypes = [C.c_void_p, C.c_void_p]
    L.acvm_batch_solve.argtypes = [C.c_void_p]
    L.acvm_batch_solve_then_import.argtypes = [C.c_void_p, C.c_void_p]
    L.acvm_batch_reset.argtypes = [C.c_void_p]
    L.acvm_batch_set_instances.argtypes = [C.c_void_p, C.c_uint32]
    L.acvm_batch_set_force_slow_path.argtypes = [C.c_void_p, C.c_int]
    L.acvm_batch_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.acvm_batch_results.argtypes = [C.c_void_p, C.c_void_p]
    L.acvm_batch_witness.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.acvm_batch_witness_map.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.acvm_batch_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.acvm_batch_pending_foreign_call.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(ForeignCallInfo)]
    L.acvm_batch_pending_foreign_call_inputs.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.acvm_batch_resolve_foreign_call.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_void_p, C.c_char_p]
    L.acvm_circuit_assert_message.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]
    L.acvm_circuit_witness_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]
    L.acvm_batch_error_string.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]
    L.acvm_batch_extract_witnesses.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.acvm_batch_digest.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.acvm_batch_digest_blake2s.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    for f in (L.acvm_witness_map_decode, L.acvm_witness_map_encode, L.acvm_batch_witness_map_bytes):
        f.restype = C.c_longlong
    L.acvm_witness_map_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint32]
    L.acvm_witness_map_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_void_p, C.c_size_t]
    L.acvm_batch_witness_map_bytes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
    L.acvm_debug_modmul_rate.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.acvm_debug_secp_rate.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.acvm_circuit_opcode_kinds.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.acvm_batch_error_expression.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(ExpressionHead), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
    L.acvm_device_release_tables.restype = C.c_longlong
    L.acvm_device_release_tables.argtypes = [C.c_int]
    L.acvm_node_new.restype = C.c_void_p
    L.acvm_node_new.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    L.acvm_node_free.restype = None
    L.acvm_node_free.argtypes = [C.c_void_p]
    L.acvm_node_tile_instances.argtypes = [C.c_void_p]
    L.acvm_node_tile_instances.restype = C.c_uint32
    L.acvm_node_num_devices.argtypes = [C.c_void_p]
    L.acvm_node_num_devices.restype = C.c_uint32
    L.acvm_node_solve.restype = C.c_longlong
    L.acvm_node_solve.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.acvm_node_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.acvm_debug_stream_rate.argtypes = [C.c_size_t, C.POINTER(C.c_double)]
    L.acvm_tuning_set.argtypes = [C.c_char_p, C.c_longlong]
    L.acvm_tuning_get.argtypes = [C.c_char_p, C.POINTER(C.c_longlong)]
    L.acvm_tuning_key.restype = C.c_char_p
    L.acvm_tuning_key.argtypes = [C.c_uint]
    L.acvm_device_malloc.restype = C.c_void_p
    L.acvm_device_malloc.argtypes = [C.c_size_t]
    L.acvm_device_free.argtypes = [C.c_void_p]
    L.acvm_device_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.acvm_batch_solve_opcode.argtypes = [C.c_void_p]
    L.acvm_bb_stubbed.restype = C.c_void_p
    L.acvm_bb_dummy.restype = C.c_void_p
    L.acvm_new.restype = C.c_void_p
    L.acvm_new.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint32]
    for f in (L.acvm_free, L.acvm_solve, L.acvm_solve_opcode):
        f.argtypes = [C.c_void_p]
    L.acvm_free.restype = None
    L.acvm_get_status.argtypes = [C.c_void_p, C.POINTER(Result)]
    L.acvm_instruction_pointer.restype = C.c_uint32
    L.acvm_instruction_pointer.argtypes = [C.c_void_p]
    for f in (L.acvm_witness_map, L.acvm_finalize):
        f.restype = C.c_longlong
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.acvm_get_pending_foreign_call.argtypes = [C.c_void_p, C.POINTER(ForeignCallInfo)]
    L.acvm_pending_foreign_call_inputs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.acvm_resolve_pending_foreign_call.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_void_p, C.c_char_p]
    L.acvm_multi_new.restype = C.c_void_p
    L.acvm_multi_new.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_char_p]
    L.acvm_multi_free.restype = None
    L.acvm_multi_free.argtypes = [C.c_void_p]
    L.acvm_multi_num_groups.restype = C.c_uint32
    L.acvm_multi_num_groups.argtypes = [C.c_void_p]
    L.acvm_multi_num_witnesses.restype = C.c_uint32
    L.acvm_multi_num_witnesses.argtypes = [C.c_void_p]
    L.acvm_multi_solve.argtypes = [C.c_void_p]
    L.acvm_multi_results.argtypes = [C.c_void_p, C.c_void_p]
    L.acvm_multi_witness_map.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.acvm_multi_locate.restype = C.c_void_p
    L.acvm_multi_locate.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        raise AcvmError(f"acvm_amd error {rc}: {lib().acvm_last_error().decode()}")
    return rc


def device_count():
    return lib().acvm_device_count()


_current_device = 0


def set_device(i):
    """hipSetDevice for the calling thread (HIP's current device is per thread: helper threads call this with current_device())"""
    global _current_device
    _check(lib().acvm_set_device(i))
    _current_device = i


def current_device():
    return _current_device


def synchronize():
    _check(lib().acvm_device_synchronize())


def selftest(n=1 << 16, seed=1):
    return _check(lib().acvm_selftest(n, seed))


def parse_cpulist(text):
    """a sysfs cpulist ("0-15,32-47") as the node driver reads it (host only)"""
    buf = (C.c_uint32 * 4096)()
    n = lib().acvm_debug_cpulist(text.encode(), buf, 4096)
    _check(n)
    return list(buf[:min(n, 4096)])


def device_locality(pci_root, bus_id):
    """(NUMA node, local CPUs) of a PCI device from a sysfs-shaped tree (host only): what a lane of the node driver pins its threads to"""
    buf = (C.c_uint32 * 4096)()
    node = C.c_int(-1)
    n = lib().acvm_debug_device_locality(pci_root.encode(), bus_id.encode(), C.byref(node), buf, 4096)
    _check(n)
    return node.value, list(buf[:min(n, 4096)])


def tuning_keys():
    """every planner / scheduler mode and device limit of the library (csrc/tuning.hpp)"""
    out, i = [], 0
    while True:
        k = lib().acvm_tuning_key(i)
        if not k:
            return out
        out.append(k.decode())
        i += 1


def tuning_get(key):
    v = C.c_longlong()
    _check(lib().acvm_tuning_get(key.encode(), C.byref(v)))
    return v.value


def tuning_set(key, value):
    _check(lib().acvm_tuning_set(key.encode(), int(value)))


class tuning:
    """with acvm_amd.tuning(scale=0, pairs=0): ... -- the modes hold for batches CREATED inside the block"""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: tuning_get(k) for k in self.kv}
        for k, v in self.kv.items():
            tuning_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            tuning_set(k, v)
        return False


def release_tables(device=0):
    """acvm_device_release_tables: frees the lookup tables of `device` (bytes given back); AcvmError while a handle of the device still uses them"""
    r = lib().acvm_device_release_tables(device)
    if r < 0:
        raise AcvmError(lib().acvm_last_error().decode())
    return r


def stream_rate(nbytes=4 << 30):
    """GB/s moved by a two-rows-in, one-row-out stream of nbytes per row (the gate kernel's access shape): the measured streaming ceiling
    beside the spec peak of the HBM roofline"""
    r = C.c_double()
    _check(lib().acvm_debug_stream_rate(nbytes, C.byref(r)))
    return r.value


def modmul_rate(iters=400, waves_per_simd=8):
    """(modmul/s, products per launch) of the back-to-back fr29_mul probe: the peak of the ALU roofline"""
    r, n = C.c_double(), C.c_uint64()
    _check(lib().acvm_debug_modmul_rate(iters, waves_per_simd, C.byref(r), C.byref(n)))
    return r.value, n.value


def secp_rate(curve, iters=400, waves_per_simd=8):
    """(products/s, products per launch) of the back-to-back sp_mul / sp_sqr probe of secp256k1 (0) / secp256r1 (1): the peak of the ECDSA kernels' ALU roofline"""
    r, n = C.c_double(), C.c_uint64()
    _check(lib().acvm_debug_secp_rate(curve, iters, waves_per_simd, C.byref(r), C.byref(n)))
    return r.value, n.value


def modmul_probe_cus():
    """compute units the probe's grid is sized by (blocks = CUs x waves_per_simd): products per launch / (256 x waves x iters x 2)"""
    _, n = modmul_rate(1, 1)
    return n // 512


def device_arch():
    buf = C.create_string_buffer(64)
    _check(lib().acvm_device_arch(buf, 64))
    return buf.value.decode()


def debug_grumpkin(what, param, inputs=()):
    """Component probe of the Grumpkin kernels (see include/acvm_amd.h); returns (x, y) as ints."""
    out = C.create_string_buffer(64)
    data = b"".join(int(v).to_bytes(32, "big") for v in inputs)
    _check(lib().acvm_debug_grumpkin(what, param, data, len(inputs), out))
    return int.from_bytes(out.raw[:32], "big"), int.from_bytes(out.raw[32:], "big")


SECP_PROBE_WORDS = ((2, 1), (1, 1), (2, 1), (2, 1), (1, 1), (1, 1), (3, 3), (5, 3), (1, 1), (2, 1), (2, 1), (1, 1))


def debug_secp(curve, what, items):
    """Component probe of the ECDSA kernels (see include/acvm_amd.h): items = tuples of ints (values < p); returns one tuple of ints per item."""
    wi, wo = SECP_PROBE_WORDS[what]
    assert all(len(it) == wi for it in items)
    data = b"".join(int(v).to_bytes(32, "big") for it in items for v in it)
    out = C.create_string_buffer(32 * wo * max(len(items), 1))
    _check(lib().acvm_debug_secp(curve, what, data, len(items), out))
    raw = out.raw
    return [tuple(int.from_bytes(raw[32 * (wo * i + k):32 * (wo * i + k + 1)], "big") for k in range(wo)) for i in range(len(items))]


def decompress_witness(data: bytes) -> dict:
    """WitnessMap::try_from(&[u8]) (witness_map.rs:135-146; acvm_js decompressWitness): {witness index: int}."""
    n = _check(lib().acvm_witness_map_decode(data, len(data), None, None, 0))
    ids = (C.c_uint32 * max(n, 1))()
    vals = C.create_string_buffer(32 * max(n, 1))
    _check(lib().acvm_witness_map_decode(data, len(data), ids, vals, n))
    return {ids[i]: int.from_bytes(vals.raw[32 * i:32 * i + 32], "big") for i in range(n)}


def compress_witness(witness_map: dict) -> bytes:
    """Vec<u8>::try_from(WitnessMap) (witness_map.rs:108-119; acvm_js compressWitness)."""
    items = sorted(witness_map.items())
    ids = (C.c_uint32 * max(len(items), 1))(*[k for k, _ in items])
    vals = b"".join(int(v).to_bytes(32, "big") for _, v in items)
    n = _check(lib().acvm_witness_map_encode(ids, vals, len(items), None, 0))
    out = C.create_string_buffer(max(n, 1))
    _check(lib().acvm_witness_map_encode(ids, vals, len(items), out, n))
    return out.raw[:n]


class Circuit:
    """acir::circuit::Circuit::read (circuit/mod.rs:154-161)."""

    def __init__(self, data: bytes):
        self._h = lib().acvm_circuit_from_bytes(data, len(data))
        if not self._h:
            raise AcvmError(lib().acvm_last_error().decode())
        self.num_opcodes = lib().acvm_circuit_num_opcodes(self._h)
        self.num_witnesses = lib().acvm_circuit_num_witnesses(self._h)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.acvm_circuit_free(self._h)
            self._h = None

    LOCATION_ACIR = 0xFFFFFFFF
    SETS = {"private_parameters": 0, "public_parameters": 1, "return_values": 2, "public_inputs": 3, "circuit_arguments": 4}

    def opcode_kinds(self):
        """ACVM::opcodes seen through the ABI: [(Opcode variant, sub-kind)] per opcode (acvm_circuit_opcode_kinds)"""
        import numpy as np
        n = lib().acvm_circuit_num_opcodes(self._h)
        k = np.zeros((max(n, 1), 2), dtype=np.uint32)
        _check(lib().acvm_circuit_opcode_kinds(self._h, 0, n, k.ctypes.data))
        return [(int(a), int(b)) for a, b in k[:n]]

    def get_assert_message(self, acir_index: int, brillig_index: int = None):
        """Circuit::get_assert_message (circuit/mod.rs:43-51) for OpcodeLocation::Acir / ::Brillig; None if there is none."""
        buf = C.create_string_buffer(4096)
        n = lib().acvm_circuit_assert_message(self._h, acir_index, self.LOCATION_ACIR if brillig_index is None else brillig_index, buf, 4096)
        return None if n < 0 else buf.value.decode()

    def witness_set(self, which: str):
        """private_parameters / public_parameters / return_values / public_inputs() / circuit_arguments() (circuit/mod.rs:25-32,109-121)."""
        n = _check(lib().acvm_circuit_witness_set(self._h, self.SETS[which], None, 0))
        arr = (C.c_uint32 * max(n, 1))()
        _check(lib().acvm_circuit_witness_set(self._h, self.SETS[which], arr, n))
        return list(arr[:n])

    def plan_stats(self, initial_ids, fold_digest=False, reuse_slots=False, keep=()) -> dict:
        """Host-only levelisation (no device): statistics of the static plan; raises if an opcode has no kernel."""
        ids = list(initial_ids)
        arr = (C.c_uint32 * max(len(ids), 1))(*ids)
        keep = list(keep)
        karr = (C.c_uint32 * max(len(keep), 1))(*keep)
        s = Stats()
        _check(lib().acvm_circuit_plan_stats_ex(self._h, arr, len(ids), (1 if fold_digest else 0) | (2 if reuse_slots else 0), karr, len(keep), C.byref(s)))
        return s.as_dict()

    def plans_built(self) -> int:
        """how often this circuit handle has been levelised (handles of equal options share one plan)"""
        return int(lib().acvm_circuit_plans_built(self._h))

    def check_schedule(self, initial_ids, n_instances=4096, fold_digest=False, reuse_slots=False, keep=(), host_solver=False, drop_wait=None) -> dict:
        """Host-only hazard check of the level schedule a handle of these options would enqueue (acvm_circuit_check_schedule): every
        launch's reads and writes derived from the record words, happens-before from stream order + events. drop_wait = k leaves the
        k-th cross-stream wait out (mutation testing). Returns {ok, n_launches, n_waits, n_accesses, n_records, n_findings, report}."""
        ids = list(initial_ids)
        arr = (C.c_uint32 * max(len(ids), 1))(*ids)
        keep = list(keep)
        karr = (C.c_uint32 * max(len(keep), 1))(*keep)
        counts = (C.c_uint64 * 8)()
        text = C.create_string_buffer(1 << 14)
        rc = lib().acvm_circuit_check_schedule(self._h, arr, len(ids), (1 if fold_digest else 0) | (2 if reuse_slots else 0) | (0x100 if host_solver else 0),
                                               karr, len(keep), n_instances, 0xFFFFFFFF if drop_wait is None else drop_wait, counts, text, len(text))
        _check(rc)
        return {"ok": rc == 0, "n_launches": counts[0], "n_waits": counts[1], "n_accesses": counts[2], "n_records": counts[3], "n_findings": counts[4],
                "report": text.value.decode()}

    def plan_fingerprint(self, initial_ids, fold_digest=False, reuse_slots=False, keep=(), host_solver=False) -> list:
        """Host-only: 64-bit fingerprints of the static plan, one per component (acvm_debug_plan_fingerprint)."""
        ids = list(initial_ids)
        arr = (C.c_uint32 * max(len(ids), 1))(*ids)
        keep = list(keep)
        karr = (C.c_uint32 * max(len(keep), 1))(*keep)
        out = (C.c_uint64 * 128)()
        n = lib().acvm_debug_plan_fingerprint(self._h, arr, len(ids), (1 if fold_digest else 0) | (2 if reuse_slots else 0) | (0x100 if host_solver else 0),
                                              karr, len(keep), out, 128)
        _check(n)
        return list(out[:n])


class NodeOpts(C.Structure):
    _fields_ = [("n_devices", C.c_uint32), ("devices", C.POINTER(C.c_int)), ("tile_instances", C.c_uint32), ("batch_flags", C.c_uint32)]


class NodeStats(C.Structure):
    _fields_ = [("n_devices", C.c_uint32), ("tile_instances", C.c_uint32), ("n_instances", C.c_uint64), ("total_ms", C.c_double),
                ("device", C.c_int * 16), ("async_exact", C.c_uint32 * 16), ("tiles", C.c_uint32 * 16), ("exact_instances", C.c_uint32 * 16),
                ("lane_ms", C.c_double * 16), ("solve_device_ms", C.c_double * 16), ("h2d_wait_ms", C.c_double * 16), ("export_ms", C.c_double * 16),
                ("numa_node", C.c_int * 16), ("n_cpus_pinned", C.c_uint32 * 16), ("first_cpu", C.c_int * 16),
                ("plans_built", C.c_uint32), ("create_ms", C.c_double), ("plan_ms", C.c_double), ("host_rss_bytes", C.c_uint64)]


class Node:
    """acvm_node_*: one call solves a global batch on every listed device (one handle + one host thread per device, tiles inside a
    device, pinned double-buffered uploads, the exact path of a tile beside the next one). The caller loop it replaces:
    acvm_js/src/execute.rs:60-119 once per instance."""

    def __init__(self, circuit: "Circuit", initial_ids, keep=(), devices=None, tile=0, fold_digest=False, reuse_slots=False, solver: "BbSolver" = None):
        self.ids, self.keep = list(initial_ids), list(keep)
        self._solver = solver
        arr = (C.c_uint32 * max(len(self.ids), 1))(*self.ids)
        karr = (C.c_uint32 * max(len(self.keep), 1))(*self.keep)
        opts = NodeOpts()
        if devices is not None:
            self._dev = (C.c_int * len(devices))(*devices)
            opts.n_devices, opts.devices = len(devices), self._dev
        opts.tile_instances = tile
        opts.batch_flags = (Batch.FOLD_DIGEST if fold_digest else 0) | (Batch.REUSE_SLOTS if reuse_slots else 0)
        self._h = lib().acvm_node_new(circuit._h, C.byref(solver) if solver is not None else None, arr, len(self.ids), karr, len(self.keep), C.byref(opts))
        if not self._h:
            raise AcvmError(lib().acvm_last_error().decode())
        self.tile = lib().acvm_node_tile_instances(self._h)
        self.n_devices = lib().acvm_node_num_devices(self._h)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.acvm_node_free(self._h)
            self._h = None

    def free(self):
        self.__del__()

    def solve(self, values_be, n_instances: int, results=True, kept=True, digests=True):
        """values_be: n_instances * len(ids) * 32 bytes (bytes or a uint8 array). Returns (not_solved, results or None, kept uint8
        [n][len(keep)][32] or None, kept_assigned uint8 [n][len(keep)] or None, digests uint8 [n][32] or None)."""
        import numpy as np
        buf = np.frombuffer(values_be, dtype=np.uint8) if not isinstance(values_be, np.ndarray) else values_be
        if buf.size != n_instances * len(self.ids) * 32:
            raise ValueError("values_be has the wrong size")
        res = (Result * max(n_instances, 1))() if results else None
        nk = len(self.keep)
        kv = np.zeros((n_instances, nk, 32), dtype=np.uint8) if kept and nk else None
        ka = np.zeros((n_instances, nk), dtype=np.uint8) if kept and nk else None
        dg = np.zeros((n_instances, 32), dtype=np.uint8) if digests else None
        rc = lib().acvm_node_solve(self._h, n_instances, buf.ctypes.data if buf.size else None, C.cast(res, C.c_void_p) if res is not None else None,
                                   kv.ctypes.data if kv is not None else None, ka.ctypes.data if ka is not None else None, dg.ctypes.data if dg is not None else None)
        _check(rc)
        return rc, res, kv, ka, dg

    def stats(self):
        st = NodeStats()
        _check(lib().acvm_node_stats(self._h, C.byref(st)))
        n = st.n_devices
        return {"n_devices": n, "tile_instances": st.tile_instances, "n_instances": st.n_instances, "total_ms": st.total_ms,
                "plans_built": st.plans_built, "create_ms": st.create_ms, "plan_ms": st.plan_ms, "host_rss_bytes": st.host_rss_bytes,
                **{f: list(getattr(st, f))[:n] for f in ("device", "async_exact", "tiles", "exact_instances", "lane_ms", "solve_device_ms", "h2d_wait_ms", "export_ms", "numa_node", "n_cpus_pinned",
                                                              "first_cpu")}}


class Batch:
    """ACVM::new / solve / witness_map for n_instances instances of one circuit."""

    FOLD_DIGEST, REUSE_SLOTS = 1, 2

    def __init__(self, circuit: Circuit, n_instances: int, initial_ids, solver: "BbSolver" = None, fold_digest=False, reuse_slots=False, keep=()):
        """fold_digest: the map digests are computed during the solve; reuse_slots: witness-slot liveness reuse (only the initial
        witnesses and `keep` can be read back afterwards, plus results and digests): acvm_batch_new_ex"""
        self.circuit = circuit
        self.B = n_instances
        self.ids = list(initial_ids)
        self._solver = solver  # keeps the callbacks alive: the vtable must outlive the batch
        arr = (C.c_uint32 * max(len(self.ids), 1))(*self.ids)
        keep = list(keep)
        karr = (C.c_uint32 * max(len(keep), 1))(*keep)
        flags = (self.FOLD_DIGEST if fold_digest else 0) | (self.REUSE_SLOTS if reuse_slots else 0)
        self._h = lib().acvm_batch_new_ex(circuit._h, C.byref(solver) if solver is not None else None, n_instances, arr, len(self.ids), flags, karr, len(keep))
        if not self._h:
            raise AcvmError(lib().acvm_last_error().decode())
        self.nw = self.stats()["n_witnesses"]

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.acvm_batch_free(self._h)
            self._h = None

    def free(self):
        self.__del__()

    def set_initial_witness(self, values_be: bytes):
        """values_be: B * len(ids) * 32 bytes, instance-major canonical big-endian."""
        import numpy as np
        buf = np.frombuffer(values_be, dtype=np.uint8)
        if buf.size != self.B * len(self.ids) * 32:
            raise ValueError("initial witness buffer has the wrong size")
        _check(lib().acvm_batch_set_initial_witness(self._h, buf.ctypes.data))

    def solve(self, then_import: int = 0) -> int:
        """ACVM::solve for every instance; then_import: device pointer of the NEXT tile's inputs, imported behind the solve when no instance
        left the generic path (acvm_batch_solve_then_import)"""
        if then_import:
            return _check(lib().acvm_batch_solve_then_import(self._h, then_import))
        return _check(lib().acvm_batch_solve(self._h))

    def solve_opcode(self) -> int:
        """ACVM::solve_opcode for the batch (see include/acvm_amd.h): one opcode, returns the number of instances not Solved."""
        return _check(lib().acvm_batch_solve_opcode(self._h))

    def set_initial_witness_device(self, d_ptr: int):
        """values already resident on the device (same layout as set_initial_witness), e.g. a slice of a DeviceBuffer."""
        _check(lib().acvm_batch_set_initial_witness_device(self._h, d_ptr))

    def reset(self):
        _check(lib().acvm_batch_reset(self._h))

    def set_instances(self, n):
        """from the next set_initial_witness on the handle covers instances [0, n), n <= the size it was created with"""
        _check(lib().acvm_batch_set_instances(self._h, n))
        self.B = n

    def set_force_slow_path(self, on: bool):
        _check(lib().acvm_batch_set_force_slow_path(self._h, int(on)))

    def set_profiling(self, on: bool):
        _check(lib().acvm_batch_set_profiling(self._h, int(on)))

    def results(self):
        out = (Result * max(self.B, 1))()
        _check(lib().acvm_batch_results(self._h, C.cast(out, C.c_void_p)))
        return out

    def witness(self, w: int):
        import numpy as np
        vals = np.zeros((self.B, 32), dtype=np.uint8)
        asg = np.zeros((self.B,), dtype=np.uint8)
        _check(lib().acvm_batch_witness(self._h, w, vals.ctypes.data, asg.ctypes.data))
        return vals, asg

    def witness_map(self, first=0, n=None):
        import numpy as np
        n = self.B - first if n is None else n
        asg = np.zeros((n, self.nw), dtype=np.uint8)
        vals = np.zeros((n, self.nw, 32), dtype=np.uint8)
        _check(lib().acvm_batch_witness_map(self._h, first, n, asg.ctypes.data, vals.ctypes.data))
        return asg, vals

    def error_string(self, instance: int) -> str:
        """The string acvm_js reports for a failed instance (execute.rs:79-108): assert message or the error's Display text."""
        buf = C.create_string_buffer(1024)
        _check(lib().acvm_batch_error_string(self._h, self.circuit._h, instance, buf, 1024))
        return buf.value.decode()

    def error_expression(self, instance: int):
        """ExpressionHasTooManyUnknowns(Expression) of a failed instance as data (acvm_batch_error_expression): None if the instance did not fail
        that way, else {"opcode_index", "mul": [(coef, wl, wr)], "lin": [(coef, w)], "q_c"} with coefficients as integers"""
        import numpy as np
        head = ExpressionHead()
        cap = 64
        while True:
            mc, mw = np.zeros((cap, 32), dtype=np.uint8), np.zeros((cap, 2), dtype=np.uint32)
            lc, lw = np.zeros((cap, 32), dtype=np.uint8), np.zeros(cap, dtype=np.uint32)
            rc = _check(lib().acvm_batch_error_expression(self._h, self.circuit._h, instance, C.byref(head), mc.ctypes.data, mw.ctypes.data, cap,
                                                          lc.ctypes.data, lw.ctypes.data, cap))
            if rc == 0:
                return None
            if max(head.n_mul, head.n_lin) <= cap:
                break
            cap = max(head.n_mul, head.n_lin)
        be = lambda a: int.from_bytes(bytes(a), "big")
        return {"opcode_index": head.opcode_index, "q_c": be(head.q_c),
                "mul": [(be(mc[i]), int(mw[i, 0]), int(mw[i, 1])) for i in range(head.n_mul)],
                "lin": [(be(lc[i]), int(lw[i])) for i in range(head.n_lin)]}

    def extract(self, witnesses, first=0, n=None):
        """extract_indices (public_witness.rs:10-21) for instances [first, first + n): uint8 array [n][len(witnesses)][32];
        raises if a witness is unassigned. Use with Circuit.witness_set("return_values" | "public_parameters" | "public_inputs")."""
        import numpy as np
        n = self.B - first if n is None else n
        ws = list(witnesses)
        arr = (C.c_uint32 * max(len(ws), 1))(*ws)
        vals = np.zeros((n, len(ws), 32), dtype=np.uint8)
        _check(lib().acvm_batch_extract_witnesses(self._h, arr, len(ws), first, n, vals.ctypes.data))
        return vals

    def digest(self, first=0, n=None):
        """Per-instance 32-byte digest of the witness map (acvm_batch_digest): uint8 array [n][32]."""
        import numpy as np
        n = self.B - first if n is None else n
        out = np.zeros((n, 32), dtype=np.uint8)
        _check(lib().acvm_batch_digest(self._h, first, n, out.ctypes.data))
        return out

    def digest_blake2s(self, first=0, n=None):
        """Per-instance Blake2s tree digest of the witness map's BYTES (acvm_batch_digest_blake2s): uint8 array [n][32]."""
        import numpy as np
        n = self.B - first if n is None else n
        out = np.zeros((n, 32), dtype=np.uint8)
        _check(lib().acvm_batch_digest_blake2s(self._h, first, n, out.ctypes.data))
        return out

    def witness_map_bytes(self, instance: int) -> bytes:
        """The instance's WitnessMap in the reference's wire format (finalize() + compressWitness)."""
        n = _check(lib().acvm_batch_witness_map_bytes(self._h, instance, None, 0))
        out = C.create_string_buffer(max(n, 1))
        _check(lib().acvm_batch_witness_map_bytes(self._h, instance, out, n))
        return out.raw[:n]

    def get_pending_foreign_call(self, instance: int):
        """ACVM::get_pending_foreign_call: None, or (function, [[int, ...] per input])."""
        info = ForeignCallInfo()
        if _check(lib().acvm_batch_pending_foreign_call(self._h, instance, C.byref(info))) == 0:
            return None
        lens = (C.c_uint32 * max(info.n_inputs, 1))()
        vals = C.create_string_buffer(32 * max(info.n_values, 1))
        _check(lib().acvm_batch_pending_foreign_call_inputs(self._h, instance, lens, vals))
        out, k = [], 0
        for i in range(info.n_inputs):
            out.append([int.from_bytes(vals.raw[32 * (k + c):32 * (k + c + 1)], "big") for c in range(lens[i])])
            k += lens[i]
        return info.function.decode(), out

    def resolve_pending_foreign_call(self, instance: int, values):
        """values: list of int (Single) or list[int] (Array), like ForeignCallResult."""
        is_arr = bytes(0 if isinstance(v, int) else 1 for v in values)
        lens = (C.c_uint32 * max(len(values), 1))(*[1 if isinstance(v, int) else len(v) for v in values])
        flat = b"".join(int(v).to_bytes(32, "big") if isinstance(v, int) else b"".join(int(x).to_bytes(32, "big") for x in v) for v in values)
        _check(lib().acvm_batch_resolve_foreign_call(self._h, instance, len(values), is_arr, lens, flat))

    def stats(self) -> dict:
        s = Stats()
        _check(lib().acvm_batch_stats(self._h, C.byref(s)))
        return s.as_dict()


class DeviceBuffer:
    """hipMalloc'd bytes on the current device (acvm_device_malloc): inputs a caller keeps resident in HBM."""

    def __init__(self, data: bytes = None, size: int = None):
        self.size = len(data) if data is not None else size
        self.ptr = lib().acvm_device_malloc(self.size)
        if not self.ptr:
            raise AcvmError(lib().acvm_last_error().decode())
        if data is not None:
            self.upload(data)

    def upload(self, data, offset=0):
        import numpy as np
        buf = np.frombuffer(data, dtype=np.uint8)
        if offset + buf.size > self.size:
            raise ValueError("upload past the end of the device buffer")
        _check(lib().acvm_device_upload(self.ptr + offset, buf.ctypes.data, buf.size))

    def free(self):
        if getattr(self, "ptr", None) and _lib is not None:
            _lib.acvm_device_free(self.ptr)
            self.ptr = None

    __del__ = free


class Acvm:
    """struct ACVM (acvm/src/pwg/mod.rs:129-304) for one instance: the acvm_new / acvm_solve / acvm_finalize shim."""

    def __init__(self, circuit: Circuit, initial_witness: dict, backend: "BbSolver" = None):
        self.circuit = circuit
        self._backend = backend
        items = list(initial_witness.items())
        ids = (C.c_uint32 * max(len(items), 1))(*[k for k, _ in items])
        vals = b"".join(int(v % (1 << 256)).to_bytes(32, "big") for _, v in items)
        self._h = lib().acvm_new(circuit._h, C.byref(backend) if backend is not None else None, ids, vals, len(items))
        if not self._h:
            raise AcvmError(lib().acvm_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.acvm_free(self._h)
            self._h = None

    def solve(self) -> int:
        return _check(lib().acvm_solve(self._h))

    def solve_opcode(self) -> int:
        return _check(lib().acvm_solve_opcode(self._h))

    def status(self) -> Result:
        r = Result()
        _check(lib().acvm_get_status(self._h, C.byref(r)))
        return r

    def instruction_pointer(self) -> int:
        return lib().acvm_instruction_pointer(self._h)

    def _pairs(self, fn):
        n = fn(self._h, None, None, 0)
        if n < 0:
            raise AcvmError(f"acvm_amd error {n}: {lib().acvm_last_error().decode()}")
        ids = (C.c_uint32 * max(n, 1))()
        vals = C.create_string_buffer(32 * max(n, 1))
        _check(fn(self._h, ids, vals, n))
        return {ids[i]: int.from_bytes(vals.raw[32 * i:32 * i + 32], "big") for i in range(n)}

    def witness_map(self) -> dict:
        return self._pairs(lib().acvm_witness_map)

    def finalize(self) -> dict:
        """ACVM::finalize: raises unless the status is Solved (the reference panics)."""
        return self._pairs(lib().acvm_finalize)

    def get_pending_foreign_call(self):
        info = ForeignCallInfo()
        if _check(lib().acvm_get_pending_foreign_call(self._h, C.byref(info))) == 0:
            return None
        lens = (C.c_uint32 * max(info.n_inputs, 1))()
        vals = C.create_string_buffer(32 * max(info.n_values, 1))
        _check(lib().acvm_pending_foreign_call_inputs(self._h, lens, vals))
        out, k = [], 0
        for i in range(info.n_inputs):
            out.append([int.from_bytes(vals.raw[32 * (k + c):32 * (k + c + 1)], "big") for c in range(lens[i])])
            k += lens[i]
        return info.function.decode(), out

    def resolve_pending_foreign_call(self, values):
        is_arr = bytes(0 if isinstance(v, int) else 1 for v in values)
        lens = (C.c_uint32 * max(len(values), 1))(*[1 if isinstance(v, int) else len(v) for v in values])
        flat = b"".join(int(v).to_bytes(32, "big") if isinstance(v, int) else b"".join(int(x).to_bytes(32, "big") for x in v) for v in values)
        _check(lib().acvm_resolve_pending_foreign_call(self._h, len(values), is_arr, lens, flat))


class MultiBatch:
    """acvm_multi_*: instances whose initial witness maps assign different id sets (list of {witness: int} dicts)."""

    def __init__(self, circuit: Circuit, initial_maps, solver: "BbSolver" = None):
        import numpy as np
        self.circuit = circuit
        self._solver = solver
        self.n = len(initial_maps)
        offsets = np.zeros(self.n + 1, dtype=np.uint64)
        ids, vals = [], []
        for i, m in enumerate(initial_maps):
            for k, v in m.items():
                ids.append(k)
                vals.append(int(v % (1 << 256)).to_bytes(32, "big"))
            offsets[i + 1] = len(ids)
        ids_arr = np.asarray(ids if ids else [0], dtype=np.uint32)
        self._h = lib().acvm_multi_new(circuit._h, C.byref(solver) if solver is not None else None, self.n, offsets.ctypes.data, ids_arr.ctypes.data,
                                       b"".join(vals))
        if not self._h:
            raise AcvmError(lib().acvm_last_error().decode())
        self.n_groups = lib().acvm_multi_num_groups(self._h)
        self.nw = lib().acvm_multi_num_witnesses(self._h)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.acvm_multi_free(self._h)
            self._h = None

    def solve(self) -> int:
        return _check(lib().acvm_multi_solve(self._h))

    def results(self):
        out = (Result * max(self.n, 1))()
        _check(lib().acvm_multi_results(self._h, C.cast(out, C.c_void_p)))
        return out

    def witness_map(self, instance: int):
        import numpy as np
        asg = np.zeros((self.nw,), dtype=np.uint8)
        vals = np.zeros((self.nw, 32), dtype=np.uint8)
        _check(lib().acvm_multi_witness_map(self._h, instance, asg.ctypes.data, vals.ctypes.data))
        return asg, vals
