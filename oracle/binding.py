"""ctypes binding of the CPU oracle (oracle/liboracle-<cpu tag>.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package (acvm_amd) never does.
"""
import ctypes as C
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))


def _cpu_tag():
    """the library is built -march=native (oracle/Makefile): one file per kind of host CPU"""
    import hashlib
    try:
        with open("/proc/cpuinfo") as f:
            flags = next((line for line in f if line.startswith("flags")), "")
    except OSError:
        flags = ""
    return hashlib.sha1(" ".join(sorted(flags.split(":")[-1].split())).encode()).hexdigest()[:10]


_LIB_NAME = f"liboracle-{_cpu_tag()}.so"
_LIB_PATH = os.path.join(_DIR, _LIB_NAME)

ST_SOLVED, ST_IN_PROGRESS, ST_FAILURE, ST_REQUIRES_FOREIGN_CALL = 0, 1, 2, 3
(E_NONE, E_MISSING_ASSIGNMENT, E_TOO_MANY_UNKNOWNS, E_UNSUPPORTED_BLACKBOX, E_UNSATISFIED, E_INDEX_OOB,
 E_BLACKBOX_FAILED, E_BRILLIG_FAILED, E_PANIC) = range(9)
BACKEND_BARRETENBERG, BACKEND_STUBBED, BACKEND_DUMMY = 0, 1, 2


class Result(C.Structure):
    _fields_ = [("status", C.c_uint32), ("err", C.c_uint32), ("opcode_index", C.c_uint32), ("aux0", C.c_uint32),
                ("aux1", C.c_uint32), ("n_call_stack", C.c_uint32), ("call_stack", C.c_uint32 * 16),
                ("message", C.c_char * 200)]

    def as_tuple(self):
        return (self.status, self.err, self.opcode_index, self.aux0, self.aux1)


def build(force=False):
    srcs = [os.path.join(_DIR, f) for f in os.listdir(_DIR) if f.endswith((".c", ".h")) or f == "Makefile"]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _DIR, "-s", f"OUT={_LIB_NAME}", _LIB_NAME])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()  # (rebuilds when a source is newer than the library: a stale checker checks nothing)
        L = C.CDLL(_LIB_PATH)
        L.oracle_circuit_from_bytes.restype = C.c_void_p
        L.oracle_circuit_from_bytes.argtypes = [C.c_char_p, C.c_size_t]
        L.oracle_circuit_free.argtypes = [C.c_void_p]
        for f in ("oracle_circuit_num_witnesses", "oracle_circuit_num_opcodes", "oracle_circuit_current_witness_index"):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [C.c_void_p]
        L.oracle_result_size.restype = C.c_size_t
        L.oracle_acvm_create.restype = C.c_void_p
        L.oracle_acvm_create.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
        L.oracle_acvm_destroy.argtypes = [C.c_void_p]
        for f in ("oracle_acvm_run", "oracle_acvm_step", "oracle_acvm_instruction_pointer", "oracle_acvm_num_witnesses",
                  "oracle_acvm_pending_num_inputs"):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [C.c_void_p]
        L.oracle_acvm_result.argtypes = [C.c_void_p, C.POINTER(Result)]
        L.oracle_acvm_witness_map.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_acvm_pending_function.restype = C.c_char_p
        L.oracle_acvm_pending_function.argtypes = [C.c_void_p]
        L.oracle_acvm_pending_input_len.restype = C.c_uint32
        L.oracle_acvm_pending_input_len.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_acvm_pending_input.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_acvm_resolve.restype = C.c_int
        L.oracle_acvm_resolve.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_solve_batch.restype = C.c_int
        L.oracle_solve_batch.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
        L.oracle_solve_batch_mode.restype = C.c_int
        L.oracle_solve_batch_mode.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int]
        L.oracle_fr_op.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p]
        L.oracle_fr_num_bits.restype = C.c_uint32
        L.oracle_fr_num_bits.argtypes = [C.c_char_p]
        L.oracle_fr_from_bytes_reduce.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.oracle_fr_fetch_nearest_bytes.restype = C.c_int
        L.oracle_fr_fetch_nearest_bytes.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p]
        for f in ("oracle_sha256", "oracle_keccak256", "oracle_blake2s"):
            getattr(L, f).argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.oracle_grumpkin_generator.argtypes = [C.c_uint32, C.c_char_p]
        L.oracle_grumpkin_mul_g.argtypes = [C.c_char_p, C.c_char_p]
        L.oracle_pedersen_compress.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p]
        L.oracle_pedersen_hash_single.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
        L.oracle_pedersen.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p]
        L.oracle_fixed_base.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.oracle_schnorr_verify.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.oracle_schnorr_sign.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p]
        L.oracle_sorting_route.restype = C.c_size_t
        L.oracle_sorting_route.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_ecdsa_verify.restype = C.c_int
        L.oracle_ecdsa_verify.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p]
        assert L.oracle_result_size() == C.sizeof(Result)
        _lib = L
    return _lib


def be32(x: int) -> bytes:
    return int(x).to_bytes(32, "big")


class Circuit:
    def __init__(self, data: bytes):
        self._h = lib().oracle_circuit_from_bytes(data, len(data))
        if not self._h:
            raise ValueError("oracle: malformed circuit bytes")
        self.num_witnesses = lib().oracle_circuit_num_witnesses(self._h)
        self.num_opcodes = lib().oracle_circuit_num_opcodes(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_circuit_free(self._h)
            self._h = None


class ACVM:
    """Single-instance handle with the reference's call shape (pwg/mod.rs:145-304)."""

    def __init__(self, circuit: Circuit, initial_witness: dict, backend=BACKEND_BARRETENBERG):
        self.circuit = circuit
        ids = sorted(initial_witness)
        arr = (C.c_uint32 * len(ids))(*ids)
        vals = b"".join(be32(initial_witness[i]) for i in ids)
        self._h = lib().oracle_acvm_create(circuit._h, backend, len(ids), arr, vals)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_acvm_destroy(self._h)
            self._h = None

    def solve(self):
        return lib().oracle_acvm_run(self._h)

    def solve_opcode(self):
        return lib().oracle_acvm_step(self._h)

    def result(self) -> Result:
        r = Result()
        lib().oracle_acvm_result(self._h, C.byref(r))
        return r

    def instruction_pointer(self):
        return lib().oracle_acvm_instruction_pointer(self._h)

    def witness_map(self) -> dict:
        n = lib().oracle_acvm_num_witnesses(self._h)
        a = C.create_string_buffer(n)
        v = C.create_string_buffer(32 * n)
        lib().oracle_acvm_witness_map(self._h, a, v)
        return {w: int.from_bytes(v.raw[32 * w:32 * w + 32], "big") for w in range(n) if a.raw[w]}

    def get_pending_foreign_call(self):
        f = lib().oracle_acvm_pending_function(self._h)
        if f is None:
            return None
        inputs = []
        for i in range(lib().oracle_acvm_pending_num_inputs(self._h)):
            n = lib().oracle_acvm_pending_input_len(self._h, i)
            buf = C.create_string_buffer(32 * max(n, 1))
            lib().oracle_acvm_pending_input(self._h, i, buf)
            inputs.append([int.from_bytes(buf.raw[32 * k:32 * k + 32], "big") for k in range(n)])
        return f.decode(), inputs

    def resolve_pending_foreign_call(self, values):
        """values: list of int (Single) or list[int] (Array)."""
        is_arr = bytes(0 if isinstance(v, int) else 1 for v in values)
        lens = (C.c_uint32 * max(len(values), 1))(*[1 if isinstance(v, int) else len(v) for v in values])
        flat = b"".join(be32(v) if isinstance(v, int) else b"".join(be32(x) for x in v) for v in values)
        rc = lib().oracle_acvm_resolve(self._h, len(values), is_arr, lens, flat)
        if rc != 0:
            raise RuntimeError("ACVM is not expecting a foreign call response as no call was made")


MODE_SPARSE_MAP, MODE_CACHE_INV = 1, 2  # ORACLE_MODE_* of pwg.h: timing modes of the CPU baseline, results identical


def solve_batch(circuit: Circuit, ids, values_be: bytes, B: int, want_witness=True, backend=BACKEND_BARRETENBERG,
                n_threads=1, mode=0):
    """values_be: B * len(ids) * 32 bytes, instance-major. Returns (results[B], assigned bytes, values bytes).
    mode: MODE_SPARSE_MAP (every witness access also walks a BTreeMap-shaped tree: the reference's data structure) and / or
    MODE_CACHE_INV (constant divisors inverted once per circuit and thread instead of once per solved witness)."""
    import numpy as np
    n_in = len(ids)
    nw = circuit.num_witnesses
    for i in ids:
        nw = max(nw, i + 1)
    arr = (C.c_uint32 * max(n_in, 1))(*ids)
    res = (Result * B)()
    assigned = np.zeros((B, nw), dtype=np.uint8) if want_witness else None
    vals = np.zeros((B, nw, 32), dtype=np.uint8) if want_witness else None
    buf = np.frombuffer(values_be, dtype=np.uint8)
    assert buf.size == B * n_in * 32
    lib().oracle_solve_batch_mode(circuit._h, backend, B, n_in, arr, buf.ctypes.data, C.cast(res, C.c_void_p),
                                  assigned.ctypes.data if want_witness else None,
                                  vals.ctypes.data if want_witness else None, nw, n_threads, mode)
    return res, assigned, vals


P_BN254 = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_DIGEST_POWERS = {}


def _digest_powers(nw):
    """[g^(w+1)], [h^(w+1)] for w < nw; g, h = Blake2s-256("acvm_amd witness map digest: g" / "... h") as big-endian integers mod p"""
    import hashlib
    got = _DIGEST_POWERS.get("t")
    if got is None or len(got[0]) < nw:
        g = int.from_bytes(hashlib.blake2s(b"acvm_amd witness map digest: g").digest(), "big") % P_BN254
        h = int.from_bytes(hashlib.blake2s(b"acvm_amd witness map digest: h").digest(), "big") % P_BN254
        gp, hp, gs, hs = g, h, [], []
        for _ in range(nw):
            gs.append(gp)
            hs.append(hp)
            gp = gp * g % P_BN254
            hp = hp * h % P_BN254
        got = _DIGEST_POWERS["t"] = (gs, hs)
    return got


def witness_map_blake2s(assigned, values) -> bytes:
    """CPU restatement (hashlib) of acvm_batch_digest_blake2s (include/acvm_amd.h) for ONE instance -- checker only: leaves of 256 witnesses (32
    big-endian bytes each, 0xFF x 32 where unassigned), the root over u32_le(nw) and the leaves."""
    import hashlib
    nw = len(assigned)
    raw = bytes(values) if isinstance(values, (bytes, bytearray)) else values.tobytes() if hasattr(values, "tobytes") else b"".join(bytes(v) for v in values)
    leaves = []
    for k in range(0, nw, 256):
        leaves.append(hashlib.blake2s(b"".join(raw[32 * w:32 * w + 32] if assigned[w] else b"\xff" * 32 for w in range(k, min(k + 256, nw)))).digest())
    return hashlib.blake2s(nw.to_bytes(4, "little") + b"".join(leaves)).digest()


def witness_map_digest(assigned, values) -> bytes:
    """CPU restatement (Python big integers + hashlib) of the definition of acvm_batch_digest in include/acvm_amd.h, for ONE instance --
    checker only: assigned[w] truthy, values[w] = 32 big-endian bytes. D = sum over the assigned witnesses of value_w * g^(w+1) + h^(w+1)
    modulo p; digest = Blake2s-256(D as 32 big-endian bytes)."""
    import hashlib
    nw = len(assigned)
    gs, hs = _digest_powers(nw)
    raw = bytes(values) if isinstance(values, (bytes, bytearray)) else values.tobytes() if hasattr(values, "tobytes") else b"".join(bytes(v) for v in values)
    acc = 0
    for w in range(nw):
        if assigned[w]:
            acc += int.from_bytes(raw[32 * w:32 * w + 32], "big") * gs[w] + hs[w]
    return hashlib.blake2s((acc % P_BN254).to_bytes(32, "big")).digest()
