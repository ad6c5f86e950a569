#!/usr/bin/env python3
"""bench.py -- witnesses solved / second on MI355X for the batched ACIR witness solver.

Default workload = BASELINE.json's metric: 10k-gate arithmetic-only ACIR, batch 2^20 witnesses over the whole node
(synthetic circuit and inputs from acvm_amd.synth). The global batch is sharded contiguously over the ranks (strong
scaling: 2^20 / N instances per GPU, no data-path collective); a rank solves its shard in tiles of 2^--tile-log2 instances (default 2^17)
through one reused batch handle (the 335 GB witness table of 2^20 instances does not fit one GPU's 288 GB: SURVEY 8e). The
inputs of the whole shard are resident in HBM before the timed region starts; a "step" = ACVM::new + ACVM::solve of every
instance of the global batch (per tile: import of the resident inputs, then the level kernels -- the same for every N and every tile
count), witness maps left in HBM.
Other workloads (parity-test configs, measured for DESIGN.md; 2^16 instances per GPU unless --total-log2 is given):
    --workload hash            config 3: SHA256 + Keccak256 + RANGE circuit
    --workload grumpkin        config 4: Pedersen + FixedBaseScalarMul + SchnorrVerify circuit
    --workload ecdsa           EcdsaSecp256k1 + EcdsaSecp256r1
    --workload arith_pedersen  the north-star shape: 10k arithmetic gates + 8 Pedersen commitments
    --workload mixed           the config-5 opcode mix at --gates opcodes (every kernel class in one circuit)
    --workload config5         config 5 at circuit size: the 10^6-opcode mixed circuit (--gates), tiles of --tile-log2 (default 2^12) instances,
                               --total-log2 (default 2^14) per GPU; a step = solve + per-instance map digest of every tile (what config 5 keeps)
The default run (N = 1) also appends `other_workloads`: legs of arith_pedersen (the north-star target shape, at the metric's batch: 2^20 in
tiles), hash, grumpkin and ecdsa, each parity-checked against the oracle and carrying its own roofline / alu_roofline (--no-legs skips
them), then the same legs once more as the compact `legs`, and `summary` as the last key of the line.

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1 without a launcher: spawns the N ranks itself
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One process per GPU. Rank 0 prints ONE JSON line: whole-job witnesses/s (max-over-ranks time), per-rank rates, the digest of
digests over the per-instance witness-map digests of ALL ranks (identical for every N and tile size: the ranks provably solved
disjoint shards of one batch), `roofline` of the dominant kernel (algorithmic bytes of its launches / their HIP-event durations,
events on the batch's own streams; HBM traffic from in-run rocprofv3 PMC passes on rank 0's device), `end_to_end` (H2D of the inputs +
solve + D2H of the return witnesses), and `cpu_baseline` (the CPU oracle -- a port of the reference's in-order solver -- timed on every
host core on a bounded sample of the same workload, which is also the bit-exact parity check of the run; variants of BASELINE.md
section 2). Rank 0 computes `cpu_baseline`, `parity` and `roofline.traffic` for EVERY world size, after the timed region and the last barrier.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CLASS_KERNEL = ["light_level_kernel", "hash_level_kernel", "grumpkin_level_kernel", "brillig_level_kernel"]
# substring of the kernel's symbol in the rocprofv3 output
# substrings of the symbols of the kernels a class's launches run (a class may have several: the byte-message hashes have a kernel of their own)
KERNEL_SYMBOL = {"arith_level_kernel": ["arith_level_kernel", "arith_light_level_kernel"],  # (the second: a level whose light records ride along)
                 "inverse_batch_kernel": ["inverse_batch_kernel"],
                 "light_level_kernel": ["record_level_kernel<acvm::LightOp", "record_level_kernel<acvm::LightSlOp"],
                 "hash_level_kernel": ["hash_coop_level_kernel", "record_level_kernel<acvm::HashOp"],
                 "grumpkin_level_kernel": ["record_level_kernel<acvm::GrumpkinOp", "pedersen_quad_level_kernel", "record_level_kernel<acvm::EcdsaOp"],
                 "brillig_level_kernel": ["record_level_kernel<acvm::BrilligOp"]}
# FETCH_SIZE reports half the bytes of a 16 B/lane coalesced read on gfx950 (MI355X_MICROARCH.md, HBM section); both factors
# were re-derived on this part from a 4 GiB copy (profiles/traffic.json: x1.99998 / x1.00000)
PMC_READ_CORRECTION, PMC_WRITE_CORRECTION = 2.0, 1.0


def make_workload(args, first, n):
    """(circuit, initial ids, values of instances [first, first + n) of the global batch, name)"""
    from acvm_amd import synth
    if args.workload == "arith":
        circ, ids = synth.arithmetic_circuit(args.gates, seed=0xAC1D0002)
        values = synth.witness_batch(n, seed=0xAC1D0002, first_instance=first)
        name = f"{args.gates}-gate arithmetic-only ACIR"
    elif args.workload == "hash":
        circ, ids = synth.hash_circuit()
        values = synth.byte_batch(n, len(ids), first_instance=first)
        name = "sha256 + keccak256 (64-byte messages) + 96 RANGE(8) ACIR"
    elif args.workload == "grumpkin":
        circ, ids = synth.grumpkin_circuit()
        # the edge-case / flipped-signature pattern repeats every 1024 instances (row generation is host Python)
        import numpy as np
        base = synth.grumpkin_rows(min(n, 1024), first_instance=0)
        arr = np.frombuffer(synth.values_from_rows(base), dtype=np.uint8).reshape(len(base), -1)
        values = arr[(first + np.arange(n)) % len(base)].tobytes()
        name = "Pedersen + FixedBaseScalarMul + SchnorrVerify ACIR"
    elif args.workload == "ecdsa":
        circ, ids = synth.ecdsa_circuit()
        values = synth.ecdsa_batch(n, first_instance=first)
        name = "EcdsaSecp256k1 + EcdsaSecp256r1 ACIR (the reference's two vectors, every 7th instance tampered)"
    elif args.workload == "arith_pedersen":
        circ, ids = synth.arith_pedersen_circuit(args.gates, args.pedersen)
        values = synth.witness_batch(n, seed=0xAC1D0006, first_instance=first)
        name = f"{args.gates}-gate arithmetic + {args.pedersen} Pedersen ACIR"
    elif args.workload in ("mixed", "config5"):
        circ, ids = synth.mixed_circuit(args.gates)
        values = synth.witness_batch(n, seed=0xAC1D0005, first_instance=first)
        name = f"{args.gates}-opcode mixed ACIR (config-5 mix: arithmetic, range/logic, directives, memory, Brillig, hashes, Pedersen)"
        if args.workload == "config5":
            name += "; a step = solve + per-instance map digest of every tile"
    else:
        raise SystemExit(f"unknown workload {args.workload}")
    return circ, ids, values, name


def spawn_ranks(n):
    """--gpus N without a launcher: start the N ranks (one per GPU) and relay their output"""
    import acvm_amd
    have = acvm_amd.device_count()
    if have < n:
        raise SystemExit(f"bench.py: --gpus {n} but only {have} HIP device(s) visible; refusing to label a {have}-GPU run as {n}")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def pmc_pass(args, counter, inner=None):
    """One rocprofv3 pass (`--pmc <counter> --kernel-trace`, nothing else) over a one-tile run of this script (or over `inner`, another
    invocation of it): {kernel symbol: [sum of the counter over its launches, launches]}, or None if rocprofv3 is missing or fails."""
    import csv
    import shutil
    import tempfile
    if os.environ.get("ACVM_BENCH_NO_PMC") or not shutil.which("rocprofv3"):
        return None
    if inner is None:
        inner = [sys.executable, os.path.abspath(__file__), "--inner", "--workload", args.workload, "--gates", str(args.gates), "--pedersen", str(args.pedersen),
                 "--total-log2", str(args.eff_tile_log2), "--tile-log2", str(args.eff_tile_log2), "--steps", "1", "--warmup", "1"]
    tmp = tempfile.mkdtemp(prefix="acvm_pmc_", dir="/tmp")
    per = {}
    try:
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "pmc", "--"] + inner
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            return None
        for root, _, files in os.walk(tmp):
            for f in files:
                if f.endswith("counter_collection.csv"):
                    with open(os.path.join(root, f)) as fh:
                        for row in csv.DictReader(fh):
                            if row["Counter_Name"] == counter:
                                e = per.setdefault(row["Kernel_Name"], [0.0, 0])
                                e[0] += float(row["Counter_Value"])
                                e[1] += 1
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return per or None


def pick(per, substrs):
    """[sum, launches] over the kernels whose symbol holds one of substrs"""
    tot, n = 0.0, 0
    for name, (v, k) in (per or {}).items():
        if any(x in name for x in substrs):
            tot += v
            n += k
    return tot, n


def rocprof_names(per, substrs):
    """the rocprofv3 kernel names (demangled, arguments cut) of a kernel class that a PMC pass actually saw, with their launch counts"""
    out = {}
    for name, (v, k) in (per or {}).items():
        if any(x in name for x in substrs):
            short = name.split("(")[0].replace("void ", "")
            out[short] = out.get(short, 0) + k
    return out


def measure_traffic(args, kernel_substr):
    """HBM bytes per launch of the dominant kernel from the PMC counters, measured in THIS run: two rocprofv3 passes (FETCH_SIZE,
    WRITE_SIZE separately, kernel-trace only) over a one-tile run of this script. None if rocprofv3 is missing or fails."""
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tot, n = pick(pmc_pass(args, counter), kernel_substr)
        if not n:
            return None
        out[counter] = (tot / n, n)
    rd = out["FETCH_SIZE"][0] * 1024 * PMC_READ_CORRECTION
    wr = out["WRITE_SIZE"][0] * 1024 * PMC_WRITE_CORRECTION
    return {"bytes_per_launch": rd + wr, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "launches_counted": out["FETCH_SIZE"][1],
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over a one-tile run inside this bench run; "
                      "KiB x 1024, reads x2 (gfx950 correction of MI355X_MICROARCH.md), writes x1"}


# kernels of the integer-bound record classes (judged against an ALU roofline, not HBM)
ALU_KERNELS = {"grumpkin_level_kernel": KERNEL_SYMBOL["grumpkin_level_kernel"], "hash_level_kernel": KERNEL_SYMBOL["hash_level_kernel"],
               "brillig_level_kernel": KERNEL_SYMBOL["brillig_level_kernel"]}


def measure_alu(args, cls_kernel, cls_ms_per_tile, peak, probe_modmuls, sclk_mhz=None):
    """ALU roofline of an integer-bound kernel class, in modmul-equivalents: the VALU instructions its launches execute (PMC
    SQ_INSTS_VALU, own pass) divided by the VALU instructions of ONE Montgomery product (the same counter over the library's
    back-to-back fr29_mul probe, whose product count is known), per second of the class's HIP-event time, against the probe's rate."""
    per = pmc_pass(args, "SQ_INSTS_VALU")
    v_probe, n_probe = pick(per, ["modmul_rate_kernel"])
    v_cls, n_cls = pick(per, ALU_KERNELS[cls_kernel])
    if not n_probe or not n_cls or cls_ms_per_tile <= 0:
        return None
    valu_per_modmul = (v_probe / n_probe) / probe_modmuls  # wave instructions per lane product (x 64 lanes = per wave product)
    solves = 2  # the profiled inner run: one warm-up and one timed solve of one tile
    equiv = v_cls / solves / valu_per_modmul
    achieved = equiv / (cls_ms_per_tile / 1e3)
    import acvm_amd
    n_simd = 4 * acvm_amd.modmul_probe_cus()
    return {"bound": "valu", "unit": "modmul/s", "achieved": achieved, "peak": peak, "frac": achieved / peak,
            # the hardware-referenced figure beside the probe-referenced one: the share of the chip's VALU issue slots the class's launches fill
            # (a better product routine would lower `frac` and leave this one alone)
            "valu_issue_frac": None if not sclk_mhz else (v_cls / solves) * 4.0 / (n_simd * (cls_ms_per_tile / 1e3) * sclk_mhz * 1e6),
            "valu_issue_definition": "SQ_INSTS_VALU of the class's launches per tile x 4 cycles / (SIMDs x the class's HIP-event seconds x the shader clock sampled under this workload)",
            "sclk_mhz_under_load": sclk_mhz, "simds": n_simd,
            "modmul_equivalents_per_tile": equiv, "valu_wave_insts_per_tile": v_cls / solves, "valu_wave_insts_per_wave_modmul": valu_per_modmul * 64,
            "kernel_ms_per_tile": cls_ms_per_tile, "kernels": ALU_KERNELS[cls_kernel],
            "definition": "modmul-equivalents = SQ_INSTS_VALU of the class's launches / SQ_INSTS_VALU per product of the back-to-back fr29_mul probe "
                          "(acvm_debug_modmul_rate: 8 interleaved chains per SIMD); peak = that probe's measured modmul/s in this run"}


# Base-field products (multiplications + squarings of secp_device.hpp) of ONE verification, counted from the routine (NOTEBOOK.md section 6):
# the curve equation 3 (x^2, x^3, y^2: the batch's keys are on the curve, so the given y is the root and the square-root chain of the
# decompression -- 253 S + 13 / 7 M -- never runs; round 3 counted it); the window table {Q .. 8Q} 119 (+ 8 for the beta x of secp256k1); the ladder
# 128 / 256 doublings x 7 / 8 and on average 61 / 60.5 mixed additions x 11; 16 additions of generator-table points (16-bit windows) x 11; the final
# x 2; secp256r1: 5 more for the Montgomery form (x and y in, the result out, R^3 behind two inversions).
# Not counted: three safegcd inversions and ~12 products of the scalar field per verification (about 8 % of the instructions).
ECDSA_PRODUCTS = {0: 3 + 119 + 8 + 128 * 7 + 61 * 11 + 176 + 2, 1: 3 + 120 + 256 * 8 + 665.5 + 176 + 2 + 5}


def ecdsa_alu_roofline(args, tile, kernel_ms, sclk_mhz=None):
    """ALU roofline of the ECDSA kernel in base-field products: what one launch computes (both curves, one verification each per instance)
    per second of its HIP-event time, against the back-to-back s29_mul / s29_sqr probe of each curve measured in this run; and, from one
    SQ_INSTS_VALU pass, the instructions per counted product and the kernel's share of the chip's VALU issue slots"""
    import acvm_amd
    peaks = [acvm_amd.secp_rate(c, 400, 8)[0] for c in (0, 1)]
    per_instance = ECDSA_PRODUCTS[0] + ECDSA_PRODUCTS[1]
    at_peak_s = tile * (ECDSA_PRODUCTS[0] / peaks[0] + ECDSA_PRODUCTS[1] / peaks[1])
    achieved = tile * per_instance / (kernel_ms / 1e3) if kernel_ms > 0 else 0.0
    hw = {}
    v, n = pick(pmc_pass(args, "SQ_INSTS_VALU"), ["EcdsaOp"]) if args is not None else (0, 0)
    if n:
        solves = 2  # the profiled inner run: one warm-up and one timed solve of one tile
        n_simd = 4 * acvm_amd.modmul_probe_cus()
        hw = {"valu_wave_insts_per_tile": v / solves, "valu_per_counted_product": (v / solves) * 64.0 / (tile * per_instance),
              "valu_issue_frac": None if not sclk_mhz or kernel_ms <= 0 else (v / solves) * 4.0 / (n_simd * (kernel_ms / 1e3) * sclk_mhz * 1e6),
              "sclk_mhz_under_load": sclk_mhz, "simds": n_simd,
              "valu_issue_definition": "SQ_INSTS_VALU of the kernel's launches per tile x 4 cycles / (SIMDs x its HIP-event seconds x the shader clock sampled under this workload)"}
    return {"bound": "valu", "unit": "field products/s", "achieved": achieved, "peak": tile * per_instance / at_peak_s,
            "frac": at_peak_s / (kernel_ms / 1e3) if kernel_ms > 0 else None, **hw,
            "products_per_verification": {"secp256k1": ECDSA_PRODUCTS[0], "secp256r1": ECDSA_PRODUCTS[1]},
            "probe_products_per_s": {"secp256k1": peaks[0], "secp256r1": peaks[1]}, "kernel_ms_per_tile": kernel_ms,
            "definition": "products (multiplications and squarings of the curve's base field) one launch executes, counted from the routine, / its HIP-event "
                          "time; peak = the time the same products take in the back-to-back s29_mul / s29_sqr probe of the 29-bit working form (acvm_debug_secp_rate, 8 chains per SIMD)"}


class ClockSampler:
    """shader clock (MHz) while the timed region runs: rocm-smi polled from a thread; median of the samples, or None"""

    def __init__(self, device, enabled=True):
        import shutil
        import threading
        self.samples, self.stop, self.device = [], False, device
        self.th = threading.Thread(target=self._run, daemon=True) if enabled and shutil.which("rocm-smi") and not os.environ.get("ACVM_BENCH_NO_PMC") else None

    def _run(self):
        import re
        while not self.stop:
            try:
                out = subprocess.run(["rocm-smi", "-d", str(self.device), "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
                if m:
                    self.samples.append(int(m.group(1)))
            except (OSError, subprocess.SubprocessError):
                return
            time.sleep(0.05)

    def __enter__(self):
        if self.th:
            self.th.start()
        return self

    def __exit__(self, *exc):
        self.stop = True
        if self.th:
            self.th.join(timeout=15)
        return False

    def median(self):
        s = sorted(self.samples)
        return s[len(s) // 2] if s else None


def timed_one_core(ob, oc, ids, values, row, n0, n_max, mode, target_s=1.2):
    """median of three one-core runs of at least about a second each (the sample grows until one run takes target_s): (instances, seconds)"""
    n = max(1, min(n0, n_max))
    while True:
        c = time.perf_counter()
        ob.solve_batch(oc, ids, values[: n * row], n, want_witness=False, n_threads=1, mode=mode)
        dt = time.perf_counter() - c
        if dt >= 0.8 * target_s or n >= n_max:
            break
        n = min(n_max, max(n + 1, int(n * target_s / max(dt, 1e-4)) + 1))
    runs = [dt]
    for rep in range(2):
        c = time.perf_counter()
        ob.solve_batch(oc, ids, values[: n * row], n, want_witness=False, n_threads=1, mode=mode)
        runs.append(time.perf_counter() - c)
    return n, sorted(runs)[1]


from acvm_amd.shard import cpu_budget  # (CPUs this process may use, the cgroup quota, os.cpu_count())


def cpu_baseline_and_parity(args, data, ids, values, batch, sh, tile, row, inst_digests):
    """The CPU oracle (a port of the reference's in-order solver, oracle/) on a bounded sample of tile 0: the parity check of the run
    (results, assigned sets, every witness, digests) and the cpu_baseline of the line -- median of three timed runs per variant.
    Rank 0 runs it for every world size (after the timed region and the last barrier: the other ranks are done), on ALL host cores."""
    import numpy as np
    from oracle import binding as ob
    cores, quota, host_cores = cpu_budget()
    per = {"arith": 96, "hash": 1024, "grumpkin": 256, "ecdsa": 64, "arith_pedersen": 48, "mixed": 32, "config5": 1}[args.workload]  # about a second of all host cores
    sh.load_tile(0)
    batch.solve()
    results0 = batch.results()
    oc = ob.Circuit(data)
    # every CPU the process may use (the cgroup's quota, not the host's count) is offered to the oracle; the thread count reported is the one that
    # solves fastest: twice the budget (SMT, throttling slack), the budget and half of it on a short sample each, then the timed runs
    threads, tried = cores, {}
    if args.workload != "config5" and cores >= 8:
        probe = min(tile, max(64, per * cores // 8))
        for t in sorted({min(2 * cores, host_cores), cores, max(1, cores // 2)}, reverse=True):
            c0 = time.perf_counter()
            ob.solve_batch(oc, ids, values[: probe * row], probe, want_witness=False, n_threads=t, mode=ob.MODE_CACHE_INV)
            tried[t] = round(probe / (time.perf_counter() - c0), 1)
        threads = max(tried, key=tried.get)
    sample = args.cpu_sample or min(tile, max(threads if args.workload == "config5" else 64, per * threads))
    sample_vals = values[: sample * row]
    runs = []
    for rep in range(3):
        c0 = time.perf_counter()
        ores, oasg, ovals = ob.solve_batch(oc, ids, sample_vals, sample, want_witness=True, n_threads=threads, mode=ob.MODE_CACHE_INV)
        runs.append(time.perf_counter() - c0)
    cpu_s = sorted(runs)[len(runs) // 2]
    if args.workload == "config5":  # full maps of a 10^6-opcode circuit: compare through the digests and the return witnesses
        dig = batch.digest(0, sample)
        ok = all(results0[j].as_tuple() == ores[j].as_tuple() for j in range(sample))
        ok = ok and all(bytes(dig[j]) == ob.witness_map_digest(oasg[j], ovals[j]) for j in range(min(sample, 4)))
        n_dig = min(sample, 4)
    else:
        gasg, gvals = batch.witness_map(0, sample)
        ok = all(results0[j].as_tuple() == ores[j].as_tuple() for j in range(sample))
        ok = ok and bool(np.array_equal(gasg, oasg[:, : gasg.shape[1]])) and bool(np.array_equal(gvals, ovals[:, : gvals.shape[1]]))
        n_dig = min(sample, 16) if inst_digests is not None else 0
        if n_dig:  # the digests that feed the digest of digests, against hashlib over the oracle's maps
            ok = ok and all(bytes(inst_digests[j]) == ob.witness_map_digest(oasg[j], ovals[j]) for j in range(n_dig))
    parity = {"checked_instances": sample, "bit_exact": bool(ok), "digests_checked": n_dig}
    cpu = {"value": sample / cpu_s, "unit": "witnesses/s", "cores": min(threads, cores), "threads": threads, "host_cores": host_cores, "cgroup_cpu_quota": quota,
           "kind": "port", "variant": "cpu_ref_dense_mt", "runs_s": [round(x, 3) for x in runs], "threads_tried_witnesses_per_s": tried,
           "sample": f"{sample} instances of the same circuit, oracle/ (gcc -O3 -march=native; dense witness vector, constant divisors inverted once, one solver "
                     f"object per thread), {threads} threads on the {cores} CPUs this process may use (cgroup quota {quota}, host os.cpu_count() = {host_cores}; the fastest "
                     f"of twice / once / half that many threads), median of {len(runs)} runs: {cpu_s:.2f} s"}
    if args.workload != "config5":  # the two single-core variants, at least about a second per run, median of three
        n_max = min(tile, len(values) // row)
        one, one_s = timed_one_core(ob, oc, ids, values, row, int(round(1.2 * sample / (cpu_s * threads))), n_max, ob.MODE_CACHE_INV)
        cpu["cpu_ref_dense"] = {"value": one / one_s, "unit": "witnesses/s", "cores": 1, "sample": f"{one} instances, median of 3 runs: {one_s:.2f} s"}
        few, few_s = timed_one_core(ob, oc, ids, values, row, max(1, one // 6), n_max, ob.MODE_SPARSE_MAP)
        cpu["cpu_ref_faithful"] = {"value": few / few_s, "unit": "witnesses/s", "cores": 1,
                                   "sample": f"{few} instances, median of 3 runs: {few_s:.2f} s; BTreeMap-shaped witness map, one field inversion per solved witness (the reference's data structures)"}
    return cpu, parity


def roofline_block(args, st, tile, world, sclk_mhz, pmc=True):
    """`roofline` (+ `alu_roofline`) of the dominant kernel of the last profiled tile: algorithmic bytes of its launches / their HIP-event
    durations against the HBM peak; HBM traffic, VALU issue fraction and the ALU roofline from in-run PMC passes (rank 0 on its own device,
    for every world size: the passes profile a one-tile run of the size this rank's tiles have)."""
    import acvm_amd
    arith_ms, dyn_ms, cls_ms = st["arith_kernel_ms"], st["dyn_kernel_ms"], list(st["class_kernel_ms"])
    cand = {"arith_level_kernel": (arith_ms, st["arith_algorithmic_bytes_per_instance"], st["n_arith_launches"]),
            "inverse_batch_kernel": (dyn_ms, st["dyn_algorithmic_bytes_per_instance"], 0)}
    for k in range(4):
        cand[CLASS_KERNEL[k]] = (cls_ms[k], st["class_algorithmic_bytes_per_instance"][k], 0)
    # byte planes (acvm_amd.h acvm_stats_t): the hash kernel reads 4 bytes instead of a 32-byte row for those inputs -- its own bytes, not the reference's unit
    saved = 28 * st.get("n_byte_plane_reads", 0)
    cand[CLASS_KERNEL[1]] = (cls_ms[1], st["class_algorithmic_bytes_per_instance"][1] - saved, 0)
    dominant = "arith_level_kernel" if args.workload in ("arith", "config5") else max(cand, key=lambda k: cand[k][0])
    k_ms, k_bytes, k_launches = cand[dominant]
    if dominant in ALU_KERNELS and st["solve_device_ms"] > 0 and k_ms > st["solve_device_ms"]:
        # the lanes of the class (Pedersen | FixedBase, Schnorr, ECDSA) ran side by side: their HIP-event durations overlap and add up to more
        # than the solve; the class was busy for the solve's device time
        k_ms = st["solve_device_ms"]
    achieved = k_bytes * tile / (k_ms / 1e3) / 1e9 if k_ms > 0 else 0.0
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "kernel": dominant, "kernel_class_symbols": KERNEL_SYMBOL.get(dominant, [dominant]), "kernel_ms_per_tile": k_ms, "algorithmic_bytes_per_tile": k_bytes * tile,
            "kernel_timing": "HIP events around every launch of the last tile of the last timed step (on the launching stream)",
            "launches_per_tile_all_kernels": st["n_kernel_launches"],
            "other_kernels_ms_per_tile": {k: v[0] for k, v in cand.items() if k != dominant and v[0] > 0}}
    if k_launches:
        roof["kernel_launches_per_tile"] = k_launches
        roof["kernel_avg_launch_ms"] = k_ms / k_launches
        roof["algorithmic_bytes_per_launch"] = k_bytes * tile / k_launches
    alu = None
    if pmc:
        tr = measure_traffic(args, KERNEL_SYMBOL.get(dominant, [dominant]))
        if tr:
            roof["traffic"] = tr["bytes_per_launch"]
            roof["traffic_detail"] = tr
            # `kernel` above is the library's class of launches; these are the names rocprofv3 prints for them (what profiles/*_profile_*.txt lists)
            roof["rocprof_kernels"] = rocprof_names(pmc_pass(args, "FETCH_SIZE"), KERNEL_SYMBOL.get(dominant, [dominant]))
        if dominant == "arith_level_kernel" and k_launches:
            # share of the VALU issue slots the gate kernel fills: its VALU wave instructions (PMC, own pass) x 4 cycles (a wave64 instruction
            # on a 16-lane SIMD) / (SIMDs x launch time x the shader clock sampled while the kernel ran)
            v, n = pick(pmc_pass(args, "SQ_INSTS_VALU"), KERNEL_SYMBOL[dominant]) if sclk_mhz else (0, 0)
            if n and sclk_mhz:
                n_simd = 4 * acvm_amd.modmul_probe_cus()
                roof["valu_issue_frac"] = (v / n) * 4.0 / (n_simd * (k_ms / k_launches / 1e3) * sclk_mhz * 1e6)
                roof["valu_issue_detail"] = {"valu_wave_insts_per_launch": v / n, "sclk_mhz_under_load": sclk_mhz, "simds": n_simd,
                                             "definition": "SQ_INSTS_VALU per launch x 4 cycles / (SIMDs x launch seconds x sclk)"}
        if args.workload == "ecdsa":
            alu = ecdsa_alu_roofline(args, tile, k_ms, sclk_mhz)
            roof["note"] = "integer-ALU bound class: the HBM fraction is for information, see alu_roofline"
        elif dominant in ALU_KERNELS:
            peak, probe_n = acvm_amd.modmul_rate(400, 8)
            alu = measure_alu(args, dominant, k_ms, peak, 8 * 100 * 2 * 256 * acvm_amd.modmul_probe_cus(), sclk_mhz) or {"bound": "valu", "unit": "modmul/s", "peak": peak, "achieved": None, "frac": None}
            roof["note"] = "integer-ALU bound class: the HBM fraction is for information, see alu_roofline"
    return roof, alu


def run_leg(name, total_log2=16, tile_log2=16, steps=3, warmup=2, pmc=True):
    """one of the other workloads as a short leg of the default run (N = 1), parity-checked. A step is what it is for the metric's workload:
    per tile, import of the resident inputs + solve (arith_pedersen, the north-star target shape, runs at the metric's batch: 2^20 in tiles)"""
    import acvm_amd
    from acvm_amd import tiling
    a = argparse.Namespace(workload=name, gates=10000, pedersen=8, cpu_sample={"hash": 4096, "grumpkin": 512, "ecdsa": 512, "arith_pedersen": 256}.get(name, 0),
                           eff_tile_log2=tile_log2)
    n, tile = 1 << total_log2, 1 << tile_log2
    circ, ids, values, wname = make_workload(a, 0, n)
    data = circ.to_bytes()
    gc = acvm_amd.Circuit(data)
    sh = tiling.ResidentShard(gc, ids, values, n, tile)
    batch = sh.batch
    n_tiles = len(sh.starts)
    batch.set_profiling(True)
    for _ in range(warmup):
        sh.solve_pass()
    batch.set_profiling(False)
    acvm_amd.synchronize()
    solve_ms = 0.0
    t0 = time.perf_counter()
    for i in range(steps):
        for k in range(n_tiles):
            if i == steps - 1 and k == n_tiles - 1:
                batch.set_profiling(True)
            sh.load_tile(k)
            sh.solve_tile(k)
            if i == steps - 1:
                solve_ms += batch.stats()["solve_device_ms"]
    acvm_amd.synchronize()
    elapsed = time.perf_counter() - t0
    st = batch.stats()
    batch.set_profiling(False)
    sclk = None
    if pmc and name in ("hash", "grumpkin", "ecdsa"):
        # the shader clock under THIS workload, for the hardware-referenced ALU figures: sampled over ~0.8 s of back-to-back passes outside the timed region
        # (the sampler forks rocm-smi: it would disturb the sub-millisecond steps themselves)
        with ClockSampler(acvm_amd.current_device()) as cs:
            t_end = time.perf_counter() + 0.8
            while time.perf_counter() < t_end:
                sh.solve_pass()
            acvm_amd.synchronize()
        sclk = cs.median()
    roof, alu = roofline_block(a, st, tile, 1, sclk, pmc=pmc)
    # the whole step against the HBM peak: what the import moves (the caller's 32 bytes in, the row out, 4 bytes per byte plane) + the solve's bytes
    step_bytes = (len(ids) * 64 + 4 * st.get("n_byte_planes", 0) + st["algorithmic_bytes_per_instance"] - 28 * st.get("n_byte_plane_reads", 0)) * tile
    step_ms = elapsed / steps / n_tiles * 1e3
    roof["per_step"] = {"bytes_per_tile": step_bytes, "ms_per_tile": step_ms, "achieved": step_bytes / (step_ms / 1e3) / 1e9, "unit": "GB/s",
                        "frac": step_bytes / (step_ms / 1e3) / 1e9 / HBM_PEAK_GBS, "what": "import of the tile's resident inputs + every kernel of its solve, wall clock"}
    cpu, parity = cpu_baseline_and_parity(a, data, ids, values, batch, sh, tile, len(ids) * 32, None)
    sh.free()
    out = {"workload": wname, "value": n * steps / elapsed, "unit": "witnesses/s", "instances": n, "tile_instances": tile, "steps": steps,
           "ms_per_step": elapsed / steps * 1e3, "step": "per tile: import of the resident inputs + solve",
           "solve_device_ms_last_step": solve_ms, "value_solve_only": n / (solve_ms / 1e3) if solve_ms > 0 else None,
           "levels": st["n_levels"], "launches": st["n_kernel_launches"], "roofline": roof, "alu_roofline": alu,
           "cpu_baseline": {k: cpu[k] for k in ("value", "unit", "cores", "host_cores", "kind", "sample")}, "parity": parity}
    return out


# kernels of the level schedule by class, for the config-5 leg's instruction accounting (substrings of the rocprofv3 kernel names; the kernels that
# build lookup tables -- pedersen_window_table_kernel and friends, once per process -- are none of them)
CONFIG5_CLASSES = {"gates (+ fused light records)": ["arith_level_kernel", "arith_light_level_kernel", "arith_l"], "inversion batches": ["inverse_batch_kernel"],
                   "Pedersen": ["pedersen_quad_level_kernel", "pedersen_bundle_level_kernel"], "digest leaves": ["digest_fold_level_kernel"],
                   "hashes": ["hash_coop_level_kernel", "HashOp"], "light / inlined Brillig": ["LightOp", "LightSlOp"], "Grumpkin": ["GrumpkinOp"], "Brillig VM": ["BrilligOp"]}


def run_config5_leg(tile_log2=13, timed_tiles=2, audit=32, inner=False):
    """BASELINE config 5 at circuit size as a leg of the default run: the 10^6-opcode mixed circuit (SURVEY 8d generator), ONE handle with
    witness-slot liveness reuse and the digest folded into the solve, `timed_tiles` tiles of 2^tile_log2 fresh instances (a step = ACVM::new of the
    tile from its resident inputs + solve + the tile's per-instance digests; no per-launch events in the timed tiles), one more tile with per-launch HIP
    events for the classes' times, an audit sample of the first timed tile re-solved by the CPU oracle (results, return witnesses, map digests bit
    for bit), and one in-run PMC pass (SQ_INSTS_VALU) over the same sequence for the instructions per class and the tile-wide VALU issue fraction.
    inner: the run that pass profiles (no oracle, nothing printed)."""
    import acvm_amd
    from acvm_amd import synth
    tile = 1 << tile_log2
    t0 = time.perf_counter()
    circ, ids = synth.mixed_circuit(1_000_000)
    data = circ.to_bytes()
    t1 = time.perf_counter()
    gc = acvm_amd.Circuit(data)
    ret = gc.witness_set("return_values")
    batch = acvm_amd.Batch(gc, tile, ids, reuse_slots=True, keep=ret)  # (slot reuse folds the digest)
    t2 = time.perf_counter()
    row = len(ids) * 32
    tiles = [synth.witness_batch(tile, seed=0xAC1D0005, first_instance=k * tile) for k in range(timed_tiles + 2)]
    resident = [acvm_amd.DeviceBuffer(t) for t in tiles]  # like the metric's own step: inputs resident in HBM when the timed region starts
    batch.set_initial_witness_device(resident[0].ptr)  # warm-up tile: tables built, clocks up, the exact path's side table allocated (its edge-case instances)
    batch.solve()
    batch.digest()
    acvm_amd.synchronize()
    step_ms, dev_ms, not_solved, first = [], [], 0, None
    with ClockSampler(acvm_amd.current_device(), enabled=not inner) as clock:
        for k in range(1, timed_tiles + 1):
            w0 = time.perf_counter()
            batch.set_initial_witness_device(resident[k].ptr)
            not_solved += batch.solve()
            dig = batch.digest()
            step_ms.append((time.perf_counter() - w0) * 1e3)
            dev_ms.append(batch.stats()["solve_device_ms"])
            if k == 1 and not inner:
                first = (batch.results(), dig, [batch.extract(ret, j, 1)[0] for j in range(0, tile, max(tile // audit, 1))][:audit])
    if inner:
        batch.free()
        for r in resident:
            r.free()
        return None
    sclk = clock.median()
    batch.set_profiling(True)  # the classes' own times: per-launch events cost a tenth of a tile of 1 200 launches, so they bracket a tile of their own
    batch.set_initial_witness_device(resident[timed_tiles + 1].ptr)
    batch.solve()
    st = batch.stats()
    batch.set_profiling(False)
    # ---- audit of the first timed tile (instances tile .. 2 tile - 1 of the synthetic batch: no edge cases among them)
    from oracle import binding as ob
    picks = list(range(0, tile, max(tile // audit, 1)))[:audit]
    sub = b"".join(tiles[1][j * row:(j + 1) * row] for j in picks)
    threads = min(len(picks), cpu_budget()[0])
    oc, oracle_runs = ob.Circuit(data), []
    for rep in range(3):  # (the median of three)
        a0 = time.perf_counter()
        ores, oasg, ovals = ob.solve_batch(oc, ids, sub, len(picks), n_threads=threads)
        oracle_runs.append(time.perf_counter() - a0)
    oracle_s = sorted(oracle_runs)[1]
    res, dig1, kept = first
    ok = True
    for i, j in enumerate(picks):
        ok &= res[j].as_tuple() == ores[i].as_tuple()
        ok &= bytes(dig1[j]) == ob.witness_map_digest(oasg[i], ovals[i])
        if ores[i].status == 0 and ret:
            ok &= all(bytes(kept[i][n]) == bytes(ovals[i][w]) for n, w in enumerate(ret))
    batch.free()
    for r in resident:
        r.free()
    ms = sum(step_ms) / len(step_ms)
    solve_ms = sum(dev_ms) / len(dev_ms)
    cls_names = ["light (range / logic / directives / memory / inlined Brillig)", "hashes", "Grumpkin + Pedersen + ECDSA", "Brillig VM"]
    classes = {"arith_level_kernel": {"ms_profiled_tile": st["arith_kernel_ms"], "algorithmic_bytes_per_tile": st["arith_algorithmic_bytes_per_instance"] * tile},
               "inverse_batch_kernel": {"ms_profiled_tile": st["dyn_kernel_ms"], "algorithmic_bytes_per_tile": st["dyn_algorithmic_bytes_per_instance"] * tile}}
    for k in range(4):
        classes[cls_names[k]] = {"ms_profiled_tile": st["class_kernel_ms"][k], "algorithmic_bytes_per_tile": st["class_algorithmic_bytes_per_instance"][k] * tile}
    for c in classes.values():  # (the classes run side by side on their own streams: the sum of their times exceeds the step)
        c["achieved_GBps"] = c["algorithmic_bytes_per_tile"] / (c["ms_profiled_tile"] / 1e3) / 1e9 if c["ms_profiled_tile"] > 0 else None
        c["frac_of_hbm_peak"] = None if c["achieved_GBps"] is None else c["achieved_GBps"] / HBM_PEAK_GBS
    # ---- instructions: one PMC pass over the same sequence of tiles (this function with inner=True), summed per class and per solve
    valu = None
    per = pmc_pass(None, "SQ_INSTS_VALU", inner=[sys.executable, os.path.abspath(__file__), "--inner-config5-leg", "--tile-log2", str(tile_log2)])
    if per:
        n_solves = pick(per, ["import_witness_kernel"])[1]  # one import per tile of the profiled sequence (it also resets the event words: there is no reset launch to count)
        by_class = {name: pick(per, subs)[0] / max(n_solves, 1) for name, subs in CONFIG5_CLASSES.items()}
        total = sum(by_class.values())
        n_simd = 4 * acvm_amd.modmul_probe_cus()
        valu = {"valu_wave_insts_per_tile": total, "by_class": {k: v for k, v in by_class.items() if v}, "solves_profiled": n_solves, "sclk_mhz_under_load": sclk, "simds": n_simd,
                "solve_device_ms": solve_ms,
                "valu_issue_frac": None if not sclk else total * 4.0 / (n_simd * (solve_ms / 1e3) * sclk * 1e6),
                "definition": "sum over the level schedule's kernels of SQ_INSTS_VALU per solve (rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace over this leg's own sequence of tiles, "
                              "inside this bench run; table-building kernels excluded) x 4 cycles / (SIMDs x the tile's solve_device_ms x the shader clock sampled while the timed tiles ran)"}
    achieved = st["algorithmic_bytes_per_instance"] * tile / (ms / 1e3) / 1e9
    return {"workload": "10^6-opcode mixed ACIR (config 5: 94 % arithmetic, range / logic, directives, memory, Brillig, hashes, Pedersen), "
                        f"tiles of {tile} instances through one handle, witness-slot reuse, digest folded into the solve",
            "value": tile / (ms / 1e3), "unit": "witnesses/s", "instances": tile * timed_tiles, "tile_instances": tile, "steps": timed_tiles, "ms_per_step": ms,
            "step": "per tile: ACVM::new (import of the tile's resident inputs) + solve + per-instance map digests", "ms_of_each_step": step_ms,
            "solve_device_ms_of_each_step": dev_ms, "not_solved": not_solved, "opcodes": st["n_opcodes"], "witnesses_per_instance": st["n_witnesses"],
            "table_rows": st["n_table_rows"], "levels": st["n_levels"], "launches": st["n_kernel_launches"], "launches_by_stream": dict(zip(["main", "inversions", "lane0 (hashes)", "lane1 (Pedersen)", "lane2 (Brillig VM)", "digest"], st["n_stream_launches"])),
            "cross_stream_waits": st["n_stream_waits"], "brillig_opcodes_inlined": st["n_brillig_inlined"],
            "generate_s": round(t1 - t0, 1), "parse_plan_alloc_s": round(t2 - t1, 1), "plan_ms": st["plan_ms"],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "the whole step (every class of the level schedule; they overlap)", "algorithmic_bytes_per_tile": st["algorithmic_bytes_per_instance"] * tile,
                         "gate_kernel_frac_inside_the_tile": classes["arith_level_kernel"]["frac_of_hbm_peak"], "classes": classes,
                         "valu_issue_frac": None if not valu else valu["valu_issue_frac"], "valu": valu,
                         "note": "the tile is bound by VALU issue, not by HBM: its kernels' instructions add up (valu.by_class); the HBM fraction is for information"},
            "alu_roofline": None,
            "cpu_baseline": {"value": len(picks) / oracle_s, "unit": "witnesses/s", "cores": threads, "host_cores": os.cpu_count(), "cgroup_cpu_quota": cpu_budget()[1], "kind": "port",
                             "runs_s": [round(x, 2) for x in oracle_runs],
                             "sample": f"the {len(picks)} audit instances, one oracle thread each on {threads} threads, median of 3 runs: {oracle_s:.1f} s (a 10^6-opcode instance is ~1.4 s of one core)"},
            "parity": {"bit_exact": bool(ok), "instances": picks, "of_tile": 1, "checked": "result records, return witnesses, map digests (hashlib over the oracle's full map)"}}


def leg_summary(leg):
    """[witnesses/s, ms per step, HBM fraction of the dominant kernel, ALU fraction or None, parity ok] of a leg, for the tail of the line"""
    if not leg or "error" in leg:
        return None
    r, a = leg.get("roofline") or {}, leg.get("alu_roofline") or {}
    rnd = lambda x, k: None if x is None else round(x, k)
    return [round(leg["value"]), round(leg["ms_per_step"], 3), rnd(r.get("frac"), 4), rnd(a.get("frac"), 4), bool((leg.get("parity") or {}).get("bit_exact"))]


def end_to_end_node(gc, ids, values, total, tile, n_dev_rank):
    """PCIe-inclusive rate through the node-level driver (acvm_node_solve: host buffers in, the count of unsolved instances + return witnesses out):
    pinned double-buffered uploads beside the solves, the exact path of a tile beside the next one"""
    import acvm_amd
    ret = gc.witness_set("return_values")
    node = acvm_amd.Node(gc, ids, keep=ret, devices=[acvm_amd.current_device()], tile=tile)
    node.solve(values[: tile * len(ids) * 32], tile, results=False)  # warm-up: staging buffers touched, tables built
    best = None
    for rep in range(2):
        t0 = time.perf_counter()
        not_solved, _, kept, asg, dig = node.solve(values, total, results=False, digests=False)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, node.stats(), not_solved)
    node.free()
    dt, st, not_solved = best
    return {"value": total / dt, "unit": "witnesses/s", "total_ms": dt * 1e3, "solve_device_ms": st["solve_device_ms"], "h2d_exposed_ms": st["h2d_wait_ms"],
            "export_ms": st["export_ms"], "exact_path_instances": st["exact_instances"], "not_solved": not_solved, "input_bytes_per_witness": len(ids) * 32,
            "returned_per_witness": f"{len(ret)} return witness(es) x 32 B",
            "note": "acvm_node_solve on pageable host buffers over PCIe (the library's own pinned staging, uploads of tile k + 1 beside the solve of tile k, "
                    "diverging instances re-solved beside the next tile); `value` of the line keeps inputs resident"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)  # the device reaches its clocks after about two solves (profiles/README.md)
    ap.add_argument("--workload", default="arith", choices=["arith", "hash", "grumpkin", "ecdsa", "arith_pedersen", "mixed", "config5"])
    ap.add_argument("--gates", type=int, default=None, help="opcodes of the synthetic circuit (default 10000; 1000000 for config5)")
    ap.add_argument("--pedersen", type=int, default=8)
    ap.add_argument("--total-log2", type=int, default=None, help="global batch = 2^this (default: 20 for arith = the metric; 14 per GPU for config5; 16 per GPU otherwise)")
    ap.add_argument("--tile-log2", type=int, default=None, help="instances per batch handle = 2^this (default 17: 10k gates x 2^17 = 42 GB of witness table, measured 2^16 / 2^17 / 2^18: 5.22 / 5.41 / 5.30 M witnesses/s; 12 for config5)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="instances for the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-digest", action="store_true", help="skip the digest-of-digests pass")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the other_workloads legs of the default run")
    ap.add_argument("--no-pipeline", action="store_true", help="plain acvm_batch_solve per tile instead of acvm_batch_solve_then_import (A/B of the tile boundary)")
    ap.add_argument("--inner", action="store_true", help=argparse.SUPPRESS)  # the one-tile run the PMC passes profile
    ap.add_argument("--inner-config5-leg", action="store_true", help=argparse.SUPPRESS)  # the config-5 leg's sequence of tiles, for its PMC pass
    args = ap.parse_args()
    if args.inner_config5_leg:
        import acvm_amd
        acvm_amd.set_device(0)
        run_config5_leg(tile_log2=args.tile_log2 or 13, inner=True)
        return
    if args.gates is None:
        args.gates = 1000000 if args.workload == "config5" else 10000
    if args.tile_log2 is None:
        args.tile_log2 = 12 if args.workload == "config5" else 17
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.inner:
        spawn_ranks(args.gpus)

    import numpy as np
    import acvm_amd
    from acvm_amd import shard, tiling

    rank, local_rank, world = shard.env_rank()
    # timing barrier / reductions only; the data path has no exchange step, so a CPU (gloo) group is enough and keeps
    # torch's bundled HIP runtime out of this process (the kernels run on the system ROCm runtime of libacvm_amd.so)
    dist = shard.init_group(rank, world)
    n_dev = acvm_amd.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py: no HIP device visible; this benchmark has no CPU fallback")
    if world > 1 and n_dev < world and os.environ.get("LOCAL_WORLD_SIZE", str(world)) == str(world) and not os.environ.get("ACVM_BENCH_SHARE_GPU"):
        raise SystemExit(f"bench.py: {world} ranks but {n_dev} HIP device(s) visible (set ACVM_BENCH_SHARE_GPU=1 to let ranks share a GPU in tests)")
    acvm_amd.set_device(local_rank % n_dev)

    config5 = args.workload == "config5"
    strong = args.workload == "arith" or (args.total_log2 is not None and not config5)
    if strong:
        total = 1 << (args.total_log2 if args.total_log2 is not None else 20)
    else:
        total = (1 << (args.total_log2 if args.total_log2 is not None else (14 if config5 else 16))) * world
    if total % world or ((total // world) % shard.DIGEST_CHUNK and not args.inner):
        raise SystemExit(f"bench.py: the global batch {total} must split into {world} shards that are multiples of {shard.DIGEST_CHUNK}")
    n_rank = total // world
    first = rank * n_rank
    tile = min(1 << args.tile_log2, n_rank)
    args.eff_tile_log2 = tile.bit_length() - 1  # what the PMC passes profile: one tile of the size this run uses
    circ, ids, values, workload_name = make_workload(args, first, n_rank)
    data = circ.to_bytes()
    gc = acvm_amd.Circuit(data)
    row = len(ids) * 32

    t_h2d0 = time.perf_counter()
    sh = tiling.ResidentShard(gc, ids, values, n_rank, tile)  # H2D of the whole shard: outside the timed region
    acvm_amd.synchronize()
    h2d_resident_s = time.perf_counter() - t_h2d0  # includes the batch handle (plan + table allocation)
    batch = sh.batch
    n_tiles = len(sh.starts)

    def barrier():
        shard.barrier(dist)
        acvm_amd.synchronize()

    batch.set_profiling(True)  # warm-up with per-launch events, so that the event pool exists before the timed region
    for _ in range(args.warmup):
        sh.solve_pass()
    if config5 and args.warmup:
        batch.digest(0, tile)  # (the digest's coefficient tables are built at its first use: not inside the timed region)
    batch.set_profiling(False)
    barrier()
    dev_ms = 0.0
    n_failed = 0
    digest_ms = 0.0
    # (the sampler forks rocm-smi every 50 ms: harmless beside the 190 ms steps of the gate kernel, whose VALU issue fraction needs the clock,
    # but it would disturb the sub-millisecond steps of the hash / Grumpkin workloads, and N ranks of it would disturb each other: it runs
    # where its value is used -- rank 0 of a one-rank run of the gate workloads)
    with ClockSampler(acvm_amd.current_device(), enabled=rank == 0 and world == 1 and args.workload in ("arith", "config5")) as clock:
        t0 = time.perf_counter()
        for i in range(args.steps):
            for k in range(n_tiles):
                # per-launch HIP events (two per launch) cost 3 % of a solve: they bracket the launches of the LAST tile of the LAST step only
                last = i == args.steps - 1 and k == n_tiles - 1
                if last:
                    batch.set_profiling(True)
                # the same step for every N and every tile count: ACVM::new of the tile (import of its resident inputs into the reused table),
                # then ACVM::solve
                sh.load_tile(k)
                # (the next tile's import rides behind this solve: acvm_batch_solve_then_import; config 5 reads the map's digest between tiles)
                n_failed += sh.solve_tile(k, pipelined=not config5 and not args.no_pipeline)
                if i == args.steps - 1:
                    dev_ms += batch.stats()["solve_device_ms"]
                if config5:  # what config 5 keeps of a tile: the per-instance digest of the map (the return witness rides along)
                    d0 = time.perf_counter()
                    batch.digest(0, tile)
                    digest_ms += (time.perf_counter() - d0) * 1e3
        acvm_amd.synchronize()
        my_elapsed = time.perf_counter() - t0
    barrier()
    elapsed = shard.max_over_ranks(my_elapsed, dist)
    st = batch.stats()
    batch.set_profiling(False)
    rank_rates = shard.gather_floats(n_rank * args.steps / my_elapsed, dist)

    if args.inner:  # the profiled one-tile run: nothing to report; the modmul probe runs so that the PMC passes see it
        acvm_amd.modmul_rate(100, 8)
        sh.free()
        return

    # ---- digest of digests: every rank hashes the witness maps of its shard (one more pass, untimed), rank 0 combines
    dod = None
    inst_digests = None
    if not args.no_digest:
        d0 = time.perf_counter()
        inst_digests = sh.digests()
        chunks = shard.chunk_digests(inst_digests)
        digest_s = time.perf_counter() - d0
        all_chunks = shard.gather_bytes(b"".join(chunks), dist)
        dod = {"value": shard.digest_of_digests(all_chunks), "chunk_instances": shard.DIGEST_CHUNK, "chunks": sum(len(c) for c in all_chunks) // 32,
               "per_rank": [shard.digest_of_digests([c]) for c in all_chunks], "pass_s_rank0": round(digest_s, 3),
               "definition": "Blake2s-256 over the chunk digests in global instance order; chunk digest = Blake2s-256 over the acvm_batch_digest "
                             "values of its instances: the same value for every number of ranks and every tile size"}

    # ---- end to end through the node-level driver, per rank on its own device (host buffers in, results out)
    e2e = None
    if not args.no_end_to_end:
        sh.free()  # the node driver allocates its own handle: give the table back first
        sh = None
        barrier()
        mine = end_to_end_node(gc, ids, values, n_rank, tile, n_dev)
        e2e_s = shard.max_over_ranks(mine["total_ms"] / 1e3, dist)
        e2e = dict(mine, value=total / e2e_s, total_ms=e2e_s * 1e3)

    # ---- everything below is rank 0's: the other ranks have delivered what the line needs from them (times, rates, chunk digests, their
    # end-to-end time) and leave; rank 0 computes the CPU baseline + parity and the PMC passes for EVERY world size, on its own device
    barrier()
    if rank != 0:
        if sh is not None:
            sh.free()
        if dist is not None:
            dist.destroy_process_group()
        return
    cpu = parity = None
    if not args.no_cpu_baseline:
        if sh is None:
            sh = tiling.ResidentShard(gc, ids, values, n_rank, tile)
            batch = sh.batch
        cpu, parity = cpu_baseline_and_parity(args, data, ids, values, batch, sh, tile, row, inst_digests)
        if not parity["bit_exact"]:
            print(json.dumps({"error": "parity check failed; the measurement is void", "parity": parity}), flush=True)
            raise SystemExit(2)
    if sh is not None:
        sh.free()

    value = total * args.steps / elapsed
    roof, alu = roofline_block(args, st, tile, world, clock.median())
    if not os.environ.get("ACVM_BENCH_NO_PMC"):
        try:
            roof["peak_measured_copy"] = acvm_amd.stream_rate(4 << 30)
            roof["peak_measured_copy_note"] = "GB/s moved by a two-rows-in, one-row-out stream (the gate kernel's access shape, 4 GiB per row) on this device in this run: what streaming code reaches of the 8 TB/s spec peak; `achieved` counts ALGORITHMIC bytes, the kernel's real traffic is `traffic`"
            if roof.get("traffic") and roof.get("kernel_avg_launch_ms"):
                roof["traffic_frac_of_measured_copy"] = roof["traffic"] / (roof["kernel_avg_launch_ms"] / 1e3) / 1e9 / roof["peak_measured_copy"]
        except acvm_amd.AcvmError:
            pass
    legs = None
    if world == 1 and args.workload == "arith" and not args.no_legs and not args.no_cpu_baseline:
        legs = {}
        # arith_pedersen is north_star's target shape: it runs at the metric's batch and tile (2^20 in tiles of 2^17; tiles of 2^16 measure 5 % lower:
        # the Pedersen launches of a tile of 2^17 hold enough waves for several records to share their inversions, NOTEBOOK.md section 9)
        # (the short legs first, before the long one has the part power-limited for a second and a half)
        # (a step of these is 0.15-2.5 ms: five of them from a cold start end before the device has reached its clocks and measure the ramp -- the hash step reads
        # 0.171 ms over 3 + 5 steps and 0.149 ms over 20 + 200 on the same box, NOTEBOOK.md 6.13 -- so the short legs time 30-120 ms of steps)
        for name, kw in (("hash", dict(warmup=20, steps=200)), ("grumpkin", dict(warmup=5, steps=50)), ("ecdsa", dict(warmup=5, steps=50)),
                         ("arith_pedersen", dict(total_log2=20, tile_log2=17, steps=5, warmup=1))):
            try:
                legs[name] = run_leg(name, **kw)
            except (acvm_amd.AcvmError, OSError, ValueError) as e:  # a leg must not void the metric's line
                legs[name] = {"error": str(e)[:300]}
        try:  # BASELINE config 5 at circuit size: 10^6 opcodes, tiles of 8 192 with slot reuse, 32-instance oracle audit, one PMC pass (~2 min of the run)
            legs["config5"] = run_config5_leg()
        except (acvm_amd.AcvmError, OSError, ValueError, MemoryError) as e:
            legs["config5"] = {"error": str(e)[:300]}
    line = {
        "metric": "witnesses solved/sec (whole node)",
        "value": value,
        "unit": "witnesses/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "u256 (8x u32 limbs, BN254-Fr Montgomery)",
        "data": "synthetic",
        "config": {"workload": f"{workload_name}, batch 2^{total.bit_length() - 1} witnesses over {world} GPU(s)", "opcodes": st["n_opcodes"],
                   "global_batch": total, "instances_per_gpu": n_rank, "tile_instances": tile, "tiles_per_gpu_per_step": n_tiles, "levels": st["n_levels"],
                   "step": "per tile: import of the resident inputs (ACVM::new) + solve, the same for every N"
                           + ("" if config5 or args.no_pipeline else "; the next tile's import is enqueued behind the solve (acvm_batch_solve_then_import)"),
                   "not_solved_rank0_all_steps": n_failed, "slow_path_instances_last_tile": st["n_slow_instances"],
                   "algorithmic_bytes_per_witness": st["algorithmic_bytes_per_instance"], "brillig_opcodes_inlined": st["n_brillig_inlined"],
                   "device_ms_per_step_rank0": dev_ms, "inputs_resident_setup_s_rank0": round(h2d_resident_s, 3), "host_cores": os.cpu_count(),
                   "parallelism": f"instances sharded x{world}, no collectives"},
        "per_rank_witnesses_per_s": rank_rates,
        "digest_of_digests": dod,
        "roofline": roof,
        "alu_roofline": alu,
        "end_to_end": e2e,
        "cpu_baseline": cpu,
        "parity": parity,
    }
    if config5:
        line["config"]["digest_ms_per_step_rank0"] = digest_ms / args.steps
    if legs is not None:
        line["other_workloads"] = legs
        # the same legs once more, compact, at the END of the line (a reader that keeps only the tail still sees them):
        # [witnesses/s, ms per step, HBM fraction of the dominant kernel, ALU fraction, parity ok]
        line["legs"] = {k: leg_summary(v) for k, v in legs.items()}
    rnd = lambda x, k: None if x is None else round(x, k)
    line["summary"] = {"value": round(value), "n_gpus": world, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "roofline_frac": rnd(roof.get("frac"), 4),
                       "kernel_avg_launch_ms": rnd(roof.get("kernel_avg_launch_ms"), 5), "traffic_bytes_per_launch": roof.get("traffic"),
                       "algorithmic_bytes_per_launch": roof.get("algorithmic_bytes_per_launch"), "valu_issue_frac": rnd(roof.get("valu_issue_frac"), 4),
                       "end_to_end": None if not e2e else round(e2e["value"]), "cpu_baseline": None if not cpu else [round(cpu["value"], 1), cpu["cores"]],
                       "parity_ok": None if not parity else parity["bit_exact"], "digest_of_digests": None if not dod else dod["value"][:16]}
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
