#!/usr/bin/env python3
"""bench.py -- witnesses solved / second on MI355X for the batched ACIR witness solver.

Default workload = BASELINE.json configs[1]: 10k-gate arithmetic-only ACIR, batch 2^16 instances per GPU, synthetic
circuit and inputs from acvm_amd.synth. A "step" = ACVM::solve() of the whole per-GPU batch, inputs already
resident in HBM (Montgomery SoA), witness map left in HBM. Other workloads (parity-test configs, measured for DESIGN.md):
    --workload hash            config 3: SHA256 + Keccak256 + RANGE circuit
    --workload grumpkin        config 4: Pedersen + FixedBaseScalarMul + SchnorrVerify circuit
    --workload arith_pedersen  the north-star shape: 10k arithmetic gates + 8 Pedersen commitments
    --workload mixed           the config-5 opcode mix at --gates opcodes (every kernel class in one circuit)

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One process per GPU, instances sharded contiguously (acvm_amd.shard), no data-path collective (weak scaling: the per-GPU
batch is fixed). Rank 0 prints ONE JSON line. The roofline entry is for the workload's dominant kernel: achieved =
algorithmic bytes of all its launches in a solve / their summed HIP-event durations (events on the batch's own streams,
inside the timed region). The cpu_baseline entry times the CPU oracle (a port of the reference's in-order solver) on a
bounded sample of the same workload; the same sample is the bit-exact parity check of the run.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CLASS_KERNEL = ["light_level_kernel", "hash_level_kernel", "grumpkin_level_kernel", "brillig_level_kernel"]


def make_workload(args, rank, B):
    from acvm_amd import synth
    first = rank * B
    if args.workload == "arith":
        circ, ids = synth.arithmetic_circuit(args.gates, seed=0xAC1D0002)
        values = synth.witness_batch(B, seed=0xAC1D0002, first_instance=first)
        name = f"{args.gates}-gate arithmetic-only ACIR, batch 2^{args.batch_log2} witnesses per GPU"
    elif args.workload == "hash":
        circ, ids = synth.hash_circuit()
        values = synth.byte_batch(B, len(ids), first_instance=first)
        name = f"sha256 + keccak256 (64-byte messages) + 96 RANGE(8) ACIR, batch 2^{args.batch_log2} per GPU"
    elif args.workload == "grumpkin":
        circ, ids = synth.grumpkin_circuit()
        # the edge-case / flipped-signature pattern repeats every 1024 instances (row generation is host Python)
        import numpy as np
        base = synth.grumpkin_rows(min(B, 1024), first_instance=0)
        arr = np.frombuffer(synth.values_from_rows(base), dtype=np.uint8).reshape(len(base), -1)
        values = arr[(first + np.arange(B)) % len(base)].tobytes()
        name = f"Pedersen + FixedBaseScalarMul + SchnorrVerify ACIR, batch 2^{args.batch_log2} per GPU"
    elif args.workload == "arith_pedersen":
        circ, ids = synth.arith_pedersen_circuit(args.gates, args.pedersen)
        values = synth.witness_batch(B, seed=0xAC1D0006, first_instance=first)
        name = f"{args.gates}-gate arithmetic + {args.pedersen} Pedersen ACIR, batch 2^{args.batch_log2} per GPU"
    elif args.workload == "mixed":
        circ, ids = synth.mixed_circuit(args.gates)
        values = synth.witness_batch(B, seed=0xAC1D0005, first_instance=first)
        name = f"{args.gates}-opcode mixed ACIR (config-5 mix: arithmetic, range/logic, directives, memory, Brillig, hashes, Pedersen), batch 2^{args.batch_log2} per GPU"
    else:
        raise SystemExit(f"unknown workload {args.workload}")
    return circ, ids, values, name


def load_traffic(workload, kernel):
    """Measured HBM bytes per launch of `kernel` from the committed PMC profile of this workload (profiles/traffic.json,
    written by tools/prof_summary.py --json from separate --pmc FETCH_SIZE / WRITE_SIZE passes), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        e = t.get(workload, {}).get(kernel)
        return e
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)  # the device reaches its clocks after about two solves (profiles/README.md)
    ap.add_argument("--workload", default="arith", choices=["arith", "hash", "grumpkin", "arith_pedersen", "mixed"])
    ap.add_argument("--gates", type=int, default=10000)
    ap.add_argument("--pedersen", type=int, default=8)
    ap.add_argument("--batch-log2", type=int, default=16, help="instances per GPU = 2^this")
    ap.add_argument("--cpu-sample", type=int, default=0, help="instances for the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import acvm_amd
    from acvm_amd import shard

    rank, local_rank, world = shard.env_rank()
    # timing barrier / max only; the data path has no exchange step, so a CPU (gloo) group is enough and keeps
    # torch's bundled HIP runtime out of this process (the kernels run on the system ROCm runtime of libacvm_amd.so)
    dist = shard.init_group(rank, world)

    if acvm_amd.device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible; this benchmark has no CPU fallback")
    acvm_amd.set_device(local_rank % max(acvm_amd.device_count(), 1))

    B = 1 << args.batch_log2
    circ, ids, values, workload_name = make_workload(args, rank, B)
    data = circ.to_bytes()
    gc = acvm_amd.Circuit(data)
    batch = acvm_amd.Batch(gc, B, ids)
    batch.set_initial_witness(values)  # H2D + Montgomery import: outside the timed region
    batch.set_profiling(True)

    def barrier():
        shard.barrier(dist)
        acvm_amd.synchronize()

    for _ in range(args.warmup):  # with per-launch events, so that the event pool exists before the timed region
        batch.reset()
        batch.solve()
    barrier()
    t0 = time.perf_counter()
    arith_ms = dyn_ms = dev_ms = 0.0
    cls_ms = [0.0] * 4
    for i in range(args.steps):
        # per-launch HIP events (two per launch) cost 3 % of a solve: they bracket every launch of the LAST timed step only
        batch.set_profiling(i == args.steps - 1)
        batch.reset()
        batch.solve()
        st = batch.stats()
        dev_ms += st["solve_device_ms"]
        if i == args.steps - 1:
            arith_ms = st["arith_kernel_ms"]
            dyn_ms = st["dyn_kernel_ms"]
            cls_ms = list(st["class_kernel_ms"])
    acvm_amd.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    elapsed = shard.max_over_ranks(elapsed, dist)

    st = batch.stats()
    results = batch.results()
    n_solved = sum(1 for r in results if r.status == 0)

    # ---- CPU baseline + parity on a bounded sample (rank 0 only)
    cpu = None
    parity = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import binding as ob
        cores = os.cpu_count() or 1
        threads = min(cores, 64)
        per = {"arith": 24, "hash": 1024, "grumpkin": 256, "arith_pedersen": 16, "mixed": 16}[args.workload]  # a few seconds of all host cores
        sample = args.cpu_sample or min(B, max(64, per * threads))
        oc = ob.Circuit(data)
        sample_vals = values[: sample * len(ids) * 32]
        c0 = time.perf_counter()
        ores, oasg, ovals = ob.solve_batch(oc, ids, sample_vals, sample, want_witness=True, n_threads=threads)
        cpu_s = time.perf_counter() - c0
        gasg, gvals = batch.witness_map(0, sample)
        ok = all(results[j].as_tuple() == ores[j].as_tuple() for j in range(sample))
        ok = ok and bool(np.array_equal(gasg, oasg[:, : gasg.shape[1]])) and bool(np.array_equal(gvals, ovals[:, : gvals.shape[1]]))
        parity = {"checked_instances": sample, "bit_exact": ok}
        cpu = {"value": sample / cpu_s, "unit": "witnesses/s", "cores": threads, "kind": "port",
               "sample": f"{sample} instances of the same circuit, oracle/liboracle.so, {threads} threads, {cpu_s:.2f} s"}
        # SURVEY 8d asks for one core as well: a few seconds of the same oracle on one thread
        one = max(1, min(sample, int(round(3.0 * sample / (cpu_s * threads)))))
        c1 = time.perf_counter()
        ob.solve_batch(oc, ids, values[: one * len(ids) * 32], one, want_witness=False, n_threads=1)
        one_s = time.perf_counter() - c1
        cpu["single_thread"] = {"value": one / one_s, "unit": "witnesses/s", "sample": f"{one} instances, 1 thread, {one_s:.2f} s"}
        if not ok:
            print(json.dumps({"error": "parity check failed; the measurement is void", "parity": parity}), flush=True)
            raise SystemExit(2)

    if rank == 0:
        total_instances = B * world * args.steps
        value = total_instances / elapsed
        # dominant kernel of the workload: the arithmetic level kernel, or the record class that took the most time
        cand = {"arith_level_kernel": (arith_ms, st["arith_algorithmic_bytes_per_instance"]),
                "inverse_batch_kernel": (dyn_ms, st["dyn_algorithmic_bytes_per_instance"])}
        for k in range(4):
            cand[CLASS_KERNEL[k]] = (cls_ms[k], st["class_algorithmic_bytes_per_instance"][k])
        dominant = "arith_level_kernel" if args.workload == "arith" else max(cand, key=lambda k: cand[k][0])
        k_ms, k_bytes = cand[dominant]
        achieved = k_bytes * B / (k_ms / 1e3) / 1e9 if k_ms > 0 else 0.0
        # the committed PMC profile is of the default size of the workload: quote it only for that size
        default_size = args.gates == 10000 and args.batch_log2 == 16 and args.pedersen == 8
        tr = load_traffic(args.workload, dominant) if default_size else None
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": tr["bytes_per_launch"] if tr else None, "kernel": dominant,
                "kernel_ms_per_step": k_ms, "algorithmic_bytes_per_step": k_bytes * B,
                "kernel_timing": f"HIP events around every launch of timed step {args.steps} of {args.steps} (on the launching stream)",
                "launches_per_step_all_kernels": st["n_kernel_launches"],
                "other_kernels_ms_per_step": {k: v[0] for k, v in cand.items() if k != dominant and v[0] > 0}}
        if tr:
            roof["traffic_source"] = tr.get("source")
            # the profile ran 3 timed + 1 warm-up solves: launches per solve = launches_profiled / 4
            per_solve = max(1, tr.get("launches_profiled", 4) // 4)
            roof["traffic_algorithmic_bytes_per_launch"] = k_bytes * B / per_solve
            roof["kernel_launches_per_step"] = per_solve
            roof["kernel_avg_launch_ms"] = k_ms / per_solve
        if dominant == "grumpkin_level_kernel":
            roof["note"] = "integer-ALU bound (about 1e3 field multiplications per 128-256 B moved): the HBM fraction is for information"
        line = {
            "metric": "witnesses solved/sec (whole node)",
            "value": value,
            "unit": "witnesses/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u256 (8x u32 limbs, BN254-Fr Montgomery)",
            "data": "synthetic",
            "config": {"workload": workload_name, "opcodes": st["n_opcodes"], "instances_per_gpu": B, "levels": st["n_levels"],
                       "solved_instances_rank0": n_solved, "slow_path_instances_rank0": st["n_slow_instances"],
                       "algorithmic_bytes_per_witness": st["algorithmic_bytes_per_instance"],
                       "device_ms_per_step": dev_ms / args.steps, "parallelism": f"instances sharded x{world}, no collectives"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "parity": parity,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
