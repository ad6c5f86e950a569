#!/usr/bin/env python3
"""bench.py -- witnesses solved / second on MI355X for the batched ACIR witness solver.

Workload (BASELINE.json configs[1]): 10k-gate arithmetic-only ACIR, batch 2^16 instances per GPU,
synthetic circuit and inputs from acvm_amd.synth. A "step" = ACVM::solve() of the whole
per-GPU batch, inputs already resident in HBM (Montgomery SoA), witness map left in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One process per GPU, instances sharded contiguously, no data-path collective (weak scaling: the per-GPU
batch is fixed). Rank 0 prints ONE JSON line. The roofline entry is for arith_level_kernel:
achieved = algorithmic bytes of all its launches in a solve / their summed HIP-event durations.
The cpu_baseline entry times the CPU oracle (a port of the reference's in-order solver) on a bounded
sample of the same workload; the same sample is the bit-exact parity check of the run.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gates", type=int, default=10000)
    ap.add_argument("--batch-log2", type=int, default=16, help="instances per GPU = 2^this")
    ap.add_argument("--cpu-sample", type=int, default=0, help="instances for the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        # timing barrier / max only; the data path has no exchange step, so a CPU (gloo) group is enough and keeps
        # torch's bundled HIP runtime out of this process (the kernels run on the system ROCm runtime of libacvm_amd.so)
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    import numpy as np
    import acvm_amd
    from acvm_amd import synth

    if acvm_amd.device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible; this benchmark has no CPU fallback")
    acvm_amd.set_device(local_rank % max(acvm_amd.device_count(), 1))

    B = 1 << args.batch_log2
    seed = 0xAC1D0002
    circ, ids = synth.arithmetic_circuit(args.gates, seed=seed)
    data = circ.to_bytes()
    values = synth.witness_batch(B, seed=seed, first_instance=rank * B)
    gc = acvm_amd.Circuit(data)
    batch = acvm_amd.Batch(gc, B, ids)
    batch.set_initial_witness(values)  # H2D + Montgomery import: outside the timed region
    batch.set_profiling(True)

    def barrier():
        if dist is not None:
            dist.barrier()
        acvm_amd.synchronize()

    for _ in range(args.warmup):
        batch.reset()
        batch.solve()
    barrier()
    t0 = time.perf_counter()
    arith_ms = 0.0
    dyn_ms = 0.0
    dev_ms = 0.0
    for _ in range(args.steps):
        batch.reset()
        batch.solve()
        st = batch.stats()
        arith_ms += st["arith_kernel_ms"]
        dyn_ms += st["dyn_kernel_ms"]
        dev_ms += st["solve_device_ms"]
    acvm_amd.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

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
        sample = args.cpu_sample or min(B, max(64, 8 * threads))
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
               "sample": f"{sample} instances of the same {args.gates}-gate circuit, oracle/liboracle.so, {threads} threads, {cpu_s:.2f} s"}
        if not ok:
            print(json.dumps({"error": "parity check failed; the measurement is void", "parity": parity}), flush=True)
            raise SystemExit(2)

    if rank == 0:
        total_instances = B * world * args.steps
        value = total_instances / elapsed
        alg_bytes_per_solve = st["arith_algorithmic_bytes_per_instance"] * B
        achieved = alg_bytes_per_solve * args.steps / (arith_ms / 1e3) / 1e9 if arith_ms > 0 else 0.0
        line = {
            "metric": "witnesses solved/sec (whole node)",
            "value": value,
            "unit": "witnesses/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u256 (8x u32 limbs, BN254-Fr Montgomery)",
            "data": "synthetic",
            "config": {"workload": f"{args.gates}-gate arithmetic-only ACIR, batch 2^{args.batch_log2} witnesses per GPU",
                       "gates": args.gates, "instances_per_gpu": B, "levels": st["n_levels"],
                       "solved_instances_rank0": n_solved, "slow_path_instances_rank0": st["n_slow_instances"],
                       "algorithmic_bytes_per_witness": st["algorithmic_bytes_per_instance"],
                       "device_ms_per_step": dev_ms / args.steps, "parallelism": f"instances sharded x{world}, no collectives"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "arith_level_kernel",
                         "launches_per_step": st["n_kernel_launches"], "kernel_ms_per_step": arith_ms / args.steps,
                         "other_kernels": {"arith_dyn_level_kernel_ms_per_step": dyn_ms / args.steps,
                                           "note": "batched-inversion gates, overlapped on a second stream"}},
            "cpu_baseline": cpu,
            "parity": parity,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
