"""bench.py as the driver launches it: N ranks under torch.distributed.run, one shard each, no collective on the data path.
Two ranks share the one GPU of the test box (ACVM_BENCH_SHARE_GPU=1): their digest of digests must equal the single-rank
run's, i.e. the two shards ARE the two halves of the one batch (acvm/src/pwg/mod.rs:146,236-241: one solver per instance)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--gates", "400", "--total-log2", "13", "--tile-log2", "11", "--steps", "1", "--warmup", "0", "--no-end-to-end"]


def run(cmd, **env):
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", ACVM_BENCH_NO_PMC="1", **env)
    out = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=900, cwd=ROOT)
    return out


def last_json(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads(lines[-1])


def test_two_ranks_equal_one_batch():
    one = last_json(run([sys.executable, "bench.py", "--gpus", "1"] + ARGS))
    two = last_json(run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", "29547", "bench.py", "--gpus", "2"] + ARGS + ["--no-cpu-baseline"], ACVM_BENCH_SHARE_GPU="1"))
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["parity"]["bit_exact"] and one["parity"]["digests_checked"] > 0
    assert one["config"]["global_batch"] == two["config"]["global_batch"] == 1 << 13
    assert two["config"]["instances_per_gpu"] == 1 << 12 and len(two["per_rank_witnesses_per_s"]) == 2
    assert one["digest_of_digests"]["value"] == two["digest_of_digests"]["value"]
    assert len(two["digest_of_digests"]["per_rank"]) == 2 and two["digest_of_digests"]["per_rank"][0] != two["digest_of_digests"]["per_rank"][1]
    assert one["scaling"] == two["scaling"] == "strong"
    for line in (one, two):
        assert line["roofline"]["frac"] > 0 and line["unit"] == "witnesses/s"


def test_more_gpus_than_devices_is_refused():
    import acvm_amd
    n = acvm_amd.device_count() + 1
    out = run([sys.executable, "bench.py", "--gpus", str(n)] + ARGS)
    assert out.returncode != 0 and "refusing" in (out.stdout + out.stderr)
