"""Multi-GPU path on CPU: world_size-2 gloo processes exercise acvm_amd.shard (instance ranges, barrier, max / sum over
ranks) exactly as bench.py uses them; the data path itself has no collective."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_single_process():
    from acvm_amd import shard
    assert shard.shard_range(0, 1, 65536) == (0, 65536)
    assert [shard.shard_range(r, 8, 1 << 17) for r in (0, 7)] == [(0, 1 << 17), (7 << 17, 8 << 17)]
    assert shard.split_total(1 << 20, 8)[-1] == (7 << 17, 1 << 20)
    assert shard.max_over_ranks(3.5, None) == 3.5


def test_world_size_2_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "_dist_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


def test_digest_of_digests_is_independent_of_the_sharding():
    """what bench.py's ranks compute: chunk digests per shard, concatenated in rank order = the chunk digests of the whole batch"""
    import numpy as np
    from acvm_amd import shard
    rng = np.random.default_rng(7)
    d = rng.integers(0, 256, size=(4 * shard.DIGEST_CHUNK, 32), dtype=np.uint8)
    whole = shard.digest_of_digests(shard.chunk_digests(d))
    for world in (2, 4):
        per = d.shape[0] // world
        parts = [b"".join(shard.chunk_digests(d[r * per:(r + 1) * per])) for r in range(world)]
        assert shard.digest_of_digests(parts) == whole
    d2 = d.copy()
    d2[-1, 0] ^= 1
    assert shard.digest_of_digests(shard.chunk_digests(d2)) != whole


def test_node_lane_placement_reads_sysfs(tmp_path):
    """acvm_node_* pins a lane's two host threads to the CPUs next to its device (csrc/node.cpp: /sys/bus/pci/devices/<bus id>/numa_node and
    local_cpulist); the parsing runs here on a made-up tree, no GPU. A missing or malformed entry means "not pinned", never an error of the solve."""
    import acvm_amd
    assert acvm_amd.parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    assert acvm_amd.parse_cpulist("5") == [5] and acvm_amd.parse_cpulist("") == [] and acvm_amd.parse_cpulist("\n") == []
    assert acvm_amd.parse_cpulist("0-1,x") == [0, 1] and acvm_amd.parse_cpulist("7-3") == []   # malformed: the list ends where it stands
    dev = tmp_path / "0000:c1:00.0"
    dev.mkdir()
    (dev / "numa_node").write_text("1\n")
    (dev / "local_cpulist").write_text("32-63,96-127\n")
    node, cpus = acvm_amd.device_locality(str(tmp_path), "0000:C1:00.0")   # (hipDeviceGetPCIBusId prints the bus id in upper or lower case)
    assert node == 1 and cpus == list(range(32, 64)) + list(range(96, 128))
    other = tmp_path / "0000:05:00.0"
    other.mkdir()
    (other / "numa_node").write_text("-1\n")
    (other / "local_cpulist").write_text("\n")
    assert acvm_amd.device_locality(str(tmp_path), "0000:05:00.0") == (-1, [])
    with pytest.raises(acvm_amd.AcvmError):
        acvm_amd.device_locality(str(tmp_path), "0000:ff:00.0")
