import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run on the GPU box only")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.build()
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(autouse=True, scope="session")
def every_batch_has_a_hazard_free_schedule():
    """Every batch handle a test creates -- i.e. every circuit shape, option set and planner mode the suite runs on the device -- has the schedule it is
    about to enqueue proved free of cross-stream hazards by the host-only checker (acvm_circuit_check_schedule) first. A test that passes bit for bit on
    this box's timing but would race on another box's fails here. (Handles created inside the library -- the node driver's lanes -- share the plans
    checked for the batch handles of the same circuit; tests/test_schedule_hazards.py covers the corpus without a GPU.)"""
    import acvm_amd
    orig = acvm_amd.Batch.__init__
    seen = set()

    def init(self, circuit, n_instances, initial_ids, solver=None, fold_digest=False, reuse_slots=False, keep=()):
        orig(self, circuit, n_instances, initial_ids, solver, fold_digest, reuse_slots, keep)
        ids, keep = list(initial_ids), list(keep)
        mode = tuple(acvm_amd.tuning_get(k) for k in acvm_amd.tuning_keys())
        key = (id(circuit), tuple(ids), bool(fold_digest), bool(reuse_slots), tuple(keep), solver is not None, max(n_instances, 1).bit_length(), mode)
        if key in seen:
            return
        seen.add(key)
        r = circuit.check_schedule(ids, n_instances=n_instances, fold_digest=fold_digest, reuse_slots=reuse_slots, keep=keep, host_solver=solver is not None)
        assert r["ok"], "the level schedule of this handle has a hazard:\n" + r["report"]

    acvm_amd.Batch.__init__ = init
    yield
    acvm_amd.Batch.__init__ = orig
