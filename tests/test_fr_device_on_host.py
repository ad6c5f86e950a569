"""The device field library (acvm_amd/csrc/fr_device.hpp is __host__ __device__) executed on the host against the
planner's independent 4x64-bit implementation (fr_host.hpp): products, the 29-bit working form with lazy reduction at the
edges of its contracts, add/sub/neg, both inversions, and the 5^-1 known answer of acvm_js/test/shared/foreign_call.ts.
No GPU is needed: hipcc builds the host side of tools/fr_device_host_test.hip and nothing is launched."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_device_field_library_on_host(tmp_path):
    exe = str(tmp_path / "fr_host_test")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tools", "fr_device_host_test.hip"), "-o", exe],
                   check=True, timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-2000:]
