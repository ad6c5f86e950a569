"""The C-ABI library loads and exports every symbol include/acvm_amd.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "acvm_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(acvm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import acvm_amd
    from acvm_amd import build
    build.build()
    lib = ctypes.CDLL(acvm_amd.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/acvm_amd.h but not exported"
    assert sorted(acvm_amd.ABI_SYMBOLS) == declared


def test_abi_version_and_error_string():
    import acvm_amd
    L = acvm_amd.lib()
    assert L.acvm_abi_version() == 6
    assert acvm_amd.Circuit  # python mirror present


def test_no_device_fails_loudly():
    """Without a gfx950 device batch creation must raise; there is no CPU fallback."""
    import acvm_amd
    from acvm_amd import synth
    if acvm_amd.device_count() > 0:
        import pytest
        pytest.skip("a GPU is present")
    circ, ids = synth.arithmetic_circuit(10, seed=1)
    c = acvm_amd.Circuit(circ.to_bytes())
    import pytest
    with pytest.raises(acvm_amd.AcvmError):
        acvm_amd.Batch(c, 4, ids)


def test_product_does_not_reference_oracle():
    """The shipped package must never import, link or call anything under oracle/."""
    import subprocess
    pkg = os.path.join(ROOT, "acvm_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath.split(os.sep)[-1:]:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".inc", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
                code = "\n".join(l for l in txt.splitlines() if not l.lstrip().startswith(("//", "#", "*", "/*")) or l.lstrip().startswith("#include"))
                assert "oracle/" not in code and "oracle\\" not in code, f
    out = subprocess.run(["ldd", os.path.join(pkg, "libacvm_amd.so")], capture_output=True, text=True).stdout
    assert "liboracle" not in out


def test_device_field_library_on_host():
    """fr_device.hpp compiled for the host against the planner's independent implementation (tools/fr_device_host_test.hip)."""
    import subprocess
    exe = "/tmp/acvm_fr_device_host_test"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17",
                           os.path.join(ROOT, "tools", "fr_device_host_test.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout
