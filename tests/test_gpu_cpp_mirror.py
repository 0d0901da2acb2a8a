"""-m gpu: builds and runs the C++ host-mirror test (tests/cpp/test_host_mirror.cpp) against libbp_msm_ntt.so"""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror(tmp_path):
    exe = str(tmp_path / "test_host_mirror")
    libdir = os.path.join(ROOT, "baby_plonk_rust_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp"), "-o", exe,
                           "-L" + libdir, "-lbp_msm_ntt", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host mirror ok" in out.stdout
    # Prover mirror: the toy circuit of tests/verify_proof_test.rs built in C++, blinders of the committed golden proof
    import hashlib
    import random
    from tests import bigint_model as M
    blinders = [random.Random(99).randrange(1, M.Q) for _ in range(11)]
    out = subprocess.run([exe, "".join(b.to_bytes(32, "little").hex() for b in blinders)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    proof = bytes.fromhex(out.stdout.split("proof ")[1].split()[0])
    golden = open(os.path.join(ROOT, "tests", "golden", "toy_proof_blinders_seed99.sha256")).read().strip()
    assert len(proof) == 624 and hashlib.sha256(proof).hexdigest() == golden
