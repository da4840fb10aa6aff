"""CPU only: the host mirror (VM, deserializer, verifier: everything that parses untrusted bytes) rebuilt with
AddressSanitizer + UndefinedBehaviorSanitizer and driven by tests/c_driver/host_fuzz.cpp over mutated golden proofs and
random programs.  (GPU AddressSanitizer is not available on this pool; the kernels' index arithmetic is covered by the
parity tests on ragged sizes instead.)"""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "zigz_amd", "csrc", "host")
LIB = os.path.join(ROOT, "zigz_amd", "lib")
G = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))


@pytest.fixture(scope="module")
def fuzz_exe(tmp_path_factory):
    import zigz_amd  # noqa: F401  (libzigz_hip.so must exist: the mirror links its host-side SHA3)
    exe = str(tmp_path_factory.mktemp("san") / "host_fuzz")
    srcs = [os.path.join(HOST, f) for f in ("capi.cpp", "core.cpp", "prover.cpp", "vm.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-pthread", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "zigz_amd", "csrc"), os.path.join(ROOT, "tests", "c_driver", "host_fuzz.cpp")]
                          + srcs + ["-o", exe, "-L", LIB, "-lzigz_hip", "-Wl,-rpath," + LIB])
    return exe


@pytest.mark.parametrize("i", [0, 3, len(G["prove"]) - 1])
def test_mutated_proofs_and_random_programs_under_asan_ubsan(fuzz_exe, tmp_path, i):
    g = G["prove"][i]
    (tmp_path / "proof.bin").write_bytes(bytes.fromhex(g["proof"]))
    (tmp_path / "prog.bin").write_bytes(bytes.fromhex(g["program"]))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([fuzz_exe, str(tmp_path / "proof.bin"), str(tmp_path / "prog.bin"), "4000"], capture_output=True,
                       text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-3000:] + r.stderr[-6000:])
    assert r.stdout.startswith("ok:"), r.stdout
