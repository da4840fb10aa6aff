"""The drop-in boundary consumed from plain C99 (tests/c_driver/abi_driver.c): compiled with gcc against
include/zigz_hip.h and linked to libzigz_hip.so -- no Python between the caller and the ABI.
CPU: the header is valid C99 / C++17, the host-only entry points answer their known values, and a context cannot be
created without a GPU.  GPU: the reference's own small known answers through the HIP path."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
LIB = os.path.join(ROOT, "zigz_amd", "lib")
SRC = os.path.join(ROOT, "tests", "c_driver", "abi_driver.c")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    import zigz_amd  # noqa: F401  (makes sure the libraries are built)
    exe = str(tmp_path_factory.mktemp("c_driver") / "abi_driver")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O1", "-I", INC, SRC, "-o", exe,
                           "-L", LIB, "-lzigz_hip", "-Wl,-rpath," + LIB])
    return exe


@pytest.mark.parametrize("header", ["zigz_hip.h", "zigz_host.h"])
@pytest.mark.parametrize("lang", [("gcc", "c", "-std=c99"), ("g++", "c++", "-std=c++17")])
def test_headers_are_valid_c_and_cxx(header, lang):
    cc, x, std = lang
    subprocess.check_call([cc, std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x", x,
                           os.path.join(INC, header)])


def test_c_driver_host_entry_points(driver):
    r = subprocess.run([driver, "host"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host: 0 failure(s)" in r.stdout


@pytest.mark.gpu
def test_c_driver_gpu_known_answers(driver):
    r = subprocess.run([driver, "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gpu: 0 failure(s)" in r.stdout


def test_device_keccak_source_on_host(tmp_path):
    """zigz_amd/csrc/keccak.hpp (the bit-interleaved permutation, leaf / node hashing, tree form <-> SHA3 bytes) built
    for the host with its C fallbacks: equals the 64-bit formulation on random states and hashlib on leaves / nodes."""
    import hashlib
    exe = str(tmp_path / "keccak_forms")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "zigz_amd", "csrc"),
                           os.path.join(ROOT, "tests", "c_driver", "keccak_forms.cpp"), "-o", exe])
    vals = [0, 1, 2, 5, 2013265920, 1234567, 0xFFFFFFFF, 2 ** 31, 2 ** 63 + 12345, 2 ** 64 - 1]
    out = subprocess.run([exe] + [str(v) for v in vals], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    assert lines[0] == "permutation ok"
    leaves = [hashlib.sha3_256(v.to_bytes(8, "little")).digest() for v in vals]
    got_leaves = [l.split()[2] for l in lines if l.startswith("leaf ")]
    assert got_leaves == [d.hex() for d in leaves]
    got_nodes = [l.split()[1] for l in lines if l.startswith("node ")]
    want_nodes = [hashlib.sha3_256(leaves[k - 1] + leaves[k]).digest().hex() for k in range(1, len(vals), 2)]
    assert got_nodes == want_nodes
