"""CPU-only: the C-ABI libraries load and export every symbol the headers declare; the ctypes tables
cover them; error names map 1:1; creating a context without a GPU fails loudly (no CPU fallback)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zigzh?_[a-z0-9_]+)\s*\(", text)))


def exported(so):
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "zigz_amd", "lib", so)], text=True)
    return {l.split()[-1] for l in out.splitlines() if " T " in l}


def test_hip_abi_symbols():
    from zigz_amd import _ffi
    names = declared("zigz_hip.h")
    assert len(names) >= 40
    exp = exported("libzigz_hip.so")
    for n in names:
        assert n in exp, f"{n} declared in zigz_hip.h but not exported"
        assert n in _ffi.SIGNATURES, f"{n} has no ctypes signature"
    assert _ffi.lib.zigz_abi_version() == 1


def test_host_abi_symbols():
    from zigz_amd import host
    names = [n for n in declared("zigz_host.h") if n.startswith("zigzh_")]
    exp = exported("libzigz_host.so")
    for n in names:
        assert n in exp and n in host.SIGNATURES, n


def test_status_names_match_zig_errors():
    from zigz_amd import _ffi, errors
    expect = {1: "EmptyEvaluations", 2: "LengthNotPowerOfTwo", 3: "WrongNumberOfVariables", 4: "NoVariablesToFix",
              5: "NoVariables", 6: "ProtocolError", 7: "EmptyValues", 8: "TooManyValues", 9: "IndexOutOfBounds",
              10: "PointDimensionMismatch", 11: "NoQueries", 12: "TooManyQueries", 13: "MappingLengthMismatch",
              14: "InvalidMapping", 15: "QueryTableMismatch", 16: "EmptyTrace", 17: "OutOfMemory",
              18: "WrongNumberOfChallenges", 100: "NoDevice", 102: "NotCanonical"}
    for code, name in expect.items():
        assert _ffi.lib.zigz_status_name(code).decode() == name
    assert errors.LENGTH_NOT_POWER_OF_TWO == 2 and errors.NO_DEVICE == 100


def test_no_cpu_fallback_without_gpu():
    import zigz_amd
    if zigz_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(zigz_amd.ZigzError) as e:
        zigz_amd.Context(0)
    assert e.value.code == zigz_amd.errors.NO_DEVICE


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under zigz_amd/ or include/ may reference it."""
    for base in ("zigz_amd", "include"):
        for d, _, fs in os.walk(os.path.join(ROOT, base)):
            if os.sep + "build" in d or os.sep + "lib" in d or "__pycache__" in d:
                continue
            for f in fs:
                if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                    txt = open(os.path.join(d, f), errors="replace").read()
                    assert "oracle_lib" not in txt and "zigz_oracle" not in txt and "orc_" not in txt, os.path.join(d, f)
