"""Builds the in-tree native libraries for gfx950 (hipcc cross-compiles without a GPU):
    zigz_amd/lib/libzigz_hip.so   -- HIP kernels + C ABI (include/zigz_hip.h)
    zigz_amd/lib/libzigz_host.so  -- C++ mirror of the Zig host (Prover/Verifier/Serializer/VM), on top of the C ABI
Usage: python -m zigz_amd.build [--force]
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
OBJ = os.path.join(HERE, "build")
INC = os.path.join(ROOT, "include")
ARCH = "gfx950"


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP backend cannot be built")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _headers():
    hs = [os.path.join(INC, f) for f in os.listdir(INC) if f.endswith(".h")]
    for d, _, fs in os.walk(CSRC):
        hs += [os.path.join(d, f) for f in fs if f.endswith(".hpp")]
    return hs


def _host_cxx():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/bin/amdclang++"):
        if os.path.exists(c):
            return c
    return "g++"


def build_hip(force=False):
    os.makedirs(LIB, exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = _headers()
    objs = []
    extra = os.environ.get("ZIGZ_EXTRA_HIPCC_FLAGS", "").split()  # A/B builds of kernel variants
    units = [
        ("kernels.hip", [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", f"-I{INC}", f"-I{CSRC}"] + extra),
        ("merkle_levels.hip", [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", f"-I{INC}", f"-I{CSRC}"] + extra),
        ("api.cpp", [hipcc, "-O3", "-std=c++17", "-fPIC", f"-I{INC}", f"-I{CSRC}"]),
        ("api_mle.cpp", [hipcc, "-O3", "-std=c++17", "-fPIC", f"-I{INC}", f"-I{CSRC}"]),
        ("api_commit.cpp", [hipcc, "-O3", "-std=c++17", "-fPIC", f"-I{INC}", f"-I{CSRC}"]),
        ("api_misc.cpp", [hipcc, "-O3", "-std=c++17", "-fPIC", f"-I{INC}", f"-I{CSRC}"]),
        # the host sponge is the sequential critical path of a proof: ROCm's clang schedules the scalar / BMI2 Keccak-f
        # 7 % faster than g++ on the EPYC 9575F of the GPU box (tools/host_keccak_rate.cpp: 0.178 vs 0.192 us)
        ("host_hash.cpp", [_host_cxx(), "-O3", "-std=c++17", "-fPIC", f"-I{CSRC}"]),
        ("host_keccak_avx512.cpp", [_host_cxx(), "-O3", "-std=c++17", "-fPIC", f"-I{CSRC}"]),
        ("host_keccak_bmi.cpp", [_host_cxx(), "-O3", "-std=c++17", "-fPIC", f"-I{CSRC}"]),
        ("host_keccak_avx512vl.cpp", [_host_cxx(), "-O3", "-std=c++17", "-fPIC", f"-I{CSRC}"]),
        ("host_sponge_batch.cpp", [_host_cxx(), "-O3", "-std=c++17", "-fPIC", "-pthread", f"-I{CSRC}"]),
        ("shm_comm.cpp", [_host_cxx(), "-O2", "-std=c++17", "-fPIC", f"-I{INC}", f"-I{CSRC}"]),
        # RCCL transport: compiled against rccl.h, librccl.so itself is dlopen-ed on first use (570 MB: never at load time)
        ("rccl_comm.cpp", [hipcc, "-O2", "-std=c++17", "-fPIC", f"-I{INC}", f"-I{CSRC}", "-I/opt/rocm/include"]),
    ]
    for src, cmd in units:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".", "_") + ".o")
        if force or _newer(o, [s] + hdrs):
            _run(cmd + ["-c", s, "-o", o])
        objs.append(o)
    so = os.path.join(LIB, "libzigz_hip.so")
    if force or _newer(so, objs):
        _run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", so] + objs + ["-lrt", "-pthread", "-ldl"])
    return so


def build_host(force=False):
    hdir = os.path.join(CSRC, "host")
    if not os.path.isdir(hdir):
        return None
    srcs = sorted(os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith(".cpp"))
    if not srcs:
        return None
    so = os.path.join(LIB, "libzigz_host.so")
    if force or _newer(so, srcs + _headers() + [os.path.join(LIB, "libzigz_hip.so")]):
        _run(["g++", "-O3", "-std=c++17", "-fPIC", "-pthread", "-shared", f"-I{INC}", f"-I{CSRC}", "-o", so] + srcs +
             [f"-L{LIB}", "-lzigz_hip", "-Wl,-rpath,$ORIGIN"])
    return so


def build_all(force=False):
    a = build_hip(force)
    b = build_host(force)
    return a, b


if __name__ == "__main__":
    print(build_all("--force" in sys.argv))
