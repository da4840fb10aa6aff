#!/usr/bin/env python3
"""Per-kernel static instruction mix of the gfx950 code of zigz_amd/csrc/kernels.hip and merkle_levels.hip.

    python tools/isa_counts.py [--write]      # --write refreshes profiles/isa_counts.json

Compiles the kernels to assembly with the flags of zigz_amd/build.py (hipcc cross-compiles without a GPU) and
counts, for every kernel, the instructions of each class in the function body.  bench.py converts Keccak
permutations/s into VALU lane-instructions/s with the per-hash figure of the loop body of k_keccak_leaves /
k_keccak_level (the permutation is fully unrolled, so the static count of the loop body IS the dynamic count per
hash); tests/test_abi.py re-derives the numbers and compares them with the committed file.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zigz_amd", "csrc")
OUT = os.path.join(ROOT, "profiles", "isa_counts.json")


SOURCES = ("kernels.hip", "merkle_levels.hip")


def assembly(extra=()):
    hipcc = "/opt/rocm/bin/hipcc"
    out = []
    with tempfile.TemporaryDirectory() as d:
        for src in SOURCES:
            s = os.path.join(d, src + ".s")
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{CSRC}",
                                   "--cuda-device-only", "-S", "-o", s, os.path.join(CSRC, src), *extra,
                                   *os.environ.get("ZIGZ_EXTRA_HIPCC_FLAGS", "").split()],
                                  stderr=subprocess.DEVNULL)
            out.append(open(s).read())
    return "\n".join(out)


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.split("\n")


def classify(op):
    if op.startswith("v_bitop3"):
        return "v_bitop3"
    if op.startswith("v_alignbit"):
        return "v_alignbit"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("s_sleep"):
        return "s_sleep"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def count(asm):
    kernels = {}
    cur = None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur = m.group(1)
            kernels[cur] = {}
            continue
        if cur is None:
            continue
        t = line.strip()
        if t.startswith(".Lfunc_end"):
            cur = None
            continue
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        c = classify(t.split()[0])
        kernels[cur][c] = kernels[cur].get(c, 0) + 1
        vg = None
    # VGPR counts from the kernel descriptors
    for m in re.finditer(r"\.amdhsa_kernel (\w+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        nm = m.group(1)
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(2))
        if nm in kernels and vg:
            kernels[nm]["vgprs"] = int(vg.group(1))
    names = list(kernels)
    pretty = demangle(names)
    out = {}
    for n, p in zip(names, pretty):
        k = kernels[n]
        k["valu"] = k.get("v_bitop3", 0) + k.get("v_alignbit", 0) + k.get("valu_other", 0)
        short = re.sub(r"\(.*$", "", p).replace("zk::", "")
        out[short.replace("void ", "")] = k
    return out


def keccak_per_hash(counts):
    """VALU instructions per hash of the two hashing kernels.  The hash loop is `#pragma unroll 1` around one fully
    unrolled permutation, so the function body holds exactly one permutation plus set-up: body VALU count = per-hash
    count (set-up instructions outside the loop are < 1 % and are included, i.e. the figure is an upper bound on the
    permutation itself and the right number for instructions issued per hash at HPT = 1; at HPT = 4 the per-hash count
    is lower by the amortised set-up)."""
    leaves = counts["k_keccak_leaves"]
    level = counts["k_keccak_level<4>"]
    return {"k_keccak_leaves": leaves, "k_keccak_level<4>": level}


def main():
    counts = count(assembly())
    sel = {k: v for k, v in counts.items() if "keccak" in k or "bind" in k or "sums" in k or "radix" in k or "witness" in k
           or "lasso" in k or "k_level_hash" in k or "k_merkle_top" in k or "k_runs_" in k or "k_cons_" in k}
    doc = {"source": "zigz_amd/csrc/kernels.hip + merkle_levels.hip, hipcc --offload-arch=gfx950 -O3 (tools/isa_counts.py)",
           "kernels": sel}
    print(json.dumps(doc, indent=1, sort_keys=True))
    if "--write" in sys.argv:
        with open(OUT, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
