#!/usr/bin/env python3
"""Extract the known-answer DATA (program bytes, run limits, expected register values) that the
reference's own VM tests hold, into tests/golden/vm_kats.json.

Runs only in the build container (reads /root/reference as text; nothing from the reference is
executed or copied as source).  Sources: tests/test_rv64i.zig, tests/test_rv64m.zig and the inline
tests at the bottom of src/vm/state.zig.  The committed JSON is the fixture; this script is kept so
the extraction is reproducible.
"""
import json, re, sys, os

REF = "/root/reference"
FILES = ["tests/test_rv64i.zig", "tests/test_rv64m.zig", "src/vm/state.zig"]

def parse_int(tok):
    tok = tok.replace("_", "")
    return int(tok, 0)

def main():
    kats = []
    for rel in FILES:
        text = open(os.path.join(REF, rel)).read()
        # split into test blocks
        for m in re.finditer(r'test "([^"]+)" \{(.*?)\n\}\n', text, re.S):
            name, body = m.group(1), m.group(2)
            pm = re.search(r'const program = \[_\]u8\{(.*?)\};', body, re.S)
            im = re.search(r'VMState\.init\([^,]+,\s*&program,\s*(0x[0-9A-Fa-f]+|\d+),\s*null\)', body)
            if not pm or not im:
                continue
            code = re.sub(r'//[^\n]*', '', pm.group(1))
            prog = [parse_int(t) for t in re.findall(r'0x[0-9A-Fa-f]+|\b\d+\b', code)]
            rm = re.search(r'vm\.run\((\d+)\)', body)
            steps = re.findall(r'vm\.step\(\)', body)
            expected = {}
            for em in re.finditer(r'expectEqual\(@as\(u64,\s*([^)]+)\),\s*vm\.regs\.read\((\d+)\)\)', body):
                expected[int(em.group(2))] = parse_int(em.group(1).strip())
            em = re.search(r'const expected: u64 = @bitCast\(@as\(i64, (-?\d+)\)\);', body)
            if em:
                rm2 = re.search(r'expectEqual\(expected, vm\.regs\.read\((\d+)\)\)', body)
                if rm2:
                    expected[int(rm2.group(1))] = int(em.group(1)) & (2**64 - 1)
            pcm = re.search(r'expectEqual\(@as\(u64,\s*(0x[0-9A-Fa-f]+|\d+)\),\s*vm\.pc\)', body)
            kats.append({
                "source": rel, "name": name, "program": prog, "entry_pc": parse_int(im.group(1)),
                "run_limit": int(rm.group(1)) if rm else None, "n_step_calls": len(steps),
                "expected_regs": {str(k): str(v) for k, v in sorted(expected.items())},
                "expected_pc": str(parse_int(pcm.group(1))) if pcm else None,
            })
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vm_kats.json")
    json.dump(kats, open(out, "w"), indent=1)
    print(f"{len(kats)} KATs -> {out}")
    for k in kats:
        print(" ", k["source"], "|", k["name"], "| run", k["run_limit"], "steps", k["n_step_calls"], "| exp", k["expected_regs"], k["expected_pc"])

if __name__ == "__main__":
    main()
