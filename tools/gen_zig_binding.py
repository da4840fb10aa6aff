#!/usr/bin/env python3
"""Generate bindings/zig/zigz_hip.zig -- the `extern "c"` face of include/zigz_hip.h for the Zig host -- from the header.

    python tools/gen_zig_binding.py [--write]

The reference host is Zig 0.15.2; this image has no Zig toolchain, so the file cannot be compiled here.  Generating it
mechanically from the header (every function, opaque handle, value struct, callback type and constant) at least keeps it
complete and in step with the C ABI: tests/test_abi.py regenerates it and compares with the committed file.  Hand-written
glue (error mapping, the body replacements of the six reference functions) lives in INTEGRATION.md.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "zigz_hip.h")
OUT = os.path.join(ROOT, "bindings", "zig", "zigz_hip.zig")

SCALARS = {"void": "void", "int": "c_int", "unsigned": "c_uint", "double": "f64", "char": "u8", "int32_t": "i32", "uint32_t": "u32",
           "int64_t": "i64", "uint64_t": "u64", "uint8_t": "u8", "size_t": "usize", "zigz_status": "Status"}


def zig_name(c):
    """zigz_commit_job -> CommitJob"""
    return "".join(p.capitalize() for p in c.replace("zigz_", "").split("_"))


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


class Types:
    def __init__(self, opaque, structs, callbacks):
        self.opaque, self.structs, self.callbacks = opaque, structs, callbacks

    def zig(self, ctype, is_param=True):
        """C type text (without the parameter name) -> Zig type."""
        t = " ".join(ctype.replace("*", " * ").split())
        const = False
        toks = t.split()
        stars = toks.count("*")
        toks = [x for x in toks if x != "*"]
        if toks and toks[0] == "const":
            const = True
            toks = toks[1:]
        toks = [x for x in toks if x != "const"]  # `T *const` qualifiers do not matter to the caller
        base = " ".join(toks)
        if base in self.callbacks and stars == 0:
            return zig_name(base)
        if base in self.opaque:
            z = zig_name(base)
            if stars == 1:
                return ("?*const " if const else "?*") + z
            if stars == 2:
                return "[*c]?*" + z
        if base in self.structs:
            z = zig_name(base)
            if stars == 0:
                return z
            if stars == 1:
                return ("[*c]const " if const else "[*c]") + z
        if base == "void" and stars == 1:
            return "?*const anyopaque" if const else "?*anyopaque"
        if base == "void" and stars == 2:
            return "[*c]?*anyopaque"
        if base in SCALARS:
            z = SCALARS[base]
            if stars == 0:
                return z
            if stars == 1:
                return ("[*c]const " if const else "[*c]") + z
            if stars == 2:
                return "[*c][*c]" + z
        raise ValueError("unmapped C type: %r" % ctype)


def split_params(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "(":
            depth += 1
        if ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def param(p, types, idx):
    """'const uint64_t *in' / 'uint8_t root[32]' -> (name, zig type)"""
    p = p.strip()
    if p == "void":
        return None
    m = re.match(r"^(.*?)(\w+)\s*\[\w*\]$", p)  # array parameter decays to a pointer
    if m:
        return m.group(2), types.zig(m.group(1).strip() + " *")
    m = re.match(r"^(.*?)(\w+)$", p)
    ctype, name = m.group(1).strip(), m.group(2)
    if not ctype:  # unnamed parameter
        ctype, name = name, "arg%d" % idx
    return name, types.zig(ctype)


def fn_pointer(text, types):
    """'zigz_status (*block_sums)(void *user, unsigned k, uint64_t *sums)' -> (name, zig fn-pointer type)"""
    m = re.match(r"^(.*?)\(\s*\*\s*(\w+)\s*\)\s*\((.*)\)$", text.strip(), re.S)
    ret, name, params = m.group(1).strip(), m.group(2), m.group(3)
    ps = [param(x, types, i) for i, x in enumerate(split_params(params))]
    ps = [x for x in ps if x]
    sig = ", ".join("%s: %s" % (zid(n), t) for n, t in ps)
    return name, "?*const fn (%s) callconv(.c) %s" % (sig, types.zig(ret))


ZIG_KEYWORDS = {"error", "type", "test", "align", "var", "fn", "packed", "export", "union", "opaque", "suspend", "resume", "async",
                "await", "try", "catch", "defer", "unreachable", "and", "or", "orelse", "struct", "enum", "const", "volatile"}


def zid(n):
    return '@"%s"' % n if n in ZIG_KEYWORDS else n


def generate():
    raw = open(HEADER).read()
    text = strip_comments(raw)
    text = re.sub(r"#ifdef __cplusplus.*?#endif", "", text, flags=re.S)
    opaque = re.findall(r"typedef struct (\w+) \1;", text)
    struct_defs = re.findall(r"typedef struct (\w+) \{(.*?)\} \1;", text, flags=re.S)
    structs = [n for n, _ in struct_defs]
    opaque = [o for o in opaque if o not in structs]
    callbacks = re.findall(r"typedef\s+[\w\s\*]+?\(\s*\*\s*(\w+)\s*\)\s*\(", text)
    types = Types(set(opaque), set(structs), set(callbacks))
    out = []
    w = out.append
    w("//! extern \"c\" binding of include/zigz_hip.h (libzigz_hip.so): the MI355X backend of zigz's prover hot path.")
    w("//! GENERATED by tools/gen_zig_binding.py from the header -- do not edit; regenerate.  Written for Zig 0.15; it has")
    w("//! not been compiled (the build image has no Zig toolchain): tests/test_abi.py keeps it in step with the header.")
    w("//! F must be BabyBear = Field(u64, 2013265921): its memory image is one canonical little-endian u64 per element")
    w("//! (src/core/field.zig:27), so a `[]const F` is passed as `[*c]const u64` without copying.  Error mapping and the")
    w("//! bodies of the replaced reference functions: INTEGRATION.md.")
    w("")
    w("pub const Status = i32;")
    for name, val in re.findall(r"#define (ZIGZ_[A-Z_0-9]+) (\d+)(?:ull)?\b", raw):
        w("pub const %s = %s;" % (name.replace("ZIGZ_", ""), val))
    w("")
    w("// zigz_status values (the Zig error each one maps back to is in the header comment of the code)")
    em = re.search(r"enum \{(.*?)\};", text, flags=re.S)
    for name, val in re.findall(r"(ZIGZ_\w+)\s*=\s*(\d+)", em.group(1)):
        w("pub const %s: Status = %s;" % (name.replace("ZIGZ_", ""), val))
    w("")
    for o in opaque:
        w("pub const %s = opaque {};" % zig_name(o))
    w("")
    for m in re.finditer(r"typedef\s+([\w\s\*]+?)\(\s*\*\s*(\w+)\s*\)\s*\((.*?)\);", text, flags=re.S):
        if m.group(2) in callbacks:
            _, t = fn_pointer("%s (*%s)(%s)" % (m.group(1).strip(), m.group(2), m.group(3)), types)
            w("pub const %s = %s;" % (zig_name(m.group(2)), t))
    w("")
    for name, body in struct_defs:
        w("pub const %s = extern struct {" % zig_name(name))
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            if "(*" in decl:
                fname, ftype = fn_pointer(decl, types)
                w("    %s: %s," % (zid(fname), ftype))
                continue
            m = re.match(r"^([\w\s]+?)\s+([\w\s,\*]+)$", decl)
            ctype, names = m.group(1).strip(), m.group(2)
            for nm in [x.strip() for x in names.split(",")]:
                stars = nm.count("*")
                nm = nm.replace("*", "").strip()
                w("    %s: %s," % (zid(nm), types.zig(ctype + " *" * stars)))
        w("};")
    w("")
    body = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", text, flags=re.S)
    body = re.sub(r"typedef[^;]*;", "", body)
    body = re.sub(r"enum \{.*?\};", "", body, flags=re.S)
    body = re.sub(r"^\s*#.*$", "", body, flags=re.M)
    body = body.replace('extern "C" {', "").replace("}", "")
    for decl in [d.strip() for d in body.split(";")]:
        decl = " ".join(decl.split())
        m = re.match(r"^(.*?)\b(zigz_\w+)\s*\((.*)\)$", decl)
        if not m:
            continue
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3)
        ps = [param(x, types, i) for i, x in enumerate(split_params(params))]
        ps = [x for x in ps if x]
        sig = ", ".join("%s: %s" % (zid(n), t) for n, t in ps)
        w("pub extern \"c\" fn %s(%s) %s;" % (name, sig, types.zig(ret)))
    return "\n".join(out) + "\n"


LAYOUT_OUT = os.path.join(ROOT, "bindings", "zig", "zigz_hip_layout_check.c")
ZIG_SIZES = {"u8": 1, "i8": 1, "u16": 2, "i16": 2, "u32": 4, "i32": 4, "c_int": 4, "c_uint": 4, "u64": 8, "i64": 8, "f64": 8, "f32": 4,
             "usize": 8, "isize": 8, "Status": 4}


def layout_check(zig_text=None):
    """No Zig compiler here, so what CAN be checked without one: the layout of every `extern struct` of the generated binding.
    Parsed from the ZIG side (field names and Zig types of bindings/zig/zigz_hip.zig), laid out by the C ABI rules an
    `extern struct` follows on x86-64 (natural alignment, size rounded up to the largest member's), and written as C
    _Static_asserts on offsetof / sizeof of the header's own struct: tests/test_c_driver.py compiles the file against
    include/zigz_hip.h.  A field the generator mistyped (a u32 where the header has a size_t), dropped or reordered fails
    the compile."""
    zig_text = zig_text if zig_text is not None else generate()
    out = ["/* GENERATED by tools/gen_zig_binding.py from the ZIG-side field lists of bindings/zig/zigz_hip.zig -- do not edit.",
           " * The layout Zig gives every `extern struct` of the binding (C ABI, x86-64) must be the layout the C compiler gives the",
           " * header's struct of the same name. */", "#include <stddef.h>", '#include "zigz_hip.h"', ""]
    names = {}
    for cname in re.findall(r"typedef struct (\w+) \{", strip_comments(open(HEADER).read())):
        names[zig_name(cname)] = cname
    for m in re.finditer(r"pub const (\w+) = extern struct \{(.*?)\n\};", zig_text, flags=re.S):
        zname, body = m.group(1), m.group(2)
        cname = names[zname]
        off, maxal = 0, 1
        for fm in re.finditer(r"^\s*(@\"\w+\"|\w+): (.*),$", body, flags=re.M):
            fname, ztype = fm.group(1).strip('@"'), fm.group(2).strip()
            if ztype in ZIG_SIZES:
                size = ZIG_SIZES[ztype]
            elif ztype.startswith(("?*", "[*c]", "*")):
                size = 8  # pointers, optional pointers, function pointers
            else:
                raise ValueError("layout_check: no size for Zig type %r (%s.%s)" % (ztype, zname, fname))
            al = size
            off = (off + al - 1) // al * al
            out.append("_Static_assert(offsetof(%s, %s) == %d, \"%s.%s: Zig-side offset %d\");" % (cname, fname, off, zname, fname, off))
            off += size
            maxal = max(maxal, al)
        total = (off + maxal - 1) // maxal * maxal
        out.append("_Static_assert(sizeof(%s) == %d, \"%s: Zig-side size %d\");" % (cname, total, zname, total))
        out.append("")
    return "\n".join(out)


def main():
    txt = generate()
    if "--write" in sys.argv:
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        open(OUT, "w").write(txt)
        print("wrote", OUT, "(%d lines)" % txt.count("\n"))
        open(LAYOUT_OUT, "w").write(layout_check(txt))
        print("wrote", LAYOUT_OUT)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main()
