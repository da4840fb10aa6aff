# generates the bit-interleaved Keccak-f[1600] round body (e/o = even/odd bits of each lane)
RHO = {}
x, y = 1, 0
for t in range(24):
    RHO[(x, y)] = ((t + 1) * (t + 2) // 2) % 64
    x, y = y, (2 * x + 3 * y) % 5
RHO[(0, 0)] = 0
def rot(v, k):
    k %= 32
    return v if k == 0 else f"ZK_ROT32({v}, {k})"
out = []
for x in range(5):
    out.append(f"        const uint32_t ce{x} = ZK_X3(ZK_X3(e[{x}], e[{x+5}], e[{x+10}]), e[{x+15}], e[{x+20}]);")
    out.append(f"        const uint32_t co{x} = ZK_X3(ZK_X3(o[{x}], o[{x+5}], o[{x+10}]), o[{x+15}], o[{x+20}]);")
for x in range(5):
    out.append(f"        const uint32_t re{x} = ZK_ROT32(co{x}, 1);  // rot64(C[{x}], 1): even <- rot32(odd, 1), odd <- even")
for y in range(5):
    for x in range(5):
        i = x + 5 * y
        r = RHO[(x, y)]
        dst = y + 5 * ((2 * x + 3 * y) % 5)
        xm, xp = (x + 4) % 5, (x + 1) % 5
        te = f"ZK_X3(e[{i}], ce{xm}, re{xp})"
        to = f"ZK_X3(o[{i}], co{xm}, ce{xp})"
        if r % 2 == 0:
            k = r // 2
            be, bo = rot("te", k), rot("to", k)
        else:
            k = r // 2
            be, bo = rot("to", k + 1), rot("te", k)
        out.append(f"        {{ const uint32_t te = {te}, to = {to};  // lane {i}, rho {r} -> {dst}")
        out.append(f"          be[{dst}] = {be}; bo[{dst}] = {bo}; }}")
for y in range(5):
    for x in range(5):
        i = x + 5 * y
        a, b, c = i, 5 * y + (x + 1) % 5, 5 * y + (x + 2) % 5
        out.append(f"        e[{i}] = ZK_CHI(be[{a}], be[{b}], be[{c}]); o[{i}] = ZK_CHI(bo[{a}], bo[{b}], bo[{c}]);")
print("\n".join(out))
RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808a, 0x8000000080008000, 0x000000000000808b, 0x0000000080000001,
      0x8000000080008081, 0x8000000000008009, 0x000000000000008a, 0x0000000000000088, 0x0000000080008009, 0x000000008000000a,
      0x000000008000808b, 0x800000000000008b, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
      0x000000000000800a, 0x800000008000000a, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
def il(v):
    e = o = 0
    for b in range(32):
        e |= ((v >> (2 * b)) & 1) << b
        o |= ((v >> (2 * b + 1)) & 1) << b
    return e, o
print("RC_E = {" + ", ".join("0x%08xu" % il(v)[0] for v in RC) + "};")
print("RC_O = {" + ", ".join("0x%08xu" % il(v)[1] for v in RC) + "};")
