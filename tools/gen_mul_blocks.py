#!/usr/bin/env python3
"""tools/gen_mul_blocks.py > acvm_amd/csrc/fr_blocks.inc -- the column scans of fr29_mul / fr29_dot_add (fr_device.hpp) as asm blocks.

Why: hipcc re-associates `acc = (acc >> 29) + products`, starts every column from 0 ahead of time and joins it with the carry of the column before by
a v_lshl_add_u64 -- 17 of the 221 VALU instructions of a product, and an instruction costs its issue slot on this part. A multiply-add has one free
addend: a column whose FIRST multiply-add takes the carry needs no join. The compiler cannot be told not to re-associate, so the scan is written as
asm; dependent v_mad_u64_u32 may sit back to back (tools/mad_hazard_probe.hip: the hardware interlocks), but hipcc puts a wait state behind every asm
STATEMENT whose result the next instruction reads, so a statement holds as many instructions as its operand budget allows (30 operands, a read-write
one counts twice). The low word of the accumulator cannot be named inside asm: m_k = column * (-1/p) mod 2^29 and the result limbs are C between
the statements. Measured (tools/serial_mul_probe.hip): products back to back x1.04 - x1.06 of the compiler's form at 4 - 8 waves per SIMD, x1.00 at 2.

Emits, all __device__:  fr29_mul_blk(a, b);  fr29_dot1_add_blk_{v,u}(a0, b0, h);  fr29_dot2_add_blk_{vv,vu,uu}(a0, b0, a1, b1, h)
(v / u per product: the second factor b_t is per-lane / WAVE-UNIFORM and then stays in scalar registers -- a VOP3 instruction takes one scalar source).
Same contracts as the C forms (fr_device.hpp fr29_mul, fr29_dot_impl)."""
BUDGET = 27  # inputs per statement: 30 operands - the accumulator (read-write: 2) - the carry-out scalar pair


class Stmt:
    def __init__(self):
        self.names = {}
        self.lines = []

    def ref(self, name, cons):
        key = (name, cons)
        if key not in self.names:
            self.names[key] = 2 + len(self.names)
        return "%%%d" % self.names[key]

    def would_need(self, opnds):
        return len(self.names) + sum(1 for o in opnds if o not in self.names)


class Emitter:
    def __init__(self):
        self.out = []
        self.cur = Stmt()
        self.defined = False  # acc has a value

    def mad(self, x, y, y_cons="v", addend_zero=False):
        """acc = x * y + acc (x per-lane; y per-lane or uniform)"""
        opnds = [(x, "v"), (y, y_cons)]
        if self.cur.would_need(opnds) > BUDGET:
            self.flush()
        ax, ay = self.cur.ref(x, "v"), self.cur.ref(y, y_cons)
        self.cur.lines.append("v_mad_u64_u32 %%0, %%1, %s, %s, %s" % (ax, ay, "0" if addend_zero else "%0"))

    def add32(self, x):
        if self.cur.would_need([(x, "v")]) > BUDGET:
            self.flush()
        self.cur.lines.append("v_mad_u64_u32 %%0, %%1, %s, 1, %%0" % self.cur.ref(x, "v"))

    def shift(self):
        self.cur.lines.append("v_lshrrev_b64 %0, 29, %0")

    def flush(self):
        if not self.cur.lines:
            return
        ins = ", ".join('"%s"(%s)' % (c, n) for (n, c), _ in sorted(self.cur.names.items(), key=lambda kv: kv[1]))
        out = '"+v"(acc)' if self.defined else '"=&v"(acc)'  # (early clobber: later instructions of the statement still read its inputs)
        self.out.append('    asm("%s" : %s, "=&s"(cy) : %s);' % ("\\n\\t".join(self.cur.lines), out, ins))
        self.defined = True
        self.cur = Stmt()

    def c(self, text):
        self.flush()
        self.out.append("    " + text)


def scan(name, n, uniform, add):
    """uniform: tuple of bools per product"""
    args = ", ".join("const Fr29 &a%d, const Fr29 &b%d" % (t, t) for t in range(n)) + (", const Fr29 &h" if add else "")
    e = Emitter()
    first = True
    for k in range(9):
        if k > 0:
            e.mad("m[%d]" % (k - 1), "fr_p29(0)", "s")
            e.shift()
        for t in range(n):
            for i in range(k + 1):
                e.mad("a%d.v[%d]" % (t, i), "b%d.v[%d]" % (t, k - i), "s" if uniform[t] else "v", addend_zero=first)
                first = False
        for i in range(k):
            e.mad("m[%d]" % i, "fr_p29(%d)" % (k - i), "s")
        e.c("m[%d] = ((uint32_t)acc * 0x0fffffffu) & M;" % k)
    for k in range(9, 17):
        if k == 9:
            e.mad("m[8]", "fr_p29(0)", "s")
        e.shift()
        for t in range(n):
            for i in range(k - 8, 9):
                e.mad("a%d.v[%d]" % (t, i), "b%d.v[%d]" % (t, k - i), "s" if uniform[t] else "v")
        for i in range(k - 8, 9):
            e.mad("m[%d]" % i, "fr_p29(%d)" % (k - i), "s")
        if add:
            e.add32("h.v[%d]" % (k - 9))
        e.c("r.v[%d] = (uint32_t)acc & M;" % (k - 9))
    e.c("r.v[8] = (uint32_t)(acc >> 29)%s;" % (" + h.v[8]" if add else ""))
    body = "\n".join(e.out)
    return ("__device__ __forceinline__ Fr29 %s(%s) {\n    constexpr uint32_t M = 0x1fffffffu;\n    uint64_t acc, cy;\n    uint32_t m[9];\n    Fr29 r;\n%s\n    return r;\n}\n"
            % (name, args, body))


if __name__ == "__main__":
    print("// fr_blocks.inc -- GENERATED by tools/gen_mul_blocks.py (read its header); included by fr_device.hpp in the device pass. Do not edit.")
    print(scan("fr29_mul_blk", 1, (False,), False).replace("const Fr29 &a0, const Fr29 &b0", "const Fr29 &a0, const Fr29 &b0"))
    print(scan("fr29_dot1_add_blk_v", 1, (False,), True))
    print(scan("fr29_dot1_add_blk_u", 1, (True,), True))
    print(scan("fr29_dot2_add_blk_vv", 2, (False, False), True))
    print(scan("fr29_dot2_add_blk_vu", 2, (False, True), True))
    print(scan("fr29_dot2_add_blk_uu", 2, (True, True), True))
