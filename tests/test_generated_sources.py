"""acvm_amd/csrc/fr_blocks.inc (the gate kernel's asm-block column scans) is generated: the committed file must be what tools/gen_mul_blocks.py emits, every
asm statement must stay within the 30-operand budget, and every form must hold exactly the multiply-adds of the scan it replaces."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fr_blocks_inc_is_the_generators_output():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_mul_blocks.py")], capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(ROOT, "acvm_amd", "csrc", "fr_blocks.inc")).read()


def test_fr_blocks_statements_fit_the_operand_budget_and_count_their_products():
    txt = open(os.path.join(ROOT, "acvm_amd", "csrc", "fr_blocks.inc")).read()
    want = {"fr29_mul_blk": 162, "fr29_dot1_add_blk_v": 162 + 8, "fr29_dot1_add_blk_u": 162 + 8, "fr29_dot2_add_blk_vv": 243 + 8, "fr29_dot2_add_blk_vu": 243 + 8,
            "fr29_dot2_add_blk_uu": 243 + 8}  # N x 81 products + 81 of the reduction (+ the eight limbs of h that ride in the upper columns)
    seen = {}
    for fn in re.split(r"\n(?=__device__)", txt)[1:]:
        name = re.match(r"__device__ __forceinline__ Fr29 (\w+)", fn).group(1)
        mads = 0
        for body, outs, ins in re.findall(r'asm\("(.*?)" : (.*?) : (.*?)\);', fn):
            n_in = len(re.findall(r'"[vs]"\(', ins))
            assert n_in + 3 <= 30, name  # inputs + the read-write accumulator (2) + the carry-out pair
            assert max(int(x) for x in re.findall(r"%(\d+)", body)) <= n_in + 1, name
            assert '"=&s"(cy)' in outs and ('"+v"(acc)' in outs or '"=&v"(acc)' in outs), name
            mads += body.count("v_mad_u64_u32")
        seen[name] = mads
    assert seen == want


# ---- the generated forms EXECUTED on the host: a small interpreter of the two instructions and three C lines fr_blocks.inc is made of, against
# Python integers (the device self test compares the same forms with the C forms on the GPU; this one needs none)
P29 = [0x10000001, 0x1f0fac9f, 0x0e5c2450, 0x07d090f3, 0x1585d283, 0x02db40c0, 0x00a6e141, 0x0e5c2634, 0x0030644e]
P = sum(v << (29 * i) for i, v in enumerate(P29))
MASK64 = (1 << 64) - 1


def _value(expr, env):
    expr = expr.strip()
    m = re.fullmatch(r"fr_p29\((\d+)\)", expr)
    if m:
        return P29[int(m.group(1))]
    m = re.fullmatch(r"(\w+)\.v\[(\d+)\]", expr)
    if m:
        return env[m.group(1)][int(m.group(2))]
    m = re.fullmatch(r"m\[(\d+)\]", expr)
    if m:
        return env["m"][int(m.group(1))]
    raise AssertionError(expr)


def _run(fn_text, env):
    """env: {'a0': [9 limbs], 'b0': ..., 'h': ...}; returns r's nine limbs"""
    env = dict(env, m=[None] * 9, r=[None] * 9)
    acc = None
    for line in fn_text.splitlines():
        line = line.strip()
        st = re.fullmatch(r'asm\("(.*)" : (.*?) : (.*)\);', line)
        if st:
            body, outs, ins = st.groups()
            ops = {}
            for i, (cons, expr) in enumerate(re.findall(r'"([vs])"\(((?:[^()]|\([^()]*\))*)\)', ins)):
                ops[i + 2] = _value(expr, env)
            if '"=&v"(acc)' in outs:
                acc = None  # write-only: its old value must not be read
            for ins_text in body.split("\\n\\t"):
                mad = re.fullmatch(r"v_mad_u64_u32 %0, %1, %(\d+), (%\d+|1), (%0|0)", ins_text)
                if mad:
                    x = ops[int(mad.group(1))]
                    y = 1 if mad.group(2) == "1" else ops[int(mad.group(2)[1:])]
                    assert x < 2 ** 32 and y < 2 ** 32
                    add = 0 if mad.group(3) == "0" else acc
                    assert add is not None
                    acc = (x * y + add) & MASK64
                    assert x * y + add < 2 ** 64  # the column accumulator never wraps
                else:
                    assert ins_text == "v_lshrrev_b64 %0, 29, %0", ins_text
                    acc >>= 29
            continue
        c = re.fullmatch(r"m\[(\d+)\] = \(\(uint32_t\)acc \* 0x0fffffffu\) & M;", line)
        if c:
            env["m"][int(c.group(1))] = (((acc & 0xffffffff) * 0x0fffffff) & 0xffffffff) & 0x1fffffff
            continue
        c = re.fullmatch(r"r\.v\[(\d+)\] = \(uint32_t\)acc & M;", line)
        if c:
            env["r"][int(c.group(1))] = acc & 0x1fffffff
            continue
        c = re.fullmatch(r"r\.v\[8\] = \(uint32_t\)\(acc >> 29\)( \+ h\.v\[8\])?;", line)
        if c:
            env["r"][8] = (((acc >> 29) & 0xffffffff) + (env["h"][8] if c.group(1) else 0)) & 0xffffffff
            continue
        assert line in ("", "{", "}", "constexpr uint32_t M = 0x1fffffffu;", "uint64_t acc, cy;", "uint32_t m[9];", "Fr29 r;", "return r;") or line.startswith(("__device__", "//")), line
    return env["r"]


def _limbs(x):
    return [(x >> (29 * i)) & 0x1fffffff for i in range(8)] + [x >> 232]


def test_fr_blocks_forms_compute_the_montgomery_products_on_the_host():
    import random
    rnd = random.Random(0xB10C)
    txt = open(os.path.join(ROOT, "acvm_amd", "csrc", "fr_blocks.inc")).read()
    fns = {re.match(r"__device__ __forceinline__ Fr29 (\w+)", fn).group(1): fn for fn in re.split(r"\n(?=__device__)", txt)[1:]}
    rinv = pow(2, -261, P)
    cases = [(0, 0), (P - 1, P - 1), (2 ** 256 - 1, 2 ** 256 - 1), (1, 2 ** 256 - 1)] + [(rnd.randrange(2 ** 256), rnd.randrange(2 ** 256)) for _ in range(60)]
    for name, fn in fns.items():
        n = 2 if "dot2" in name else 1
        add = "_add_" in name
        for a0, b0 in cases:
            a1, b1 = rnd.randrange(2 ** 256), rnd.randrange(2 ** 256)
            h = [rnd.randrange(2 ** 32) for _ in range(8)] + [rnd.randrange(2 ** 16)] if add else [0] * 9
            env = {"a0": _limbs(a0), "b0": _limbs(b0), "a1": _limbs(a1), "b1": _limbs(b1), "h": h}
            r = _run(fn, env)
            assert all(v < 2 ** 29 for v in r[:8])
            got = sum(v << (29 * i) for i, v in enumerate(r))
            s = a0 * b0 + (a1 * b1 if n == 2 else 0)
            hv = sum(v << (29 * i) for i, v in enumerate(h))
            # exactly (sum a_t b_t + m p) / 2^261 + h: the same residue as s / R + h, and below p + s / 2^261 + h (fr_device.hpp's contract)
            assert (got - hv) % P == s * rinv % P, name
            assert got - hv <= P + s // 2 ** 261, name
