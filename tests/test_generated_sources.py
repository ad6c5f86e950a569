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
