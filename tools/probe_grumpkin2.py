import ctypes as C, sys, random
sys.path.insert(0, '.')
import acvm_amd
P = acvm_amd.acir.P
BETA = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
def add(p, q):
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if (p[1] + q[1]) % P == 0: return None
        l = 3 * p[0] * p[0] * pow(2 * p[1], -1, P) % P
    else: l = (q[1] - p[1]) * pow(q[0] - p[0], -1, P) % P
    x = (l * l - p[0] - q[0]) % P
    return (x, (l * (p[0] - x) - p[1]) % P)
def tbl(i): return acvm_amd.debug_grumpkin(0, i)
for v in [0, 12345678901234567890123]:
    for par in (0, 1):
        acc = None
        for i in range(15): acc = add(acc, tbl((par * 15 + i) * 512 + ((v >> (18 * i)) & 511)))
        print("acc0 loop", par, acvm_amd.debug_grumpkin(5, par, [v]) == acc)
for v in [1, 5, P - 1, 0x1234567890abcdef << 100]:
    g = acvm_amd.debug_grumpkin(6, 0, [v])
    print("beta mul", g[0] == v * BETA % P, g[1] == BETA)
p0, p1 = tbl(0), tbl(512)
print("gj_add", acvm_amd.debug_grumpkin(7, 0) == add(p0, add(p1, p0)))
