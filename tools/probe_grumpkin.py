import ctypes as C, sys, random
sys.path.insert(0, '.')
import acvm_amd
from oracle import binding as ob
P = acvm_amd.acir.P
out = C.create_string_buffer(64)
r = random.Random(5)
def xy(b): return (int.from_bytes(b.raw[:32], 'big'), int.from_bytes(b.raw[32:], 'big'))
# device table points
for prm in [0, 511, 512 * 29 + 3, (1 << 24) | 0, (1 << 24) | (32 * 255 * 3 + 31 * 255 + 254), (2 << 24) | 44, (3 << 24) | 2]:
    print("table", hex(prm), acvm_amd.debug_grumpkin(0, prm) == acvm_amd.debug_grumpkin(4, prm))
for v in [0, 1, 2, 511, 512, P - 1, r.randrange(P), r.randrange(P)]:
    for par in (0, 1):
        ob.lib().oracle_pedersen_hash_single(v.to_bytes(32, 'big'), par, out)
        g = acvm_amd.debug_grumpkin(1, par, [v])
        print("hash_single", hex(v)[:12], par, xy(out) == g)
for vs in [[1], [0, 1, 2], [P - 1, 5, 6], [r.randrange(P) for _ in range(3)]]:
    o32 = C.create_string_buffer(32)
    ob.lib().oracle_pedersen_compress(b"".join(v.to_bytes(32, 'big') for v in vs), len(vs), o32)
    g = acvm_amd.debug_grumpkin(2, 0, vs)
    print("compress", len(vs), int.from_bytes(o32.raw, 'big') == g[0])
for k in [1, 2, 255, 256, r.randrange(1 << 254)]:
    ob.lib().oracle_grumpkin_mul_g(k.to_bytes(32, 'big'), out)
    print("mul_g", xy(out) == acvm_amd.debug_grumpkin(3, 0, [k]))
