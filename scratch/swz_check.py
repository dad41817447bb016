"""bank-conflict check of the K=1024 tile swizzle for every LDS access pattern of channelizer_kernel (8-byte elements).
ds_read_b64: lane groups {0-31},{32-63}, slot = element mod 32; ds_write_b64: groups of 16 lanes, slot = element mod 16."""
import numpy as np
K, T, F = 1024, 512, 16
def swz(e, row):
    e = np.asarray(e)
    m = (((e >> 5) ^ (e >> 9)) & 1) | (((e >> 6) & 1) << 1) | (((e >> 7) & 1) << 2) | (((e >> 8) & 1) << 3) | (((e >> 6) & 1) << 4)
    return (e ^ m ^ ((((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 4))) + row * K
def worst(addr, group, mod):
    w = 1
    addr = np.asarray(addr)
    for g in range(0, len(addr), group):
        a = np.unique(addr[g:g + group])
        slots = a % mod
        w = max(w, np.max(np.bincount(slots, minlength=mod)))
    return w
tid = np.arange(T)
assert len(np.unique(swz(np.arange(K), 3) - 3 * K)) == K          # bijection within a row
res = {}
for st in range(3):
    L = K >> (2 * st); q4 = L >> 2
    for i in range(4):
        q = tid + i * T
        f, j = q // (K // 4), q % (K // 4)
        grp, pos = j // q4, j % q4
        e0 = grp * L + pos
        for r in range(4):
            res["radix4 st%d leg%d" % (st, r)] = max(res.get("radix4 st%d leg%d" % (st, r), 1),
                worst(swz(e0 + r * q4, f)[:64], 32, 32), worst(swz(e0 + r * q4, f)[:64], 16, 16))
g = tid                       # final stage: NG = 8 * 64 = 512 groups, one per thread
f, gi = g // (K // F), g % (K // F)
for m in range(F):
    res["final m=%d" % m] = max(worst(swz(gi * F + m, f)[:64], 32, 32), worst(swz(gi * F + m, f)[:64], 16, 16))
n0 = 2 * tid
for c in range(2):
    res["fir write c=%d" % c] = worst(swz(n0 + c, 5)[:64], 16, 16)
def dif_pos(k):
    Lh, pos = K, 0
    k = np.asarray(k).copy()
    for s in range(3):
        pos = pos + (k & 3) * (Lh >> 2); k >>= 2; Lh >>= 2
    return pos + k
for i in range(4):
    o = tid + i * T
    ch, rp = o // 4, o % 4
    res["store read i=%d" % i] = worst(swz(dif_pos(ch), 2 * rp)[:64], 32, 32)
for k, v in res.items():
    print("%-22s worst %d-way" % (k, v))
