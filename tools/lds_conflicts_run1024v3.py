#!/usr/bin/env python3
"""Static LDS bank-conflict count of k_run1024v3's access patterns (addresses per lane exactly as the kernel computes them).
Model (gfx9 LDS, 64 banks x 4 B): a ds_read/write_b64 wave instruction is served in 2 groups of 32 lanes, a b128 in 4 groups of 16;
within a group, lanes that hit the same bank at DIFFERENT addresses serialise: cost of a group = max over banks of the number of distinct
dwords addressed in it.  Prints, per access, the cycles per wave instruction and the conflict-free minimum."""
import sys
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 64      # LDS banks of the model (64: the model whose four findings the SQ counters confirmed; 32: stricter, names the b128 accesses too)

def cost(addrs, width):                       # addrs: byte address per lane (64), width: 8 or 16
    per = {8: 32 * NB // 64, 16: 16 * NB // 64}[width]
    total = 0
    for g in range(0, 64, per):
        banks = {}
        for l in range(g, g + per):
            a = addrs[l]
            if a is None: continue
            for w in range(width // 4):
                dw = a // 4 + w
                banks.setdefault(dw % NB, set()).add(dw)
        total += max((len(v) for v in banks.values()), default=0)
    return total

def report(name, fn, width, count, variants):
    worst = 0; tot = 0
    for v in variants:
        c = cost([fn(l, v) for l in range(64)], width); tot += c; worst = max(worst, c)
    ideal = 64 * width // (4 * NB)
    print(f"{name:46s} b{width * 8:<3d} x{count:3d}/thread/tile: {tot / len(variants):5.2f} cycles per instruction (ideal {ideal}), worst {worst}")
    return tot / len(variants) * count, ideal * count

acc = []
# ---- front wave w (0..3), lane l: j = q = 64 w + l
for w in range(4):
    pass
def q_of(l, w): return 64 * w + l
W = [0, 1, 2, 3]
# scan: raw_a ^ (i << 4), raw_a = q * 128 + (((q >> 1) & 7) << 4), i = 0..7 (b128 read + write)
acc.append(report("front scan read/write (raw image)", lambda l, v: (q_of(l, v[0]) * 128 + ((((q_of(l, v[0]) >> 1) & 7) ^ v[1]) << 4)), 16, 16, [(w, i) for w in W for i in range(8)]))
def col_off(j): return 16 * (j >> 4) + 2 * (((j & 15) >> 1) ^ (j >> 5)) + (j & 1)
acc.append(report("front column read / X write Bf[256 g + col_off]", lambda l, v: 8 * (256 * v[1] + col_off(q_of(l, v[0]))), 8, 32, [(w, g) for w in W for g in range(16)]))
# pass 1 (front wave f): b1 = l
def x_a(b1): return 8 * ((16 * (b1 >> 4)) | (b1 & 1) | (2 * ((((b1 & 15) >> 1) ^ (b1 >> 5)) & 7)))
def z1w(b1): return 128 * (b1 & 3) + 8 * ((b1 >> 2) & 1) + 16 * (((b1 >> 3) ^ ((b1 & 3) >> 1) ^ (((b1 & 3) >> 1) << 2)) & 7)
acc.append(report("pass 1 X read fb + 512 a + (x_a ^ ((a&3)<<5))", lambda l, v: 8192 * v[0] + 512 * v[1] + (x_a(l) ^ ((v[1] & 3) << 5)), 8, 16, [(f, a) for f in W for a in range(16)]))
TW1 = 4 * 32768
acc.append(report("pass 1 twiddle tw1[64 (k1 - 1) + b1]", lambda l, v: TW1 + 8 * (64 * (v - 1) + l), 8, 15, list(range(1, 16))))
acc.append(report("pass 1 Z1 write fb + 512 k1 + (z1w ^ ..)", lambda l, v: 8192 * v[0] + 512 * v[1] + (z1w(l) ^ (((2 * v[1]) & 6) << 4)), 8, 16, [(f, k) for f in W for k in range(16)]))
# ---- back wave f, lane l2
acc.append(report("pass 2 Z1 read z1r ^ (i << 4)", lambda l, v: (8192 * v[0] + l * 128 + ((((l >> 1) & 7) ^ (((l >> 1) & 1) << 2)) << 4)) ^ (v[1] << 4), 16, 8, [(f, i) for f in W for i in range(8)]))
TW2 = TW1 + 8 * (960 + 1024 + 16)
acc.append(report("pass 2 twiddle tw2[4 k2 + d]", lambda l, v: (TW2 + 8 * (4 * v + (l & 3))) if (l & 3) else None, 8, 15, list(range(1, 16))))
acc.append(report("pass 2 Z2 write (halves swapped for kk & 8)", lambda l, v: 8192 * v[0] + 32 * (l >> 2) + 16 * (((l >> 1) & 1) ^ ((l >> 5) & 1)) + 8 * (l & 1) + 512 * v[1], 8, 16, [(f, k) for f in W for k in range(16)]))
acc.append(report("pass 3 Z2 read 8192 f + 32 kk (+16)", lambda l, v: (8192 * v[1] + 32 * (64 * v[0] + l) + 16 * ((l >> 3) & 1)) ^ (16 * v[2]), 16, 8, [(w, f, h) for w in W for f in range(4) for h in range(2)]))
ST = TW1 + 8 * 960
acc.append(report("freqdem stash read / write ST + 32 kk (+16)", lambda l, v: (ST + 32 * (64 * v[0] + l) + 16 * ((l >> 3) & 1)) ^ (16 * v[1]), 16, 4, [(w, h) for w in W for h in range(2)]))
# flush (per block of 8 tiles: / 8 per tile)
def fl_w(l, wv): return 8192 * (l >> 4) + 2048 * wv + 128 * (l & 15) + (((l >> 1) & 7) << 4)
def fl_r(l, wv, m): return (2048 * wv + 128 * (l >> 3) + 16 * ((l & 7) ^ (l >> 4)) if (m & 1) == 0 else 2048 * wv + 1024 + 128 * (l >> 3) + 16 * ((l & 7) ^ (4 + (l >> 4)))) + 8192 * (m >> 1)
acc.append(report("flush write fl_w ^ (p << 4)   [per 8 tiles]", lambda l, v: fl_w(l, v[0]) ^ (v[1] << 4), 16, 4, [(w, p) for w in W for p in range(8)]))
acc.append(report("flush read fl_r(m)            [per 8 tiles]", lambda l, v: fl_r(l, v[0], v[1]), 16, 4, [(w, m) for w in W for m in range(8)]))
tot = sum(a for a, _ in acc); ideal = sum(b for _, b in acc)
print(f"LDS cycles per tile and wave (front + back thread both counted once): {tot:.0f}, conflict-free {ideal:.0f}")
