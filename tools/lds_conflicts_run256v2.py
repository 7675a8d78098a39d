#!/usr/bin/env python3
"""Static LDS bank-conflict count of k_run256v2<FM>'s access patterns (whole band), same model as tools/lds_conflicts_run1024v3.py.
Usage: lds_conflicts_run256v2.py [PAD=1] [banks=64]   PAD=1: round 5's image (frames 2176 bytes apart, Z swizzle (f >> 1) & 7: kernels_fused_v2.hip
V2_PAD=1); PAD=0: round 4's dense image (382 cycles per tile and wave against 286 conflict-free)."""
import sys
PAD = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 64
FSB = 2048 + 128 * PAD
def zsw(f): return ((f >> 1) & 7) if PAD else (f & 7)
def cost(addrs, width):
    per = {8: 32 * NB // 64, 16: 16 * NB // 64}[width]; total = 0
    for g in range(0, 64, per):
        banks = {}
        for l in range(g, g + per):
            a = addrs[l]
            if a is None: continue
            for w in range(width // 4):
                dw = a // 4 + w; banks.setdefault(dw % NB, set()).add(dw)
        total += max((len(v) for v in banks.values()), default=0)
    return total
def report(name, fn, width, count, variants):
    tot = 0; worst = 0
    for v in variants:
        c = cost([fn(l, v) for l in range(64)], width); tot += c; worst = max(worst, c)
    ideal = 64 * width // (4 * NB)
    print(f"{name:52s} b{width * 8:<3d} x{count:3d}/thread/tile: {tot / len(variants):5.2f} cycles per instruction (ideal {ideal}), worst {worst}")
    return tot / len(variants) * count, ideal * count
W = range(4); acc = []
tid = lambda l, w: 64 * w + l
def col_off(j): return 16 * (j >> 4) + 2 * (((j & 15) >> 1) ^ (j >> 5)) + (j & 1)
acc.append(report("scan read/write raw_a ^ (i << 4)", lambda l, v: ((tid(l, v[0]) >> 4) * FSB + (tid(l, v[0]) & 15) * 128 + ((((tid(l, v[0]) >> 1) & 7) ^ v[1]) << 4)), 16, 16, [(w, i) for w in W for i in range(8)]))
acc.append(report("column read / X write Bf[256 f + col_off]", lambda l, v: FSB * v[1] + 8 * col_off(tid(l, v[0])), 8, 32, [(w, f) for w in W for f in range(16)]))
def x_a(t): return (t >> 4) * FSB + (t & 15) * 8
acc.append(report("pass 1 X read (x_a ^ ((a >> 1) << 4)) + 128 a", lambda l, v: (x_a(tid(l, v[0])) ^ ((v[1] >> 1) << 4)) + 128 * v[1], 8, 16, [(w, a) for w in W for a in range(16)]))
TW = 2 * 16 * FSB + 8 * (256 + 32)
acc.append(report("pass 1 twiddle tw_s[16 k1 + b1]", lambda l, v: TW + 8 * (16 * v + (l & 15)), 8, 15, list(range(1, 16))))
def zw_a(t): f1, b1 = t >> 4, t & 15; return f1 * FSB + ((((b1 >> 1) ^ zsw(f1)) << 1) | (b1 & 1)) * 8
acc.append(report("pass 1 Z write zw_a + 128 k1", lambda l, v: zw_a(tid(l, v[0])) + 128 * v[1], 8, 16, [(w, k) for w in W for k in range(16)]))
def z_a(t): k1, f2 = t >> 4, t & 15; return f2 * FSB + k1 * 128 + (zsw(f2) << 4)
acc.append(report("pass 2 Z read z_a ^ (i << 4)", lambda l, v: z_a(tid(l, v[0])) ^ (v[1] << 4), 16, 8, [(w, i) for w in W for i in range(8)]))
STB = 2 * 16 * FSB
acc.append(report("stash read ST + k1 16 + i (all lanes of a k1 row)", lambda l, v: STB + 8 * ((tid(l, v[0]) >> 4) * 16 + v[1]), 16, 8, [(w, i) for w in W for i in range(0, 16, 2)]))
tot = sum(a for a, _ in acc); ideal = sum(b for _, b in acc)
print(f"LDS cycles per tile and wave: {tot:.0f}, conflict-free {ideal:.0f}")
