"""Search of an XOR swizzle for the LDS layout of csrc/fft.hip: conflicts (16-byte units per bank and
16-lane group) of every access pattern of the kernels for transform lengths 2^10 .. 2^13."""
import itertools, numpy as np
def bitrev(i, logn):
    r = 0
    for b in range(logn): r |= ((i >> b) & 1) << (logn - 1 - b)
    return r
def patterns(logn):
    N = 1 << logn
    pats = []
    # P1 bitrev store: 16 consecutive i
    for i0 in range(0, N, 16):
        pats.append([bitrev(i0 + t, logn) for t in range(16)])
    s = 0
    if logn & 1:
        for q0 in range(0, N >> 1, 16):
            for k in range(2): pats.append([((q0 + t) << 1) + k for t in range(16)])
        s = 1
    while s < logn:
        h = 1 << s
        for q0 in range(0, N >> 2, 16):
            for k in range(4):
                pats.append([(((q0 + t) >> s) << (s + 2)) + ((q0 + t) & (h - 1)) + k * h for t in range(16)])
        s += 2
    # untangle: k and N-k
    for k0 in range(0, N // 2, 16):
        pats.append([k0 + t for t in range(16)])
        pats.append([(N - (k0 + t)) & (N - 1) for t in range(16)])
    return pats
def score(f, logn):
    worst = 0; total = 0
    for p in patterns(logn):
        units = [f(i) & 15 for i in p]
        c = max(np.bincount(units, minlength=16))
        worst = max(worst, c); total += c
    return worst, total
def mk(shifts):
    def f(i):
        x = 0
        for sh in shifts: x ^= (i >> sh)
        return i ^ (x & 15)
    return f
best = []
for r in range(1, 4):
    for shifts in itertools.combinations(range(4, 12), r):
        f = mk(shifts)
        # bijective? sources are bits >= 4 only -> yes
        w = 0; tot = 0
        for logn in (10, 11, 12, 13):
            a, b = score(f, logn); w = max(w, a); tot += b
        best.append((w, tot, shifts))
best.sort()
print(best[:8])
print("pad i+(i>>6):", [score(lambda i: i + (i >> 6), l) for l in (10, 11, 12, 13)])
