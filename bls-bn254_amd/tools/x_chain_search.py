"""Search for a short signed addition chain for the BN parameter x (used by BN_X_CHAIN, csrc/curve.h): for every dictionary of
up to four odd values below 64 (plus 1), the optimal signed recoding x = sum +-d 2^k (dynamic programme over position and
carry) and the cheapest way to build the dictionary from 1 by doublings and additions/subtractions; cost in Fp multiplications
with a cyclotomic squaring = 18 and a product = 54.  Prints the best candidates; takes a few minutes.
Result used: D = {1, 17, 35}: 62 squarings + 13 products (the unsigned chain of rounds 1-2: 62 + 17)."""
import itertools, functools, sys
X = 0x44E992B44A6909F1
NB = X.bit_length()
def recode(D):
    """min #terms to write X = sum +-d 2^k (d in D), processing from LSB; returns (count, digits list of (k, d))"""
    Ds = sorted(set(D))
    signed = [d for d in Ds] + [-d for d in Ds]
    maxd = max(Ds)
    @functools.lru_cache(None)
    def go(i, c):
        v = (X >> i) + c
        if v == 0: return (0, ())
        if i > NB + 8: return (999, ())
        if v % 2 == 0: return go(i + 1, (v >> 1) - (X >> (i + 1)))
        best = (999, ())
        for d in signed:
            w = v - d
            # w even
            nv = w >> 1
            nc = nv - (X >> (i + 1))
            if abs(nc) > maxd + 2: continue
            r = go(i + 1, nc)
            if r[0] + 1 < best[0]: best = (r[0] + 1, ((i, d),) + r[1])
        return best
    return go(0, 0)
def build_cost(D):
    """greedy cost to build odd set D from 1 with doublings (sqr) and add/sub (mul): returns (muls, sqrs, steps) by BFS over small sets"""
    target = set(D) - {1}
    best = None
    # iterative deepening over sequences: have = set of values; each step: double an element (cost sqr) or add/sub two elements (cost mul)
    import heapq
    start = (frozenset([1]),)
    pq = [(0, 0, 0, frozenset([1]))]
    seen = {}
    while pq:
        cost, m, s, have = heapq.heappop(pq)
        if target <= have: return (m, s, have)
        if seen.get(have, 1e9) <= cost: continue
        seen[have] = cost
        if len(have) > len(target) + 5: continue
        hv = sorted(have)
        for a in hv:
            v = 2 * a
            if v not in have and v <= 64: heapq.heappush(pq, (cost + 18, m, s + 1, have | {v}))
        for a in hv:
            for b in hv:
                if a < b:
                    for v in (a + b, b - a):
                        if v > 0 and v not in have and v < 80 and (v in target or v % 2 == 0 and v <= 16):
                            heapq.heappush(pq, (cost + 54, m + 1, s, have | {v}))
    return None
odds = [d for d in range(3, 64, 2)]
results = []
for k in range(0, 5):
    for extra in itertools.combinations(odds, k):
        D = (1,) + extra
        cnt, digs = recode(D)
        if cnt >= 999: continue
        # main chain: start from top digit; shifts = position of top digit k_top; total doublings in main chain = k_top; top digit value d_top costs nothing extra
        ktop = max(kk for kk, _ in digs)
        bc = build_cost(D)
        if bc is None: continue
        m, s, have = bc
        muls = m + cnt - 1
        sqrs = s + ktop
        cost = muls * 54 + sqrs * 18
        results.append((cost, muls, sqrs, D, digs))
    results.sort(key=lambda r: r[0])
    print("k", k, "best", results[0][:4], file=sys.stderr)
for r in results[:8]: print(r[:4], r[4])
