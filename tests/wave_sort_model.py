"""Lane-by-lane CPU model of csrc/sot_wave_sort.hpp (round 6): ONE wavefront sorts a whole array of 64 KPL keys in registers.

  * every key becomes ONE 32-bit word: a row-adaptive quantisation q = trunc((x - min) * C / (max - min)) in the high bits (monotone in x),
    the element's index in the low IDXBITS = 6 + log2(KPL) bits -- ordering the words orders (q, index);
  * a payload-free bitonic network in "flip" form (every compare-exchange ascending) on the blocked layout position = lane KPL + register:
    exchanges between registers are v_min_u32 + v_max_u32, exchanges between lanes are one cross-lane move (DPP / ds_swizzle / ds_bpermute:
    source lane = lane ^ m) + v_med3_u32 against a per-lane bound (0: keep the minimum, ~0: keep the maximum);
  * runs of EQUAL q are then put into the exact order by an insertion sort on the full keys (stable: the network left each run in index
    order); a run longer than RUN_LIMIT, non-finite keys or a degenerate range return "not sorted" (the caller takes the merge sort).

The model mirrors the kernel's data movement (the lane / register index algebra of every stage, the skewed transposition image) so that
a wrong partner or bound shows up here, without a GPU.  tests/test_wave_sort_model.py checks it against numpy's stable argsort."""
import numpy as np

RUN_LIMIT = 8


def order_bits(f):
    """float32 -> uint32 whose unsigned order is the float order (-0 == +0), as sot_device.hpp: float_order_bits."""
    u = np.asarray(f, np.float32).view(np.uint32).copy()
    u[u == 0x80000000] = 0
    neg = (u & 0x80000000) != 0
    return np.where(neg, ~u, u ^ np.uint32(0x80000000)).astype(np.uint32)


def pack_words(keys, n, kpl):
    """(words[64 * kpl] in load order e = r 64 + lane, ok): the kernel's pre-pass + packing, float32 arithmetic operation for operation."""
    npad = 64 * kpl
    idxbits = 6 + int(np.log2(kpl))
    qbits = 32 - idxbits
    qmax = (1 << qbits) - 1
    x = np.asarray(keys[:n], np.float32)
    if np.isnan(x).any():
        return None, False
    mn, mx = np.float32(x.min()), np.float32(x.max())
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        rng = np.float32(mx - mn)
        scale = np.float32(np.float32(qmax - 8) / rng)
    if not (np.isfinite(rng) and np.isfinite(scale) and rng > 0):
        return None, False
    with np.errstate(over="ignore", invalid="ignore"):
        q = ((x - mn).astype(np.float32) * scale).astype(np.float32)
    q = np.trunc(q).astype(np.uint64)
    assert q.max() <= qmax - 1
    w = np.empty(npad, np.uint32)
    w[:n] = ((q << idxbits) | np.arange(n, dtype=np.uint64)).astype(np.uint32)
    w[n:] = 0xFFFFFFFF                     # every pad is the SAME word: pads are behind the data and never form a run (xor == 0)
    return w, True


def lane_xor(w, m):
    """what every lane receives from lane ^ m (w: [64, kpl])"""
    return w[np.arange(64) ^ m]


def med3(a, b, c):
    return np.maximum(np.minimum(a, b), np.minimum(np.maximum(a, b), c))


TRANSPOSE_MIN = 4        # sot_wave_sort.hpp: SOT_WSORT_TRANSPOSE_MIN


def transpose32(w, count):
    """register r of lane l <-> register (l & 31) of lane (l & 32) + r through the skewed image (position p at p + p / 32), as the kernel:
    store with the blocked map (lane l, register r at 33 l + r), load with the crossed map (lane L, register R at 33 (32 (L >> 5) + R) + (L & 31))"""
    lane = np.arange(64)
    image = np.zeros(64 * 33 + 2, np.uint32)
    for r in range(32):
        addr = 33 * lane + r
        assert len(set((addr[:32] % 32).tolist())) == 32 and len(set((addr[32:] % 32).tolist())) == 32
        image[addr] = w[:, r]
        count("lds_store")
    out = np.empty_like(w)
    for r in range(32):
        addr = 33 * (32 * (lane >> 5) + r) + (lane & 31)
        assert len(set((addr[:32] % 32).tolist())) == 32 and len(set((addr[32:] % 32).tolist())) == 32
        out[:, r] = image[addr]
        count("lds_load")
    return out


def network(w, kpl, stats=None):
    """the bitonic network on w[lane, r] (position = lane * kpl + r); returns the sorted image in the same layout"""
    w = w.copy()
    lane = np.arange(64)

    def count(kind, n=1):
        if stats is not None:
            stats[kind] = stats.get(kind, 0) + n

    def ce_regs(j_pairs):
        for a, b in j_pairs:
            lo, hi = np.minimum(w[:, a], w[:, b]), np.maximum(w[:, a], w[:, b])
            w[:, a], w[:, b] = lo, hi          # (in place: `w` is rebound by the cross-lane stages, and this closure follows the rebinding)
            count("valu_inreg", 2)

    def bound(bit):
        return np.where(lane & bit, np.uint32(0xFFFFFFFF), np.uint32(0))

    def tail(j_top):                       # half cleaners between registers: distances j_top, j_top / 2, ..., 1
        j = j_top
        while j >= 1:
            ce_regs([(r, r | j) for r in range(kpl) if not r & j])
            j >>= 1

    p = 1
    while p < kpl:                         # the lane's own registers: Batcher's odd-even merge sort (Knuth's iterative form, as the kernel)
        k = p
        while k >= 1:
            pairs = []
            for j in range(k % p, kpl - k, 2 * k):
                for i in range(k):
                    a, b = i + j, i + j + k
                    if b < kpl and a // (2 * p) == b // (2 * p):
                        pairs.append((a, b))
            ce_regs(pairs)
            k //= 2
        p *= 2
    s = 2
    while s <= 64:                         # merges across lanes: runs of (s / 2) kpl -> s kpl
        src = lane_xor(w, s - 1)           # flip: partner = (lane ^ (s - 1), kpl - 1 - r)
        b = bound(s >> 1)
        new = np.empty_like(w)
        for r in range(kpl):
            new[:, r] = med3(w[:, r], src[:, kpl - 1 - r], b)
            count("move"); count("valu_med3")
        w = new
        if kpl == 32 and (s >> 2) >= TRANSPOSE_MIN:
            # the lane stages at distances s / 4 ... 1 as exchanges between REGISTERS on the transposed layout (the lane's low five bits and the
            # register index change places through the skewed LDS image), then back
            w = transpose32(w, count)
            tail(s >> 2)
            w = transpose32(w, count)
        else:
            d = s >> 2
            while d >= 1:                  # half cleaners between lanes at distance d
                src = lane_xor(w, d)
                b = bound(d)
                new = np.empty_like(w)
                for r in range(kpl):
                    new[:, r] = med3(w[:, r], src[:, r], b)
                    count("move"); count("valu_med3")
                w = new
                d >>= 1
        tail(kpl >> 1)
        s <<= 1
    return w


BUCKET_LIMIT = 16        # sot_wave_sort.hpp: kWaveSortBucketLimit


def bucket_sort32(w, n, stats=None, arrival=None):
    """The DISTRIBUTION form of the sort for 32 keys per lane (round 6, second form; sot_wave_sort.hpp: wsort_bucket_sort32): the packed words' top
    11 bits -- 2048 equal bins over the row's range -- are a bucket number; a histogram (LDS atomics), an exclusive scan (lane l owns buckets
    32 l ... 32 l + 31) and a scatter put every word into its bucket's position range, in whatever order the atomics arrived (`arrival`: a
    permutation the test varies); two passes of 32-register sorts on windows [32 l, 32 l + 32) and [32 l + 16, 32 l + 48) then order every
    bucket of at most BUCKET_LIMIT words completely (a bucket lies inside a window of one of the two passes).  Returns the sorted image in the
    blocked layout [lane, r], or None when a bucket is over the limit (the caller runs the network on the untouched words)."""
    words = w.reshape(-1).copy()                    # register order is irrelevant to the result: every word carries its own index
    real = words != 0xFFFFFFFF
    assert int(real.sum()) == n
    buckets = (words >> 21).astype(np.int64)
    counts = np.bincount(buckets[real], minlength=2048)
    if counts.max() > BUCKET_LIMIT:
        return None
    base = np.concatenate([[0], np.cumsum(counts)[:-1]])
    image = np.full(2048, 0xFFFFFFFF, np.uint32)    # pads keep positions n ... 2047
    order = np.arange(len(words)) if arrival is None else arrival
    fill = np.zeros(2048, np.int64)
    for i in order:                                 # the atomics' return values: arrival order inside a bucket
        if real[i]:
            b = buckets[i]
            image[base[b] + fill[b]] = words[i]
            fill[b] += 1
    lane = np.arange(64)
    for j in range(32):                             # the two window reads / writes touch 32 banks per half wave on the skewed image
        for addr in (33 * lane + j, 33 * lane[:63] + 16 + j + (1 if j >= 16 else 0)):
            half = addr[:32] % 32
            assert len(set(half.tolist())) == len(half)
    for start in (0, 16):                           # pass 1: windows [32 l, 32 l + 32); pass 2: [32 l + 16, 32 l + 48), lane 63 idle
        for l in range(64 if start == 0 else 63):
            lo = 32 * l + start
            image[lo:lo + 32] = np.sort(image[lo:lo + 32])
        if stats is not None:
            stats["valu_inreg"] = stats.get("valu_inreg", 0) + 2 * 191
    assert np.all(image[:-1] <= image[1:]), "two window passes did not finish the buckets"
    return image.reshape(64, 32)


def collide(a, b, idxbits):
    """two neighbouring sorted words share q (identical words = two pads: not a collision): the kernel's (a ^ b) - 1 < 2^idxbits - 1"""
    return ((int(a) ^ int(b)) - 1) & 0xFFFFFFFF < (1 << idxbits) - 1


def fix_runs(words, keys_nat, idxbits, limit=RUN_LIMIT):
    """exact order inside runs of equal q; words: sorted, natural order.  Returns False when a run is longer than `limit`."""
    n_all = len(words)
    mask = (1 << idxbits) - 1
    ob = order_bits(keys_nat)
    p = 0
    while p + 1 < n_all:
        if not collide(words[p], words[p + 1], idxbits):
            p += 1
            continue
        e = p + 1
        while e + 1 < n_all and collide(words[e + 1], words[p], idxbits):
            e += 1
        ln = e - p + 1
        if ln > limit:
            return False
        for i in range(1, ln):             # insertion sort, strict '>' : stable w.r.t. the index order the network left
            wi = int(words[p + i]); ki = int(ob[wi & mask])
            j = i
            while j > 0:
                wj = int(words[p + j - 1])
                if int(ob[wj & mask]) > ki:
                    words[p + j] = wj
                    j -= 1
                else:
                    break
            words[p + j] = wi
        p = e + 1
    return True


def skew(p):
    return p + (p >> 5)


def wave_sort_model(keys, kpl, stats=None, buckets=True, arrival=None):
    """(sorted_keys[n], indices[n]) or None when the fast path declines (caller falls back to the merge sort)."""
    n = len(keys)
    npad = 64 * kpl
    assert 1 <= n <= npad
    idxbits = 6 + int(np.log2(kpl))
    nat = np.full(npad, np.inf, np.float32)
    nat[:n] = keys
    words, ok = pack_words(nat, n, kpl)
    if not ok:
        return None
    # load order e = r 64 + lane -> register image [lane, r]
    w = words.reshape(kpl, 64).T.copy()
    sorted_image = bucket_sort32(w, n, stats, arrival) if (kpl == 32 and buckets) else None
    w = sorted_image if sorted_image is not None else network(w, kpl, stats)
    # blocked -> striped through the skewed scratch image: write position p = lane kpl + r at skew(p), read p' = r 64 + lane at skew(p')
    scratch = np.zeros(npad + (npad >> 5) + 2, np.uint32)
    lane = np.arange(64)
    for r in range(kpl):
        addr = skew(lane * kpl + r)
        assert len(set((addr[:32] % 32).tolist())) == 32 and len(set((addr[32:] % 32).tolist())) == 32   # conflict-free ds_write_b32
        scratch[addr] = w[:, r]
    nat_words = np.empty(npad, np.uint32)
    for r in range(kpl):
        addr = skew(r * 64 + lane)
        assert len(set((addr[:32] % 32).tolist())) == 32 and len(set((addr[32:] % 32).tolist())) == 32
        nat_words[r * 64 + lane] = scratch[addr]
    assert np.all(nat_words[:-1] <= nat_words[1:]), "the network did not sort the words"
    x = (nat_words[:-1] ^ nat_words[1:]).astype(np.int64)
    if (((x - 1) & 0xFFFFFFFF) < (1 << idxbits) - 1).any():
        if not fix_runs(nat_words, nat, idxbits):
            return None
    idx = (nat_words & ((1 << idxbits) - 1)).astype(np.int64)
    return nat[idx][:n], idx[:n]
