"""CPU model of the in-LDS merge sort of csrc/sot_device.hpp (merge_sort_kv, round 4): the same block sort, merge-path bisection and
8-element merges, thread by thread, in numpy -- the index algebra the kernel relies on (stable order, in-place rounds, pads behind the
data) checked against numpy's stable argsort.  The kernel itself is checked on the GPU against torch.sort (tests/test_gpu_parity.py)."""
import numpy as np
import pytest


def merge_sort_kv_model(key, idx, npad):
    nv = npad >> 3
    for v in range(nv):                                   # phase 0: odd-even transposition on blocks of 8 (strict '>' : stable)
        k, x = list(key[8 * v:8 * v + 8]), list(idx[8 * v:8 * v + 8])
        for p in range(8):
            for e in range(p & 1, 7, 2):
                if k[e] > k[e + 1]:
                    k[e], k[e + 1] = k[e + 1], k[e]
                    x[e], x[e + 1] = x[e + 1], x[e]
        key[8 * v:8 * v + 8], idx[8 * v:8 * v + 8] = k, x
    run, log_run = 8, 3
    while run < npad:
        out_k, out_i = np.empty_like(key), np.empty_like(idx)
        for v in range(nv):                               # every "thread" reads the OLD arrays, writes after the barrier
            o = 8 * v
            a0 = o & ~(2 * run - 1)
            b0, d = a0 + run, o - a0
            lo, hi = max(0, d - run), min(d, run)
            for _ in range(log_run + 1):                  # fixed trip count, predicated updates
                if lo < hi:
                    mid = (lo + hi) >> 1
                    if key[a0 + mid] <= key[b0 + d - 1 - mid]:
                        lo = mid + 1
                    else:
                        hi = mid
            assert lo >= hi
            ia, ib = lo, d - lo
            ka, kb = key[a0 + min(ia, run - 1)], key[b0 + min(ib, run - 1)]
            xa, xb = idx[a0 + min(ia, run - 1)], idx[b0 + min(ib, run - 1)]
            for e in range(8):
                take_a = ia < run and (ib >= run or ka <= kb)
                out_k[o + e], out_i[o + e] = (ka, xa) if take_a else (kb, xb)
                if take_a:
                    ia += 1
                    ka, xa = key[a0 + min(ia, run - 1)], idx[a0 + min(ia, run - 1)]
                else:
                    ib += 1
                    kb, xb = key[b0 + min(ib, run - 1)], idx[b0 + min(ib, run - 1)]
        key[:], idx[:] = out_k, out_i
        run, log_run = run << 1, log_run + 1


@pytest.mark.parametrize("seed", range(40))
def test_merge_sort_model_is_a_stable_sort(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 700))
    npad = max(8, 1 << int(np.ceil(np.log2(n))))
    vals = rng.random(n).astype(np.float32)
    if seed % 3 == 0:
        vals = (np.round(vals * 8) / 8).astype(np.float32)      # many ties
    if seed % 7 == 0:
        vals[rng.integers(0, n)] = np.inf                       # a real +inf ties with the pads
    if seed % 11 == 0:
        vals = np.sort(vals)[::-1].copy()                       # descending input
    key = np.full(npad, np.inf, np.float32)
    key[:n] = vals
    idx = np.full(npad, 2 ** 31 - 1, np.int64)
    idx[:n] = np.arange(n)
    merge_sort_kv_model(key, idx, npad)
    order = np.argsort(vals, kind="stable")
    assert np.array_equal(idx[:n], order)
    assert np.array_equal(key[:n], vals[order])
    assert np.all(idx[n:] == 2 ** 31 - 1)


# ---------------------------------------------------------------------------------------------------------------------------------
# Model of the 16-elements-per-thread merge sort (merge_sort16_kv, round 4): ordered-integer keys, a skewed LDS image with one spare
# slot behind every block of 16 (conflict-free block accesses) that holds the run's SENTINEL or a COPY of the next block's first
# element, so that the sequential merge reads "the element after the one just consumed" without bounds tests.
# ---------------------------------------------------------------------------------------------------------------------------------
def batcher_network(n):
    """Comparators of Batcher's odd-even merge sort for n = 2^k inputs (63 for 16)."""
    def merge(lo, hi, r):
        step = r * 2
        if step < hi - lo:
            yield from merge(lo, hi, step)
            yield from merge(lo + r, hi, step)
            yield from [(i, i + r) for i in range(lo + r, hi - r, step)]
        else:
            yield (lo, lo + r)

    def sort(lo, hi):
        if hi - lo >= 1:
            mid = lo + (hi - lo) // 2
            yield from sort(lo, mid)
            yield from sort(mid + 1, hi)
            yield from merge(lo, hi, 1)
    return list(sort(0, n - 1))


def float_order_bits(v):
    u = np.asarray(v, np.float32).view(np.uint32).astype(np.uint64)
    u = np.where(u == 0x80000000, 0, u)                      # -0.0 sorts with +0.0
    return np.where(u & 0x80000000, u ^ 0xFFFFFFFF, u ^ 0x80000000).astype(np.uint64)


SENT = 0xFFFFFFFF


def sk(i):
    return i + (i >> 4)


def sk_next(i):       # i >= 1: the read that follows the consumption of element i - 1 of the same run
    return i + ((i - 1) >> 4)


def merge_sort16_model(vals, n, npad):
    """Returns (sorted order bits, indices) in natural layout; vals: float32[n]."""
    assert npad >= 16 and npad & (npad - 1) == 0
    nb = npad >> 4
    cap = npad + nb
    key = np.zeros(cap, np.uint64)
    idx = np.zeros(cap, np.int64)
    net = batcher_network(16)
    assert len(net) == 63
    raw = np.full(npad, np.inf, np.float32)
    raw[:n] = vals
    for b in range(nb):                                       # phase 0: block sort in "registers" by (key, index)
        k = [int(float_order_bits(raw[16 * b + e])) for e in range(16)]
        x = [16 * b + e if 16 * b + e < n else 2 ** 31 - 1 for e in range(16)]
        for i, j in net:
            if (k[i], x[i]) > (k[j], x[j]):
                k[i], k[j], x[i], x[j] = k[j], k[i], x[j], x[i]
        for e in range(16):
            key[17 * b + e], idx[17 * b + e] = k[e], x[e]
        key[17 * b + 16] = SENT                               # every block is a run: its spare slot is the sentinel
    run, log_run = 16, 4
    while run < npad:
        outs = []
        for b in range(nb):                                   # all reads of the round come before its writes (barrier)
            o = 16 * b
            a0 = o & ~(2 * run - 1)
            b0, d = a0 + run, o - a0
            lo, hi = max(0, d - run), min(d, run)
            for _ in range(log_run + 1):
                if lo < hi:
                    mid = (lo + hi) >> 1
                    if key[sk(a0 + mid)] <= key[sk(b0 + d - 1 - mid)]:
                        lo = mid + 1
                    else:
                        hi = mid
            assert lo >= hi
            ia, ib = a0 + lo, b0 + d - lo                      # global indices of the two heads
            pa = sk_next(ia) if lo > 0 else sk(ia)
            pb = sk_next(ib) if d - lo > 0 else sk(ib)
            ka, xa, kb, xb = key[pa], idx[pa], key[pb], idx[pb]
            ok, oi = [], []
            for e in range(16):
                take_a = ka <= kb
                assert (ka if take_a else kb) != SENT
                ok.append(ka if take_a else kb)
                oi.append(xa if take_a else xb)
                if take_a:
                    ia += 1
                    ka, xa = key[sk_next(ia)], idx[sk_next(ia)]
                else:
                    ib += 1
                    kb, xb = key[sk_next(ib)], idx[sk_next(ib)]
            outs.append((ok, oi))
        for b, (ok, oi) in enumerate(outs):
            o = 16 * b
            for e in range(16):
                key[17 * b + e], idx[17 * b + e] = ok[e], oi[e]
            if o % (2 * run) != 0:                            # not the first block of its output run: the copy for the block in front
                key[17 * b - 1], idx[17 * b - 1] = ok[0], oi[0]
            if (o + 16) % (2 * run) == 0:                     # last block of its output run: the sentinel
                key[17 * b + 16] = SENT
        run, log_run = run << 1, log_run + 1
    nat_k = np.array([key[sk(i)] for i in range(npad)], np.uint64)
    nat_i = np.array([idx[sk(i)] for i in range(npad)], np.int64)
    return nat_k, nat_i


@pytest.mark.parametrize("seed", range(40))
def test_merge_sort16_model_is_a_stable_sort(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 900))
    npad = max(16, 1 << int(np.ceil(np.log2(n))))
    vals = (rng.random(n).astype(np.float32) - np.float32(0.3))   # negative keys too
    if seed % 3 == 0:
        vals = (np.round(vals * 8) / 8).astype(np.float32)       # many ties, +-0
    if seed % 7 == 0:
        vals[rng.integers(0, n)] = np.inf                        # a real +inf ties with the pads
    if seed % 11 == 0:
        vals = np.sort(vals)[::-1].copy()
    k, i = merge_sort16_model(vals, n, npad)
    order = np.argsort(vals, kind="stable")
    assert np.array_equal(i[:n], order)
    assert np.array_equal(k[:n], float_order_bits(vals[order]))
    assert np.all(i[n:] == 2 ** 31 - 1)
