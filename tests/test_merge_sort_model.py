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
