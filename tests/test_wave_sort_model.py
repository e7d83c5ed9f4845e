"""The one-wavefront sort of csrc/sot_wave_sort.hpp as a lane-by-lane CPU model (tests/wave_sort_model.py) against numpy's stable argsort:
the packed word, the network's partners and bounds, the skewed transposition image (its bank-conflict freedom is asserted inside the model),
the run repair, and every way the fast path has to DECLINE (the kernels then take the merge sort).  The kernels themselves are checked on
the GPU against torch.sort (tests/test_gpu_parity.py: test_segmented_sort_*, test_rowpos_presort_*)."""
import numpy as np
import pytest

from wave_sort_model import BUCKET_LIMIT, RUN_LIMIT, bucket_sort32, collide, network, order_bits, pack_words, wave_sort_model


def _check(keys, kpl, expect_decline=False):
    keys = np.asarray(keys, np.float32)
    out = wave_sort_model(keys, kpl)
    if expect_decline:
        assert out is None
        return
    assert out is not None, "the fast path declined a case it should take"
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(out[1], order)
    assert np.array_equal(out[0].view(np.uint32), keys[order].view(np.uint32))


@pytest.mark.parametrize("kpl", [1, 2, 4, 8, 16, 32])
@pytest.mark.parametrize("seed", range(4))
def test_wave_sort_model_is_the_stable_sort(kpl, seed):
    rng = np.random.default_rng(100 * kpl + seed)
    npad = 64 * kpl
    n = npad if seed == 0 else int(rng.integers(max(2, npad // 2), npad + 1))
    keys = rng.random(n).astype(np.float32)
    if seed == 2:
        keys = (rng.standard_normal(n) * 3).astype(np.float32)
    if seed == 3:
        keys = (np.round(keys * 40) / 40).astype(np.float32)        # many exact ties: runs of equal words' q, already in index order
        if np.unique(keys, return_counts=True)[1].max() > RUN_LIMIT:
            _check(keys, kpl, expect_decline=True)                   # a run longer than the limit: declined, never wrong
            return
    _check(keys, kpl)


@pytest.mark.parametrize("kpl", [8, 32])
def test_keys_inside_one_binade_interval_are_separated_by_the_adaptive_range(kpl):
    """All keys inside one 2^-14 interval: the float's own top 21 order bits would put them into ONE bin; the row-adaptive range does not."""
    rng = np.random.default_rng(kpl)
    n = 64 * kpl
    keys = (0.5 + rng.random(n) * 2.0 ** -14).astype(np.float32)
    top21 = order_bits(keys) >> 11
    assert len(np.unique(top21)) == 1
    _check(keys, kpl)


@pytest.mark.parametrize("kpl", [8, 32])
def test_runs_are_repaired_by_the_full_key(kpl):
    """Pairs and short runs that share q but differ in the key, placed in reverse order across the array (also across a lane boundary)."""
    rng = np.random.default_rng(7 + kpl)
    n = 64 * kpl
    keys = np.sort(rng.random(n).astype(np.float32))
    keys = keys[rng.permutation(n)]
    # neighbours in value that collide in q: nudge some keys to one ulp above another key
    order = np.argsort(keys)
    for j in range(0, n - 8, 37):
        a, b = order[j], order[j + 1]
        keys[max(a, b)] = keys[min(a, b)]                            # the later index ...
        keys[min(a, b)] = np.nextafter(keys[min(a, b)], np.float32(2), dtype=np.float32)   # ... gets the smaller key: the repair must swap them
    w, ok = pack_words(np.concatenate([keys, np.full(0, np.inf, np.float32)]), n, kpl)
    assert ok
    _check(keys, kpl)


@pytest.mark.parametrize("kpl", [4, 32])
def test_the_fast_path_declines_what_it_cannot_order(kpl):
    rng = np.random.default_rng(kpl)
    n = 64 * kpl - 3
    base = rng.random(n).astype(np.float32)
    cluster = (base * 1e-9).astype(np.float32); cluster[-1] = 1.0       # all but one key in ONE bin
    _check(cluster, kpl, expect_decline=True)
    for bad in (np.nan, np.inf, -np.inf):
        k = base.copy(); k[n // 3] = bad
        _check(k, kpl, expect_decline=True)
    _check(np.full(n, 0.25, np.float32), kpl, expect_decline=True)      # zero range
    _check((base - 0.5) * np.float32(6e38), kpl, expect_decline=True)    # max - min overflows
    _check(base * np.float32(1e-44), kpl, expect_decline=True)           # the scale overflows


def test_signed_zeros_and_denormals():
    rng = np.random.default_rng(3)
    keys = rng.random(512).astype(np.float32) - 0.5
    keys[[3, 77, 200, 411]] = 0.0                                        # (at most RUN_LIMIT equal keys: a longer run is declined)
    keys[[4, 90, 300]] = -0.0
    _check(keys, 8)                                                      # -0 == +0: index order among them (numpy's stable sort agrees)
    den = (rng.random(2048) * 1e-39).astype(np.float32)
    _check(den, 32, expect_decline=True)                                 # a denormal RANGE: the scale overflows
    mixed = rng.random(2048).astype(np.float32)
    mixed[[5, 600, 601, 1500, 2047]] = den[:5]
    _check(mixed, 32)                                                    # a few denormal keys inside a normal range: they share bin 0 and are repaired by the full key


def test_network_sorts_any_words_and_pads_stay_behind():
    rng = np.random.default_rng(11)
    for kpl in (2, 16):
        w = rng.integers(0, 2 ** 32 - 1, size=(64, kpl), dtype=np.uint64).astype(np.uint32)
        w[rng.integers(0, 64, 5), rng.integers(0, kpl, 5)] = 0xFFFFFFFF   # pads
        out = network(w, kpl).reshape(-1)
        assert np.array_equal(out, np.sort(w.reshape(-1)))
    assert not collide(0xFFFFFFFF, 0xFFFFFFFF, 11) and collide(0x12345000, 0x12345001, 12) and not collide(0x12345000, 0x12346000, 12)


def test_distribution_form_does_not_depend_on_the_order_the_atomics_arrive_in():
    """32 keys per lane (round 6, second form): histogram -> scan -> scatter by the top 11 bits of the packed word, then two passes of 32-register sorts on
    windows offset by 16.  The rank an LDS atomic returns is an arrival order the hardware owns; the result must not depend on it."""
    rng = np.random.default_rng(5)
    for trial in range(6):
        n = 2048 if trial < 2 else int(rng.integers(1100, 2048))
        keys = rng.random(n).astype(np.float32) if trial % 2 == 0 else (rng.standard_normal(n) * 1.5).astype(np.float32)
        want = np.argsort(keys, kind="stable")
        stats = {}
        for arrival in (None, rng.permutation(2048), np.arange(2048)[::-1]):
            out = wave_sort_model(keys, 32, stats, arrival=arrival)
            assert out is not None and np.array_equal(out[1], want)
        assert "move" not in stats                                        # the network never ran: the distribution form took these rows


def test_distribution_form_declines_an_overfull_bucket_and_the_network_takes_the_row():
    """moderately clustered positions: most keys inside 1 % of the row's range -- buckets of ~100 words (limit 16), but still one word per quantisation bin:
    the distribution form declines, the network sorts the same words, the result is the same stable order"""
    rng = np.random.default_rng(6)
    keys = np.concatenate([rng.random(2000) * 0.01, rng.random(48)]).astype(np.float32)
    keys = keys[rng.permutation(2048)]
    words, ok = pack_words(keys, 2048, 32)
    assert ok and np.bincount((words >> 21).astype(np.int64), minlength=2048).max() > BUCKET_LIMIT
    assert bucket_sort32(words.reshape(32, 64).T.copy(), 2048) is None
    stats = {}
    out = wave_sort_model(keys, 32, stats)
    assert out is not None and np.array_equal(out[1], np.argsort(keys, kind="stable")) and stats.get("move", 0) > 0   # the network ran
    same = wave_sort_model(keys, 32, buckets=False)
    assert np.array_equal(out[1], same[1])
