"""The one-wavefront batched FFT of csrc/sot_mss.hip as a CPU model (tests/wave_fft_model.py): layouts, stages, exchanges and address maps
against numpy's FFT for every transform size of MSSLoss (n_fft 64 ... 2048), plus the LDS bank-conflict count of every exchange."""
import numpy as np
import pytest

import wave_fft_model as wm


def random_frames(geo, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((geo.F, geo.m)) + 1j * rng.standard_normal((geo.F, geo.m))


@pytest.mark.parametrize("M", range(5, 11))
def test_forward_is_the_dft_at_bit_reversed_positions(M):
    geo = wm.Geometry(M)
    frames = random_frames(geo, M)
    lds = wm.Lds()
    data = wm.forward_transform(geo, lds, wm.load_frames(geo, frames))
    want = np.fft.fft(frames, axis=1)
    for lane in range(wm.LANES):
        for reg in range(wm.REGS):
            j, k = wm.freq_of(geo, lane, reg)
            assert abs(data[lane][reg] - want[j][k]) < 1e-9 * geo.m
    # every exchange is conflict-free: 16 stores + 16 loads per exchange, one LDS cycle per lane group
    assert lds.write_cycles == 4 * lds.write_ops and lds.read_cycles == 2 * lds.read_ops
    assert lds.write_ops == 16 * (len(geo.phase_bits) - 1)


@pytest.mark.parametrize("M", range(5, 11))
def test_real_frame_bins_from_the_packed_transform(M):
    geo = wm.Geometry(M)
    rng = np.random.default_rng(100 + M)
    real = rng.standard_normal((geo.F, 2 * geo.m))
    frames = real[:, 0::2] + 1j * real[:, 1::2]
    lds = wm.Lds()
    data = wm.forward_transform(geo, lds, wm.load_frames(geo, frames))
    w0, r0 = lds.write_cycles, lds.read_cycles
    wo, ro = lds.write_ops, lds.read_ops
    wm.write_natural(geo, lds, data)
    pairs = wm.read_pairs(geo, lds)
    assert lds.write_cycles - w0 == 4 * (lds.write_ops - wo), "natural-order stores conflict"
    assert lds.read_cycles - r0 == 2 * (lds.read_ops - ro), "pair loads conflict"
    want = np.fft.rfft(real, axis=1)
    seen = np.zeros((geo.F, geo.m + 1), int)
    for lane in range(wm.LANES):
        for q in range(9):
            idx = wm.pair_indices(geo, lane, q)
            if idx is None:
                assert pairs[lane][q] is None
                continue
            j, k = idx
            xk, xm = wm.unpack_pair(geo, k, *pairs[lane][q])
            assert abs(xk - want[j][k]) < 1e-9 * geo.m and abs(xm - want[j][geo.m - k]) < 1e-9 * geo.m
            seen[j][k] += 1
            if geo.m - k != k:
                seen[j][geo.m - k] += 1
    assert (seen == 1).all()      # every bin 0 .. m of every frame is produced by exactly one lane


@pytest.mark.parametrize("M", range(5, 11))
def test_inverse_network_returns_the_frame_gradient_in_load_order(M):
    """Hermitian spectrum -> packed G -> transposed network = irfft (unnormalised), in the phase-1 (coalesced) layout."""
    geo = wm.Geometry(M)
    rng = np.random.default_rng(200 + M)
    n = 2 * geo.m
    zin = rng.standard_normal((geo.F, geo.m + 1)) + 1j * rng.standard_normal((geo.F, geo.m + 1))
    lds = wm.Lds()
    pairs = [[None] * 9 for _ in range(wm.LANES)]
    for lane in range(wm.LANES):
        for q in range(9):
            idx = wm.pair_indices(geo, lane, q)
            if idx is None:
                continue
            j, k = idx
            hk, hm = 0.5 * zin[j][k], 0.5 * zin[j][geo.m - k]
            if k == 0:
                hk, hm = zin[j][0].real + 0j, zin[j][geo.m].real + 0j
            pairs[lane][q] = wm.pack_gradient_pair(geo, k, hk, hm)
    wm.write_gradient_pairs(geo, lds, pairs)
    w0, wo = lds.write_cycles, lds.write_ops
    data = wm.inverse_transform(geo, lds)
    # the inverse runs the exchanges backwards (stores in the later layout, loads in the earlier one): conflict-free up to n_fft 512;
    # at 1024 / 2048 one store group and one load half per exchange go two-way (no padding of the form pos + c1 (pos >> a1) + c2 (pos >> a2)
    # with any lane-bit order of the middle layout serves all four directions: searched) -- ~100 of ~1300 LDS cycles of a wave task
    slack = 1 if M <= 8 else 2
    assert lds.read_cycles <= slack * 2 * lds.read_ops, "inverse-side loads conflict"
    assert lds.write_cycles - w0 <= slack * 4 * (lds.write_ops - wo)
    # y_i = Re(sum_{k=0}^{m} Zin_k e^{+2 pi i k i / n}): samples 2 i, 2 i + 1 are the real and imaginary part of packed point i
    t = np.arange(n)
    kk = np.arange(geo.m + 1)
    lay = geo.layouts[0]
    for j in range(geo.F):
        y = np.real(np.exp(2j * np.pi * np.outer(t, kk) / n) @ zin[j])
        for lane in range(wm.LANES):
            for reg in range(wm.REGS):
                p = lay.pos(lane, reg)
                if p >> geo.M != j:
                    continue
                i = p & (geo.m - 1)
                assert i == geo.L * reg + lane % geo.L
                assert abs(data[lane][reg].real - y[2 * i]) < 1e-8 * n and abs(data[lane][reg].imag - y[2 * i + 1]) < 1e-8 * n


@pytest.mark.parametrize("M", range(5, 11))
def test_addresses_are_lane_part_plus_register_part(M):
    """Every LDS address of the kernel is (a value computed once per lane) + (a compile-time constant per register): immediate offsets."""
    geo = wm.Geometry(M)
    for lay in geo.layouts:
        for lane in range(wm.LANES):
            for reg in range(wm.REGS):
                assert geo.addr_mid(lay.pos(lane, reg)) == geo.addr_mid(lay.lane_part(lane)) + geo.addr_mid(lay.reg_part(reg)) - geo.addr_mid(0)
    for lane in range(wm.LANES):
        j0, k0 = wm.freq_of(geo, lane, 0)
        for reg in range(wm.REGS):
            j, k = wm.freq_of(geo, 0, reg)
            jj, kk = wm.freq_of(geo, lane, reg)
            assert geo.addr_nat(jj, kk) == geo.addr_nat(j0, k0) + geo.addr_nat(j, k)


@pytest.mark.parametrize("M", range(5, 11))
def test_twiddle_loads_are_conflict_free(M):
    """The compact per-stage table: the lanes of a wave read consecutive (or equal) slots -- two LDS cycles per ds_read_b64, the minimum."""
    cycles, reads = wm.twiddle_read_cycles(wm.Geometry(M))
    assert reads > 0 and cycles == 2 * reads


def test_compact_twiddle_table_layout():
    seen = set()
    for beta in range(2, 10):
        for mult in (1, 2, 3):
            for lam in range(1 << (beta - 1)):
                seen.add(wm.tw_addr(beta, mult, lam))
    assert seen == set(range(wm.TW_ENTRIES)) and wm.TW_ENTRIES == 1530
