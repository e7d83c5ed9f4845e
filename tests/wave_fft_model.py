"""CPU model of the one-wavefront batched FFT of csrc/sot_mss.hip (round 5): 64 lanes x 16 complex registers = 1024 packed points =
F = 1024 / m frames of m = 2^M points each (n_fft = 2 m = 64 ... 2048).  The model executes the kernel's data flow lane by lane --
register layouts, radix-4 / radix-2 decimation-in-frequency stages on the bits a lane holds, exchanges through the wave's LDS buffer
with the kernel's address maps -- and counts LDS bank conflicts with the rules of MI355X_MICROARCH.md (ds_write_b64: groups of 16
contiguous lanes, 32 banks; ds_read_b64: two halves of 32 lanes, 64 banks).  tests/test_wave_fft_model.py checks it against numpy's FFT;
the HIP code follows the same functions (same names) with the layouts as compile-time constants.

Vocabulary: `pos` = the 10-bit slot number of a point inside the wave: bits [M, 10) = frame j, bits [0, M) = the point's in-place position
i (time index before the transform, bit-reversed frequency after it).  A LAYOUT says which pos bit each of the 4 register-index bits and 6
lane bits carries."""
import numpy as np

LANES, REGS = 64, 16


def bitrev(v, bits):
    r = 0
    for b in range(bits):
        if v >> b & 1:
            r |= 1 << (bits - 1 - b)
    return r


class Layout:
    def __init__(self, regbits, lanebits):
        assert len(regbits) == 4 and len(lanebits) == 6 and sorted(regbits + lanebits) == list(range(10))
        self.regbits, self.lanebits = list(regbits), list(lanebits)

    def pos(self, lane, reg):
        p = 0
        for t, b in enumerate(self.regbits):
            p |= (reg >> t & 1) << b
        for t, b in enumerate(self.lanebits):
            p |= (lane >> t & 1) << b
        return p

    def lane_part(self, lane):
        return self.pos(lane, 0)

    def reg_part(self, reg):
        return self.pos(0, reg)


class Geometry:
    """Phases (the bits processed in registers, high to low) and layouts of the transform of 2^M points per frame."""

    def __init__(self, M):
        assert 5 <= M <= 10
        self.M, self.m, self.J = M, 1 << M, 10 - M
        self.F, self.L = 1 << (10 - M), (1 << M) // 16          # frames per wave, lanes per frame
        bits = list(range(M - 1, -1, -1))
        self.phase_bits = [p for p in (bits[0:4], bits[4:8], bits[8:]) if p]
        frame = list(range(M, 10))
        # phase 1 (the load layout): registers = the 4 top bits of i, lanes = (frame | low bits of i): point i = L * reg + l, coalesced loads
        lay = [Layout(list(range(M - 4, M)), list(range(0, M - 4)) + frame)]
        if M >= 9:   # a middle phase: registers = the next 4 bits; lanes = (low bits | the 4 processed bits | frame)
            lay.append(Layout(list(range(M - 8, M - 4)), list(range(0, M - 8)) + list(range(M - 4, M)) + frame))
        # last phase: registers = pos bits 3..0 (processed here, or already processed); lanes = (k's low bits | frame): lane bit t carries pos
        # bit M-1-t = frequency bit t, so that a register holds L consecutive frequencies per frame
        lay.append(Layout([0, 1, 2, 3], [M - 1 - t for t in range(M - 4)] + frame))
        self.layouts = lay
        assert len(self.layouts) == len(self.phase_bits)

    # address maps (in units of one complex point = 8 bytes)
    @staticmethod
    def addr_mid(pos, strides=None):
        """exchange between two phases: one pad slot per 16 points and one more per 512 (found by search: the only two-term padding that
        keeps all 16-lane store groups and 32-lane load halves of every exchange of every size on distinct banks); 1088 points"""
        return pos + (pos >> 4) + (pos >> 9)

    def addr_nat(self, j, k):
        """frame-major natural frequency order with L pad slots per frame: F * (m + L) = 1088 points"""
        return j * (self.m + self.L) + k


def stage_plan(phase_bits):
    """radix-4 on (b, b-1) pairs from the top; a last single bit is a radix-2 stage"""
    out, i = [], 0
    while i < len(phase_bits):
        if i + 1 < len(phase_bits):
            out.append((phase_bits[i], phase_bits[i + 1]))
            i += 2
        else:
            out.append((phase_bits[i],))
            i += 1
    return out


def tw_addr(beta, mult, lam):
    """Slot of W_{2^(beta+1)}^(mult * lam) in the COMPACT twiddle table of csrc/sot_wave_fft.hpp (round 5, second form): one block of
    3 * 2^(beta-1) entries per radix-4 stage position beta >= 2 -- [mult - 1][lam], lam < 2^(beta-1) -- so that the lanes of a wave, which
    differ in the LOW bits of lam, read consecutive slots (the strided single table W_1024^t had up to 8 lanes per bank)."""
    half = 1 << (beta - 1)
    assert beta >= 2 and 1 <= mult <= 3 and 0 <= lam < half
    return 3 * (half - 2) + (mult - 1) * half + lam


TW_ENTRIES = 3 * ((1 << 9) - 2)    # beta = 2 ... 9


def tw_table():
    tab = np.zeros(TW_ENTRIES, complex)
    for beta in range(2, 10):
        for mult in (1, 2, 3):
            for lam in range(1 << (beta - 1)):
                idx = mult * (lam << (9 - beta))          # the exponent on the 1024-point circle (what build_tables derives the entry from)
                assert idx < 768
                tab[tw_addr(beta, mult, lam)] = np.exp(-2j * np.pi * idx / 1024.0)
    return tab


_TW = tw_table()


def twiddle(M, beta, lam, mult, inverse=False):
    """W_{2^(beta+1)}^(mult * lam) from the compact table (radix-4 stages: lam < 2^(beta-1)); beta = 1 and the radix-2 stage on bit 0 have lam = 0."""
    if lam == 0:
        w = 1.0 + 0j
    else:
        w = _TW[tw_addr(beta, mult, lam)]
        assert abs(w - np.exp(-2j * np.pi * mult * lam / float(1 << (beta + 1)))) < 1e-15
    return np.conj(w) if inverse else w


def twiddle_read_cycles(geo):
    """(LDS cycles, reads) of the twiddle loads of one transform: per radix-4 butterfly three ds_read_b64 of the wave at
    tw_addr(beta, mult, lam(lane, base)); 2 cycles per read = conflict-free (Lds.read's bank model)."""
    table = Lds(TW_ENTRIES)
    for layout, bits in zip(geo.layouts, geo.phase_bits):
        for st in stage_plan(bits):
            if len(st) != 2:
                continue
            tbit = {b: layout.regbits.index(b) for b in st}
            beta, low = st[0], st[-1]
            for base in range(REGS):
                if any(base >> tbit[b] & 1 for b in st):
                    continue
                lams = [(layout.pos(lane, base) & ((1 << geo.M) - 1)) & ((1 << low) - 1) for lane in range(LANES)]
                if beta < 2 or all(v == 0 for v in lams):
                    continue                                # trivial butterfly: no loads
                for mult in (1, 2, 3):
                    table.read([tw_addr(beta, mult, v) for v in lams])
    return table.read_cycles, table.read_ops


def run_phase(geo, data, layout, bits, inverse=False):
    """In-register stages of one phase.  data[lane][reg].  Forward: decimation in frequency; inverse: the transposed network (see
    inverse_transform), which runs the stages of a phase in REVERSE order with conjugate twiddles applied BEFORE the butterfly."""
    M = geo.M
    plan = stage_plan(bits)
    if inverse:
        plan = plan[::-1]
    for st in plan:
        tbit = {b: layout.regbits.index(b) for b in st}          # register-index bit of each processed pos bit
        low = st[-1]
        for lane in range(LANES):
            for base in range(REGS):
                if any(base >> tbit[b] & 1 for b in st):
                    continue
                lam = (layout.pos(lane, base) & ((1 << M) - 1)) & ((1 << low) - 1)      # i mod 2^low: the bits below the stage
                if len(st) == 2:
                    beta = st[0]
                    r = [base | (p >> 1 & 1) << tbit[st[0]] | (p & 1) << tbit[st[1]] for p in range(4)]   # p = 2 b_beta + b_(beta-1)
                    a = [data[lane][x] for x in r]
                    w1, w2, w3 = (twiddle(M, beta, lam, q, inverse) for q in (1, 2, 3))
                    if not inverse:
                        s02, d02, s13, d13 = a[0] + a[2], a[0] - a[2], a[1] + a[3], a[1] - a[3]
                        v = [s02 + s13, (s02 - s13) * w2, (d02 - 1j * d13) * w1, (d02 + 1j * d13) * w3]
                    else:
                        v0, v1, v2, v3 = a[0], a[1] * w2, a[2] * w1, a[3] * w3
                        s01, d01, s23, d23 = v0 + v1, v0 - v1, v2 + v3, v2 - v3
                        v = [s01 + s23, d01 + 1j * d23, s01 - s23, d01 - 1j * d23]
                    for x, val in zip(r, v):
                        data[lane][x] = val
                else:
                    beta = st[0]
                    r0, r1 = base, base | 1 << tbit[beta]
                    a, b = data[lane][r0], data[lane][r1]
                    w = twiddle(M, beta, lam, 1, inverse)
                    if not inverse:
                        data[lane][r0], data[lane][r1] = a + b, (a - b) * w
                    else:
                        data[lane][r0], data[lane][r1] = a + b * w, a - b * w


class Lds:
    """One wave's exchange buffer with conflict accounting."""

    def __init__(self, size=1088):
        self.mem = np.zeros(size, complex)
        self.size = size
        self.write_cycles = self.write_ops = self.read_cycles = self.read_ops = 0

    def write(self, addrs, vals):          # one ds_write_b64 of the wave: addrs[lane] (None = lane masked off)
        for g in range(4):
            banks = {}
            for lane in range(16 * g, 16 * g + 16):
                if addrs[lane] is not None:
                    banks.setdefault(addrs[lane] % 16, set()).add(addrs[lane])
            self.write_cycles += max([len(s) for s in banks.values()] or [1])
        self.write_ops += 1
        for lane in range(LANES):
            if addrs[lane] is not None:
                assert 0 <= addrs[lane] < self.size
                self.mem[addrs[lane]] = vals[lane]

    def read(self, addrs):                 # one ds_read_b64
        for g in range(2):
            banks = {}
            for lane in range(32 * g, 32 * g + 32):
                if addrs[lane] is not None:
                    banks.setdefault(addrs[lane] % 32, set()).add(addrs[lane])
            self.read_cycles += max([len(s) for s in banks.values()] or [1])
        self.read_ops += 1
        return [self.mem[a] if a is not None else 0.0 for a in addrs]


def exchange(geo, lds, data, lay_from, lay_to):
    for reg in range(REGS):
        lds.write([geo.addr_mid(lay_from.pos(lane, reg)) for lane in range(LANES)], [data[lane][reg] for lane in range(LANES)])
    out = [[0j] * REGS for _ in range(LANES)]
    for reg in range(REGS):
        vals = lds.read([geo.addr_mid(lay_to.pos(lane, reg)) for lane in range(LANES)])
        for lane in range(LANES):
            out[lane][reg] = vals[lane]
    return out


def load_frames(geo, frames):
    """frames[j][i] (complex packed points) -> phase-1 registers: lane (j, l) holds i = L * reg + l"""
    lay = geo.layouts[0]
    data = [[0j] * REGS for _ in range(LANES)]
    for lane in range(LANES):
        for reg in range(REGS):
            p = lay.pos(lane, reg)
            data[lane][reg] = frames[p >> geo.M][p & (geo.m - 1)]
    return data


def forward_transform(geo, lds, data):
    """phase-1 registers -> last-phase registers holding the m-point DFT of every frame at bit-reversed positions"""
    for p, bits in enumerate(geo.phase_bits):
        if p:
            data = exchange(geo, lds, data, geo.layouts[p - 1], geo.layouts[p])
        run_phase(geo, data, geo.layouts[p], bits)
    return data


def freq_of(geo, lane, reg):
    """(frame, frequency) held by (lane, reg) in the last layout"""
    p = geo.layouts[-1].pos(lane, reg)
    return p >> geo.M, bitrev(p & (geo.m - 1), geo.M)


def write_natural(geo, lds, data):
    for reg in range(REGS):
        lds.write([geo.addr_nat(*freq_of(geo, lane, reg)) for lane in range(LANES)], [data[lane][reg] for lane in range(LANES)])


def pair_indices(geo, lane, q):
    """the q-th bin pair (k, m - k) of a lane, q = 0 .. 8: k = L q + l for q < 8 (k < m / 2), and k = m / 2 for l = 0 at q = 8"""
    j, l = lane // geo.L, lane % geo.L
    if q < 8:
        return j, geo.L * q + l
    return (j, geo.m // 2) if l == 0 else None


def read_pairs(geo, lds):
    """-> pairs[lane][q] = (Z_k, Z_(m-k) (index taken mod m)) or None"""
    out = [[None] * 9 for _ in range(LANES)]
    for q in range(9):
        idx = [pair_indices(geo, lane, q) for lane in range(LANES)]
        zk = lds.read([geo.addr_nat(*x) if x else None for x in idx])
        zm = lds.read([geo.addr_nat(x[0], (geo.m - x[1]) % geo.m) if x else None for x in idx])
        for lane in range(LANES):
            if idx[lane]:
                out[lane][q] = (zk[lane], zm[lane])
    return out


def unpack_pair(geo, k, zk, zm):
    """bins k and m - k of the real frame of n = 2 m samples from the packed transform (csrc/sot_stft.hip: unpack_pair)"""
    ze = 0.5 * (zk + np.conj(zm))
    zo = -0.5j * (zk - np.conj(zm))
    wz = np.exp(-2j * np.pi * k / (2 * geo.m)) * zo
    return ze + wz, np.conj(ze - wz)


def pack_gradient_pair(geo, k, hk, hm):
    """G_k, G_(m-k) of the Hermitian spectrum H (H_k for 0 < k < m already halved, H_0 and H_m real) -- csrc/sot_stft.hip: backward"""
    w = np.exp(-2j * np.pi * k / (2 * geo.m))
    s, d = hk + np.conj(hm), hk - np.conj(hm)
    return s + 1j * np.conj(w) * d, np.conj(s) + 1j * w * np.conj(d)


def write_gradient_pairs(geo, lds, pairs):
    """pairs[lane][q] = (G_k, G_(m-k)) -> natural order (k = 0 and k = m / 2 are their own partners: one store)"""
    for q in range(9):
        idx = [pair_indices(geo, lane, q) for lane in range(LANES)]
        lds.write([geo.addr_nat(*x) if x else None for x in idx], [pairs[lane][q][0] if idx[lane] else 0 for lane in range(LANES)])
        second = [x if (x and 0 < x[1] < geo.m - x[1]) else None for x in idx]
        lds.write([geo.addr_nat(x[0], geo.m - x[1]) if x else None for x in second], [pairs[lane][q][1] if second[lane] else 0 for lane in range(LANES)])


def inverse_transform(geo, lds, lds_nat_loaded=True):
    """natural-order G in the buffer -> phase-1 registers holding the unnormalised inverse m-point transform (time order: lane (j, l), reg r
    = point L r + l).  The transposed network of forward_transform with conjugate twiddles: phases and stages in reverse order."""
    last = geo.layouts[-1]
    data = [[0j] * REGS for _ in range(LANES)]
    for reg in range(REGS):
        vals = lds.read([geo.addr_nat(*freq_of(geo, lane, reg)) for lane in range(LANES)])
        for lane in range(LANES):
            data[lane][reg] = vals[lane]
    for p in range(len(geo.phase_bits) - 1, -1, -1):
        run_phase(geo, data, geo.layouts[p], geo.phase_bits[p], inverse=True)
        if p:
            data = exchange(geo, lds, data, geo.layouts[p], geo.layouts[p - 1])
    return data
