// sot_wave_fft.hpp -- the one-wavefront batched FFT of round 5 (gfx950): 64 lanes x 16 complex registers = 1024 packed points = F = 1024 / m
// frames of m = 2^M points each (n_fft = 2 m = 64 ... 2048): radix-4 / radix-2 stages on the position bits a lane holds in its register index,
// one or two exchanges through the wave's own LDS buffer (1088 points) with padded, conflict-free, additive address maps, and the transposed
// network for the inverse.  The index algebra is modelled lane by lane in tests/wave_fft_model.py and checked against numpy's FFT
// (tests/test_wave_fft_model.py).  Used by csrc/sot_mss.hip (MSSLoss in two launches) and csrc/sot_stft.hip (n_fft 2048 producer kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef WFFT_DIAG_TW_NOLANE
#define WFFT_DIAG_TW_NOLANE 0
#endif

namespace sot_wfft {

typedef float v2f __attribute__((ext_vector_type(2)));   // one complex point; arithmetic maps to v_pk_*_f32

constexpr int kBuf = 1088;            // complex points of one wave's exchange buffer: 1024 + pads (both address maps)
constexpr int kTw = 1530;             // table 1: the stage twiddles, one compact block per radix-4 stage position (tw_block)
constexpr int kWnMax = 520;           // table 2: -i W_2048^k / 2, k <= 512 (scale M reads index k << (10 - M))

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// complex products on the packed-fp32 unit: one v_pk_mul_f32 + one v_pk_fma_f32 (operand halves picked by op_sel, signs by neg_*).
// (csrc/sot_stft.hip keeps the three-rounding form for the SOT chain's knife edge; nothing here has one, and the fused form is the more accurate.)
#ifndef MSS_CMUL_3OP
#define MSS_CMUL_3OP 0    /* diagnostic: 1 = the three-rounding products of csrc/sot_stft.hip */
#endif
__device__ __forceinline__ v2f cmul(v2f a, v2f b)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(b));          // (-a.y b.y, a.y b.x)
#if MSS_CMUL_3OP
    return a.xx * b + t;
#else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));       // (a.x b.x, a.x b.y) + t
    return r;
#endif
}
__device__ __forceinline__ v2f cmul_conj(v2f a, v2f b)   // a * conj(b)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));                                   // (a.y b.y, a.y b.x)
#if MSS_CMUL_3OP
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));                      // (a.x b.x, -a.x b.y)
    return r + t;
#else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));    // (a.x b.x, -a.x b.y) + t
    return r;
#endif
}
__device__ __forceinline__ v2f add_mi(v2f a, v2f b)      // a - i b
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f add_pi(v2f a, v2f b)      // a + i b
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f cconj(v2f a) { return (v2f){a.x, -a.y}; }
__device__ __forceinline__ v2f mul_i(v2f a) { return (v2f){-a.y, a.x}; }
__device__ __forceinline__ v2f mul_mi(v2f a) { return (v2f){a.y, -a.x}; }
template <bool INV>
__device__ __forceinline__ v2f ctw(v2f a, v2f w) { return INV ? cmul_conj(a, w) : cmul(a, w); }

// ---------------------------------------------------------------------------------------------
// Geometry of the transform of m = 2^M packed points per frame on one wavefront (tests/wave_fft_model.py: class Geometry).
// `pos` = 10-bit slot of a point in the wave: bits [M, 10) = frame, bits [0, M) = in-place position (time index before, bit-reversed
// frequency after).  Phase ph processes up to four position bits held in the REGISTER index: phase 0 = the top four (load layout: point
// i = L reg + l, coalesced), a middle phase for M >= 9, and a last phase with registers = position bits 3..0 whose lanes carry the low
// frequency bits (a register then holds L consecutive frequencies of each frame).
// ---------------------------------------------------------------------------------------------
template <int M>
struct Geo {
    static constexpr int m = 1 << M, n = 2 * m, L = m / 16, logL = M - 4, F = 1024 / m, nph = (M <= 8) ? 2 : 3, nb = m + 1;
    static constexpr int regbit(int ph, int t) { return ph == 0 ? M - 4 + t : (ph == nph - 1 ? t : M - 8 + t); }
    static constexpr int pos_reg(int ph, int reg)
    {
        int p = 0;
        for (int t = 0; t < 4; ++t) p |= ((reg >> t) & 1) << regbit(ph, t);
        return p;
    }
    static constexpr int tbit(int ph, int posbit)
    {
        for (int t = 0; t < 4; ++t)
            if (regbit(ph, t) == posbit) return t;
        return -1;
    }
    static constexpr int reg_posmask(int ph) { return pos_reg(ph, 15); }
    static constexpr int phase_hi(int ph) { return M - 1 - 4 * ph; }
    static constexpr int phase_cnt(int ph) { return (M - 4 * ph) >= 4 ? 4 : (M - 4 * ph); }
};

template <int M, int PH>
__device__ __forceinline__ int pos_lane(int lane)
{
    using G = Geo<M>;
    if constexpr (PH == 0) return (lane & (G::L - 1)) | ((lane >> G::logL) << M);
    else if constexpr (PH == G::nph - 1) return ((int)(__brev((unsigned)(lane & (G::L - 1))) >> (32 - G::logL)) << 4) | ((lane >> G::logL) << M);
    else return (lane & ((1 << (M - 8)) - 1)) | ((lane >> (M - 8)) << (M - 4));
}

__host__ __device__ constexpr int addr_mid(int pos) { return pos + (pos >> 4) + (pos >> 9); }   // exchange between phases (conflict-free: model)

// ---- in-register stages -------------------------------------------------------------------------------------------------------------
// radix-4 on position bits (BETA, BETA - 1), both carried by the register index in phase PH.  Register p = 2 b_BETA + b_(BETA-1) of each
// group of four; lam = the position bits below the stage; twiddles W_(2^(BETA+1))^(q lam) = tw[q * (lam << (9 - BETA))] (index < 768).
// first entry of the table block of the radix-4 stage on position bits (BETA, BETA - 1), BETA >= 2: blocks of 3 * 2^(BETA-1) entries in
// ascending BETA (tests/wave_fft_model.py: tw_addr); BETA = 1 has no twiddle (q = 0)
__host__ __device__ constexpr int tw_block(int beta) { return beta >= 2 ? 3 * ((1 << (beta - 1)) - 2) : 0; }

// Forward (decimation in frequency): v0 = s02 + s13, v1 = (s02 - s13) w2, v2 = (d02 - i d13) w1, v3 = (d02 + i d13) w3.
// INV: the transposed butterfly with conjugate twiddles applied first (the inverse network runs the forward one backwards).
template <int M, int PH, int BETA, bool INV>
__device__ __forceinline__ void radix4_stage(v2f (&r)[16], const v2f* tw, int lanepos)
{
    using G = Geo<M>;
    constexpr int th = G::tbit(PH, BETA), tl = G::tbit(PH, BETA - 1);
    static_assert(th >= 0 && tl >= 0, "stage bits must be register bits");
    constexpr int mask = (1 << (BETA - 1)) - 1, half = 1 << (BETA - 1);
    constexpr bool lane_low = ((~G::reg_posmask(PH)) & mask) != 0;       // some of the bits below the stage are lane bits
    // q = the position bits below the stage; the stage's block of the table holds W^q, W^2q, W^3q (W = the 2^(BETA+1)-th root) as three runs
    // of `half` consecutive entries: the lanes of a wave differ in the LOW bits of q and read consecutive slots -- no bank conflicts (the
    // single strided table W_1024^(q << (9 - BETA)) of the first form had up to 8 lanes per bank: 35 % of the MSS kernel's LDS cycles)
#if WFFT_DIAG_TW_NOLANE   /* diagnostic only (wrong values): lane-independent twiddle addresses, to price the reads' bank conflicts */
    const int ll = 0;
#else
    const int ll = lane_low ? (lanepos & mask) : 0;
#endif
    const v2f* const t1 = tw + tw_block(BETA) + ll;
#pragma unroll
    for (int base = 0; base < 16; ++base) {
        if (((base >> th) & 1) || ((base >> tl) & 1)) continue;
        const int o = G::pos_reg(PH, base) & mask;                        // compile-time after unrolling
        const int i0 = base, i1 = base | (1 << tl), i2 = base | (1 << th), i3 = base | (1 << th) | (1 << tl);
        const bool trivial = !lane_low && o == 0;
        if (!INV) {
            const v2f s02 = r[i0] + r[i2], d02 = r[i0] - r[i2], s13 = r[i1] + r[i3], d13 = r[i1] - r[i3];
            r[i0] = s02 + s13;
            const v2f v1 = s02 - s13, v2 = add_mi(d02, d13), v3 = add_pi(d02, d13);
            if (trivial) { r[i1] = v1; r[i2] = v2; r[i3] = v3; }
            else { r[i1] = cmul(v1, t1[half + o]); r[i2] = cmul(v2, t1[o]); r[i3] = cmul(v3, t1[2 * half + o]); }
        } else {
            v2f v1 = r[i1], v2 = r[i2], v3 = r[i3];
            if (!trivial) { v1 = cmul_conj(v1, t1[half + o]); v2 = cmul_conj(v2, t1[o]); v3 = cmul_conj(v3, t1[2 * half + o]); }
            const v2f s01 = r[i0] + v1, d01 = r[i0] - v1, s23 = v2 + v3, d23 = v2 - v3;
            r[i0] = s01 + s23; r[i2] = s01 - s23;
            r[i1] = add_pi(d01, d23); r[i3] = add_mi(d01, d23);
        }
    }
}

// radix-2 on position bit 0 (the only single bit any size ends with): no twiddle, its own transpose
template <int M, int PH>
__device__ __forceinline__ void radix2_stage(v2f (&r)[16])
{
    constexpr int t0 = Geo<M>::tbit(PH, 0);
    static_assert(t0 >= 0, "bit 0 must be a register bit");
#pragma unroll
    for (int base = 0; base < 16; ++base) {
        if ((base >> t0) & 1) continue;
        const v2f a = r[base], b = r[base | (1 << t0)];
        r[base] = a + b; r[base | (1 << t0)] = a - b;
    }
}

template <int M, int PH, bool INV>
__device__ __forceinline__ void run_phase(v2f (&r)[16], const v2f* tw, int lane)
{
    using G = Geo<M>;
    constexpr int hi = G::phase_hi(PH), cnt = G::phase_cnt(PH);
    const int lp = pos_lane<M, PH>(lane);
    if (!INV) {
        if constexpr (cnt >= 2) radix4_stage<M, PH, hi, false>(r, tw, lp);
        if constexpr (cnt == 4) radix4_stage<M, PH, hi - 2, false>(r, tw, lp);
        if constexpr (cnt == 3 || cnt == 1) radix2_stage<M, PH>(r);
    } else {
        if constexpr (cnt == 3 || cnt == 1) radix2_stage<M, PH>(r);
        if constexpr (cnt == 4) radix4_stage<M, PH, hi - 2, true>(r, tw, lp);
        if constexpr (cnt >= 2) radix4_stage<M, PH, hi, true>(r, tw, lp);
    }
}

// registers of layout FROM -> registers of layout TO through the wave's buffer (16 ds_write_b64 + 16 ds_read_b64, immediate offsets)
template <int M, int FROM, int TO>
__device__ __forceinline__ void exchange(v2f (&r)[16], v2f* zl, int lane)
{
    using G = Geo<M>;
    v2f* const wp = zl + addr_mid(pos_lane<M, FROM>(lane));
#pragma unroll
    for (int q = 0; q < 16; ++q) wp[addr_mid(G::pos_reg(FROM, q))] = r[q];
    wave_sync();
    const v2f* const rp = zl + addr_mid(pos_lane<M, TO>(lane));
#pragma unroll
    for (int q = 0; q < 16; ++q) r[q] = rp[addr_mid(G::pos_reg(TO, q))];
    wave_sync();
}

// phase-0 registers (time order) -> last-phase registers: the m-point DFT of every frame at bit-reversed positions
template <int M>
__device__ __forceinline__ void forward_transform(v2f (&r)[16], v2f* zl, const v2f* tw, int lane)
{
    using G = Geo<M>;
    run_phase<M, 0, false>(r, tw, lane);
    exchange<M, 0, 1>(r, zl, lane);
    run_phase<M, 1, false>(r, tw, lane);
    if constexpr (G::nph == 3) {
        exchange<M, 1, 2>(r, zl, lane);
        run_phase<M, 2, false>(r, tw, lane);
    }
}

// last-phase registers -> phase-0 registers: the unnormalised INVERSE transform (transposed network, conjugate twiddles)
template <int M>
__device__ __forceinline__ void inverse_transform(v2f (&r)[16], v2f* zl, const v2f* tw, int lane)
{
    using G = Geo<M>;
    if constexpr (G::nph == 3) {
        run_phase<M, 2, true>(r, tw, lane);
        exchange<M, 2, 1>(r, zl, lane);
    }
    run_phase<M, 1, true>(r, tw, lane);
    exchange<M, 1, 0>(r, zl, lane);
    run_phase<M, 0, true>(r, tw, lane);
}

__host__ __device__ constexpr int brev4(int v) { return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3); }


// a + conj(b), a - conj(b), conj(a - b): one packed add each (negated halves)
__device__ __forceinline__ v2f add_conj(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f sub_conj(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f conj_sub(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}


// last-phase registers -> the wave's buffer in natural frequency order, frame-major with L pad slots per frame (address j (m + L) + k);
// Z_0 is stored a second time at slot m, where the partner read of bin 0 looks for "Z_m"
template <int M>
__device__ __forceinline__ void write_natural(const v2f (&r)[16], v2f* zl, int lane)
{
    using G = Geo<M>;
    const int j = lane >> G::logL, kl = lane & (G::L - 1);
    v2f* const p = zl + j * (G::m + G::L) + kl;
#pragma unroll
    for (int q = 0; q < 16; ++q) p[G::L * brev4(q)] = r[q];
    if (kl == 0) p[G::m] = r[0];
}

// the two tables from W_4096^j, j <= 1024 (csrc/sot_stft_tables.inc: kWn), by all THREADS threads of the workgroup; the caller synchronises.
// Every load of a thread is issued before the first is used (compile-time trip counts): ONE round trip to L2 -- a plain strided loop waits
// per element, 9 serial round trips for a 256-thread workgroup (~5 us at the head of every workgroup of a short kernel).
template <int THREADS>
__device__ __forceinline__ void build_tables(const float2* __restrict__ w4096, v2f* tw, v2f* wn)
{
    constexpr int NT = (kTw + THREADS - 1) / THREADS, NW = (513 + THREADS - 1) / THREADS;
    float2 a[NT], b[NW];
    int quarter[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        // slot t = tw_block(beta) + (mult - 1) half + q, half = 2^(beta-1): W_(4 half)^(mult q) = W_1024^e, e = mult q (256 / half) < 768;
        // W_1024^(256 a + b) = W_4096^(4 b) (-i)^a: exact quarter turns of the committed table (the values of the first form's table)
        const int t = min((int)threadIdx.x + i * THREADS, kTw - 1);
        const int half = 1 << (31 - __builtin_clz((unsigned)(t / 3 + 2)));
        const int rem = t - 3 * (half - 2);
        const int mult = rem / half + 1, q = rem & (half - 1);
        const int e = mult * q * (256 / half);
        quarter[i] = e >> 8;
        a[i] = w4096[4 * (e & 255)];
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) b[i] = w4096[2 * min((int)threadIdx.x + i * THREADS, 512)];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int t = (int)threadIdx.x + i * THREADS;
        v2f w = (v2f){a[i].x, a[i].y};
        if (quarter[i] == 1) w = mul_mi(w); else if (quarter[i] == 2) w = -w;
        if (t < kTw) tw[t] = w;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int k = (int)threadIdx.x + i * THREADS;
        if (k <= 512) wn[k] = (v2f){0.5f * b[i].y, -0.5f * b[i].x};   // -i W_2048^k / 2
    }
}

}  // namespace sot_wfft
