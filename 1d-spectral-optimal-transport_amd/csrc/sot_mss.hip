// sot_mss.hip -- MI355X (gfx950): the reference's multi-scale spectrogram loss `MSSLoss` (losses.py:365-425: for every FFT size the
// magnitude STFT of target and estimate -- compute_mag, features.py:191-237: hann window, 75 % overlap, end padding (utils.py:252-275),
// normalized -- and mean_difference (losses.py:7-36) of the magnitudes and / or their safe_log (utils.py:145-151), summed over the sizes)
// TOGETHER WITH ITS GRADIENT w.r.t. the estimate's audio, in TWO launches for all scales (round 4: 36 launches and ~40 B of HBM traffic per
// spectrogram bin; here no spectrogram ever leaves the chip):
//
//  mss_fused_kernel   one 512-thread workgroup per (scale, clip, chunk of 4096 samples).  Each of its 8 wavefronts takes 1024 packed
//                     points = F = 2048 / n_fft consecutive frames: both signals' frames -> window -> one-wavefront FFT (16 points per lane,
//                     radix-4 stages in registers, 1-2 exchanges through the wave's LDS buffer; index algebra: tests/wave_fft_model.py, checked
//                     against numpy) -> bins (k, m - k) of the real frames -> |T|, |V| -> distance terms (fp64 partial sums) -> the gradient
//                     w.r.t. the estimate's spectrum g_k V_k / |V_k| as a Hermitian packing, written over the spectrum in LDS -> transposed
//                     (inverse) network -> window -> the workgroup overlap-adds its 8 F frames in LDS and stores the chunk's span.
//  mss_finish_kernel  sums, per sample, the spans of all scales in a fixed order (deterministic; no atomics) and the loss partials.
//
// The gradient is that of the loss value itself (upstream gradient 1); the autograd node multiplies by the upstream scalar (or per-clip vector).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/sot_hip.h"

namespace sot_mss {

#include "sot_stft_tables.inc"      // kWn = W_4096^j, j <= 1024 (csrc/gen/make_stft_tables.py); kPassTw unused here

typedef float v2f __attribute__((ext_vector_type(2)));   // one complex point; arithmetic maps to v_pk_*_f32

constexpr int kThreads = 512, kWaves = 8;
constexpr int kBuf = 1088;            // complex points of one wave's exchange buffer: 1024 + pads (both address maps below)
constexpr int kTw = 768;              // W_1024^t, t < 768
constexpr int kWnMax = 520;           // W_n^k, k <= m / 2 <= 512
constexpr size_t kLdsBytes = ((size_t)kWaves * kBuf + kTw + kWnMax) * sizeof(float2);
constexpr int kMaxScales = 8;

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// complex products on the packed-fp32 unit (see csrc/sot_stft.hip: cmul / cmul_conj / add_mi / add_pi)
__device__ __forceinline__ v2f cmul(v2f a, v2f b)
{
    v2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(b));   // (-a.y b.y, a.y b.x)
    return a.xx * b + t;
}
__device__ __forceinline__ v2f cmul_conj(v2f a, v2f b)   // a * conj(b)
{
    v2f t1, t2;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t1) : "v"(a), "v"(b));   // (a.x b.x, -a.x b.y)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t2) : "v"(a), "v"(b));               // (a.y b.y, a.y b.x)
    return t1 + t2;
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
// Forward (decimation in frequency): v0 = s02 + s13, v1 = (s02 - s13) w2, v2 = (d02 - i d13) w1, v3 = (d02 + i d13) w3.
// INV: the transposed butterfly with conjugate twiddles applied first (the inverse network runs the forward one backwards).
template <int M, int PH, int BETA, bool INV>
__device__ __forceinline__ void radix4_stage(v2f (&r)[16], const v2f* tw, int lanepos)
{
    using G = Geo<M>;
    constexpr int th = G::tbit(PH, BETA), tl = G::tbit(PH, BETA - 1);
    static_assert(th >= 0 && tl >= 0, "stage bits must be register bits");
    constexpr int mask = (1 << (BETA - 1)) - 1, sh = 9 - BETA;
    constexpr bool lane_low = ((~G::reg_posmask(PH)) & mask) != 0;       // some of the bits below the stage are lane bits
    const int ll = lane_low ? ((lanepos & mask) << sh) : 0;
    const v2f* const t1 = tw + ll;
    const v2f* const t2 = tw + 2 * ll;
    const v2f* const t3 = tw + 3 * ll;
#pragma unroll
    for (int base = 0; base < 16; ++base) {
        if (((base >> th) & 1) || ((base >> tl) & 1)) continue;
        const int o = (G::pos_reg(PH, base) & mask) << sh;                // compile-time after unrolling
        const int i0 = base, i1 = base | (1 << tl), i2 = base | (1 << th), i3 = base | (1 << th) | (1 << tl);
        const bool trivial = !lane_low && o == 0;
        if (!INV) {
            const v2f s02 = r[i0] + r[i2], d02 = r[i0] - r[i2], s13 = r[i1] + r[i3], d13 = r[i1] - r[i3];
            r[i0] = s02 + s13;
            const v2f v1 = s02 - s13, v2 = add_mi(d02, d13), v3 = add_pi(d02, d13);
            if (trivial) { r[i1] = v1; r[i2] = v2; r[i3] = v3; }
            else { r[i1] = cmul(v1, t2[2 * o]); r[i2] = cmul(v2, t1[o]); r[i3] = cmul(v3, t3[3 * o]); }
        } else {
            v2f v1 = r[i1], v2 = r[i2], v3 = r[i3];
            if (!trivial) { v1 = cmul_conj(v1, t2[2 * o]); v2 = cmul_conj(v2, t1[o]); v3 = cmul_conj(v3, t3[3 * o]); }
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

// ---------------------------------------------------------------------------------------------
struct MssArgs {
    const float* target; const float* value;          // [batch, samples], row strides in floats
    int64_t batch, samples, stride_t, stride_v;
    int n_scales;
    int logm[kMaxScales];                             // log2(n_fft / 2) per scale
    const float* window[kMaxScales];                  // n_fft taps per scale
    int frames[kMaxScales];                           // ceil(samples / hop), hop = n_fft / 4
    int chunks[kMaxScales];                           // workgroups per clip: ceil(frames / (8 F))
    int block_base[kMaxScales + 1];                   // first workgroup of each scale
    float coef[kMaxScales];                           // d loss / d (sum of the scale's distance terms): 1 / count (all clips) or 1 / (frames bins)
    double inv_count[kMaxScales];                     // the same in double, for the loss value
    int64_t grad_base[kMaxScales];                    // offset (floats) of the scale's spans in partial_grad
    int64_t loss_base[kMaxScales];                    // offset of the scale's partial sums in partial_loss
    float mag_weight, logmag_weight, eps; int l2, per_clip, want_grad;
    double* partial_loss;                             // [scale][clip][chunk]
    float* partial_grad;                              // [scale][clip][chunk][span]: span = 4096 + 3 hop samples
    float* loss; float* grad;                         // outputs: [1] or [batch]; [batch, samples] (contiguous)
};

__device__ __forceinline__ float safe_logf(float x, float eps) { return logf(x <= eps ? eps : x); }

// |x| for a frame whose windowed samples passed the range test (csrc/sot_stft.hip: magnitude_plain / frame_is_plain)
__device__ __forceinline__ float magnitude_plain(v2f x)
{
    const float s = fmaf(x.x, x.x, x.y * x.y);
    const float r = __builtin_amdgcn_sqrtf(s);
    const float h = 0.5f * __builtin_amdgcn_rsqf(s);
    const float e = fmaf(-r, r, s);
    const float v = fmaf(e, h, r);
    return s == 0.0f ? 0.0f : v;
}
__device__ __forceinline__ bool frame_is_plain(float amax) { return (amax > 1e-9f && amax < 1e15f) || amax == 0.0f; }

__device__ __forceinline__ float wave_max_f32(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// one bin: the distance term D(t, v) (weighted) into `acc`, and d(term) / d v into the return value
__device__ __forceinline__ float bin_term(const MssArgs& a, float t, float v, double& acc)
{
    float gv = 0.0f;
    if (a.mag_weight > 0.0f) {
        const float d = t - v;
        acc += (double)a.mag_weight * (a.l2 ? (double)(d * d) : (double)fabsf(d));
        const float g = a.l2 ? 2.0f * d : (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));   // torch: sgn(0) = 0
        gv -= a.mag_weight * g;
    }
    if (a.logmag_weight > 0.0f) {
        const float d = safe_logf(t, a.eps) - safe_logf(v, a.eps);
        acc += (double)a.logmag_weight * (a.l2 ? (double)(d * d) : (double)fabsf(d));
        const float g = a.l2 ? 2.0f * d : (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
        gv -= (v <= a.eps) ? 0.0f : a.logmag_weight * g / v;    // where(x <= eps, eps, x): no gradient below eps
    }
    return gv;
}

// frames [frame0, frame0 + F) of one signal -> windowed packed points in phase-0 registers; returns the largest |sample * tap| of the lane
template <int M>
__device__ __forceinline__ float load_frames(const float* __restrict__ clip, int64_t samples, int frames, int frame0, const float2* __restrict__ win,
                                             int lane, v2f (&r)[16])
{
    using G = Geo<M>;
    constexpr int hop = G::n / 4;
    const int j = lane >> G::logL, l = lane & (G::L - 1);
    const int f = frame0 + j;
    const int64_t t0 = (int64_t)f * hop;
    const float* const s0 = clip + t0;
    // the wave's fast path: every frame it owns exists and lies inside the clip, 8-byte aligned
    const int64_t last = ((int64_t)frame0 + G::F - 1) * hop + G::n;
    const bool fast = frame0 + G::F <= frames && last <= samples && (reinterpret_cast<uintptr_t>(clip) & 7u) == 0;
    float amax = 0.0f;
    if (fast) {
        const float2* const s2 = reinterpret_cast<const float2*>(s0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float2 v = s2[G::L * q + l], w = win[G::L * q + l];
            r[q] = (v2f){v.x * w.x, v.y * w.y};
            amax = fmaxf(amax, fmaxf(fabsf(r[q].x), fabsf(r[q].y)));
        }
    } else {
        const int64_t left = (f < frames) ? samples - t0 : 0;     // samples of the frame that exist: zeros beyond (utils.py:252-275)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = G::L * q + l;
            const float2 w = win[i];
            const float v0 = (2 * i < left) ? s0[2 * i] * w.x : 0.0f;
            const float v1 = (2 * i + 1 < left) ? s0[2 * i + 1] * w.y : 0.0f;
            r[q] = (v2f){v0, v1};
            amax = fmaxf(amax, fmaxf(fabsf(v0), fabsf(v1)));
        }
    }
    return amax;
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

// bins k and m - k of the real frame from the packed transform (csrc/sot_stft.hip: unpack_pair); w = W_n^k
__device__ __forceinline__ void unpack_pair(v2f zk, v2f zm, v2f w, v2f& xk, v2f& xm)
{
    const v2f ze = 0.5f * (zk + cconj(zm));
    const v2f zo = 0.5f * mul_mi(zk - cconj(zm));
    const v2f wz = cmul(w, zo);
    xk = ze + wz;
    xm = cconj(ze - wz);
}

// The bin pairs of the lane: q < 8: k = L q + l (all lanes), q = 8: k = m / 2 (lane l = 0 of each frame).  Pass 1 (target): |T| of both
// bins into tm[].  Pass 2 (estimate): |V|, distance terms, and -- GRAD -- the Hermitian packing G of the gradient w.r.t. the spectrum written
// over Z in the buffer (the two slots of a pair are read and written by the same lane only).
template <int M, bool PLAIN, bool SECOND, bool GRAD>
__device__ __forceinline__ void pair_pass(const MssArgs& a, float coef, v2f* zl, const v2f* wn, int lane, bool active, float (&tm)[18], double& acc)
{
    using G = Geo<M>;
    const int j = lane >> G::logL, l = lane & (G::L - 1);
    v2f* const pk = zl + j * (G::m + G::L) + l;         // + L q
    v2f* const pm = zl + j * (G::m + G::L) - l;         // + L (16 - q)
    const float scale = 1.0f / sqrtf((float)G::n);      // normalized=True: frame_length^-0.5
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        if (q == 8 && l != 0) break;
        const int k = (q < 8) ? G::L * q + l : G::m / 2;
        const v2f zk = (q < 8) ? pk[G::L * q] : pk[G::m / 2], zm = (q < 8) ? pm[G::L * (16 - q)] : zk;
        v2f xk, xm;
        unpack_pair(zk, zm, wn[k], xk, xm);
        float mk, mm;
        if (PLAIN) { mk = magnitude_plain(xk); mm = magnitude_plain(xm); }
        else { mk = hypotf(xk.x, xk.y); mm = hypotf(xm.x, xm.y); }
        if (!SECOND) { tm[2 * q] = mk * scale; tm[2 * q + 1] = mm * scale; continue; }
        const bool both = q < 8;                                   // k = m / 2 (q = 8) is its own partner: one bin; k = 0 pairs with bin m
        float gk = 0.0f, gm = 0.0f;
        if (active) {
            gk = bin_term(a, tm[2 * q], mk * scale, acc);
            if (both) gm = bin_term(a, tm[2 * q + 1], mm * scale, acc);
        }
        if (GRAD) {
            // Zin_k = g_k X_k / |X_k| (torch: sgn(0) = 0); H_k = Zin_k / 2 (0 < k < m), H_0 = Re Zin_0, H_m = Re Zin_m;
            // G_k = (H_k + conj H_(m-k)) + i conj(W) (H_k - conj H_(m-k)),  G_(m-k) = conj(s) + i W conj(d)   (csrc/sot_stft.hip, backward)
            float ck, cm;
            if (PLAIN) {
                const float sk2 = fmaf(xk.x, xk.x, xk.y * xk.y), sm2 = fmaf(xm.x, xm.x, xm.y * xm.y);
                ck = sk2 > 0.0f ? (gk * coef) * __builtin_amdgcn_rsqf(sk2) : 0.0f;
                cm = sm2 > 0.0f ? (gm * coef) * __builtin_amdgcn_rsqf(sm2) : 0.0f;
            } else {
                ck = mk > 0.0f ? (gk * coef) / mk : 0.0f;
                cm = mm > 0.0f ? (gm * coef) / mm : 0.0f;
            }
            v2f hk = (0.5f * ck) * xk, hm = (0.5f * cm) * xm;
            if (k == 0) { hk = (v2f){ck * xk.x, 0.0f}; hm = (v2f){cm * xm.x, 0.0f}; }
            if (q == 8) hm = hk;                                   // k = m / 2: H_(m-k) is H_k itself
            const v2f sk = hk + cconj(hm), dk = hk - cconj(hm);
            const v2f w = wn[k];
            const v2f g_k = sk + mul_i(cmul(cconj(w), dk));
            if (q < 8) {
                pk[G::L * q] = g_k;
                if (k != 0) pm[G::L * (16 - q)] = cconj(sk) + mul_i(cmul(w, cconj(dk)));
            } else {
                pk[G::m / 2] = g_k;
            }
        }
    }
}

template <int M, bool GRAD>
__device__ __forceinline__ void scale_body(const MssArgs& a, int s, int unit)
{
    using G = Geo<M>;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    v2f* const tw = reinterpret_cast<v2f*>(smem_f);      // tables first: every lane-part address into a wave's buffer stays positive
    v2f* const wn = tw + kTw;
    v2f* const bufs = wn + kWnMax;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    v2f* const zl = bufs + wave * kBuf;
    const int chunks = a.chunks[s], frames = a.frames[s];
    const int b = unit / chunks, c = unit - b * chunks;
    const float* const tclip = a.target + (int64_t)b * a.stride_t;
    const float* const vclip = a.value + (int64_t)b * a.stride_v;
    const float2* const win = reinterpret_cast<const float2*>(a.window[s]);
    const int frame0 = (c * kWaves + wave) * G::F;

    // the target's frames are in flight while the tables are built
    v2f r[16];
    float amax = load_frames<M>(tclip, a.samples, frames, frame0, win, lane, r);
    for (int t = threadIdx.x; t < kTw; t += kThreads) {     // W_1024^(256 a + b) = W_4096^(4 b) (-i)^a: exact quarter turns of the committed table
        const float2 w0 = kWn[4 * (t & 255)];
        v2f w = (v2f){w0.x, w0.y};
        const int qa = t >> 8;
        if (qa == 1) w = mul_mi(w); else if (qa == 2) w = -w;
        tw[t] = w;
    }
    for (int k = threadIdx.x; k <= G::m / 2; k += kThreads) { const float2 w0 = kWn[k << (11 - M)]; wn[k] = (v2f){w0.x, w0.y}; }
    __syncthreads();

    const int jf = frame0 + (lane >> G::logL);
    const bool active = jf < frames;
    float tm[18];
    double acc = 0.0;
    const bool wave_has_frames = frame0 < frames;          // wave-uniform: the last chunk of a clip may own fewer than 8 F frames
    if (wave_has_frames) {
    // ---- target: |T| of the lane's bins
    {
        const bool plain = __builtin_amdgcn_readfirstlane((int)frame_is_plain(wave_max_f32(amax))) != 0;
        forward_transform<M>(r, zl, tw, lane);
        write_natural<M>(r, zl, lane);
        wave_sync();
        if (plain) pair_pass<M, true, false, false>(a, 0.0f, zl, wn, lane, active, tm, acc);
        else pair_pass<M, false, false, false>(a, 0.0f, zl, wn, lane, active, tm, acc);
        wave_sync();
    }
    // ---- estimate: |V|, distance, gradient packing
    amax = load_frames<M>(vclip, a.samples, frames, frame0, win, lane, r);
    {
        const bool plain = __builtin_amdgcn_readfirstlane((int)frame_is_plain(wave_max_f32(amax))) != 0;
        forward_transform<M>(r, zl, tw, lane);
        write_natural<M>(r, zl, lane);
        wave_sync();
        const float coef = a.coef[s];
        if (plain) pair_pass<M, true, true, GRAD>(a, coef, zl, wn, lane, active, tm, acc);
        else pair_pass<M, false, true, GRAD>(a, coef, zl, wn, lane, active, tm, acc);
        wave_sync();
    }
    if (GRAD) {
        // ---- inverse: G (natural order) -> last-phase registers -> time order; windowed, scaled frame gradients back into the buffer
        const int j = lane >> G::logL, kl = lane & (G::L - 1);
        const v2f* const p = zl + j * (G::m + G::L) + kl;
#pragma unroll
        for (int q = 0; q < 16; ++q) r[q] = p[G::L * brev4(q)];
        wave_sync();
        inverse_transform<M>(r, zl, tw, lane);
        const float scale = 1.0f / sqrtf((float)G::n);
        v2f* const o = zl + j * (G::m + G::L) + kl;          // packed point i = L q + l of frame j at j (m + L) + i
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float2 w = win[G::L * q + kl];
            o[G::L * q] = (v2f){w.x * r[q].x * scale, w.y * r[q].y * scale};
        }
    }
    }   // wave_has_frames
    // ---- the workgroup's distance sum: lanes -> wave (fixed shuffle tree) -> workgroup (wave order)
    double* const red = reinterpret_cast<double*>(wn);     // the W_n table is dead once every wave has finished its pair passes (barrier below)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    __syncthreads();
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < kWaves; ++w) tot += red[w];
        a.partial_loss[a.loss_base[s] + unit] = tot;
    }
    if (GRAD) {
        // overlap-add of the chunk's 8 F frames (frame fc starts at packed point fc m / 4) in ascending frame order, span = 2048 + 3 m / 4 points
        constexpr int hp = G::m / 4, span = 2048 + 3 * hp;
        const int nfr = min(kWaves * G::F, frames - c * kWaves * G::F);
        float2* const dst = reinterpret_cast<float2*>(a.partial_grad + a.grad_base[s]) + (int64_t)unit * span;
        for (int p = threadIdx.x; p < span; p += kThreads) {
            const int f_hi = min(p / hp, nfr - 1);
            int f_lo = (p - (G::m - 1) + hp - 1) / hp;
            if (p - (G::m - 1) <= 0) f_lo = 0;
            v2f sum = (v2f){0.0f, 0.0f};
            for (int f = f_lo; f <= f_hi; ++f)
                sum += bufs[(f >> (10 - M)) * kBuf + (f & (G::F - 1)) * (G::m + G::L) + (p - f * hp)];
            dst[p] = make_float2(sum.x, sum.y);
        }
    }
}

template <bool GRAD>
__global__ __launch_bounds__(kThreads) void mss_fused_kernel(const MssArgs a)
{
    int s = 0;
    const int blk = blockIdx.x;
    while (s + 1 < a.n_scales && blk >= a.block_base[s + 1]) ++s;
    const int unit = blk - a.block_base[s];
    switch (a.logm[s]) {
        case 5: scale_body<5, GRAD>(a, s, unit); break;
        case 6: scale_body<6, GRAD>(a, s, unit); break;
        case 7: scale_body<7, GRAD>(a, s, unit); break;
        case 8: scale_body<8, GRAD>(a, s, unit); break;
        case 9: scale_body<9, GRAD>(a, s, unit); break;
        default: scale_body<10, GRAD>(a, s, unit); break;
    }
}

// Finish: workgroup 0 also turns the partial sums into the loss (per scale: fixed-order sum, mean as float32, `loss += mean` in the
// reference's scale order, losses.py:411-424); every workgroup sums the spans covering its samples, scales in order.
constexpr int kFinishThreads = 256;
__global__ __launch_bounds__(kFinishThreads) void mss_finish_kernel(const MssArgs a)
{
    __shared__ double red[kFinishThreads];
    if (a.per_clip) {        // one value per clip: a thread per clip, its chunks in order
        for (int64_t o = (int64_t)blockIdx.x * kFinishThreads + threadIdx.x; o < a.batch; o += (int64_t)gridDim.x * kFinishThreads) {
            float total = 0.0f;
            for (int s = 0; s < a.n_scales; ++s) {
                double acc = 0.0;
                for (int c = 0; c < a.chunks[s]; ++c) acc += a.partial_loss[a.loss_base[s] + o * a.chunks[s] + c];
                total += (float)(acc * a.inv_count[s]);
            }
            a.loss[o] = total;
        }
    } else if (blockIdx.x == 0) {
        float total = 0.0f;
        for (int s = 0; s < a.n_scales; ++s) {
            const int64_t count = a.batch * a.chunks[s];
            double acc = 0.0;
            for (int64_t i = threadIdx.x; i < count; i += kFinishThreads) acc += a.partial_loss[a.loss_base[s] + i];
            red[threadIdx.x] = acc;
            __syncthreads();
            for (int off = kFinishThreads / 2; off > 0; off >>= 1) {
                if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
                __syncthreads();
            }
            total += (float)(red[0] * a.inv_count[s]);
            __syncthreads();
        }
        if (threadIdx.x == 0) a.loss[0] = total;
    }
    if (!a.want_grad) return;
    const int64_t half = (a.samples + 1) / 2;                   // packed points per clip
    const int64_t total = a.batch * half;
    for (int64_t idx = (int64_t)blockIdx.x * kFinishThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kFinishThreads) {
        const int64_t b = idx / half;
        const int p = (int)(idx - b * half);                    // packed point of the clip: samples 2 p, 2 p + 1
        float gx = 0.0f, gy = 0.0f;
        for (int s = 0; s < a.n_scales; ++s) {
            const int hp = (1 << a.logm[s]) / 4, span = 2048 + 3 * hp;
            const int c = p >> 11, pp = p & 2047;
            const float2* const base = reinterpret_cast<const float2*>(a.partial_grad + a.grad_base[s]) + (b * a.chunks[s]) * (int64_t)span;
            if (c > 0 && pp < 3 * hp) { const float2 v = base[(int64_t)(c - 1) * span + 2048 + pp]; gx += v.x; gy += v.y; }   // the previous chunk's tail
            if (c < a.chunks[s]) { const float2 v = base[(int64_t)c * span + pp]; gx += v.x; gy += v.y; }
        }
        float* const dst = a.grad + b * a.samples + 2 * (int64_t)p;
        dst[0] = gx;
        if (2 * (int64_t)p + 1 < a.samples) dst[1] = gy;
    }
}

static int fill(const float* target, int64_t stride_t, const float* value, int64_t stride_v, int64_t batch, int64_t samples, const int* fft_sizes,
                const float* const* windows, int n_scales, float mag_weight, float logmag_weight, float eps, int l2, int per_clip, MssArgs* a,
                size_t* workspace_bytes, size_t* grad_offset_bytes)
{
    if (batch < 0 || samples < 1 || n_scales < 1 || n_scales > kMaxScales || stride_t < samples || stride_v < samples) return SOT_ERR_BAD_SHAPE;
    if (samples > (1LL << 30)) return SOT_ERR_UNSUPPORTED_SIZE;
    a->target = target; a->value = value; a->batch = batch; a->samples = samples; a->stride_t = stride_t; a->stride_v = stride_v;
    a->n_scales = n_scales; a->mag_weight = mag_weight; a->logmag_weight = logmag_weight; a->eps = eps; a->l2 = l2; a->per_clip = per_clip;
    int64_t blocks = 0, gfloats = 0, ldoubles = 0;
    for (int s = 0; s < n_scales; ++s) {
        const int n = fft_sizes[s];
        int logn = 0;
        while ((1 << logn) < n) ++logn;
        if ((1 << logn) != n || n < 64 || n > 2048) return SOT_ERR_UNSUPPORTED_SIZE;     // 64 ... 2048, hop = n / 4
        if (windows != nullptr && (windows[s] == nullptr || reinterpret_cast<uintptr_t>(windows[s]) % 8 != 0)) return SOT_ERR_BAD_SHAPE;
        const int m = n / 2, hop = n / 4, F = 1024 / m;
        a->logm[s] = logn - 1;
        a->window[s] = windows ? windows[s] : nullptr;
        const int64_t frames = (samples + hop - 1) / hop;        // utils.py:265
        a->frames[s] = (int)frames;
        a->chunks[s] = (int)((frames + 8 * F - 1) / (8 * F));
        a->block_base[s] = (int)blocks;
        blocks += batch * a->chunks[s];
        const double count = (double)frames * (double)(m + 1) * (per_clip ? 1.0 : (double)batch);
        a->inv_count[s] = 1.0 / count;
        a->coef[s] = (float)(1.0 / count);
        a->grad_base[s] = gfloats;
        a->loss_base[s] = ldoubles;
        gfloats += 2 * batch * a->chunks[s] * (int64_t)(2048 + 3 * (m / 4));
        ldoubles += batch * a->chunks[s];
    }
    a->block_base[n_scales] = (int)blocks;
    if (blocks > 0x7fffffffLL) return SOT_ERR_UNSUPPORTED_SIZE;
    *workspace_bytes = sizeof(double) * (size_t)ldoubles + sizeof(float) * (size_t)gfloats;
    *grad_offset_bytes = sizeof(double) * (size_t)ldoubles;    // workspace = [partial sums (double) | spans (float)]
    return SOT_OK;
}

}  // namespace sot_mss

extern "C" {

size_t sot_mss_workspace_bytes(int64_t batch, int64_t samples, const int* fft_sizes, int n_scales)
{
    sot_mss::MssArgs a{};
    size_t bytes = 0, off = 0;
    if (fft_sizes == nullptr || sot_mss::fill(nullptr, samples, nullptr, samples, batch, samples, fft_sizes, nullptr, n_scales, 1.0f, 0.0f, 1e-5f, 0, 0, &a, &bytes, &off) != SOT_OK)
        return 0;
    return bytes;
}

int sot_mss_loss_and_grad(const float* target, int64_t target_row_stride, const float* value, int64_t value_row_stride, int64_t batch,
                          int64_t samples, const int* fft_sizes, const float* const* windows, int n_scales, float mag_weight,
                          float logmag_weight, float eps, int l2, int per_clip, float* loss, float* grad_value, void* workspace,
                          size_t workspace_bytes, void* stream)
{
    using namespace sot_mss;
    if (fft_sizes == nullptr || windows == nullptr) return SOT_ERR_NULL_POINTER;
    MssArgs a{};
    size_t need = 0, grad_off = 0;
    const int rc = fill(target, target_row_stride, value, value_row_stride, batch, samples, fft_sizes, windows, n_scales, mag_weight, logmag_weight,
                        eps, l2, per_clip, &a, &need, &grad_off);
    if (rc != SOT_OK) return rc;
    if (!(mag_weight > 0.0f) && !(logmag_weight > 0.0f)) return SOT_ERR_BAD_SHAPE;
    if (batch == 0) return SOT_OK;
    if (target == nullptr || value == nullptr || loss == nullptr || workspace == nullptr) return SOT_ERR_NULL_POINTER;
    if (workspace_bytes < need) return SOT_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) % 8 != 0) return SOT_ERR_BAD_SHAPE;
    a.partial_loss = reinterpret_cast<double*>(workspace);
    a.partial_grad = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + grad_off);
    a.loss = loss; a.grad = grad_value; a.want_grad = grad_value != nullptr;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    static bool attr_done[64][2] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    void (*kern)(const MssArgs) = a.want_grad ? mss_fused_kernel<true> : mss_fused_kernel<false>;
    if (dev < 0 || dev >= 64 || !attr_done[dev][a.want_grad]) {   // idempotent per device; a benign race sets it twice
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes) != hipSuccess)
            (void)hipGetLastError();
        if (dev >= 0 && dev < 64) attr_done[dev][a.want_grad] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)a.block_base[n_scales]), dim3(kThreads), kLdsBytes, st, a);
    if (hipGetLastError() != hipSuccess) return SOT_ERR_LAUNCH;
    const int64_t work = a.want_grad ? (batch * ((samples + 1) / 2) + kFinishThreads - 1) / kFinishThreads : 1;
    hipLaunchKernelGGL(mss_finish_kernel, dim3((unsigned)(work < 4096 ? (work < 1 ? 1 : work) : 4096)), dim3(kFinishThreads), 0, st, a);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

}  // extern "C"
