// sot_mss.hip -- MI355X (gfx950): the reference's multi-scale spectrogram loss `MSSLoss` (losses.py:365-425: for every FFT size the
// magnitude STFT of target and estimate -- compute_mag, features.py:191-237: hann window, 75 % overlap, end padding (utils.py:252-275),
// normalized -- and mean_difference (losses.py:7-36) of the magnitudes and / or their safe_log (utils.py:145-151), summed over the sizes)
// TOGETHER WITH ITS GRADIENT w.r.t. the estimate's audio, in TWO launches for all scales (round 4: 36 launches and ~40 B of HBM traffic per
// spectrogram bin; here no spectrogram ever leaves the chip):
//
//  mss_fused_kernel   persistent 512-thread workgroups; every WAVEFRONT takes tasks on its own (no workgroup barrier after the twiddle tables
//                     are built): a task = 1024 packed points = F = 2048 / n_fft consecutive frames of one clip at one scale.  Both signals'
//                     frames -> window -> one-wavefront FFT (16 points per lane, radix-4 stages in registers, 1-2 exchanges through the wave's
//                     LDS buffer; index algebra: tests/wave_fft_model.py, checked against numpy) -> bins (k, m - k) of the real frames -> |T|, |V|
//                     -> distance terms -> the gradient w.r.t. the estimate's spectrum g_k V_k / |V_k| as a Hermitian packing, written over the
//                     spectrum in LDS -> transposed (inverse) network -> window -> the wave overlap-adds its F frames and stores their span
//                     (512 samples + 3 hops of tail).
//  mss_finish_kernel  sums, per sample, the spans of all scales that cover it in a fixed order (deterministic; no atomics) and the loss partials.
//
// The gradient is that of the loss value itself (upstream gradient 1); the autograd node multiplies by the upstream scalar (or per-clip vector).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/sot_hip.h"
#include "sot_wave_fft.hpp"

namespace sot_mss {

#include "sot_stft_tables.inc"      // kWn = W_4096^j, j <= 1024 (csrc/gen/make_stft_tables.py); kPassTw unused here

using namespace sot_wfft;             // the one-wavefront FFT engine (csrc/sot_wave_fft.hpp)

// workgroups of 8 waves (two per CU: 2 x 80 KB of LDS) for batches that take more than one round of the chip's 16 x CUs wave slots; of 4 waves
// (three per CU) for smaller ones -- the paper's 64 clips are 3072 tasks: 384 workgroups of 8 leave half the CUs with 16 waves and half with 8,
// 768 of 4 give every CU 12 (41.6 -> 37.9 us; 256 clips: 98.7 -> 106.9 us, so the large form stays for those)
constexpr size_t lds_bytes(int waves) { return ((size_t)waves * kBuf + kTw + kWnMax) * sizeof(float2); }
constexpr int kMaxScales = 8;
// Diagnostic build only (-DMSS_STAMPS): wave 0 of the workgroups 0, 1, 2, ... (at most 64) stamps the shader clock at its phase boundaries
// (tools/r5/mss_stamps.py reads them through sot_mss_debug_read_stamps)
#ifdef MSS_STAMPS
__device__ unsigned long long g_mss_stamps[64 * 16];
#ifndef MSS_STAMP_EVERY
#define MSS_STAMP_EVERY 1     /* sample workgroups 0, E, 2 E, ... (64 of them) */
#endif
#define MSS_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x % MSS_STAMP_EVERY == 0 && blockIdx.x / MSS_STAMP_EVERY < 64) { __builtin_amdgcn_sched_barrier(0); g_mss_stamps[(blockIdx.x / MSS_STAMP_EVERY) * 16 + (i)] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define MSS_STAMP(i) do { } while (0)
#endif
#ifndef MSS_WAVES_PER_EU
#define MSS_WAVES_PER_EU 4   /* two 512-thread workgroups per CU (LDS: 2 x 80 KB): 128 VGPRs */
#endif

// ---------------------------------------------------------------------------------------------
struct MssArgs {
    const float* target; const float* value;          // [batch, samples], row strides in floats
    int64_t batch, samples, stride_t, stride_v;
    int n_scales;
    int logm[kMaxScales];                             // log2(n_fft / 2) per scale
    const float* window[kMaxScales];                  // n_fft taps per scale
    int frames[kMaxScales];                           // ceil(samples / hop), hop = n_fft / 4
    int waves[kMaxScales];                            // wave tasks per clip: ceil(frames / F), F = 2048 / n_fft frames = 512 samples of hop positions each
    int task_base[kMaxScales + 1];                    // first task of each scale; task = task_base[s] + clip * waves[s] + w
    float coef[kMaxScales];                           // d loss / d (sum of the scale's distance terms): 1 / count (all clips) or 1 / (frames bins)
    double inv_count[kMaxScales];                     // the same in double, for the loss value
    int slot_base[kMaxScales + 1];                    // partial_grad: first 256-point slot of the scale inside a (clip, range) block; [n_scales] = slots per block
    int ranges;                                       // 256-point ranges per clip: ceil(ceil(samples / 2) / 256)
    int slot_info[32];                                // scale | piece << 8: which scale and which 256-point piece of its waves' spans a slot holds (dwords: scalar loads)
    float mag_weight, logmag_weight, eps; int l2, per_clip, want_grad;
    float post_scale;                                 // multiplies the finished float32 loss value(s) and gradient entries (a caller's `* weight`, losses.py:360): one more float32 product
    double* partial_loss;                             // [task]
    float* partial_grad;                              // [clip][range r][slot][256 points]: the wave w of a scale writes piece j of its span (256 + 3 m / 4
                                                      // packed points) into slot slot_base[s] + j of range r = w + j: all a range needs is contiguous
    float* loss; float* grad;                         // outputs: [1] or [batch]; [batch, samples] (contiguous)
};

__device__ __forceinline__ float safe_logf(float x, float eps) { return logf(x <= eps ? eps : x); }

// range test of a frame's windowed samples (csrc/sot_stft.hip: frame_is_plain): then re^2 + im^2 of its bins neither overflows nor loses the bins that matter
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

// frames [frame0, frame0 + F) of one signal: raw packed points (samples 2 i, 2 i + 1) and the window's tap pairs into registers -- loads only,
// nothing waits for them here (the estimate's frames are requested before the target's pair pass and used after it)
template <int M>
__device__ __forceinline__ void fetch_frames(const float* __restrict__ clip, int64_t samples, int frames, int frame0, const float2* __restrict__ win,
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
    if (fast) {
        const float2* const s2 = reinterpret_cast<const float2*>(s0);
#pragma unroll
        for (int q = 0; q < 16; ++q) { const float2 v = s2[G::L * q + l]; r[q] = (v2f){v.x, v.y}; }
    } else {
        const int64_t left = (f < frames) ? samples - t0 : 0;     // samples of the frame that exist: zeros beyond (utils.py:252-275)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = G::L * q + l;
            r[q] = (v2f){(2 * i < left) ? s0[2 * i] : 0.0f, (2 * i + 1 < left) ? s0[2 * i + 1] : 0.0f};
        }
    }
}

// sample * tap for BOTH signals' frames with one fetch of the taps (every wave of the scale reads the same n_fft floats: L1 / L2 hits);
// plain_t / plain_v: whether each signal's frames pass the range test (wave-uniform)
template <int M>
__device__ __forceinline__ void window_frames(v2f (&rt)[16], v2f (&rv)[16], const float2* __restrict__ win, int lane, bool& plain_t, bool& plain_v)
{
    using G = Geo<M>;
    const float2* const wl = win + (lane & (G::L - 1));
    float at = 0.0f, av = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float2 w = wl[G::L * q];
        const v2f wv = (v2f){w.x, w.y};
        rt[q] = rt[q] * wv;
        rv[q] = rv[q] * wv;
        at = fmaxf(at, fmaxf(fabsf(rt[q].x), fabsf(rt[q].y)));
        av = fmaxf(av, fmaxf(fabsf(rv[q].x), fabsf(rv[q].y)));
    }
    plain_t = __builtin_amdgcn_readfirstlane((int)frame_is_plain(wave_max_f32(at))) != 0;
    plain_v = __builtin_amdgcn_readfirstlane((int)frame_is_plain(wave_max_f32(av))) != 0;
}

// The bin pairs of the lane: q < 8: k = L q + l (all lanes), q = 8: k = m / 2 (lane l = 0 of each frame).  With the packed transform Z,
// e = Z_k + conj Z_(m-k), o = Z_k - conj Z_(m-k) and the table entry w' = -i W_n^k / 2:   X_k = e / 2 + w' o,   conj X_(m-k) = e / 2 - w' o
// (csrc/sot_stft.hip: unpack_pair, regrouped: 8 packed operations per pair).
// Pass 1 (target): |T| of both bins into tm[].  Pass 2 (estimate): |V|, the distance terms, and -- GRAD -- the Hermitian packing of the
// gradient w.r.t. the spectrum written over Z in the buffer (the two slots of a pair are read and written by the same lane only):
//   Zin_k = g_k X_k / |X_k| (torch: sgn(0) = 0),  H_k = Zin_k / 2 (0 < k < m), H_0 = Re Zin_0, H_m = Re Zin_m,
//   s = H_k + conj H_(m-k), P = i conj(W) (H_k - conj H_(m-k)):   G_k = s + P,   G_(m-k) = conj(s - P)     (csrc/sot_stft.hip, backward).
// KIND 0: the paper's distance (L1 on the magnitudes only: mag_weight |t - v|, float32 partial sums per lane); KIND 1: every combination
// (L1 / L2, magnitudes and / or their safe_log; float64 partial sums).  PLAIN: the frame passed the range test: |x| = v_sqrt(re^2 + im^2),
// 1 / |x| = v_rsq; otherwise hypotf and an IEEE division.
template <int M, bool PLAIN, bool SECOND, bool GRAD, int KIND>
__device__ __forceinline__ void pair_pass(const MssArgs& a, float coef, v2f* zl, const v2f* wn, int lane, float (&tm)[18], float& accf, double& acc)
{
    using G = Geo<M>;
    const int j = lane >> G::logL, l = lane & (G::L - 1);
    v2f* const pk = zl + j * (G::m + G::L) + l;         // + L q
    v2f* const pm = zl + j * (G::m + G::L) - l;         // + L (16 - q)
    const float scale = 1.0f / sqrtf((float)G::n);      // normalized=True: frame_length^-0.5
    const float cw = coef * a.mag_weight;
    v2f gk_out[(SECOND && GRAD) ? 9 : 1], gm_out[(SECOND && GRAD) ? 9 : 1];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        if (q == 8 && l != 0) break;
        const int k = (q < 8) ? G::L * q + l : G::m / 2;
        const v2f zk = (q < 8) ? pk[G::L * q] : pk[G::m / 2], zm = (q < 8) ? pm[G::L * (16 - q)] : zk;
        const v2f wp = wn[k << (10 - M)];
        const v2f e = add_conj(zk, zm), o = sub_conj(zk, zm);
        const v2f wz = cmul(o, wp), eh = 0.5f * e;
        const v2f xk = eh + wz, xc = eh - wz;            // X_k, conj X_(m-k)
        const float sk2 = fmaf(xk.x, xk.x, xk.y * xk.y), sm2 = fmaf(xc.x, xc.x, xc.y * xc.y);
        float mk, mm;
        if (PLAIN) { mk = __builtin_amdgcn_sqrtf(sk2); mm = __builtin_amdgcn_sqrtf(sm2); }
        else { mk = hypotf(xk.x, xk.y); mm = hypotf(xc.x, xc.y); }
        mk *= scale; mm *= scale;
        if (!SECOND) { tm[2 * q] = mk; tm[2 * q + 1] = mm; continue; }
        float gk, gm;                                     // d term / d |V| of the two bins (times coef)
        if (KIND == 0) {
            const float dk = tm[2 * q] - mk, dm = tm[2 * q + 1] - mm;
            accf += fabsf(dk);
            if (q < 8) accf += fabsf(dm);                 // k = m / 2 (q = 8) is its own partner: one bin; k = 0 pairs with bin m
            gk = dk > 0.0f ? -cw : cw; gk = dk == 0.0f ? 0.0f : gk;    // torch: sgn(0) = 0
            gm = dm > 0.0f ? -cw : cw; gm = dm == 0.0f ? 0.0f : gm;
        } else {
            gk = coef * bin_term(a, tm[2 * q], mk, acc);
            gm = (q < 8) ? coef * bin_term(a, tm[2 * q + 1], mm, acc) : 0.0f;
        }
        if (GRAD) {
            float ck, cm;
            if (PLAIN) {
                ck = sk2 > 0.0f ? gk * __builtin_amdgcn_rsqf(sk2) : 0.0f;
                cm = sm2 > 0.0f ? gm * __builtin_amdgcn_rsqf(sm2) : 0.0f;
            } else {
                ck = mk > 0.0f ? gk / (mk / scale) : 0.0f;
                cm = mm > 0.0f ? gm / (mm / scale) : 0.0f;
            }
            if (q == 0 && k == 0) { ck *= 2.0f; cm *= 2.0f; }   // H_0 and H_m are the (real) Zin themselves, not halves
            const v2f hk = ck * xk;                        // 2 H_k
            const v2f hc = (q < 8) ? cm * xc : cconj(hk);  // 2 conj H_(m-k)   (k = m / 2: H_(m-k) is H_k itself)
            const v2f sk = 0.5f * (hk + hc), dd = hk - hc;
            const v2f P = cmul_conj(dd, wp);               // conj(w') (2 d) = i conj(W) d
            gk_out[q] = sk + P;
            gm_out[q] = conj_sub(sk, P);
        }
    }
    if (SECOND && GRAD) {
        // stored after every pair has been read: the loads of all nine pairs can be in flight together (a store between two pairs' loads
        // would order them: the compiler cannot know that a pair's slots are private to its lane)
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            if (q == 8 && l != 0) break;
            if (q < 8) {
                pk[G::L * q] = gk_out[q];
                if (q != 0 || l != 0) pm[G::L * (16 - q)] = gm_out[q];
            } else {
                pk[G::m / 2] = gk_out[q];
            }
        }
    }
}

// One task of one wavefront: frames [w F, w F + F) of clip b at scale s.
template <int M, bool GRAD, int KIND>
__device__ __forceinline__ void wave_task(const MssArgs& a, int s, int task, const v2f* tw, const v2f* wn, v2f* zl)
{
    using G = Geo<M>;
    // the lane number as an opaque value per task: hipcc otherwise hoists the lane-derived LDS addresses of all six scales out of the caller's
    // loop and pays for them with hundreds of spilled registers
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int waves = a.waves[s], frames = a.frames[s];
    const int local = task - a.task_base[s];
    const int b = local / waves, w = local - b * waves;
    const float* const tclip = a.target + (int64_t)b * a.stride_t;
    const float* const vclip = a.value + (int64_t)b * a.stride_v;
    const float2* const win = reinterpret_cast<const float2*>(a.window[s]);
    const int frame0 = w * G::F;

    MSS_STAMP(1);
    float tm[18];
    float accf = 0.0f;
    double acc = 0.0;
    v2f r[16], rv[16];
    // ---- both signals' frames and the taps in one round of loads; the estimate's windowed frames wait in registers
    fetch_frames<M>(tclip, a.samples, frames, frame0, win, lane, r);
    fetch_frames<M>(vclip, a.samples, frames, frame0, win, lane, rv);
    bool plain, plain_v;
    window_frames<M>(r, rv, win, lane, plain, plain_v);
    MSS_STAMP(2);
    // ---- target: |T| of the lane's bins
    forward_transform<M>(r, zl, tw, lane);
    MSS_STAMP(3);
    write_natural<M>(r, zl, lane);
    wave_sync();
    if (plain) pair_pass<M, true, false, false, KIND>(a, 0.0f, zl, wn, lane, tm, accf, acc);
    else pair_pass<M, false, false, false, KIND>(a, 0.0f, zl, wn, lane, tm, accf, acc);
    wave_sync();
    MSS_STAMP(4);
    // ---- estimate: |V|, distance, gradient packing
    forward_transform<M>(rv, zl, tw, lane);
    MSS_STAMP(5);
    write_natural<M>(rv, zl, lane);
    wave_sync();
    const float coef = a.coef[s];
    if (plain_v) pair_pass<M, true, true, GRAD, KIND>(a, coef, zl, wn, lane, tm, accf, acc);
    else pair_pass<M, false, true, GRAD, KIND>(a, coef, zl, wn, lane, tm, accf, acc);
    wave_sync();
    MSS_STAMP(6);
    // ---- the task's distance sum: lanes -> wave (fixed shuffle tree)
    if (KIND == 0) acc = (double)accf * (double)a.mag_weight;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) a.partial_loss[task] = acc;
    if (GRAD) {
        // ---- inverse: G (natural order) -> last-phase registers -> time order; windowed, scaled frame gradients
        const int j = lane >> G::logL, kl = lane & (G::L - 1);
        const v2f* const p = zl + j * (G::m + G::L) + kl;
#pragma unroll
        for (int q = 0; q < 16; ++q) r[q] = p[G::L * brev4(q)];
        v2f wt[16];                                                // the synthesis taps arrive during the transform (tm[] is dead: the registers are free)
#pragma unroll
        for (int q = 0; q < 16; ++q) { const float2 wv = win[G::L * q + kl]; wt[q] = (v2f){wv.x, wv.y}; }
        wave_sync();
        inverse_transform<M>(r, zl, tw, lane);
        MSS_STAMP(7);
        const float scale = 1.0f / sqrtf((float)G::n);
        constexpr int hp = G::m / 4, span = 256 + 3 * hp;          // packed points: the F frames start hp apart (F hp = 256) and are m long
        // piece j = points [256 j, 256 j + 256) of the span belongs to range w + j of the clip; ranges past the clip's end are dropped
        const int slots = a.slot_base[a.n_scales], ranges = a.ranges;
        float2* const blk = reinterpret_cast<float2*>(a.partial_grad) + (((int64_t)b * ranges + w) * slots + a.slot_base[s]) * 256;
        const int64_t rstride = (int64_t)slots * 256 + 256;        // from (range r, piece j) to (range r + 1, piece j + 1)
        if constexpr (M == 10) {                                   // one frame per wave: its span is the frame, straight from the registers
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const v2f o = (wt[q] * r[q]) * scale;
                if (w + (q >> 2) < ranges) blk[(q >> 2) * rstride + 64 * (q & 3) + lane] = make_float2(o.x, o.y);
            }
        } else {
            v2f* const o = zl + j * (G::m + G::L) + kl;              // packed point i = L q + l of frame j at j (m + L) + i
#pragma unroll
            for (int q = 0; q < 16; ++q) o[G::L * q] = (wt[q] * r[q]) * scale;
            wave_sync();
            // overlap-add of the wave's frames (frame j starts at packed point j hp) in ascending frame order
            const int nfr = min(G::F, frames - frame0);
#pragma unroll
            for (int t = 0; t < (span + 63) / 64; ++t) {
                const int pp = 64 * t + lane;
                if (pp < span && w + (t >> 2) < ranges) {
                    const int f_hi = min(pp / hp, nfr - 1);
                    int f_lo = (pp - (G::m - 1) + hp - 1) / hp;
                    if (pp - (G::m - 1) <= 0) f_lo = 0;
                    v2f sum = (v2f){0.0f, 0.0f};
                    for (int f = f_lo; f <= f_hi; ++f) sum += zl[f * (G::m + G::L) + (pp - f * hp)];
                    blk[(t >> 2) * rstride + (pp & 255)] = make_float2(sum.x, sum.y);
                }
            }
            wave_sync();      // the overlap-add has read the buffer before the next task writes it
        }
    }
    MSS_STAMP(8);
}

// Persistent workgroups: each builds the two twiddle tables once (the stage twiddles in compact per-stage blocks, sot_wave_fft.hpp: tw_block; -i W_2048^k / 2 for the real-frame bins of
// every scale); after that barrier its waves (4 or 16) are on their own, each with a CONTIGUOUS range of tasks -- mostly one scale, so the
// scale's code stays in the instruction cache.
template <bool GRAD, int KIND, int kWaves>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(MSS_WAVES_PER_EU, MSS_WAVES_PER_EU))) void mss_fused_kernel(const MssArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    v2f* const tw = reinterpret_cast<v2f*>(smem_f);      // tables first: every lane-part address into a wave's buffer stays positive
    v2f* const wn = tw + kTw;
    v2f* const bufs = wn + kWnMax;
    MSS_STAMP(0);
    build_tables<64 * kWaves>(kWn, tw, wn);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    v2f* const zl = bufs + wave * kBuf;
    const int total = a.task_base[a.n_scales];
    // Tasks are ordered by scale; wave g takes tasks g, g + (waves of the grid), ...: at any moment the waves resident on a CU work on the
    // same one or two scales.  (The instruction cache is shared by the waves of a CU pair and one scale's path is ~30 KB of straight-line
    // code: with a contiguous range per wave every CU ran all six scales at once and refetched code from L2 all the time.)
    const int stride = (int)gridDim.x * kWaves;
    int s = 0;
    for (int task = (int)blockIdx.x * kWaves + wave; task < total; task += stride) {
        while (s + 1 < a.n_scales && task >= a.task_base[s + 1]) ++s;
#ifdef MSS_ONLY_M      /* diagnostic builds: one transform size (compile time, register pressure of one body) */
        wave_task<MSS_ONLY_M, GRAD, KIND>(a, s, task, tw, wn, zl);
#else
        switch (a.logm[s]) {
            case 5: wave_task<5, GRAD, KIND>(a, s, task, tw, wn, zl); break;
            case 6: wave_task<6, GRAD, KIND>(a, s, task, tw, wn, zl); break;
            case 7: wave_task<7, GRAD, KIND>(a, s, task, tw, wn, zl); break;
            case 8: wave_task<8, GRAD, KIND>(a, s, task, tw, wn, zl); break;
            case 9: wave_task<9, GRAD, KIND>(a, s, task, tw, wn, zl); break;
            default: wave_task<10, GRAD, KIND>(a, s, task, tw, wn, zl); break;
        }
#endif
    }
}

// Finish: the first workgroup turns the partial sums into the loss (per scale: fixed-order sum, mean as float32, `loss += mean` in the
// reference's scale order, losses.py:411-424); every workgroup sums the wave spans covering its samples, scales and waves in order.
#ifndef MSS_DIAG_NO_LOSS
#define MSS_DIAG_NO_LOSS 0     /* diagnostic (timing only): the finish kernel without its loss reduction */
#endif
constexpr int kFinishThreads = 256;
constexpr int kLossLoads = 4;       // partial sums per thread, scale and pass of the loss workgroup (8: one pass for 256 clips, but 148 VGPRs for the whole kernel -- three workgroups per CU instead of four: 12.7 -> 14.2 us)
__global__ __launch_bounds__(kFinishThreads) void mss_finish_kernel(const MssArgs a)
{
    __shared__ double red[(kFinishThreads / 64) * kMaxScales];
    if (a.per_clip) {        // one value per clip: a thread per clip, its tasks in order
        for (int64_t o = (int64_t)blockIdx.x * kFinishThreads + threadIdx.x; o < a.batch; o += (int64_t)gridDim.x * kFinishThreads) {
            float total = 0.0f;
            for (int s = 0; s < a.n_scales; ++s) {
                double acc = 0.0;
                for (int c = 0; c < a.waves[s]; ++c) acc += a.partial_loss[a.task_base[s] + o * a.waves[s] + c];
                total += (float)(acc * a.inv_count[s]);
            }
            a.loss[o] = total * a.post_scale;
        }
    } else if (blockIdx.x == 0 && !MSS_DIAG_NO_LOSS) {      // (the FIRST workgroup: it is resident from the start, so the reduction runs beside the other workgroups' gathers, not behind them)
        // all clips: per scale a fixed-order sum of the task partials -- thread t adds partials t, t + 256, ... of every scale, the waves reduce by
        // shuffles, thread 0 adds the four wave sums per scale in order.  The loads of ALL scales of a pass (kLossLoads per thread and scale) are issued
        // before the first is used (a loop that waits per load: 48 serialised round trips for 256 clips).
        double acc[kMaxScales];
#pragma unroll
        for (int s = 0; s < kMaxScales; ++s) acc[s] = 0.0;
        int longest = 0;
        for (int s = 0; s < a.n_scales; ++s) longest = max(longest, a.task_base[s + 1] - a.task_base[s]);
        for (int i0 = 0; i0 < longest; i0 += kLossLoads * kFinishThreads) {
            double part[kMaxScales][kLossLoads];
#pragma unroll
            for (int s = 0; s < kMaxScales; ++s) {
                const bool on = s < a.n_scales;               // uniform
                const int count = on ? a.task_base[s + 1] - a.task_base[s] : 1;
                const double* const src = a.partial_loss + (on ? a.task_base[s] : 0);
#pragma unroll
                for (int u = 0; u < kLossLoads; ++u) {
                    const int i = i0 + u * kFinishThreads + (int)threadIdx.x;
                    part[s][u] = 0.0;
                    if (on && i0 + u * kFinishThreads < count) part[s][u] = src[min(i, count - 1)];     // uniform branch
                }
            }
#pragma unroll
            for (int s = 0; s < kMaxScales; ++s) {
                const int count = (s < a.n_scales) ? a.task_base[s + 1] - a.task_base[s] : 0;
#pragma unroll
                for (int u = 0; u < kLossLoads; ++u) acc[s] += (i0 + u * kFinishThreads + (int)threadIdx.x < count) ? part[s][u] : 0.0;
            }
        }
#pragma unroll
        for (int s = 0; s < kMaxScales; ++s) {
            if (s < a.n_scales) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) acc[s] += __shfl_xor(acc[s], off);
                if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * kMaxScales + s] = acc[s];
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {                                  // lane s of the first wave finishes scale s; then `loss += mean` in float32, scale by scale
            const int sl = (int)threadIdx.x;
            double tot = 0.0;
            if (sl < a.n_scales) {
#pragma unroll
                for (int w = 0; w < kFinishThreads / 64; ++w) tot += red[w * kMaxScales + sl];
                tot *= a.inv_count[min(sl, kMaxScales - 1)];
            }
            const float mean_s = (float)tot;
            float total = 0.0f;
#pragma unroll
            for (int sc = 0; sc < kMaxScales; ++sc) {
                const float ms = __shfl(mean_s, sc);
                if (sc < a.n_scales) total += ms;                // losses.py:411-424
            }
            if (sl == 0) a.loss[0] = total * a.post_scale;
        }
    }
    if (!a.want_grad) return;
    // One workgroup per (clip, 256 packed points = one wave range): the spans that cover its points are the same for all its threads -- waves
    // w_hi - 3 ... w_hi of every scale (span = 256 + 3 m / 4 <= 1024 points) -- so the address arithmetic is scalar; scales and waves in order.
    const int half = (int)((a.samples + 1) / 2);                // packed points per clip
    const int ranges = (half + 255) >> 8;
    // COMPACT code on purpose: a rolled loop over the block's slots, sixteen loads in flight per pass (the six MSSLoss scales: 15 slots, one pass; a load of this
    // freshly written scratch takes ~2 us, so passes are what the gather costs).  (The fully unrolled form of this gather
    // -- 32 predicated loads, 19 KB of straight-line code -- took 10 us for 64 clips and 24 us for 256 however its loads were arranged:
    // every workgroup executes the code once, and the first workgroup of each CU fetches all of it from L2.)
    // two packed points per thread (16-byte loads): 128 threads per (clip, range), so each half of the workgroup takes a range of its own -- every
    // thread works and 256 clips' 2048 ranges are 1024 workgroups: ONE round of a ~4 us gather (with one range per workgroup and half the threads idle the
    // chip held 1280 of 2048 workgroups at a time: two rounds)
    MSS_STAMP(11);
    const int slots = a.slot_base[a.n_scales];
    const int side = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7)), tid = (int)threadIdx.x & 127;
    // (all clips: workgroup 0 is the loss reduction and nothing else -- its serial passes would otherwise sit in front of a gather)
    const int first = a.per_clip ? 0 : 1;
    if ((int)blockIdx.x < first) return;
    for (int64_t blk = 2 * ((int64_t)blockIdx.x - first) + side; blk < a.batch * ranges; blk += 2 * ((int64_t)gridDim.x - first)) {
        const int b = (int)(blk / ranges), w_hi = (int)(blk - (int64_t)b * ranges);
        const int t2 = 2 * tid;
        const int p = 256 * w_hi + t2;
        const float4* const blk4 = reinterpret_cast<const float4*>(a.partial_grad) + (((int64_t)b * ranges + w_hi) * slots) * 128 + tid;
        float g0x = 0.0f, g0y = 0.0f, g1x = 0.0f, g1y = 0.0f;
#pragma unroll 1
        for (int s0 = 0; s0 < slots; s0 += 16) {                 // slots in order = scales in order, pieces ascending (= waves descending)
            float4 got[16];
            int q[16], span[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int slot = min(s0 + u, slots - 1);
                const int info = a.slot_info[slot], sc = info & 255, piece = info >> 8, w = w_hi - piece;
                const bool covers = s0 + u < slots && w >= 0 && w < a.waves[sc];       // uniform: a scalar branch, the load stays in flight past it
                span[u] = covers ? 256 + 3 * ((1 << a.logm[sc]) / 4) : 0;
                q[u] = 256 * piece + t2;
                got[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (covers) got[u] = blk4[slot * 128];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {                      // the per-thread end of a span masks the VALUE (a load inside a divergent branch is waited for)
                const bool k0 = q[u] < span[u], k1 = q[u] + 1 < span[u];
                g0x += k0 ? got[u].x : 0.0f; g0y += k0 ? got[u].y : 0.0f;
                g1x += k1 ? got[u].z : 0.0f; g1y += k1 ? got[u].w : 0.0f;
            }
        }
        MSS_STAMP(12);
        g0x *= a.post_scale; g0y *= a.post_scale; g1x *= a.post_scale; g1y *= a.post_scale;
        float* const dst = a.grad + (int64_t)b * a.samples + 2 * (int64_t)p;
        const int64_t left = a.samples - 2 * (int64_t)p;        // samples of the clip from 2 p on
        if (left >= 4 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) *reinterpret_cast<float4*>(dst) = make_float4(g0x, g0y, g1x, g1y);
        else {
            if (left >= 1) dst[0] = g0x;
            if (left >= 2) dst[1] = g0y;
            if (left >= 3) dst[2] = g1x;
            if (left >= 4) dst[3] = g1y;
        }
        MSS_STAMP(13);
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
        a->waves[s] = (int)((frames + F - 1) / F);
        a->task_base[s] = (int)blocks;
        blocks += batch * a->waves[s];
        const double count = (double)frames * (double)(m + 1) * (per_clip ? 1.0 : (double)batch);
        a->inv_count[s] = 1.0 / count;
        a->coef[s] = (float)(1.0 / count);
        a->slot_base[s] = (int)gfloats;                          // (slots so far)
        for (int j = 0; j < (256 + 3 * (m / 4) + 255) / 256; ++j) a->slot_info[gfloats + j] = s | (j << 8);
        gfloats += (256 + 3 * (m / 4) + 255) / 256;             // 256-point pieces of a span: 4, 3, 2, 2, 2, 2 for n_fft 2048 ... 64
        ldoubles += batch * a->waves[s];
    }
    a->slot_base[n_scales] = (int)gfloats;
    a->ranges = (int)(((samples + 1) / 2 + 255) / 256);
    gfloats = 2 * 256 * gfloats * a->ranges * batch;            // floats: [clip][range][slot][256 points][2]
    a->task_base[n_scales] = (int)blocks;
    if (blocks > 0x7fffffffLL) return SOT_ERR_UNSUPPORTED_SIZE;
    // workspace = [partial sums (double) | padding to 16 bytes | spans (float)]: mss_finish_kernel reads the spans with 16-byte loads
    const size_t sums = (sizeof(double) * (size_t)ldoubles + 15) & ~(size_t)15;
    *workspace_bytes = sums + sizeof(float) * (size_t)gfloats;
    *grad_offset_bytes = sums;
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
                          float logmag_weight, float eps, int l2, int per_clip, float post_scale, float* loss, float* grad_value, void* workspace,
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
    if (reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return SOT_ERR_BAD_SHAPE;   // 16-byte loads of the span pieces
    a.partial_loss = reinterpret_cast<double*>(workspace);
    a.partial_grad = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + grad_off);
    a.loss = loss; a.grad = grad_value; a.want_grad = grad_value != nullptr; a.post_scale = post_scale;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) { (void)hipGetLastError(); cus = 256; }
    const int kind = (l2 == 0 && !(logmag_weight > 0.0f)) ? 0 : 1;     // the paper's configuration (L1 on the magnitudes) has its own instantiation
    const int tasks = a.task_base[n_scales], slots = 16 * cus;         // one wave per task slot: 16 waves per CU (LDS)
    const bool small = tasks <= slots;                                 // a single round: 4-wave workgroups spread it evenly
    void (*kern)(const MssArgs) =
        small ? (kind == 0 ? (a.want_grad ? mss_fused_kernel<true, 0, 4> : mss_fused_kernel<false, 0, 4>)
                           : (a.want_grad ? mss_fused_kernel<true, 1, 4> : mss_fused_kernel<false, 1, 4>))
              : (kind == 0 ? (a.want_grad ? mss_fused_kernel<true, 0, 16> : mss_fused_kernel<false, 0, 16>)
                           : (a.want_grad ? mss_fused_kernel<true, 1, 16> : mss_fused_kernel<false, 1, 16>));
    const int wg_waves = small ? 4 : 16;     // LDS: tables 16.4 KB + 8.7 KB per wave: three 4-wave workgroups or one of 16 waves per CU
    static bool attr_done[64][8] = {};
    const int which = 4 * (small ? 1 : 0) + 2 * kind + a.want_grad;
    if (dev < 0 || dev >= 64 || !attr_done[dev][which]) {   // idempotent per device; a benign race sets it twice
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(wg_waves)) != hipSuccess)
            (void)hipGetLastError();
        if (dev >= 0 && dev < 64) attr_done[dev][which] = true;
    }
    // persistent grid: at most 16 waves per CU; wave g takes tasks g, g + (waves of the grid), ...
    const int waves_needed = tasks < slots ? tasks : slots;
#ifndef MSS_SKIP_FUSED     /* diagnostic (timing only, results are garbage): the finish kernel on its own */
    hipLaunchKernelGGL(kern, dim3((unsigned)((waves_needed + wg_waves - 1) / wg_waves)), dim3(64 * wg_waves), lds_bytes(wg_waves), st, a);
#endif
    if (hipGetLastError() != hipSuccess) return SOT_ERR_LAUNCH;
    const int64_t work = a.want_grad ? (batch * ((((samples + 1) / 2) + 255) / 256) + 1) / 2 : 1;      // one workgroup per TWO (clip, 256 packed points) blocks
    const int64_t blocks = (work < 16384 ? (work < 1 ? 1 : work) : 16384) + ((a.want_grad && !per_clip) ? 1 : 0);      // + the loss workgroup
    hipLaunchKernelGGL(mss_finish_kernel, dim3((unsigned)blocks), dim3(kFinishThreads), 0, st, a);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

#ifdef MSS_STAMPS
int sot_mss_debug_read_stamps(unsigned long long* host_out, int count)   // diagnostic build only (synchronises)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(sot_mss::g_mss_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
