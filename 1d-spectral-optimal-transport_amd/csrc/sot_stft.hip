// sot_stft.hip -- MI355X (gfx950) kernels for the producer in front of the SOT loss: the magnitude STFT of the
// reference's `features.TorchSTFT` (features.py:85-113 -> compute_mag / stft, features.py:191-237; end padding
// utils.pad_for_stft, utils.py:252-275): torch.stft(center=False, normalized=True, onesided) of the end-padded signal,
// |.|, frames-major output [batch, frames, n_fft/2 + 1].  SURVEY §8f row 1.
//
// Forward: one frame slot (n_fft/8 threads) per frame, kernels instantiated per transform size.  The windowed REAL frame
// is packed into n_fft/2 complex points and goes through an in-LDS complex FFT of half the frame length (bit-reversed
// load, two radix-2 stages per barrier-separated pass, twiddles copied into LDS from constant tables); the n_fft/2 + 1
// bins are unpacked pairwise and reduced to |.| / sqrt(n_fft).
// Backward (closed form of abs o stft's autograd): per group of two consecutive frames, frame by frame, recompute the
// frame's spectrum X, form
// Z_k = g_k X_k / |X_k| (0 where |X_k| = 0, torch's sgn(0)), inverse-transform it as a Hermitian spectrum (again a
// half-length complex transform), multiply by window / sqrt(n_fft) and overlap-add it into the group's gradient, which
// is kept in LDS and written once to a scratch buffer; a second kernel adds, per sample, the groups that cover it in a
// fixed order: no atomics, deterministic.
// HBM traffic: forward reads n_fft samples per frame (L2-resident overlap) and writes n_fft/2+1 magnitudes; backward
// reads the audio and the magnitude gradients once, writes and re-reads the groups' partial gradients (4.5 x the audio
// for 2 frames per group at 8 frames per sample: 19 MB for 4096 frames of 2048, L2/MALL-resident) and writes the audio
// gradient once.  (4 frames per group: 2.75 x, but half the workgroups; the same time at n_fft 2048, 20 % slower over
// the six scales of MSSLoss.)
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <math.h>

#include "../../include/sot_hip.h"
#include "sot_wave_fft.hpp"

namespace sot_stft {

constexpr int kThreads = 256;
#ifndef SOT_STFT_FRAMES_PER_GROUP
#define SOT_STFT_FRAMES_PER_GROUP 2
#endif
constexpr int kFramesPerGroup = SOT_STFT_FRAMES_PER_GROUP;  // backward: one frame slot per group of consecutive frames of a clip
constexpr int kMaxFft = 4096;

#include "sot_stft_tables.inc"      // kPassTw, kWn (csrc/gen/make_stft_tables.py)

#ifndef SOT_STFT_RAW_SQRT
#define SOT_STFT_RAW_SQRT 0   /* 1: v_sqrt_f32 alone -- 14 % fewer VALU instructions, the same 24.7 us, and past the 2e-6 pin against float64 */
#endif
#ifndef SOT_STFT_PERSISTENT_WAVES
#define SOT_STFT_PERSISTENT_WAVES 6   /* waves per SIMD the persistent forward kernel is compiled for: 6 = 76 VGPRs, no spills; 8 =
                                         64 VGPRs with 10 spilled, measured slower (26.4 vs 21.9 us) */
#endif
#ifndef SOT_STFT_WAVE_FRAMES
#define SOT_STFT_WAVE_FRAMES 1
#endif
// Timing-only ablation (tools/fusion_probe.py; results are WRONG on purpose, never defined in the product build): the forward kernels
// compute every magnitude but store none -- what a consumer fused behind the transform would save on the producer's side.
#ifndef SOT_STFT_ABLATE_STORE
#define SOT_STFT_ABLATE_STORE 0
#endif
__device__ __forceinline__ void store_mag(float* dst, int k, float v)
{
    if (SOT_STFT_ABLATE_STORE) { if (v == 12345.678f) dst[k] = v; }   // keeps the value alive, never true for |.| / sqrt(n) of audio
    else dst[k] = v;
}

// synchronisation of one frame slot (see Geo::wave_sync): the LDS executes a wavefront's instructions in issue order
template <bool WAVE>
__device__ __forceinline__ void slot_sync()
{
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

typedef float v2f __attribute__((ext_vector_type(2)));   // one complex point; arithmetic maps to v_pk_*_f32

__device__ __forceinline__ void store_spec(float2* sp, int k, int mk, v2f xk, v2f xm)
{
    sp[k] = make_float2(xk.x, xk.y);
    sp[mk] = make_float2(xm.x, xm.y);
}

// Complex product on the packed-fp32 unit.  Round 4: the swizzled product a.yy * (-b.y, b.x) is ONE v_pk_mul_f32 whose operand halves
// are picked by op_sel and negated by neg_lo (hipcc builds the swizzled operand with v_xor + v_mov first: five instructions per product
// instead of three).  Same three roundings as the vector expression, so every result is bit-identical to rounds 2-3.  (A fused form --
// v_pk_mul + v_pk_fma, two instructions -- was measured too: its last-bit differences move rows of the SOT stage across the cutoff's
// knife edge, which the float64 yardstick test of the audio-in chain does not tolerate: 1.2e-5 -> 1.5e-4 of the gradient's peak.)
#ifndef SOT_STFT_ASM_CMUL
#define SOT_STFT_ASM_CMUL 1
#endif
__device__ __forceinline__ v2f cmul(v2f a, v2f b)
{
#if SOT_STFT_ASM_CMUL
    v2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(b));   // (-a.y b.y, a.y b.x)
    return a.xx * b + t;
#else
    return a.xx * b + a.yy * (v2f){-b.y, b.x};
#endif
}
// a * conj(b) = a.xx * (b.x, -b.y) + a.yy * (b.y, b.x)
__device__ __forceinline__ v2f cmul_conj(v2f a, v2f b)
{
#if SOT_STFT_ASM_CMUL
    v2f t1, t2;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t1) : "v"(a), "v"(b));   // (a.x b.x, -a.x b.y)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t2) : "v"(a), "v"(b));               // (a.y b.y, a.y b.x)
    return t1 + t2;
#else
    return a.xx * (v2f){b.x, -b.y} + a.yy * (v2f){b.y, b.x};
#endif
}
// a + (-i) b = (a.x + b.y, a.y - b.x)   and   a + (+i) b = (a.x - b.y, a.y + b.x): one packed add with swapped / negated halves
__device__ __forceinline__ v2f add_mi(v2f a, v2f b)
{
#if SOT_STFT_ASM_CMUL
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return (v2f){a.x + b.y, a.y - b.x};
#endif
}
__device__ __forceinline__ v2f add_pi(v2f a, v2f b)
{
#if SOT_STFT_ASM_CMUL
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return (v2f){a.x - b.y, a.y + b.x};
#endif
}
__device__ __forceinline__ v2f cconj(v2f a) { return (v2f){a.x, -a.y}; }
__device__ __forceinline__ v2f mul_i(v2f a) { return (v2f){-a.y, a.x}; }    // a * (+i)
__device__ __forceinline__ v2f mul_mi(v2f a) { return (v2f){a.y, -a.x}; }   // a * (-i)

// LDS index of element i of a transform buffer: one point of padding after every 32.  The bit-reversed load (lane
// stride m/2, m/4, ...) and the stride-4 accesses of the first pass would otherwise pile 32 lanes onto two banks
// (SQ_LDS_BANK_CONFLICT 72 % -> 37 % of the LDS cycles of the forward kernel).
#ifndef SOT_STFT_PAD
#define SOT_STFT_PAD 0   /* diagnostic: other paddings of the transform buffer.  tools/ab_stft.py, n_fft 2048 forward / backward:
                            0 (one point per 32) 22.4 / 53.0 us; 1 (per 16) 23.3 / 54.2; 2 (per 32 and per 256) 22.6 / 54.0;
                            3 (per 16 and per 256) 22.3 / 53.5; 4 (per 8) 23.6 / 54.8; 5 (none) 29.3 / 65.7 */
#endif
__host__ __device__ constexpr int zi(int i)
{
    return SOT_STFT_PAD == 0 ? i + (i >> 5)
         : SOT_STFT_PAD == 1 ? i + (i >> 4)
         : SOT_STFT_PAD == 2 ? i + (i >> 5) + (i >> 8)
         : SOT_STFT_PAD == 3 ? i + (i >> 4) + (i >> 8)
         : SOT_STFT_PAD == 4 ? i + (i >> 3)
         : i;
}

// Frame geometry for n_fft = 2^(LOGM+1).  The frames are REAL, so each one is transformed by a complex FFT of HALF its
// length m = n_fft/2 on the packed signal z[i] = v[2i] + i v[2i+1]:   with Ze = (Z_k + conj(Z_{m-k})) / 2,
// Zo = -i/2 (Z_k - conj(Z_{m-k})), W = exp(-2 pi i k / n):   X_k = Ze + W Zo,   X_{m-k} = conj(Ze - W Zo)   (k <= m/2).
// One frame SLOT is m/4 threads (one radix-4 butterfly each per pass; at least 16); small transforms share a workgroup:
// n_fft = 64 / 128 -> 16 frames per workgroup, 256 -> 8, 512 -> 4, 1024 -> 2, 2048 and 4096 -> 1 (4096: two butterflies per thread).
// LDS: z [slots][zi(m)] | per-pass FFT twiddles [m] | W_n^k [m/2 + 2]  (| backward: overlap-add buffers [slots][span]).
template <int LOGM>
struct Geo {
    static constexpr int logm = LOGM, m = 1 << LOGM, n = 2 * m, nb = m + 1;
    // threads per frame slot: one radix-4 butterfly per thread and pass (m/4), at least 16.  Slots of at most one wavefront
    // (n_fft <= 512) synchronise with a compiler-level ordering point instead of the workgroup barrier (slot_sync): n_fft 512
    // forward 10.7 -> 10.3 us, backward 30.2 -> 28.7 us for 8192 frames.  (Shrinking the slots of n_fft 1024 / 2048 to one
    // wavefront -- 2 / 4 butterflies per thread and pass, SOT_STFT_WAVE_FRAMES=2 -- was measured slower: n_fft 2048 forward
    // 24.6 -> 29.4 us, backward 53 -> 104 us.)
    static constexpr int tpf_full = (m / 4 > 16) ? (m / 4 > kThreads ? kThreads : m / 4) : 16;   // n_fft 4096: two butterflies per thread and pass
    static constexpr int tpf = (SOT_STFT_WAVE_FRAMES == 2 && tpf_full > 64) ? 64 : tpf_full;
    static constexpr bool wave_sync = (SOT_STFT_WAVE_FRAMES != 0) && tpf <= 64;
    static constexpr int slots = kThreads / tpf;
    static constexpr int zpoints = zi(m - 1) + 2;
    static constexpr int table_points = m + m / 2 + 2;
    static constexpr size_t lds_points = (size_t)slots * zpoints + table_points;
};

// twiddles from the constant tables (L2-resident) into LDS: tw[2 (h - 1) + pos] = exp(-2 pi i pos / 2h),
// tw[2 (h - 1) + h + pos] = exp(-2 pi i pos / 4h) for every pass half size h <= m/4;  wn[k] = exp(-2 pi i k / n)
template <int LOGM>
__device__ __forceinline__ void load_tables(v2f* tw, v2f* wn)
{
    using G = Geo<LOGM>;
    for (int i = threadIdx.x; i < G::m - 2; i += kThreads) { const float2 t = kPassTw[i]; tw[i] = (v2f){t.x, t.y}; }
    for (int k = threadIdx.x; k <= G::m / 2; k += kThreads) { const float2 t = kWn[k << (11 - LOGM)]; wn[k] = (v2f){t.x, t.y}; }
}

// In-place decimation-in-time FFT of m = 2^LOGM points held in LDS (element i at z[zi(i)]) in BIT-REVERSED order on
// entry, natural order on exit, executed by the tpf threads lid = 0 .. tpf-1 of one frame slot (the barriers are
// workgroup-wide, every slot runs the same passes).  Two radix-2 stages are executed per pass (a radix-4 butterfly on the
// elements i0, i0+h, i0+2h, i0+3h: the same operations, in the same order per element, as two separate stages -- half
// the LDS round trips and barriers); an odd LOGM starts with one plain radix-2 stage.  INVERSE: conjugate twiddles
// (no 1/m).  Ends with a barrier.
template <int LOGM, bool INVERSE>
__device__ __forceinline__ void fft_inplace(v2f* z, const v2f* tw, int lid)
{
    using G = Geo<LOGM>;
    if (LOGM & 1) {  // stage 1: half = 1, twiddle 1
        slot_sync<G::wave_sync>();
        for (int j = lid; j < G::m / 2; j += G::tpf) {
            const int p0 = zi(2 * j), p1 = zi(2 * j + 1);
            const v2f a = z[p0], b = z[p1];
            z[p0] = a + b;
            z[p1] = a - b;
        }
    }
#pragma unroll
    for (int s = (LOGM & 1) ? 2 : 1; s <= LOGM; s += 2) {
        const int h = 1 << (s - 1);       // half size of stage s; stage s+1 has half size 2h
        slot_sync<G::wave_sync>();
        for (int j = lid; j < G::m / 4; j += G::tpf) {
            const int pos = j & (h - 1);
            const int i0 = ((j >> (s - 1)) << (s + 1)) + pos;
            const int p0 = zi(i0), p1 = zi(i0 + h), p2 = zi(i0 + 2 * h), p3 = zi(i0 + 3 * h);
            v2f w1 = tw[2 * (h - 1) + pos], w2 = tw[2 * (h - 1) + h + pos];
            if (INVERSE) { w1 = cconj(w1); w2 = cconj(w2); }
            const v2f a = z[p0], b = cmul(z[p1], w1), c = z[p2], d = cmul(z[p3], w1);
            const v2f a1 = a + b, b1 = a - b, c1 = c + d, d1 = c - d;
            const v2f c2 = cmul(c1, w2);
            // exp(-2 pi i (pos + h) / (4h)) = w2 * (-i)  (forward),  w2 * (+i)  (inverse)
            const v2f d2 = cmul(d1, INVERSE ? mul_i(w2) : mul_mi(w2));
            z[p0] = a1 + c2;
            z[p2] = a1 - c2;
            z[p1] = b1 + d2;
            z[p3] = b1 - d2;
        }
    }
    slot_sync<G::wave_sync>();
}

__device__ __forceinline__ int bitrev(int v, int logn) { return (int)(__brev((unsigned)v) >> (32 - logn)); }

struct StftArgs {
    const float* audio; int64_t batch, samples, row_stride;
    const float* audio_b; int64_t split, row_stride_b;   // forward of two signals in one launch: clips >= split come from audio_b
    const float* window; int n_fft, logm, hop; int64_t frames;   // logm = log2(n_fft / 2)
    float* mag;                 // forward output [batch, frames, n_fft/2+1]
    float2* spec;               // forward, optional: the complex spectrum X of the clips >= spec_first, [batch - spec_first, frames, n_fft/2+1]
    int64_t spec_first;         //   (what abs()'s autograd would save: the backward then needs no second forward transform)
    const float2* spec_in;      // backward, optional: that spectrum for all `batch` clips (then `audio` is not read)
    const float* grad_mag;      // backward input, same shape
    const float* grad_scale;    // backward: optional device scalar multiplying grad_mag (an upstream gradient), or null
    int accumulate;             // backward: grad_audio += result (the sum over the scales of MSSLoss)
    float* grad_audio;          // backward output [batch, samples] (contiguous)
    float* partial;             // backward scratch [batch, groups, span]: each frame group's overlap-added gradient
    int64_t groups; int span;   // span = n_fft + hop * (kFramesPerGroup - 1)
};

// windowed, end-padded, packed frame -> LDS in bit-reversed order (zeros for an idle slot)
template <int LOGM>
__device__ __forceinline__ void load_frame(const StftArgs& a, const float* src, int64_t t0, v2f* z, bool active, int lid)
{
    using G = Geo<LOGM>;
    const float2* win = reinterpret_cast<const float2*>(a.window);   // 8-byte aligned (checked by the host)
    const bool inside = active && t0 + G::n <= a.samples;            // the whole frame lies inside the clip
    const float* const s0 = src + t0;                                // frame start: 32-bit offsets from here on
    const int left = (int)min((int64_t)G::n, a.samples - t0);        // samples of the frame that exist
#pragma unroll
    for (int i = lid; i < G::m; i += G::tpf) {
        const float2 w = win[i];
        float v0, v1;
        if (inside) {
            v0 = s0[2 * i] * w.x; v1 = s0[2 * i + 1] * w.y;
        } else {                                                      // end padding: zeros (utils.py:252-275)
            v0 = (active && 2 * i < left) ? s0[2 * i] * w.x : 0.0f;
            v1 = (active && 2 * i + 1 < left) ? s0[2 * i + 1] * w.y : 0.0f;
        }
        z[zi(bitrev(i, LOGM))] = (v2f){v0, v1};
    }
}

// the same with the thread's window taps (points lid + j tpf) already in registers
template <int LOGM>
__device__ __forceinline__ void load_frame_w(const StftArgs& a, const float* src, int64_t t0, v2f* z, bool active, int lid,
                                             const float2 (&w)[Geo<LOGM>::m / Geo<LOGM>::tpf])
{
    using G = Geo<LOGM>;
    const bool inside = active && t0 + G::n <= a.samples;
    const float* const s0 = src + t0;
    const int left = (int)min((int64_t)G::n, a.samples - t0);
#pragma unroll
    for (int j = 0; j < G::m / G::tpf; ++j) {
        const int i = lid + j * G::tpf;
        float v0, v1;
        if (inside) {
            v0 = s0[2 * i] * w[j].x; v1 = s0[2 * i + 1] * w[j].y;
        } else {
            v0 = (active && 2 * i < left) ? s0[2 * i] * w[j].x : 0.0f;
            v1 = (active && 2 * i + 1 < left) ? s0[2 * i + 1] * w[j].y : 0.0f;
        }
        z[zi(bitrev(i, LOGM))] = (v2f){v0, v1};
    }
}

// |x| as torch's abs(complex) gives it (hypot): sqrt(re^2 + im^2) wherever the squares stay normal (a correctly rounded
// sqrt of a sum that is good to 1 ulp); the scaled hypotf when any lane of the wave holds a tiny or huge value
__device__ __forceinline__ float magnitude(v2f x)
{
    const float s = fmaf(x.x, x.x, x.y * x.y);
    const bool plain = (s > 1e-30f && s < 1e30f) || (x.x == 0.0f && x.y == 0.0f);
    if (__builtin_expect(__ballot(!plain) != 0ull, 0)) return hypotf(x.x, x.y);
#if SOT_STFT_RAW_SQRT
    return __builtin_amdgcn_sqrtf(s);   // v_sqrt_f32 (1 ulp) without the range scaling and the two refinement steps of sqrtf: s is normal here
#else
    return sqrtf(s);
#endif
}

// |x| for a frame whose input amplitude has been checked once (frame_is_plain below): re^2 + im^2 can neither overflow nor lose the bins
// that matter to underflow, so the per-value range test, the wave vote and the libm sqrtf (range scaling + special cases: ~20 VALU
// instructions) shrink to v_sqrt_f32, v_rsq_f32 and one Newton step on the residual (8 instructions; the result is within half an ulp of
// sqrt(s) up to the rounding of s itself, like torch's hypot-based abs(complex)); s == 0 gives 0.
__device__ __forceinline__ float magnitude_plain(v2f x)
{
    const float s = fmaf(x.x, x.x, x.y * x.y);
    const float r = __builtin_amdgcn_sqrtf(s);
    const float h = 0.5f * __builtin_amdgcn_rsqf(s);
    const float e = fmaf(-r, r, s);
    const float v = fmaf(e, h, r);
    return s == 0.0f ? 0.0f : v;
}
// amax = largest |sample * tap| of the frame.  |X_k| <= n amax, so the squares cannot overflow below 1e15; the spectrum's peak is >= amax
// (Parseval), so with amax > 1e-9 every bin within 1e-10 of the peak keeps a normal square.  An all-zero frame is plain too (every bin 0).
__device__ __forceinline__ bool frame_is_plain(float amax) { return (amax > 1e-9f && amax < 1e15f) || amax == 0.0f; }

// spectrum bins k and m-k of the real frame from the packed transform (see Geo)
template <int LOGM>
__device__ __forceinline__ void unpack_pair(const v2f* z, const v2f* wn, int k, v2f& xk, v2f& xm)
{
    constexpr int m = 1 << LOGM;
    const v2f zk = z[zi(k)], zm = z[zi((m - k) & (m - 1))];
    const v2f ze = 0.5f * (zk + cconj(zm));
    const v2f zo = 0.5f * mul_mi(zk - cconj(zm));          // -i/2 (Z_k - conj(Z_{m-k}))
    const v2f wz = cmul(wn[k], zo);
    xk = ze + wz;
    xm = cconj(ze - wz);
}

// the same with the bin's twiddle W_n^k passed by value
template <int LOGM>
__device__ __forceinline__ void unpack_pair_w(const v2f* z, v2f w, int k, v2f& xk, v2f& xm)
{
    constexpr int m = 1 << LOGM;
    const v2f zk = z[zi(k)], zm = z[zi((m - k) & (m - 1))];
    const v2f ze = 0.5f * (zk + cconj(zm));
    const v2f zo = 0.5f * mul_mi(zk - cconj(zm));
    const v2f wz = cmul(w, zo);
    xk = ze + wz;
    xm = cconj(ze - wz);
}

template <int LOGM>
__global__ __launch_bounds__(kThreads) void stft_mag_forward_kernel(const StftArgs a)
{
    using G = Geo<LOGM>;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int slot = threadIdx.x / G::tpf, lid = threadIdx.x - slot * G::tpf;
    v2f* const zall = reinterpret_cast<v2f*>(smem_f);
    v2f* const z = zall + slot * G::zpoints;
    v2f* const tw = zall + G::slots * G::zpoints;
    v2f* const wn = tw + G::m;
    const float scale = 1.0f / sqrtf((float)G::n);  // normalized=True: frame_length^-0.5
    const unsigned total = (unsigned)(a.batch * a.frames), frames = (unsigned)a.frames;   // < 2^31 (host)
    load_tables<LOGM>(tw, wn);
    if constexpr (G::wave_sync) __syncthreads();   // the tables are shared by the slots, which then run wave-synchronised
    const unsigned fr = blockIdx.x * G::slots + slot;
    const bool active = fr < total;
    const unsigned b = active ? fr / frames : 0u, f = active ? fr - b * frames : 0u;
    const float* src = (a.audio_b != nullptr && (int64_t)b >= a.split) ? a.audio_b + ((int64_t)b - a.split) * a.row_stride_b
                                                                      : a.audio + (int64_t)b * a.row_stride;
    load_frame<LOGM>(a, src, (int64_t)f * a.hop, z, active, lid);
    fft_inplace<LOGM, false>(z, tw, lid);
    if (active) {
        float* dst = a.mag + (int64_t)fr * G::nb;
        float2* sp = (a.spec != nullptr && (int64_t)b >= a.spec_first) ? a.spec + ((int64_t)fr - a.spec_first * frames) * G::nb : nullptr;
#pragma unroll
        for (int k = lid; k <= G::m / 2; k += G::tpf) {
            v2f xk, xm;
            unpack_pair<LOGM>(z, wn, k, xk, xm);
            store_mag(dst, k, magnitude(xk) * scale);
            store_mag(dst, G::m - k, magnitude(xm) * scale);
            if (sp != nullptr) store_spec(sp, k, G::m - k, xk, xm);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same transform with PERSISTENT workgroups (SOT_STFT_PERSISTENT): a workgroup loads the twiddle tables and its threads' window
// taps once, then walks over frame groups g = blockIdx.x, blockIdx.x + gridDim.x, ...; the audio of the NEXT frame is fetched into
// registers before the passes of the current one start, so its latency (and the tables') is paid once per workgroup instead of once
// per frame.  Results are identical to stft_mag_forward_kernel's (the same operations on the same values).
// (LDS: z [slots][zi(m)] | pass twiddles [m]; the unpacking twiddles W_n^k of a thread's bins sit in registers: 16.6 KB at n_fft 2048,
// eight workgroups per CU, which the 64-VGPR budget of amdgpu_waves_per_eu(8) matches.)
template <int LOGM>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(SOT_STFT_PERSISTENT_WAVES, 8))) void stft_mag_forward_persistent_kernel(const StftArgs a)
{
    using G = Geo<LOGM>;
    constexpr int PER = G::m / G::tpf;                       // packed points per thread and frame
    constexpr int PERK = (G::m / 2 + G::tpf) / G::tpf;       // bin pairs (k, m - k), k <= m/2, per thread and frame
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int slot = threadIdx.x / G::tpf, lid = threadIdx.x - slot * G::tpf;
    v2f* const zall = reinterpret_cast<v2f*>(smem_f);
    v2f* const z = zall + slot * G::zpoints;
    v2f* const tw = zall + G::slots * G::zpoints;
    v2f wnr[PERK];
#pragma unroll
    for (int j = 0; j < PERK; ++j) {
        const int k = min(lid + j * G::tpf, G::m / 2);
        const float2 t = kWn[k << (11 - LOGM)];
        wnr[j] = (v2f){t.x, t.y};
    }
    const float scale = 1.0f / sqrtf((float)G::n);
    const unsigned total = (unsigned)(a.batch * a.frames), frames = (unsigned)a.frames;
    const unsigned ngroups = (total + G::slots - 1) / G::slots;
    const float2* win = reinterpret_cast<const float2*>(a.window);
    float2 w[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) w[j] = win[lid + j * G::tpf];

    float2 cur[PER];
    auto fetch = [&](unsigned g) {   // raw samples 2i, 2i+1 of frame g*slots+slot for i = lid + j tpf; zeros outside the clip / for an idle slot
        const unsigned fr = g * G::slots + slot;
        const bool active = fr < total;
        const unsigned b = active ? fr / frames : 0u, f = active ? fr - b * frames : 0u;
        const float* src = (a.audio_b != nullptr && (int64_t)b >= a.split) ? a.audio_b + ((int64_t)b - a.split) * a.row_stride_b
                                                                          : a.audio + (int64_t)b * a.row_stride;
        const int64_t t0 = (int64_t)f * a.hop;
        const float* const s0 = src + t0;
        const int left = active ? (int)min((int64_t)G::n, a.samples - t0) : 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = lid + j * G::tpf;
            cur[j].x = (2 * i < left) ? s0[2 * i] : 0.0f;
            cur[j].y = (2 * i + 1 < left) ? s0[2 * i + 1] : 0.0f;
        }
    };
    unsigned g = blockIdx.x;
    if (g < ngroups) fetch(g);
    for (int i = threadIdx.x; i < G::m - 2; i += kThreads) { const float2 t = kPassTw[i]; tw[i] = (v2f){t.x, t.y}; }   // load_tables() without W_n
    if constexpr (G::wave_sync) __syncthreads();
    for (; g < ngroups; g += gridDim.x) {
#pragma unroll
        for (int j = 0; j < PER; ++j)   // the products of load_frame(): sample * tap (zero * tap = 0 for the padding)
            z[zi(bitrev(lid + j * G::tpf, LOGM))] = (v2f){cur[j].x * w[j].x, cur[j].y * w[j].y};
        const unsigned fr = g * G::slots + slot;
        if (g + gridDim.x < ngroups) fetch(g + gridDim.x);   // in flight during the passes below
        fft_inplace<LOGM, false>(z, tw, lid);
        if (fr < total) {
            float* dst = a.mag + (int64_t)fr * G::nb;
            float2* sp = (a.spec != nullptr && (int64_t)(fr / frames) >= a.spec_first) ? a.spec + ((int64_t)fr - a.spec_first * frames) * G::nb : nullptr;
#pragma unroll
            for (int j = 0; j < PERK; ++j) {
                const int k = lid + j * G::tpf;
                if (k <= G::m / 2) {
                    v2f xk, xm;
                    unpack_pair_w<LOGM>(z, wnr[j], k, xk, xm);
                    store_mag(dst, k, magnitude(xk) * scale);
                    store_mag(dst, G::m - k, magnitude(xm) * scale);
                    if (sp != nullptr) store_spec(sp, k, G::m - k, xk, xm);
                }
            }
        }
        slot_sync<G::wave_sync>();   // the spectrum has been read before the next frame overwrites it
    }
}

// n_fft = 2048, one WAVEFRONT per frame (SOT_STFT_WAVE_KERNEL): the 1024-point complex transform of the packed frame as five
// radix-4 decimation-in-frequency stages on 16 points per lane.  With the index written in base 4, i = (d4 d3 d2 d1 d0), a lane
// keeps two digits in its 16 registers and the other three are its lane number: stages 1-2 (digits d4, d3) on
// r = 4 d4 + d3, lane = 16 d2 + 4 d1 + d0 (element 64 r + lane: the load is coalesced), one exchange through the wave's LDS
// buffer, stages 3-4 (d2, d1), a second exchange, stage 5 (d0); the results are written to LDS in natural frequency order
// (k = q4 + 4 q3 + 16 q2 + 64 q1 + 256 q0) for the pairwise unpacking of the real transform.  No workgroup barrier inside the
// frame loop (slot_sync<true>), twiddles W_1024^j from one LDS table per workgroup, 4 frames in flight per workgroup in a
// persistent grid.  Index algebra checked against numpy's FFT (6e-14) before it was written down here.
// ---------------------------------------------------------------------------------------------
constexpr int kWaveBuf = 64 * 17;   // v2f slots of one wave's exchange buffer: [lane][16 registers + 1 pad]; >= zi(1024) = 1056

// radix-4 butterfly, outputs q = 0..3: sum_p a_p W_4^{p q} (forward: W_4 = -i; INVERSE: +i)
template <bool INVERSE>
__device__ __forceinline__ void bf4(const v2f a0, const v2f a1, const v2f a2, const v2f a3, v2f& o0, v2f& o1, v2f& o2, v2f& o3)
{
    const v2f s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    const v2f rot = INVERSE ? mul_i(d13) : mul_mi(d13);
    o0 = s02 + s13; o1 = d02 + rot; o2 = s02 - s13; o3 = d02 - rot;
}

template <bool INVERSE>
__device__ __forceinline__ v2f twid(const v2f* tw, int j) { const v2f w = tw[j & 1023]; return INVERSE ? cconj(w) : w; }

// r[q] = z[64 q + lane] on entry; on return the transform sits in zl[zi(k)], k = 0 .. 1023 (after the caller's slot_sync)
template <bool INVERSE>
__device__ __forceinline__ void fft1024_wave(v2f (&r)[16], v2f* zl, const v2f* tw, int lane)
{
    v2f o[16];
    // stage 1 (digit d4; registers 4 p + d3): twiddle W_1024^{(64 d3 + lane) q}
#pragma unroll
    for (int d3 = 0; d3 < 4; ++d3) {
        const int j = 64 * d3 + lane;
        bf4<INVERSE>(r[d3], r[4 + d3], r[8 + d3], r[12 + d3], o[d3], o[4 + d3], o[8 + d3], o[12 + d3]);
#pragma unroll
        for (int q = 1; q < 4; ++q) o[4 * q + d3] = cmul(o[4 * q + d3], twid<INVERSE>(tw, j * q));
    }
    // stage 2 (digit d3; registers 4 q4 + p): twiddle W_256^{lane q} = W_1024^{4 lane q}
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        bf4<INVERSE>(o[4 * q4], o[4 * q4 + 1], o[4 * q4 + 2], o[4 * q4 + 3], r[4 * q4], r[4 * q4 + 1], r[4 * q4 + 2], r[4 * q4 + 3]);
#pragma unroll
        for (int q = 1; q < 4; ++q) r[4 * q4 + q] = cmul(r[4 * q4 + q], twid<INVERSE>(tw, 4 * lane * q));
    }
    // exchange 1: register (q4, q3) of lane (d2, d1, d0) -> register (d2, d1) of lane (q4, q3, d0)
    {
        const int d0 = lane & 3, rr = lane >> 2;   // rr = 4 d2 + d1
#pragma unroll
        for (int q = 0; q < 16; ++q) zl[(4 * q + d0) * 17 + rr] = r[q];
        slot_sync<true>();
#pragma unroll
        for (int q = 0; q < 16; ++q) r[q] = zl[lane * 17 + q];
        slot_sync<true>();
    }
    // stage 3 (digit d2; registers 4 p + d1): twiddle W_64^{(4 d1 + d0) q} = W_1024^{16 (4 d1 + d0) q}
    const int d0 = lane & 3;
#pragma unroll
    for (int d1 = 0; d1 < 4; ++d1) {
        const int j = 16 * (4 * d1 + d0);
        bf4<INVERSE>(r[d1], r[4 + d1], r[8 + d1], r[12 + d1], o[d1], o[4 + d1], o[8 + d1], o[12 + d1]);
#pragma unroll
        for (int q = 1; q < 4; ++q) o[4 * q + d1] = cmul(o[4 * q + d1], twid<INVERSE>(tw, j * q));
    }
    // stage 4 (digit d1; registers 4 q2 + p): twiddle W_16^{d0 q} = W_1024^{64 d0 q}
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {
        bf4<INVERSE>(o[4 * q2], o[4 * q2 + 1], o[4 * q2 + 2], o[4 * q2 + 3], r[4 * q2], r[4 * q2 + 1], r[4 * q2 + 2], r[4 * q2 + 3]);
#pragma unroll
        for (int q = 1; q < 4; ++q) r[4 * q2 + q] = cmul(r[4 * q2 + q], twid<INVERSE>(tw, 64 * d0 * q));
    }
    // exchange 2: register (q2, q1) of lane (q4, q3, d0) -> register (d0, q1) of lane (q4, q3, q2)
    {
        const int hi = lane >> 2;   // 4 q4 + q3
#pragma unroll
        for (int q = 0; q < 16; ++q) zl[(4 * hi + (q >> 2)) * 17 + 4 * d0 + (q & 3)] = r[q];
        slot_sync<true>();
#pragma unroll
        for (int q = 0; q < 16; ++q) r[q] = zl[lane * 17 + q];
        slot_sync<true>();
    }
    // stage 5 (digit d0; registers 4 p + q1), no twiddle; result (q4 q3 q2 q1 q0) is frequency k = q4 + 4 q3 + 16 q2 + 64 q1 + 256 q0
    const int kb = (lane >> 4) + 4 * ((lane >> 2) & 3) + 16 * (lane & 3);
#pragma unroll
    for (int q1 = 0; q1 < 4; ++q1) {
        bf4<INVERSE>(r[q1], r[4 + q1], r[8 + q1], r[12 + q1], o[q1], o[4 + q1], o[8 + q1], o[12 + q1]);
#pragma unroll
        for (int q0 = 0; q0 < 4; ++q0) zl[zi(kb + 64 * q1 + 256 * q0)] = o[4 * q0 + q1];
    }
}

__global__ __launch_bounds__(kThreads) void stft_mag_forward_wave_kernel(const StftArgs a)
{
    constexpr int LOGM = 10, m = 1024, n = 2048, nb = m + 1;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    v2f* const tw = reinterpret_cast<v2f*>(smem_f);          // W_1024^j, j < 1024
    v2f* const wn = tw + 1024;                                // W_2048^k, k <= 512 (+ pad)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    v2f* const zl = wn + 520 + wave * kWaveBuf;
    // W_1024^{256 a + b} = W_2048^{2 b} (-i)^a  (exact quarter turns of the committed table)
    for (int j = threadIdx.x; j < 1024; j += kThreads) {
        const float2 t = kWn[4 * (j & 255)];
        v2f w = (v2f){t.x, t.y};
        const int qa = j >> 8;
        if (qa == 1) w = mul_mi(w); else if (qa == 2) w = -w; else if (qa == 3) w = mul_i(w);
        tw[j] = w;
    }
    for (int k = threadIdx.x; k <= 512; k += kThreads) { const float2 t = kWn[2 * k]; wn[k] = (v2f){t.x, t.y}; }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)n);
    const unsigned total = (unsigned)(a.batch * a.frames), frames = (unsigned)a.frames;
    const float2* win = reinterpret_cast<const float2*>(a.window);
    for (unsigned fr = blockIdx.x * 4 + wave; fr < total; fr += gridDim.x * 4) {
        const unsigned b = fr / frames, f = fr - b * frames;
        const float* src = (a.audio_b != nullptr && (int64_t)b >= a.split) ? a.audio_b + ((int64_t)b - a.split) * a.row_stride_b
                                                                          : a.audio + (int64_t)b * a.row_stride;
        const int64_t t0 = (int64_t)f * a.hop;
        const bool inside = t0 + n <= a.samples;
        v2f r[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = 64 * q + lane;
            const int64_t t = t0 + 2 * i;
            const float2 w = win[i];
            float v0, v1;
            if (inside) { v0 = src[t] * w.x; v1 = src[t + 1] * w.y; }
            else { v0 = (t < a.samples) ? src[t] * w.x : 0.0f; v1 = (t + 1 < a.samples) ? src[t + 1] * w.y : 0.0f; }
            r[q] = (v2f){v0, v1};
        }
        fft1024_wave<false>(r, zl, tw, lane);
        slot_sync<true>();
        float* dst = a.mag + (int64_t)fr * nb;
        float2* sp = (a.spec != nullptr && (int64_t)b >= a.spec_first) ? a.spec + ((int64_t)fr - a.spec_first * frames) * nb : nullptr;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int k = lane + 64 * j;
            if (k <= m / 2) {
                v2f xk, xm;
                unpack_pair<LOGM>(zl, wn, k, xk, xm);
                store_mag(dst, k, magnitude(xk) * scale);
                store_mag(dst, m - k, magnitude(xm) * scale);
                if (sp != nullptr) store_spec(sp, k, m - k, xk, xm);
            }
        }
        slot_sync<true>();   // the unpack reads are issued before the next frame's exchange writes
    }
}

// ---------------------------------------------------------------------------------------------
// Round 4: the one-wavefront-per-frame transform again, rebuilt around occupancy (stft_mag_forward_wave2_kernel, n_fft 2048).
// What the round-2 kernel above paid for: 194 VGPRs (two waves per SIMD) and three quarters of a CU's LDS per four frames.  Here
//  * ONE 1024-thread workgroup per CU: sixteen waves = sixteen frames in flight per CU share one twiddle table (W_1024^j, 8 KB) and one
//    copy of the window (8 KB); 159.8 KB of LDS; the 128-VGPR budget of a 1024-thread workgroup is met by doing every radix-4 stage IN
//    PLACE on the sixteen points a lane holds (no second 16-point array);
//  * the frame's |.| values take magnitude_plain() after ONE range test per frame on the windowed samples (frame_is_plain) instead of a
//    range test, a wave vote and libm's sqrtf per value;
//  * samples and taps are fetched as 8-byte pairs (the clip's row and hop keep frames 8-byte aligned in every reference configuration; a
//    scalar path covers the rest).
// No workgroup barrier inside the frame loop: a wave's exchanges through its private LDS buffer need only wave-level ordering.
// ---------------------------------------------------------------------------------------------
#ifndef SOT_STFT_XCHG_SWZ
#define SOT_STFT_XCHG_SWZ 1
#endif
template <bool INVERSE>
__device__ __forceinline__ v2f ctw(v2f a, v2f w) { return INVERSE ? cmul_conj(a, w) : cmul(a, w); }   // a * twiddle (inverse: conjugate twiddle)

template <bool INVERSE>
__device__ __forceinline__ void bf4_ip(v2f& a0, v2f& a1, v2f& a2, v2f& a3)
{
    const v2f s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13; a2 = s02 - s13;
    a1 = INVERSE ? add_pi(d02, d13) : add_mi(d02, d13);   // d02 + (-+i) d13
    a3 = INVERSE ? add_mi(d02, d13) : add_pi(d02, d13);   // d02 - (-+i) d13
}

// r[q] = z[64 q + lane] on entry; on return the transform sits in zl[zi(k)], k = 0 .. 1023 (after the caller's slot_sync).  The index
// algebra is fft1024_wave's (five radix-4 decimation-in-frequency stages, two exchanges), every stage in place.
template <bool INVERSE>
__device__ __forceinline__ void fft1024_wave_ip(v2f (&r)[16], v2f* zl, const v2f* tw, int lane)
{
    // stage 1 (digit d4; registers 4 p + d3 -> 4 q + d3): twiddle W_1024^{(64 d3 + lane) q}
#pragma unroll
    for (int d3 = 0; d3 < 4; ++d3) {
        const int j = 64 * d3 + lane;
        const v2f w1 = tw[j & 1023], w2 = tw[(2 * j) & 1023], w3 = tw[(3 * j) & 1023];   // requested ahead of the butterfly
        bf4_ip<INVERSE>(r[d3], r[4 + d3], r[8 + d3], r[12 + d3]);
        r[4 + d3] = ctw<INVERSE>(r[4 + d3], w1); r[8 + d3] = ctw<INVERSE>(r[8 + d3], w2); r[12 + d3] = ctw<INVERSE>(r[12 + d3], w3);
    }
    // stage 2 (digit d3; registers 4 q4 + p -> 4 q4 + q): twiddle W_256^{lane q} = W_1024^{4 lane q}
    {
        const v2f w1 = tw[(4 * lane) & 1023], w2 = tw[(8 * lane) & 1023], w3 = tw[(12 * lane) & 1023];   // the same three for every q4
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            bf4_ip<INVERSE>(r[4 * q4], r[4 * q4 + 1], r[4 * q4 + 2], r[4 * q4 + 3]);
            r[4 * q4 + 1] = ctw<INVERSE>(r[4 * q4 + 1], w1); r[4 * q4 + 2] = ctw<INVERSE>(r[4 * q4 + 2], w2); r[4 * q4 + 3] = ctw<INVERSE>(r[4 * q4 + 3], w3);
        }
    }
    // exchange 1: register (q4, q3) of lane (d2, d1, d0) -> register (d2, d1) of lane (q4, q3, d0): logical element (column 4 q + d0, row rr)
    // written by lane (rr, d0), row q of column L read by lane L.  Storage [column ^ g(row)][17]: the reader's lanes still sweep 64 different
    // columns per instruction (conflict-free as before), the writer's 16-lane groups no longer pile four lanes on one bank pair
    // (SQ_LDS_BANK_CONFLICT was 49 % of the kernel's LDS cycles, all of it these stores): g(row) = 4 (row & 3) here (two-way left),
    // g(row) = row >> 2 in exchange 2 (conflict-free).  SOT_STFT_XCHG_SWZ=0: the plain [column][17] image.
    const int d0 = lane & 3;
    {
        const int rr = lane >> 2;   // 4 d2 + d1
#pragma unroll
        for (int q = 0; q < 16; ++q) zl[((4 * q + d0) ^ (SOT_STFT_XCHG_SWZ ? 4 * (rr & 3) : 0)) * 17 + rr] = r[q];
        slot_sync<true>();
#pragma unroll
        for (int q = 0; q < 16; ++q) r[q] = zl[(lane ^ (SOT_STFT_XCHG_SWZ ? 4 * (q & 3) : 0)) * 17 + q];
        slot_sync<true>();
    }
    // stage 3 (digit d2; registers 4 p + d1 -> 4 q + d1): twiddle W_64^{(4 d1 + d0) q} = W_1024^{16 (4 d1 + d0) q}
#pragma unroll
    for (int d1 = 0; d1 < 4; ++d1) {
        const int j = 16 * (4 * d1 + d0);
        const v2f w1 = tw[j & 1023], w2 = tw[(2 * j) & 1023], w3 = tw[(3 * j) & 1023];
        bf4_ip<INVERSE>(r[d1], r[4 + d1], r[8 + d1], r[12 + d1]);
        r[4 + d1] = ctw<INVERSE>(r[4 + d1], w1); r[8 + d1] = ctw<INVERSE>(r[8 + d1], w2); r[12 + d1] = ctw<INVERSE>(r[12 + d1], w3);
    }
    // stage 4 (digit d1; registers 4 q2 + p -> 4 q2 + q): twiddle W_16^{d0 q} = W_1024^{64 d0 q}
    {
        const v2f w1 = tw[(64 * d0) & 1023], w2 = tw[(128 * d0) & 1023], w3 = tw[(192 * d0) & 1023];   // the same three for every q2
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
            bf4_ip<INVERSE>(r[4 * q2], r[4 * q2 + 1], r[4 * q2 + 2], r[4 * q2 + 3]);
            r[4 * q2 + 1] = ctw<INVERSE>(r[4 * q2 + 1], w1); r[4 * q2 + 2] = ctw<INVERSE>(r[4 * q2 + 2], w2); r[4 * q2 + 3] = ctw<INVERSE>(r[4 * q2 + 3], w3);
        }
    }
    // exchange 2: register (q2, q1) of lane (q4, q3, d0) -> register (d0, q1) of lane (q4, q3, q2)
    {
        const int hi = lane >> 2;   // 4 q4 + q3
#pragma unroll
        for (int q = 0; q < 16; ++q) zl[((4 * hi + (q >> 2)) ^ (SOT_STFT_XCHG_SWZ ? d0 : 0)) * 17 + 4 * d0 + (q & 3)] = r[q];   // row 4 d0 + (q & 3): g = d0
        slot_sync<true>();
#pragma unroll
        for (int q = 0; q < 16; ++q) r[q] = zl[(lane ^ (SOT_STFT_XCHG_SWZ ? (q >> 2) : 0)) * 17 + q];
        slot_sync<true>();
    }
    // stage 5 (digit d0; registers 4 p + q1 -> 4 q0 + q1), no twiddle; result (q4 q3 q2 q1 q0) is frequency k = q4 + 4 q3 + 16 q2 + 64 q1 + 256 q0
    const int kb = (lane >> 4) + 4 * ((lane >> 2) & 3) + 16 * (lane & 3);
#pragma unroll
    for (int q1 = 0; q1 < 4; ++q1) {
        bf4_ip<INVERSE>(r[q1], r[4 + q1], r[8 + q1], r[12 + q1]);
#pragma unroll
        for (int q0 = 0; q0 < 4; ++q0) zl[zi(kb + 64 * q1 + 256 * q0)] = r[4 * q0 + q1];
    }
}

__device__ __forceinline__ float wave_max_f32(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

constexpr int kWave2Threads = 1024;
constexpr int kWave2Waves = kWave2Threads / 64;
constexpr size_t kWave2LdsBytes = (1024 + 520 + 1024 + (size_t)kWave2Waves * kWaveBuf) * sizeof(float2);

// the frame's bins from the transform in LDS: |.| / sqrt(n) (and the complex spectrum on request); PLAIN: magnitude_plain(), else the careful form
template <bool PLAIN>
__device__ __forceinline__ void wave2_unpack_store(const v2f* zl, const v2f* wn, int lane, float scale, float* dst, float2* sp)
{
    constexpr int LOGM = 10, m = 1024;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int k = lane + 64 * j;
        if (j < 8 || k <= m / 2) {
            v2f xk, xm;
            unpack_pair<LOGM>(zl, wn, k, xk, xm);
            store_mag(dst, k, (PLAIN ? magnitude_plain(xk) : magnitude(xk)) * scale);
            store_mag(dst, m - k, (PLAIN ? magnitude_plain(xm) : magnitude(xm)) * scale);
            if (sp != nullptr) store_spec(sp, k, m - k, xk, xm);
        }
    }
}

__global__ __launch_bounds__(kWave2Threads) void stft_mag_forward_wave2_kernel(const StftArgs a)
{
    constexpr int m = 1024, n = 2048, nb = m + 1;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    v2f* const tw = reinterpret_cast<v2f*>(smem_f);          // W_1024^j, j < 1024
    v2f* const wn = tw + 1024;                                // W_2048^k, k <= 512 (+ pad)
    v2f* const wl = wn + 520;                                 // window taps (2 i, 2 i + 1), i < 1024
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;   // the wave number in a scalar register:
    v2f* const zl = wl + 1024 + wave * kWaveBuf;                                                        // frame, clip and row pointers stay scalar
    const float2* win = reinterpret_cast<const float2*>(a.window);
    const unsigned total = (unsigned)(a.batch * a.frames), frames = (unsigned)a.frames;
    const unsigned stride = gridDim.x * kWave2Waves;
    v2f r[16];
    auto fetch = [&](unsigned fr) {   // raw samples (2 i, 2 i + 1), i = 64 q + lane, of frame fr; zeros past the clip's end (utils.py:252-275)
        const unsigned b = fr / frames, f = fr - b * frames;
        const float* src = (a.audio_b != nullptr && (int64_t)b >= a.split) ? a.audio_b + ((int64_t)b - a.split) * a.row_stride_b
                                                                          : a.audio + (int64_t)b * a.row_stride;
        const int64_t t0 = (int64_t)f * a.hop;
        const float* const s0 = src + t0;
        const bool pairs = t0 + n <= a.samples && (reinterpret_cast<uintptr_t>(s0) & 7u) == 0;   // wave-uniform
        if (pairs) {
            const float2* const s2 = reinterpret_cast<const float2*>(s0);
#pragma unroll
            for (int q = 0; q < 16; ++q) { const float2 v = s2[64 * q + lane]; r[q] = (v2f){v.x, v.y}; }
        } else {
            const int left = (int)min((int64_t)n, a.samples - t0);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = 64 * q + lane;
                r[q] = (v2f){(2 * i < left) ? s0[2 * i] : 0.0f, (2 * i + 1 < left) ? s0[2 * i + 1] : 0.0f};
            }
        }
    };
    unsigned fr = blockIdx.x * kWave2Waves + wave;
    if (fr < total) fetch(fr);   // the first frame's samples are in flight while the tables are built
    for (int j = threadIdx.x; j < 1024; j += kWave2Threads) {   // W_1024^{256 a + b} = W_2048^{2 b} (-i)^a  (exact quarter turns of the committed table)
        const float2 t = kWn[4 * (j & 255)];
        v2f w = (v2f){t.x, t.y};
        const int qa = j >> 8;
        if (qa == 1) w = mul_mi(w); else if (qa == 2) w = -w; else if (qa == 3) w = mul_i(w);
        tw[j] = w;
        const float2 tap = win[j];
        wl[j] = (v2f){tap.x, tap.y};
    }
    for (int k = threadIdx.x; k <= 512; k += kWave2Threads) { const float2 t = kWn[2 * k]; wn[k] = (v2f){t.x, t.y}; }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)n);
    for (; fr < total; fr += stride) {
        // the lane number as an opaque value per frame: hipcc otherwise hoists every lane-derived LDS address out of this (two-trip) loop
        // and pays for the ~40 live registers with spills (23 dwords at the 128-register budget of a 1024-thread workgroup)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        float amax = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            r[q] = r[q] * wl[64 * q + ln];
            amax = fmaxf(amax, fmaxf(fabsf(r[q].x), fabsf(r[q].y)));
        }
        // one range test per FRAME (NaN samples fail it and take the careful path); a scalar, so the two forms below are two branches
        const bool plain = __builtin_amdgcn_readfirstlane((int)frame_is_plain(wave_max_f32(amax))) != 0;
        fft1024_wave_ip<false>(r, zl, tw, ln);
        if (fr + stride < total) fetch(fr + stride);   // the next frame's samples arrive while this one is unpacked
        slot_sync<true>();
        const unsigned b = fr / frames;
        float* dst = a.mag + (int64_t)fr * nb;
        float2* sp = (a.spec != nullptr && (int64_t)b >= a.spec_first) ? a.spec + ((int64_t)fr - a.spec_first * frames) * nb : nullptr;
        if (plain) wave2_unpack_store<true>(zl, wn, ln, scale, dst, sp);
        else wave2_unpack_store<false>(zl, wn, ln, scale, dst, sp);
        slot_sync<true>();   // the unpack reads are issued before the next frame's exchange writes
    }
}

// Backward.  With Zin_k = g_k X_k / |X_k| (k = 0 .. m) the gradient of the windowed frame is
//   y_i = Re(sum_{k=0}^{m} Zin_k e^{+2 pi i k i / n}) / sqrt(n),
// i.e. the (unnormalised) inverse real transform of the Hermitian spectrum H_k = Zin_k / 2 (0 < k < m), H_0 = Re Zin_0,
// H_m = Re Zin_m, again through a half-length complex transform:  G_k = (H_k + conj(H_{m-k})) + i conj(W) (H_k - conj(H_{m-k})),
// g = IFFT_m(G) (no 1/m), y_{2i} = Re g_i, y_{2i+1} = Im g_i.
// Pass 1 (this kernel): one frame slot per group of kFramesPerGroup consecutive frames; their windowed gradients are
// overlap-added in LDS and stored as the group's partial result.  Pass 2 (stft_overlap_add_kernel) adds, per sample, the
// partial results of the groups that cover it in ascending group order: deterministic, no atomics.
// SPEC: the frames' complex spectra come from the forward pass (StftArgs::spec_in: what the forward kernels store on request, bit for
// bit the X this kernel would recompute): no audio is read and the forward transform of every frame is skipped -- the kernel is
// the magnitude / phase arithmetic, ONE (inverse) transform per frame and the overlap-add.
template <int LOGM, bool SPEC>
__device__ __forceinline__ void backward_partial_body(const StftArgs& a)
{
    using G = Geo<LOGM>;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int m = G::m;
    const int slot = threadIdx.x / G::tpf, lid = threadIdx.x - slot * G::tpf;
    constexpr int kPairIters = (m / 2 + 1 + G::tpf - 1) / G::tpf;   // m/2 + 1 pairs (k, m-k) over the slot's threads (3 at m/4 threads)
    v2f* const zall = reinterpret_cast<v2f*>(smem_f);
    v2f* const z = zall + slot * G::zpoints;
    v2f* const tw = zall + G::slots * G::zpoints;
    v2f* const wn = tw + m;
    float* const acc = reinterpret_cast<float*>(zall + G::lds_points) + slot * a.span;  // this group's overlap-added gradient [span]
    const float2* win = reinterpret_cast<const float2*>(a.window);
    const float scale = 1.0f / sqrtf((float)G::n);
    const float up = a.grad_scale ? *a.grad_scale : 1.0f;
    const unsigned total = (unsigned)(a.batch * a.groups), groups = (unsigned)a.groups;
    constexpr int PER = m / G::tpf;
    float2 wt[PER];   // this thread's window taps: used twice per frame (analysis and synthesis side)
#pragma unroll
    for (int j = 0; j < PER; ++j) wt[j] = win[lid + j * G::tpf];
    load_tables<LOGM>(tw, wn);
    if constexpr (G::wave_sync) __syncthreads();
    const unsigned w = blockIdx.x * G::slots + slot;
    const bool active = w < total;
    const unsigned b = active ? w / groups : 0u, grp = active ? w - b * groups : 0u;
    const float* src = a.audio + (int64_t)b * a.row_stride;
    const int64_t f_begin = (int64_t)grp * kFramesPerGroup;
    float2 cur[PER];   // raw samples of the frame about to be transformed (fetched while the frame before it is processed)
    auto fetch_audio = [&](int64_t f) {
        const int64_t t0 = f * a.hop;
        const float* const s0 = src + t0;
        const int left = (active && f < a.frames) ? (int)min((int64_t)G::n, a.samples - t0) : 0;   // zeros past the clip (utils.py:252-275)
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = lid + j * G::tpf;
            cur[j].x = (2 * i < left) ? s0[2 * i] : 0.0f;
            cur[j].y = (2 * i + 1 < left) ? s0[2 * i + 1] : 0.0f;
        }
    };
    if constexpr (!SPEC) fetch_audio(f_begin);
    for (int t = lid; t < a.span; t += G::tpf) acc[t] = 0.0f;
    for (int fi = 0; fi < kFramesPerGroup; ++fi) {
        const int64_t f = f_begin + fi;
        const bool has = active && f < a.frames;   // idle slots / missing frames run the same passes on zeros
        slot_sync<G::wave_sync>();
        if constexpr (!SPEC) {
#pragma unroll
            for (int j = 0; j < PER; ++j)
                z[zi(bitrev(lid + j * G::tpf, LOGM))] = (v2f){cur[j].x * wt[j].x, cur[j].y * wt[j].y};
            if (fi + 1 < kFramesPerGroup) fetch_audio(f + 1);
        }
        // this frame's upstream gradients: requested before the transform, consumed after it
        const float* g = a.grad_mag + ((int64_t)b * a.frames + (has ? f : 0)) * G::nb;
        float gup_k[kPairIters], gup_m[kPairIters];
#pragma unroll
        for (int r = 0; r < kPairIters; ++r) {
            const int k = lid + r * G::tpf;
            const bool use = has && k <= m / 2;
            gup_k[r] = use ? g[k] : 0.0f;
            gup_m[r] = use ? g[m - k] : 0.0f;
        }
        // the frame's spectrum: recomputed here, or the one the forward pass stored
        const float2* sp = SPEC ? a.spec_in + ((int64_t)b * a.frames + (has ? f : 0)) * G::nb : nullptr;
        v2f sxk[SPEC ? kPairIters : 1], sxm[SPEC ? kPairIters : 1];
        if constexpr (SPEC) {
#pragma unroll
            for (int r = 0; r < kPairIters; ++r) {
                const int k = lid + r * G::tpf;
                const bool use = has && k <= m / 2;
                const float2 pk = use ? sp[k] : make_float2(0.0f, 0.0f), pm = use ? sp[m - k] : make_float2(0.0f, 0.0f);
                sxk[r] = (v2f){pk.x, pk.y}; sxm[r] = (v2f){pm.x, pm.y};
            }
        } else {
            fft_inplace<LOGM, false>(z, tw, lid);
        }
        // pairs (k, m-k): spectrum -> Zin -> H -> G, kept in registers until every thread has read z
        v2f gk[kPairIters], gm[kPairIters];
#pragma unroll
        for (int r = 0; r < kPairIters; ++r) {
            const int k = lid + r * G::tpf;
            gk[r] = (v2f){0.0f, 0.0f}; gm[r] = (v2f){0.0f, 0.0f};
            if (has && k <= m / 2) {
                v2f xk, xm;
                if constexpr (SPEC) { xk = sxk[r]; xm = sxm[r]; }
                else unpack_pair<LOGM>(z, wn, k, xk, xm);
                const float mk = magnitude(xk), mm = magnitude(xm);
                const float ck = mk > 0.0f ? (gup_k[r] * up) / mk : 0.0f;      // torch: sgn(0) = 0
                const float cm = mm > 0.0f ? (gup_m[r] * up) / mm : 0.0f;
                v2f hk = (0.5f * ck) * xk, hm = (0.5f * cm) * xm;
                if (k == 0) { hk = (v2f){ck * xk.x, 0.0f}; hm = (v2f){cm * xm.x, 0.0f}; }   // H_0, H_m are real
                const v2f sk = hk + cconj(hm);      // H_k + conj(H_{m-k})
                const v2f dk = hk - cconj(hm);      // H_k - conj(H_{m-k})
                const v2f wk = wn[k];
                gk[r] = sk + mul_i(cmul(cconj(wk), dk));                 // s + i conj(W) d
                // G_{m-k} = (H_{m-k} + conj(H_k)) + i (-W) (H_{m-k} - conj(H_k)) = conj(s) + i W conj(d)
                gm[r] = cconj(sk) + mul_i(cmul(wk, cconj(dk)));
            }
        }
        if constexpr (!SPEC) slot_sync<G::wave_sync>();   // (SPEC: nobody has read z since the barrier at the top of the frame)
#pragma unroll
        for (int r = 0; r < kPairIters; ++r) {
            const int k = lid + r * G::tpf;
            if (k <= m / 2) {
                z[zi(bitrev(k, LOGM))] = gk[r];
                if (k > 0 && k < m - k) z[zi(bitrev(m - k, LOGM))] = gm[r];
            }
        }
        fft_inplace<LOGM, true>(z, tw, lid);
        if (has) {
            const int off = fi * a.hop;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int i = lid + j * G::tpf;
                const v2f v = z[zi(i)];
                acc[off + 2 * i] += wt[j].x * v.x * scale;
                acc[off + 2 * i + 1] += wt[j].y * v.y * scale;
            }
        }
    }
    slot_sync<G::wave_sync>();
    if (active) {
        float* dst = a.partial + (int64_t)w * a.span;
        for (int t = lid; t < a.span; t += G::tpf) dst[t] = acc[t];
    }
}

template <int LOGM>
__global__ __launch_bounds__(kThreads) void stft_mag_backward_partial_kernel(const StftArgs a) { backward_partial_body<LOGM, false>(a); }

template <int LOGM>
__global__ __launch_bounds__(kThreads) void stft_mag_backward_spec_kernel(const StftArgs a) { backward_partial_body<LOGM, true>(a); }

// ---------------------------------------------------------------------------------------------
// Round 4: the backward from the stored spectrum with ONE WAVEFRONT per frame group (n_fft 2048, hop a multiple of 128; the counterpart
// of stft_mag_forward_wave2_kernel).  A wave takes the kFramesPerGroup = 2 consecutive frames of its group one after the other: spectrum
// and upstream gradient of the lane's nine bin pairs (k, m - k) -> Zin -> Hermitian packing G (as backward_partial_body, with
// 1 / |X| = rsq(re^2 + im^2) after one range test per frame instead of a careful |.| and an IEEE division per value) -> G to the wave's LDS
// buffer in natural order -> the 16 points 64 q + lane into registers -> fft1024_wave_ip<true> -> windowed, scaled, and overlap-added IN
// REGISTERS: the second frame starts QS = hop / 128 register slots behind the first (its packed point i sits at point i + hop / 2 of the
// group's span), same lane.  The group's span leaves as 16 + QS coalesced 8-byte stores into the scratch buffer stft_overlap_add_kernel
// reads (same layout as the slot kernel's).  512-thread workgroups: 8 frame groups in flight per CU (tables 20 KB + 8 x 8.7 KB of LDS).
// ---------------------------------------------------------------------------------------------
constexpr int kBwdWave2Threads = 512;
constexpr int kBwdWave2Waves = kBwdWave2Threads / 64;
constexpr size_t kBwdWave2LdsBytes = (1024 + 520 + 1024 + (size_t)kBwdWave2Waves * kWaveBuf) * sizeof(float2);

// tables of the one-wave kernels: W_1024^j, W_2048^k (k <= 512), the window as tap pairs; all threads of the workgroup, then a barrier
template <int THREADS>
__device__ __forceinline__ void wave2_load_tables(const StftArgs& a, v2f* tw, v2f* wn, v2f* wl)
{
    const float2* win = reinterpret_cast<const float2*>(a.window);
    for (int j = threadIdx.x; j < 1024; j += THREADS) {   // W_1024^{256 a + b} = W_2048^{2 b} (-i)^a  (exact quarter turns of the committed table)
        const float2 t = kWn[4 * (j & 255)];
        v2f w = (v2f){t.x, t.y};
        const int qa = j >> 8;
        if (qa == 1) w = mul_mi(w); else if (qa == 2) w = -w; else if (qa == 3) w = mul_i(w);
        tw[j] = w;
        const float2 tap = win[j];
        wl[j] = (v2f){tap.x, tap.y};
    }
    for (int k = threadIdx.x; k <= 512; k += THREADS) { const float2 t = kWn[2 * k]; wn[k] = (v2f){t.x, t.y}; }
    __syncthreads();
}

// The Hermitian packing G of one frame's Zin_k = g_k X_k / |X_k| into the wave's LDS buffer (natural order), pair by pair -- nothing is held
// across pairs.  PLAIN: 1 / |X| = rsq(re^2 + im^2); the pass also returns the frame's largest and smallest non-zero component magnitude,
// from which the caller decides whether PLAIN was legitimate (and repeats the pass in the careful form if not: rare).
template <bool PLAIN>
__device__ __forceinline__ void wave2_pack_gradient(const float2* __restrict__ sp, const float* __restrict__ g, float up, v2f* zl, const v2f* wn,
                                                    int lane, float& peak, float& least)
{
    constexpr int m = 1024;
    peak = 0.0f; least = INFINITY;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int k = lane + 64 * j;
        if (j < 8 || k <= m / 2) {
            const float2 pk = sp[k], pm = sp[m - k];
            const v2f xk = (v2f){pk.x, pk.y}, xm = (v2f){pm.x, pm.y};
            const float gk_up = g[k], gm_up = g[m - k];
            float ck, cm;
            if (PLAIN) {
                const float ak = fmaxf(fabsf(pk.x), fabsf(pk.y)), am = fmaxf(fabsf(pm.x), fabsf(pm.y));
                peak = fmaxf(peak, fmaxf(ak, am));
                least = fminf(least, fminf(ak > 0.0f ? ak : INFINITY, am > 0.0f ? am : INFINITY));
                const float sk2 = fmaf(xk.x, xk.x, xk.y * xk.y), sm2 = fmaf(xm.x, xm.x, xm.y * xm.y);
                ck = sk2 > 0.0f ? (gk_up * up) * __builtin_amdgcn_rsqf(sk2) : 0.0f;      // torch: sgn(0) = 0
                cm = sm2 > 0.0f ? (gm_up * up) * __builtin_amdgcn_rsqf(sm2) : 0.0f;
            } else {
                const float mk = magnitude(xk), mm = magnitude(xm);
                ck = mk > 0.0f ? (gk_up * up) / mk : 0.0f;
                cm = mm > 0.0f ? (gm_up * up) / mm : 0.0f;
            }
            v2f hk = (0.5f * ck) * xk, hm = (0.5f * cm) * xm;
            if (k == 0) { hk = (v2f){ck * xk.x, 0.0f}; hm = (v2f){cm * xm.x, 0.0f}; }   // H_0, H_m are real
            const v2f sk = hk + cconj(hm);      // H_k + conj(H_{m-k})
            const v2f dk = hk - cconj(hm);      // H_k - conj(H_{m-k})
            const v2f wk = wn[k];
            zl[k] = sk + mul_i(cmul(cconj(wk), dk));                                   // G_k = s + i conj(W) d
            if (k > 0 && k < m - k) zl[m - k] = cconj(sk) + mul_i(cmul(wk, cconj(dk)));   // G_{m-k} = conj(s) + i W conj(d)
        }
    }
}

// One frame of the backward from the stored spectrum on one wavefront: out[q] = tap * (inverse transform of the Hermitian packing of
// Zin_k = g_k X_k / |X_k|) * scale at the packed points 64 q + lane, i.e. the frame's windowed audio gradient (samples 2 i, 2 i + 1).
// sp / g: the frame's spectrum and upstream gradient rows.  Ends with a wave-level ordering point (zl may be reused).
__device__ __forceinline__ void wave2_frame_gradient(const float2* __restrict__ sp, const float* __restrict__ g, float up, float scale, v2f* zl,
                                                     const v2f* tw, const v2f* wn, const v2f* wl, int lane_in, v2f (&out)[16])
{
    int lane = lane_in;
    asm volatile("" : "+v"(lane));   // lane-derived LDS addresses are recomputed per frame instead of living in registers across frames
    float peak, least;
    wave2_pack_gradient<true>(sp, g, up, zl, wn, lane, peak, least);
    // re^2 + im^2 of every non-zero bin must be a normal number that cannot overflow (NaNs fail the test): else the careful form
    const float wpeak = wave_max_f32(peak), wleast = -wave_max_f32(-least);
    if (__builtin_amdgcn_readfirstlane((int)(wpeak < 1e15f && wleast > 1e-18f)) == 0) {
        slot_sync<true>();
        wave2_pack_gradient<false>(sp, g, up, zl, wn, lane, peak, least);
    }
    slot_sync<true>();
    v2f r[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) r[q] = zl[64 * q + lane];
    slot_sync<true>();   // every lane holds its points before the transform's exchanges reuse the buffer
    fft1024_wave_ip<true>(r, zl, tw, lane);
    slot_sync<true>();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const v2f v = zl[zi(64 * q + lane)], tap = wl[64 * q + lane];
        out[q] = (v2f){tap.x * v.x * scale, tap.y * v.y * scale};
    }
    slot_sync<true>();   // the reads are issued before the buffer is written again
}

template <int QS>
__global__ __launch_bounds__(kBwdWave2Threads) void stft_mag_backward_spec_wave2_kernel(const StftArgs a)
{
    static_assert(kFramesPerGroup == 2, "two frames per wave: 16 + QS register slots");
    constexpr int n = 2048, nb = 1025;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    v2f* const tw = reinterpret_cast<v2f*>(smem_f);          // W_1024^j, j < 1024
    v2f* const wn = tw + 1024;                                // W_2048^k, k <= 512 (+ pad)
    v2f* const wl = wn + 520;                                 // window taps (2 i, 2 i + 1), i < 1024
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    v2f* const zl = wl + 1024 + wave * kWaveBuf;
    wave2_load_tables<kBwdWave2Threads>(a, tw, wn, wl);
    const float scale = 1.0f / sqrtf((float)n);
    const float up = a.grad_scale ? *a.grad_scale : 1.0f;
    const unsigned total = (unsigned)(a.batch * a.groups), groups = (unsigned)a.groups;
    for (unsigned w = blockIdx.x * kBwdWave2Waves + wave; w < total; w += gridDim.x * kBwdWave2Waves) {
        const unsigned b = w / groups, grp = w - b * groups;
        v2f acc[16 + QS];
#pragma unroll
        for (int q = 0; q < 16 + QS; ++q) acc[q] = (v2f){0.0f, 0.0f};
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) {
            const int64_t f = (int64_t)grp * 2 + fi;
            if (f < a.frames) {   // wave-uniform
                v2f out[16];
                wave2_frame_gradient(a.spec_in + ((int64_t)b * a.frames + f) * nb, a.grad_mag + ((int64_t)b * a.frames + f) * nb, up, scale, zl, tw, wn,
                                     wl, lane, out);
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q + QS * fi] += out[q];
            }
        }
        float2* dst = reinterpret_cast<float2*>(a.partial + (int64_t)w * a.span);
#pragma unroll
        for (int q = 0; q < 16 + QS; ++q) dst[64 * q + lane] = make_float2(acc[q].x, acc[q].y);
    }
}

// ---------------------------------------------------------------------------------------------
// The same with the overlap-add INSIDE the kernel, for clips of at most 16 frames (config 5 and the paper's step: 4096 samples, hop 256):
// one 1024-thread workgroup per clip, wave f transforms frame f and leaves the windowed frame gradient in its own LDS buffer; after ONE
// workgroup barrier every thread adds, for its two packed points of the clip, the frames that cover them in ascending frame order
// (deterministic) and stores the clip's gradient: no scratch buffer, no second kernel (stft_overlap_add_kernel: 7 us at 256 clips).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave2Threads) void stft_mag_backward_spec_clip_kernel(const StftArgs a)
{
    constexpr int n = 2048, nb = 1025, m = 1024;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    v2f* const tw = reinterpret_cast<v2f*>(smem_f);
    v2f* const wn = tw + 1024;
    v2f* const wl = wn + 520;
    v2f* const bufs = wl + 1024;                              // 16 frame buffers of kWaveBuf points
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    v2f* const zl = bufs + wave * kWaveBuf;
    wave2_load_tables<kWave2Threads>(a, tw, wn, wl);
    const float scale = 1.0f / sqrtf((float)n);
    const float up = a.grad_scale ? *a.grad_scale : 1.0f;
    const int frames = (int)a.frames, hp = a.hop >> 1;        // hop in packed points (hop is even: host)
    const int samples = (int)a.samples;
    for (int64_t b = blockIdx.x; b < a.batch; b += gridDim.x) {
        if (wave < frames) {
            v2f out[16];
            wave2_frame_gradient(a.spec_in + (b * a.frames + wave) * nb, a.grad_mag + (b * a.frames + wave) * nb, up, scale, zl, tw, wn, wl, lane, out);
#pragma unroll
            for (int q = 0; q < 16; ++q) zl[zi(64 * q + lane)] = out[q];
        }
        __syncthreads();
        float* const dst = a.grad_audio + b * a.samples;
        for (int p = threadIdx.x; 2 * p < samples; p += kWave2Threads) {   // packed point p of the clip = samples 2 p, 2 p + 1
            const int f_hi = min(p / hp, frames - 1);
            int f_lo = (p - (m - 1) + hp - 1) / hp;
            if (p - (m - 1) <= 0) f_lo = 0;
            v2f sum = (v2f){0.0f, 0.0f};
            for (int f = f_lo; f <= f_hi; ++f) sum += bufs[f * kWaveBuf + zi(p - f * hp)];
            if (2 * p + 1 < samples) {
                float2* d2 = reinterpret_cast<float2*>(dst + 2 * p);
                if ((reinterpret_cast<uintptr_t>(d2) & 7u) == 0) {
                    float2 o = make_float2(sum.x, sum.y);
                    if (a.accumulate) { const float2 old = *d2; o.x += old.x; o.y += old.y; }
                    *d2 = o;
                } else {
                    dst[2 * p] = a.accumulate ? dst[2 * p] + sum.x : sum.x;
                    dst[2 * p + 1] = a.accumulate ? dst[2 * p + 1] + sum.y : sum.y;
                }
            } else {
                dst[2 * p] = a.accumulate ? dst[2 * p] + sum.x : sum.x;
            }
        }
        __syncthreads();   // the frame buffers are read before the next clip overwrites them
    }
}

// ---------------------------------------------------------------------------------------------
// Round 5: n_fft 2048 on the one-wavefront FFT of csrc/sot_wave_fft.hpp (written for the two-launch MSSLoss: 16 points per lane, radix-4 stages
// in registers, padded additive exchange maps, fused complex products, a transposed network for the inverse -- a frame's forward transform is
// ~415 instructions and ~4 000 clocks on one wave against the 28 000-clock chain of the round-4 wave kernels).
//  * stft_mag_backward_spec_clipw_kernel: the backward from the stored spectrum with the overlap-add inside (clips of at most 16 frames),
//    same structure as stft_mag_backward_spec_clip_kernel: wave f of a 1024-thread workgroup turns frame f's spectrum and upstream gradient
//    into the Hermitian packing G (natural order in its LDS buffer), runs the inverse network and leaves the windowed frame gradient in its
//    buffer; after one barrier the workgroup adds the frames that cover each sample in ascending frame order.
//  * stft_mag_forward_wavew_kernel: forward (single or pair form, optional spectrum) with one wavefront per frame in 512-thread workgroups,
//    for the batches BELOW the round-4 wave kernel's threshold (512 ... 3071 frames: the paper's 64 clips), where the slot kernel ran.
// ---------------------------------------------------------------------------------------------
// Diagnostic build only (-DSTFT_STAMPS; tools/r5/stft_stamps.py): wave 0 of workgroups 0 .. 63 of the wavew forward kernel (slots 0-4) and of the clipw backward kernel (slots 8-14) stamps the shader clock
#ifdef STFT_STAMPS
__device__ unsigned long long g_stft_stamps[64 * 16];
#define STFT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 64) { __builtin_amdgcn_sched_barrier(0); g_stft_stamps[blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define STFT_STAMP(i) do { } while (0)
#endif

constexpr size_t kClipwLdsBytes = ((size_t)16 * sot_wfft::kBuf + sot_wfft::kTw + sot_wfft::kWnMax) * sizeof(float2);
constexpr int kFwdwThreads = 256, kFwdwWaves = 4;   // 51.2 KB of LDS per workgroup (tables 16.4 KB + 8.7 KB per wave): three per CU; a 1088-frame launch reaches every CU
constexpr size_t kFwdwLdsBytes = ((size_t)kFwdwWaves * sot_wfft::kBuf + sot_wfft::kTw + sot_wfft::kWnMax) * sizeof(float2);

// Hermitian packing G of Zin_k = g_k X_k / |X_k| for the lane's bin pairs (k, m - k), k = 64 q + lane (q < 8) and k = m / 2 (lane 0), written to the
// wave's buffer in natural order (csrc/sot_mss.hip: pair_pass, here from the stored spectrum).  PLAIN: 1 / |X| = rsq(re^2 + im^2); the pass also
// returns the largest and the smallest non-zero component magnitude, from which the caller decides whether PLAIN was legitimate.
template <bool PLAIN>
__device__ __forceinline__ void clipw_pack_gradient(const float2* __restrict__ sp, const float* __restrict__ g, float up, sot_wfft::v2f* zl,
                                                    const sot_wfft::v2f* wn, int lane, float& peak, float& least)
{
    using namespace sot_wfft;
    constexpr int m = 1024;
    peak = 0.0f; least = INFINITY;
    // every load of the pass first (34 per lane; bin m / 2 by all lanes: one address): ONE round trip.  Loads inside the pair loop were waited
    // for pair by pair -- nine serial round trips, 10 000 of a wave's 29 000 clocks (tools/r5/stft_stamps.py)
    float2 lpk[9], lpm[9];
    float lgk[9], lgm[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int k = (q < 8) ? 64 * q + lane : m / 2;
        lpk[q] = sp[k]; lpm[q] = sp[m - k];
        lgk[q] = g[k]; lgm[q] = g[m - k];
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        if (q == 8 && lane != 0) break;
        const int k = (q < 8) ? 64 * q + lane : m / 2;
        const float2 pk = lpk[q], pm = lpm[q];
        const sot_wfft::v2f xk = (sot_wfft::v2f){pk.x, pk.y}, xm = (sot_wfft::v2f){pm.x, pm.y};
        const float gk = lgk[q] * up, gm = lgm[q] * up;
        float ck, cm;
        if (PLAIN) {
            const float ak = fmaxf(fabsf(pk.x), fabsf(pk.y)), am = fmaxf(fabsf(pm.x), fabsf(pm.y));
            peak = fmaxf(peak, fmaxf(ak, am));
            least = fminf(least, fminf(ak > 0.0f ? ak : INFINITY, am > 0.0f ? am : INFINITY));
            const float sk2 = fmaf(xk.x, xk.x, xk.y * xk.y), sm2 = fmaf(xm.x, xm.x, xm.y * xm.y);
            ck = sk2 > 0.0f ? gk * __builtin_amdgcn_rsqf(sk2) : 0.0f;      // torch: sgn(0) = 0
            cm = sm2 > 0.0f ? gm * __builtin_amdgcn_rsqf(sm2) : 0.0f;
        } else {
            const float mk = hypotf(xk.x, xk.y), mm = hypotf(xm.x, xm.y);
            ck = mk > 0.0f ? gk / mk : 0.0f;
            cm = mm > 0.0f ? gm / mm : 0.0f;
        }
        if (q == 0 && k == 0) { ck *= 2.0f; cm *= 2.0f; }                  // H_0 and H_m are the (real) Zin themselves, not halves
        const sot_wfft::v2f hk = ck * xk;                                   // 2 H_k
        const sot_wfft::v2f hc = (q < 8) ? cm * cconj(xm) : cconj(hk);     // 2 conj H_(m-k)
        const sot_wfft::v2f sk = 0.5f * (hk + hc), dd = hk - hc;
        const sot_wfft::v2f P = cmul_conj(dd, wn[k]);                      // i conj(W) (H_k - conj H_(m-k))
        zl[k] = sk + P;
        if (q < 8 && k != 0) zl[m - k] = conj_sub(sk, P);
    }
}

__global__ __launch_bounds__(kWave2Threads) void stft_mag_backward_spec_clipw_kernel(const StftArgs a)
{
    using namespace sot_wfft;
    constexpr int n = 2048, nb = 1025, m = 1024;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    sot_wfft::v2f* const tw = reinterpret_cast<sot_wfft::v2f*>(smem_f);
    sot_wfft::v2f* const wn = tw + kTw;
    sot_wfft::v2f* const bufs = wn + kWnMax;                   // 16 frame buffers of kBuf points
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    sot_wfft::v2f* const zl = bufs + wave * kBuf;
    STFT_STAMP(8);
    build_tables<kWave2Threads>(kWn, tw, wn);
    __syncthreads();
    STFT_STAMP(9);
    const float scale = 1.0f / sqrtf((float)n);
    const float up = a.grad_scale ? *a.grad_scale : 1.0f;
    const int frames = (int)a.frames, hp = a.hop >> 1;        // hop in packed points (hop is even: host)
    const int samples = (int)a.samples;
    const float2* const win2 = reinterpret_cast<const float2*>(a.window);
    for (int64_t b = blockIdx.x; b < a.batch; b += gridDim.x) {
        if (wave < frames) {
            int lane = threadIdx.x & 63;
            asm volatile("" : "+v"(lane));   // lane-derived LDS addresses are recomputed per clip instead of living in registers across clips
            const float2* const sp = a.spec_in + (b * a.frames + wave) * nb;
            const float* const g = a.grad_mag + (b * a.frames + wave) * nb;
            float peak, least;
            clipw_pack_gradient<true>(sp, g, up, zl, wn, lane, peak, least);
            // re^2 + im^2 of every non-zero bin must be a normal number that cannot overflow: else the careful form.  (A NaN bin does NOT fail
            // this test -- the maxima drop NaN operands -- and needs no care: it propagates through either form.)
            const float wpeak = wave_max_f32(peak), wleast = -wave_max_f32(-least);
            if (__builtin_amdgcn_readfirstlane((int)(wpeak < 1e15f && wleast > 1e-18f)) == 0) {
                wave_sync();
                clipw_pack_gradient<false>(sp, g, up, zl, wn, lane, peak, least);
            }
            wave_sync();
            STFT_STAMP(10);
            sot_wfft::v2f r[16], wt[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) r[q] = zl[lane + 64 * brev4(q)];
#pragma unroll
            for (int q = 0; q < 16; ++q) { const float2 wv = win2[64 * q + lane]; wt[q] = (sot_wfft::v2f){wv.x, wv.y}; }   // arrive during the transform
            wave_sync();
            inverse_transform<10>(r, zl, tw, lane);
            STFT_STAMP(11);
#pragma unroll
            for (int q = 0; q < 16; ++q) zl[64 * q + lane] = (wt[q] * r[q]) * scale;      // the windowed frame gradient, packed point 64 q + lane
        }
        __syncthreads();
        STFT_STAMP(12);
        float* const dst = a.grad_audio + b * a.samples;
        for (int p = threadIdx.x; 2 * p < samples; p += kWave2Threads) {   // packed point p of the clip = samples 2 p, 2 p + 1
            const int f_hi = min(p / hp, frames - 1);
            int f_lo = (p - (m - 1) + hp - 1) / hp;
            if (p - (m - 1) <= 0) f_lo = 0;
            sot_wfft::v2f sum = (sot_wfft::v2f){0.0f, 0.0f};
            for (int f = f_lo; f <= f_hi; ++f) sum += bufs[f * kBuf + (p - f * hp)];
            if (2 * p + 1 < samples) {
                float2* d2 = reinterpret_cast<float2*>(dst + 2 * p);
                if ((reinterpret_cast<uintptr_t>(d2) & 7u) == 0) {
                    float2 o = make_float2(sum.x, sum.y);
                    if (a.accumulate) { const float2 old = *d2; o.x += old.x; o.y += old.y; }
                    *d2 = o;
                } else {
                    dst[2 * p] = a.accumulate ? dst[2 * p] + sum.x : sum.x;
                    dst[2 * p + 1] = a.accumulate ? dst[2 * p + 1] + sum.y : sum.y;
                }
            } else {
                dst[2 * p] = a.accumulate ? dst[2 * p] + sum.x : sum.x;
            }
        }
        STFT_STAMP(13);
        __syncthreads();   // the frame buffers are read before the next clip overwrites them
        STFT_STAMP(14);
    }
}

// the frame's bins from the packed transform in the wave's buffer (natural order, Z_0 again at slot m): |.| / sqrt(n) (and the complex spectrum
// on request); PLAIN: magnitude_plain(), else the careful form
template <bool PLAIN>
__device__ __forceinline__ void wavew_unpack_store(const sot_wfft::v2f* zl, const sot_wfft::v2f* wn, int lane, float scale, float* dst, float2* sp)
{
    using namespace sot_wfft;
    constexpr int m = 1024;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        if (q == 8 && lane != 0) break;
        const int k = (q < 8) ? 64 * q + lane : m / 2;
        const sot_wfft::v2f zk = zl[k], zm = zl[m - k];                  // (k = 0: slot m holds the copy of Z_0)
        const sot_wfft::v2f e = add_conj(zk, zm), o = sub_conj(zk, zm);
        const sot_wfft::v2f wz = sot_wfft::cmul(o, wn[k]), eh = 0.5f * e;
        const sot_wfft::v2f xk = eh + wz, xc = eh - wz;                   // X_k, conj X_(m-k)
        const v2f xk_ = (v2f){xk.x, xk.y}, xc_ = (v2f){xc.x, xc.y};
        store_mag(dst, k, (PLAIN ? magnitude_plain(xk_) : magnitude(xk_)) * scale);
        store_mag(dst, m - k, (PLAIN ? magnitude_plain(xc_) : magnitude(xc_)) * scale);
        if (sp != nullptr) { sp[k] = make_float2(xk.x, xk.y); sp[m - k] = make_float2(xc.x, -xc.y); }
    }
}

__global__ __launch_bounds__(kFwdwThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void stft_mag_forward_wavew_kernel(const StftArgs a)
{
    using namespace sot_wfft;
    constexpr int n = 2048, nb = 1025;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    sot_wfft::v2f* const tw = reinterpret_cast<sot_wfft::v2f*>(smem_f);
    sot_wfft::v2f* const wn = tw + kTw;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    sot_wfft::v2f* const zl = wn + kWnMax + wave * kBuf;
    const float2* const win2 = reinterpret_cast<const float2*>(a.window);
    const unsigned total = (unsigned)(a.batch * a.frames), frames = (unsigned)a.frames;
    const unsigned stride = gridDim.x * kFwdwWaves;
    STFT_STAMP(0);
    // (the first frame's 32 loads issued ahead of the table build change nothing: the build then waits behind them -- tables + loads are
    //  ~6 300 clocks of a wave's 13 000 either way, tools/r5/stft_stamps.py)
    build_tables<kFwdwThreads>(kWn, tw, wn);
    __syncthreads();
    STFT_STAMP(1);
    const float scale = 1.0f / sqrtf((float)n);
    for (unsigned fr = blockIdx.x * kFwdwWaves + wave; fr < total; fr += stride) {
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        const unsigned b = fr / frames, f = fr - b * frames;
        const float* src = (a.audio_b != nullptr && (int64_t)b >= a.split) ? a.audio_b + ((int64_t)b - a.split) * a.row_stride_b
                                                                          : a.audio + (int64_t)b * a.row_stride;
        const int64_t t0 = (int64_t)f * a.hop;
        const float* const s0 = src + t0;
        sot_wfft::v2f r[16];
        const bool pairs = t0 + n <= a.samples && (reinterpret_cast<uintptr_t>(s0) & 7u) == 0;   // wave-uniform
        if (pairs) {
            const float2* const s2 = reinterpret_cast<const float2*>(s0);
#pragma unroll
            for (int q = 0; q < 16; ++q) { const float2 v = s2[64 * q + lane]; r[q] = (sot_wfft::v2f){v.x, v.y}; }
        } else {
            const int left = (int)min((int64_t)n, a.samples - t0);      // zeros past the clip's end (utils.py:252-275)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = 64 * q + lane;
                r[q] = (sot_wfft::v2f){(2 * i < left) ? s0[2 * i] : 0.0f, (2 * i + 1 < left) ? s0[2 * i + 1] : 0.0f};
            }
        }
        float amax = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float2 wv = win2[64 * q + lane];
            r[q] = r[q] * (sot_wfft::v2f){wv.x, wv.y};
            amax = fmaxf(amax, fmaxf(fabsf(r[q].x), fabsf(r[q].y)));
        }
        const bool plain = __builtin_amdgcn_readfirstlane((int)frame_is_plain(wave_max_f32(amax))) != 0;   // one range test per FRAME
        STFT_STAMP(2);
        forward_transform<10>(r, zl, tw, lane);
        STFT_STAMP(3);
        write_natural<10>(r, zl, lane);
        wave_sync();
        float* const dst = a.mag + (int64_t)fr * nb;
        float2* const sp = (a.spec != nullptr && (int64_t)b >= a.spec_first) ? a.spec + ((int64_t)fr - a.spec_first * frames) * nb : nullptr;
        if (plain) wavew_unpack_store<true>(zl, wn, lane, scale, dst, sp);
        else wavew_unpack_store<false>(zl, wn, lane, scale, dst, sp);
        STFT_STAMP(4);
        wave_sync();   // the unpack reads are issued before the next frame's exchange writes
    }
}

__global__ __launch_bounds__(kThreads) void stft_overlap_add_kernel(const StftArgs a)
{
    // clips over blockIdx.y; 32-bit arithmetic inside a clip (samples < 2^31 is checked by the host)
    const int samples = (int)a.samples, span = a.span, groups = (int)a.groups;
    const int gstep = kFramesPerGroup * a.hop;                  // samples between the starts of consecutive groups
    for (int64_t b = blockIdx.y; b < a.batch; b += gridDim.y) {
        const float* part = a.partial + b * (int64_t)groups * span;
        float* out = a.grad_audio + b * a.samples;
        for (int t = blockIdx.x * kThreads + threadIdx.x; t < samples; t += gridDim.x * kThreads) {
            int g_lo = (t - span + gstep) / gstep;              // first group whose span [g * gstep, g * gstep + span) holds t
            if (t - span + 1 <= 0) g_lo = 0;
            const int g_hi = min(t / gstep, groups - 1);
            float sum = 0.0f;
            for (int g = g_lo; g <= g_hi; ++g) sum += part[(int64_t)g * span + (t - g * gstep)];
            out[t] = a.accumulate ? out[t] + sum : sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Spectral distance of the reference's MSSLoss (losses.py:365-425; mean_difference, losses.py:7-36; safe_log,
// utils.py:145-151): over `count` magnitudes,
//   d = mag_weight * mean(D(t - v)) + logmag_weight * mean(D(slog(t) - slog(v))),  D = |.| (L1) or (.)^2 (L2),
//   slog(x) = log(x <= eps ? eps : x).
// Forward: per-workgroup fp64 partial sums (fixed assignment of elements to workgroups), a second one-workgroup kernel
// adds the partials in index order: deterministic.  Backward: elementwise.
// ---------------------------------------------------------------------------------------------
struct DistArgs {
    const float* target; const float* value; int64_t count;
    float mag_weight, logmag_weight, eps; int l2;
    double* partial; int n_partial; float* out; int accumulate;   // forward; accumulate: out += d (the sum over the scales of MSSLoss)
    const float* upstream; float grad_scale;               // backward: d(loss)/d(d) as a device scalar, times grad_scale
    float* grad_target; float* grad_value;                 // either may be null
};

__device__ __forceinline__ float safe_logf(float x, float eps) { return logf(x <= eps ? eps : x); }

__global__ __launch_bounds__(kThreads) void spec_distance_partial_kernel(const DistArgs a)
{
    __shared__ double red[kThreads / 64];
    double acc_m = 0.0, acc_l = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < a.count; i += (int64_t)gridDim.x * kThreads) {
        const float t = a.target[i], v = a.value[i];
        if (a.mag_weight > 0.0f) { const float d = t - v; acc_m += a.l2 ? (double)(d * d) : (double)fabsf(d); }
        if (a.logmag_weight > 0.0f) { const float d = safe_logf(t, a.eps) - safe_logf(v, a.eps); acc_l += a.l2 ? (double)(d * d) : (double)fabsf(d); }
    }
    double acc = (double)a.mag_weight * acc_m + (double)a.logmag_weight * acc_l;
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < kThreads / 64; ++w) tot += red[w];
        a.partial[blockIdx.x] = tot;
    }
}

// one workgroup: thread t adds partials t, t + 256, ... in index order, then a fixed tree over the 256 threads
__global__ __launch_bounds__(kThreads) void spec_distance_finish_kernel(const DistArgs a)
{
    __shared__ double red[kThreads];
    double acc = 0.0;
    for (int i = threadIdx.x; i < a.n_partial; i += kThreads) acc += a.partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = kThreads / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float d = (float)(red[0] / (double)a.count);
        a.out[0] = a.accumulate ? a.out[0] + d : d;
    }
}

__global__ __launch_bounds__(kThreads) void spec_distance_backward_kernel(const DistArgs a)
{
    const float gs = a.upstream[0] * a.grad_scale / (float)a.count;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < a.count; i += (int64_t)gridDim.x * kThreads) {
        const float t = a.target[i], v = a.value[i];
        float gt = 0.0f, gv = 0.0f;   // d(distance * count)/dt, /dv
        if (a.mag_weight > 0.0f) {
            const float d = t - v;
            const float g = a.l2 ? 2.0f * d : (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));   // torch: sgn(0) = 0
            gt += a.mag_weight * g; gv -= a.mag_weight * g;
        }
        if (a.logmag_weight > 0.0f) {
            const float d = safe_logf(t, a.eps) - safe_logf(v, a.eps);
            const float g = a.l2 ? 2.0f * d : (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
            gt += (t <= a.eps) ? 0.0f : a.logmag_weight * g / t;    // where(x <= eps, eps, x): no gradient below eps
            gv -= (v <= a.eps) ? 0.0f : a.logmag_weight * g / v;
        }
        if (a.grad_target) a.grad_target[i] = gs * gt;
        if (a.grad_value) a.grad_value[i] = gs * gv;
    }
}

// The same distance PER ROW (MSSLoss called with dims = the two spectrogram axes: one value per clip, losses.py:406-425 with
// mean_difference's `dims`): row r owns `count` consecutive magnitudes; one workgroup per row, thread t adds elements t, t + 1024, ...
// in fp64, then a fixed tree: deterministic.  Backward: elementwise with the row's own upstream gradient.
constexpr int kRowDistThreads = 1024;
__global__ __launch_bounds__(kRowDistThreads) void spec_distance_rows_kernel(const DistArgs a)
{
    __shared__ double red[kRowDistThreads];
    const float* t_ = a.target + (int64_t)blockIdx.x * a.count;
    const float* v_ = a.value + (int64_t)blockIdx.x * a.count;
    double acc_m = 0.0, acc_l = 0.0;
    for (int64_t i = threadIdx.x; i < a.count; i += kRowDistThreads) {
        const float t = t_[i], v = v_[i];
        if (a.mag_weight > 0.0f) { const float d = t - v; acc_m += a.l2 ? (double)(d * d) : (double)fabsf(d); }
        if (a.logmag_weight > 0.0f) { const float d = safe_logf(t, a.eps) - safe_logf(v, a.eps); acc_l += a.l2 ? (double)(d * d) : (double)fabsf(d); }
    }
    red[threadIdx.x] = (double)a.mag_weight * acc_m + (double)a.logmag_weight * acc_l;
    __syncthreads();
    for (int off = kRowDistThreads / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float d = (float)(red[0] / (double)a.count);
        a.out[blockIdx.x] = a.accumulate ? a.out[blockIdx.x] + d : d;
    }
}

__global__ __launch_bounds__(kThreads) void spec_distance_rows_backward_kernel(const DistArgs a, int64_t rows)
{
    const int64_t total = rows * a.count;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const float gs = a.upstream[i / a.count] * a.grad_scale / (float)a.count;
        const float t = a.target[i], v = a.value[i];
        float gt = 0.0f, gv = 0.0f;
        if (a.mag_weight > 0.0f) {
            const float d = t - v;
            const float g = a.l2 ? 2.0f * d : (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
            gt += a.mag_weight * g; gv -= a.mag_weight * g;
        }
        if (a.logmag_weight > 0.0f) {
            const float d = safe_logf(t, a.eps) - safe_logf(v, a.eps);
            const float g = a.l2 ? 2.0f * d : (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
            gt += (t <= a.eps) ? 0.0f : a.logmag_weight * g / t;
            gv -= (v <= a.eps) ? 0.0f : a.logmag_weight * g / v;
        }
        if (a.grad_target) a.grad_target[i] = gs * gt;
        if (a.grad_value) a.grad_value[i] = gs * gv;
    }
}

constexpr int kDistBlocks = 1024;

static int ilog2_exact(int v)
{
    int l = 0;
    while ((1 << l) < v) ++l;
    return ((1 << l) == v) ? l : -1;
}

static int fill_args(const float* audio, int64_t batch, int64_t samples, int64_t row_stride, const float* window, int n_fft, int hop,
                     StftArgs* a)
{
    if (batch < 0 || samples < 1 || hop < 1 || row_stride < samples) return SOT_ERR_BAD_SHAPE;
    const int logn = ilog2_exact(n_fft);
    if (logn < 6 || n_fft > kMaxFft) return SOT_ERR_UNSUPPORTED_SIZE;  // 64 ... 4096, powers of two
    if (batch > 0 && (audio == nullptr || window == nullptr)) return SOT_ERR_NULL_POINTER;
    if (reinterpret_cast<uintptr_t>(window) % 8 != 0) return SOT_ERR_BAD_SHAPE;   // the kernels read the window two taps at a time
    a->audio = audio; a->batch = batch; a->samples = samples; a->row_stride = row_stride;
    a->window = window; a->n_fft = n_fft; a->logm = logn - 1; a->hop = hop;
    a->frames = (samples + hop - 1) / hop;  // utils.py:265: -(-signal_len // hop_length)
    if (batch * a->frames > 0x7fffffffLL) return SOT_ERR_UNSUPPORTED_SIZE;
    return SOT_OK;
}

// launches KERNEL<log2(n_fft / 2)> with one frame slot per unit of `work`
#define SOT_STFT_LAUNCH(KERNEL, work, extra_lds_per_slot, st, a)                                                              \
    do {                                                                                                                      \
        switch ((a).logm) {                                                                                                   \
            case 5: launch_slots<5>(KERNEL<5>, work, extra_lds_per_slot, st, a); break;                                       \
            case 6: launch_slots<6>(KERNEL<6>, work, extra_lds_per_slot, st, a); break;                                       \
            case 7: launch_slots<7>(KERNEL<7>, work, extra_lds_per_slot, st, a); break;                                       \
            case 8: launch_slots<8>(KERNEL<8>, work, extra_lds_per_slot, st, a); break;                                       \
            case 9: launch_slots<9>(KERNEL<9>, work, extra_lds_per_slot, st, a); break;                                       \
            case 10: launch_slots<10>(KERNEL<10>, work, extra_lds_per_slot, st, a); break;                                    \
            default: launch_slots<11>(KERNEL<11>, work, extra_lds_per_slot, st, a); break;                                    \
        }                                                                                                                     \
    } while (0)

template <int LOGM>
static void launch_slots(void (*kernel)(const StftArgs), int64_t work, size_t extra_lds_per_slot, hipStream_t st, const StftArgs& a)
{
    using G = Geo<LOGM>;
    const size_t lds = G::lds_points * sizeof(float2) + (size_t)G::slots * extra_lds_per_slot;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
            (void)hipGetLastError();
    }
    const unsigned grid = (unsigned)((work + G::slots - 1) / G::slots);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), lds, st, a);
}

#ifndef SOT_STFT_PERSISTENT
#define SOT_STFT_PERSISTENT 1
#endif
static int cu_count()
{
    static std::atomic<int> cus[64];   // zero-initialised; per device, written with the same value by whoever gets there first
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) { (void)hipGetLastError(); v = 256; }
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// the persistent forward kernel on as many workgroups as stay resident (LDS / 2048 threads per CU)
template <int LOGM>
static void launch_forward_persistent(int64_t work, hipStream_t st, const StftArgs& a)
{
    using G = Geo<LOGM>;
    const size_t lds = ((size_t)G::slots * G::zpoints + G::m) * sizeof(float2);
    static std::atomic<int> per_cu_cache[64];   // per device, as in cu_count()
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); dev = 0; }
    int per_cu = per_cu_cache[dev].load(std::memory_order_relaxed);
    if (per_cu == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stft_mag_forward_persistent_kernel<LOGM>, kThreads, lds) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            per_cu = 4;
        }
        per_cu_cache[dev].store(per_cu, std::memory_order_relaxed);
    }
    const int64_t groups = (work + G::slots - 1) / G::slots, cap = (int64_t)cu_count() * per_cu;
    hipLaunchKernelGGL(stft_mag_forward_persistent_kernel<LOGM>, dim3((unsigned)(groups < cap ? groups : cap)), dim3(kThreads), lds, st, a);
}

// Measured (tools/ab_stft.py, 256 clips x 4096 samples): n_fft 2048 forward 25.2 -> 21.9 us, forward of a pair 42.4 -> 38.1 us; n_fft 512
// 10.2 -> 10.8 / 16.7 -> 17.8 us -- the small transforms already share a workgroup's tables among 4-16 frame slots, so only n_fft 2048
// (one frame per workgroup) takes the persistent form.
static void launch_forward(int64_t work, hipStream_t st, const StftArgs& a)
{
    if (SOT_STFT_PERSISTENT && a.logm == 10) { launch_forward_persistent<10>(work, st, a); return; }
    SOT_STFT_LAUNCH(stft_mag_forward_kernel, work, 0, st, a);
}

// OFF by default.  Measured (MI355X, 256 clips x 16 frames, tools/ab_stft.py): forward 24.2 us against 24.7 us for the slot
// kernel, forward of a pair 52.5 against 42.3 us.  The kernel is correct (tests/test_stft_producer.py passes with it) and does
// a frame in ~1000 instructions per lane without a workgroup barrier, but at 16-32 frames per CU there is no steady state to
// amortise anything over: a frame's dependent chain (loads -> five stages -> unpack) on ONE wavefront at two waves per SIMD
// (194 VGPRs) takes longer than the same frame spread over four wavefronts.  It is the form to switch on for batches of
// >= 64 frames per CU, after a register diet (the two 16-point arrays).
#ifndef SOT_STFT_WAVE_KERNEL
#define SOT_STFT_WAVE_KERNEL 0
#endif
// n_fft 2048: the one-wavefront-per-frame kernel, persistent grid (3 workgroups of 4 frames per CU fit the LDS); returns
// false for other sizes (the caller launches the slot kernel)
// Measured (round 4, tools/r4/stft_sizes.py, forward of a pair, n_fft 2048 / hop 256, us per call, slot kernel | this kernel;
// profiles/r4j_stft_wave2.txt): 512 frames 10.6 | 15.0, 1024: 10.6 | 15.3, 2048: 14.4 | 16.1, 4096: 21.6 | 17.3, 8192 (config 5): 37.3 | 28.5,
// 16384: 64.0 | 48.0 -- a frame is a 28 000-clock dependent chain on one wave (~15 us), so the kernel needs at least one frame per
// wave slot of the chip (256 CUs x 16) to pay: from 3072 frames.
#ifndef SOT_STFT_WAVE2_KERNEL
#define SOT_STFT_WAVE2_KERNEL 1
#endif
#ifndef SOT_STFT_WAVE2_MIN_FRAMES
#define SOT_STFT_WAVE2_MIN_FRAMES 3072
#endif
static bool launch_forward_wave2(const StftArgs& a, int64_t frames_total, hipStream_t st)
{
    if (!SOT_STFT_WAVE2_KERNEL || a.logm != 10 || frames_total < SOT_STFT_WAVE2_MIN_FRAMES) return false;
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {   // idempotent per device; a benign race sets it twice
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(stft_mag_forward_wave2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kWave2LdsBytes) != hipSuccess)
            (void)hipGetLastError();
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    const int64_t want = (frames_total + kWave2Waves - 1) / kWave2Waves, cap = cu_count();   // one 1024-thread workgroup per CU
    hipLaunchKernelGGL(stft_mag_forward_wave2_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(kWave2Threads), kWave2LdsBytes, st, a);
    return true;
}

#ifndef SOT_STFT_BWD_WAVE2_KERNEL
#define SOT_STFT_BWD_WAVE2_KERNEL 1
#endif
#ifndef SOT_STFT_BWD_WAVE2_MIN_GROUPS
#define SOT_STFT_BWD_WAVE2_MIN_GROUPS 1024
#endif
#ifndef SOT_STFT_BWD_CLIP_KERNEL
#define SOT_STFT_BWD_CLIP_KERNEL 1
#endif
#ifndef SOT_STFT_BWD_CLIP_MIN_CLIPS
#define SOT_STFT_BWD_CLIP_MIN_CLIPS 64
#endif
#ifndef SOT_STFT_BWD_CLIPW_KERNEL
#define SOT_STFT_BWD_CLIPW_KERNEL 1
#endif
#ifndef SOT_STFT_FWD_WAVEW_MIN_FRAMES
#define SOT_STFT_FWD_WAVEW_MIN_FRAMES 512     /* below: the slot kernels (a frame is a dependent chain on one wave: small batches want the frame spread over four) */
#endif
// the backward from the stored spectrum with the overlap-add inside: clips of at most 16 frames of 2048, one workgroup per clip.  Returns true
// when it has launched the WHOLE backward (the caller then skips stft_overlap_add_kernel).
static bool launch_backward_spec_clip(const StftArgs& a, hipStream_t st)
{
    if (!SOT_STFT_BWD_CLIP_KERNEL || a.logm != 10 || a.spec_in == nullptr || a.frames > kWave2Waves || a.frames < 1 || (a.hop & 1) != 0 ||
        a.batch < SOT_STFT_BWD_CLIP_MIN_CLIPS)
        return false;
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(stft_mag_backward_spec_clip_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kWave2LdsBytes) != hipSuccess)
            (void)hipGetLastError();
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    const int64_t cap = cu_count();
#if SOT_STFT_BWD_CLIPW_KERNEL     /* round 5: the same structure on the one-wavefront FFT of csrc/sot_wave_fft.hpp */
    {
        static bool attrw_done[64] = {};
        if (dev < 0 || dev >= 64 || !attrw_done[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(stft_mag_backward_spec_clipw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kClipwLdsBytes) != hipSuccess)
                (void)hipGetLastError();
            if (dev >= 0 && dev < 64) attrw_done[dev] = true;
        }
        hipLaunchKernelGGL(stft_mag_backward_spec_clipw_kernel, dim3((unsigned)(a.batch < cap ? a.batch : cap)), dim3(kWave2Threads), kClipwLdsBytes, st, a);
        return true;
    }
#endif
    hipLaunchKernelGGL(stft_mag_backward_spec_clip_kernel, dim3((unsigned)(a.batch < cap ? a.batch : cap)), dim3(kWave2Threads), kWave2LdsBytes, st, a);
    return true;
}

// the backward from the stored spectrum, one wavefront per group of two frames: n_fft 2048, hop 256 or 512, scratch rows on 8-byte boundaries
static bool launch_backward_spec_wave2(const StftArgs& a, int64_t groups_total, hipStream_t st)
{
    if (!SOT_STFT_BWD_WAVE2_KERNEL || a.logm != 10 || groups_total < SOT_STFT_BWD_WAVE2_MIN_GROUPS || (a.hop != 256 && a.hop != 512) ||
        (reinterpret_cast<uintptr_t>(a.partial) & 7u) != 0 || (a.span & 1) != 0)
        return false;
    void (*kern)(const StftArgs) = a.hop == 256 ? stft_mag_backward_spec_wave2_kernel<2> : stft_mag_backward_spec_wave2_kernel<4>;
    static bool attr_done[64][2] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    const int which = a.hop == 256 ? 0 : 1;
    if (dev < 0 || dev >= 64 || !attr_done[dev][which]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdWave2LdsBytes) != hipSuccess)
            (void)hipGetLastError();
        if (dev >= 0 && dev < 64) attr_done[dev][which] = true;
    }
    const int64_t want = (groups_total + kBwdWave2Waves - 1) / kBwdWave2Waves, cap = cu_count();   // one 512-thread workgroup per CU
    hipLaunchKernelGGL(kern, dim3((unsigned)(want < cap ? want : cap)), dim3(kBwdWave2Threads), kBwdWave2LdsBytes, st, a);
    return true;
}

// n_fft 2048, SOT_STFT_FWD_WAVEW_MIN_FRAMES <= frames < SOT_STFT_WAVE2_MIN_FRAMES: one wavefront per frame on the round-5 FFT (the round-4
// wave kernel keeps the larger batches: its results on BASELINE config 5 are what the cutoff mode's knife-edge statistics were taken with)
static bool launch_forward_wavew(const StftArgs& a, int64_t frames_total, hipStream_t st)
{
    if (a.logm != 10 || frames_total < SOT_STFT_FWD_WAVEW_MIN_FRAMES || (SOT_STFT_WAVE2_KERNEL && frames_total >= SOT_STFT_WAVE2_MIN_FRAMES)) return false;
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(stft_mag_forward_wavew_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kFwdwLdsBytes) != hipSuccess)
            (void)hipGetLastError();
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    const int64_t want = (frames_total + kFwdwWaves - 1) / kFwdwWaves, cap = 3 * (int64_t)cu_count();   // three 256-thread workgroups per CU (LDS)
    hipLaunchKernelGGL(stft_mag_forward_wavew_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(kFwdwThreads), kFwdwLdsBytes, st, a);
    return true;
}

static bool launch_forward_wave(const StftArgs& a, int64_t frames_total, hipStream_t st)
{
    if (launch_forward_wavew(a, frames_total, st)) return true;
    if (launch_forward_wave2(a, frames_total, st)) return true;
    if (!SOT_STFT_WAVE_KERNEL || a.logm != 10) return false;
    const size_t lds = (1024 + 520 + 4 * (size_t)kWaveBuf) * sizeof(float2);
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {   // idempotent per device; a benign race sets it twice
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(stft_mag_forward_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            (void)hipGetLastError();
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    const int64_t want = (frames_total + 3) / 4;
    const int64_t cap = 256 * 3;
    hipLaunchKernelGGL(stft_mag_forward_wave_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(kThreads), lds, st, a);
    return true;
}

}  // namespace sot_stft

extern "C" {

#ifdef STFT_STAMPS
int sot_stft_debug_read_stamps(unsigned long long* host_out, int count)   // diagnostic build only (synchronises)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(sot_stft::g_stft_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : -1;
}
#endif

int64_t sot_stft_frames(int64_t samples, int hop) { return (samples < 1 || hop < 1) ? 0 : (samples + hop - 1) / hop; }

int sot_stft_mag_forward(const float* audio, int64_t batch, int64_t samples, int64_t audio_row_stride, const float* window,
                         int n_fft, int hop, float* mag, void* stream)
{
    return sot_stft_mag_forward_spec(audio, batch, samples, audio_row_stride, window, n_fft, hop, mag, nullptr, stream);
}

int sot_stft_mag_forward_spec(const float* audio, int64_t batch, int64_t samples, int64_t audio_row_stride, const float* window,
                              int n_fft, int hop, float* mag, float* spec, void* stream)
{
    using namespace sot_stft;
    StftArgs a{};
    const int rc = fill_args(audio, batch, samples, audio_row_stride, window, n_fft, hop, &a);
    if (rc != SOT_OK) return rc;
    if (batch == 0) return SOT_OK;
    if (mag == nullptr) return SOT_ERR_NULL_POINTER;
    if (spec != nullptr && reinterpret_cast<uintptr_t>(spec) % 8 != 0) return SOT_ERR_BAD_SHAPE;
    a.mag = mag;
    a.spec = reinterpret_cast<float2*>(spec); a.spec_first = 0;
    (void)hipGetLastError();
    if (!launch_forward_wave(a, batch * a.frames, reinterpret_cast<hipStream_t>(stream)))
        launch_forward(batch * a.frames, reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

int sot_stft_mag_forward_pair(const float* audio_a, int64_t row_stride_a, const float* audio_b, int64_t row_stride_b,
                              int64_t batch_each, int64_t samples, const float* window, int n_fft, int hop, float* mag, void* stream)
{
    return sot_stft_mag_forward_pair_spec(audio_a, row_stride_a, audio_b, row_stride_b, batch_each, samples, window, n_fft, hop, mag, nullptr, stream);
}

int sot_stft_mag_forward_pair_spec(const float* audio_a, int64_t row_stride_a, const float* audio_b, int64_t row_stride_b,
                                   int64_t batch_each, int64_t samples, const float* window, int n_fft, int hop, float* mag, float* spec_b,
                                   void* stream)
{
    using namespace sot_stft;
    if (row_stride_b < samples) return SOT_ERR_BAD_SHAPE;
    StftArgs a{};
    const int rc = fill_args(audio_a, 2 * batch_each, samples, row_stride_a, window, n_fft, hop, &a);
    if (rc != SOT_OK) return rc;
    if (batch_each == 0) return SOT_OK;
    if (mag == nullptr || audio_b == nullptr) return SOT_ERR_NULL_POINTER;
    if (spec_b != nullptr && reinterpret_cast<uintptr_t>(spec_b) % 8 != 0) return SOT_ERR_BAD_SHAPE;
    a.audio_b = audio_b; a.split = batch_each; a.row_stride_b = row_stride_b;
    a.mag = mag;
    a.spec = reinterpret_cast<float2*>(spec_b); a.spec_first = batch_each;   // the spectrum of the SECOND signal only (the estimate: the one that is differentiated)
    (void)hipGetLastError();
    if (!launch_forward_wave(a, 2 * batch_each * a.frames, reinterpret_cast<hipStream_t>(stream)))
        launch_forward(2 * batch_each * a.frames, reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

size_t sot_stft_backward_workspace_bytes(int64_t batch, int64_t samples, int n_fft, int hop)
{
    if (batch < 1 || samples < 1 || hop < 1 || n_fft < 1) return 0;
    const int64_t frames = (samples + hop - 1) / hop;
    const int64_t groups = (frames + sot_stft::kFramesPerGroup - 1) / sot_stft::kFramesPerGroup;
    const int64_t span = n_fft + (int64_t)hop * (sot_stft::kFramesPerGroup - 1);
    return sizeof(float) * (size_t)(batch * groups * span);
}

int sot_stft_mag_backward(const float* audio, int64_t batch, int64_t samples, int64_t audio_row_stride, const float* window,
                          int n_fft, int hop, const float* grad_mag, const float* grad_scale, float* grad_audio, int accumulate,
                          void* workspace, size_t workspace_bytes, void* stream)
{
    return sot_stft_mag_backward_spec(audio, nullptr, batch, samples, audio_row_stride, window, n_fft, hop, grad_mag, grad_scale, grad_audio,
                                      accumulate, workspace, workspace_bytes, stream);
}

int sot_stft_mag_backward_spec(const float* audio, const float* spec, int64_t batch, int64_t samples, int64_t audio_row_stride,
                               const float* window, int n_fft, int hop, const float* grad_mag, const float* grad_scale, float* grad_audio,
                               int accumulate, void* workspace, size_t workspace_bytes, void* stream)
{
    using namespace sot_stft;
    StftArgs a{};
    if (spec != nullptr && audio == nullptr) { audio = spec; audio_row_stride = samples; }   // never read: the spectra replace the audio
    if (spec != nullptr && reinterpret_cast<uintptr_t>(spec) % 8 != 0) return SOT_ERR_BAD_SHAPE;
    const int rc = fill_args(audio, batch, samples, audio_row_stride, window, n_fft, hop, &a);
    if (rc != SOT_OK) return rc;
    a.spec_in = reinterpret_cast<const float2*>(spec);
    if (batch == 0) return SOT_OK;
    if (grad_mag == nullptr || grad_audio == nullptr || workspace == nullptr) return SOT_ERR_NULL_POINTER;
    if (workspace_bytes < sot_stft_backward_workspace_bytes(batch, samples, n_fft, hop)) return SOT_ERR_WORKSPACE;
    const int64_t span = n_fft + (int64_t)hop * (kFramesPerGroup - 1);
    if (span > 8192) return SOT_ERR_UNSUPPORTED_SIZE;   // the groups' gradients live in LDS
    if (samples > 0x7fffffffLL) return SOT_ERR_UNSUPPORTED_SIZE;   // overlap-add kernel: 32-bit sample indices inside a clip
    a.grad_mag = grad_mag; a.grad_scale = grad_scale; a.grad_audio = grad_audio; a.accumulate = accumulate;
    a.partial = reinterpret_cast<float*>(workspace);
    a.groups = (a.frames + kFramesPerGroup - 1) / kFramesPerGroup;
    a.span = (int)span;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (spec != nullptr && launch_backward_spec_clip(a, st)) return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;   // overlap-add inside
    if (spec != nullptr && launch_backward_spec_wave2(a, batch * a.groups, st)) { /* one wavefront per frame group */ }
    else if (spec != nullptr) SOT_STFT_LAUNCH(stft_mag_backward_spec_kernel, batch * a.groups, sizeof(float) * (size_t)span, st, a);
    else SOT_STFT_LAUNCH(stft_mag_backward_partial_kernel, batch * a.groups, sizeof(float) * (size_t)span, st, a);
    if (hipGetLastError() != hipSuccess) return SOT_ERR_LAUNCH;
    const int64_t per_clip = (samples + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(stft_overlap_add_kernel, dim3((unsigned)(per_clip < 64 ? per_clip : 64), (unsigned)(batch < 65535 ? batch : 65535)), dim3(kThreads), 0, st, a);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

size_t sot_spec_distance_workspace_bytes(void) { return sizeof(double) * (size_t)sot_stft::kDistBlocks; }

int sot_spec_distance_forward(const float* target, const float* value, int64_t count, float mag_weight, float logmag_weight,
                              float eps, int l2, float* out, int accumulate, void* workspace, size_t workspace_bytes, void* stream)
{
    using namespace sot_stft;
    if (count < 1) return SOT_ERR_BAD_SHAPE;
    if (target == nullptr || value == nullptr || out == nullptr || workspace == nullptr) return SOT_ERR_NULL_POINTER;
    if (workspace_bytes < sot_spec_distance_workspace_bytes()) return SOT_ERR_WORKSPACE;
    DistArgs a{};
    a.target = target; a.value = value; a.count = count; a.mag_weight = mag_weight; a.logmag_weight = logmag_weight; a.eps = eps;
    a.l2 = l2; a.partial = reinterpret_cast<double*>(workspace); a.out = out; a.accumulate = accumulate;
    const int64_t need = (count + kThreads - 1) / kThreads;
    a.n_partial = (int)(need < kDistBlocks ? need : kDistBlocks);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    hipLaunchKernelGGL(spec_distance_partial_kernel, dim3(a.n_partial), dim3(kThreads), 0, st, a);
    hipLaunchKernelGGL(spec_distance_finish_kernel, dim3(1), dim3(kThreads), 0, st, a);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

int sot_spec_distance_backward(const float* target, const float* value, int64_t count, float mag_weight, float logmag_weight,
                               float eps, int l2, const float* upstream, float grad_scale, float* grad_target, float* grad_value,
                               void* stream)
{
    using namespace sot_stft;
    if (count < 1) return SOT_ERR_BAD_SHAPE;
    if (target == nullptr || value == nullptr || upstream == nullptr) return SOT_ERR_NULL_POINTER;
    if (grad_target == nullptr && grad_value == nullptr) return SOT_OK;
    DistArgs a{};
    a.target = target; a.value = value; a.count = count; a.mag_weight = mag_weight; a.logmag_weight = logmag_weight; a.eps = eps;
    a.l2 = l2; a.upstream = upstream; a.grad_scale = grad_scale; a.grad_target = grad_target; a.grad_value = grad_value;
    const int64_t need = (count + kThreads - 1) / kThreads;
    const int grid = (int)(need < 256 * 32 ? need : 256 * 32);
    (void)hipGetLastError();
    hipLaunchKernelGGL(spec_distance_backward_kernel, dim3(grid), dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

int sot_spec_distance_rows_forward(const float* target, const float* value, int64_t rows, int64_t count_per_row, float mag_weight,
                                   float logmag_weight, float eps, int l2, float* out, int accumulate, void* stream)
{
    using namespace sot_stft;
    if (rows < 1 || count_per_row < 1 || rows > 0x7fffffff) return SOT_ERR_BAD_SHAPE;
    if (target == nullptr || value == nullptr || out == nullptr) return SOT_ERR_NULL_POINTER;
    DistArgs a{};
    a.target = target; a.value = value; a.count = count_per_row; a.mag_weight = mag_weight; a.logmag_weight = logmag_weight; a.eps = eps;
    a.l2 = l2; a.out = out; a.accumulate = accumulate;
    (void)hipGetLastError();
    hipLaunchKernelGGL(spec_distance_rows_kernel, dim3((unsigned)rows), dim3(kRowDistThreads), 0, reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

int sot_spec_distance_rows_backward(const float* target, const float* value, int64_t rows, int64_t count_per_row, float mag_weight,
                                    float logmag_weight, float eps, int l2, const float* upstream, float grad_scale, float* grad_target,
                                    float* grad_value, void* stream)
{
    using namespace sot_stft;
    if (rows < 1 || count_per_row < 1) return SOT_ERR_BAD_SHAPE;
    if (target == nullptr || value == nullptr || upstream == nullptr) return SOT_ERR_NULL_POINTER;
    if (grad_target == nullptr && grad_value == nullptr) return SOT_OK;
    DistArgs a{};
    a.target = target; a.value = value; a.count = count_per_row; a.mag_weight = mag_weight; a.logmag_weight = logmag_weight; a.eps = eps;
    a.l2 = l2; a.upstream = upstream; a.grad_scale = grad_scale; a.grad_target = grad_target; a.grad_value = grad_value;
    const int64_t need = (rows * count_per_row + kThreads - 1) / kThreads;
    const int grid = (int)(need < 256 * 32 ? need : 256 * 32);
    (void)hipGetLastError();
    hipLaunchKernelGGL(spec_distance_rows_backward_kernel, dim3(grid), dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream), a, rows);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

}  // extern "C"
