// sot_stft.hip -- MI355X (gfx950) kernels for the producer in front of the SOT loss: the magnitude STFT of the
// reference's `features.TorchSTFT` (features.py:85-113 -> compute_mag / stft, features.py:191-237; end padding
// utils.pad_for_stft, utils.py:252-275): torch.stft(center=False, normalized=True, onesided) of the end-padded signal,
// |.|, frames-major output [batch, frames, n_fft/2 + 1].  SURVEY §8f row 1.
//
// Forward: one workgroup per frame.  The windowed REAL frame is packed into n_fft/2 complex points and goes through an
// in-LDS radix-2 complex FFT of half the frame length (bit-reversed load, log2(n_fft/2) butterfly stages, twiddles from
// LDS tables built with sincospi); the n_fft/2 + 1 bins are unpacked pairwise and reduced to hypot(re, im) / sqrt(n_fft).
// Backward (closed form of abs o stft's autograd): per group of four consecutive frames, frame by frame, recompute the
// frame's spectrum X, form
// Z_k = g_k X_k / |X_k| (0 where |X_k| = 0, torch's sgn(0)), inverse-transform it as a Hermitian spectrum (again a
// half-length complex transform), multiply by window / sqrt(n_fft) and overlap-add it into the group's gradient, which
// is kept in LDS and written once to a scratch buffer; a second kernel adds, per sample, the groups that cover it in a
// fixed order: no atomics, deterministic.
// HBM traffic: forward reads n_fft samples per frame (L2-resident overlap) and writes n_fft/2+1 magnitudes; backward
// reads the audio and the magnitude gradients once, writes and re-reads the groups' partial gradients (1.4 x the audio
// for 4 frames per group at 8 frames per sample) and writes the audio gradient once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/sot_hip.h"

namespace sot_stft {

constexpr int kThreads = 256;      // forward: one 256-thread workgroup per frame
constexpr int kFramesPerGroup = 4;  // backward: one workgroup per group of consecutive frames of a clip
constexpr int kMaxFft = 2048;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// twiddle table T[k] = exp(-2 pi i k / n), k < n/2 (accurate: sincospi)
template <int T>
__device__ __forceinline__ void build_twiddles(float2* tw, int n)
{
    for (int k = threadIdx.x; k < n / 2; k += T) {
        float s, c;
        sincospif(2.0f * (float)k / (float)n, &s, &c);
        tw[k] = make_float2(c, -s);
    }
}

// In-place decimation-in-time FFT of n = 2^logn points held in LDS in BIT-REVERSED order on entry, natural order on exit,
// executed by the `nthr` threads lid = 0 .. nthr-1 of one frame slot (a workgroup holds kThreads / nthr slots, each with
// its own z; the barriers are workgroup-wide, every slot runs the same passes).
// Two radix-2 stages are executed per pass (a radix-4 butterfly on the elements i0, i0+h, i0+2h, i0+3h: the same
// operations, in the same order per element, as two separate stages -- half the LDS round trips and barriers); an odd
// logn starts with one plain radix-2 stage.  inverse: conjugate twiddles (no 1/n).  Ends with a barrier.
__device__ __forceinline__ void fft_inplace(float2* z, const float2* tw, int n, int logn, bool inverse, int lid, int nthr)
{
    int s = 1;
    if (logn & 1) {  // stage 1: half = 1, twiddle 1
        __syncthreads();
        for (int j = lid; j < n / 2; j += nthr) {
            const float2 a = z[2 * j], b = z[2 * j + 1];
            z[2 * j] = make_float2(a.x + b.x, a.y + b.y);
            z[2 * j + 1] = make_float2(a.x - b.x, a.y - b.y);
        }
        s = 2;
    }
    for (; s <= logn; s += 2) {
        const int h = 1 << (s - 1);       // half size of stage s; stage s+1 has half size 2h
        const int t1 = n >> s;            // stage s:     exp(-2 pi i pos / (2h)) = T[pos * n / (2h)]
        const int t2 = n >> (s + 1);      // stage s + 1: exp(-2 pi i pos / (4h)) = T[pos * n / (4h)]
        __syncthreads();
        for (int j = lid; j < n / 4; j += nthr) {
            const int pos = j & (h - 1);
            const int i0 = ((j >> (s - 1)) << (s + 1)) + pos;
            float2 w1 = tw[pos * t1], w2 = tw[pos * t2];
            if (inverse) { w1.y = -w1.y; w2.y = -w2.y; }
            const float2 a = z[i0], b = cmul(z[i0 + h], w1), c = z[i0 + 2 * h], d = cmul(z[i0 + 3 * h], w1);
            const float2 a1 = make_float2(a.x + b.x, a.y + b.y), b1 = make_float2(a.x - b.x, a.y - b.y);
            const float2 c1 = make_float2(c.x + d.x, c.y + d.y), d1 = make_float2(c.x - d.x, c.y - d.y);
            const float2 c2 = cmul(c1, w2);
            // exp(-2 pi i (pos + h) / (4h)) = w2 * (-i)  (forward),  w2 * (+i)  (inverse)
            const float2 w3 = inverse ? make_float2(-w2.y, w2.x) : make_float2(w2.y, -w2.x);
            const float2 d2 = cmul(d1, w3);
            z[i0] = make_float2(a1.x + c2.x, a1.y + c2.y);
            z[i0 + 2 * h] = make_float2(a1.x - c2.x, a1.y - c2.y);
            z[i0 + h] = make_float2(b1.x + d2.x, b1.y + d2.y);
            z[i0 + 3 * h] = make_float2(b1.x - d2.x, b1.y - d2.y);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int bitrev(int v, int logn) { return (int)(__brev((unsigned)v) >> (32 - logn)); }

struct StftArgs {
    const float* audio; int64_t batch, samples, row_stride;
    const float* window; int n_fft, logm, hop; int64_t frames;   // logm = log2(n_fft / 2)
    int tpf;                    // threads per frame slot: max(16, n_fft / 8); a workgroup holds kThreads / tpf slots
    float* mag;                 // forward output [batch, frames, n_fft/2+1]
    const float* grad_mag;      // backward input, same shape
    float* grad_audio;          // backward output [batch, samples] (contiguous)
    float* partial;             // backward scratch [batch, groups, span]: each frame group's overlap-added gradient
    int64_t groups; int span;   // span = n_fft + hop * (kFramesPerGroup - 1)
};

// The frames are REAL, so each one is transformed by a complex FFT of HALF its length m = n_fft/2 on the packed signal
// z[i] = v[2i] + i v[2i+1]:   with Ze = (Z_k + conj(Z_{m-k})) / 2, Zo = -i/2 (Z_k - conj(Z_{m-k})), W = exp(-2 pi i k / n):
//   X_k = Ze + W Zo,   X_{m-k} = conj(Ze - W Zo)      (k = 0 .. m/2; Z_m := Z_0)
// LDS: z [slots][m] | FFT twiddles exp(-2 pi i k / m) [m/2] | W_n^k [m/2 + 1].  Small transforms share a workgroup:
// n_fft = 64 / 128 -> 16 frames per workgroup, 256 -> 8, 512 -> 4, 1024 -> 2, 2048 -> 1.
template <int T>
__device__ __forceinline__ void build_tables(float2* tw, float2* wn, int m)
{
    build_twiddles<T>(tw, m);
    for (int k = threadIdx.x; k <= m / 2; k += T) {
        float s, c;
        sincospif((float)k / (float)m, &s, &c);   // 2 pi k / n = pi k / m
        wn[k] = make_float2(c, -s);
    }
}

// windowed, end-padded, packed frame -> LDS in bit-reversed order (zeros for an idle slot)
__device__ __forceinline__ void load_frame(const StftArgs& a, const float* src, int64_t t0, float2* z, int m, bool active, int lid, int nthr)
{
    for (int i = lid; i < m; i += nthr) {
        const int64_t t = t0 + 2 * i;
        const float v0 = (active && t < a.samples) ? src[t] * a.window[2 * i] : 0.0f;          // end padding: zeros (utils.py:252-275)
        const float v1 = (active && t + 1 < a.samples) ? src[t + 1] * a.window[2 * i + 1] : 0.0f;
        z[bitrev(i, a.logm)] = make_float2(v0, v1);
    }
}

__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// spectrum bins k and m-k of the real frame from the packed transform (see above)
__device__ __forceinline__ void unpack_pair(const float2* z, const float2* wn, int k, int m, float2& xk, float2& xm)
{
    const float2 zk = z[k], zm = z[(m - k) & (m - 1)];
    const float2 ze = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
    const float2 d = make_float2(zk.x - zm.x, zk.y + zm.y);          // Z_k - conj(Z_{m-k})
    const float2 zo = make_float2(0.5f * d.y, -0.5f * d.x);          // -i/2 * d
    const float2 wz = cmul(wn[k], zo);
    xk = make_float2(ze.x + wz.x, ze.y + wz.y);
    xm = make_float2(ze.x - wz.x, -(ze.y - wz.y));
}

__global__ __launch_bounds__(kThreads) void stft_mag_forward_kernel(const StftArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int n = a.n_fft, m = n / 2, nb = m + 1;
    const int nthr = a.tpf, slots = kThreads / nthr;
    const int slot = threadIdx.x / nthr, lid = threadIdx.x - slot * nthr;
    float2* const zall = reinterpret_cast<float2*>(smem_f);
    float2* const z = zall + slot * m;
    float2* const tw = zall + slots * m;
    float2* const wn = tw + m / 2;
    const float scale = 1.0f / sqrtf((float)n);  // normalized=True: frame_length^-0.5
    const int64_t total = a.batch * a.frames;
    build_tables<kThreads>(tw, wn, m);
    for (int64_t base = (int64_t)blockIdx.x * slots; base < total; base += (int64_t)gridDim.x * slots) {
        const int64_t fr = base + slot;
        const bool active = fr < total;
        const int64_t b = active ? fr / a.frames : 0, f = active ? fr - b * a.frames : 0;
        __syncthreads();  // previous frames' reads of z are done
        load_frame(a, a.audio + b * a.row_stride, f * a.hop, z, m, active, lid, nthr);
        fft_inplace(z, tw, m, a.logm, false, lid, nthr);
        if (active) {
            float* dst = a.mag + fr * nb;
            for (int k = lid; k <= m / 2; k += nthr) {
                float2 xk, xm;
                unpack_pair(z, wn, k, m, xk, xm);
                dst[k] = hypotf(xk.x, xk.y) * scale;
                dst[m - k] = hypotf(xm.x, xm.y) * scale;
            }
        }
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
__global__ __launch_bounds__(kThreads) void stft_mag_backward_partial_kernel(const StftArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int n = a.n_fft, m = n / 2, nb = m + 1;
    const int nthr = a.tpf, slots = kThreads / nthr;
    const int slot = threadIdx.x / nthr, lid = threadIdx.x - slot * nthr;
    constexpr int kPairIters = 3;   // m/2 + 1 pairs (k, m-k) over nthr >= m/4 threads
    float2* const zall = reinterpret_cast<float2*>(smem_f);
    float2* const z = zall + slot * m;
    float2* const tw = zall + slots * m;
    float2* const wn = tw + m / 2;
    float* const acc = reinterpret_cast<float*>(wn + m / 2 + 2) + slot * a.span;  // this group's overlap-added gradient [span]
    const float scale = 1.0f / sqrtf((float)n);
    const int64_t total = a.batch * a.groups;
    build_tables<kThreads>(tw, wn, m);
    for (int64_t base0 = (int64_t)blockIdx.x * slots; base0 < total; base0 += (int64_t)gridDim.x * slots) {
        const int64_t w = base0 + slot;
        const bool active = w < total;
        const int64_t b = active ? w / a.groups : 0, grp = active ? w - b * a.groups : 0;
        const float* src = a.audio + b * a.row_stride;
        const int64_t f_begin = grp * kFramesPerGroup;
        __syncthreads();
        for (int t = lid; t < a.span; t += nthr) acc[t] = 0.0f;
        for (int fi = 0; fi < kFramesPerGroup; ++fi) {
            const int64_t f = f_begin + fi;
            const bool has = active && f < a.frames;   // idle slots / missing frames run the same passes on zeros
            const int64_t t0 = f * a.hop;
            __syncthreads();
            load_frame(a, src, t0, z, m, has, lid, nthr);
            fft_inplace(z, tw, m, a.logm, false, lid, nthr);
            // pairs (k, m-k): spectrum -> Zin -> H -> G, kept in registers until every thread has read z
            const float* g = a.grad_mag + (b * a.frames + (has ? f : 0)) * nb;
            float2 gk[kPairIters], gm[kPairIters];
#pragma unroll
            for (int r = 0; r < kPairIters; ++r) {
                const int k = lid + r * nthr;
                gk[r] = make_float2(0.0f, 0.0f); gm[r] = make_float2(0.0f, 0.0f);
                if (has && k <= m / 2) {
                    float2 xk, xm;
                    unpack_pair(z, wn, k, m, xk, xm);
                    const float mk = hypotf(xk.x, xk.y), mm = hypotf(xm.x, xm.y);
                    const float ck = mk > 0.0f ? g[k] / mk : 0.0f;          // torch: sgn(0) = 0
                    const float cm = mm > 0.0f ? g[m - k] / mm : 0.0f;
                    float2 hk = make_float2(0.5f * ck * xk.x, 0.5f * ck * xk.y);
                    float2 hm = make_float2(0.5f * cm * xm.x, 0.5f * cm * xm.y);
                    if (k == 0) { hk = make_float2(ck * xk.x, 0.0f); hm = make_float2(cm * xm.x, 0.0f); }   // H_0, H_m are real
                    const float2 sk = make_float2(hk.x + hm.x, hk.y - hm.y);      // H_k + conj(H_{m-k})
                    const float2 dk = make_float2(hk.x - hm.x, hk.y + hm.y);      // H_k - conj(H_{m-k})
                    const float2 wk = wn[k];
                    const float2 cw = cmul(cconj(wk), dk);                         // conj(W) d
                    gk[r] = make_float2(sk.x - cw.y, sk.y + cw.x);                 // s + i conj(W) d
                    // G_{m-k} = (H_{m-k} + conj(H_k)) + i (-W) (H_{m-k} - conj(H_k)) = conj(s) + i W conj(d)
                    const float2 wd = cmul(wk, cconj(dk));
                    gm[r] = make_float2(sk.x - wd.y, -sk.y + wd.x);
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kPairIters; ++r) {
                const int k = lid + r * nthr;
                if (k <= m / 2) {
                    z[bitrev(k, a.logm)] = gk[r];
                    if (k > 0 && k < m - k) z[bitrev(m - k, a.logm)] = gm[r];
                }
            }
            fft_inplace(z, tw, m, a.logm, true, lid, nthr);
            if (has) {
                const int off = fi * a.hop;
                for (int i = lid; i < m; i += nthr) {
                    acc[off + 2 * i] += a.window[2 * i] * z[i].x * scale;
                    acc[off + 2 * i + 1] += a.window[2 * i + 1] * z[i].y * scale;
                }
            }
        }
        __syncthreads();
        if (active) {
            float* dst = a.partial + w * a.span;
            for (int t = lid; t < a.span; t += nthr) dst[t] = acc[t];
        }
    }
}

__global__ __launch_bounds__(kThreads) void stft_overlap_add_kernel(const StftArgs a)
{
    const int64_t total = a.batch * a.samples;
    const int64_t gstep = (int64_t)kFramesPerGroup * a.hop;   // samples between the starts of consecutive groups
    for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kThreads) {
        const int64_t b = idx / a.samples, t = idx - b * a.samples;
        int64_t g_lo = (t - a.span + gstep) / gstep;           // first group whose span [g * gstep, g * gstep + span) holds t
        if (t - a.span + 1 <= 0) g_lo = 0;
        const int64_t g_hi = min(t / gstep, a.groups - 1);
        float sum = 0.0f;
        for (int64_t g = g_lo; g <= g_hi; ++g) sum += a.partial[(b * a.groups + g) * a.span + (t - g * gstep)];
        a.grad_audio[idx] = sum;
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
    double* partial; int n_partial; float* out;           // forward
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
    if (threadIdx.x == 0) a.out[0] = (float)(red[0] / (double)a.count);
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
    if (logn < 6 || n_fft > kMaxFft) return SOT_ERR_UNSUPPORTED_SIZE;  // 64 ... 2048, powers of two
    if (batch > 0 && (audio == nullptr || window == nullptr)) return SOT_ERR_NULL_POINTER;
    a->audio = audio; a->batch = batch; a->samples = samples; a->row_stride = row_stride;
    a->window = window; a->n_fft = n_fft; a->logm = logn - 1; a->hop = hop;
    a->tpf = (n_fft / 8 > 16) ? n_fft / 8 : 16;   // threads per frame slot (n_fft / 8 radix-4 butterflies per pass)
    a->frames = (samples + hop - 1) / hop;  // utils.py:265: -(-signal_len // hop_length)
    return SOT_OK;
}

}  // namespace sot_stft

extern "C" {

int64_t sot_stft_frames(int64_t samples, int hop) { return (samples < 1 || hop < 1) ? 0 : (samples + hop - 1) / hop; }

int sot_stft_mag_forward(const float* audio, int64_t batch, int64_t samples, int64_t audio_row_stride, const float* window,
                         int n_fft, int hop, float* mag, void* stream)
{
    using namespace sot_stft;
    StftArgs a{};
    const int rc = fill_args(audio, batch, samples, audio_row_stride, window, n_fft, hop, &a);
    if (rc != SOT_OK) return rc;
    if (batch == 0) return SOT_OK;
    if (mag == nullptr) return SOT_ERR_NULL_POINTER;
    a.mag = mag;
    const int slots = kThreads / a.tpf, m = n_fft / 2;
    const size_t lds = sizeof(float2) * ((size_t)slots * m + m + 2);
    const int64_t work = (batch * a.frames + slots - 1) / slots;
    const int grid = (int)(work < 256 * 16 ? work : 256 * 16);
    (void)hipGetLastError();
    hipLaunchKernelGGL(stft_mag_forward_kernel, dim3(grid), dim3(kThreads), lds, reinterpret_cast<hipStream_t>(stream), a);
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
                          int n_fft, int hop, const float* grad_mag, float* grad_audio, void* workspace, size_t workspace_bytes,
                          void* stream)
{
    using namespace sot_stft;
    StftArgs a{};
    const int rc = fill_args(audio, batch, samples, audio_row_stride, window, n_fft, hop, &a);
    if (rc != SOT_OK) return rc;
    if (batch == 0) return SOT_OK;
    if (grad_mag == nullptr || grad_audio == nullptr || workspace == nullptr) return SOT_ERR_NULL_POINTER;
    if (workspace_bytes < sot_stft_backward_workspace_bytes(batch, samples, n_fft, hop)) return SOT_ERR_WORKSPACE;
    const int64_t span = n_fft + (int64_t)hop * (kFramesPerGroup - 1);
    if (span > 8192) return SOT_ERR_UNSUPPORTED_SIZE;   // the groups' gradients live in LDS
    a.grad_mag = grad_mag; a.grad_audio = grad_audio;
    a.partial = reinterpret_cast<float*>(workspace);
    a.groups = (a.frames + kFramesPerGroup - 1) / kFramesPerGroup;
    a.span = (int)span;
    const int slots = kThreads / a.tpf, m = n_fft / 2;
    const size_t lds = sizeof(float2) * ((size_t)slots * m + m + 2) + sizeof(float) * (size_t)slots * (size_t)span;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(stft_mag_backward_partial_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
            (void)hipGetLastError();
        attr_set = true;
    }
    const int64_t work = (batch * a.groups + slots - 1) / slots;
    const int grid = (int)(work < 256 * 16 ? work : 256 * 16);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    hipLaunchKernelGGL(stft_mag_backward_partial_kernel, dim3(grid), dim3(kThreads), lds, st, a);
    if (hipGetLastError() != hipSuccess) return SOT_ERR_LAUNCH;
    const int64_t total = batch * samples;
    const int grid2 = (int)((total + kThreads - 1) / kThreads < 256 * 32 ? (total + kThreads - 1) / kThreads : 256 * 32);
    hipLaunchKernelGGL(stft_overlap_add_kernel, dim3(grid2), dim3(kThreads), 0, st, a);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

size_t sot_spec_distance_workspace_bytes(void) { return sizeof(double) * (size_t)sot_stft::kDistBlocks; }

int sot_spec_distance_forward(const float* target, const float* value, int64_t count, float mag_weight, float logmag_weight,
                              float eps, int l2, float* out, void* workspace, size_t workspace_bytes, void* stream)
{
    using namespace sot_stft;
    if (count < 1) return SOT_ERR_BAD_SHAPE;
    if (target == nullptr || value == nullptr || out == nullptr || workspace == nullptr) return SOT_ERR_NULL_POINTER;
    if (workspace_bytes < sot_spec_distance_workspace_bytes()) return SOT_ERR_WORKSPACE;
    DistArgs a{};
    a.target = target; a.value = value; a.count = count; a.mag_weight = mag_weight; a.logmag_weight = logmag_weight; a.eps = eps;
    a.l2 = l2; a.partial = reinterpret_cast<double*>(workspace); a.out = out;
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

}  // extern "C"
