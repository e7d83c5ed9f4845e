// sot_stft.hip -- MI355X (gfx950) kernels for the producer in front of the SOT loss: the magnitude STFT of the
// reference's `features.TorchSTFT` (features.py:85-113 -> compute_mag / stft, features.py:191-237; end padding
// utils.pad_for_stft, utils.py:252-275): torch.stft(center=False, normalized=True, onesided) of the end-padded signal,
// |.|, frames-major output [batch, frames, n_fft/2 + 1].  SURVEY §8f row 1.
//
// Forward: one workgroup per frame.  The windowed frame goes through an in-LDS radix-2 complex FFT of n_fft points
// (bit-reversed load, log2(n_fft) butterfly stages, twiddles from an LDS table built with sincospi), the first
// n_fft/2 + 1 bins are reduced to hypot(re, im) / sqrt(n_fft).
// Backward (closed form of abs o stft's autograd): per clip, frame by frame, recompute the frame's spectrum X, form
// Z_k = g_k X_k / |X_k| (0 where |X_k| = 0, torch's sgn(0)), inverse-transform the one-sided Z (other bins zero), take
// window * Re(.) / sqrt(n_fft) and overlap-add it into the clip's gradient, which is kept in LDS and written once:
// no atomics, deterministic.
// HBM traffic: forward reads n_fft samples per frame (L2-resident overlap) and writes n_fft/2+1 magnitudes; backward
// reads the audio and the magnitude gradients once and writes the audio gradient once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/sot_hip.h"

namespace sot_stft {

constexpr int kThreads = 256;      // forward: one 256-thread workgroup per frame
constexpr int kBwdThreads = 1024;  // backward: one 1024-thread workgroup per clip (its 2 x frames transforms run back to back)
constexpr int kMaxFft = 2048;
constexpr int kMaxClip = 8192;  // samples + end padding a backward workgroup can hold in LDS

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

// In-place radix-2 decimation-in-time FFT of n = 2^logn points held in LDS in BIT-REVERSED order on entry, natural
// order on exit.  inverse: conjugate twiddles (no 1/n).  Ends with a barrier.
template <int T>
__device__ __forceinline__ void fft_inplace(float2* z, const float2* tw, int n, int logn, bool inverse)
{
    for (int s = 1; s <= logn; ++s) {
        const int half = 1 << (s - 1);
        const int tstride = n >> s;  // twiddle index step: exp(-2 pi i pos / (2 half)) = T[pos * n / (2 half)]
        __syncthreads();
        for (int j = threadIdx.x; j < n / 2; j += T) {
            const int pos = j & (half - 1);
            const int i0 = ((j >> (s - 1)) << s) + pos;
            const int i1 = i0 + half;
            float2 w = tw[pos * tstride];
            if (inverse) w.y = -w.y;
            const float2 a = z[i0];
            const float2 b = cmul(z[i1], w);
            z[i0] = make_float2(a.x + b.x, a.y + b.y);
            z[i1] = make_float2(a.x - b.x, a.y - b.y);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int bitrev(int v, int logn) { return (int)(__brev((unsigned)v) >> (32 - logn)); }

struct StftArgs {
    const float* audio; int64_t batch, samples, row_stride;
    const float* window; int n_fft, logn, hop; int64_t frames;
    float* mag;                 // forward output [batch, frames, n_fft/2+1]
    const float* grad_mag;      // backward input, same shape
    float* grad_audio;          // backward output [batch, samples] (contiguous)
};

__global__ __launch_bounds__(kThreads) void stft_mag_forward_kernel(const StftArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float2* const z = reinterpret_cast<float2*>(smem_f);
    float2* const tw = z + a.n_fft;
    const int n = a.n_fft, nb = n / 2 + 1;
    const float scale = 1.0f / sqrtf((float)n);  // normalized=True: frame_length^-0.5
    build_twiddles<kThreads>(tw, n);
    for (int64_t fr = blockIdx.x; fr < a.batch * a.frames; fr += gridDim.x) {
        const int64_t b = fr / a.frames, f = fr - b * a.frames;
        const float* src = a.audio + b * a.row_stride;
        const int64_t t0 = f * a.hop;
        __syncthreads();  // previous frame's reads of z are done
        for (int i = threadIdx.x; i < n; i += kThreads) {
            const int64_t t = t0 + i;
            const float v = (t < a.samples) ? src[t] * a.window[i] : 0.0f;  // end padding: zeros (utils.py:252-275)
            z[bitrev(i, a.logn)] = make_float2(v, 0.0f);
        }
        fft_inplace<kThreads>(z, tw, n, a.logn, false);
        float* dst = a.mag + fr * nb;
        for (int k = threadIdx.x; k < nb; k += kThreads) dst[k] = hypotf(z[k].x, z[k].y) * scale;
    }
}

__global__ __launch_bounds__(kBwdThreads) void stft_mag_backward_kernel(const StftArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float2* const z = reinterpret_cast<float2*>(smem_f);
    float2* const tw = z + a.n_fft;
    float* const acc = reinterpret_cast<float*>(tw + a.n_fft / 2);  // gradient of the (padded) clip
    const int n = a.n_fft, nb = n / 2 + 1;
    const float scale = 1.0f / sqrtf((float)n);
    const int64_t padded = a.n_fft + a.hop * (a.frames - 1);
    build_twiddles<kBwdThreads>(tw, n);
    for (int64_t b = blockIdx.x; b < a.batch; b += gridDim.x) {
        const float* src = a.audio + b * a.row_stride;
        __syncthreads();
        for (int64_t t = threadIdx.x; t < padded; t += kBwdThreads) acc[t] = 0.0f;
        for (int64_t f = 0; f < a.frames; ++f) {
            const int64_t t0 = f * a.hop;
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += kBwdThreads) {
                const int64_t t = t0 + i;
                const float v = (t < a.samples) ? src[t] * a.window[i] : 0.0f;
                z[bitrev(i, a.logn)] = make_float2(v, 0.0f);
            }
            fft_inplace<kBwdThreads>(z, tw, n, a.logn, false);  // X (unscaled)
            // Z_k = g_k * X_k / |X_k| for the one-sided bins, 0 elsewhere; each thread rewrites the natural-order
            // spectrum into bit-reversed order for the inverse transform through registers (two passes, barrier between)
            const float* g = a.grad_mag + (b * a.frames + f) * nb;
            float2 zk[kMaxFft / kBwdThreads];
#pragma unroll
            for (int r = 0; r < kMaxFft / kBwdThreads; ++r) {
                const int k = threadIdx.x + r * kBwdThreads;
                float2 v = make_float2(0.0f, 0.0f);
                if (k < nb) {
                    const float2 x = z[k];
                    const float m = hypotf(x.x, x.y);
                    if (m > 0.0f) { const float c = g[k] / m; v = make_float2(c * x.x, c * x.y); }
                }
                zk[r] = v;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kMaxFft / kBwdThreads; ++r) {
                const int k = threadIdx.x + r * kBwdThreads;
                if (k < n) z[bitrev(k, a.logn)] = zk[r];
            }
            fft_inplace<kBwdThreads>(z, tw, n, a.logn, true);  // c_i = sum_k Z_k e^{+2 pi i k i / n}
            for (int i = threadIdx.x; i < n; i += kBwdThreads) acc[t0 + i] += a.window[i] * z[i].x * scale;  // disjoint i per thread
        }
        __syncthreads();
        float* dst = a.grad_audio + b * a.samples;
        for (int64_t t = threadIdx.x; t < a.samples; t += kBwdThreads) dst[t] = acc[t];
    }
}

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
    a->window = window; a->n_fft = n_fft; a->logn = logn; a->hop = hop;
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
    const size_t lds = sizeof(float2) * ((size_t)n_fft + n_fft / 2);
    const int64_t work = batch * a.frames;
    const int grid = (int)(work < 256 * 16 ? work : 256 * 16);
    (void)hipGetLastError();
    hipLaunchKernelGGL(stft_mag_forward_kernel, dim3(grid), dim3(kThreads), lds, reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

int sot_stft_mag_backward(const float* audio, int64_t batch, int64_t samples, int64_t audio_row_stride, const float* window,
                          int n_fft, int hop, const float* grad_mag, float* grad_audio, void* stream)
{
    using namespace sot_stft;
    StftArgs a{};
    const int rc = fill_args(audio, batch, samples, audio_row_stride, window, n_fft, hop, &a);
    if (rc != SOT_OK) return rc;
    if (batch == 0) return SOT_OK;
    if (grad_mag == nullptr || grad_audio == nullptr) return SOT_ERR_NULL_POINTER;
    const int64_t padded = n_fft + (int64_t)hop * (a.frames - 1);
    if (padded > kMaxClip) return SOT_ERR_UNSUPPORTED_SIZE;
    a.grad_mag = grad_mag; a.grad_audio = grad_audio;
    const size_t lds = sizeof(float2) * ((size_t)n_fft + n_fft / 2) + sizeof(float) * (size_t)padded;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(stft_mag_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                64 * 1024) != hipSuccess)
            (void)hipGetLastError();
        attr_set = true;
    }
    const int grid = (int)(batch < 1024 ? batch : 1024);
    (void)hipGetLastError();
    hipLaunchKernelGGL(stft_mag_backward_kernel, dim3(grid), dim3(kBwdThreads), lds, reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

}  // extern "C"
