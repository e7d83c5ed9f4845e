// sot_osc.hip -- MI355X (gfx950) kernels for the additive oscillator bank in front of the STFT in the reference's
// training step: ddsp.oscillator_bank (ddsp.py:208-263, use_angular_cumsum=False, sum_sinusoids=True)
// with remove_above_nyquist (ddsp.py:25-49):
//   a' = (f >= sr/2) ? 0 : a;   omega = (f * 2pi) / sr;   phase_t = fp32( sum_{i<=t} omega_i )  (fp64 accumulation, as
//   ATen's CPU cumsum);   audio_t = sum_k a'_{t,k} sin(phase_{t,k}).
// Layout: frequency / amplitude envelopes [batch, samples, sinusoids] (sinusoid innermost), audio [batch, samples].
//
// One 256-thread workgroup per clip walks the time axis in tiles of 256 x kChunk samples; within a tile each thread owns
// kChunk consecutive samples.  Per sinusoid: thread-local fp64 sum of its omegas, wave scan (shuffles) + cross-wave
// offsets through LDS, then the thread re-walks its samples (phase -> sin -> amplitude) and accumulates the audio in
// registers.  The running phase of every sinusoid is carried from tile to tile in LDS (fp64).
// Backward: d audio / d a = sin(phase) (0 above Nyquist), d audio / d phase = a' cos(phase); the gradient w.r.t. omega
// is the REVERSE cumulative sum over time, so the tiles are walked backwards with a carried suffix sum (fp64); the phase
// at each tile's start comes from a forward pre-pass.  Every element is written by exactly one thread: deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/sot_hip.h"

namespace sot_osc {

constexpr int kThreads = 256;
constexpr int kChunk = 8;                    // consecutive samples per thread within a tile
constexpr int kTile = kThreads * kChunk;     // 2048 samples per tile
constexpr int kWaves = kThreads / 64;
constexpr int kMaxTiles = 64;                // samples <= 131072
constexpr float kTwoPi = 6.283185307179586f;  // float32(2 * np.pi), as `frequency_envelopes * (2.0 * np.pi)` rounds it

struct OscArgs {
    const float* freq; const float* amp; int64_t batch, samples; int sinusoids; float sample_rate;
    float* audio;                                   // forward output [batch, samples]
    const float* grad_audio;                        // backward input  [batch, samples]
    float* grad_freq; float* grad_amp;              // backward outputs [batch, samples, sinusoids]; either may be null
};

__device__ __forceinline__ float omega_of(float f, float sr) { return (f * kTwoPi) / sr; }

// inclusive scan of one double per thread over the workgroup (wave shuffles + LDS across the 4 waves); returns the
// EXCLUSIVE prefix of the calling thread and the workgroup total.  `scratch`: kWaves doubles.  Two barriers.
__device__ __forceinline__ double block_exclusive(double v, double* scratch, double& total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    __syncthreads();  // scratch free
    if (lane == 63) scratch[wv] = inc;
    __syncthreads();
    double before = 0.0, tot = 0.0;
    for (int w = 0; w < kWaves; ++w) { const double s = scratch[w]; if (w < wv) before += s; tot += s; }
    total = tot;
    return before + (inc - v);
}

// same for a SUFFIX (reverse) scan: returns the sum of the values of all threads AFTER the calling one
__device__ __forceinline__ double block_exclusive_suffix(double v, double* scratch, double& total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_down(inc, off);
        if (lane + off < 64) inc += o;
    }
    __syncthreads();
    if (lane == 0) scratch[wv] = inc;
    __syncthreads();
    double after = 0.0, tot = 0.0;
    for (int w = 0; w < kWaves; ++w) { const double s = scratch[w]; if (w > wv) after += s; tot += s; }
    total = tot;
    return after + (inc - v);
}

__global__ __launch_bounds__(kThreads) void oscillator_bank_forward_kernel(const OscArgs a)
{
    __shared__ double scratch[kWaves];
    extern __shared__ double carry[];   // [sinusoids]: each sinusoid's phase at the start of the current tile (fp64)
    const int K = a.sinusoids;
    const float nyq = a.sample_rate / 2.0f;
    for (int64_t b = blockIdx.x; b < a.batch; b += gridDim.x) {
        const float* fb = a.freq + b * a.samples * K;
        const float* ab = a.amp + b * a.samples * K;
        for (int64_t tile0 = 0; tile0 < a.samples; tile0 += kTile) {
            const int64_t t0 = tile0 + (int64_t)threadIdx.x * kChunk;
            float out[kChunk];
#pragma unroll
            for (int j = 0; j < kChunk; ++j) out[j] = 0.0f;
            for (int k = 0; k < K; ++k) {
                double local = 0.0;
#pragma unroll
                for (int j = 0; j < kChunk; ++j) {
                    const int64_t t = t0 + j;
                    if (t < a.samples) local += (double)omega_of(fb[t * K + k], a.sample_rate);
                }
                double total;
                double run = block_exclusive(local, scratch, total);
                run += (tile0 == 0) ? 0.0 : carry[k];   // phase carried over from the earlier tiles
#pragma unroll
                for (int j = 0; j < kChunk; ++j) {
                    const int64_t t = t0 + j;
                    if (t < a.samples) {
                        const float f = fb[t * K + k];
                        run += (double)omega_of(f, a.sample_rate);
                        const float am = (f >= nyq) ? 0.0f : ab[t * K + k];
                        out[j] += am * sinf((float)run);
                    }
                }
                __syncthreads();   // every thread has read carry[k]
                if (threadIdx.x == 0) carry[k] = ((tile0 == 0) ? 0.0 : carry[k]) + total;
            }
            float* dst = a.audio + b * a.samples;
#pragma unroll
            for (int j = 0; j < kChunk; ++j) {
                const int64_t t = t0 + j;
                if (t < a.samples) dst[t] = out[j];
            }
            __syncthreads();   // carry[] updates visible before the next tile reads them
        }
    }
}

__global__ __launch_bounds__(kThreads) void oscillator_bank_backward_kernel(const OscArgs a)
{
    __shared__ double scratch[kWaves];
    extern __shared__ double shm[];   // [kMaxTiles] phase at each tile's start, reused per sinusoid
    const int K = a.sinusoids;
    const float nyq = a.sample_rate / 2.0f;
    const float dscale_num = kTwoPi;   // d omega / d f = 2pi / sr, applied as (g / sr) * 2pi like autograd does
    const int64_t ntiles = (a.samples + kTile - 1) / kTile;
    for (int64_t b = blockIdx.x; b < a.batch; b += gridDim.x) {
        const float* fb = a.freq + b * a.samples * K;
        const float* ab = a.amp + b * a.samples * K;
        const float* gb = a.grad_audio + b * a.samples;
        float* gfb = a.grad_freq ? a.grad_freq + b * a.samples * K : nullptr;
        float* gab = a.grad_amp ? a.grad_amp + b * a.samples * K : nullptr;
        for (int k = 0; k < K; ++k) {
            // forward pre-pass: phase at the start of every tile
            double carry = 0.0;
            for (int64_t ti = 0; ti < ntiles; ++ti) {
                const int64_t t0 = ti * kTile + (int64_t)threadIdx.x * kChunk;
                double local = 0.0;
#pragma unroll
                for (int j = 0; j < kChunk; ++j) {
                    const int64_t t = t0 + j;
                    if (t < a.samples) local += (double)omega_of(fb[t * K + k], a.sample_rate);
                }
                double total;
                (void)block_exclusive(local, scratch, total);
                if (threadIdx.x == 0) shm[ti] = carry;
                carry += total;   // identical in every thread
            }
            __syncthreads();
            // reverse pass over the tiles with the carried suffix sum of d L / d phase
            double suffix = 0.0;
            for (int64_t ti = ntiles - 1; ti >= 0; --ti) {
                const int64_t t0 = ti * kTile + (int64_t)threadIdx.x * kChunk;
                float om[kChunk], dphi[kChunk];
                double local = 0.0;
#pragma unroll
                for (int j = 0; j < kChunk; ++j) {
                    const int64_t t = t0 + j;
                    om[j] = (t < a.samples) ? omega_of(fb[t * K + k], a.sample_rate) : 0.0f;
                    local += (double)om[j];
                }
                double total;
                double run = block_exclusive(local, scratch, total) + shm[ti];
                double dsum = 0.0;
#pragma unroll
                for (int j = 0; j < kChunk; ++j) {
                    const int64_t t = t0 + j;
                    dphi[j] = 0.0f;
                    if (t < a.samples) {
                        run += (double)om[j];
                        const float ph = (float)run;
                        const float f = fb[t * K + k];
                        const bool audible = !(f >= nyq);
                        const float g = gb[t];
                        if (gab) gab[t * K + k] = audible ? g * sinf(ph) : 0.0f;
                        dphi[j] = audible ? g * ab[t * K + k] * cosf(ph) : 0.0f;
                        dsum += (double)dphi[j];
                    }
                }
                double dtotal;
                double after = block_exclusive_suffix(dsum, scratch, dtotal) + suffix;   // sum over all later samples
                if (gfb) {
#pragma unroll
                    for (int j = kChunk - 1; j >= 0; --j) {
                        const int64_t t = t0 + j;
                        after += (double)dphi[j];
                        if (t < a.samples) gfb[t * K + k] = ((float)after / a.sample_rate) * dscale_num;
                    }
                }
                suffix += dtotal;
            }
            __syncthreads();   // shm[] is rewritten for the next sinusoid
        }
    }
}

}  // namespace sot_osc

extern "C" {

int sot_oscillator_bank_forward(const float* freq, const float* amp, int64_t batch, int64_t samples, int sinusoids, float sample_rate,
                                float* audio, void* stream)
{
    using namespace sot_osc;
    if (batch < 0 || samples < 1 || sinusoids < 1 || !(sample_rate > 0.0f)) return SOT_ERR_BAD_SHAPE;
    if (samples > (int64_t)kMaxTiles * kTile || sinusoids > 1024) return SOT_ERR_UNSUPPORTED_SIZE;
    if (batch == 0) return SOT_OK;
    if (freq == nullptr || amp == nullptr || audio == nullptr) return SOT_ERR_NULL_POINTER;
    OscArgs a{};
    a.freq = freq; a.amp = amp; a.batch = batch; a.samples = samples; a.sinusoids = sinusoids; a.sample_rate = sample_rate; a.audio = audio;
    const int grid = (int)(batch < 4096 ? batch : 4096);
    (void)hipGetLastError();
    hipLaunchKernelGGL(oscillator_bank_forward_kernel, dim3(grid), dim3(kThreads), sizeof(double) * (size_t)sinusoids,
                       reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

int sot_oscillator_bank_backward(const float* freq, const float* amp, int64_t batch, int64_t samples, int sinusoids, float sample_rate,
                                 const float* grad_audio, float* grad_freq, float* grad_amp, void* stream)
{
    using namespace sot_osc;
    if (batch < 0 || samples < 1 || sinusoids < 1 || !(sample_rate > 0.0f)) return SOT_ERR_BAD_SHAPE;
    if (samples > (int64_t)kMaxTiles * kTile || sinusoids > 1024) return SOT_ERR_UNSUPPORTED_SIZE;
    if (batch == 0 || (grad_freq == nullptr && grad_amp == nullptr)) return SOT_OK;
    if (freq == nullptr || amp == nullptr || grad_audio == nullptr) return SOT_ERR_NULL_POINTER;
    OscArgs a{};
    a.freq = freq; a.amp = amp; a.batch = batch; a.samples = samples; a.sinusoids = sinusoids; a.sample_rate = sample_rate;
    a.grad_audio = grad_audio; a.grad_freq = grad_freq; a.grad_amp = grad_amp;
    const int grid = (int)(batch < 4096 ? batch : 4096);
    (void)hipGetLastError();
    hipLaunchKernelGGL(oscillator_bank_backward_kernel, dim3(grid), dim3(kThreads), sizeof(double) * (size_t)kMaxTiles,
                       reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

}  // extern "C"
