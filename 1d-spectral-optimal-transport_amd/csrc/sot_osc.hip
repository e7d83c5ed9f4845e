// sot_osc.hip -- MI355X (gfx950) kernels for the additive oscillator bank in front of the STFT in the reference's
// training step: ddsp.oscillator_bank (ddsp.py:208-263, use_angular_cumsum=False, sum_sinusoids=True)
// with remove_above_nyquist (ddsp.py:25-49):
//   a' = (f >= sr/2) ? 0 : a;   omega = (f * 2pi) / sr;   phase_t = fp32( sum_{i<=t} omega_i )  (fp64 accumulation, as
//   ATen's CPU cumsum);   audio_t = sum_k a'_{t,k} sin(phase_{t,k}).
// Layout: frequency / amplitude envelopes [batch, samples, sinusoids] (sinusoid innermost), audio [batch, samples].
//
// The time axis of every clip is cut into SEGMENTS of S samples (S a power of two, S * sinusoids <= 4096 floats); one
// 256-thread workgroup owns one segment of one clip, so the grid is clips x segments and every global access is a
// contiguous, fully coalesced copy of the segment's [S, sinusoids] tile between HBM and LDS.  Inside the tile the work
// items are (run of 8 consecutive samples, sinusoid) pairs: a thread sums its run's omegas in fp64, the runs before it
// in the segment are added from LDS, and the phase at the segment's start comes from a scan over per-segment totals:
//   forward : [segment totals] -> [exclusive scan over segments] -> [tile kernel: phases, a' sin, sum over sinusoids]
//   backward: [segment totals] -> [scan] -> [tile kernel: grad_amp = g sin, dphi = g a' cos, segment totals of dphi]
//             -> [reverse scan] -> [suffix kernel: grad_freq = (sum_{t' >= t} dphi) / sr * 2pi]
// (the gradient of a cumulative sum is the REVERSE cumulative sum).  All sums run in a fixed order: deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/sot_hip.h"

namespace sot_osc {

constexpr int kThreads = 256;
constexpr int kRun = 8;                       // consecutive samples per work item
constexpr int kTileElems = 4096;              // floats per envelope tile (S * sinusoids)
constexpr int kMaxSinusoids = kTileElems / kRun;
constexpr int64_t kMaxSamples = 1 << 20;
constexpr float kTwoPi = 6.283185307179586f;  // float32(2 * np.pi), as `frequency_envelopes * (2.0 * np.pi)` rounds it

enum Mode { kTotals = 0, kForward = 1, kBackward = 2 };

struct OscArgs {
    const float* freq; const float* amp; int64_t batch, samples; int sinusoids; float sample_rate;
    int seg_len; int64_t nseg;                      // S and the number of segments per clip
    double* phase0;                                 // [batch, nseg, sinusoids] segment totals -> phase at each segment's start
    double* dcarry;                                 // [batch, nseg, sinusoids] totals of dphi -> sum over all later segments
    float* audio;                                   // forward output [batch, samples]
    const float* grad_audio;                        // backward input  [batch, samples]
    float* grad_freq; float* grad_amp;              // backward outputs [batch, samples, sinusoids]; either may be null
};

__device__ __forceinline__ float omega_of(float f, float sr) { return (f * kTwoPi) / sr; }

// LDS: ls[slots*K] doubles | ls2[slots*K] doubles (backward) | tf[S*K] | ta[S*K] | tg[S]
inline size_t lds_bytes(int S, int K, int mode)
{
    const size_t items = (size_t)(S / kRun) * K;
    size_t b = items * sizeof(double) * (mode == kBackward ? 2 : 1) + (size_t)S * K * sizeof(float);
    if (mode != kTotals) b += (size_t)S * K * sizeof(float);
    if (mode == kBackward) b += (size_t)S * sizeof(float);
    return b;
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void oscillator_tile_kernel(const OscArgs a)
{
    extern __shared__ double smem[];
    const int K = a.sinusoids, S = a.seg_len, slots = S / kRun, items = slots * K, tile = S * K;
    double* ls = smem;
    double* ls2 = ls + items;
    float* tf = reinterpret_cast<float*>(ls + (MODE == kBackward ? 2 : 1) * items);
    float* ta = tf + tile;
    float* tg = ta + tile;
    const float sr = a.sample_rate, nyq = sr / 2.0f;

    const int64_t b = blockIdx.x / a.nseg, seg = blockIdx.x - b * a.nseg;
    const int64_t t_base = seg * S;
    const int rows = (int)((a.samples - t_base) < S ? (a.samples - t_base) : S);
    const int n = rows * K;
    const int64_t ebase = (b * a.samples + t_base) * K;     // first envelope element of the tile
    const int64_t wbase = (b * a.nseg + seg) * K;           // this segment's entry in the per-segment arrays

    for (int e = threadIdx.x; e < tile; e += kThreads) {
        tf[e] = e < n ? a.freq[ebase + e] : 0.0f;            // f = 0 past the clip's end: omega 0, amplitude 0
        if (MODE != kTotals) ta[e] = e < n ? a.amp[ebase + e] : 0.0f;
    }
    if (MODE == kBackward)
        for (int t = threadIdx.x; t < S; t += kThreads) tg[t] = t < rows ? a.grad_audio[b * a.samples + t_base + t] : 0.0f;
    __syncthreads();

    for (int i = threadIdx.x; i < items; i += kThreads) {
        const int slot = i / K, k = i - slot * K, base = slot * kRun * K + k;
        double local = 0.0;
#pragma unroll
        for (int j = 0; j < kRun; ++j) local += (double)omega_of(tf[base + j * K], sr);
        ls[i] = local;
    }
    __syncthreads();

    if (MODE == kTotals) {
        for (int k = threadIdx.x; k < K; k += kThreads) {
            double s = 0.0;
            for (int slot = 0; slot < slots; ++slot) s += ls[slot * K + k];
            a.phase0[wbase + k] = s;
        }
        return;
    }

    for (int i = threadIdx.x; i < items; i += kThreads) {
        const int slot = i / K, k = i - slot * K, base = slot * kRun * K + k;
        double run = a.nseg > 1 ? a.phase0[wbase + k] : 0.0;
        for (int s = 0; s < slot; ++s) run += ls[s * K + k];
        double dlocal = 0.0;
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
            const int idx = base + j * K;
            const float f = tf[idx];
            run += (double)omega_of(f, sr);
            const float ph = (float)run;
            const bool muted = f >= nyq;
            const float am = muted ? 0.0f : ta[idx];
            if (MODE == kForward) {
                ta[idx] = am * sinf(ph);
            } else {
                float sn, cs;
                sincosf(ph, &sn, &cs);
                const float g = tg[slot * kRun + j];
                const float dphi = (g * am) * cs;
                ta[idx] = muted ? 0.0f : g * sn;
                tf[idx] = dphi;
                dlocal += (double)dphi;
            }
        }
        if (MODE == kBackward) ls2[i] = dlocal;
    }
    __syncthreads();

    if (MODE == kForward) {
        float* dst = a.audio + b * a.samples + t_base;
        if (S >= kThreads) {
            for (int t = threadIdx.x; t < S; t += kThreads) {
                float s = 0.0f;
                for (int k = 0; k < K; ++k) s += ta[t * K + k];
                if (t < rows) dst[t] = s;
            }
        } else {
            const int tps = kThreads / S;                    // threads per sample: a power of two <= 32
            const int t = threadIdx.x / tps, sub = threadIdx.x - t * tps;
            float s = 0.0f;
            for (int k = sub; k < K; k += tps) s += ta[t * K + k];
            for (int off = tps >> 1; off >= 1; off >>= 1) s += __shfl_xor(s, off);
            if (sub == 0 && t < rows) dst[t] = s;
        }
    } else {
        for (int e = threadIdx.x; e < n; e += kThreads) {
            if (a.grad_amp) a.grad_amp[ebase + e] = ta[e];
            if (a.grad_freq) a.grad_freq[ebase + e] = tf[e];  // dphi for now; the suffix kernel turns it into the gradient
        }
        if (a.grad_freq)
            for (int k = threadIdx.x; k < K; k += kThreads) {
                double s = 0.0;
                for (int slot = 0; slot < slots; ++slot) s += ls2[slot * K + k];
                a.dcarry[wbase + k] = s;
            }
    }
}

// exclusive scan of ws[b, :, k] along the segment axis, in place: forward (phase at a segment's start) or reverse (sum over
// all LATER segments).  One thread per (clip, sinusoid); consecutive threads touch consecutive doubles.
__global__ __launch_bounds__(kThreads) void oscillator_scan_kernel(double* ws, int64_t batch, int64_t nseg, int K, int reverse)
{
    const int64_t id = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (id >= batch * K) return;
    const int64_t b = id / K, k = id - b * K;
    double* p = ws + b * nseg * K + k;
    double acc = 0.0;
    if (!reverse) {
        for (int64_t s = 0; s < nseg; ++s) { const double v = p[s * K]; p[s * K] = acc; acc += v; }
    } else {
        for (int64_t s = nseg - 1; s >= 0; --s) { const double v = p[s * K]; p[s * K] = acc; acc += v; }
    }
}

// grad_freq holds dphi; rewrite it in place as ((sum over this and all later samples of dphi) / sr) * 2pi
__global__ __launch_bounds__(kThreads) void oscillator_suffix_kernel(const OscArgs a)
{
    extern __shared__ double smem[];
    const int K = a.sinusoids, S = a.seg_len, slots = S / kRun, items = slots * K, tile = S * K;
    double* ls = smem;
    float* tf = reinterpret_cast<float*>(ls + items);
    const int64_t b = blockIdx.x / a.nseg, seg = blockIdx.x - b * a.nseg;
    const int64_t t_base = seg * S;
    const int rows = (int)((a.samples - t_base) < S ? (a.samples - t_base) : S);
    const int n = rows * K;
    float* g = a.grad_freq + (b * a.samples + t_base) * K;
    const int64_t wbase = (b * a.nseg + seg) * K;

    for (int e = threadIdx.x; e < tile; e += kThreads) tf[e] = e < n ? g[e] : 0.0f;
    __syncthreads();
    for (int i = threadIdx.x; i < items; i += kThreads) {
        const int slot = i / K, k = i - slot * K, base = slot * kRun * K + k;
        double local = 0.0;
#pragma unroll
        for (int j = 0; j < kRun; ++j) local += (double)tf[base + j * K];
        ls[i] = local;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < items; i += kThreads) {
        const int slot = i / K, k = i - slot * K, base = slot * kRun * K + k;
        double after = a.nseg > 1 ? a.dcarry[wbase + k] : 0.0;
        for (int s = slots - 1; s > slot; --s) after += ls[s * K + k];
#pragma unroll
        for (int j = kRun - 1; j >= 0; --j) {
            const int idx = base + j * K;
            after += (double)tf[idx];
            tf[idx] = ((float)after / a.sample_rate) * kTwoPi;   // d omega / d f as autograd applies it: (g / sr) * 2pi
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += kThreads) g[e] = tf[e];
}

// segment length: a power of two with S * K <= kTileElems, halved while the grid would leave most of the 256 CUs idle
inline int pick_segment(int64_t batch, int64_t samples, int K)
{
    int S = 512;
    while (S > kRun && (int64_t)S * K > kTileElems) S >>= 1;
    while (S > 64 && (int64_t)(S / kRun) * K >= 2 * kThreads && batch * ((samples + S - 1) / S) < 2048) S >>= 1;
    return S;
}

inline int check_common(int64_t batch, int64_t samples, int sinusoids, float sample_rate)
{
    if (batch < 0 || samples < 1 || sinusoids < 1 || !(sample_rate > 0.0f)) return SOT_ERR_BAD_SHAPE;
    if (samples > kMaxSamples || sinusoids > kMaxSinusoids) return SOT_ERR_UNSUPPORTED_SIZE;
    return SOT_OK;
}

inline size_t segment_array_bytes(int64_t batch, int64_t samples, int K)
{
    const int S = pick_segment(batch, samples, K);
    return (size_t)batch * (size_t)((samples + S - 1) / S) * (size_t)K * sizeof(double);
}

inline bool launched() { return hipGetLastError() == hipSuccess; }

// fills a.phase0 with the phase of every sinusoid at the start of every segment
inline bool launch_segment_starts(const OscArgs& a, hipStream_t st)
{
    if (a.nseg <= 1) return true;
    const unsigned grid = (unsigned)(a.batch * a.nseg);
    hipLaunchKernelGGL(oscillator_tile_kernel<kTotals>, dim3(grid), dim3(kThreads), lds_bytes(a.seg_len, a.sinusoids, kTotals), st, a);
    const unsigned sgrid = (unsigned)((a.batch * a.sinusoids + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(oscillator_scan_kernel, dim3(sgrid), dim3(kThreads), 0, st, a.phase0, a.batch, a.nseg, a.sinusoids, 0);
    return launched();
}

}  // namespace sot_osc

extern "C" {

size_t sot_oscillator_bank_workspace_bytes(int64_t batch, int64_t samples, int sinusoids)
{
    using namespace sot_osc;
    if (batch < 1 || samples < 1 || sinusoids < 1 || samples > kMaxSamples || sinusoids > kMaxSinusoids) return 0;
    return 2 * segment_array_bytes(batch, samples, sinusoids);
}

int sot_oscillator_bank_forward(const float* freq, const float* amp, int64_t batch, int64_t samples, int sinusoids, float sample_rate,
                                float* audio, void* workspace, size_t workspace_bytes, void* stream)
{
    using namespace sot_osc;
    if (const int rc = check_common(batch, samples, sinusoids, sample_rate)) return rc;
    if (batch == 0) return SOT_OK;
    if (freq == nullptr || amp == nullptr || audio == nullptr) return SOT_ERR_NULL_POINTER;
    OscArgs a{};
    a.freq = freq; a.amp = amp; a.batch = batch; a.samples = samples; a.sinusoids = sinusoids; a.sample_rate = sample_rate; a.audio = audio;
    a.seg_len = pick_segment(batch, samples, sinusoids);
    a.nseg = (samples + a.seg_len - 1) / a.seg_len;
    if (batch * a.nseg > 0x7fffffffLL) return SOT_ERR_UNSUPPORTED_SIZE;
    if (a.nseg > 1) {
        if (workspace == nullptr) return SOT_ERR_NULL_POINTER;
        if (workspace_bytes < segment_array_bytes(batch, samples, sinusoids)) return SOT_ERR_BAD_SHAPE;
        a.phase0 = static_cast<double*>(workspace);
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (!launch_segment_starts(a, st)) return SOT_ERR_LAUNCH;
    hipLaunchKernelGGL(oscillator_tile_kernel<kForward>, dim3((unsigned)(batch * a.nseg)), dim3(kThreads),
                       lds_bytes(a.seg_len, sinusoids, kForward), st, a);
    return launched() ? SOT_OK : SOT_ERR_LAUNCH;
}

int sot_oscillator_bank_backward(const float* freq, const float* amp, int64_t batch, int64_t samples, int sinusoids, float sample_rate,
                                 const float* grad_audio, float* grad_freq, float* grad_amp, void* workspace, size_t workspace_bytes,
                                 int workspace_from_forward, void* stream)
{
    using namespace sot_osc;
    if (const int rc = check_common(batch, samples, sinusoids, sample_rate)) return rc;
    if (batch == 0 || (grad_freq == nullptr && grad_amp == nullptr)) return SOT_OK;
    if (freq == nullptr || amp == nullptr || grad_audio == nullptr) return SOT_ERR_NULL_POINTER;
    OscArgs a{};
    a.freq = freq; a.amp = amp; a.batch = batch; a.samples = samples; a.sinusoids = sinusoids; a.sample_rate = sample_rate;
    a.grad_audio = grad_audio; a.grad_freq = grad_freq; a.grad_amp = grad_amp;
    a.seg_len = pick_segment(batch, samples, sinusoids);
    a.nseg = (samples + a.seg_len - 1) / a.seg_len;
    if (batch * a.nseg > 0x7fffffffLL) return SOT_ERR_UNSUPPORTED_SIZE;
    const size_t one = segment_array_bytes(batch, samples, sinusoids);
    if (workspace == nullptr) return SOT_ERR_NULL_POINTER;
    if (workspace_bytes < 2 * one) return SOT_ERR_BAD_SHAPE;
    a.phase0 = static_cast<double*>(workspace);
    a.dcarry = reinterpret_cast<double*>(static_cast<char*>(workspace) + one);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)(batch * a.nseg);
    (void)hipGetLastError();
    // the forward of the SAME envelopes left every segment's start phase in the first half of this workspace
    if (!workspace_from_forward && !launch_segment_starts(a, st)) return SOT_ERR_LAUNCH;
    hipLaunchKernelGGL(oscillator_tile_kernel<kBackward>, dim3(grid), dim3(kThreads), lds_bytes(a.seg_len, sinusoids, kBackward), st, a);
    if (grad_freq != nullptr) {
        if (a.nseg > 1) {
            const unsigned sgrid = (unsigned)((batch * sinusoids + kThreads - 1) / kThreads);
            hipLaunchKernelGGL(oscillator_scan_kernel, dim3(sgrid), dim3(kThreads), 0, st, a.dcarry, batch, a.nseg, sinusoids, 1);
        }
        hipLaunchKernelGGL(oscillator_suffix_kernel, dim3(grid), dim3(kThreads), lds_bytes(a.seg_len, sinusoids, kTotals), st, a);
    }
    return launched() ? SOT_OK : SOT_ERR_LAUNCH;
}

}  // extern "C"
