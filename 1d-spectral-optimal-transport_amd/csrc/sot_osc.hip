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
#include <mutex>

#include "../../include/sot_hip.h"

namespace sot_osc {

constexpr int kThreads = 256;
constexpr int kRun = 8;                       // consecutive samples per work item
constexpr int kTileElems = 4096;              // most floats per envelope tile (S * sinusoids)
#ifndef SOT_OSC_TILE
#define SOT_OSC_TILE 2048                     /* preferred floats per envelope tile, see pick_segment */
#endif
constexpr int kMaxSinusoids = kTileElems / kRun;
constexpr int64_t kMaxSamples = 1 << 20;
constexpr float kTwoPi = 6.283185307179586f;  // float32(2 * np.pi), as `frequency_envelopes * (2.0 * np.pi)` rounds it

enum Mode { kTotals = 0, kForward = 1, kBackward = 2, kBackwardFrames = 3 };   // 3: backward reduced to the frame-rate controls in the tile

// frame-rate controls of the synthesiser (see the envelope section below for the arithmetic)
struct EnvArgs {
    const float* amp; const float* freq;     // [batch, frames, K]; freq [batch, frames, 1] when harmonic
    const float* window;                     // [2 hop]
    int64_t batch; int frames, K, harmonic; int64_t samples; int hop; float nyquist, scale;
    float inv_hop, inv_K;                    // 1 / hop, 1 / K for quotient_of()
    float* amp_env; float* freq_env;         // forward outputs [batch, samples, K]
    const float* g_amp_env; const float* g_freq_env;   // backward inputs (either may be null)
    float* g_amp; float* g_freq;             // backward outputs [batch, frames, K] / [batch, frames, K or 1] (either may be null)
};

// floor(t / d) for 0 <= t < 2^20 from inv = 1.0f / d: (t + 0.5) / d is at least 0.5 / d away from every integer and the two
// float roundings move it by less than 2^-22 t / d, so truncation gives the exact quotient (checked exhaustively on the host)
__device__ __forceinline__ int quotient_of(int t, float inv) { return (int)(((float)t + 0.5f) * inv); }

// frame-rate frequency of sinusoid k in frame f of the clip whose controls start at freq_b
__device__ __forceinline__ float frame_freq(const EnvArgs& a, const float* freq_b, int f, int k)
{
    return a.harmonic ? freq_b[f] * (float)(k + 1) : freq_b[f * a.K + k];
}

__device__ __forceinline__ const float* clip_freq(const EnvArgs& a, int64_t b) { return a.freq + b * a.frames * (a.harmonic ? 1 : a.K); }
__device__ __forceinline__ const float* clip_amp(const EnvArgs& a, int64_t b) { return a.amp + b * a.frames * a.K; }

__device__ __forceinline__ void linear_taps(const EnvArgs& a, int t, int& i0, int& i1, float& l0, float& l1)
{
    float s = fmaf(a.scale, (float)t + 0.5f, -0.5f);   // one rounding, like the contracted expression of ATen's CPU build
    s = s < 0.0f ? 0.0f : s;
    i0 = min((int)s, a.frames - 1);
    i1 = min(i0 + 1, a.frames - 1);
    l1 = fminf(fmaxf(s - (float)i0, 0.0f), 1.0f);
    l0 = 1.0f - l1;
}

// the two envelopes of sample t, sinusoid k of clip b (what synth_envelopes_forward_kernel stores)
__device__ __forceinline__ void envelopes_at(const EnvArgs& a, const float* amp_b, const float* freq_b, int t, int k, bool want_amp,
                                             float& f, float& am)
{
    int i0, i1; float l0, l1;
    linear_taps(a, t, i0, i1, l0, l1);
    f = fmaf(l0, frame_freq(a, freq_b, i0, k), l1 * frame_freq(a, freq_b, i1, k));
    if (!want_amp) return;
    const int fa = quotient_of(t, a.inv_hop), u = t - fa * a.hop, fb = min(fa + 1, a.frames - 1);
    float A0 = amp_b[fa * a.K + k], A1 = amp_b[fb * a.K + k];
    if (frame_freq(a, freq_b, fa, k) >= a.nyquist) A0 = 0.0f;
    if (frame_freq(a, freq_b, fb, k) >= a.nyquist) A1 = 0.0f;
    am = A0 * a.window[u + a.hop] + A1 * a.window[u];
}

struct OscArgs {
    const float* freq; const float* amp; int64_t batch, samples; int sinusoids; float sample_rate;
    int seg_len; int64_t nseg;                      // S and the number of segments per clip
    double* phase0;                                 // [batch, nseg, sinusoids] segment totals -> phase at each segment's start
    double* dcarry;                                 // [batch, nseg, sinusoids] totals of dphi -> sum over all later segments
    float* audio;                                   // forward output [batch, samples]
    const float* grad_audio;                        // backward input  [batch, samples]
    float* grad_freq; float* grad_amp;              // backward outputs [batch, samples, sinusoids]; either may be null
    int scanned;                                    // phase0 / dcarry hold scanned values (oscillator_scan_kernel ran) rather than raw totals
    EnvArgs ctl;                                    // CTL kernels: the envelopes are evaluated from these frame-rate controls
    // kBackwardFrames: cumulative linear-tap weights and Hann-window weights [frames, 3 hop], per-segment partial sums
    // [batch, nseg, nslot, sinusoids] x 2
    const double* wtab; const float* atab; double* part_amp; double* part_freq; int nslot;
};

__device__ __forceinline__ float omega_of(float f, float sr) { return (f * kTwoPi) / sr; }

// Clips of at most kFusedScanSegments segments would skip the two scan launches: a tile adds up the totals of the segments before it
// (after it, for the reverse direction) itself, in the order oscillator_scan_kernel uses -- the same doubles, bit for bit.  Measured
// with 64 (256 clips x 8 segments): the tile kernels lose more (forward 45.3 -> 50.5 us, backward 74.5 -> 79.2 us) than the 4.7 us
// scan launches they replace: 0 = always scan.
constexpr int64_t kFusedScanSegments = 0;

__device__ __forceinline__ double sum_before(const double* totals, int64_t b, int64_t nseg, int64_t seg, int K, int k, int scanned)
{
    if (nseg <= 1) return 0.0;
    const double* p = totals + b * nseg * K + k;
    if (scanned) return p[seg * K];
    double acc = 0.0;
    for (int64_t s = 0; s < seg; ++s) acc += p[s * K];
    return acc;
}

__device__ __forceinline__ double sum_after(const double* totals, int64_t b, int64_t nseg, int64_t seg, int K, int k, int scanned)
{
    if (nseg <= 1) return 0.0;
    const double* p = totals + b * nseg * K + k;
    if (scanned) return p[seg * K];
    double acc = 0.0;
    for (int64_t s = nseg - 1; s > seg; --s) acc += p[s * K];
    return acc;
}

// LDS: ls[slots*K] doubles | ls2[slots*K] doubles (backward) | tf[S*K] | ta[S*K] | tg[S]
inline size_t lds_bytes(int S, int K, int mode)
{
    const size_t items = (size_t)(S / kRun) * K;
    size_t b = items * sizeof(double) * (mode == kBackward ? 2 : 1) + (size_t)S * K * sizeof(float);
    if (mode != kTotals) b += (size_t)S * K * sizeof(float);
    if (mode == kBackward) b += (size_t)S * sizeof(float);
    return b;
}

// frame slots a segment of S samples can touch: the frames whose 3-hop support [(f - 1) hop, (f + 2) hop) meets the segment
inline int frame_slots(int S, int hop) { return (S + hop - 1) / hop + 3; }

// kBackwardFrames: the backward tile + two arrays of partial sums (one entry per (frame slot, sinusoid, sub-range))
inline size_t lds_bytes_frames(int S, int K, int hop)
{
    const size_t pairs = (size_t)frame_slots(S, hop) * K;
    return lds_bytes(S, K, kBackward) + 8 + 2 * sizeof(double) * (pairs > (size_t)kThreads ? pairs : (size_t)kThreads);
}

template <int MODE, bool CTL = false>
__global__ __launch_bounds__(kThreads) void oscillator_tile_kernel(const OscArgs a)
{
    extern __shared__ double smem[];
    const int K = a.sinusoids, S = a.seg_len, slots = S / kRun, items = slots * K, tile = S * K;
    double* ls = smem;
    double* ls2 = ls + items;
    constexpr bool BWD = MODE == kBackward || MODE == kBackwardFrames;
    float* tf = reinterpret_cast<float*>(ls + (BWD ? 2 : 1) * items);
    float* ta = tf + tile;
    float* tg = ta + tile;
    const float sr = a.sample_rate, nyq = sr / 2.0f;

    const int64_t b = blockIdx.x / a.nseg, seg = blockIdx.x - b * a.nseg;
    const int64_t t_base = seg * S;
    const int rows = (int)((a.samples - t_base) < S ? (a.samples - t_base) : S);
    const int n = rows * K;
    const int64_t ebase = (b * a.samples + t_base) * K;     // first envelope element of the tile
    const int64_t wbase = (b * a.nseg + seg) * K;           // this segment's entry in the per-segment arrays

    if (CTL) {
        // the synthesiser's form: no envelope arrays exist; every tile element is evaluated from the frame-rate controls
        // (the same float32 operations as synth_envelopes_forward_kernel: identical values)
        const float* amp_b = clip_amp(a.ctl, b);
        const float* freq_b = clip_freq(a.ctl, b);
#pragma unroll 4   // (runtime trip count: unrolled so that the loads of four elements are in flight together)
        for (int e = threadIdx.x; e < tile; e += kThreads) {
            const int tl = quotient_of(e, a.ctl.inv_K), k = e - tl * K;
            float f = 0.0f, am = 0.0f;
            if (e < n) envelopes_at(a.ctl, amp_b, freq_b, (int)t_base + tl, k, MODE != kTotals, f, am);
            tf[e] = f;
            if (MODE != kTotals) ta[e] = am;
        }
    } else {
        // all loads of up to eight elements per thread first, then the LDS stores (a plain loop with this runtime trip count waits for
        // every element before it requests the next)
        for (int e0 = threadIdx.x; e0 < tile; e0 += 8 * kThreads) {
            float f[8], am[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = e0 + j * kThreads;
                f[j] = e < n ? a.freq[ebase + e] : 0.0f;        // f = 0 past the clip's end: omega 0, amplitude 0
                am[j] = (MODE != kTotals && e < n) ? a.amp[ebase + e] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = e0 + j * kThreads;
                if (e < tile) { tf[e] = f[j]; if (MODE != kTotals) ta[e] = am[j]; }
            }
        }
    }
    if (BWD)
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
        double run = sum_before(a.phase0, b, a.nseg, seg, K, k, a.scanned);
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
        if (BWD) ls2[i] = dlocal;
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
    } else if (MODE == kBackward) {
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
    } else {
        // The synthesiser's backward: nothing at sample rate leaves the tile.  The gradient of a frame-rate control is a
        // weighted sum over the samples of its 3-hop support:
        //   amplitude: sum_t (g sin)_t wamp_f(t)                               (the two Hann-window halves),
        //   frequency: sum_t l_f(t) sum_{t' >= t} dphi_t'  =  sum_t' dphi_t' W_f(t'),   W_f(t') = sum_{t <= t'} l_f(t)
        // (order of summation exchanged: W_f is the cumulative tap weight a.wtab holds; it stays at its last value behind the
        // support, where whole segments enter through their dphi totals -- synth_frames_reduce_kernel).  Here: the part of both
        // sums that lies in this tile, for every frame whose support meets it; fixed order.
        const EnvArgs& e = a.ctl;
        const bool want_amp = a.part_amp != nullptr, want_freq = a.part_freq != nullptr;
        if (want_freq)
            for (int k = threadIdx.x; k < K; k += kThreads) {
                double s = 0.0;
                for (int slot = 0; slot < slots; ++slot) s += ls2[slot * K + k];
                a.dcarry[wbase + k] = s;
            }
        double* pa_l = reinterpret_cast<double*>(reinterpret_cast<uintptr_t>(tg + S + 1) & ~(uintptr_t)7);
        const int hop = e.hop, tb = (int)t_base, T = (int)a.samples;
        const int f_lo = max(0, quotient_of(tb, e.inv_hop) - 1);
        const int f_hi = min(e.frames - 1, quotient_of(tb + rows - 1, e.inv_hop) + 1);
        const int P = (f_hi - f_lo + 1) * K;                        // (frame slot, sinusoid) pairs; <= a.nslot * K
        const int subs = P >= kThreads ? 1 : kThreads / P;          // sub-ranges of the tile's samples per pair
        const int chunk = (rows + subs - 1) / subs;
        double* pf_l = pa_l + (P > kThreads ? P : kThreads);
        for (int item = threadIdx.x; item < P * subs; item += kThreads) {
            const int sub = item / P, pr = item - sub * P, sl = pr / K, k = pr - sl * K, f = f_lo + sl;
            const int ts = max(0, (f - 1) * hop), te = min(T, (f + 2) * hop);
            const int lo = max(tb + sub * chunk, ts), hi = min(tb + min(rows, (sub + 1) * chunk), want_freq ? T : te);
            const double* wt = a.wtab + (int64_t)f * 3 * hop;
            const float* at = a.atab + (int64_t)f * 3 * hop;
            double sa = 0.0, sf = 0.0;
            // (8 table loads in flight per thread: the tables come from L2, one load at a time would cost its full latency per sample)
            if (want_amp) {
                const int hia = min(hi, te);
                int t = lo;
                for (; t + 8 <= hia; t += 8) {
                    float w[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = at[t + j - ts];
#pragma unroll
                    for (int j = 0; j < 8; ++j) sa += (double)(ta[(t + j - tb) * K + k] * w[j]);
                }
                for (; t < hia; ++t) sa += (double)(ta[(t - tb) * K + k] * at[t - ts]);
            }
            if (want_freq) {
                int t = lo;
                for (; t + 8 <= hi; t += 8) {
                    double w[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = wt[min(t + j - ts, 3 * hop - 1)];
#pragma unroll
                    for (int j = 0; j < 8; ++j) sf += (double)tf[(t + j - tb) * K + k] * w[j];
                }
                for (; t < hi; ++t) sf += (double)tf[(t - tb) * K + k] * wt[min(t - ts, 3 * hop - 1)];
            }
            pa_l[item] = sa; pf_l[item] = sf;
        }
        __syncthreads();
        for (int pr = threadIdx.x; pr < P; pr += kThreads) {
            double sa = 0.0, sf = 0.0;
            for (int sub = 0; sub < subs; ++sub) { sa += pa_l[sub * P + pr]; sf += pf_l[sub * P + pr]; }
            const int sl = pr / K, k = pr - sl * K;
            const int64_t o = ((b * a.nseg + seg) * a.nslot + sl) * K + k;
            if (want_amp) a.part_amp[o] = sa;
            if (want_freq) a.part_freq[o] = sf;
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

    for (int e0 = threadIdx.x; e0 < tile; e0 += 8 * kThreads) {   // loads first, as in the tile kernel
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int e = e0 + j * kThreads; v[j] = e < n ? g[e] : 0.0f; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int e = e0 + j * kThreads; if (e < tile) tf[e] = v[j]; }
    }
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
        double after = sum_after(a.dcarry, b, a.nseg, seg, K, k, a.scanned);
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
    // preferred tile: 2048 floats per envelope (18 KB of LDS forward, 25 KB backward: 8 / 6 resident workgroups per CU instead of 4 / 3).
    // 256 clips x 4096 samples x 8 partials (tools/ab_synth.py): forward 64.5 -> 57.3 us, backward 79.2 -> 66.8 us; 1024: 65.7 / 79.6 us
    while (S > 64 && (int64_t)S * K > SOT_OSC_TILE) S >>= 1;
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
    if (a.ctl.amp != nullptr)
        hipLaunchKernelGGL((oscillator_tile_kernel<kTotals, true>), dim3(grid), dim3(kThreads), lds_bytes(a.seg_len, a.sinusoids, kTotals), st, a);
    else
        hipLaunchKernelGGL(oscillator_tile_kernel<kTotals>, dim3(grid), dim3(kThreads), lds_bytes(a.seg_len, a.sinusoids, kTotals), st, a);
    if (a.scanned) {
        const unsigned sgrid = (unsigned)((a.batch * a.sinusoids + kThreads - 1) / kThreads);
        hipLaunchKernelGGL(oscillator_scan_kernel, dim3(sgrid), dim3(kThreads), 0, st, a.phase0, a.batch, a.nseg, a.sinusoids, 0);
    }
    return launched();
}

// ---------------------------------------------------------------------------------------------
// Envelope upsampling of the synthesiser in front of the oscillator bank (synths.Sinusoidal.get_controls / get_signal,
// synths.py:62-113): frame-rate controls [batch, frames, sinusoids] -> sample-rate envelopes [batch, samples, sinusoids].
//   frequencies: harmonic -> f0 * k (ddsp.get_harmonic_frequencies, ddsp.py:6-22);  amplitudes of partials whose FRAME-rate
//   frequency is at or above Nyquist are zeroed (ddsp.remove_above_nyquist, ddsp.py:25-49);
//   amplitudes : ddsp.resample(method="window", add_endpoint=True) = half-overlapping Hann windows (ddsp.py:121-205): with
//                hop = samples / frames, sample t = a hop + u gets  A[a] w[u + hop] + A[min(a + 1, frames - 1)] w[u]  (two products,
//                one sum -- what the reference's fold() adds up); `window` is torch.hann_window(2 hop) as the caller computed it;
//   frequencies: ddsp.resample(method="bilinear", add_endpoint=True) = F.interpolate(align_corners=False): source index
//                s = max(fma(scale, t + 0.5, -0.5), 0), scale = frames / samples, i0 = min(int(s), frames - 1), l = s - i0, and -- the
//                operation order of ATen's CPU kernel, found by matching its output bit for bit --  fma(1 - l, F[i0], l * F[i1]).
// The backward hands every frame the weighted sum of the gradients of the samples it contributed to (fixed order: deterministic).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void synth_envelopes_forward_kernel(const EnvArgs a)
{
    const int64_t per_clip = a.samples * a.K;            // < 2^29
    for (int64_t b = blockIdx.y; b < a.batch; b += gridDim.y) {
        const float* amp_b = clip_amp(a, b);
        const float* freq_b = clip_freq(a, b);
        float* amp_env = a.amp_env + b * per_clip;
        float* freq_env = a.freq_env + b * per_clip;
        for (unsigned e = blockIdx.x * kThreads + threadIdx.x; e < (unsigned)per_clip; e += gridDim.x * kThreads) {
            const int t = (int)(e / (unsigned)a.K), k = (int)(e - (unsigned)t * (unsigned)a.K);
            float f, am;
            envelopes_at(a, amp_b, freq_b, t, k, true, f, am);
            amp_env[e] = am;
            freq_env[e] = f;
        }
    }
}

// one workgroup per (clip, frame): thread (sub, k) adds up its share of the frame's samples in ascending order, thread k then adds
// the `subs` partial sums in order
__global__ __launch_bounds__(kThreads) void synth_envelopes_backward_kernel(const EnvArgs a)
{
    extern __shared__ double red[];                     // [2][subs * K]
    const int64_t b = blockIdx.x / a.frames;
    const int f = (int)(blockIdx.x - b * a.frames);
    const int K = a.K, subs = max(1, kThreads / K);
    const float* freq_b = clip_freq(a, b);
    const int tlo = (int)max((int64_t)0, (int64_t)(f - 1) * a.hop), thi = (int)min(a.samples, (int64_t)(f + 2) * a.hop);
    const int span = thi - tlo, per = (span + subs - 1) / subs;
    for (int item = threadIdx.x; item < subs * K; item += kThreads) {
        const int k = item % K, sub = item / K;
        double ga = 0.0, gf = 0.0;
        const int t1 = min(thi, tlo + (sub + 1) * per);
        for (int t = tlo + sub * per; t < t1; ++t) {
            const int64_t e = (b * a.samples + t) * K + k;
            if (a.g_amp_env != nullptr) {
                const int fa = quotient_of(t, a.inv_hop), u = t - fa * a.hop, fb = min(fa + 1, a.frames - 1);
                const float g = a.g_amp_env[e];
                if (fa == f) ga += (double)(g * a.window[u + a.hop]);
                if (fb == f) ga += (double)(g * a.window[u]);
            }
            if (a.g_freq_env != nullptr) {
                int i0, i1; float l0, l1;
                linear_taps(a, t, i0, i1, l0, l1);
                const float g = a.g_freq_env[e];
                if (i0 == f) gf += (double)(g * l0);
                if (i1 == f) gf += (double)(g * l1);
            }
        }
        red[item] = ga; red[subs * K + item] = gf;
    }
    __syncthreads();
    double f0_sum = 0.0;   // harmonic: d/d f0 = sum_k (k + 1) d/d f_k, added in ascending k by thread 0
    for (int k = threadIdx.x; k < K; k += kThreads) {
        double ga = 0.0, gf = 0.0;
        for (int sub = 0; sub < subs; ++sub) { ga += red[sub * K + k]; gf += red[subs * K + sub * K + k]; }
        if (a.g_amp != nullptr) a.g_amp[(b * a.frames + f) * K + k] = (frame_freq(a, freq_b, f, k) >= a.nyquist) ? 0.0f : (float)ga;
        if (a.g_freq != nullptr && !a.harmonic) a.g_freq[(b * a.frames + f) * K + k] = (float)gf;
        red[k] = gf;       // this thread's own slot of sub-range 0: safe to overwrite after it has read its column
    }
    if (a.g_freq != nullptr && a.harmonic) {
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 0; k < K; ++k) f0_sum += (double)(k + 1) * red[k];
            a.g_freq[b * a.frames + f] = (float)f0_sum;
        }
    }
}

// The weights sample t = ts(f) + j puts on frame f, for the 3 hop samples from ts(f) = max(0, (f - 1) hop) on (the same for every
// clip and sinusoid; zero past the clip's end): atab [frames, 3 hop] floats -- the Hann-window weight of the amplitude upsampling
// (both halves added where the held last frame makes them meet); wtab [frames, 3 hop] doubles -- the CUMULATIVE linear-
// interpolation weight W_f(j) = sum_{j' <= j} l_f(ts + j').  One workgroup per frame.
__global__ __launch_bounds__(kThreads) void synth_tap_table_kernel(const EnvArgs a, double* wtab, float* atab)
{
    __shared__ double sums[kThreads];
    const int f = blockIdx.x, n = 3 * a.hop, ts = max(0, (f - 1) * a.hop), T = (int)a.samples;
    const int chunk = (n + kThreads - 1) / kThreads, j0 = min(n, (int)threadIdx.x * chunk), j1 = min(n, j0 + chunk);
    auto weight = [&](int t) -> double {
        if (t >= T) return 0.0;
        int i0, i1; float l0, l1;
        linear_taps(a, t, i0, i1, l0, l1);
        return (double)((i0 == f ? l0 : 0.0f) + (i1 == f ? l1 : 0.0f));
    };
    double s = 0.0;
    for (int j = j0; j < j1; ++j) s += weight(ts + j);
    sums[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double run = 0.0;
        for (int i = 0; i < kThreads; ++i) { const double v = sums[i]; sums[i] = run; run += v; }
    }
    __syncthreads();
    double run = sums[threadIdx.x];
    for (int j = j0; j < j1; ++j) {
        const int t = ts + j;
        run += weight(t);
        wtab[(int64_t)f * n + j] = run;
        float w = 0.0f;
        if (t < T) {
            const int fa = quotient_of(t, a.inv_hop), u = t - fa * a.hop, fb = min(fa + 1, a.frames - 1);
            w = (fa == f ? a.window[u + a.hop] : 0.0f) + (fb == f ? a.window[u] : 0.0f);
        }
        atab[(int64_t)f * n + j] = w;
    }
}

// gradients of the frame-rate controls from the per-segment partial sums of oscillator_tile_kernel<kBackwardFrames>: one workgroup
// per (clip, frame), one thread per sinusoid; segments are added in ascending order.
__global__ __launch_bounds__(64) void synth_frames_reduce_kernel(const OscArgs a, float* g_amp, float* g_freq)
{
    extern __shared__ double fsum[];     // [K]: harmonic only
    const EnvArgs& e = a.ctl;
    const int64_t b = blockIdx.x / e.frames;
    const int f = (int)(blockIdx.x - b * e.frames);
    const int K = e.K, hop = e.hop, S = a.seg_len, T = (int)a.samples;
    const float* freq_b = clip_freq(e, b);
    const int ts = max(0, (f - 1) * hop), te = min(T, (f + 2) * hop);
    const int seg_a = ts / S, seg_b = (te - 1) / S;
    const double wtot = a.part_freq != nullptr ? a.wtab[(int64_t)f * 3 * hop + 3 * hop - 1] : 0.0;
    for (int k = threadIdx.x; k < K; k += 64) {
        double sa = 0.0, sf = 0.0;
        for (int seg = seg_a; seg <= seg_b; ++seg) {
            const int sl = f - max(0, quotient_of(seg * S, e.inv_hop) - 1);
            const int64_t o = ((b * a.nseg + seg) * a.nslot + sl) * K + k;
            if (a.part_amp != nullptr) sa += a.part_amp[o];
            if (a.part_freq != nullptr) sf += a.part_freq[o];
        }
        if (g_amp != nullptr) g_amp[(b * e.frames + f) * K + k] = (frame_freq(e, freq_b, f, k) >= e.nyquist) ? 0.0f : (float)sa;
        if (g_freq != nullptr) {
            sf += wtot * sum_after(a.dcarry, b, a.nseg, seg_b, K, k, a.scanned);       // all later segments
            sf = sf / (double)a.sample_rate * (double)kTwoPi;                        // d omega / d f
            if (!e.harmonic) g_freq[(b * e.frames + f) * K + k] = (float)sf;
            else fsum[k] = sf;
        }
    }
    if (g_freq != nullptr && e.harmonic) {
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0.0;   // d / d f0 = sum_k (k + 1) d / d f_k, in ascending k
            for (int k = 0; k < K; ++k) s += (double)(k + 1) * fsum[k];
            g_freq[b * e.frames + f] = (float)s;
        }
    }
}

static int fill_env_args(int64_t batch, int frames, int sinusoids, int harmonic, int64_t samples, float sample_rate, EnvArgs* a)
{
    if (batch < 0 || frames < 1 || sinusoids < 1 || samples < 1 || !(sample_rate > 0.0f)) return SOT_ERR_BAD_SHAPE;
    if (frames >= samples || samples % frames != 0) return SOT_ERR_BAD_SHAPE;   // ddsp.py:155-170: upsampling only, whole hops
    if (sinusoids > kMaxSinusoids || samples > kMaxSamples || batch * (int64_t)frames > 0x7fffffffLL) return SOT_ERR_UNSUPPORTED_SIZE;
    a->batch = batch; a->frames = frames; a->K = sinusoids; a->harmonic = harmonic; a->samples = samples;
    a->hop = (int)(samples / frames); a->nyquist = sample_rate / 2.0f; a->scale = (float)frames / (float)samples;
    a->inv_hop = 1.0f / (float)a->hop; a->inv_K = 1.0f / (float)sinusoids;
    return SOT_OK;
}

}  // namespace sot_osc

extern "C" {

int sot_synth_envelopes_forward(const float* amp_frames, const float* freq_frames, const float* window, int64_t batch, int frames,
                                int sinusoids, int harmonic, int64_t samples, float sample_rate, float* amp_env, float* freq_env,
                                void* stream)
{
    using namespace sot_osc;
    EnvArgs a{};
    if (const int rc = fill_env_args(batch, frames, sinusoids, harmonic, samples, sample_rate, &a)) return rc;
    if (batch == 0) return SOT_OK;
    if (!amp_frames || !freq_frames || !window || !amp_env || !freq_env) return SOT_ERR_NULL_POINTER;
    a.amp = amp_frames; a.freq = freq_frames; a.window = window; a.amp_env = amp_env; a.freq_env = freq_env;
    const int64_t want = (samples * sinusoids + kThreads - 1) / kThreads;
    const int64_t cap = 256 * 16 / batch > 0 ? 256 * 16 / batch : 1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(synth_envelopes_forward_kernel, dim3((unsigned)(want < cap ? want : cap), (unsigned)(batch < 65535 ? batch : 65535)), dim3(kThreads), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

int sot_synth_envelopes_backward(const float* amp_frames, const float* freq_frames, const float* window, int64_t batch, int frames,
                                 int sinusoids, int harmonic, int64_t samples, float sample_rate, const float* grad_amp_env,
                                 const float* grad_freq_env, float* grad_amp_frames, float* grad_freq_frames, void* stream)
{
    using namespace sot_osc;
    EnvArgs a{};
    if (const int rc = fill_env_args(batch, frames, sinusoids, harmonic, samples, sample_rate, &a)) return rc;
    if (batch == 0) return SOT_OK;
    if (!amp_frames || !freq_frames || !window) return SOT_ERR_NULL_POINTER;
    if ((grad_amp_frames && !grad_amp_env) || (grad_freq_frames && !grad_freq_env)) return SOT_ERR_NULL_POINTER;
    a.amp = amp_frames; a.freq = freq_frames; a.window = window;
    a.g_amp_env = grad_amp_frames ? grad_amp_env : nullptr; a.g_freq_env = grad_freq_frames ? grad_freq_env : nullptr;
    a.g_amp = grad_amp_frames; a.g_freq = grad_freq_frames;
    const int subs = kThreads / sinusoids > 0 ? kThreads / sinusoids : 1;
    const size_t lds = 2 * sizeof(double) * (size_t)subs * sinusoids;
    (void)hipGetLastError();
    hipLaunchKernelGGL(synth_envelopes_backward_kernel, dim3((unsigned)(batch * frames)), dim3(kThreads), lds,
                       reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}


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
    a.scanned = a.nseg > kFusedScanSegments;
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
    a.scanned = a.nseg > kFusedScanSegments;
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
        if (a.nseg > 1 && a.scanned) {
            const unsigned sgrid = (unsigned)((batch * sinusoids + kThreads - 1) / kThreads);
            hipLaunchKernelGGL(oscillator_scan_kernel, dim3(sgrid), dim3(kThreads), 0, st, a.dcarry, batch, a.nseg, sinusoids, 1);
        }
        hipLaunchKernelGGL(oscillator_suffix_kernel, dim3(grid), dim3(kThreads), lds_bytes(a.seg_len, sinusoids, kTotals), st, a);
    }
    return launched() ? SOT_OK : SOT_ERR_LAUNCH;
}

// backward workspace: [segment start phases][segment dphi totals][wtab: frames x 3 hop doubles][partial sums x 2][atab: frames x 3 hop floats]
static size_t synth_partial_entries(int64_t batch, int frames, int64_t samples, int sinusoids)
{
    using namespace sot_osc;
    const int S = pick_segment(batch, samples, sinusoids);
    const int64_t nseg = (samples + S - 1) / S;
    return (size_t)batch * (size_t)nseg * (size_t)frame_slots(S, (int)(samples / frames)) * (size_t)sinusoids;
}

size_t sot_synth_workspace_bytes(int64_t batch, int frames, int64_t samples, int sinusoids, int backward)
{
    using namespace sot_osc;
    if (batch < 1 || frames < 1 || samples < 1 || sinusoids < 1 || samples > kMaxSamples || sinusoids > kMaxSinusoids) return 0;
    if (frames >= samples || samples % frames != 0) return 0;
    const size_t seg = 2 * segment_array_bytes(batch, samples, sinusoids);
    if (!backward) return seg;
    return seg + sizeof(double) * (3 * (size_t)samples + 2 * synth_partial_entries(batch, frames, samples, sinusoids)) + sizeof(float) * 3 * (size_t)samples;
}

static int fill_synth_args(const float* amp_frames, const float* freq_frames, const float* window, int64_t batch, int frames, int sinusoids,
                           int harmonic, int64_t samples, float sample_rate, sot_osc::OscArgs* a)
{
    using namespace sot_osc;
    if (const int rc = fill_env_args(batch, frames, sinusoids, harmonic, samples, sample_rate, &a->ctl)) return rc;
    a->ctl.amp = amp_frames; a->ctl.freq = freq_frames; a->ctl.window = window;
    a->batch = batch; a->samples = samples; a->sinusoids = sinusoids; a->sample_rate = sample_rate;
    a->seg_len = pick_segment(batch, samples, sinusoids);
    a->nseg = (samples + a->seg_len - 1) / a->seg_len;
    a->scanned = a->nseg > kFusedScanSegments;
    return batch * a->nseg > 0x7fffffffLL ? SOT_ERR_UNSUPPORTED_SIZE : SOT_OK;
}

int sot_synth_forward(const float* amp_frames, const float* freq_frames, const float* window, int64_t batch, int frames, int sinusoids,
                      int harmonic, int64_t samples, float sample_rate, float* audio, void* workspace, size_t workspace_bytes, void* stream)
{
    using namespace sot_osc;
    OscArgs a{};
    if (const int rc = fill_synth_args(amp_frames, freq_frames, window, batch, frames, sinusoids, harmonic, samples, sample_rate, &a)) return rc;
    if (batch == 0) return SOT_OK;
    if (!amp_frames || !freq_frames || !window || !audio) return SOT_ERR_NULL_POINTER;
    a.audio = audio;
    if (a.nseg > 1) {
        if (workspace == nullptr) return SOT_ERR_NULL_POINTER;
        if (workspace_bytes < segment_array_bytes(batch, samples, sinusoids)) return SOT_ERR_WORKSPACE;
        a.phase0 = static_cast<double*>(workspace);
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (!launch_segment_starts(a, st)) return SOT_ERR_LAUNCH;
    hipLaunchKernelGGL((oscillator_tile_kernel<kForward, true>), dim3((unsigned)(batch * a.nseg)), dim3(kThreads),
                       lds_bytes(a.seg_len, sinusoids, kForward), st, a);
    return launched() ? SOT_OK : SOT_ERR_LAUNCH;
}

size_t sot_synth_tap_table_bytes(int frames, int64_t samples)
{
    if (frames < 1 || samples < 1 || samples > sot_osc::kMaxSamples || frames >= samples || samples % frames != 0) return 0;
    return (sizeof(double) + sizeof(float)) * 3 * (size_t)samples;
}

int sot_synth_tap_tables(const float* window, int frames, int64_t samples, void* tables, void* stream)
{
    using namespace sot_osc;
    EnvArgs e{};
    if (const int rc = fill_env_args(1, frames, 1, 0, samples, 2.0f, &e)) return rc;
    if (!window || !tables) return SOT_ERR_NULL_POINTER;
    e.window = window;
    double* wtab = static_cast<double*>(tables);
    (void)hipGetLastError();
    hipLaunchKernelGGL(synth_tap_table_kernel, dim3((unsigned)frames), dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream), e, wtab,
                       reinterpret_cast<float*>(wtab + 3 * (size_t)samples));
    return launched() ? SOT_OK : SOT_ERR_LAUNCH;
}

int sot_synth_backward(const float* amp_frames, const float* freq_frames, const float* window, int64_t batch, int frames, int sinusoids,
                       int harmonic, int64_t samples, float sample_rate, const float* grad_audio, float* grad_amp_frames,
                       float* grad_freq_frames, const void* tap_tables, void* workspace, size_t workspace_bytes, int workspace_from_forward,
                       void* stream)
{
    using namespace sot_osc;
    OscArgs a{};
    if (const int rc = fill_synth_args(amp_frames, freq_frames, window, batch, frames, sinusoids, harmonic, samples, sample_rate, &a)) return rc;
    if (batch == 0 || (grad_amp_frames == nullptr && grad_freq_frames == nullptr)) return SOT_OK;
    if (!amp_frames || !freq_frames || !window || !grad_audio || !workspace) return SOT_ERR_NULL_POINTER;
    if (workspace_bytes < sot_synth_workspace_bytes(batch, frames, samples, sinusoids, 1)) return SOT_ERR_WORKSPACE;
    const size_t one = segment_array_bytes(batch, samples, sinusoids);
    const size_t entries = synth_partial_entries(batch, frames, samples, sinusoids);
    char* ws = static_cast<char*>(workspace);
    a.phase0 = reinterpret_cast<double*>(ws);
    a.dcarry = reinterpret_cast<double*>(ws + one);
    double* wtab = reinterpret_cast<double*>(ws + 2 * one);
    double* parts = wtab + 3 * (size_t)samples;
    float* atab = reinterpret_cast<float*>(parts + 2 * entries);
    if (tap_tables != nullptr) {   // the caller's tables (sot_synth_tap_tables for the same window, frames and samples)
        a.wtab = static_cast<const double*>(tap_tables);
        a.atab = reinterpret_cast<const float*>(a.wtab + 3 * (size_t)samples);
    } else {
        a.wtab = wtab;
        a.atab = atab;
    }
    a.part_amp = grad_amp_frames ? parts : nullptr;
    a.part_freq = grad_freq_frames ? parts + entries : nullptr;
    a.nslot = frame_slots(a.seg_len, a.ctl.hop);
    a.grad_audio = grad_audio;
    const size_t lds = lds_bytes_frames(a.seg_len, sinusoids, a.ctl.hop);
    if (lds > 160 * 1024) return SOT_ERR_UNSUPPORTED_SIZE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)(batch * a.nseg);
    (void)hipGetLastError();
    if (!workspace_from_forward && !launch_segment_starts(a, st)) return SOT_ERR_LAUNCH;
    if (tap_tables == nullptr) hipLaunchKernelGGL(synth_tap_table_kernel, dim3((unsigned)frames), dim3(kThreads), 0, st, a.ctl, wtab, atab);
    auto kern = oscillator_tile_kernel<kBackwardFrames, true>;
    if (lds > 64 * 1024) {   // opt this kernel in to the full LDS once per DEVICE (the attribute is per device; setting it twice is harmless)
        static std::mutex mu;
        static bool done[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); dev = -1; }
        std::lock_guard<std::mutex> lock(mu);
        if (dev < 0 || !done[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                (void)hipGetLastError();
            if (dev >= 0) done[dev] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, a);
    if (a.part_freq != nullptr && a.nseg > 1 && a.scanned) {
        const unsigned sgrid = (unsigned)((batch * sinusoids + kThreads - 1) / kThreads);
        hipLaunchKernelGGL(oscillator_scan_kernel, dim3(sgrid), dim3(kThreads), 0, st, a.dcarry, batch, a.nseg, sinusoids, 1);
    }
    hipLaunchKernelGGL(synth_frames_reduce_kernel, dim3((unsigned)(batch * frames)), dim3(64), sizeof(double) * (size_t)sinusoids, st, a,
                       grad_amp_frames, grad_freq_frames);
    return launched() ? SOT_OK : SOT_ERR_LAUNCH;
}

}  // extern "C"
