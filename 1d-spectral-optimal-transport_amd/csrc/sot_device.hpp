// sot_device.hpp -- device-side building blocks shared by the SOT kernels (gfx950, wave64).
//
// Everything here works on one "row group": G consecutive threads of a workgroup that own one
// spectrum pair (row) staged in LDS.  G is a multiple of 64, so a row group is a whole number of
// wavefronts and wave-level primitives never straddle two rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sot {

constexpr int kWave = 64;
constexpr float kMassEps = 1e-7f;  // utils.py:135-142 safe_divide epsilon (a float32 tensor)

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// ---------------------------------------------------------------------------------------------
// Row mass in ATen's CPU summation order (losses.py:177,184 call torch.sum(x, dim=1)).
//
// ATen's x86 kernel (SumKernel.cpp: cascade_sum -> vectorized_inner_sum -> row_sum ->
// multi_row_sum) views a contiguous fp32 row as 8-lane vectors with 4 ILP accumulators, i.e. 32
// interleaved columns (column of element e = e mod 32).  Each column is summed sequentially in
// chunks of 16 steps; chunk sums cascade through 4 levels.  The left-over vectors, the 4 ILP
// groups, the 8 lanes and the scalar tail are then folded in a fixed order.  Reproducing that order
// makes the row mass S -- and through it the knife-edge of the `limit_quantile_range` cutoff
//  -- bit-identical to the reference's CPU path.
//
// Work split: "task" (column c, chunk h) = sequential sum of <= 16 elements; 32 * ceil(steps/16)
// independent tasks per array, spread over all threads of the row group; then one half-wave (32
// lanes = 32 columns) per array folds the chunk sums.
// ---------------------------------------------------------------------------------------------
struct MassPlan {
    int n;       // row length
    int steps;   // n / 32   (multi_row_sum "size")
    int nchunk;  // ceil(steps / 16)
};

__device__ __forceinline__ MassPlan make_mass_plan(int n)
{
    MassPlan p;
    p.n = n;
    p.steps = (n >= 8) ? (n >> 5) : 0;
    p.nchunk = (p.steps + 15) >> 4;
    return p;
}

// Phase A: chunk sums.  raw: LDS row (n floats), part: LDS scratch [nchunk*32].
// SQ: the staged values are squared on the fly (square_dist, losses.py:172-174: x**2 == x*x exactly).
template <bool SQ>
__device__ __forceinline__ float ldw(const float* raw, int e) { const float w = raw[e]; return SQ ? w * w : w; }

template <int G, bool SQ>
__device__ __forceinline__ void mass_chunk_sums(const float* raw, float* part, const MassPlan& mp, int t)
{
    const int ntask = mp.nchunk << 5;
    for (int task = t; task < ntask; task += G) {
        const int c = task & 31, h = task >> 5;
        const int s0 = h << 4;
        const int s1 = min(s0 + 16, mp.steps);
        float acc = 0.0f;
        for (int s = s0; s < s1; ++s) acc += ldw<SQ>(raw, (s << 5) + c);
        part[task] = acc;
    }
}

// Phase B: executed by 32 consecutive lanes (c = 0..31) of one wave; `half_base` is the lane id of
// column 0 (0 or 32).  Returns S in every participating lane.
template <bool SQ>
__device__ __forceinline__ float mass_fold(const float* raw, const float* part, const MassPlan& mp, int c, int half_base)
{
    const int n = mp.n;
    if (n < 8) {  // scalar_inner_sum: 4 columns, then the tail into column 0
        float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
        int i = 0;
        if (n >= 4) { p0 += ldw<SQ>(raw, 0); p1 += ldw<SQ>(raw, 1); p2 += ldw<SQ>(raw, 2); p3 += ldw<SQ>(raw, 3); i = 4; }
        for (; i < n; ++i) p0 += ldw<SQ>(raw, i);
        return ((p0 + p1) + p2) + p3;
    }
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    const int nfull = mp.steps >> 4;
    int i = 0;
    for (int h = 0; h < nfull; ++h) {
        a0 = part[(h << 5) + c];
        i += 16;
        a1 += a0; a0 = 0.0f;
        if ((i & (15 << 4)) == 0) {
            a2 += a1; a1 = 0.0f;
            if ((i & (15 << 8)) == 0) { a3 += a2; a2 = 0.0f; }
        }
    }
    if (mp.steps & 15) a0 = part[(nfull << 5) + c];
    float col = ((a0 + a1) + a2) + a3;
    // left-over 8-lane vectors go to ILP group 0 (columns 0..7)
    const int vec_size = n >> 3;
    if (c < 8)
        for (int v = (mp.steps << 2); v < vec_size; ++v) col += ldw<SQ>(raw, (v << 3) + c);
    // fold the 4 ILP groups: p0[l] = ((col[l] + col[8+l]) + col[16+l]) + col[24+l]
    const int l = c & 7;
    float p0 = __shfl(col, half_base + l);
    p0 += __shfl(col, half_base + 8 + l);
    p0 += __shfl(col, half_base + 16 + l);
    p0 += __shfl(col, half_base + 24 + l);
    // scalar tail first, then the 8 lanes, sequentially
    float fin = 0.0f;
    for (int k = vec_size << 3; k < n; ++k) fin += ldw<SQ>(raw, k);
#pragma unroll
    for (int k = 0; k < 8; ++k) fin += __shfl(p0, half_base + k);
    return fin;
}

__device__ __forceinline__ float guard_mass(float s) { return (s <= kMassEps) ? kMassEps : s; }

// ---------------------------------------------------------------------------------------------
// Wave / group scans and reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_incl_scan(double v)
{
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const double o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;  // lane 0 holds the total
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// ---------------------------------------------------------------------------------------------
// Merge-path partition: number of U elements among the first D merged elements, with U before V
// on ties (the order of a stable sort of cat(U, V)).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int merge_path(const float* U, const float* V, int n, int m, int D)
{
    int lo = max(0, D - m), hi = min(D, n);
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (U[mid] <= V[D - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ float transport_cost(float xa, float yb, float p)
{
    const float d = fabsf(xa - yb);
    if (p == 1.0f) return d;          // losses.py:311-312: no pow for p == 1
    if (p == 2.0f) return d * d;      // torch.pow(., 2) is an exact square
    return powf(d, p);
}

// ---------------------------------------------------------------------------------------------
// In-LDS bitonic sort of (key, index) pairs, ascending by (key, index): a stable sort.
// npad = power of two >= n; entries >= n must be pre-filled with (+inf, INT_MAX).
// All T threads of the (sub)group call it; `sync()` must be a barrier over exactly those threads.
// ---------------------------------------------------------------------------------------------
template <typename Sync>
__device__ __forceinline__ void bitonic_sort_kv(float* key, int* idx, int npad, int t, int T, Sync sync)
{
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < npad; i += T) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const float ka = key[i], kb = key[ixj];
                    const int ia = idx[i], ib = idx[ixj];
                    const bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
                    const bool up = (i & k) == 0;
                    if (a_gt_b == up) { key[i] = kb; key[ixj] = ka; idx[i] = ib; idx[ixj] = ia; }
                }
            }
            sync();
        }
    }
}

}  // namespace sot
