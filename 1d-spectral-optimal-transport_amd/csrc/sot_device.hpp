// sot_device.hpp -- device-side building blocks shared by the SOT kernels (gfx950, wave64).
//
// Everything here works on one "row group": G consecutive threads of a workgroup that own one
// spectrum pair (row) staged in LDS.  G is a multiple of 64, so a row group is a whole number of
// wavefronts and wave-level primitives never straddle two rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>

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

#ifndef SOT_CHUNK_UNROLL
#define SOT_CHUNK_UNROLL 1
#endif
template <int G, bool SQ>
__device__ __forceinline__ void mass_chunk_sums(const float* raw, float* part, const MassPlan& mp, int t)
{
    const int ntask = mp.nchunk << 5;
    for (int task = t; task < ntask; task += G) {
        const int c = task & 31, h = task >> 5;
        const int s0 = h << 4;
        const int s1 = min(s0 + 16, mp.steps);
        float acc = 0.0f;
        if (SOT_CHUNK_UNROLL && s1 - s0 == 16) {  // full chunk: issue the 16 LDS reads back to back, then the ordered adds
            float v[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) v[s] = ldw<SQ>(raw, ((s0 + s) << 5) + c);
#pragma unroll
            for (int s = 0; s < 16; ++s) acc += v[s];
        } else {
            for (int s = s0; s < s1; ++s) acc += ldw<SQ>(raw, (s << 5) + c);
        }
        part[task] = acc;
    }
}

// Phase B: executed by 32 consecutive lanes (c = 0..31), one per column: cascade of the chunk sums
// plus the left-over 8-lane vectors (which ATen adds to ILP group 0).  Returns the column total.
// For n < 8 (ATen's scalar_inner_sum path) lane c == 0 returns the complete row sum instead.
#ifndef SOT_COLUMN_PREFETCH
#define SOT_COLUMN_PREFETCH 1   /* -0.3 us */
#endif
template <bool SQ>
__device__ __forceinline__ float mass_column(const float* raw, const float* part, const MassPlan& mp, int c)
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
#if SOT_COLUMN_PREFETCH
    // the first four chunk sums (all of them for n <= 2048) are fetched together: one LDS round trip instead of a
    // dependent load -> add chain inside the serial mass fold
    float pre[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) pre[h] = (h < nfull) ? part[(h << 5) + c] : 0.0f;
#endif
    int i = 0;
    for (int h = 0; h < nfull; ++h) {
#if SOT_COLUMN_PREFETCH
        a0 = (h < 4) ? pre[h & 3] : part[(h << 5) + c];
#else
        a0 = part[(h << 5) + c];
#endif
        i += 16;
        a1 += a0; a0 = 0.0f;
        if ((i & (15 << 4)) == 0) {
            a2 += a1; a1 = 0.0f;
            if ((i & (15 << 8)) == 0) { a3 += a2; a2 = 0.0f; }
        }
    }
    if (mp.steps & 15) a0 = part[(nfull << 5) + c];
    float col = ((a0 + a1) + a2) + a3;
    const int vec_size = n >> 3;
    if (c < 8)
        for (int v = (mp.steps << 2); v < vec_size; ++v) col += ldw<SQ>(raw, (v << 3) + c);
    return col;
}

// Phase C: every thread folds the 32 column totals itself (reads are LDS broadcasts), which removes
// a serial single-wave step and a barrier from the row pipeline:
//   p0[l] = ((col[l] + col[8+l]) + col[16+l]) + col[24+l];  S = (0 + tail...) + p0[0] + ... + p0[7].
template <bool SQ>
__device__ __forceinline__ float mass_fold(const float* raw, const float* colbuf, int n)
{
    if (n < 8) return colbuf[0];
    const float4* c4 = reinterpret_cast<const float4*>(colbuf);
    const float4 a0 = c4[0], a1 = c4[1], b0 = c4[2], b1 = c4[3], c0 = c4[4], c1 = c4[5], d0 = c4[6], d1 = c4[7];
    const float p0 = ((a0.x + b0.x) + c0.x) + d0.x, p1 = ((a0.y + b0.y) + c0.y) + d0.y;
    const float p2 = ((a0.z + b0.z) + c0.z) + d0.z, p3 = ((a0.w + b0.w) + c0.w) + d0.w;
    const float p4 = ((a1.x + b1.x) + c1.x) + d1.x, p5 = ((a1.y + b1.y) + c1.y) + d1.y;
    const float p6 = ((a1.z + b1.z) + c1.z) + d1.z, p7 = ((a1.w + b1.w) + c1.w) + d1.w;
    float fin = 0.0f;
    for (int k = (n >> 3) << 3; k < n; ++k) fin += ldw<SQ>(raw, k);
    fin += p0; fin += p1; fin += p2; fin += p3; fin += p4; fin += p5; fin += p6; fin += p7;
    return fin;
}

__device__ __forceinline__ float guard_mass(float s) { return (s <= kMassEps) ? kMassEps : s; }

// ---------------------------------------------------------------------------------------------
// Correctly rounded x / S for a per-row constant S (utils.py:141 `numerator / safe_denominator`).
// q0 = x*r with r = RN(1/S), e = x - q0*S (exact, one FMA), q1 = RN(q0 + e*r) is the IEEE quotient
// (Markstein's reciprocal-refinement theorem) as long as the residual cannot underflow and the
// quotient is normal; both hold when x >= 2^-78 and 2^-24 < S <= 2^40.  `risk` tracks the smallest
// non-zero operand seen; the caller redoes the chunk with the IEEE sequence when it is below the bound.
// 5 VALU per element instead of ~12 (v_div_scale x2, v_rcp, 4 FMA, v_div_fmas, v_div_fixup).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kFastDivMinBits = 0x18800000u;  // 2^-78
__device__ __forceinline__ float div_by_row_constant(float x, float S, float r, uint32_t& risk)
{
    const float q0 = x * r;
    const float e = fmaf(-q0, S, x);
    const float q1 = fmaf(e, r, q0);
    risk = min(risk, __float_as_uint(x) - 1u);  // x == 0 (exact either way) wraps to UINT_MAX
    return q1;
}

// ---------------------------------------------------------------------------------------------
// Wave-level scans / reductions.  Forward scans use DPP (row_shr 1/2/4/8, row_bcast 15/31,
// wave_shr 1): a handful of VALU ops instead of 6 dependent LDS-crossbar shuffles (12 for fp64).
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float v)  // value of the DPP source lane; +0.0 where there is none
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
// row_shr with all rows and banks enabled and bound_ctrl: lanes without a source read 0 and no lane keeps its old value,
// so the destination needs no zero-initialisation (2 VALU less per fp64 step than dpp_f64)
template <int CTRL>
__device__ __forceinline__ double dpp_f64_shr(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
constexpr int kRowShr1 = 0x111, kRowShr2 = 0x112, kRowShr4 = 0x114, kRowShr8 = 0x118;
constexpr int kRowBcast15 = 0x142, kRowBcast31 = 0x143, kWaveShr1 = 0x138;

__device__ __forceinline__ double wave_incl_scan(double v)
{
    v += dpp_f64_shr<kRowShr1>(v);
    v += dpp_f64_shr<kRowShr2>(v);
    v += dpp_f64_shr<kRowShr4>(v);
    v += dpp_f64_shr<kRowShr8>(v);
    v += dpp_f64<kRowBcast15, 0xA>(v);
    v += dpp_f64<kRowBcast31, 0xC>(v);
    return v;
}
__device__ __forceinline__ double wave_shift_right1(double v) { return dpp_f64<kWaveShr1>(v); }  // lane 0 gets 0
__device__ __forceinline__ double wave_last(double v)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ float wave_sum(float v)  // total in every lane
{
    v += dpp_f32<kRowShr1>(v);
    v += dpp_f32<kRowShr2>(v);
    v += dpp_f32<kRowShr4>(v);
    v += dpp_f32<kRowShr8>(v);
    v += dpp_f32<kRowBcast15, 0xA>(v);
    v += dpp_f32<kRowBcast31, 0xC>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ double wave_sum(double v) { return wave_last(wave_incl_scan(v)); }

// suffix (right-to-left) inclusive scan, used by the backward's reverse cumsum
__device__ __forceinline__ double wave_suffix_incl_scan(double v)
{
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const double o = __shfl_down(v, off);
        if (lane + off < kWave) v += o;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// Synchronisation of ONE row group.  A row that is owned by a single wavefront (64 threads: 129 / 257 / 512-bin rows) never
// exchanges data with the other rows of its workgroup, and the LDS executes one wave's instructions in issue order: a
// compiler-level ordering point is all it needs.  Rows of several wavefronts use the workgroup barrier (all row groups of a
// workgroup execute the same number of them).  SOT_WAVE_ROWS=0 restores workgroup barriers everywhere (A/B switch).
// ---------------------------------------------------------------------------------------------
#ifndef SOT_WAVE_ROWS
#define SOT_WAVE_ROWS 1
#endif
template <int NW>
__device__ __forceinline__ void row_sync()
{
    if constexpr (NW == 1 && SOT_WAVE_ROWS) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// "does any thread of the row group see `flag`" + the row group's synchronisation point
template <int NW>
__device__ __forceinline__ bool row_any(bool flag)
{
    if constexpr (NW == 1 && SOT_WAVE_ROWS) {
        const bool any = __builtin_amdgcn_ballot_w64(flag) != 0ull;
        row_sync<NW>();
        return any;
    } else {
        return __syncthreads_or(flag ? 1 : 0) != 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Batch mean inside the kernel that produces the row losses (losses.py:203-211): the workgroup that finishes LAST reduces
// row_loss[0, B) with the arithmetic of sot_reduce_mean_kernel, operation for operation (1024 "virtual threads" of 8
// consecutive rows each, DPP wave sums, 16 wave totals added in order), so the result is bit-identical to the separate
// kernel's and independent of timing.  Nobody waits: a workgroup only learns, from the value its counter add returns,
// whether it was the last one.  Hand-off protocol (cdna_hip_programming.md Guideline 16): row losses are written with
// write-through (sc1) stores; every wave drains them (s_waitcnt vmcnt(0)); workgroup barrier; ONE lane adds to the
// counter of its shard (blocks b and b + 8 tend to share an XCD: eight shard counters instead of one hot word), the
// shard's last arriver adds to the top counter, the overall last arriver runs an agent-scope acquire and the workgroup
// reads the row losses with sc1 loads.  The last arrivers put the counters back to zero: the caller zero-fills
// `counters` once and may reuse it for every later launch on the same stream (one launch in flight per counter buffer).
// ---------------------------------------------------------------------------------------------
struct MeanTail {
    unsigned* counters;   // [9] device words, zero between launches: 8 shard counters + the top counter; null = no tail
    double denom;
    float* mean_out;      // may be null
    double* sum_out;      // may be null
    int apply_hinge;
    float hinge;
};

__device__ __forceinline__ void store_row_loss(float* row_loss, int64_t row, float v, bool write_through)
{
    if (write_through)
        __hip_atomic_store(reinterpret_cast<unsigned*>(row_loss) + row, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        row_loss[row] = v;
}

__device__ __forceinline__ float load_row_loss_sc1(const float* row_loss, int64_t row)
{
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(row_loss) + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Called by every thread of the workgroup after its last row (all BLOCK threads reach it).  `scratch`: >= 17 doubles of LDS
// that nothing else uses any more.
template <int BLOCK>
__device__ __forceinline__ void batch_mean_tail(const MeanTail& mt, const float* row_loss, int64_t B, double* scratch)
{
    static_assert(BLOCK % kWave == 0, "whole wavefronts");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's row-loss stores have left the CU
    __syncthreads();
    int* const flag = reinterpret_cast<int*>(scratch + 16);
    if (threadIdx.x == 0) {
        const unsigned nshard = gridDim.x < 8u ? gridDim.x : 8u;
        const unsigned shard = blockIdx.x % nshard;
        const unsigned members = (gridDim.x - shard + nshard - 1u) / nshard;
        int last = 0;
        unsigned old = __hip_atomic_fetch_add(mt.counters + shard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == members - 1u) {
            __hip_atomic_store(mt.counters + shard, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __hip_atomic_fetch_add(mt.counters + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == nshard - 1u) {
                __hip_atomic_store(mt.counters + 8, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        *flag = last;
    }
    __syncthreads();
    if (*flag == 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    for (int vw = threadIdx.x >> 6; vw < 16; vw += BLOCK / kWave) {   // virtual wave vw of sot_reduce_mean_kernel's 16
        const int64_t v = (int64_t)vw * kWave + lane;                  // virtual thread
        double acc = 0.0;
        for (int64_t base = v * 8; base < B; base += 8192) {
            float w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = (base + k < B) ? load_row_loss_sc1(row_loss, base + k) : 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float u = w[k];
                if (mt.apply_hinge) u = (base + k < B) ? fmaxf(u - mt.hinge, 0.0f) : 0.0f;
                acc += (double)u;
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) scratch[vw] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < 16; ++w) tot += scratch[w];
        if (mt.sum_out) *mt.sum_out = tot;
        if (mt.mean_out) *mt.mean_out = (float)(tot / mt.denom);
    }
}

// ---------------------------------------------------------------------------------------------
// Merge-path partition: i0 = number of U elements among the first D merged elements, with U before V on ties (the
// order of a stable sort of cat(U, V)): P(i) := U[i] <= V[D-1-i] is true for i < i0 and false from i0 on.
// Branch-free search with a fixed (wave-uniform) number of rounds and the descending steps 2^k + 1, ..., 9, 5, 3, 2, 1
// from the lower end of the diagonal's range: a step is taken when it stays inside the range and the predicate holds
// at its last element; any step sequence with s_k <= 1 + (sum of the later steps) finds every count up to the sum of
// all steps.  Compared with a bisection loop: no divergent loop, 5 VALU per round instead of 12, and probes of lanes
// whose searches have diverged by a multiple of 32 elements do not pile up in one LDS bank (steps 2^k do).  Probes
// outside the range read neighbouring LDS and are masked by the range test.  (A 4-ary search with three probes per round
// was slower: the search is bound by LDS reads and bank conflicts, not by its number of dependent rounds.)
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr int merge_steps_top(int range)  // reach of (2^k + 1, ..., 5, 3, 2, 1) = 2^(k+1) + k + 1
{
    int k = 1;
    while ((2 << k) + k + 1 < range) ++k;
    return k;
}

// 32-bit LDS byte address of an LDS pointer, and a load from such an address (+ immediate): the merge phases keep their
// address arithmetic in single VGPRs in byte units (no shift and no re-basing add per access).
typedef __attribute__((address_space(3))) const float lds_cfloat;
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
__device__ __forceinline__ float lds_load(uint32_t addr) { return *reinterpret_cast<lds_cfloat*>((uintptr_t)addr); }

// The search on LDS byte addresses: ub1 = address of U[-1], vd1 = address of V[D] + ub1, topk = merge_steps_top(min(n, m));
// returns the address of U[i0 - 1].
__device__ __forceinline__ uint32_t merge_path_steps32(uint32_t ub1, uint32_t vd1, int n, int m, int D, int topk)
{
    const int lo = max(0, D - m), hi = min(D, n);
    uint32_t posb = ub1 + 4u * (uint32_t)lo;   // address of U[pos - 1]
    const uint32_t hib = ub1 + 4u * (uint32_t)hi;
    for (int k = topk; k >= 1; --k) {
        const uint32_t candb = posb + (4u << k) + 4u;   // step 2^k + 1
        const int take = (int)(candb <= hib) & (int)(lds_load(candb) <= lds_load(vd1 - candb));
        posb = take ? candb : posb;
    }
#pragma unroll
    for (int step = 2; step >= 1; --step) {
        const uint32_t candb = posb + 4u * (uint32_t)step;
        const int take = (int)(candb <= hib) & (int)(lds_load(candb) <= lds_load(vd1 - candb));
        posb = take ? candb : posb;
    }
    return posb;
}

// d^p for d >= 0, p >= 1 (losses.py:313 `diff_quantiles.pow(p)` for a general p) on the transcendental units: with d = m 2^e,
// m in [0.5, 1), d^p = exp2(p (e + log2 m)).  v_log_f32 / v_exp_f32 are good to about an ulp, but the exponent p (e + log2 m)
// reaches hundreds, so it is carried as an unevaluated sum hi + lo (Fast2Sum of e + log2 m, product error from one FMA) and the
// low part applied as exp2(hi) (1 + lo ln 2): 14 VALU instructions instead of libm powf's ~120 (which made a p = 3 call 4x slower
// than p = 2: 183 vs 49 us at 8192 x 2048), relative error <= ~5e-7 for p <= 3 (a few ulp; torch's CPU pow is a 1-ulp Sleef
// routine) -- inside the 1e-5 the tests hold general-p results to.
__device__ __forceinline__ float pow_nonneg(float d, float p)
{
    const float m = __builtin_amdgcn_frexp_mantf(d);
    const float ef = (float)__builtin_amdgcn_frexp_expf(d);
    const float l = __builtin_amdgcn_logf(m);          // log2 m in [-1, 0)
    const float s_hi = ef + l;
    const float s_lo = l - (s_hi - ef);                // exact: |ef| >= |l| or ef == 0
    const float hi = p * s_hi;
    const float lo = fmaf(p, s_hi, -hi) + p * s_lo;
    const float r = __builtin_amdgcn_exp2f(hi);
    const float v = fmaf(r, lo * 0.6931472f, r);
    return d == 0.0f ? 0.0f : v;                       // log2(0) = -inf would turn lo into a NaN
}

// PM: 1 -> p == 1 (losses.py:311-312: no pow), 2 -> p == 2 (torch.pow(., 2) is an exact square), 0 -> any other p >= 1
template <int PM>
__device__ __forceinline__ float transport_cost(float xa, float yb, float p)
{
    const float d = fabsf(xa - yb);
    if (PM == 1) return d;
    if (PM == 2) return d * d;
    return pow_nonneg(d, p);
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

// ---------------------------------------------------------------------------------------------
// In-LDS MERGE sort of (key, index) pairs (round 4): the same result as bitonic_sort_kv -- ascending keys, equal keys in ascending index order
// when idx starts as 0 .. n-1 with the pads (+inf, INT_MAX) behind: the sort is STABLE -- in O(N log N) instead of O(N log^2 N) and with
// 1 + 2 log2(npad / 8) barriers instead of log2(npad) (log2(npad) + 1) / 2 (2048 points: 17 instead of 66; ~1700 instead of ~13 000
// instructions per thread).  Phase 0: every aligned block of 8 elements is sorted in the registers of one thread by odd-even
// transposition (adjacent exchanges on strict "greater": stable).  Round r: runs of L = 8 * 2^(r-1) are merged pairwise; the thread that
// owns outputs [8 v, 8 v + 8) of the array finds its split of the pair by bisection on the merge path (left run first on ties: stable),
// merges eight elements into registers, and after a barrier writes them back IN PLACE (no second LDS buffer).
// npad: power of two >= 8.  All T threads call it; at most MAXV * T blocks of 8 (npad <= 8 MAXV T).  sync(): barrier over those threads.
// ---------------------------------------------------------------------------------------------
template <int MAXV, typename Sync>
__device__ __forceinline__ void merge_sort_kv(float* key, int* idx, int npad, int t, int T, Sync sync)
{
    const int NV = npad >> 3;
    // phase 0: blocks of 8 in registers
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int v = t + j * T;
        if (v < NV) {
            float k[8]; int x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { k[e] = key[8 * v + e]; x[e] = idx[8 * v + e]; }
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
#pragma unroll
                for (int e = pass & 1; e + 1 < 8; e += 2) {
                    const bool sw = k[e] > k[e + 1];
                    const float ka = sw ? k[e + 1] : k[e], kb = sw ? k[e] : k[e + 1];
                    const int xa = sw ? x[e + 1] : x[e], xb = sw ? x[e] : x[e + 1];
                    k[e] = ka; k[e + 1] = kb; x[e] = xa; x[e + 1] = xb;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { key[8 * v + e] = k[e]; idx[8 * v + e] = x[e]; }
        }
    }
    sync();
    int logL = 3;
    for (int L = 8; L < npad; L <<= 1, ++logL) {
        float ok[MAXV][8]; int oi[MAXV][8];
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int v = t + j * T;
            if (v < NV) {
                const int o = 8 * v;                        // first output position of this thread
                const int a0 = o & ~(2 * L - 1), b0 = a0 + L;  // the pair of runs it falls into
                const int d = o - a0;                       // outputs of the pair in front of it
                int lo = max(0, d - L), hi = min(d, L);
                for (int it = 0; it <= logL; ++it) {        // bisection: lo = number of left-run elements among the first d merged
                    if (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        const bool take = key[a0 + mid] <= key[b0 + d - 1 - mid];
                        lo = take ? mid + 1 : lo;
                        hi = take ? hi : mid;
                    }
                }
                int ia = lo, ib = d - lo;
                float ka = key[a0 + min(ia, L - 1)], kb = key[b0 + min(ib, L - 1)];
                int xa = idx[a0 + min(ia, L - 1)], xb = idx[b0 + min(ib, L - 1)];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool ta = (ia < L) && ((ib >= L) || (ka <= kb));
                    ok[j][e] = ta ? ka : kb;
                    oi[j][e] = ta ? xa : xb;
                    ia += ta ? 1 : 0;
                    ib += ta ? 0 : 1;
                    const int nx = ta ? a0 + min(ia, L - 1) : b0 + min(ib, L - 1);
                    const float kn = key[nx];
                    const int xn = idx[nx];
                    ka = ta ? kn : ka; xa = ta ? xn : xa;
                    kb = ta ? kb : kn; xb = ta ? xb : xn;
                }
            }
        }
        sync();   // every read of this round is done
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int v = t + j * T;
            if (v < NV) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { key[8 * v + e] = ok[j][e]; idx[8 * v + e] = oi[j][e]; }
            }
        }
        sync();
    }
}

// ---------------------------------------------------------------------------------------------
// Merge sort with SIXTEEN elements per thread on a skewed LDS image (round 4, second form).  Measured on the 8-per-thread form
// above (4096 x 2048 keys, rocprofv3 PMC): the LDS array busy 80 % of the time with 74 % of its cycles bank conflicts (blocks of 8
// dwords at a lane stride of 8 dwords: 8 lanes per bank), 2774 VALU instructions per wave and row of which the bisections are 30 %.
// This form:
//  * keys as ORDER-PRESERVING unsigned integers (float_order_bits), so that a run's end can carry a sentinel above every key;
//  * a block of 16 elements per thread, sorted in registers by Batcher's odd-even merge network (63 compare-exchanges on (key, index):
//    a total order, hence the stable result), then log2(npad / 16) merge rounds: half the bisections per element, one round less;
//  * the image is SKEWED: element i lives at i + (i >> 4), i.e. every block of 16 is followed by one spare slot and a thread's block
//    starts 17 dwords behind its neighbour's: block reads and writes touch 32 different banks;
//  * the spare slot behind a block holds the run's SENTINEL (0xFFFFFFFF) when the block ends a run, otherwise a COPY of the next
//    block's first element: the sequential merge reads "the element after the one just consumed", element g >= 1 of the same run, at
//    g + ((g - 1) >> 4) -- the real slot inside a block, the copy at a block boundary, the sentinel at the run's end -- with no
//    bounds test and no clamp (14 VALU per merged element instead of ~20);
//  * TWO arrays at once (a row's x and y positions): one barrier sequence for both.
// Entry: job.key holds the float keys in natural order on [0, npad) (+inf behind the n real ones); job.idx is scratch.  Exit: sorted
// float keys and their original indices (INT_MAX for the pads) in natural order.  Both arrays need sort16_capacity(npad) dwords.
// tests/test_merge_sort_model.py models this index algebra on the CPU.
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr int sort16_npad(int n) { int p = 16; while (p < n) p <<= 1; return p; }
__host__ __device__ constexpr int sort16_capacity(int npad) { return npad + (npad >> 4); }

__device__ __forceinline__ uint32_t float_order_bits(float f)
{
    uint32_t u = __float_as_uint(f);
    u = (u == 0x80000000u) ? 0u : u;   // -0.0 sorts with +0.0 (torch.sort compares values)
    return u ^ ((u & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float order_bits_float(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o); }

typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_ld_u32(uint32_t addr) { return *reinterpret_cast<const lds_u32*>((uintptr_t)addr); }
__device__ __forceinline__ void lds_st_u32(uint32_t addr, uint32_t v) { *reinterpret_cast<lds_u32*>((uintptr_t)addr) = v; }

struct SortJob { float* key; int* idx; int n; int npad; };   // npad = sort16_npad(n), or 0 for "no array"

// Batcher's odd-even merge sort for 16 inputs
__device__ constexpr signed char kNet16[63][2] = {
    {0, 1}, {2, 3}, {0, 2}, {1, 3}, {1, 2}, {4, 5}, {6, 7}, {4, 6}, {5, 7}, {5, 6}, {0, 4}, {2, 6}, {2, 4}, {1, 5}, {3, 7}, {3, 5},
    {1, 2}, {3, 4}, {5, 6}, {8, 9}, {10, 11}, {8, 10}, {9, 11}, {9, 10}, {12, 13}, {14, 15}, {12, 14}, {13, 15}, {13, 14}, {8, 12},
    {10, 14}, {10, 12}, {9, 13}, {11, 15}, {11, 13}, {9, 10}, {11, 12}, {13, 14}, {0, 8}, {4, 12}, {4, 8}, {2, 10}, {6, 14}, {6, 10},
    {2, 4}, {6, 8}, {10, 12}, {1, 9}, {5, 13}, {5, 9}, {3, 11}, {7, 15}, {7, 11}, {3, 5}, {7, 9}, {11, 13}, {1, 2}, {3, 4}, {5, 6},
    {7, 8}, {9, 10}, {11, 12}, {13, 14}};

template <int MAXB, typename Sync>
__device__ __forceinline__ void merge_sort16_kv2(const SortJob& jx, const SortJob& jy, int t, int T, Sync sync)
{
    const int NBx = jx.npad >> 4, NB = NBx + (jy.npad >> 4);
    const int maxpad = jx.npad > jy.npad ? jx.npad : jy.npad;
    uint32_t ok[MAXB][16]; int oi[MAXB][16];
    // ---- phase 0: blocks of 16 in registers
#pragma unroll
    for (int j = 0; j < MAXB; ++j) {
        const int vb = t + j * T;
        if (vb < NB) {
            const SortJob& job = vb < NBx ? jx : jy;
            const int b = vb < NBx ? vb : vb - NBx;
            const float4* src = reinterpret_cast<const float4*>(job.key + 16 * b);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = src[q];
                ok[j][4 * q] = float_order_bits(v.x); ok[j][4 * q + 1] = float_order_bits(v.y);
                ok[j][4 * q + 2] = float_order_bits(v.z); ok[j][4 * q + 3] = float_order_bits(v.w);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) oi[j][e] = (16 * b + e < job.n) ? 16 * b + e : INT_MAX;
#pragma unroll
            for (int c = 0; c < 63; ++c) {
                const int p = kNet16[c][0], q = kNet16[c][1];
                // (key, index) as one 64-bit integer: a single v_cmp_gt_u64 (and no short-circuit control flow: 63 nested ||/&& sent
                // the compiler into a path explosion)
                const bool sw = (((unsigned long long)ok[j][p] << 32) | (uint32_t)oi[j][p]) > (((unsigned long long)ok[j][q] << 32) | (uint32_t)oi[j][q]);
                const uint32_t k0 = sw ? ok[j][q] : ok[j][p], k1 = sw ? ok[j][p] : ok[j][q];
                const int x0 = sw ? oi[j][q] : oi[j][p], x1 = sw ? oi[j][p] : oi[j][q];
                ok[j][p] = k0; ok[j][q] = k1; oi[j][p] = x0; oi[j][q] = x1;
            }
        }
    }
    sync();   // every natural-order read is done: the skewed image may overwrite it
#pragma unroll
    for (int j = 0; j < MAXB; ++j) {
        const int vb = t + j * T;
        if (vb < NB) {
            const SortJob& job = vb < NBx ? jx : jy;
            const int b = vb < NBx ? vb : vb - NBx;
            if (job.npad == 16) {   // a single block: done; natural order, float keys
#pragma unroll
                for (int e = 0; e < 16; ++e) { job.key[e] = order_bits_float(ok[j][e]); job.idx[e] = oi[j][e]; }
            } else {
                const uint32_t ka = lds_addr(job.key) + 68u * (uint32_t)b, xa = lds_addr(job.idx) + 68u * (uint32_t)b;
#pragma unroll
                for (int e = 0; e < 16; ++e) { lds_st_u32(ka + 4u * e, ok[j][e]); lds_st_u32(xa + 4u * e, (uint32_t)oi[j][e]); }
                lds_st_u32(ka + 64u, 0xFFFFFFFFu);   // every block is a run of its own: the sentinel
            }
        }
    }
    sync();
    // ---- merge rounds: runs of L -> runs of 2 L
    int logL = 4;
    for (int L = 16; L < maxpad; L <<= 1, ++logL) {
#pragma unroll
        for (int j = 0; j < MAXB; ++j) {
            const int vb = t + j * T;
            const SortJob& job = vb < NBx ? jx : jy;
            if (vb < NB && L < job.npad) {
                const int b = vb < NBx ? vb : vb - NBx;
                const uint32_t kbase = lds_addr(job.key), xbase = lds_addr(job.idx);
                const int o = 16 * b;                          // first output of this block
                const int a0 = o & ~(2 * L - 1), b0 = a0 + L;  // the pair of runs it falls into
                const int d = o - a0;                          // outputs of the pair in front of it
                int lo = max(0, d - L), hi = min(d, L);
                for (int it = 0; it <= logL; ++it) {           // bisection on the merge path (left run first on ties: stable)
                    if (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        const int ga = a0 + mid, gb = b0 + d - 1 - mid;
                        const bool take = lds_ld_u32(kbase + 4u * (uint32_t)(ga + (ga >> 4))) <= lds_ld_u32(kbase + 4u * (uint32_t)(gb + (gb >> 4)));
                        lo = take ? mid + 1 : lo;
                        hi = take ? hi : mid;
                    }
                }
                int ia = a0 + lo;                              // the two heads (element numbers of the array): ia, and ib = ctot + e - ia
                const int ib = b0 + d - lo, ctot = ia + ib;
                // a head behind a consumed element of its run reads like every later one (copy / sentinel aware); the first element of a
                // run is read at its real slot (the slot in front of it is the PREVIOUS run's sentinel)
                const int pa = ia + ((ia - (lo > 0 ? 1 : 0)) >> 4), pb = ib + ((ib - (d - lo > 0 ? 1 : 0)) >> 4);
                uint32_t ka = lds_ld_u32(kbase + 4u * (uint32_t)pa), kb = lds_ld_u32(kbase + 4u * (uint32_t)pb);
                int xa = (int)lds_ld_u32(xbase + 4u * (uint32_t)pa), xb = (int)lds_ld_u32(xbase + 4u * (uint32_t)pb);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const bool ta = ka <= kb;                  // an exhausted run shows its sentinel and never wins
                    ok[j][e] = ta ? ka : kb;
                    oi[j][e] = ta ? xa : xb;
                    if (e < 15) {
                        const int gb = ctot + (e + 1) - ia;    // B's next element if B's head is the one consumed
                        ia += ta ? 1 : 0;
                        const int g = ta ? ia : gb;
                        const uint32_t ph = 4u * (uint32_t)(g + ((g - 1) >> 4));
                        const uint32_t kn = lds_ld_u32(kbase + ph);
                        const int xn = (int)lds_ld_u32(xbase + ph);
                        ka = ta ? kn : ka; xa = ta ? xn : xa;
                        kb = ta ? kb : kn; xb = ta ? xb : xn;
                    }
                }
            }
        }
        sync();   // every read of this round is done
#pragma unroll
        for (int j = 0; j < MAXB; ++j) {
            const int vb = t + j * T;
            const SortJob& job = vb < NBx ? jx : jy;
            if (vb < NB && L < job.npad) {
                const int b = vb < NBx ? vb : vb - NBx;
                if (2 * L == job.npad) {   // the last round of this array: natural order, float keys
                    float4* dk = reinterpret_cast<float4*>(job.key + 16 * b);
                    int4* dx = reinterpret_cast<int4*>(job.idx + 16 * b);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        dk[q] = make_float4(order_bits_float(ok[j][4 * q]), order_bits_float(ok[j][4 * q + 1]),
                                            order_bits_float(ok[j][4 * q + 2]), order_bits_float(ok[j][4 * q + 3]));
                        dx[q] = make_int4(oi[j][4 * q], oi[j][4 * q + 1], oi[j][4 * q + 2], oi[j][4 * q + 3]);
                    }
                } else {
                    const uint32_t ka = lds_addr(job.key) + 68u * (uint32_t)b, xa = lds_addr(job.idx) + 68u * (uint32_t)b;
#pragma unroll
                    for (int e = 0; e < 16; ++e) { lds_st_u32(ka + 4u * e, ok[j][e]); lds_st_u32(xa + 4u * e, (uint32_t)oi[j][e]); }
                    const int o = 16 * b;
                    if ((o & (2 * L - 1)) != 0) {              // not the first block of its new run: the copy for the block in front
                        lds_st_u32(ka - 4u, ok[j][0]); lds_st_u32(xa - 4u, (uint32_t)oi[j][0]);
                    }
                    if (((o + 16) & (2 * L - 1)) == 0) lds_st_u32(ka + 64u, 0xFFFFFFFFu);   // the last one: the sentinel
                }
            }
        }
        sync();
    }
}

#ifndef SOT_MERGE_SORT
#define SOT_MERGE_SORT 1   /* 0: the bitonic network of rounds 1-3 everywhere */
#endif
// the sort the kernels call: merge sort where its preconditions hold (npad >= 8, npad <= 8 MAXV T), the bitonic network otherwise
template <int MAXV, typename Sync>
__device__ __forceinline__ void sort_kv(float* key, int* idx, int npad, int t, int T, Sync sync)
{
    if (SOT_MERGE_SORT && npad >= 8 && npad <= 8 * MAXV * T) merge_sort_kv<MAXV>(key, idx, npad, t, T, sync);
    else bitonic_sort_kv(key, idx, npad, t, T, sync);
}

}  // namespace sot
