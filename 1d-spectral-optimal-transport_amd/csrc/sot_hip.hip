// sot_hip.hip -- MI355X (gfx950) kernels and C ABI for the 1-D spectral optimal-transport loss.
//
// Reference semantics: losses.py:129-313 (Wasserstein1D.forward, wasserstein_1d,
// quantile_function) and utils.py:135-142 (safe_divide) of
// bernardo-torres/1d-spectral-optimal-transport; the reference composes ~25 ATen ops
// (SURVEY.md table 2.2), this file replaces them with one fused, LDS-resident pipeline per row.
//
// Data layout: x [B,n], y [B,m] fp32 row-major in HBM, read exactly once with 16-byte-per-lane
// coalesced loads; one row pair lives in LDS as four arrays U|V|PX|PY (CDFs and support
// positions, each with one sentinel slot) from staging to the final reduction; HBM traffic per
// row is 4(n+m) bytes in and 4 bytes out (the algorithmic minimum of SURVEY §8d).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <mutex>
#include <type_traits>

#include "../../include/sot_hip.h"
#include "sot_device.hpp"
#include "sot_wave_sort.hpp"

// The file can be compiled whole (default) or in parts that are linked into one library, so that the many
// kernel instantiations build in parallel (build.py): bit 0 forward/shared positions, bit 1 forward/per-row
// positions, bit 2 backward/shared, bit 3 backward/per-row, bit 4 everything else (small kernels, host glue, C ABI),
// bit 5 the CSR (ragged) forward, bit 6 the cutoff (limit_quantile_range) family of forward/shared positions (bit 0 then
// holds the no-cutoff family), bit 7 the compile-time-length forward kernels, bit 8 the compile-time-length backward kernels,
// bits 9 / 10 their run-time-length forms, bit 11 the position-gradient kernel.
#ifndef SOT_PART
#define SOT_PART 4095
#endif
// Timing-only ablation builds (tools/ablate.py; results are WRONG on purpose): bit 0 no merge walk, bit 1 no partition
// search, bit 2 no row mass, bit 3 no division, bit 4 no CDF scan.  Never defined in the product build.
#ifndef SOT_WAVE_SORT
#define SOT_WAVE_SORT 1   /* 0: the in-LDS merge sort of round 4 everywhere (A/B switch) */
#endif
#ifndef SOT_ABLATE
#define SOT_ABLATE 0
#endif
// Waves inside the partition search + merge walk (dependent LDS chains) run at raised priority so that they win issue
// slots over co-resident waves in throughput phases: measured -2.7 % kernel time (interleaved A/B, steady state).
#ifndef SOT_WALK_PRIO
#define SOT_WALK_PRIO 1
#endif
#ifndef SOT_MASS_PRIO
#define SOT_MASS_PRIO 2
#endif
// tuning knobs of the A/B harness (tools/ab_probe.py); the defaults are the measured best
#ifndef SOT_E_MODE
#define SOT_E_MODE 0   /* 0: ceil(K/G) forced odd (58.45 us with walk unroll 2); 1: plain (59.2); 2: even with odd half */
#endif
#ifndef SOT_WALK_UNROLL
#define SOT_WALK_UNROLL 2  /* 0: compiler's choice (x4): 59.45 us; 1: 59.95; 2: 58.85 */
#endif

namespace sot {

// ---------------------------------------------------------------------------------------------
// LDS layout of one row group (identical arithmetic on host and device)
// ---------------------------------------------------------------------------------------------
struct RowLayout {
    int padcap;          // spare floats in front of U and of PX (front padding of the merge walk)
    int nU, nV;          // floats reserved for U (padcap + n+1 incl. sentinel) and V (m+1), multiples of 4
    int poff;            // PX = U + poff, PY = V + poff
    int part_x, part_y;  // chunk-sum scratch of the two row masses
    int colbuf;          // 2 x 32 column totals of the row masses (16-B aligned)
    int wtot;            // 2 * NW doubles (float offset, 8-B aligned)
    int red;             // NW floats + 2 floats (S_x, S_y)
    int grad;            // backward only: offset of the gradient arrays from U / V (regions mirror [U|V])
    int row_floats;      // total, multiple of 4
};

__host__ __device__ constexpr int align4(int v) { return (v + 3) & ~3; }
__host__ __device__ constexpr int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
__host__ __device__ constexpr int imax(int a, int b) { return a > b ? a : b; }

// merged elements handled by one thread: ceil(K / G) forced odd, so that the per-lane LDS address stride of
// the merge walk (~E/2 floats) is not a multiple of the 32-bank period on regular data
__host__ __device__ constexpr int merge_steps(int K, int G)
{
    const int e = (K + G - 1) / G;
    if (SOT_E_MODE == 1) return e;
    if (SOT_E_MODE == 2) return ((e + 1) & ~3) + 2;  // E/2 odd: 2, 6, 10, 14, 18, ...
    return e | 1;
}

__host__ __device__ constexpr RowLayout make_layout(int n, int m, int G, bool rowpos, bool with_grad = false)
{
    RowLayout L{};
    // The U and PX regions start with `padcap` spare floats: the forward walk prepends pad < E zero-valued
    // levels to U (zero width => zero contribution) so that every thread walks exactly E merged elements.
    L.padcap = align4(merge_steps(n + m, G));
    L.nU = L.padcap + align4(n + 1);
    L.nV = align4(m + 1);
    if (rowpos) {  // the per-row position sort works on power-of-two arrays in a skewed image (sot_device.hpp: sort16_capacity)
        L.nU = imax(L.nU, L.padcap + align4(sort16_capacity(sort16_npad(n))));
        L.nV = imax(L.nV, align4(sort16_capacity(sort16_npad(m))));
    }
    L.poff = L.nU + L.nV;
    const int nchx = (((n >= 8) ? (n >> 5) : 0) + 15) >> 4;
    const int nchy = (((m >= 8) ? (m >> 5) : 0) + 15) >> 4;
    L.part_x = 2 * L.poff;
    L.part_y = L.part_x + 32 * nchx;
    L.colbuf = align4(L.part_y + 32 * nchy);
    L.wtot = L.colbuf + 64;
    const int NW = G / kWave;
    L.red = L.wtot + 4 * NW;  // 2 arrays * NW doubles = 4*NW floats
    L.grad = align4(L.red + NW + 4);  // red[NW] | S_x, S_y | 1/S_x^, 1/S_y^ (guarded masses' reciprocals)
    L.row_floats = with_grad ? L.grad + L.nU + L.nV : L.grad;  // GU = U + grad, GV = V + grad
    return L;
}

// Diagnostic build only (-DSOT_STAMPS -> libsot_hip_stamps.so): workgroup 0 stamps the shader clock at the
// phase boundaries of its second row into a buffer of its own; no output value depends on a stamp.
#ifdef SOT_STAMPS
__device__ unsigned long long g_stamps[64];
#define SOT_STAMP(i) do { if (stamp_on) { __builtin_amdgcn_sched_barrier(0); g_stamps[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define SOT_STAMP(i) do { } while (0)
#endif

struct FwdArgs {
    const float* x; const float* y;
    const float* xpos; const float* ypos;   // sorted positions when !ROWPOS
    const int* xperm; const int* yperm;     // shared-position sort permutations (may be null)
    const int* ident;                       // [2] device flags: permutation is the identity (may be null)
    uint16_t* perm_out; const uint16_t* perm_in;   // per-row positions: [B, n + m] sort permutations written / reused (may be null)
    int64_t B; int n, m;
    int64_t xs, ys, xps, yps;               // row strides (elements)
    float p; uint32_t flags;
    float* row_loss;
    // optional outputs of the quantile variant
    float* oUq; float* oVq; float* oQ; float* oU; float* oV;
    // CSR (ragged) input form: row r owns entries [off[r], off[r+1]) of the concatenated weights/positions;
    // n, m above are then the MAXIMUM row lengths (LDS is sized for them)
    const int64_t* xoff; const int64_t* yoff;
    // batch mean by the last workgroup to finish (sot_device.hpp: batch_mean_tail); mt.counters == nullptr: not requested
    MeanTail mt;
};

// ---------------------------------------------------------------------------------------------
// Per-row-group context shared by the forward and backward kernels
// ---------------------------------------------------------------------------------------------
template <int G>
struct RowCtx {
    static constexpr int NW = G / kWave;
    float* base; float* U; float* V; float* PX; float* PY;
    float* partx; float* party; float* colbuf; double* wtot; float* red;
    float* GU; float* GV;  // backward only
    RowLayout L;
    MassPlan mpx, mpy;
    int n, m, K, E, Ga, pad, topk;
    int t, lane, wv;
    bool sq, dn, lim, do_sort, prenorm, x_ident, y_ident;
    float p;
};

template <int G, bool ROWPOS>
__device__ __forceinline__ RowCtx<G> make_ctx(const FwdArgs& a, float* smem, bool with_grad_arrays)
{
    RowCtx<G> c;
    const int tid = threadIdx.x;
    const int rg = tid / G;
    c.t = tid - rg * G;
    c.lane = tid & (kWave - 1);
    c.wv = c.t >> 6;
    c.n = a.n; c.m = a.m;
    c.L = make_layout(a.n, a.m, G, ROWPOS, with_grad_arrays);
    c.base = smem + rg * c.L.row_floats;
    c.U = c.base + c.L.padcap; c.V = c.base + c.L.nU; c.PX = c.U + c.L.poff; c.PY = c.V + c.L.poff;
    c.partx = c.base + c.L.part_x; c.party = c.base + c.L.part_y;
    c.colbuf = c.base + c.L.colbuf;
    c.wtot = reinterpret_cast<double*>(c.base + c.L.wtot);
    c.red = c.base + c.L.red;
    c.GU = c.U + c.L.grad; c.GV = c.V + c.L.grad;  // same offset from U and from V
    c.prenorm = a.flags & SOT_FLAG_PRENORMALIZED;
    c.sq = !c.prenorm && (a.flags & SOT_FLAG_SQUARE);
    c.dn = c.prenorm || (a.flags & SOT_FLAG_DONT_NORMALIZE);
    c.lim = a.flags & SOT_FLAG_LIMIT_Q;
    c.do_sort = a.flags & SOT_FLAG_REQUIRE_SORT;
    c.p = a.p;
    c.mpx = make_mass_plan(a.n); c.mpy = make_mass_plan(a.m);
    c.K = a.n + a.m;
    c.E = merge_steps(c.K, G);
    c.Ga = (c.K + c.E - 1) / c.E;       // threads that take part in the walk
    c.pad = c.Ga * c.E - c.K;           // zero-valued levels prepended to U: 0 <= pad < E <= padcap
    c.topk = merge_steps_top(min(c.n + c.pad, c.m));
    for (int e = c.t; e < c.pad; e += G) c.U[e - c.pad] = 0.0f;
    c.x_ident = true; c.y_ident = true;
    if (!ROWPOS) {
        if (a.ident != nullptr) { c.x_ident = a.ident[0] != 0; c.y_ident = a.ident[1] != 0; }
        // issue every position load before the first LDS store (independent loads: one memory round trip, not one per
        // element), in batches of 8 per array so that any row length is covered
        for (int e0 = 0; e0 < max(c.n, c.m); e0 += 8 * G) {
            float px[8], py[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + c.t + k * G;
                px[k] = (e < c.n) ? a.xpos[e] : 0.0f;
                py[k] = (e < c.m) ? a.ypos[e] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + c.t + k * G;
                if (e < c.n) c.PX[e] = px[k];
                if (e < c.m) c.PY[e] = py[k];
            }
        }
        for (int e = c.t; e < c.pad; e += G) c.PX[e - c.pad] = a.xpos[0];  // any finite value: the width is 0
        if (c.t == 0) {
            c.PX[c.n] = a.xpos[c.n - 1];  // clamp of losses.py:220: ranks beyond the last index reuse it
            c.PY[c.m] = a.ypos[c.m - 1];
            c.U[c.n] = INFINITY;          // sentinels: an exhausted side never wins the merge
            c.V[c.m] = INFINITY;
        }
    }
    return c;
}

// per-row support sizes (CSR form): everything derived from n, m is recomputed; the LDS layout stays that of
// the maximum lengths.  E (and with it pad < E) can only shrink, so the front padding still fits.
template <int G>
__device__ __forceinline__ void set_row_lengths(RowCtx<G>& c, int n, int m)
{
    c.n = n; c.m = m;
    c.mpx = make_mass_plan(n); c.mpy = make_mass_plan(m);
    c.K = n + m;
    c.E = merge_steps(c.K, G);
    c.Ga = (c.K + c.E - 1) / c.E;
    c.pad = c.Ga * c.E - c.K;
    c.topk = merge_steps_top(min(n + c.pad, m));
}

// ---- global -> register -> LDS staging of one row (VEC: 16 B per lane, rows 16-B aligned) ------------
template <int G, int CPT, bool VEC>
__device__ __forceinline__ void load_row(const float* __restrict__ src, int len, int t, float (&r)[CPT])
{
    if (VEC) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
        for (int k = 0; k < CPT / 4; ++k) {
            const int q = t + k * G;
            if (4 * q < len) {
                const float4 v = s4[q];
                r[4 * k] = v.x; r[4 * k + 1] = v.y; r[4 * k + 2] = v.z; r[4 * k + 3] = v.w;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int e = t + k * G;
            if (e < len) r[k] = src[e];
        }
    }
}

template <int G, int CPT, bool VEC>
__device__ __forceinline__ void store_row(float* dst, int len, int t, const float (&r)[CPT])
{
    if (VEC) {
        float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
        for (int k = 0; k < CPT / 4; ++k) {
            const int q = t + k * G;
            if (4 * q < len) d4[q] = make_float4(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int e = t + k * G;
            if (e < len) dst[e] = r[k];
        }
    }
}

// ---- P0 (ROWPOS): this row's positions -> LDS, sorted with an index payload when REQUIRE_SORT ---------
// On return ix/iy hold, for the CPT contiguous elements this thread owns in SORTED order, their
// original column.  Ends with a barrier (U/V may be overwritten by the weights afterwards).
// MERGE: the stable merge sort of sot_device.hpp (round 4; its two 16-register output arrays cost the register-capped CSR kernel
// occupancy -- 26.5 -> 34.7 us on config 4's rows, which never need the sort -- so that instantiation keeps the bitonic network)
// perm_in (round 5): this row's two sort permutations from an earlier call on the same positions ([n + m] uint16): the sorted supports are
// GATHERED through them, nothing is sorted (the backward and position-gradient kernels of a training step).  perm_out: where this call
// leaves them.
// (An instantiation WITHOUT the sort code, launched when perm_in is given, was measured: the position-gradient kernel drops from 233 to 200
// registers -- still two waves per SIMD -- and both kernels get slower, 143.9 -> 151.5 us and 166.1 -> 173.2 us at 4096 x 2048: not kept.)
template <int G, int CPT, bool MERGE = true>
__device__ __forceinline__ void rowpos_prepare(const RowCtx<G>& c, const float* xp, const float* yp, int nmax, int mmax,
                                               int (&ix)[CPT], int (&iy)[CPT], const uint16_t* perm_in = nullptr, uint16_t* perm_out = nullptr)
{
    const int n = c.n, m = c.m, t = c.t;
    const bool gather = perm_in != nullptr && c.do_sort;   // (an image is always complete: the pre-sort kernel and the sorting row kernels both write every row)
    if (gather) {            // uniform over the threads that share barriers
        // The row's positions arrive COALESCED (element t + k G per thread), are staged in natural order in the U / V regions (free until the
        // weights arrive) and gathered from LDS through the permutation: 16 scattered 4-byte loads per thread from global memory -- up to 64
        // cache lines per wave instruction -- cost the gathering forward 83 us where rows that arrive sorted take 71 (4096 x 2048, round 6).
        float vx[CPT], vy[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int e = t + k * G;
            vx[k] = (e < n) ? xp[e] : 0.0f;
            vy[k] = (e < m) ? yp[e] : 0.0f;
        }
        // this thread's CPT consecutive entries of each permutation: one 16-byte load where eight 16-bit entries are 16-byte aligned
        const bool vec = CPT == 8 && ((n | m) & 7) == 0 && (reinterpret_cast<uintptr_t>(perm_in) & 15) == 0;
        if (vec) {
            uint4 qx = make_uint4(0, 0, 0, 0), qy = make_uint4(0, 0, 0, 0);
            if (t * CPT < n) qx = *reinterpret_cast<const uint4*>(perm_in + t * CPT);
            if (t * CPT < m) qy = *reinterpret_cast<const uint4*>(perm_in + n + t * CPT);
            const uint32_t wx[4] = {qx.x, qx.y, qx.z, qx.w}, wy[4] = {qy.x, qy.y, qy.z, qy.w};
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int e = t * CPT + k;
                ix[k] = (e < n) ? (int)((wx[(k >> 1) & 3] >> (16 * (k & 1))) & 0xFFFFu) : e;
                iy[k] = (e < m) ? (int)((wy[(k >> 1) & 3] >> (16 * (k & 1))) & 0xFFFFu) : e;
            }
        } else {
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int e = t * CPT + k;
                ix[k] = (e < n) ? (int)perm_in[e] : e;
                iy[k] = (e < m) ? (int)perm_in[n + e] : e;
            }
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int e = t + k * G;
            if (e < n) c.U[e] = vx[k];
            if (e < m) c.V[e] = vy[k];
        }
        row_sync<G / kWave>();
        float gx[CPT], gy[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int e = t * CPT + k;
            gx[k] = (e < n) ? c.U[min(ix[k], n - 1)] : 0.0f;     // (clamped: a stale or foreign image must not become a wild LDS address)
            gy[k] = (e < m) ? c.V[min(iy[k], m - 1)] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int e = t * CPT + k;
            if (e < n) c.PX[e] = gx[k];
            if (e < m) c.PY[e] = gy[k];
        }
        row_sync<G / kWave>();
        if (t == 0) { c.PX[n] = c.PX[n - 1]; c.PY[m] = c.PY[m - 1]; }
        for (int e = t; e < c.pad; e += G) { c.PX[e - c.pad] = c.PX[0]; c.U[e - c.pad] = 0.0f; }
        return;
    }
    int* const IX = reinterpret_cast<int*>(c.U);  // index payloads alias U/V until the weights arrive
    int* const IY = reinterpret_cast<int*>(c.V);
    // barriers are workgroup-wide, so the sort network is sized by the maximum lengths (identical for every
    // row group of the workgroup); rows whose positions are already sorted skip it altogether
    const int npx = MERGE ? sort16_npad(nmax) : next_pow2(nmax), npy = MERGE ? sort16_npad(mmax) : next_pow2(mmax);
    int unsorted = 0;
    // The thread's CPT slots (elements t + k G) are fetched with a compile-time trip count: all loads of a row are in flight together
    // (a runtime loop waits for each element before it requests the next: sixteen L2 round trips per row at 8 elements per thread).
    float vx[CPT], vy[CPT], wx1[CPT], wy1[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int e = t + k * G;
        vx[k] = (e < n) ? xp[e] : INFINITY;
        vy[k] = (e < m) ? yp[e] : INFINITY;
        wx1[k] = (c.do_sort && e + 1 < n) ? xp[e + 1] : INFINITY;   // right neighbours for the sortedness test
        wy1[k] = (c.do_sort && e + 1 < m) ? yp[e + 1] : INFINITY;
    }
    if (c.do_sort) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int e = t + k * G;
            if (e < npx) { c.PX[e] = vx[k]; IX[e] = (e < n) ? e : INT_MAX; }
            if (e < npy) { c.PY[e] = vy[k]; IY[e] = (e < m) ? e : INT_MAX; }
            unsorted |= (vx[k] > wx1[k]) | (vy[k] > wy1[k]);        // +inf on the right of the last element: never "unsorted"
        }
        for (int e = t + CPT * G; e < npx; e += G) { c.PX[e] = INFINITY; IX[e] = INT_MAX; }   // padding of the sort network past the slots
        for (int e = t + CPT * G; e < npy; e += G) { c.PY[e] = INFINITY; IY[e] = INT_MAX; }
    } else {
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int e = t + k * G;
            if (e < n) c.PX[e] = vx[k];
            if (e < m) c.PY[e] = vy[k];
        }
    }
    const bool need_sort = row_any<G / kWave>(unsorted != 0);  // also the barrier after the loads
    if (need_sort) {
        if constexpr (MERGE) {
            // both arrays in one barrier sequence, 16 elements per thread (sot_device.hpp: merge_sort16_kv2): npx + npy <= 2 next_pow2(G CPT), i.e.
            // one block of 16 per thread for the 8-element geometries, two for 12 / 16 elements per thread (a second, never used block
            // costs the 256 x 8 kernel 32 registers)
            constexpr int MAXB = (CPT > 8) ? 2 : 1;
            const SortJob jx{c.PX, IX, n, npx}, jy{c.PY, IY, m, npy};
            merge_sort16_kv2<MAXB>(jx, jy, t, G, [] { row_sync<G / kWave>(); });
        } else {
            bitonic_sort_kv(c.PX, IX, npx, t, G, [] { row_sync<G / kWave>(); });
            bitonic_sort_kv(c.PY, IY, npy, t, G, [] { row_sync<G / kWave>(); });
        }
    }
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int e = t * CPT + k;
        // (clamped: a NaN position orders above the +inf pads, so a pad's index INT_MAX can surface among the first n outputs -- ADVICE r4;
        // NaN positions have no defined order here or in the reference's loss, but they must not become wild gather / store indices)
        ix[k] = (need_sort && e < n) ? min(IX[e], n - 1) : e;
        iy[k] = (need_sort && e < m) ? min(IY[e], m - 1) : e;
    }
    if (perm_out != nullptr && c.do_sort) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int e = t * CPT + k;
            if (e < n) perm_out[e] = (uint16_t)ix[k];
            if (e < m) perm_out[n + e] = (uint16_t)iy[k];
        }
    }
    row_sync<G / kWave>();
    if (t == 0) { c.PX[n] = c.PX[n - 1]; c.PY[m] = c.PY[m - 1]; }
    for (int e = t; e < c.pad; e += G) { c.PX[e - c.pad] = c.PX[0]; c.U[e - c.pad] = 0.0f; }  // pads sit at the first position
}

// ---- P2-P3: row masses, safe_divide, weight gather, fp64-accumulated CDFs -----------------------------
// Entry: U/V hold the raw weights in ORIGINAL column order (a barrier has been passed since they were
// written).  Exit (after its final barrier): U/V hold the CDFs.  Each thread owns the CPT contiguous
// elements [t*CPT, t*CPT + CPT) of each array in sorted order; wx/wy receive their ORIGINAL (unsquared)
// weights (only consumed by the backward kernel).
// SQM: square_dist known at compile time (0 = no, 1 = yes) or read from the flags (2)
template <int G, int CPT, bool ROWPOS, int SQM = 2>
__device__ __forceinline__ void build_cdfs(const FwdArgs& a, const RowCtx<G>& c, const int (&ix)[CPT], const int (&iy)[CPT],
                                           float (&wx)[CPT], float (&wy)[CPT], float& Sx_out, float& Sy_out,
                                           const bool stamp_on = false)
{
    (void)stamp_on;
    constexpr int NW = G / kWave;
    const int n = c.n, m = c.m, t = c.t;
    float* const U = c.U; float* const V = c.V;
    const bool sq = (SQM == 2) ? c.sq : (SQM == 1);
    const int e0 = t * CPT;

    // ---- P2: row masses in ATen order (losses.py:177,184; the reference sums BEFORE it sorts, so the
    //      staged row is still in its original column order here) -----------------------------------
    float Sx = 1.0f, Sy = 1.0f;  // prenormalised: w / 1.0f == w exactly, weights enter the CDF unchanged
    float rSx = 1.0f, rSy = 1.0f;  // reciprocals of the guarded masses (computed once per row by the fold waves)
    if (!(SOT_ABLATE & 4) && !c.prenorm) {
        if (sq) {
            mass_chunk_sums<G, true>(U, c.partx, c.mpx, t);
            if (!c.dn) mass_chunk_sums<G, true>(V, c.party, c.mpy, (t + G / 2) & (G - 1));
        } else {
            mass_chunk_sums<G, false>(U, c.partx, c.mpx, t);
            if (!c.dn) mass_chunk_sums<G, false>(V, c.party, c.mpy, (t + G / 2) & (G - 1));
        }
        row_sync<G / kWave>();
        SOT_STAMP(2);
        // columns + fold by ONE wave per array (wave 0: x, wave 1: y; a single-wave row group does both
        // in its two half-waves): the 32 column totals are exchanged through this wave's own LDS slots.
        // (Doing this redundantly in every wave to save the barrier below was measured 10 % SLOWER.)
        float* const Sv = c.red + NW;
        __builtin_amdgcn_s_setprio(SOT_MASS_PRIO);
        if (NW >= 2) {
            if (c.wv < 2 && !(c.wv == 1 && c.dn)) {
                const float* raw = c.wv ? V : U;
                const float* part = c.wv ? c.party : c.partx;
                const MassPlan& mp = c.wv ? c.mpy : c.mpx;
                float* cb = c.colbuf + 32 * c.wv;
                if (c.lane < 32) cb[c.lane] = sq ? mass_column<true>(raw, part, mp, c.lane) : mass_column<false>(raw, part, mp, c.lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const float S = sq ? mass_fold<true>(raw, cb, mp.n) : mass_fold<false>(raw, cb, mp.n);
                if (c.lane == 0) { Sv[c.wv] = S; Sv[2 + c.wv] = 1.0f / guard_mass(S); }  // IEEE reciprocal, once per row
            }
        } else {
            const int half = c.lane >> 5, col = c.lane & 31;
            const bool use_y = half && !c.dn;
            const float* raw = use_y ? V : U;
            const float* part = use_y ? c.party : c.partx;
            const MassPlan& mp = use_y ? c.mpy : c.mpx;
            float* cb = c.colbuf + 32 * half;
            cb[col] = sq ? mass_column<true>(raw, part, mp, col) : mass_column<false>(raw, part, mp, col);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const float S = sq ? mass_fold<true>(raw, cb, mp.n) : mass_fold<false>(raw, cb, mp.n);
            if (col == 0) { Sv[half] = S; Sv[2 + half] = 1.0f / guard_mass(S); }
        }
        __builtin_amdgcn_s_setprio(0);
        row_sync<G / kWave>();
        SOT_STAMP(3);
        Sx = Sv[0];
        Sy = c.dn ? Sx : Sv[1];
        rSx = Sv[2];
        rSy = c.dn ? rSx : Sv[3];
    }
    Sx_out = Sx; Sy_out = Sy;

    // ---- P3: safe_divide (utils.py:135-142), weight gather by the position sort (losses.py:289-290)
    //      and fp64-accumulated CDFs (losses.py:292-293) --------------------------------------------
    const float Sxh = guard_mass(Sx);
    const float Syh = guard_mass(Sy);
    const bool x_perm = ROWPOS ? c.do_sort : !c.x_ident;
    const bool y_perm = ROWPOS ? c.do_sort : !c.y_ident;
    const bool fullx = (e0 + CPT <= n), fully = (e0 + CPT <= m);
    // identity order + full chunk: two 16-B LDS reads per array; otherwise element-wise (also the gather)
    if (!x_perm && fullx) {
#pragma unroll
        for (int k = 0; k < CPT; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(U + e0 + k);
            wx[k] = v.x; wx[k + 1] = v.y; wx[k + 2] = v.z; wx[k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int e = e0 + k;
            int sx = e;
            if (ROWPOS) sx = ix[k]; else if (x_perm && e < n) sx = a.xperm[e];
            wx[k] = (e < n) ? U[sx] : 0.0f;
        }
    }
    if (!y_perm && fully) {
#pragma unroll
        for (int k = 0; k < CPT; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(V + e0 + k);
            wy[k] = v.x; wy[k + 1] = v.y; wy[k + 2] = v.z; wy[k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int e = e0 + k;
            int sy = e;
            if (ROWPOS) sy = iy[k]; else if (y_perm && e < m) sy = a.yperm[e];
            wy[k] = (e < m) ? V[sy] : 0.0f;
        }
    }
    double px[CPT], py[CPT];
    double runx = 0.0, runy = 0.0;
    {
        // quotients: reciprocal + FMA residual correction (exact, see div_by_row_constant); the chunk is
        // redone with the IEEE sequence if an operand was small enough for the residual to underflow
        const float rx = rSx, ry = rSy;
        float qx[CPT], qy[CPT];
        uint32_t risk = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            if (SOT_ABLATE & 8) { qx[k] = wx[k] * rx; qy[k] = wy[k] * ry; continue; }
            qx[k] = div_by_row_constant(sq ? wx[k] * wx[k] : wx[k], Sxh, rx, risk);
            qy[k] = div_by_row_constant(sq ? wy[k] * wy[k] : wy[k], Syh, ry, risk);
        }
        // Rare fallback, taken by the whole wave if any lane needs it.  The ballot makes the branch wave-uniform and
        // the asm statement keeps hipcc from if-converting it (it would otherwise execute the 16 IEEE divisions
        // unconditionally and select afterwards: +176 VALU per thread per row, seen in the ISA).
        const bool slow = (risk < kFastDivMinBits) || !(Sxh <= 0x1p40f) || !(Syh <= 0x1p40f);
        if (__builtin_amdgcn_ballot_w64(slow) != 0ull) {
            asm volatile("; IEEE division fallback" ::: "memory");
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                qx[k] = (sq ? wx[k] * wx[k] : wx[k]) / Sxh;  // IEEE division (built without fast-math)
                qy[k] = (sq ? wy[k] * wy[k] : wy[k]) / Syh;
            }
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const bool okx = (e0 + k < n), oky = (e0 + k < m);
            runx += okx ? (double)qx[k] : 0.0;
            runy += oky ? (double)qy[k] : 0.0;
            px[k] = runx;
            py[k] = runy;
        }
    }
    const double inx = (SOT_ABLATE & 16) ? runx : wave_incl_scan(runx), iny = (SOT_ABLATE & 16) ? runy : wave_incl_scan(runy);
    double exx = (SOT_ABLATE & 16) ? 0.0 : wave_shift_right1(inx), exy = (SOT_ABLATE & 16) ? 0.0 : wave_shift_right1(iny);
    if (NW > 1 && c.lane == kWave - 1) { c.wtot[c.wv] = inx; c.wtot[NW + c.wv] = iny; }
    SOT_STAMP(4);
    row_sync<G / kWave>();  // every raw weight has been read (also through permutations) before U/V are rewritten
    if (NW > 1) {
        double ox = 0.0, oy = 0.0;
        for (int w = 0; w < c.wv; ++w) { ox += c.wtot[w]; oy += c.wtot[NW + w]; }
        exx += ox;
        exy += oy;
    }
    if (fullx) {
#pragma unroll
        for (int k = 0; k < CPT; k += 4)
            *reinterpret_cast<float4*>(U + e0 + k) = make_float4((float)(exx + px[k]), (float)(exx + px[k + 1]),
                                                                 (float)(exx + px[k + 2]), (float)(exx + px[k + 3]));
    } else {
#pragma unroll
        for (int k = 0; k < CPT; ++k) if (e0 + k < n) U[e0 + k] = (float)(exx + px[k]);
    }
    if (fully) {
#pragma unroll
        for (int k = 0; k < CPT; k += 4)
            *reinterpret_cast<float4*>(V + e0 + k) = make_float4((float)(exy + py[k]), (float)(exy + py[k + 1]),
                                                                 (float)(exy + py[k + 2]), (float)(exy + py[k + 3]));
    } else {
#pragma unroll
        for (int k = 0; k < CPT; ++k) if (e0 + k < m) V[e0 + k] = (float)(exy + py[k]);
    }
    if (ROWPOS && t == 0) { U[n] = INFINITY; V[m] = INFINITY; }
    __builtin_amdgcn_s_setprio(0);
    row_sync<G / kWave>();
}

// left rank of q in a sorted LDS array: #{A_i < q}  (torch.searchsorted side='left', losses.py:219)
__device__ __forceinline__ int lower_rank(const float* A, int len, float q)
{
    int lo = 0, hi = len;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (A[mid] < q) lo = mid + 1; else hi = mid; }
    return lo;
}

// ---------------------------------------------------------------------------------------------
// Forward kernel.
//   G      threads per row (a whole number of wavefronts); 256-thread workgroups hold 256/G rows
//   CPT    contiguous elements of each array owned by a thread during the scan (G*CPT >= max(n,m))
//   ROWPOS positions differ per row (sorted in LDS when REQUIRE_SORT)
//   QUANT  also emit the return_quantiles tensors
//   PM     cost specialisation: 1 -> p == 1, 2 -> p == 2, 0 -> powf
//   LIM    limit_quantile_range: levels Q_k > 1 contribute nothing
//   VEC    rows are 16-B aligned and n, m multiples of 4: 16-B-per-lane global loads
//   SQM    square_dist at compile time (0 / 1) in the specialised p = 1 / p = 2 variants (-2.8 % kernel time), 2 = runtime flag
// Row pipeline: the NEXT row's weights are fetched into registers while the current row is being
// processed in LDS, so HBM latency overlaps the scan/merge work of the same workgroup.
// ---------------------------------------------------------------------------------------------
// Register cap (waves per SIMD) of the one-wave-per-row CSR kernel: 146 VGPRs = three waves per SIMD without it; compiled for four
// (128 VGPRs) config 4's 8192 ragged rows take 30.8 instead of 33.0 us (five: 42.4, six: 56.5 -- spills).  The dense generic kernels
// keep their registers: the same cap on 8192 x 1000 costs 46 -> 58 us (12 elements per thread spill).
#ifndef SOT_CSR_MIN_WAVES
#define SOT_CSR_MIN_WAVES 4
#endif
// Register cap of the per-row-position FORWARD kernels on 256-thread workgroups (the in-register block sort of merge_sort16_kv2 pushes them
// to 168 ... 212 VGPRs = two waves per SIMD, i.e. two of the four workgroups the LDS would hold): SOT_ROWPOS_MIN_WAVES waves per SIMD (7 dwords
// spilled).  Measured 4096 x 2048 paper mode: unsorted rows 246 -> 192 us, sorted rows 77 -> 63 us (one sort block per thread + the cap).
#ifndef SOT_ROWPOS_MIN_WAVES
#define SOT_ROWPOS_MIN_WAVES 3
#endif
// (The generic shared-position kernels of the 2048-point geometry also sit just above a register step in some instantiations -- forward 129 ... 140
// VGPRs, backward 176 -- but holding them to the step changes nothing: forward 52.8 vs 52.8 us, both-gradient backward 120.1 vs 119.7 us at 8192 x 2048.)
template <int G, int CPT, bool ROWPOS, bool QUANT, int PM, bool LIM, bool VEC, bool CSR = false, int SQM = 2>
__global__ __launch_bounds__((G < 256 ? 256 : G), ((CSR && G == 64) ? SOT_CSR_MIN_WAVES : (ROWPOS && !CSR && G <= 256 && CPT == 8) ? SOT_ROWPOS_MIN_WAVES : 1)) void sot_forward_kernel(const FwdArgs a)
{
    static_assert(!CSR || (ROWPOS && !VEC && !QUANT), "the CSR form has per-row positions and unaligned rows");
    constexpr int BLOCK = (G < 256 ? 256 : G);
    constexpr int RPW = BLOCK / G;   // rows processed concurrently by one workgroup
    constexpr int NW = G / kWave;    // wavefronts per row
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef SOT_STAMPS
    const bool wg_stamp = (threadIdx.x == 0) && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1);
    unsigned long long* const wgs = g_stamps + (blockIdx.x == 0 ? 32 : 48);
    int wg_row = 0;
    if (wg_stamp) wgs[0] = __builtin_readcyclecounter();
#endif
    RowCtx<G> c = make_ctx<G, ROWPOS>(a, smem, false);
    const int rg = threadIdx.x / G;
    const int t = c.t;
    float* const U = c.U; float* const V = c.V; float* const PX = c.PX; float* const PY = c.PY;

    const int64_t row_step = (int64_t)gridDim.x * RPW;
    int64_t row0 = (int64_t)blockIdx.x * RPW;
    float rx[CPT], ry[CPT];
    if (!CSR && row0 < a.B) {
        const int64_t r = min(row0 + rg, a.B - 1);
        load_row<G, CPT, VEC>(a.x + r * a.xs, c.n, t, rx);
        load_row<G, CPT, VEC>(a.y + r * a.ys, c.m, t, ry);
    }
#ifdef SOT_STAMPS
    if (wg_stamp) wgs[1] = __builtin_readcyclecounter();
#endif
    for (; row0 < a.B; row0 += row_step) {
        const int64_t row = row0 + rg;
        const bool valid = row < a.B;
        const int64_t rowc = valid ? row : a.B - 1;
#ifdef SOT_STAMPS
        const bool stamp_on = (blockIdx.x == 0) && (threadIdx.x == 0) && (row0 == row_step);
#endif
        SOT_STAMP(0);
        const float* xp = nullptr; const float* yp = nullptr;
        bool bad_row = false;  // CSR: empty or over-long support -> NaN
        if (CSR) {
            const int64_t xo = a.xoff[rowc], xe = a.xoff[rowc + 1], yo = a.yoff[rowc], ye = a.yoff[rowc + 1];
            const int64_t nr = xe - xo, mr = ye - yo;
            bad_row = (nr < 1) || (mr < 1) || (nr > a.n) || (mr > a.m);
            set_row_lengths(c, bad_row ? 1 : (int)nr, bad_row ? 1 : (int)mr);
            const int64_t xb = bad_row ? 0 : xo, yb = bad_row ? 0 : yo;  // entry 0 exists (nnz >= 1 is checked on the host)
            xp = a.xpos + xb; yp = a.ypos + yb;
            load_row<G, CPT, false>(a.x + xb, c.n, t, rx);
            load_row<G, CPT, false>(a.y + yb, c.m, t, ry);
        } else if (ROWPOS) {
            xp = a.xpos + rowc * a.xps; yp = a.ypos + rowc * a.yps;
        }
        const int n = c.n, m = c.m, K = c.K;
        int ix[CPT], iy[CPT];
        if (ROWPOS) {
            const int64_t pw = (int64_t)a.n + a.m;
            rowpos_prepare<G, CPT, !CSR>(c, xp, yp, a.n, a.m, ix, iy, (!CSR && a.perm_in) ? a.perm_in + rowc * pw : nullptr,
                                         (!CSR && a.perm_out && valid) ? a.perm_out + rowc * pw : nullptr);
        }
        // ---- P1: registers -> LDS (original column order), then fetch the next row into the registers --
        store_row<G, CPT, VEC>(U, n, t, rx);
        store_row<G, CPT, VEC>(V, m, t, ry);
        if (!CSR && row0 + row_step < a.B) {
            const int64_t r = min(row0 + row_step + rg, a.B - 1);
            load_row<G, CPT, VEC>(a.x + r * a.xs, n, t, rx);
            load_row<G, CPT, VEC>(a.y + r * a.ys, m, t, ry);
        }
        row_sync<G / kWave>();
        SOT_STAMP(1);
        float wx[CPT], wy[CPT];
        float Sx, Sy;
#ifdef SOT_STAMPS
        build_cdfs<G, CPT, ROWPOS, SQM>(a, c, ix, iy, wx, wy, Sx, Sy, stamp_on);
#else
        build_cdfs<G, CPT, ROWPOS, SQM>(a, c, ix, iy, wx, wy, Sx, Sy);
#endif
        SOT_STAMP(5);

        // ---- P4: merge of the two CDFs = sort(cat(U,V)) + searchsorted + take_along_dim -----------
        //      (losses.py:295-298), level widths, cutoff mask, |.|^p, weighted sum (:301-313)
        float acc = 0.0f;
        __builtin_amdgcn_s_setprio(SOT_WALK_PRIO);
        if (t < c.Ga) {
            // Walk over (Uw, V) where Uw = pad zero levels ++ U: exactly E steps for every thread.
            const float* const Uw = U - c.pad;
            const float* const PXw = PX - c.pad;
            const int nw = n + c.pad;
            const int D0 = t * c.E;
            const uint32_t ub1 = lds_addr(Uw) - 4u;  // partition search on LDS byte addresses (merge_path_steps32)
            const int i0 = (SOT_ABLATE & 2) ? min(D0 >> 1, nw)
                                            : (int)((merge_path_steps32(ub1, lds_addr(V) + 4u * (uint32_t)D0 + ub1, nw, m, D0, c.topk) - ub1) >> 2);
            SOT_STAMP(6);
            const int j0 = D0 - i0;
            float qprev = 0.0f;  // Q_0 := 0 (the pad of losses.py:301)
            if (i0 > 0) qprev = Uw[i0 - 1];
            if (j0 > 0) qprev = fmaxf(qprev, V[j0 - 1]);
            float ua = Uw[i0], vb = V[j0], xa = PXw[i0], yb = PY[j0];
            // Byte offsets from Uw: Uw[i] at 4i; V[j] at 4(voff + j) with i + j = k => 4(voff + k) - 4i.
            const char* const lb = reinterpret_cast<const char*>(Uw);
            const uint32_t poff4 = 4u * (uint32_t)c.L.poff;
            const int voff = (int)(V - Uw);
            uint32_t iu = (uint32_t)i0;
            if (!QUANT) {
                // "Loser" form of the two-way merge (see sot_forward_full.inc): (w, wp) is the head fetched last, (r, rp) the
                // head that lost the previous comparison; only the consumed head's stream is read again, so the fetched
                // level and position become the new (w, wp) without a select.  Equal heads may be consumed in either
                // order: the pair of heads, hence the consumed level, its width and its cost factor, is the same, and the
                // second member of a tie has zero width -- the sum is bit-identical to the canonical order's.
                float w = ua, wp = xa, r = vb, rp = yb;
                const uint32_t lb32 = lds_addr(lb);  // 32-bit LDS addresses: one VGPR per stream, no re-basing add per access
                uint32_t pw = lb32 + 4u * (uint32_t)i0 + 4u, pr = lb32 + 4u * (uint32_t)(voff + j0) + 4u;
#if SOT_WALK_UNROLL > 0
#pragma unroll SOT_WALK_UNROLL
#endif
                for (int s = 0; s < ((SOT_ABLATE & 1) ? 1 : c.E); ++s) {
                    const bool cw = w <= r;
                    const float q = cw ? w : r;
                    const float cost = transport_cost<PM>(wp, rp, c.p);
                    float delta = q - qprev;
                    if (LIM && q > 1.0f) delta = 0.0f;
                    acc = fmaf(delta, cost, acc);  // fused: no worse than the reference's separate rounding
                    qprev = q;
                    const uint32_t nx = cw ? pw : pr;
                    pr = cw ? pr : pw;
                    pw = nx + 4u;
                    r = cw ? r : w;
                    rp = cw ? rp : wp;
                    w = lds_load(nx);
                    wp = lds_load(nx + poff4);
                }
            } else
#if SOT_WALK_UNROLL > 0
#pragma unroll SOT_WALK_UNROLL
#endif
            for (int s = 0; s < ((SOT_ABLATE & 1) ? 1 : c.E); ++s) {
                const bool tu = ua <= vb;
                const float q = tu ? ua : vb;
                const float cost = transport_cost<PM>(xa, yb, c.p);
                float delta = q - qprev;
                if (LIM && q > 1.0f) delta = 0.0f;
                acc = fmaf(delta, cost, acc);  // fused: no worse than the reference's separate rounding
                if (QUANT && valid) {
                    const int k = D0 + s - c.pad;  // index among the real merged levels
                    if (k >= 0) {
                        const int64_t o = row * (int64_t)K + k;
                        if (a.oQ) a.oQ[o] = q;
                        if (a.oUq || a.oVq) {
                            float uqv = xa, vqv = yb;
                            if (q == qprev && k > 0) {  // inside a tie run: searchsorted ranks of its first member
                                uqv = PX[lower_rank(U, n, q)];
                                vqv = PY[lower_rank(V, m, q)];
                            }
                            if (a.oUq) a.oUq[o] = uqv;
                            if (a.oVq) a.oVq[o] = vqv;
                        }
                    }
                }
                qprev = q;
                iu += tu ? 1u : 0u;
                const uint32_t vk = (uint32_t)(voff + D0 + s + 1);  // uniform across the wave up to D0
                const uint32_t off = 4u * (tu ? iu : (vk - iu));
                const float nv = *reinterpret_cast<const float*>(lb + off);
                const float np = *reinterpret_cast<const float*>(lb + off + poff4);
                ua = tu ? nv : ua;
                xa = tu ? np : xa;
                vb = tu ? vb : nv;
                yb = tu ? yb : np;
            }
        }
        if (QUANT && valid) {
            if (a.oU) for (int e = t; e < n; e += G) a.oU[row * (int64_t)n + e] = U[e];
            if (a.oV) for (int e = t; e < m; e += G) a.oV[row * (int64_t)m + e] = V[e];
        }
        SOT_STAMP(7);
        __builtin_amdgcn_s_setprio(0);
        acc = wave_sum(acc);
        if (CSR && bad_row) acc = __int_as_float(0x7fc00000);
        if (NW == 1) {
            if (t == 0 && valid && a.row_loss) store_row_loss(a.row_loss, row, acc, a.mt.counters != nullptr);
            row_sync<G / kWave>();  // this row's LDS reads are done before the next row's staging
        } else {
            if (c.lane == 0) c.red[c.wv] = acc;
            row_sync<G / kWave>();
            if (t == 0 && valid && a.row_loss) {
                float tot = c.red[0];
                for (int w = 1; w < NW; ++w) tot += c.red[w];
                store_row_loss(a.row_loss, row, tot, a.mt.counters != nullptr);  // NaN propagates from any wave of a bad CSR row
            }
        }
        SOT_STAMP(8);
#ifdef SOT_STAMPS
        if (wg_stamp && wg_row < 12) wgs[2 + wg_row++] = __builtin_readcyclecounter();
#endif
    }
    if (a.mt.counters != nullptr) batch_mean_tail<BLOCK>(a.mt, a.row_loss, a.B, reinterpret_cast<double*>(smem));
}

// ---------------------------------------------------------------------------------------------
// Backward kernel: closed form of the autograd graph of losses.py:172-313.
// Recomputes the CDFs in LDS, then
//   g_k  = m_k d_k - m_{k+1} d_{k+1}   per merged level; searchsorted ranks are constant along a run
//          of equal levels, so g is non-zero only at a run's LAST member (stable order: U before V,
//          lower index first), which receives  d(run) - d(next run);
//   ga_i = sum_{i' >= i} gU_i'  (reverse cumsum, fp64);   gS = -sum ga_i w_i / S^2  (if S > 1e-7);
//   dL/dx_i = (ga_i / S + gS) * (2 x_i if square_dist) * grad_row.
// ---------------------------------------------------------------------------------------------
struct BwdArgs {
    FwdArgs f;
    const float* grad_row;     // dL/d(row_loss): [B] (stride 1) or one broadcast scalar (stride 0); null = 1 for every row
    int64_t grad_row_stride;
    float grad_scale;          // multiplies every upstream gradient (1/B of the batch mean)
    float* gx; float* gy;
};

template <int G, int CPT, bool ROWPOS, int PM, bool LIM, bool VEC>
__global__ __launch_bounds__((G < 256 ? 256 : G)) void sot_backward_kernel(const BwdArgs b)   // (per-row positions capped at three waves per SIMD like the per-row forward: 76 dwords spilled, 290 -> 449 us)
{
    constexpr int BLOCK = (G < 256 ? 256 : G);
    constexpr int RPW = BLOCK / G;
    constexpr int NW = G / kWave;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdArgs& a = b.f;
    const RowCtx<G> c = make_ctx<G, ROWPOS>(a, smem, true);
    const int rg = threadIdx.x / G;
    const int n = c.n, m = c.m, t = c.t;
    float* const U = c.U; float* const V = c.V; float* const PX = c.PX; float* const PY = c.PY;

    const int64_t row_step = (int64_t)gridDim.x * RPW;
    int64_t row0 = (int64_t)blockIdx.x * RPW;
    float rx[CPT], ry[CPT];
    if (row0 < a.B) {
        const int64_t r = min(row0 + rg, a.B - 1);
        load_row<G, CPT, VEC>(a.x + r * a.xs, n, t, rx);
        load_row<G, CPT, VEC>(a.y + r * a.ys, m, t, ry);
    }
    for (; row0 < a.B; row0 += row_step) {
        const int64_t row = row0 + rg;
        const bool valid = row < a.B;
        const int64_t rowc = valid ? row : a.B - 1;
        int ix[CPT], iy[CPT];
        if (ROWPOS) {
            const int64_t pw = (int64_t)a.n + a.m;
            rowpos_prepare<G, CPT>(c, a.xpos + rowc * a.xps, a.ypos + rowc * a.yps, a.n, a.m, ix, iy,
                                                   a.perm_in ? a.perm_in + rowc * pw : nullptr, (a.perm_out && valid) ? a.perm_out + rowc * pw : nullptr);
        }
        store_row<G, CPT, VEC>(U, n, t, rx);
        store_row<G, CPT, VEC>(V, m, t, ry);
        if (row0 + row_step < a.B) {
            const int64_t r = min(row0 + row_step + rg, a.B - 1);
            load_row<G, CPT, VEC>(a.x + r * a.xs, n, t, rx);
            load_row<G, CPT, VEC>(a.y + r * a.ys, m, t, ry);
        }
        row_sync<G / kWave>();
        float wx[CPT], wy[CPT];
        float Sx, Sy;
        build_cdfs<G, CPT, ROWPOS>(a, c, ix, iy, wx, wy, Sx, Sy);

        // ---- merge walk over (pad zero levels ++ U, V), exactly E steps per thread.  The gradient of a level
        //      is known one step later (it is non-zero only if the NEXT level starts a new run), so the store
        //      of element k-1 happens at step k; the thread's last element is closed by peeking at level D0+E.
        __builtin_amdgcn_s_setprio(SOT_WALK_PRIO);
        if (t < c.Ga) {
            const float* const Uw = U - c.pad;
            const float* const PXw = PX - c.pad;
            const int nw = n + c.pad;
            const int D0 = t * c.E;
            const uint32_t ub1 = lds_addr(Uw) - 4u;
            const int i0 = (int)((merge_path_steps32(ub1, lds_addr(V) + 4u * (uint32_t)D0 + ub1, nw, m, D0, c.topk) - ub1) >> 2);
            const int j0 = D0 - i0;
            float qprev = 0.0f;
            if (i0 > 0) qprev = Uw[i0 - 1];
            if (j0 > 0) qprev = fmaxf(qprev, V[j0 - 1]);
            float ua = Uw[i0], vb = V[j0], xa = PXw[i0], yb = PY[j0];
            float dcur = 0.0f;
            if (D0 == 0) {
                qprev = __int_as_float(0x7fc00000);  // NaN: the very first level always starts a run
            } else if (fminf(ua, vb) == qprev) {      // we start inside a run: cost at its first member's ranks
                const float q0 = qprev;
                const float cst = transport_cost<PM>(PX[lower_rank(U, n, q0)], PY[lower_rank(V, m, q0)], c.p);
                dcur = (LIM && q0 > 1.0f) ? 0.0f : cst;
            }
            char* const lb = reinterpret_cast<char*>(const_cast<float*>(Uw));
            const uint32_t poff4 = 4u * (uint32_t)c.L.poff;
            const uint32_t goff4 = 4u * (uint32_t)c.L.grad;
            const int voff = (int)(V - Uw);
            uint32_t iu = (uint32_t)i0;
            uint32_t prev_off = 4u * (uint32_t)(c.pad + n);  // U[n]'s gradient slot: a scratch target for "no element yet"
            for (int s = 0; s < c.E; ++s) {
                const bool tu = ua <= vb;
                const float q = tu ? ua : vb;
                float cst = transport_cost<PM>(xa, yb, c.p);
                if (LIM && q > 1.0f) cst = 0.0f;
                const bool new_run = !(q == qprev);
                *reinterpret_cast<float*>(lb + prev_off + goff4) = new_run ? (dcur - cst) : 0.0f;
                dcur = new_run ? cst : dcur;
                qprev = q;
                const uint32_t vk = (uint32_t)(voff + D0 + s);
                prev_off = 4u * (tu ? iu : (vk - iu));  // slot of the element consumed now
                iu += tu ? 1u : 0u;
                const uint32_t off = prev_off + 4u;  // the consumed side's next element
                const float nv = *reinterpret_cast<const float*>(lb + off);
                const float np = *reinterpret_cast<const float*>(lb + off + poff4);
                ua = tu ? nv : ua;
                xa = tu ? np : xa;
                vb = tu ? vb : nv;
                yb = tu ? yb : np;
            }
            {   // close the last element with the level that follows this thread's range (0 cost past the end)
                const float qn = fminf(ua, vb);
                float cn = transport_cost<PM>(xa, yb, c.p);
                if ((LIM && qn > 1.0f) || (t == c.Ga - 1)) cn = 0.0f;
                const bool new_run = !(qn == qprev) || (t == c.Ga - 1);
                *reinterpret_cast<float*>(lb + prev_off + goff4) = new_run ? (dcur - cn) : 0.0f;
            }
        }
        __builtin_amdgcn_s_setprio(0);
        row_sync<G / kWave>();

        // ---- reverse cumsums (fp64), normalisation terms, scatter to the original columns -------------
        const int e0 = t * CPT;
        double ga[CPT], gb[CPT];
        double runx = 0.0, runy = 0.0;
#pragma unroll
        for (int k = CPT - 1; k >= 0; --k) {
            const bool okx = (e0 + k < n), oky = (e0 + k < m);
            runx += okx ? (double)c.GU[e0 + k] : 0.0;
            runy += oky ? (double)c.GV[e0 + k] : 0.0;
            ga[k] = runx;
            gb[k] = runy;
        }
        // exclusive suffix over lanes = wave total - inclusive prefix (fp64: the cancellation is harmless)
        const double pinx = wave_incl_scan(runx), piny = wave_incl_scan(runy);
        const double totwx = wave_last(pinx), totwy = wave_last(piny);
        double exx = totwx - pinx, exy = totwy - piny;
        if (NW > 1) {
            if (c.lane == 0) { c.wtot[c.wv] = totwx; c.wtot[NW + c.wv] = totwy; }
            row_sync<G / kWave>();
            double ox = 0.0, oy = 0.0;
            for (int w = NW - 1; w > c.wv; --w) { ox += c.wtot[w]; oy += c.wtot[NW + w]; }
            exx += ox;
            exy += oy;
        }
        // dot products  sum ga_i * w_i  (w = staged weight, squared if square_dist)
        double dotx = 0.0, doty = 0.0;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            ga[k] += exx;
            gb[k] += exy;
            const bool okx = (e0 + k < n), oky = (e0 + k < m);
            const float sx = c.sq ? wx[k] * wx[k] : wx[k];
            const float sy = c.sq ? wy[k] * wy[k] : wy[k];
            dotx += okx ? ga[k] * (double)sx : 0.0;
            doty += oky ? gb[k] * (double)sy : 0.0;
        }
        dotx = wave_sum(dotx);
        doty = wave_sum(doty);
        double totx = dotx, toty = doty;
        if (NW > 1) {
            row_sync<G / kWave>();  // wtot is reused
            if (c.lane == 0) { c.wtot[c.wv] = dotx; c.wtot[NW + c.wv] = doty; }
            row_sync<G / kWave>();
            totx = 0.0; toty = 0.0;
            for (int w = 0; w < NW; ++w) { totx += c.wtot[w]; toty += c.wtot[NW + w]; }
        }
        const double dx = (double)guard_mass(Sx), dy = (double)guard_mass(Sy);
        const double rdx = 1.0 / dx, rdy = 1.0 / dy;
        double gSx = -totx, gSy = -toty;
        if (c.dn) { gSx += gSy; gSy = 0.0; }
        gSx = (!c.prenorm && Sx > kMassEps) ? gSx * rdx * rdx : 0.0;
        gSy = (!c.prenorm && Sy > kMassEps) ? gSy * rdy * rdy : 0.0;
        const double gr = (b.grad_row ? (double)b.grad_row[rowc * b.grad_row_stride] : 1.0) * (double)b.grad_scale;
        if (valid) {
            const bool x_perm = ROWPOS ? c.do_sort : !c.x_ident;
            const bool y_perm = ROWPOS ? c.do_sort : !c.y_ident;
            float ox[CPT], oy[CPT];
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                double g = ga[k] * rdx + gSx;
                if (c.sq) g *= 2.0 * (double)wx[k];
                ox[k] = (float)(g * gr);
                double h = gb[k] * rdy + gSy;
                if (c.sq) h *= 2.0 * (double)wy[k];
                oy[k] = (float)(h * gr);
            }
            if (b.gx) {
                float* dst = b.gx + row * (int64_t)n;
                if (VEC && !x_perm && (e0 + CPT <= n)) {
#pragma unroll
                    for (int k = 0; k < CPT; k += 4)
                        *reinterpret_cast<float4*>(dst + e0 + k) = make_float4(ox[k], ox[k + 1], ox[k + 2], ox[k + 3]);
                } else {
#pragma unroll
                    for (int k = 0; k < CPT; ++k) {
                        const int e = e0 + k;
                        if (e < n) dst[ROWPOS ? (x_perm ? ix[k] : e) : (x_perm ? a.xperm[e] : e)] = ox[k];
                    }
                }
            }
            if (b.gy) {
                float* dst = b.gy + row * (int64_t)m;
                if (VEC && !y_perm && (e0 + CPT <= m)) {
#pragma unroll
                    for (int k = 0; k < CPT; k += 4)
                        *reinterpret_cast<float4*>(dst + e0 + k) = make_float4(oy[k], oy[k + 1], oy[k + 2], oy[k + 3]);
                } else {
#pragma unroll
                    for (int k = 0; k < CPT; ++k) {
                        const int e = e0 + k;
                        if (e < m) dst[ROWPOS ? (y_perm ? iy[k] : e) : (y_perm ? a.yperm[e] : e)] = oy[k];
                    }
                }
            }
        }
        row_sync<G / kWave>();  // GU/GV/wtot reads done before the next row reuses LDS
    }
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
constexpr size_t kLdsLimit = 160 * 1024;

// Allow a kernel to use up to the CU's full 160 KiB of dynamic LDS; leaves no sticky error behind.
static inline void allow_full_lds(const void* kernel)
{
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit) != hipSuccess)
        (void)hipGetLastError();
}

// Launch state is kept PER DEVICE and behind a mutex: a process may use several GPUs (the binding switches devices per
// call), and calls arrive from several host threads (autograd runs backward on its own thread; ctypes drops the GIL).
constexpr int kMaxDevices = 64;
static inline int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return dev;
}

static inline int device_cu_count()
{
    static std::mutex mu;
    static int cus[kMaxDevices] = {};
    const int dev = current_device();
    const bool cacheable = dev >= 0 && dev < kMaxDevices;
    std::lock_guard<std::mutex> lock(mu);
    if (cacheable && cus[dev] > 0) return cus[dev];
    hipDeviceProp_t prop;
    int n = 256;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
    else (void)hipGetLastError();
    if (cacheable) cus[dev] = n;
    return n;
}

// Experiment knob (never set in production): SOT_DEBUG_EXTRA_LDS=<bytes> pads the dynamic LDS request of the
// forward kernel to throttle its occupancy, which separates latency-bound from issue-bound behaviour.
static inline size_t debug_extra_lds()
{
    const char* e = getenv("SOT_DEBUG_EXTRA_LDS");
    return e ? (size_t)atol(e) : 0;
}

struct LaunchCfg { int G, CPT; };

// Row-group geometries: (threads per row, contiguous elements per thread).  G*CPT >= max(n, m).
// (128,12) serves the paper's row lengths just above 1024 (n_fft 2048 -> 1025 bins): two rows per workgroup.
static inline bool pick_cfg(int n, int m, bool rowpos, bool with_grad, LaunchCfg* cfg, size_t* lds_bytes, int* block, int* rpw)
{
    const int N = n > m ? n : m;
    static const LaunchCfg table[] = {{64, 8}, {128, 12}, {256, 8}, {1024, 8}, {1024, 16}};
    for (int ci = 0; ci < 5; ++ci) {
        const LaunchCfg& c = table[ci];
        if ((int64_t)c.G * c.CPT < N) continue;
        const int blk = c.G < 256 ? 256 : c.G;
        const int r = blk / c.G;
        const RowLayout L = make_layout(n, m, c.G, rowpos, with_grad);
        const size_t bytes = (size_t)r * L.row_floats * sizeof(float);
        if (bytes > kLdsLimit) continue;
        *cfg = c; *lds_bytes = bytes; *block = blk; *rpw = r;
        return true;
    }
    return false;
}

// Persistent grid: exactly as many workgroups as are co-resident (registers, LDS and wave slots all
// taken into account by the occupancy query), never more than there are row groups.
template <typename Kernel>
static inline int resident_grid(Kernel kern, int block, size_t lds, int64_t want)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, block, lds) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    const int64_t cap = (int64_t)device_cu_count() * per_cu;
    return (int)(want < cap ? want : cap);
}

// resident_grid() of one kernel instantiation, cached per device (and per LDS size: the generic kernels' LDS request
// depends on n, m); the first use on a device also opts the kernel in to the CU's full LDS THERE (hipFuncSetAttribute
// applies to the current device only).  One cache per call site: `Tag` is the kernel's own function-pointer type.
struct GridCache {
    std::mutex mu;
    struct Entry { size_t lds; int grid; bool attr; } e[kMaxDevices] = {};
};

// Grid of a persistent kernel whose workgroups stride over `want` row groups, at most `cap` of them resident.  SOT_BALANCED_GRID = 1
// gives every workgroup the same number of row groups (ceil(want / rounds) workgroups) instead of `cap` workgroups of which some run
// one round more.  Measured (8192 x 2048): merge-free and merge forward unchanged (30.0 / 45.4 us), training form 86.2 instead of
// 81.5 us (745 instead of 768 workgroups leave some CUs with two resident workgroups instead of three for the whole launch): off.
#ifndef SOT_BALANCED_GRID
#define SOT_BALANCED_GRID 0
#endif
static inline int balanced_grid(int64_t want, int cap)
{
    if (want <= cap) return (int)want;
    if (!SOT_BALANCED_GRID) return cap;
    const int64_t rounds = (want + cap - 1) / cap;
    return (int)((want + rounds - 1) / rounds);
}

template <typename Kernel>
static inline int cached_resident_grid(GridCache& gc, Kernel kern, int block, size_t lds)
{
    const int dev = current_device();
    if (dev < 0 || dev >= kMaxDevices) {
        allow_full_lds(reinterpret_cast<const void*>(kern));
        return resident_grid(kern, block, lds, INT32_MAX);
    }
    std::lock_guard<std::mutex> lock(gc.mu);
    GridCache::Entry& e = gc.e[dev];
    if (!e.attr) { allow_full_lds(reinterpret_cast<const void*>(kern)); e.attr = true; }
    if (e.grid == 0 || e.lds != lds) { e.grid = resident_grid(kern, block, lds, INT32_MAX); e.lds = lds; }
    return e.grid;
}

// once per device: opt `kernel` in to the full LDS (kernels launched with a fixed grid)
static inline void allow_full_lds_once(GridCache& gc, const void* kernel)
{
    const int dev = current_device();
    if (dev < 0 || dev >= kMaxDevices) { allow_full_lds(kernel); return; }
    std::lock_guard<std::mutex> lock(gc.mu);
    if (!gc.e[dev].attr) { allow_full_lds(kernel); gc.e[dev].attr = true; }
}

static inline int validate(const sot_problem* pr)
{
    if (pr == nullptr) return SOT_ERR_NULL_POINTER;
    if (!(pr->p >= 1.0f)) return SOT_ERR_INVALID_P;
    if (pr->B < 0 || pr->n < 1 || pr->m < 1) return SOT_ERR_BAD_SHAPE;
    if (pr->x_row_stride < pr->n || pr->y_row_stride < pr->m) return SOT_ERR_BAD_SHAPE;
    if (pr->xpos_row_stride != 0 && pr->xpos_row_stride < pr->n) return SOT_ERR_BAD_SHAPE;
    if (pr->ypos_row_stride != 0 && pr->ypos_row_stride < pr->m) return SOT_ERR_BAD_SHAPE;
    if ((pr->xpos_row_stride == 0) != (pr->ypos_row_stride == 0)) return SOT_ERR_BAD_SHAPE;
    if (pr->B > 0 && (!pr->x || !pr->y || !pr->xpos || !pr->ypos)) return SOT_ERR_NULL_POINTER;
    return SOT_OK;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WsLayout { size_t sx, sy, px, py, ident, total; };
// the permutation image of a per-row-position call (setup_launch: the pre-sort kernel's output when the caller passes no row_perm_out)
static inline size_t rowpos_perm_bytes(int64_t B, int n, int m) { return align_up((size_t)B * ((size_t)n + (size_t)m) * sizeof(uint16_t), 256); }

static inline WsLayout ws_layout(int n, int m)
{
    WsLayout w;
    size_t o = 0;
    w.sx = o; o = align_up(o + sizeof(float) * (size_t)n, 256);
    w.sy = o; o = align_up(o + sizeof(float) * (size_t)m, 256);
    w.px = o; o = align_up(o + sizeof(int) * (size_t)n, 256);
    w.py = o; o = align_up(o + sizeof(int) * (size_t)m, 256);
    w.ident = o; o = align_up(o + 2 * sizeof(int), 256);
    w.total = o;
    return w;
}

// ---- kernel-attached timing (sot_profile_next_launch, include/sot_hip.h): when armed by the calling thread, the next launch of
// a full-row kernel goes through hipExtLaunchKernelGGL with a start / stop event pair of the library's ring, i.e. the events
// bracket the dispatch itself (what rocprofv3's kernel trace measures) instead of stream time around it.
bool profile_take(hipEvent_t* start, hipEvent_t* stop);

template <typename Kernel, typename Args>
static inline void launch_maybe_profiled(Kernel kern, int grid, int block, size_t lds, hipStream_t s, const Args& a)
{
    hipEvent_t e0, e1;
    if (profile_take(&e0, &e1)) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), (uint32_t)lds, s, e0, e1, 0, a);
    else hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, a);
}

// ---- cross-part host interface (the parts are linked into one shared library) ---------------------------
struct Launch {
    FwdArgs a;
    LaunchCfg cfg;
    size_t lds;
    int block;
    int64_t want;  // row groups' worth of workgroups
    bool rowpos, vec;
    int pm;        // cost specialisation: 1 -> p == 1, 2 -> p == 2, 0 -> general
    hipStream_t s;
};

template <bool ROWPOS>
hipError_t dispatch_forward(const LaunchCfg& c, bool quant, int pm, bool vec, const FwdArgs& a, size_t lds, int64_t want, int block,
                            hipStream_t s);
template <bool ROWPOS>
hipError_t dispatch_backward(const LaunchCfg& c, int pm, bool vec, const BwdArgs& b, size_t lds, int64_t want, int block,
                             hipStream_t s);
hipError_t dispatch_forward_full(const LaunchCfg& c, int pm, const FwdArgs& a, size_t lds, int64_t want, int block, hipStream_t s);
hipError_t dispatch_forward_full_rowpos(int pm, const FwdArgs& a, hipStream_t s);   // per-row positions through handed-over permutations, 2048-point rows (round 6)
hipError_t dispatch_backward_full(const LaunchCfg& c, int pm, const BwdArgs& b, hipStream_t s);
hipError_t dispatch_backward_full_rowpos(int pm, const BwdArgs& b, hipStream_t s);   // per-row positions through handed-over permutations, 2048-point rows (round 6)
hipError_t dispatch_area_full(const FwdArgs& a, hipStream_t s);
hipError_t dispatch_area_train(const BwdArgs& b, hipStream_t s);
bool area_train_supports(int n);
bool forward_full_supports(int n, bool aligned16);
bool backward_full_supports(int n, bool aligned16);
int full_rt_capacity(int n);   // capacity of the compile-time geometry that takes a run-time row length n (0: none)
hipError_t dispatch_forward_full_rt(int pm, const FwdArgs& a, hipStream_t s);
hipError_t dispatch_area_full_rt(const FwdArgs& a, hipStream_t s);
hipError_t dispatch_backward_full_rt(int pm, const BwdArgs& b, hipStream_t s);
int launch_prepare(const float* xpos, const float* ypos, int n, int m, float* sx, float* sy, int* px, int* py, int* ident,
                   hipStream_t s, bool unit = false);
int setup_launch(const sot_problem* pr, bool with_grad, void* workspace, size_t workspace_bytes, void* stream, Launch* out);
int run_forward(const sot_problem* pr, float* row_loss, float* uq, float* vq, float* Q, float* U, float* V, bool quant,
                void* workspace, size_t workspace_bytes, void* stream, const MeanTail* mean_tail = nullptr);
int run_backward(const sot_problem* pr, const float* grad_row, int64_t grad_row_stride, float grad_scale, float* gx, float* gy,
                 void* workspace, size_t workspace_bytes, void* stream, float* row_loss_out = nullptr, bool* fused = nullptr,
                 const MeanTail* mean_tail = nullptr);
int run_forward_csr(const float* xw, const float* xp, const int64_t* xoff, int64_t x_nnz, const float* yw, const float* yp,
                    const int64_t* yoff, int64_t y_nnz, int64_t B, int max_n, int max_m, float p, uint32_t flags, float* row_loss,
                    void* stream);
int run_position_grad(const sot_problem* pr, const float* grad_row, int64_t grad_row_stride, float grad_scale, float* gxp, float* gyp,
                      void* workspace, size_t workspace_bytes, void* stream);
int run_column_sum(const float* rows, int64_t B, int n, int64_t stride, float* out, void* stream);

#ifdef SOT_STUB_MISSING_PARTS
// diagnostic single-file builds (stamps / ablation) compile a subset of the parts: resolve the rest with stubs
#if !(SOT_PART & 1)
template <> hipError_t dispatch_forward<false>(const LaunchCfg&, bool, int, bool, const FwdArgs&, size_t, int64_t, int, hipStream_t) { return hipErrorInvalidDeviceFunction; }
#endif
#if !(SOT_PART & 2)
template <> hipError_t dispatch_forward<true>(const LaunchCfg&, bool, int, bool, const FwdArgs&, size_t, int64_t, int, hipStream_t) { return hipErrorInvalidDeviceFunction; }
#endif
#if !(SOT_PART & 4)
template <> hipError_t dispatch_backward<false>(const LaunchCfg&, int, bool, const BwdArgs&, size_t, int64_t, int, hipStream_t) { return hipErrorInvalidDeviceFunction; }
#endif
#if !(SOT_PART & 8)
template <> hipError_t dispatch_backward<true>(const LaunchCfg&, int, bool, const BwdArgs&, size_t, int64_t, int, hipStream_t) { return hipErrorInvalidDeviceFunction; }
#endif
#if !(SOT_PART & 128)
hipError_t dispatch_forward_full(const LaunchCfg&, int, const FwdArgs&, size_t, int64_t, int, hipStream_t) { return hipErrorInvalidDeviceFunction; }
hipError_t dispatch_forward_full_rowpos(int, const FwdArgs&, hipStream_t) { return hipErrorInvalidDeviceFunction; }
hipError_t dispatch_backward_full(const LaunchCfg&, int, const BwdArgs&, hipStream_t) { return hipErrorInvalidDeviceFunction; }
hipError_t dispatch_backward_full_rowpos(int, const BwdArgs&, hipStream_t) { return hipErrorInvalidDeviceFunction; }
hipError_t dispatch_area_full(const FwdArgs&, hipStream_t) { return hipErrorInvalidDeviceFunction; }
hipError_t dispatch_area_train(const BwdArgs&, hipStream_t) { return hipErrorInvalidDeviceFunction; }
bool area_train_supports(int) { return false; }
bool forward_full_supports(int, bool) { return false; }
bool backward_full_supports(int, bool) { return false; }
int full_rt_capacity(int) { return 0; }
#endif
#if !(SOT_PART & 512)
hipError_t dispatch_forward_full_rt(int, const FwdArgs&, hipStream_t) { return hipErrorInvalidDeviceFunction; }
hipError_t dispatch_area_full_rt(const FwdArgs&, hipStream_t) { return hipErrorInvalidDeviceFunction; }
#endif
#if !(SOT_PART & 1024)
hipError_t dispatch_backward_full_rt(int, const BwdArgs&, hipStream_t) { return hipErrorInvalidDeviceFunction; }
#endif
#if !(SOT_PART & 32)
int run_forward_csr(const float*, const float*, const int64_t*, int64_t, const float*, const float*, const int64_t*, int64_t, int64_t, int,
                    int, float, uint32_t, float*, void*) { return SOT_ERR_LAUNCH; }
#endif
#if !(SOT_PART & 2048)
int run_position_grad(const sot_problem*, const float*, int64_t, float, float*, float*, void*, size_t, void*) { return SOT_ERR_LAUNCH; }
int run_column_sum(const float*, int64_t, int, int64_t, float*, void*) { return SOT_ERR_LAUNCH; }
#endif
#endif  // SOT_STUB_MISSING_PARTS

#if SOT_PART & 67
template <int G, int CPT, bool ROWPOS, bool QUANT, int PM, bool LIM, bool VEC, int SQM = 2>
static hipError_t launch_forward(const FwdArgs& a, size_t lds, int64_t want, int block, hipStream_t s)
{
    auto kern = sot_forward_kernel<G, CPT, ROWPOS, QUANT, PM, LIM, VEC, false, SQM>;
    static const size_t extra_lds = debug_extra_lds();
    lds += extra_lds;
    static GridCache cache;  // per instantiation (function-local static: thread-safe initialisation)
    const int grid_cap = cached_resident_grid(cache, kern, block, lds);
    const int grid = balanced_grid(want, grid_cap);
    (void)hipGetLastError();  // do not inherit a stale error from earlier runtime calls
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, a);
    return hipGetLastError();
}

template <int G, int CPT, bool ROWPOS, bool LIM, bool VEC>
static hipError_t dispatch_forward_pm(int pm, const FwdArgs& a, size_t lds, int64_t want, int block, hipStream_t s)
{
    // p = 1 and p = 2 get square_dist at compile time as well (one multiply + select per element less); any other p
    // goes through the generic variant (powf, runtime flag)
    const bool sq = (a.flags & SOT_FLAG_SQUARE) && !(a.flags & SOT_FLAG_PRENORMALIZED);
    switch (pm) {
        case 1: return sq ? launch_forward<G, CPT, ROWPOS, false, 1, LIM, VEC, 1>(a, lds, want, block, s)
                          : launch_forward<G, CPT, ROWPOS, false, 1, LIM, VEC, 0>(a, lds, want, block, s);
        case 2: return sq ? launch_forward<G, CPT, ROWPOS, false, 2, LIM, VEC, 1>(a, lds, want, block, s)
                          : launch_forward<G, CPT, ROWPOS, false, 2, LIM, VEC, 0>(a, lds, want, block, s);
        default: return launch_forward<G, CPT, ROWPOS, false, 0, LIM, VEC, 2>(a, lds, want, block, s);
    }
}

template <int G, int CPT, bool LIM>
hipError_t forward_shared(int pm, bool vec, const FwdArgs& a, size_t lds, int64_t want, int block, hipStream_t s)
{
    return vec ? dispatch_forward_pm<G, CPT, false, LIM, true>(pm, a, lds, want, block, s)
               : dispatch_forward_pm<G, CPT, false, LIM, false>(pm, a, lds, want, block, s);
}
// explicit instantiation of one LIM family per build part; the other family is an external symbol of this part
#define SOT_FWD_SHARED_ALL(PREFIX, LIMV)                                                                              \
    PREFIX template hipError_t forward_shared<64, 8, LIMV>(int, bool, const FwdArgs&, size_t, int64_t, int, hipStream_t);   \
    PREFIX template hipError_t forward_shared<128, 12, LIMV>(int, bool, const FwdArgs&, size_t, int64_t, int, hipStream_t); \
    PREFIX template hipError_t forward_shared<256, 8, LIMV>(int, bool, const FwdArgs&, size_t, int64_t, int, hipStream_t);  \
    PREFIX template hipError_t forward_shared<1024, 8, LIMV>(int, bool, const FwdArgs&, size_t, int64_t, int, hipStream_t); \
    PREFIX template hipError_t forward_shared<1024, 16, LIMV>(int, bool, const FwdArgs&, size_t, int64_t, int, hipStream_t);
#if SOT_PART & 1
SOT_FWD_SHARED_ALL(, false)
#else
SOT_FWD_SHARED_ALL(extern, false)
#endif
#if SOT_PART & 64
SOT_FWD_SHARED_ALL(, true)
#else
SOT_FWD_SHARED_ALL(extern, true)
#endif

template <int G, int CPT, bool ROWPOS>
static hipError_t dispatch_forward_g(bool quant, int pm, bool vec, const FwdArgs& a, size_t lds, int64_t want, int block, hipStream_t s)
{
    const bool lim = a.flags & SOT_FLAG_LIMIT_Q;
    if (quant)  // rare path: one generic build per cutoff flavour
        return lim ? launch_forward<G, CPT, ROWPOS, true, 0, true, false>(a, lds, want, block, s)
                   : launch_forward<G, CPT, ROWPOS, true, 0, false, false>(a, lds, want, block, s);
    if constexpr (ROWPOS) {
        return lim ? dispatch_forward_pm<G, CPT, ROWPOS, true, false>(pm, a, lds, want, block, s)
                   : dispatch_forward_pm<G, CPT, ROWPOS, false, false>(pm, a, lds, want, block, s);
    } else {
        // shared positions: the cutoff (LIM) and no-cutoff families are compiled in different build parts
#if defined(SOT_STUB_MISSING_PARTS) && !(SOT_PART & 64)
        if (lim) return hipErrorInvalidDeviceFunction;  // diagnostic build without the cutoff family
#else
        if (lim) return forward_shared<G, CPT, true>(pm, vec, a, lds, want, block, s);
#endif
        return forward_shared<G, CPT, false>(pm, vec, a, lds, want, block, s);
    }
}

template <bool ROWPOS>
hipError_t dispatch_forward(const LaunchCfg& c, bool quant, int pm, bool vec, const FwdArgs& a, size_t lds, int64_t want, int block,
                            hipStream_t s)
{
    if (c.CPT == 16) return dispatch_forward_g<1024, 16, ROWPOS>(quant, pm, vec, a, lds, want, block, s);
    switch (c.G) {
        case 64: return dispatch_forward_g<64, 8, ROWPOS>(quant, pm, vec, a, lds, want, block, s);
        case 128: return dispatch_forward_g<128, 12, ROWPOS>(quant, pm, vec, a, lds, want, block, s);
        case 256: return dispatch_forward_g<256, 8, ROWPOS>(quant, pm, vec, a, lds, want, block, s);
        default: return dispatch_forward_g<1024, 8, ROWPOS>(quant, pm, vec, a, lds, want, block, s);
    }
}

#if SOT_PART & 1
template hipError_t dispatch_forward<false>(const LaunchCfg&, bool, int, bool, const FwdArgs&, size_t, int64_t, int, hipStream_t);
#endif
#if SOT_PART & 2
template hipError_t dispatch_forward<true>(const LaunchCfg&, bool, int, bool, const FwdArgs&, size_t, int64_t, int, hipStream_t);
#endif
#endif  // forward parts

// bit 7: the compile-time-length forward kernels (merge and merge-free), bit 8: the compile-time-length backward kernels.  Diagnostic
// single-file builds (SOT_STUB_MISSING_PARTS: tools/) select both with bit 7 alone, as before the split.
// Bits 9 / 10: the same kernels for a RUN-TIME row length (any n <= 8192 on the next capacity's geometry), forward / backward.
#if defined(SOT_STUB_MISSING_PARTS) && (SOT_PART & 128)
#define SOT_FULL_FWD 1
#define SOT_FULL_BWD 1
#else
#define SOT_FULL_FWD ((SOT_PART & 128) != 0)
#define SOT_FULL_BWD ((SOT_PART & 256) != 0)
#endif
#define SOT_FULL_RT_FWD ((SOT_PART & 512) != 0)
#define SOT_FULL_RT_BWD ((SOT_PART & 1024) != 0)
#if SOT_FULL_FWD || SOT_FULL_BWD || SOT_FULL_RT_FWD || SOT_FULL_RT_BWD
#include "sot_forward_full.inc"
#endif

#if SOT_PART & 12
template <int G, int CPT, bool ROWPOS, int PM, bool LIM, bool VEC>
static hipError_t launch_backward(const BwdArgs& b, size_t lds, int64_t want, int block, hipStream_t s)
{
    auto kern = sot_backward_kernel<G, CPT, ROWPOS, PM, LIM, VEC>;
    static GridCache cache;  // per instantiation (function-local static: thread-safe initialisation)
    const int grid_cap = cached_resident_grid(cache, kern, block, lds);
    const int grid = balanced_grid(want, grid_cap);
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, b);
    return hipGetLastError();
}

template <int G, int CPT, bool ROWPOS, bool LIM, bool VEC>
static hipError_t dispatch_backward_pm(int pm, const BwdArgs& b, size_t lds, int64_t want, int block, hipStream_t s)
{
    switch (pm) {
        case 1: return launch_backward<G, CPT, ROWPOS, 1, LIM, VEC>(b, lds, want, block, s);
        case 2: return launch_backward<G, CPT, ROWPOS, 2, LIM, VEC>(b, lds, want, block, s);
        default: return launch_backward<G, CPT, ROWPOS, 0, LIM, VEC>(b, lds, want, block, s);
    }
}

template <int G, int CPT, bool ROWPOS>
static hipError_t dispatch_backward_g(int pm, bool vec, const BwdArgs& b, size_t lds, int64_t want, int block, hipStream_t s)
{
    const bool lim = b.f.flags & SOT_FLAG_LIMIT_Q;
    if (ROWPOS || !vec)
        return lim ? dispatch_backward_pm<G, CPT, ROWPOS, true, false>(pm, b, lds, want, block, s)
                   : dispatch_backward_pm<G, CPT, ROWPOS, false, false>(pm, b, lds, want, block, s);
    return lim ? dispatch_backward_pm<G, CPT, false, true, true>(pm, b, lds, want, block, s)
               : dispatch_backward_pm<G, CPT, false, false, true>(pm, b, lds, want, block, s);
}

template <bool ROWPOS>
hipError_t dispatch_backward(const LaunchCfg& c, int pm, bool vec, const BwdArgs& b, size_t lds, int64_t want, int block,
                             hipStream_t s)
{
    if (c.CPT == 16) return dispatch_backward_g<1024, 16, ROWPOS>(pm, vec, b, lds, want, block, s);
    switch (c.G) {
        case 64: return dispatch_backward_g<64, 8, ROWPOS>(pm, vec, b, lds, want, block, s);
        case 128: return dispatch_backward_g<128, 12, ROWPOS>(pm, vec, b, lds, want, block, s);
        case 256: return dispatch_backward_g<256, 8, ROWPOS>(pm, vec, b, lds, want, block, s);
        default: return dispatch_backward_g<1024, 8, ROWPOS>(pm, vec, b, lds, want, block, s);
    }
}

#if SOT_PART & 4
template hipError_t dispatch_backward<false>(const LaunchCfg&, int, bool, const BwdArgs&, size_t, int64_t, int, hipStream_t);
#endif
#if SOT_PART & 8
template hipError_t dispatch_backward<true>(const LaunchCfg&, int, bool, const BwdArgs&, size_t, int64_t, int, hipStream_t);
#endif
#endif  // backward parts

#if SOT_PART & 2048
// ---------------------------------------------------------------------------------------------
// Gradients w.r.t. the SUPPORT POSITIONS (round 4; losses.py:287-298, 214-220: the positions enter the loss through torch.sort and
// take_along_dim, both differentiable -- no reference call site asks for this gradient, but the reference's autograd supplies it).
// With delta_k the width of merged level k and (i_k, j_k) the searchsorted ranks of Q_k in U and V (clamped to n-1 / m-1),
//     d row_loss / d xs[i] =  sum_{k : i_k = i} delta_k * p |xs[i_k] - ys[j_k]|^(p-1) sign(xs[i_k] - ys[j_k]),   ys[j]: minus the same.
// The ranks of level k are the numbers of U / V levels consumed before step k of the merge walk (losses.py:219 `searchsorted` is
// side='left'; a level whose rank that misstates -- the second member of a tie -- has zero width), so the terms of xs[i] are the
// walk steps between the consumption of U[i-1] and of U[i], the latter included: a contiguous range of steps that may span several
// threads.  Deterministic, no atomics: the thread that consumes U[i] ASSIGNS the sum of its own steps since the start of its segment
// (or since its previous U) to slot i; every thread leaves the sum behind its last U as a (slot, value) TAIL; after a barrier the
// first thread of each run of equal tail slots adds the run's values, in thread order, to the slot.  Same for V.  Levels past the
// last U level rank n and are clamped to n-1 (losses.py:220): slot n collects them and is folded into slot n-1.
// One kernel for every p and for the cutoff (run-time switches: this is not a hot path).  Output: per-row gradients in the
// caller's ORIGINAL column order (through the sort permutation), already multiplied by the upstream gradient of the row.
// ---------------------------------------------------------------------------------------------
struct PosGradArgs {
    FwdArgs f;
    const float* grad_row; int64_t grad_row_stride; float grad_scale;
    float* gxp; float* gyp;   // [B, n] / [B, m], either may be null
};

__device__ __forceinline__ float cost_slope(float d, int pm, float p)
{
    if (pm == 1) return (float)(d > 0.0f) - (float)(d < 0.0f);   // d |d| / dd, 0 at 0 (torch.abs backward)
    if (pm == 2) return 2.0f * d;                                   // pow(2) backward: 2 |d| sign(d)
    return copysignf(p * pow_nonneg(fabsf(d), p - 1.0f), d);        // pow_nonneg(0, .) = 0
}

template <int G, int CPT, bool ROWPOS>
__global__ __launch_bounds__((G < 256 ? 256 : G)) void sot_position_grad_kernel(const PosGradArgs b)   // (capped at three waves per SIMD: 28 dwords spilled, 279 -> 347 us)
{
    constexpr int BLOCK = (G < 256 ? 256 : G);
    constexpr int RPW = BLOCK / G;
    constexpr int NW = G / kWave;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdArgs& a = b.f;
    const RowCtx<G> c = make_ctx<G, ROWPOS>(a, smem, true);
    // The per-thread tails (value, slot) x 2 of the Ga walking threads: in the row's own CDF region when it is large enough (long rows:
    // 4 Ga <= n + m; the CDFs are dead once every thread has finished its walk), otherwise behind the row regions (posgrad_tail_floats)
    const int rg = threadIdx.x / G;
    const bool tails_in_cdfs = 4 * c.Ga <= c.n + c.m;
    const int tstride = tails_in_cdfs ? c.Ga : G;
    float* const tail_x = tails_in_cdfs ? c.U : smem + RPW * c.L.row_floats + rg * 4 * G;   // value behind the thread's last U
    int* const tail_i = reinterpret_cast<int*>(tail_x + tstride);                             // its slot (index into Uw)
    float* const tail_y = tail_x + 2 * tstride;
    int* const tail_j = reinterpret_cast<int*>(tail_x + 3 * tstride);
    const int n = c.n, m = c.m, t = c.t;
    float* const U = c.U; float* const V = c.V; float* const PX = c.PX; float* const PY = c.PY;
    const int pm = (a.p == 1.0f) ? 1 : ((a.p == 2.0f) ? 2 : 0);
    const bool lim = c.lim;

    const int64_t row_step = (int64_t)gridDim.x * RPW;
    int64_t row0 = (int64_t)blockIdx.x * RPW;
    float rx[CPT], ry[CPT];
    if (row0 < a.B) {
        const int64_t r = min(row0 + rg, a.B - 1);
        load_row<G, CPT, false>(a.x + r * a.xs, n, t, rx);
        load_row<G, CPT, false>(a.y + r * a.ys, m, t, ry);
    }
    for (; row0 < a.B; row0 += row_step) {
        const int64_t row = row0 + rg;
        const bool valid = row < a.B;
        const int64_t rowc = valid ? row : a.B - 1;
        int ix[CPT], iy[CPT];
        if (ROWPOS) {
            const int64_t pw = (int64_t)a.n + a.m;
            rowpos_prepare<G, CPT>(c, a.xpos + rowc * a.xps, a.ypos + rowc * a.yps, a.n, a.m, ix, iy,
                                                   a.perm_in ? a.perm_in + rowc * pw : nullptr, (a.perm_out && valid) ? a.perm_out + rowc * pw : nullptr);
        }
        store_row<G, CPT, false>(U, n, t, rx);
        store_row<G, CPT, false>(V, m, t, ry);
        if (row0 + row_step < a.B) {
            const int64_t r = min(row0 + row_step + rg, a.B - 1);
            load_row<G, CPT, false>(a.x + r * a.xs, n, t, rx);
            load_row<G, CPT, false>(a.y + r * a.ys, m, t, ry);
        }
        row_sync<NW>();
        float wx[CPT], wy[CPT];
        float Sx, Sy;
        build_cdfs<G, CPT, ROWPOS>(a, c, ix, iy, wx, wy, Sx, Sy);

        float* const GUw = c.GU - c.pad;   // slot of Uw[i]; GUw[n + pad] = GU[n]: the levels past the last U level
        if (t == 0) { c.GU[n] = 0.0f; c.GV[m] = 0.0f; }   // nobody consumes the sentinels: these two slots only receive tails
        float tx = 0.0f, ty = 0.0f;
        int ti = -1, tj = -1;
        if (t < c.Ga) {
            const float* const Uw = U - c.pad;
            const float* const PXw = PX - c.pad;
            const int nw = n + c.pad;
            const int D0 = t * c.E;
            const uint32_t ub1 = lds_addr(Uw) - 4u;
            const int i0 = (int)((merge_path_steps32(ub1, lds_addr(V) + 4u * (uint32_t)D0 + ub1, nw, m, D0, c.topk) - ub1) >> 2);
            const int j0 = D0 - i0;
            float qprev = 0.0f;   // Q_0 := 0 (the pad of losses.py:301)
            if (i0 > 0) qprev = Uw[i0 - 1];
            if (j0 > 0) qprev = fmaxf(qprev, V[j0 - 1]);
            float ua = Uw[i0], vb = V[j0], xa = PXw[i0], yb = PY[j0];
            char* const lb = reinterpret_cast<char*>(const_cast<float*>(Uw));
            const uint32_t poff4 = 4u * (uint32_t)c.L.poff;
            const uint32_t goff4 = 4u * (uint32_t)c.L.grad;
            const int voff = (int)(V - Uw);
            uint32_t iu = (uint32_t)i0;
            float accx = 0.0f, accy = 0.0f;   // sums of the x-run / y-run that is open at this step
            for (int s = 0; s < c.E; ++s) {
                const bool tu = ua <= vb;   // canonical stable order: U before V on ties
                const float q = tu ? ua : vb;
                float delta = q - qprev;
                if (lim && q > 1.0f) delta = 0.0f;
                const float g = delta * cost_slope(xa - yb, pm, c.p);
                accx += g;
                accy += g;
                qprev = q;
                const uint32_t vk = (uint32_t)(voff + D0 + s);
                const uint32_t off = 4u * (tu ? iu : (vk - iu));   // the element consumed now closes its side's run
                *reinterpret_cast<float*>(lb + off + goff4) = tu ? accx : accy;
                accx = tu ? 0.0f : accx;
                accy = tu ? accy : 0.0f;
                iu += tu ? 1u : 0u;
                const float nv = *reinterpret_cast<const float*>(lb + off + 4u);
                const float np = *reinterpret_cast<const float*>(lb + off + 4u + poff4);
                ua = tu ? nv : ua;
                xa = tu ? np : xa;
                vb = tu ? vb : nv;
                yb = tu ? yb : np;
            }
            tx = accx; ty = accy;
            ti = (int)iu;                 // slot (in Uw) of the U level that is the head when this segment ends; nw = past the end
            tj = D0 + c.E - (int)iu;      // likewise in V; m = past the end
        }
        row_sync<NW>();   // every walk is done: the CDFs may be overwritten by the tails
        if (t < c.Ga) { tail_x[t] = tx; tail_i[t] = ti; tail_y[t] = ty; tail_j[t] = tj; }
        row_sync<NW>();
        if (t < c.Ga) {
            if (t == 0 || tail_i[t - 1] != ti) {   // first thread of a run of equal tail slots: one writer per slot
                float sum = tx;
                for (int u = t + 1; u < c.Ga && tail_i[u] == ti; ++u) sum += tail_x[u];
                GUw[ti] = sum + GUw[ti];
            }
            if (t == 0 || tail_j[t - 1] != tj) {
                float sum = ty;
                for (int u = t + 1; u < c.Ga && tail_j[u] == tj; ++u) sum += tail_y[u];
                c.GV[tj] = sum + c.GV[tj];
            }
        }
        row_sync<NW>();
        if (t == 0) { c.GU[n - 1] += c.GU[n]; c.GV[m - 1] += c.GV[m]; }   // clamp of losses.py:220
        row_sync<NW>();
        if (valid) {
            const float gr = (b.grad_row ? b.grad_row[rowc * b.grad_row_stride] : 1.0f) * b.grad_scale;
            const bool x_perm = ROWPOS ? c.do_sort : !c.x_ident;
            const bool y_perm = ROWPOS ? c.do_sort : !c.y_ident;
            const int e0 = t * CPT;
            if (b.gxp) {
                float* dst = b.gxp + row * (int64_t)n;
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const int e = e0 + k;
                    if (e < n) dst[ROWPOS ? (x_perm ? ix[k] : e) : (x_perm ? a.xperm[e] : e)] = c.GU[e] * gr;
                }
            }
            if (b.gyp) {
                float* dst = b.gyp + row * (int64_t)m;
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const int e = e0 + k;
                    if (e < m) dst[ROWPOS ? (y_perm ? iy[k] : e) : (y_perm ? a.yperm[e] : e)] = -(c.GV[e] * gr);
                }
            }
        }
        if (t == 0) { U[n] = INFINITY; V[m] = INFINITY; }   // the sentinels (make_ctx sets them once) may lie under the tails
        row_sync<NW>();   // slot reads done before the next row reuses LDS
    }
}

template <int G, int CPT, bool ROWPOS>
static hipError_t launch_position_grad(const PosGradArgs& b, size_t lds, int64_t want, int block, hipStream_t s)
{
    auto kern = sot_position_grad_kernel<G, CPT, ROWPOS>;
    static GridCache cache;
    const int grid_cap = cached_resident_grid(cache, kern, block, lds);
    const int grid = balanced_grid(want, grid_cap);
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, b);
    return hipGetLastError();
}

template <bool ROWPOS>
static hipError_t dispatch_position_grad(const LaunchCfg& c, const PosGradArgs& b, size_t lds, int64_t want, int block, hipStream_t s)
{
    if (c.CPT == 16) return launch_position_grad<1024, 16, ROWPOS>(b, lds, want, block, s);
    switch (c.G) {
        case 64: return launch_position_grad<64, 8, ROWPOS>(b, lds, want, block, s);
        case 128: return launch_position_grad<128, 12, ROWPOS>(b, lds, want, block, s);
        case 256: return launch_position_grad<256, 8, ROWPOS>(b, lds, want, block, s);
        default: return launch_position_grad<1024, 8, ROWPOS>(b, lds, want, block, s);
    }
}

int run_position_grad(const sot_problem* pr, const float* grad_row, int64_t grad_row_stride, float grad_scale, float* gxp, float* gyp,
                      void* workspace, size_t workspace_bytes, void* stream)
{
    Launch l;
    int rc = setup_launch(pr, true, workspace, workspace_bytes, stream, &l);
    if (rc != SOT_OK) return rc;
    if (pr->B == 0 || (gxp == nullptr && gyp == nullptr)) return SOT_OK;
    // the per-thread tails: behind the row regions unless the rows are long enough to hold them in their dead CDFs (see the kernel)
    const int E = merge_steps(pr->n + pr->m, l.cfg.G), Ga = (pr->n + pr->m + E - 1) / E;
    const size_t lds = l.lds + ((4 * Ga <= pr->n + pr->m) ? 0 : 4 * sizeof(float) * (size_t)l.block);
    if (lds > kLdsLimit) return SOT_ERR_UNSUPPORTED_SIZE;
    PosGradArgs b{};
    b.f = l.a; b.grad_row = grad_row; b.grad_row_stride = grad_row_stride; b.grad_scale = grad_scale; b.gxp = gxp; b.gyp = gyp;
    const hipError_t e = l.rowpos ? dispatch_position_grad<true>(l.cfg, b, lds, l.want, l.block, l.s)
                                  : dispatch_position_grad<false>(l.cfg, b, lds, l.want, l.block, l.s);
    return e == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

// out[c] = sum_r rows[r * stride + c] in a fixed order (the sum over the batch that autograd attaches to a position row shared by
// every batch row, losses.py:167-170 `expand`): 16 columns per workgroup, 64 row lanes each accumulating rows r = lane, lane + 64, ...
// in fp64, then the 64 partial sums of a column in lane order.
__global__ __launch_bounds__(1024) void sot_column_sum_kernel(const float* __restrict__ rows, int64_t B, int n, int64_t stride,
                                                              float* __restrict__ out)
{
    __shared__ double part[64][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;
    double acc = 0.0;
    if (col < n)
        for (int64_t r = rl; r < B; r += 64) acc += (double)rows[r * stride + col];
    part[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && col < n) {
        double tot = 0.0;
        for (int k = 0; k < 64; ++k) tot += part[k][cl];
        out[col] = (float)tot;
    }
}

int run_column_sum(const float* rows, int64_t B, int n, int64_t stride, float* out, void* stream)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(sot_column_sum_kernel, dim3((n + 15) / 16), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), rows, B, n, stride, out);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}
#endif  // position-gradient part

#if SOT_PART & 16
// ---------------------------------------------------------------------------------------------
// Shared-position preparation (losses.py:287-288 for row-invariant positions): one workgroup per
// array checks sortedness and, if needed, sorts (position, index) pairs in LDS.
// ---------------------------------------------------------------------------------------------
// UNIT: the array is divided by its maximum first (trainer.py:196-197: `x_pos = x_pos / x_pos.max()` -- torch.max's NaN rule, IEEE
// division: the same floats as the two torch kernels), so that a training step that rebuilds its grid pays one launch, not four.
template <bool UNIT>
__global__ __launch_bounds__(1024) void sot_prepare_positions_kernel(
    const float* __restrict__ xpos, const float* __restrict__ ypos, int n, int m,
    float* __restrict__ sx, float* __restrict__ sy, int* __restrict__ px, int* __restrict__ py,
    int* __restrict__ ident)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int which = blockIdx.x;
    const float* pos = which ? ypos : xpos;
    const int len = which ? m : n;
    float* spos = which ? sy : sx;
    int* perm = which ? py : px;
    const int npad = sort16_npad(len), cap = (sort16_capacity(npad) + 3) & ~3;
    float* key = smem;
    int* idx = reinterpret_cast<int*>(smem + cap);
    int* const unsorted_flag = reinterpret_cast<int*>(smem + 2 * cap);  // all LDS is dynamic (16-B aligned carve)
    const int t = threadIdx.x, T = blockDim.x;
    if (t == 0) *unsorted_flag = 0;
    float top = 1.0f;
    if (UNIT) {   // max over the array: NaN wins (torch.max), otherwise order-free
        float* const part = reinterpret_cast<float*>(unsorted_flag + 1);   // 16 floats behind the flag
        float mx = -INFINITY;
        bool nan = false;
        for (int i = t; i < len; i += T) {
            const float v = pos[i];
            nan |= (v != v);
            mx = fmaxf(mx, v);
        }
        if (nan) mx = NAN;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float o = __shfl_xor(mx, off);
            mx = (mx != mx || o != o) ? NAN : fmaxf(mx, o);
        }
        if ((t & 63) == 0) part[t >> 6] = mx;
        __syncthreads();
        top = part[0];
        for (int w = 1; w < (T >> 6); ++w) {
            const float o = part[w];
            top = (top != top || o != o) ? NAN : fmaxf(top, o);
        }
    }
    for (int i = t; i < npad; i += T) {
        key[i] = (i < len) ? (UNIT ? pos[i] / top : pos[i]) : INFINITY;
        idx[i] = (i < len) ? i : INT_MAX;
    }
    __syncthreads();
    int bad = 0;
    for (int i = t; i + 1 < len; i += T) bad |= (key[i] > key[i + 1]);
    if (bad) *unsorted_flag = 1;
    __syncthreads();
    const bool need_sort = *unsorted_flag != 0;
    if (need_sort) {   // npad <= 16384 = 16 x 1024 threads: one block of 16 per thread
        const SortJob job{key, idx, len, npad}, none{nullptr, nullptr, 0, 0};
        merge_sort16_kv2<1>(job, none, t, T, [] { __syncthreads(); });
    }
    for (int i = t; i < len; i += T) { spos[i] = key[i]; perm[i] = need_sort ? idx[i] : i; }
    if (t == 0) ident[which] = need_sort ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------
// Batch mean (losses.py:203-211): optional hinge, fixed-order fp64 accumulation, one workgroup.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sot_reduce_mean_kernel(const float* __restrict__ row_loss, int64_t B, double denom,
                                                               int apply_hinge, float hinge, float* __restrict__ mean_out,
                                                               double* __restrict__ sum_out)
{
    __shared__ double wsum[16];
    const int t = threadIdx.x;
    double acc = 0.0;
    // Fixed summation order (independent of timing): thread t accumulates rows [8t + 8192k, 8t + 8192k + 8) for k = 0, 1, ...
    // in ascending order; the loads of one chunk are independent 16-B loads, so the loop is not a chain of L2 round trips.
    const bool vec_ok = (reinterpret_cast<uintptr_t>(row_loss) & 15) == 0;
    for (int64_t base = (int64_t)t * 8; base < B; base += (int64_t)blockDim.x * 8) {
        float v[8];
        if (vec_ok && base + 8 <= B) {
            const float4 a = *reinterpret_cast<const float4*>(row_loss + base);
            const float4 b = *reinterpret_cast<const float4*>(row_loss + base + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (base + k < B) ? row_loss[base + k] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float w = v[k];
            if (apply_hinge) w = (base + k < B) ? fmaxf(w - hinge, 0.0f) : 0.0f;
            acc += (double)w;
        }
    }
    acc = wave_sum(acc);
    if ((t & 63) == 0) wsum[t >> 6] = acc;
    __syncthreads();
    if (t == 0) {
        double tot = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += wsum[w];
        if (sum_out) *sum_out = tot;
        if (mean_out) *mean_out = (float)(tot / denom);
    }
}

// data[i] *= *scalar (the upstream gradient of a loss whose gradient sot_w1d_loss_and_grad computed ahead of the backward
// pass); nothing is touched when the scalar is exactly 1 (a plain loss.backward()).
__global__ __launch_bounds__(256) void sot_scale_inplace_kernel(float* __restrict__ data, int64_t count, const float* __restrict__ scalar)
{
    const float s = *scalar;
    if (s == 1.0f) return;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t nvec = ((reinterpret_cast<uintptr_t>(data) & 15) == 0) ? count / 4 : 0;
    float4* d4 = reinterpret_cast<float4*>(data);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        float4 v = d4[i];
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        d4[i] = v;
    }
    for (int64_t i = nvec * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) data[i] *= s;
}

// ---------------------------------------------------------------------------------------------
// Standalone segmented sort (torch.sort(keys, 1) of losses.py:287-288): one workgroup per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sot_segmented_sort_kernel(const float* __restrict__ keys, int64_t B, int n, int64_t stride,
                                                                 float* __restrict__ out_keys, int64_t* __restrict__ out_idx)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int npad = sort16_npad(n), cap = (sort16_capacity(npad) + 3) & ~3;
    float* key = smem;
    int* idx = reinterpret_cast<int*>(smem + cap);
    const int t = threadIdx.x, T = blockDim.x;
    const SortJob job{key, idx, n, npad}, none{nullptr, nullptr, 0, 0};
    for (int64_t row = blockIdx.x; row < B; row += gridDim.x) {
        const float* src = keys + row * stride;
        for (int i = t; i < npad; i += T) key[i] = (i < n) ? src[i] : INFINITY;
        __syncthreads();
        merge_sort16_kv2<1>(job, none, t, T, [] { __syncthreads(); });   // the launcher sizes the block so that npad <= 16 T
        for (int i = t; i < n; i += T) {
            if (out_keys) out_keys[row * (int64_t)n + i] = key[i];
            if (out_idx) out_idx[row * (int64_t)n + i] = (int64_t)min(idx[i], n - 1);   // NaN keys: a pad's index never leaves the kernel
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// The same sort, ONE WAVEFRONT per row (round 6; rows of <= 2048 keys): sot_wave_sort.hpp -- a payload-free network on one 32-bit
// word per key in registers, no workgroup barrier; a row the fast path declines (clustered / non-finite keys) is sorted by the same
// wavefront with the merge sort.  Four independent rows per 256-thread workgroup; the next row's keys are fetched while the
// current one is stored.  FAST: n == 64 KPL, rows and outputs 16-byte aligned -- no validity tests, 16-byte loads and stores.
// ---------------------------------------------------------------------------------------------
// LDS dwords of a row's key array (natural keys; the merge-sort fallback's skewed image) and of its index / scratch array (either sort's)
template <int KPL>
__host__ __device__ constexpr int wave_sort_key_cap() { return align4(sort16_capacity(sort16_npad(64 * KPL))); }
template <int KPL>
__host__ __device__ constexpr int wave_sort_idx_cap() { return align4(imax(sort16_capacity(sort16_npad(64 * KPL)), wave_sort_scratch(KPL))); }
// waves per workgroup (the segmented sort runs the network: 17.4 KB per wave, two workgroups per CU; its 16 B per key of traffic bound it)
template <int KPL>
__host__ __device__ constexpr int wave_sort_wg_waves() { return 4; }

template <int KPL, bool FAST>
__global__ __launch_bounds__(64 * wave_sort_wg_waves<KPL>(), 2) void sot_segmented_sort_wave_kernel(const float* __restrict__ keys, int64_t B, int n, int64_t stride,
                                                                         float* __restrict__ out_keys, int64_t* __restrict__ out_idx)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NPAD = 64 * KPL, KCAP = wave_sort_key_cap<KPL>(), ICAP = wave_sort_idx_cap<KPL>(), WGW = wave_sort_wg_waves<KPL>();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* key = smem + wv * (KCAP + ICAP);
    uint32_t* idx = reinterpret_cast<uint32_t*>(key + KCAP);
    const int64_t step = (int64_t)gridDim.x * WGW;
    int64_t row = (int64_t)blockIdx.x * WGW + wv;
    float x[KPL];
    auto fetch = [&](int64_t rw) {
        const float* src = keys + rw * stride;
        if constexpr (FAST) {
#pragma unroll
            for (int r = 0; r < KPL; r += 4) {
                const float4 v = *reinterpret_cast<const float4*>(src + wsort_elem<true>(r, lane));
                x[r] = v.x; x[r + 1] = v.y; x[r + 2] = v.z; x[r + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int r = 0; r < KPL; ++r) { const int e = r * 64 + lane; const float v = src[min(e, n - 1)]; x[r] = (e < n) ? v : INFINITY; }
        }
    };
    if (row < B) fetch(row);
    for (; row < B; row += step) {
        if constexpr (FAST) {
#pragma unroll
            for (int r = 0; r < KPL; r += 4) *reinterpret_cast<float4*>(key + wsort_elem<true>(r, lane)) = make_float4(x[r], x[r + 1], x[r + 2], x[r + 3]);
        } else {
#pragma unroll
            for (int r = 0; r < KPL; ++r) key[r * 64 + lane] = x[r];
        }
        row_sync<1>();
        float sk[KPL]; uint32_t si[KPL];
        const bool fast = wave_sort_kv<KPL, false, FAST, FAST>(x, key, idx, n, lane, sk, si);
        if (!fast) {   // clustered / non-finite keys: the merge sort, by this wavefront alone (nothing of the fast path is live across it)
            constexpr int MAXB = (NPAD / 16 + 63) / 64;
            const SortJob job{key, reinterpret_cast<int*>(idx), n, sort16_npad(n)}, none{nullptr, nullptr, 0, 0};
            row_sync<1>();
            merge_sort16_kv2<MAXB>(job, none, lane, 64, [] { row_sync<1>(); });
            row_sync<1>();
#pragma unroll
            for (int r = 0; r < KPL; ++r) { sk[r] = key[wsort_elem<FAST>(r, lane)]; si[r] = idx[wsort_elem<FAST>(r, lane)]; }
        }
        if (row + step < B) fetch(row + step);   // the next row's keys travel while this one is stored
        if constexpr (FAST) {
            float* ok_row = out_keys ? out_keys + row * (int64_t)n : nullptr;
            int64_t* oi_row = out_idx ? out_idx + row * (int64_t)n : nullptr;
#pragma unroll
            for (int r = 0; r < KPL; r += 4) {
                const int e = wsort_elem<true>(r, lane);
                if (ok_row) *reinterpret_cast<float4*>(ok_row + e) = make_float4(sk[r], sk[r + 1], sk[r + 2], sk[r + 3]);
                if (oi_row) {   // (NaN keys: a pad's index never leaves the kernel)
                    typedef long long ll2 __attribute__((ext_vector_type(2)));
                    ll2 a, b;
                    a.x = min((int)si[r], n - 1); a.y = min((int)si[r + 1], n - 1); b.x = min((int)si[r + 2], n - 1); b.y = min((int)si[r + 3], n - 1);
                    *reinterpret_cast<ll2*>(oi_row + e) = a;
                    *reinterpret_cast<ll2*>(oi_row + e + 2) = b;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < KPL; ++r) {
                const int e = r * 64 + lane;
                if (e < n) {
                    if (out_keys) out_keys[row * (int64_t)n + e] = sk[r];
                    if (out_idx) out_idx[row * (int64_t)n + e] = (int64_t)min((int)si[r], n - 1);
                }
            }
        }
        row_sync<1>();
    }
}

template <int KPL>
int launch_segmented_sort_wave(const float* keys, int64_t B, int n, int64_t stride, float* out_keys, int64_t* out_idx, hipStream_t s)
{
    constexpr int WGW = wave_sort_wg_waves<KPL>();
    constexpr size_t lds = (size_t)(wave_sort_key_cap<KPL>() + wave_sort_idx_cap<KPL>()) * 4 * WGW;   // waves x (key | idx)
    constexpr bool CAN_VEC = KPL % 4 == 0;
    const bool fast = CAN_VEC && n == 64 * KPL && stride % 4 == 0 && ((reinterpret_cast<uintptr_t>(keys) | reinterpret_cast<uintptr_t>(out_keys) |
                                                                     reinterpret_cast<uintptr_t>(out_idx)) & 15) == 0;
    static GridCache cache[2];
    const void* fn = fast ? reinterpret_cast<const void*>(sot_segmented_sort_wave_kernel<KPL, CAN_VEC>)
                          : reinterpret_cast<const void*>(sot_segmented_sort_wave_kernel<KPL, false>);
    allow_full_lds_once(cache[fast ? 1 : 0], fn);
    int per_cu = (int)(kLdsLimit / lds);
    if (per_cu > 16 / WGW) per_cu = 16 / WGW;
    if (per_cu < 1) per_cu = 1;
    const int64_t groups = (B + WGW - 1) / WGW, cap = (int64_t)device_cu_count() * per_cu;
    const int grid = (int)(groups < cap ? groups : cap);
    (void)hipGetLastError();  // do not inherit a stale error from earlier runtime calls
    if (fast) hipLaunchKernelGGL((sot_segmented_sort_wave_kernel<KPL, CAN_VEC>), dim3(grid), dim3(64 * WGW), lds, s, keys, B, n, stride, out_keys, out_idx);
    else hipLaunchKernelGGL((sot_segmented_sort_wave_kernel<KPL, false>), dim3(grid), dim3(64 * WGW), lds, s, keys, B, n, stride, out_keys, out_idx);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------
// Per-row positions, sorted AHEAD of the row kernels (round 6): one wavefront per row sorts the row's two position arrays with the wave
// sort (sot_wave_sort.hpp) and leaves their permutations in the [B, n + m] uint16 image the row kernels gather through (row_perm_in).
// Inside the row kernels the same network cost their other phases the registers (loop invariants spilled on paths that never sort:
// sorted rows 71 -> 90 us, backward 276 -> 348 us at 4096 x 2048, measured inlined and as a call); a kernel of its own has its own
// allocation, every wavefront of it sorts, and it needs 8.5 KB of LDS per wavefront (the transposition image; the run repair reads the
// full keys from global memory).
// An array that arrives sorted gets the identity; an array the wave sort declines (clustered / non-finite positions) is sorted by the same wavefront with
// the stable merge sort of round 4 (17.4 KB of LDS per wave are provisioned for it: the kernel's 8 waves per CU leave the room): the image is always complete.
// ---------------------------------------------------------------------------------------------

// The stable merge sort of ONE array by ONE wavefront (what a declined wave sort falls back to): a call, not inlined -- its 64 result registers stay
// out of the pre-sort kernel's hot path.  key: natural keys in LDS with +inf behind the len real ones up to sort16_npad(len); idx: scratch in, indices out.
template <int MAXB>
static __device__ __attribute__((noinline)) void wave_merge_sort_array(float* key, int* idx, int len, int lane)
{
    const SortJob job{key, idx, len, sort16_npad(len)}, none{nullptr, nullptr, 0, 0};
    row_sync<1>();
    merge_sort16_kv2<MAXB>(job, none, lane, kWave, [] { row_sync<1>(); });
    row_sync<1>();
}
#ifndef SOT_ROWPOS_SORT_MAX_WG
#define SOT_ROWPOS_SORT_MAX_WG 4
#endif
#ifndef SOT_ROWPOS_SORT_WAVES
#define SOT_ROWPOS_SORT_WAVES 2   /* wavefronts per SIMD the pre-sort kernel is compiled for (LDS: 16.9 KB per wave of the 32-keys-per-lane form = 8 waves per CU) */
#endif

// VEC: n, m multiples of 4, position rows 16-byte aligned, permutation rows 8-byte aligned (16-byte loads, 8-byte stores); FULL: n == m == 64 KPL
template <int KPL, bool FULL, bool VEC>
__global__ __launch_bounds__(256, SOT_ROWPOS_SORT_WAVES) void sot_rowpos_sort_kernel(const float* __restrict__ xpos, const float* __restrict__ ypos, int64_t B, int n, int m,
                                                                 int64_t xps, int64_t yps, uint16_t* __restrict__ perm)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // per wave: [idx / scratch | key]: the wave sort needs the first only (8.25 KB at 32 keys per lane); the merge-sort fallback both (17.4 KB) -- at the 8 waves
    // per CU the kernel's registers allow anyway, the CU holds that
    constexpr int ICAP = wave_sort_idx_cap<KPL>(), KCAP = wave_sort_key_cap<KPL>();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* const idx = reinterpret_cast<uint32_t*>(smem) + wv * (ICAP + KCAP);
    float* const keyl = reinterpret_cast<float*>(idx + ICAP);
    // one task = ONE array (task 2 r: the x positions of row r, 2 r + 1: its y positions): 2 B tasks spread evenly over whatever number of waves is
    // resident; an array's outcome is its own business -- sorted on arrival: the identity, sorted here (wave sort, or the merge sort when that declines): its permutation
    for (int64_t task = (int64_t)blockIdx.x * 4 + wv; task < 2 * B; task += (int64_t)gridDim.x * 4) {
        const int64_t row = task >> 1;
        const int which = (int)(task & 1);
        {
            const float* src = which ? ypos + row * yps : xpos + row * xps;
            const int len = which ? m : n;
            uint16_t* const dst = perm + row * ((int64_t)n + m) + (which ? n : 0);
            float x[KPL];
            if constexpr (VEC) {
#pragma unroll
                for (int r = 0; r < KPL; r += 4) {
                    const int e = wsort_elem<true>(r, lane);
                    const float4 v = *reinterpret_cast<const float4*>(src + (FULL ? e : min(e, len - 4)));
                    const bool real = FULL || e < len;   // whole groups of four: len % 4 == 0
                    x[r] = real ? v.x : INFINITY; x[r + 1] = real ? v.y : INFINITY; x[r + 2] = real ? v.z : INFINITY; x[r + 3] = real ? v.w : INFINITY;
                }
            } else {
#pragma unroll
                for (int r = 0; r < KPL; ++r) { const int e = r * 64 + lane; const float v = src[min(e, len - 1)]; x[r] = (e < len) ? v : INFINITY; }
            }
            // sortedness (as the row kernels test it: an element greater than its right neighbour; +inf behind the last one)
            bool unsorted = false;
            if constexpr (VEC) {   // (pads are +inf: never greater than their right neighbour)
#pragma unroll
                for (int g4 = 0; g4 < KPL; g4 += 4) {
                    // the element after this lane's four: the next lane's first; for lane 63 the first of the next block of 256 (lane 0)
                    float nxt = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[g4]), 0x130 /* wave_shl:1 */, 0xF, 0xF, false));
                    const float wrap = (g4 + 4 < KPL) ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x[g4 + 4 < KPL ? g4 + 4 : 0]))) : INFINITY;
                    nxt = (lane == 63) ? wrap : nxt;
                    unsorted |= (x[g4] > x[g4 + 1]) | (x[g4 + 1] > x[g4 + 2]) | (x[g4 + 2] > x[g4 + 3]) | (x[g4 + 3] > nxt);
                }
            } else {
#pragma unroll
                for (int r = 0; r < KPL; ++r) {
                    float nxt = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[r]), 0x130 /* wave_shl:1 */, 0xF, 0xF, false));
                    const float wrap = (r + 1 < KPL) ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x[r + 1 < KPL ? r + 1 : 0]))) : INFINITY;
                    nxt = (lane == 63) ? wrap : nxt;
                    unsorted |= (r * 64 + lane + 1 < len) && (x[r] > nxt);
                }
            }
            float sk[KPL]; uint32_t si[KPL];
            if (__builtin_amdgcn_ballot_w64(unsorted) == 0ull) {      // sorted on arrival: the identity
#pragma unroll
                for (int r = 0; r < KPL; ++r) si[r] = (uint32_t)wsort_elem<VEC>(r, lane);
            } else {
                const bool done = wave_sort_core<KPL, false, FULL, VEC, false, true>(x, [src](uint32_t i) { return src[i]; }, nullptr, idx, len, lane, sk, si);
                row_sync<1>();   // the scratch image is free again
                if (!done) {     // declined (clustered / non-finite positions; wave-uniform): the stable merge sort of round 4, by this wavefront alone
                    constexpr int NP = 64 * KPL;
#pragma unroll
                    for (int r = 0; r < KPL; ++r) keyl[wsort_elem<VEC>(r, lane)] = x[r];          // natural order; +inf behind the row (x[] holds the pads)
                    for (int e = NP + lane; e < sort16_npad(len); e += 64) keyl[e] = INFINITY;    // (never: sort16_npad(len) <= 64 KPL)
                    wave_merge_sort_array<(NP / 16 + 63) / 64>(keyl, reinterpret_cast<int*>(idx), len, lane);
#pragma unroll
                    for (int r = 0; r < KPL; ++r) si[r] = (uint32_t)min(reinterpret_cast<int*>(idx)[wsort_elem<VEC>(r, lane)], len - 1);   // (a pad's INT_MAX / NaN rows: never outside the row)
                    row_sync<1>();
                }
            }
            if constexpr (VEC) {
#pragma unroll
                for (int r = 0; r < KPL; r += 4) {
                    uint2 pk;
                    pk.x = si[r] | (si[r + 1] << 16); pk.y = si[r + 2] | (si[r + 3] << 16);
                    const int e = wsort_elem<true>(r, lane);
                    if (FULL || e < len) *reinterpret_cast<uint2*>(dst + e) = pk;   // (positions < len never hold a pad: no clamp)
                }
            } else {
#pragma unroll
                for (int r = 0; r < KPL; ++r) { const int e = r * 64 + lane; if (e < len) dst[e] = (uint16_t)min((int)si[r], len - 1); }
            }
        }
    }
}

// dest: [B, n + m] uint16.  n, m in [2, 2048].
int launch_rowpos_sort(const float* xpos, const float* ypos, int64_t B, int n, int m, int64_t xps, int64_t yps, uint16_t* dest, hipStream_t s)
{
    const int N = n > m ? n : m;
    const bool aligned = ((reinterpret_cast<uintptr_t>(xpos) | reinterpret_cast<uintptr_t>(ypos)) & 15) == 0 && (xps & 3) == 0 && (yps & 3) == 0 &&
                         (reinterpret_cast<uintptr_t>(dest) & 7) == 0 && (n & 3) == 0 && (m & 3) == 0;
    (void)hipGetLastError();  // do not inherit a stale error from earlier runtime calls
    auto go = [&](auto kern, int kpl) {
        const size_t lds = (size_t)(sot::align4(sot::imax(sot::sort16_capacity(sot::sort16_npad(64 * kpl)), wave_sort_scratch(kpl, true))) + sot::align4(sot::sort16_capacity(sot::sort16_npad(64 * kpl)))) * 4 * 4;   // 4 waves x (idx / scratch | key)
        static GridCache cache;   // (one per lambda instantiation, i.e. per kernel)
        allow_full_lds_once(cache, reinterpret_cast<const void*>(kern));
        int per_cu = (int)(kLdsLimit / lds);
        if (per_cu > SOT_ROWPOS_SORT_MAX_WG) per_cu = SOT_ROWPOS_SORT_MAX_WG;   // four-wave workgroups: one wave of each per SIMD, at most 16 waves per CU
        const int64_t groups = (2 * B + 3) / 4, cap = (int64_t)device_cu_count() * per_cu;   // one wave per array
        hipLaunchKernelGGL(kern, dim3((unsigned)(groups < cap ? groups : cap)), dim3(256), lds, s, xpos, ypos, B, n, m, xps, yps, dest);
    };
    auto pick = [&](auto kpl_tag) {
        constexpr int KP = decltype(kpl_tag)::value;
        if (aligned && n == 64 * KP && m == 64 * KP) go(sot_rowpos_sort_kernel<KP, true, true>, KP);
        else if (aligned) go(sot_rowpos_sort_kernel<KP, false, true>, KP);
        else go(sot_rowpos_sort_kernel<KP, false, false>, KP);
    };
    if (N <= 512) pick(std::integral_constant<int, 8>{});
    else if (N <= 1024) pick(std::integral_constant<int, 16>{});
    else pick(std::integral_constant<int, 32>{});
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

int launch_prepare(const float* xpos, const float* ypos, int n, int m, float* sx, float* sy, int* px, int* py, int* ident,
                          hipStream_t s, bool unit)
{
    const int npad = sort16_npad(n > m ? n : m);
    if (npad > 16 * 1024) return SOT_ERR_UNSUPPORTED_SIZE;
    const size_t prep_lds = (size_t)((sort16_capacity(npad) + 3) & ~3) * 8 + 16 + 64;   // + the flag's slot + 16 partial maxima
    if (prep_lds > kLdsLimit) return SOT_ERR_UNSUPPORTED_SIZE;
    static GridCache cache, cache_unit;
    (void)hipGetLastError();  // do not inherit a stale error from earlier runtime calls
    if (unit) {
        allow_full_lds_once(cache_unit, reinterpret_cast<const void*>(sot_prepare_positions_kernel<true>));
        hipLaunchKernelGGL(sot_prepare_positions_kernel<true>, dim3(2), dim3(1024), prep_lds, s, xpos, ypos, n, m, sx, sy, px, py, ident);
    } else {
        allow_full_lds_once(cache, reinterpret_cast<const void*>(sot_prepare_positions_kernel<false>));
        hipLaunchKernelGGL(sot_prepare_positions_kernel<false>, dim3(2), dim3(1024), prep_lds, s, xpos, ypos, n, m, sx, sy, px, py, ident);
    }
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

// validation, config choice, optional position preparation, persistent-grid sizing
int setup_launch(const sot_problem* pr, bool with_grad, void* workspace, size_t workspace_bytes, void* stream, Launch* out)
{
    int rc = validate(pr);
    if (rc != SOT_OK) return rc;
    Launch& l = *out;
    l.s = reinterpret_cast<hipStream_t>(stream);
    l.rowpos = pr->xpos_row_stride != 0;
    const bool need_prep = !l.rowpos && (pr->flags & SOT_FLAG_REQUIRE_SORT);
    const int n = pr->n, m = pr->m;
    int rpw = 1;
    if (!pick_cfg(n, m, l.rowpos, with_grad, &l.cfg, &l.lds, &l.block, &rpw)) return SOT_ERR_UNSUPPORTED_SIZE;

    FwdArgs a{};
    a.x = pr->x; a.y = pr->y; a.xpos = pr->xpos; a.ypos = pr->ypos;
    a.B = pr->B; a.n = n; a.m = m;
    a.xs = pr->x_row_stride; a.ys = pr->y_row_stride; a.xps = pr->xpos_row_stride; a.yps = pr->ypos_row_stride;
    a.p = pr->p; a.flags = pr->flags;
    if (l.rowpos && (pr->flags & SOT_FLAG_REQUIRE_SORT)) { a.perm_out = pr->row_perm_out; a.perm_in = pr->row_perm_in; }
    // per-row positions nobody has sorted yet: the wave-sort kernel runs first (round 6) and the row kernel gathers through its
    // permutations -- into the caller's row_perm_out, else into the workspace when it is large enough (sot_workspace_bytes), else the
    // row kernel sorts in LDS as before
    if (SOT_WAVE_SORT && l.rowpos && (pr->flags & SOT_FLAG_REQUIRE_SORT) && a.perm_in == nullptr && pr->B > 0 && n >= 2 && m >= 2 && n <= 2048 &&
        m <= 2048 && !(pr->flags & SOT_FLAG_NO_SPECIALIZE)) {
        uint16_t* dest = pr->row_perm_out;
        if (dest == nullptr && workspace != nullptr && workspace_bytes >= rowpos_perm_bytes(pr->B, n, m)) dest = reinterpret_cast<uint16_t*>(workspace);
        if (dest != nullptr) {
            rc = launch_rowpos_sort(pr->xpos, pr->ypos, pr->B, n, m, pr->xpos_row_stride, pr->ypos_row_stride, dest, l.s);
            if (rc != SOT_OK) return rc;
            a.perm_in = dest; a.perm_out = nullptr;   // the image is complete: the row kernel only reads it
        }
    }

    if (need_prep && pr->perm_is_identity != nullptr) {  // caller-provided plan: positions are already sorted
        a.xperm = pr->xperm; a.yperm = pr->yperm; a.ident = pr->perm_is_identity;
    } else if (need_prep && pr->B > 0) {
        const WsLayout w = ws_layout(n, m);
        if (workspace == nullptr) return SOT_ERR_NULL_POINTER;
        if (workspace_bytes < w.total) return SOT_ERR_WORKSPACE;
        char* ws = reinterpret_cast<char*>(workspace);
        float* sx = reinterpret_cast<float*>(ws + w.sx);
        float* sy = reinterpret_cast<float*>(ws + w.sy);
        int* px = reinterpret_cast<int*>(ws + w.px);
        int* py = reinterpret_cast<int*>(ws + w.py);
        int* ident = reinterpret_cast<int*>(ws + w.ident);
        rc = launch_prepare(pr->xpos, pr->ypos, n, m, sx, sy, px, py, ident, l.s);
        if (rc != SOT_OK) return rc;
        a.xpos = sx; a.ypos = sy; a.xperm = px; a.yperm = py; a.ident = ident;
    }
    l.a = a;

    l.want = (pr->B + rpw - 1) / rpw;
    l.pm = (pr->p == 1.0f) ? 1 : ((pr->p == 2.0f) ? 2 : 0);
    // 16-B-per-lane loads need 16-B aligned rows of whole float4s
    l.vec = ((n & 3) == 0) && ((m & 3) == 0) && ((pr->x_row_stride & 3) == 0) && ((pr->y_row_stride & 3) == 0) &&
            ((reinterpret_cast<uintptr_t>(pr->x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(pr->y) & 15) == 0);
    return SOT_OK;
}

int run_forward(const sot_problem* pr, float* row_loss, float* uq, float* vq, float* Q, float* U, float* V,
                       bool quant, void* workspace, size_t workspace_bytes, void* stream, const MeanTail* mean_tail)
{
    Launch l;
    int rc = setup_launch(pr, false, workspace, workspace_bytes, stream, &l);
    if (rc != SOT_OK) return rc;
    if (pr->B == 0) return SOT_OK;
    l.a.row_loss = row_loss;
    if (mean_tail != nullptr && row_loss != nullptr) l.a.mt = *mean_tail;  // the batch mean comes out of this launch's last workgroup
    l.a.oUq = uq; l.a.oVq = vq; l.a.oQ = Q; l.a.oU = U; l.a.oV = V;
    // row lengths with a compile-time kernel (forward_full_supports: powers of two 512 ... 8192 and n_fft/2 + 1) take it
    bool full = !l.rowpos && !quant && pr->n == pr->m && forward_full_supports(pr->n, l.vec) &&
                !(pr->flags & (SOT_FLAG_PRENORMALIZED | SOT_FLAG_NO_SPECIALIZE));
#if defined(SOT_STUB_MISSING_PARTS) && !(SOT_PART & 128)
    full = false;
#endif
    // every other row length up to 8192 on shared positions: the same kernels with the length at run time (full_rt_capacity)
    bool full_rt = !full && !l.rowpos && !quant && pr->n == pr->m && full_rt_capacity(pr->n) != 0 &&
                   !(pr->flags & (SOT_FLAG_PRENORMALIZED | SOT_FLAG_NO_SPECIALIZE));
#if defined(SOT_STUB_MISSING_PARTS) && !(SOT_PART & 512)
    full_rt = false;
#endif
    // per-row positions with their permutations at hand (handed in, or just written by the pre-sort kernel), 2048- / 1024- / 512-point rows: the
    // compile-time-length kernel with a position copy of its own per row (sot_forward_full.inc: RP)
    bool full_rp = l.rowpos && l.a.perm_in != nullptr && !quant && (pr->n == 2048 || pr->n == 1024 || pr->n == 512) && pr->m == pr->n && l.vec && (pr->flags & SOT_FLAG_REQUIRE_SORT) &&
                   (reinterpret_cast<uintptr_t>(l.a.perm_in) & 15) == 0 && !(pr->flags & (SOT_FLAG_PRENORMALIZED | SOT_FLAG_NO_SPECIALIZE));
#if defined(SOT_STUB_MISSING_PARTS) && !(SOT_PART & 128)
    full_rp = false;
#endif
    if (full_rp) return dispatch_forward_full_rowpos(l.pm, l.a, l.s) == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
    // p = 1 on one grid shared by both measures, no cutoff: the merge-free kernel (sot_forward_full.inc: sot_area_full_kernel)
    const bool area = (full || full_rt) && l.pm == 1 && (pr->flags & SOT_FLAG_SAME_GRID) && !(pr->flags & (SOT_FLAG_LIMIT_Q | SOT_FLAG_NO_AREA));
    const hipError_t e = area       ? (full ? dispatch_area_full(l.a, l.s) : dispatch_area_full_rt(l.a, l.s))
                         : full     ? dispatch_forward_full(l.cfg, l.pm, l.a, l.lds, l.want, l.block, l.s)
                         : full_rt  ? dispatch_forward_full_rt(l.pm, l.a, l.s)
                         : l.rowpos ? dispatch_forward<true>(l.cfg, quant, l.pm, l.vec, l.a, l.lds, l.want, l.block, l.s)
                                    : dispatch_forward<false>(l.cfg, quant, l.pm, l.vec, l.a, l.lds, l.want, l.block, l.s);
    return e == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

// row_loss_out (loss-and-gradient form): the kernels that can also emit the row losses do so and *fused is set; the caller
// runs the forward kernel otherwise.  grad_row == nullptr: an upstream gradient of 1 for every row.
int run_backward(const sot_problem* pr, const float* grad_row, int64_t grad_row_stride, float grad_scale, float* gx,
                        float* gy, void* workspace, size_t workspace_bytes, void* stream, float* row_loss_out, bool* fused,
                        const MeanTail* mean_tail)
{
    Launch l;
    int rc = setup_launch(pr, true, workspace, workspace_bytes, stream, &l);
    if (rc != SOT_OK) return rc;
    if (fused) *fused = false;
    if (pr->B == 0 || (gx == nullptr && gy == nullptr)) return SOT_OK;
    BwdArgs b{};
    b.f = l.a; b.grad_row = grad_row; b.grad_row_stride = grad_row_stride; b.grad_scale = grad_scale; b.gx = gx; b.gy = gy;
    // row lengths with a compile-time kernel take it (as in run_forward)
    const bool aligned16 = l.vec && (gx == nullptr || (reinterpret_cast<uintptr_t>(gx) & 15) == 0) &&
                           (gy == nullptr || (reinterpret_cast<uintptr_t>(gy) & 15) == 0);
    bool full = !l.rowpos && pr->n == pr->m && backward_full_supports(pr->n, aligned16) &&
                !(pr->flags & (SOT_FLAG_PRENORMALIZED | SOT_FLAG_NO_SPECIALIZE));
#if defined(SOT_STUB_MISSING_PARTS) && !(SOT_PART & 128)
    full = false;
#endif
    bool full_rt = !full && !l.rowpos && pr->n == pr->m && full_rt_capacity(pr->n) != 0 && full_rt_capacity(pr->n) <= 4096 &&
                   !(pr->flags & (SOT_FLAG_PRENORMALIZED | SOT_FLAG_NO_SPECIALIZE));
#if defined(SOT_STUB_MISSING_PARTS) && !(SOT_PART & 1024)
    full_rt = false;
#endif
    // per-row positions with their permutations at hand, 2048- / 512-point rows: the compile-time-length kernel with a position copy of its own per row
    bool full_rp = l.rowpos && b.f.perm_in != nullptr && (pr->n == 2048 || pr->n == 512) && pr->m == pr->n && l.vec && (pr->flags & SOT_FLAG_REQUIRE_SORT) &&
                   (reinterpret_cast<uintptr_t>(b.f.perm_in) & 15) == 0 && !(pr->flags & (SOT_FLAG_PRENORMALIZED | SOT_FLAG_NO_SPECIALIZE));
#if defined(SOT_STUB_MISSING_PARTS) && !(SOT_PART & 128)
    full_rp = false;
#endif
    if ((full || full_rt || full_rp) && gx == nullptr && row_loss_out != nullptr) {   // the y-only full-row kernel accumulates the loss on its walk
        b.f.row_loss = row_loss_out;
        if (mean_tail != nullptr) b.f.mt = *mean_tail;
        if (fused) *fused = true;
    }
    // p = 1 on one grid, gradient w.r.t. y alone, no cutoff: the merge-free training form (sot_forward_full.inc: sot_area_train_kernel)
    // -- OPT-IN (SOT_FLAG_TIE_FREE_GRADIENT): at exactly tied levels it returns the derivative in the CDF values, not the reference's
    // float32 tie-order artefact
    const bool area = full && gx == nullptr && l.pm == 1 && area_train_supports(pr->n) && (pr->flags & SOT_FLAG_SAME_GRID) && (pr->flags & SOT_FLAG_TIE_FREE_GRADIENT) &&
                      !(pr->flags & (SOT_FLAG_LIMIT_Q | SOT_FLAG_NO_AREA));
    const hipError_t e = area       ? dispatch_area_train(b, l.s)
                         : full_rp  ? dispatch_backward_full_rowpos(l.pm, b, l.s)
                         : full     ? dispatch_backward_full(l.cfg, l.pm, b, l.s)
                         : full_rt  ? dispatch_backward_full_rt(l.pm, b, l.s)
                         : l.rowpos ? dispatch_backward<true>(l.cfg, l.pm, l.vec, b, l.lds, l.want, l.block, l.s)
                                    : dispatch_backward<false>(l.cfg, l.pm, l.vec, b, l.lds, l.want, l.block, l.s);
    return e == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

#endif  // misc part


#if SOT_PART & 32
// ---- CSR (ragged) forward: BASELINE config 4's second input form ---------------------------------------------
template <int G, int CPT, int PM, bool LIM>
static hipError_t launch_forward_csr(const FwdArgs& a, size_t lds, int64_t want, int block, hipStream_t s)
{
    auto kern = sot_forward_kernel<G, CPT, true, false, PM, LIM, false, true>;
    static GridCache cache;  // per instantiation (function-local static: thread-safe initialisation)
    const int grid_cap = cached_resident_grid(cache, kern, block, lds);
    const int grid = balanced_grid(want, grid_cap);
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, a);
    return hipGetLastError();
}

template <int G, int CPT>
static hipError_t dispatch_forward_csr_g(int pm, bool lim, const FwdArgs& a, size_t lds, int64_t want, int block, hipStream_t s)
{
    if (lim) {
        switch (pm) {
            case 1: return launch_forward_csr<G, CPT, 1, true>(a, lds, want, block, s);
            case 2: return launch_forward_csr<G, CPT, 2, true>(a, lds, want, block, s);
            default: return launch_forward_csr<G, CPT, 0, true>(a, lds, want, block, s);
        }
    }
    switch (pm) {
        case 1: return launch_forward_csr<G, CPT, 1, false>(a, lds, want, block, s);
        case 2: return launch_forward_csr<G, CPT, 2, false>(a, lds, want, block, s);
        default: return launch_forward_csr<G, CPT, 0, false>(a, lds, want, block, s);
    }
}

int run_forward_csr(const float* xw, const float* xp, const int64_t* xoff, int64_t x_nnz, const float* yw, const float* yp,
                    const int64_t* yoff, int64_t y_nnz, int64_t B, int max_n, int max_m, float p, uint32_t flags, float* row_loss,
                    void* stream)
{
    if (!(p >= 1.0f)) return SOT_ERR_INVALID_P;
    if (B < 0 || max_n < 1 || max_m < 1 || x_nnz < 1 || y_nnz < 1) return SOT_ERR_BAD_SHAPE;
    if (B == 0) return SOT_OK;
    if (!xw || !xp || !xoff || !yw || !yp || !yoff || !row_loss) return SOT_ERR_NULL_POINTER;
    LaunchCfg cfg; size_t lds = 0; int block = 0, rpw = 1;
    if (!pick_cfg(max_n, max_m, true, false, &cfg, &lds, &block, &rpw)) return SOT_ERR_UNSUPPORTED_SIZE;
    FwdArgs a{};
    a.x = xw; a.y = yw; a.xpos = xp; a.ypos = yp; a.xoff = xoff; a.yoff = yoff;
    a.B = B; a.n = max_n; a.m = max_m;
    a.p = p; a.flags = flags; a.row_loss = row_loss;
    const int pm = (p == 1.0f) ? 1 : ((p == 2.0f) ? 2 : 0);
    const bool lim = flags & SOT_FLAG_LIMIT_Q;
    const int64_t want = (B + rpw - 1) / rpw;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e;
    if (cfg.CPT == 16) e = dispatch_forward_csr_g<1024, 16>(pm, lim, a, lds, want, block, s);
    else if (cfg.G == 64) e = dispatch_forward_csr_g<64, 8>(pm, lim, a, lds, want, block, s);
    else if (cfg.G == 128) e = dispatch_forward_csr_g<128, 12>(pm, lim, a, lds, want, block, s);
    else if (cfg.G == 256) e = dispatch_forward_csr_g<256, 8>(pm, lim, a, lds, want, block, s);
    else e = dispatch_forward_csr_g<1024, 8>(pm, lim, a, lds, want, block, s);
    return e == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}
#endif  // CSR part

}  // namespace sot

// =============================================================================================
// C ABI (include/sot_hip.h)
// =============================================================================================
#if SOT_PART & 16
namespace sot {
constexpr int kProfileSlots = 64;
static hipEvent_t g_prof_start[kProfileSlots], g_prof_stop[kProfileSlots];
static bool g_prof_made[kProfileSlots];
static std::mutex g_prof_mu;
static thread_local int t_prof_armed = -1;   // slot the calling thread armed for its next full-row launch

bool profile_take(hipEvent_t* start, hipEvent_t* stop)
{
    const int slot = t_prof_armed;
    if (slot < 0) return false;
    t_prof_armed = -1;
    *start = g_prof_start[slot]; *stop = g_prof_stop[slot];
    return true;
}
}  // namespace sot

extern "C" {

int sot_profile_next_launch(int slot)
{
    if (slot < 0 || slot >= sot::kProfileSlots) return SOT_ERR_BAD_SHAPE;
    {
        std::lock_guard<std::mutex> lock(sot::g_prof_mu);
        if (!sot::g_prof_made[slot]) {
            if (hipEventCreate(&sot::g_prof_start[slot]) != hipSuccess || hipEventCreate(&sot::g_prof_stop[slot]) != hipSuccess) {
                (void)hipGetLastError();
                return SOT_ERR_LAUNCH;
            }
            sot::g_prof_made[slot] = true;
        }
    }
    sot::t_prof_armed = slot;
    return SOT_OK;
}

int sot_profile_elapsed_ms(int slot, float* ms)
{
    if (slot < 0 || slot >= sot::kProfileSlots || ms == nullptr || !sot::g_prof_made[slot]) return SOT_ERR_BAD_SHAPE;
    if (hipEventSynchronize(sot::g_prof_stop[slot]) != hipSuccess || hipEventElapsedTime(ms, sot::g_prof_start[slot], sot::g_prof_stop[slot]) != hipSuccess) {
        (void)hipGetLastError();
        return SOT_ERR_LAUNCH;
    }
    return SOT_OK;
}

int sot_abi_version(void) { return SOT_ABI_VERSION; }

#ifdef SOT_STAMPS
// diagnostic build only: copies the phase stamps of workgroup 0's second row to the host (synchronises)
int sot_debug_read_stamps(unsigned long long* host_out, int count)
{
    if (count > 64) count = 64;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(sot::g_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : -1;
}
#endif

const char* sot_status_string(int status)
{
    switch (status) {
        case SOT_OK: return "ok";
        case SOT_ERR_INVALID_P: return "The OT loss is only valid for p>=1";
        case SOT_ERR_BAD_SHAPE: return "bad shape or stride";
        case SOT_ERR_UNSUPPORTED_SIZE: return "row working set exceeds one CU's LDS (n + m too large)";
        case SOT_ERR_NULL_POINTER: return "null pointer";
        case SOT_ERR_WORKSPACE: return "workspace too small (see sot_workspace_bytes)";
        case SOT_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown status";
    }
}

size_t sot_workspace_bytes(const sot_problem* prob)
{
    if (prob == nullptr || prob->n < 1 || prob->m < 1) return 0;
    if (prob->xpos_row_stride != 0)   // per-row positions: room for the pre-sort's permutations (optional: without it the row kernel sorts in LDS)
        return (prob->flags & SOT_FLAG_REQUIRE_SORT) && prob->row_perm_in == nullptr && prob->row_perm_out == nullptr && prob->B > 0
                   ? sot::rowpos_perm_bytes(prob->B, prob->n, prob->m) : 0;
    return sot::ws_layout(prob->n, prob->m).total;
}

int sot_prepare_positions(const float* xpos, const float* ypos, int32_t n, int32_t m, float* xpos_sorted, float* ypos_sorted,
                          int32_t* xperm, int32_t* yperm, int32_t* perm_is_identity, void* stream)
{
    if (n < 1 || m < 1) return SOT_ERR_BAD_SHAPE;
    if (!xpos || !ypos || !xpos_sorted || !ypos_sorted || !xperm || !yperm || !perm_is_identity) return SOT_ERR_NULL_POINTER;
    return sot::launch_prepare(xpos, ypos, n, m, xpos_sorted, ypos_sorted, xperm, yperm, perm_is_identity,
                               reinterpret_cast<hipStream_t>(stream), false);
}

int sot_prepare_unit_positions(const float* xpos, const float* ypos, int32_t n, int32_t m, float* xpos_sorted, float* ypos_sorted,
                               int32_t* xperm, int32_t* yperm, int32_t* perm_is_identity, void* stream)
{
    if (n < 1 || m < 1) return SOT_ERR_BAD_SHAPE;
    if (!xpos || !ypos || !xpos_sorted || !ypos_sorted || !xperm || !yperm || !perm_is_identity) return SOT_ERR_NULL_POINTER;
    return sot::launch_prepare(xpos, ypos, n, m, xpos_sorted, ypos_sorted, xperm, yperm, perm_is_identity,
                               reinterpret_cast<hipStream_t>(stream), true);
}

int sot_w1d_forward(const sot_problem* prob, float* row_loss, void* workspace, size_t workspace_bytes, void* stream)
{
    if (prob != nullptr && prob->B > 0 && row_loss == nullptr) return SOT_ERR_NULL_POINTER;
    return sot::run_forward(prob, row_loss, nullptr, nullptr, nullptr, nullptr, nullptr, false, workspace, workspace_bytes, stream);
}

int sot_w1d_quantiles(const sot_problem* prob, float* uq, float* vq, float* Q, float* U, float* V, void* workspace,
                      size_t workspace_bytes, void* stream)
{
    return sot::run_forward(prob, nullptr, uq, vq, Q, U, V, true, workspace, workspace_bytes, stream);
}

int sot_w1d_reduce_mean(const float* row_loss, int64_t B, double denom, int apply_hinge, float hinge_threshold, float* mean_out,
                        double* sum_out, void* stream)
{
    if (B < 0) return SOT_ERR_BAD_SHAPE;
    if (B > 0 && row_loss == nullptr) return SOT_ERR_NULL_POINTER;
    if (mean_out == nullptr && sum_out == nullptr) return SOT_ERR_NULL_POINTER;
    (void)hipGetLastError();  // do not inherit a stale error from earlier runtime calls
    hipLaunchKernelGGL(sot::sot_reduce_mean_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), row_loss, B,
                       denom, apply_hinge, hinge_threshold, mean_out, sum_out);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

int sot_w1d_backward(const sot_problem* prob, const float* grad_row, int64_t grad_row_stride, float grad_scale, float* grad_x,
                     float* grad_y, void* workspace, size_t workspace_bytes, void* stream)
{
    if (grad_row_stride != 0 && grad_row_stride != 1) return SOT_ERR_BAD_SHAPE;
    return sot::run_backward(prob, grad_row, grad_row_stride, grad_scale, grad_x, grad_y, workspace, workspace_bytes, stream);
}

static inline sot::MeanTail make_tail(uint32_t* counters, double denom, int apply_hinge, float hinge, float* mean_out, double* sum_out)
{
    sot::MeanTail mt{};
    mt.counters = counters; mt.denom = denom; mt.apply_hinge = apply_hinge; mt.hinge = hinge; mt.mean_out = mean_out; mt.sum_out = sum_out;
    return mt;
}

int sot_w1d_loss_and_grad(const sot_problem* prob, float* row_loss, double denom, float* mean_out, double* sum_out, float grad_scale,
                          float* grad_y, uint32_t* completion_counters, void* workspace, size_t workspace_bytes, void* stream)
{
    if (prob == nullptr) return SOT_ERR_NULL_POINTER;
    if (prob->B == 0) return SOT_ERR_BAD_SHAPE;  // the mean of zero rows is undefined
    if (row_loss == nullptr || grad_y == nullptr || (mean_out == nullptr && sum_out == nullptr)) return SOT_ERR_NULL_POINTER;
    const sot::MeanTail mt = make_tail(completion_counters, denom, 0, 0.0f, mean_out, sum_out);
    const sot::MeanTail* tail = completion_counters ? &mt : nullptr;
    bool fused = false;
    int rc = sot::run_backward(prob, nullptr, 0, grad_scale, nullptr, grad_y, workspace, workspace_bytes, stream, row_loss, &fused, tail);
    if (rc != SOT_OK) return rc;
    if (!fused) {
        rc = sot::run_forward(prob, row_loss, nullptr, nullptr, nullptr, nullptr, nullptr, false, workspace, workspace_bytes, stream, tail);
        if (rc != SOT_OK) return rc;
    }
    if (tail != nullptr) return SOT_OK;  // the kernel that wrote the row losses reduced them
    return sot_w1d_reduce_mean(row_loss, prob->B, denom, 0, 0.0f, mean_out, sum_out, stream);
}

int sot_scale_inplace(float* data, int64_t count, const float* scalar, void* stream)
{
    if (count < 0) return SOT_ERR_BAD_SHAPE;
    if (count == 0) return SOT_OK;
    if (data == nullptr || scalar == nullptr) return SOT_ERR_NULL_POINTER;
    const int64_t need = (count + 4 * 256 - 1) / (4 * 256);
    const int grid = (int)(need < 256 * 16 ? need : 256 * 16);
    (void)hipGetLastError();
    hipLaunchKernelGGL(sot::sot_scale_inplace_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), data, count, scalar);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

int sot_w1d_loss(const sot_problem* prob, float* row_loss, double denom, int apply_hinge, float hinge_threshold, float* mean_out,
                 double* sum_out, uint32_t* completion_counters, void* workspace, size_t workspace_bytes, void* stream)
{
    if (prob != nullptr && prob->B > 0 && row_loss == nullptr) return SOT_ERR_NULL_POINTER;
    if (mean_out == nullptr && sum_out == nullptr) return SOT_ERR_NULL_POINTER;
    if (prob != nullptr && prob->B == 0) return SOT_ERR_BAD_SHAPE;  // the mean of zero rows is undefined
    const sot::MeanTail mt = make_tail(completion_counters, denom, apply_hinge, hinge_threshold, mean_out, sum_out);
    const int rc = sot::run_forward(prob, row_loss, nullptr, nullptr, nullptr, nullptr, nullptr, false, workspace, workspace_bytes,
                                    stream, completion_counters ? &mt : nullptr);
    if (rc != SOT_OK || completion_counters != nullptr) return rc;
    return sot_w1d_reduce_mean(row_loss, prob->B, denom, apply_hinge, hinge_threshold, mean_out, sum_out, stream);
}

int sot_w1d_position_grad(const sot_problem* prob, const float* grad_row, int64_t grad_row_stride, float grad_scale, float* grad_xpos,
                          float* grad_ypos, void* workspace, size_t workspace_bytes, void* stream)
{
    if (grad_row_stride != 0 && grad_row_stride != 1) return SOT_ERR_BAD_SHAPE;
    return sot::run_position_grad(prob, grad_row, grad_row_stride, grad_scale, grad_xpos, grad_ypos, workspace, workspace_bytes, stream);
}

int sot_column_sum(const float* rows, int64_t B, int32_t n, int64_t row_stride, float* out, void* stream)
{
    if (B < 0 || n < 1 || row_stride < n) return SOT_ERR_BAD_SHAPE;
    if (out == nullptr || (B > 0 && rows == nullptr)) return SOT_ERR_NULL_POINTER;
    return sot::run_column_sum(rows, B, (int)n, row_stride, out, stream);
}

int sot_segmented_sort(const float* keys, int64_t B, int32_t n, int64_t row_stride, float* sorted_keys, int64_t* indices,
                       void* stream)
{
    if (B < 0 || n < 1 || row_stride < n) return SOT_ERR_BAD_SHAPE;
    if (B == 0) return SOT_OK;
    if (keys == nullptr) return SOT_ERR_NULL_POINTER;
    if (SOT_WAVE_SORT && n <= 2048) {   // one wavefront per row (round 6)
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        if (n <= 128) return sot::launch_segmented_sort_wave<2>(keys, B, (int)n, row_stride, sorted_keys, indices, st);
        if (n <= 512) return sot::launch_segmented_sort_wave<8>(keys, B, (int)n, row_stride, sorted_keys, indices, st);
        if (n <= 1024) return sot::launch_segmented_sort_wave<16>(keys, B, (int)n, row_stride, sorted_keys, indices, st);
        return sot::launch_segmented_sort_wave<32>(keys, B, (int)n, row_stride, sorted_keys, indices, st);
    }
    const int npad = sot::sort16_npad((int)n);
    if (npad > 16 * 1024) return SOT_ERR_UNSUPPORTED_SIZE;   // one block of 16 elements per thread
    const size_t lds = (size_t)((sot::sort16_capacity(npad) + 3) & ~3) * 8;
    if (lds > sot::kLdsLimit) return SOT_ERR_UNSUPPORTED_SIZE;
    static sot::GridCache cache;
    sot::allow_full_lds_once(cache, reinterpret_cast<const void*>(sot::sot_segmented_sort_kernel));
    int block = 64;                                    // one thread per block of 16 elements (whole wavefronts)
    while (block < 1024 && npad > 16 * block) block <<= 1;
    int per_cu = (int)(sot::kLdsLimit / lds);
    const int wave_cap = 32 / (block / 64);            // at most 8 wavefronts per SIMD
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu > 16) per_cu = 16;
    int64_t cap = (int64_t)sot::device_cu_count() * per_cu;
    const int grid = (int)(B < cap ? B : cap);
    (void)hipGetLastError();  // do not inherit a stale error from earlier runtime calls
    hipLaunchKernelGGL(sot::sot_segmented_sort_kernel, dim3(grid), dim3(block), lds, reinterpret_cast<hipStream_t>(stream), keys, B,
                       (int)n, row_stride, sorted_keys, indices);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

}  // extern "C"
#endif  // SOT_PART & 16

#if (SOT_PART & 32) || defined(SOT_STUB_MISSING_PARTS)
extern "C" int sot_w1d_forward_csr(const float* x_weights, const float* x_positions, const int64_t* x_offsets, int64_t x_nnz,
                                   const float* y_weights, const float* y_positions, const int64_t* y_offsets, int64_t y_nnz,
                                   int64_t B, int32_t max_n, int32_t max_m, float p, uint32_t flags, float* row_loss, void* stream)
{
    return sot::run_forward_csr(x_weights, x_positions, x_offsets, x_nnz, y_weights, y_positions, y_offsets, y_nnz, B, max_n,
                                max_m, p, flags, row_loss, stream);
}
#endif
