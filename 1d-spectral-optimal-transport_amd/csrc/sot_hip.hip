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
#include <stdint.h>
#include <limits.h>
#include <math.h>

#include "../../include/sot_hip.h"
#include "sot_device.hpp"

namespace sot {

// ---------------------------------------------------------------------------------------------
// LDS layout of one row group (identical arithmetic on host and device)
// ---------------------------------------------------------------------------------------------
struct RowLayout {
    int nU, nV;      // floats reserved for U (n+1 incl. sentinel) and V (m+1), multiples of 4
    int poff;        // PX = U + poff, PY = V + poff
    int part_x, part_y;  // chunk-sum scratch of the two row masses
    int wtot;        // 2 * NW doubles (as float offset, even)
    int red;         // 2*NW floats (reduction scratch) + 2 floats (S_x, S_y)
    int grad;        // backward only: GU (nU floats) | GV (nV floats)
    int row_floats;  // total, multiple of 4
};

__host__ __device__ inline int align4(int v) { return (v + 3) & ~3; }
__host__ __device__ inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

__host__ __device__ inline RowLayout make_layout(int n, int m, int G, bool rowpos, bool with_grad = false)
{
    RowLayout L;
    L.nU = align4(n + 1);
    L.nV = align4(m + 1);
    if (rowpos) {  // per-row position sort needs power-of-two scratch for the bitonic network
        L.nU = max(L.nU, next_pow2(n));
        L.nV = max(L.nV, next_pow2(m));
    }
    L.poff = L.nU + L.nV;
    const int nchx = (((n >= 8) ? (n >> 5) : 0) + 15) >> 4;
    const int nchy = (((m >= 8) ? (m >> 5) : 0) + 15) >> 4;
    L.part_x = 2 * L.poff;
    L.part_y = L.part_x + 32 * nchx;
    L.wtot = align4(L.part_y + 32 * nchy);
    const int NW = G / kWave;
    L.red = L.wtot + 4 * NW;  // 2 arrays * NW doubles = 4*NW floats
    L.grad = align4(L.red + 2 * NW + 2);
    L.row_floats = with_grad ? L.grad + L.nU + L.nV : L.grad;
    return L;
}

struct FwdArgs {
    const float* x; const float* y;
    const float* xpos; const float* ypos;   // sorted positions when !ROWPOS
    const int* xperm; const int* yperm;     // shared-position sort permutations (may be null)
    const int* ident;                       // [2] device flags: permutation is the identity (may be null)
    int64_t B; int n, m;
    int64_t xs, ys, xps, yps;               // row strides (elements)
    float p; uint32_t flags;
    float* row_loss;
    // optional outputs of the quantile variant
    float* oUq; float* oVq; float* oQ; float* oU; float* oV;
};

// ---------------------------------------------------------------------------------------------
// Shared-position preparation (losses.py:287-288 for row-invariant positions): one workgroup per
// array checks sortedness and, if needed, sorts (position, index) pairs in LDS.
// ---------------------------------------------------------------------------------------------
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
    const int npad = next_pow2(len);
    float* key = smem;
    int* idx = reinterpret_cast<int*>(smem + npad);
    int* const unsorted_flag = reinterpret_cast<int*>(smem + 2 * npad);  // all LDS is dynamic (16-B aligned carve)
    const int t = threadIdx.x, T = blockDim.x;
    if (t == 0) *unsorted_flag = 0;
    for (int i = t; i < npad; i += T) {
        key[i] = (i < len) ? pos[i] : INFINITY;
        idx[i] = (i < len) ? i : INT_MAX;
    }
    __syncthreads();
    int bad = 0;
    for (int i = t; i + 1 < len; i += T) bad |= (key[i] > key[i + 1]);
    if (bad) *unsorted_flag = 1;
    __syncthreads();
    const bool need_sort = *unsorted_flag != 0;
    if (need_sort) bitonic_sort_kv(key, idx, npad, t, T, [] { __syncthreads(); });
    for (int i = t; i < len; i += T) { spos[i] = key[i]; perm[i] = idx[i]; }
    if (t == 0) ident[which] = need_sort ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------
// Per-row-group context shared by the forward and backward kernels
// ---------------------------------------------------------------------------------------------
template <int G>
struct RowCtx {
    static constexpr int NW = G / kWave;
    float* base; float* U; float* V; float* PX; float* PY;
    float* partx; float* party; double* wtot; float* red; float* Sv;
    float* GU; float* GV;  // backward only
    RowLayout L;
    MassPlan mpx, mpy;
    int n, m, K, E, cptn, cptm;
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
    c.U = c.base; c.V = c.base + c.L.nU; c.PX = c.U + c.L.poff; c.PY = c.V + c.L.poff;
    c.partx = c.base + c.L.part_x; c.party = c.base + c.L.part_y;
    c.wtot = reinterpret_cast<double*>(c.base + c.L.wtot);
    c.red = c.base + c.L.red;
    c.Sv = c.red + 2 * RowCtx<G>::NW;
    c.GU = c.base + c.L.grad; c.GV = c.GU + c.L.nU;
    c.prenorm = a.flags & SOT_FLAG_PRENORMALIZED;
    c.sq = !c.prenorm && (a.flags & SOT_FLAG_SQUARE);
    c.dn = c.prenorm || (a.flags & SOT_FLAG_DONT_NORMALIZE);
    c.lim = a.flags & SOT_FLAG_LIMIT_Q;
    c.do_sort = a.flags & SOT_FLAG_REQUIRE_SORT;
    c.p = a.p;
    c.mpx = make_mass_plan(a.n); c.mpy = make_mass_plan(a.m);
    c.cptn = (a.n + G - 1) / G; c.cptm = (a.m + G - 1) / G;
    c.K = a.n + a.m;
    c.E = (c.K + G - 1) / G;
    c.x_ident = true; c.y_ident = true;
    if (!ROWPOS) {
        if (a.ident != nullptr) { c.x_ident = a.ident[0] != 0; c.y_ident = a.ident[1] != 0; }
        for (int e = c.t; e < c.n; e += G) c.PX[e] = a.xpos[e];
        for (int e = c.t; e < c.m; e += G) c.PY[e] = a.ypos[e];
        if (c.t == 0) {
            c.PX[c.n] = a.xpos[c.n - 1];  // clamp of losses.py:220: ranks beyond the last index reuse it
            c.PY[c.m] = a.ypos[c.m - 1];
            c.U[c.n] = INFINITY;          // sentinels: an exhausted side never wins the merge
            c.V[c.m] = INFINITY;
        }
    }
    return c;
}

// Phases P0-P3 for one row: positions (ROWPOS), staging, row masses, safe_divide, weight gather,
// fp64-accumulated CDFs.  On return (after its final barrier) U/V hold the CDFs, PX/PY the sorted
// positions incl. clamp slots, c.Sv the two masses.  wx/wy receive the ORIGINAL (unsquared) weights of
// the elements this thread owns in sorted order; sidx/sidy their original column (for the backward
// scatter); both are dead code in the forward kernel.
template <int G, int CPT, bool ROWPOS>
__device__ __forceinline__ void build_cdfs(const FwdArgs& a, const RowCtx<G>& c, const float* xr, const float* yr, int64_t rowc,
                                           float (&wx)[CPT], float (&wy)[CPT], int (&sidx)[CPT], int (&sidy)[CPT])
{
    constexpr int NW = G / kWave;
    const int n = c.n, m = c.m, t = c.t;
    float* const U = c.U; float* const V = c.V; float* const PX = c.PX; float* const PY = c.PY;
    const bool sq = c.sq;
    const int ex0 = t * c.cptn, ey0 = t * c.cptm;

    // ---- P0 (ROWPOS only): this row's positions, sorted in LDS with an index payload --------------
    int ix[ROWPOS ? CPT : 1], iy[ROWPOS ? CPT : 1];
    if (ROWPOS) {
        const float* xp = a.xpos + rowc * a.xps;
        const float* yp = a.ypos + rowc * a.yps;
        if (c.do_sort) {
            int* const IX = reinterpret_cast<int*>(U);  // index payloads alias U/V until the weights arrive
            int* const IY = reinterpret_cast<int*>(V);
            const int npx = next_pow2(n), npy = next_pow2(m);
            for (int e = t; e < npx; e += G) { PX[e] = (e < n) ? xp[e] : INFINITY; IX[e] = (e < n) ? e : INT_MAX; }
            for (int e = t; e < npy; e += G) { PY[e] = (e < m) ? yp[e] : INFINITY; IY[e] = (e < m) ? e : INT_MAX; }
            __syncthreads();
            bitonic_sort_kv(PX, IX, npx, t, G, [] { __syncthreads(); });
            bitonic_sort_kv(PY, IY, npy, t, G, [] { __syncthreads(); });
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int ex = ex0 + k, ey = ey0 + k;
                ix[k] = ((k < c.cptn) && (ex < n)) ? IX[ex] : 0;
                iy[k] = ((k < c.cptm) && (ey < m)) ? IY[ey] : 0;
            }
            __syncthreads();
        } else {
            for (int e = t; e < n; e += G) PX[e] = xp[e];
            for (int e = t; e < m; e += G) PY[e] = yp[e];
        }
    }
    // ---- P1: stage the raw weights in ORIGINAL order, 16 B per lane when the row is aligned --------
    //      (squaring happens on the way out of LDS so that the backward keeps the unsquared value)
    if (((n & 3) == 0) && ((reinterpret_cast<uintptr_t>(xr) & 15) == 0)) {
        const float4* src = reinterpret_cast<const float4*>(xr);
        float4* dst = reinterpret_cast<float4*>(U);
        for (int e = t; e < (n >> 2); e += G) dst[e] = src[e];
    } else {
        for (int e = t; e < n; e += G) U[e] = xr[e];
    }
    if (((m & 3) == 0) && ((reinterpret_cast<uintptr_t>(yr) & 15) == 0)) {
        const float4* src = reinterpret_cast<const float4*>(yr);
        float4* dst = reinterpret_cast<float4*>(V);
        for (int e = t; e < (m >> 2); e += G) dst[e] = src[e];
    } else {
        for (int e = t; e < m; e += G) V[e] = yr[e];
    }
    if (ROWPOS && t == 0) { PX[n] = PX[n - 1]; PY[m] = PY[m - 1]; }
    __syncthreads();

    // ---- P2: row masses in ATen order (losses.py:177,184; the reference sums BEFORE it sorts, so the
    //      staged row is still in its original element order here) ----------------------------------
    if (!c.prenorm) {
        if (sq) {
            mass_chunk_sums<G, true>(U, c.partx, c.mpx, t);
            if (!c.dn) mass_chunk_sums<G, true>(V, c.party, c.mpy, (t + G / 2) & (G - 1));
        } else {
            mass_chunk_sums<G, false>(U, c.partx, c.mpx, t);
            if (!c.dn) mass_chunk_sums<G, false>(V, c.party, c.mpy, (t + G / 2) & (G - 1));
        }
        __syncthreads();
        if (c.wv == 0) {
            const int half = c.lane >> 5, col = c.lane & 31;
            const bool use_y = half && !c.dn;  // in dont_normalize mode both halves sum x: S_y := S_x
            const float S = sq ? mass_fold<true>(use_y ? V : U, use_y ? c.party : c.partx, use_y ? c.mpy : c.mpx, col, half << 5)
                               : mass_fold<false>(use_y ? V : U, use_y ? c.party : c.partx, use_y ? c.mpy : c.mpx, col, half << 5);
            if (col == 0) c.Sv[half] = S;
        }
    } else if (t == 0) {
        c.Sv[0] = 1.0f;  // w / 1.0f == w exactly: weights enter the CDF unchanged
        c.Sv[1] = 1.0f;
    }
    __syncthreads();

    // ---- P3: safe_divide (utils.py:135-142), weight gather by the position sort (losses.py:289-290)
    //      and fp64-accumulated CDFs (losses.py:292-293) --------------------------------------------
    const float Sxh = guard_mass(c.Sv[0]);
    const float Syh = guard_mass(c.Sv[1]);
    const bool x_perm = ROWPOS ? c.do_sort : !c.x_ident;
    const bool y_perm = ROWPOS ? c.do_sort : !c.y_ident;
    double px[CPT], py[CPT];
    double runx = 0.0, runy = 0.0;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int ex = ex0 + k, ey = ey0 + k;
        const bool okx = (k < c.cptn) && (ex < n), oky = (k < c.cptm) && (ey < m);
        int sx = ex, sy = ey;
        if (ROWPOS) { if (c.do_sort) { sx = ix[k]; sy = iy[k]; } }
        else { if (x_perm && okx) sx = a.xperm[ex]; if (y_perm && oky) sy = a.yperm[ey]; }
        const float ox = okx ? U[sx] : 0.0f;
        const float oy = oky ? V[sy] : 0.0f;
        wx[k] = ox; wy[k] = oy; sidx[k] = sx; sidy[k] = sy;
        const float qx = (sq ? ox * ox : ox) / Sxh;  // IEEE division (built without fast-math / contraction)
        const float qy = (sq ? oy * oy : oy) / Syh;
        runx += okx ? (double)qx : 0.0;
        runy += oky ? (double)qy : 0.0;
        px[k] = runx;
        py[k] = runy;
    }
    const double inx = wave_incl_scan(runx), iny = wave_incl_scan(runy);
    double exx = __shfl_up(inx, 1), exy = __shfl_up(iny, 1);
    if (c.lane == 0) { exx = 0.0; exy = 0.0; }
    if (NW > 1 && c.lane == kWave - 1) { c.wtot[c.wv] = inx; c.wtot[NW + c.wv] = iny; }
    __syncthreads();  // every raw weight has been read (also through permutations) before U/V are rewritten
    if (NW > 1) {
        double ox = 0.0, oy = 0.0;
        for (int w = 0; w < c.wv; ++w) { ox += c.wtot[w]; oy += c.wtot[NW + w]; }
        exx += ox;
        exy += oy;
    }
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int ex = ex0 + k, ey = ey0 + k;
        if ((k < c.cptn) && (ex < n)) U[ex] = (float)(exx + px[k]);
        if ((k < c.cptm) && (ey < m)) V[ey] = (float)(exy + py[k]);
    }
    if (ROWPOS && t == 0) { U[n] = INFINITY; V[m] = INFINITY; }
    __syncthreads();
}

// left rank of q in a sorted LDS array: #{A_i < q}  (torch.searchsorted side='left', losses.py:219)
__device__ __forceinline__ int lower_rank(const float* A, int len, float q)
{
    int lo = 0, hi = len;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (A[mid] < q) lo = mid + 1; else hi = mid; }
    return lo;
}

// ---------------------------------------------------------------------------------------------
// Forward kernel.  G threads per row, CPT = max elements of one array owned by a thread during
// the scan, ROWPOS = positions differ per row (sorted in LDS when REQUIRE_SORT), QUANT = also
// emit the return_quantiles tensors.
// ---------------------------------------------------------------------------------------------
template <int G, int CPT, bool ROWPOS, bool QUANT>
__global__ __launch_bounds__((G < 256 ? 256 : G)) void sot_forward_kernel(const FwdArgs a)
{
    constexpr int BLOCK = (G < 256 ? 256 : G);
    constexpr int RPW = BLOCK / G;   // rows processed concurrently by one workgroup
    constexpr int NW = G / kWave;    // wavefronts per row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RowCtx<G> c = make_ctx<G, ROWPOS>(a, smem, false);
    const int rg = threadIdx.x / G;
    const int n = c.n, m = c.m, K = c.K, t = c.t;
    float* const U = c.U; float* const V = c.V; float* const PX = c.PX; float* const PY = c.PY;

    const int64_t row_step = (int64_t)gridDim.x * RPW;
    for (int64_t row0 = (int64_t)blockIdx.x * RPW; row0 < a.B; row0 += row_step) {
        const int64_t row = row0 + rg;
        const bool valid = row < a.B;
        const int64_t rowc = valid ? row : a.B - 1;
        float wx[CPT], wy[CPT]; int sidx[CPT], sidy[CPT];
        build_cdfs<G, CPT, ROWPOS>(a, c, a.x + rowc * a.xs, a.y + rowc * a.ys, rowc, wx, wy, sidx, sidy);

        // ---- P4: merge of the two CDFs = sort(cat(U,V)) + searchsorted + take_along_dim -----------
        //      (losses.py:295-298), level widths, cutoff mask, |.|^p, weighted sum (:301-313)
        float acc = 0.0f;
        {
            const int D0 = min(t * c.E, K), D1 = min(D0 + c.E, K);
            if (D0 < D1) {
                int i = merge_path(U, V, n, m, D0);
                int j = D0 - i;
                float qprev = 0.0f;  // Q_0 := 0 (the pad of losses.py:301)
                if (i > 0) qprev = U[i - 1];
                if (j > 0) qprev = fmaxf(qprev, V[j - 1]);
                float ua = U[i], vb = V[j], xa = PX[i], yb = PY[j];
                for (int k = D0; k < D1; ++k) {
                    const bool tu = ua <= vb;
                    const float q = tu ? ua : vb;
                    const float cost = transport_cost(xa, yb, c.p);
                    float delta = q - qprev;
                    if (c.lim && q > 1.0f) delta = 0.0f;
                    acc += delta * cost;
                    if (QUANT && valid) {
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
                    qprev = q;
                    const int idx = tu ? ++i : (c.L.nU + ++j);
                    const float nv = c.base[idx];
                    const float np = c.base[idx + c.L.poff];
                    ua = tu ? nv : ua;
                    xa = tu ? np : xa;
                    vb = tu ? vb : nv;
                    yb = tu ? yb : np;
                }
            }
        }
        if (QUANT && valid) {
            if (a.oU) for (int e = t; e < n; e += G) a.oU[row * (int64_t)n + e] = U[e];
            if (a.oV) for (int e = t; e < m; e += G) a.oV[row * (int64_t)m + e] = V[e];
        }
        acc = wave_sum(acc);
        if (c.lane == 0) c.red[c.wv] = acc;
        __syncthreads();
        if (t == 0 && valid && a.row_loss) {
            float tot = c.red[0];
            for (int w = 1; w < NW; ++w) tot += c.red[w];
            a.row_loss[row] = tot;
        }
        // the barrier above also orders this row's last LDS reads before the next row's staging
    }
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
    const float* grad_row;
    float* gx; float* gy;
};

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

template <int G, int CPT, bool ROWPOS>
__global__ __launch_bounds__((G < 256 ? 256 : G)) void sot_backward_kernel(const BwdArgs b)
{
    constexpr int BLOCK = (G < 256 ? 256 : G);
    constexpr int RPW = BLOCK / G;
    constexpr int NW = G / kWave;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdArgs& a = b.f;
    const RowCtx<G> c = make_ctx<G, ROWPOS>(a, smem, true);
    const int rg = threadIdx.x / G;
    const int n = c.n, m = c.m, K = c.K, t = c.t;
    float* const U = c.U; float* const V = c.V; float* const PX = c.PX; float* const PY = c.PY;

    const int64_t row_step = (int64_t)gridDim.x * RPW;
    for (int64_t row0 = (int64_t)blockIdx.x * RPW; row0 < a.B; row0 += row_step) {
        const int64_t row = row0 + rg;
        const bool valid = row < a.B;
        const int64_t rowc = valid ? row : a.B - 1;
        float wx[CPT], wy[CPT]; int sidx[CPT], sidy[CPT];
        build_cdfs<G, CPT, ROWPOS>(a, c, a.x + rowc * a.xs, a.y + rowc * a.ys, rowc, wx, wy, sidx, sidy);

        // ---- merge walk: route each run's gradient to its last member --------------------------------
        {
            const int D0 = min(t * c.E, K), D1 = min(D0 + c.E, K);
            if (D0 < D1) {
                int i = merge_path(U, V, n, m, D0);
                int j = D0 - i;
                float qprev = 0.0f;
                if (i > 0) qprev = U[i - 1];
                if (j > 0) qprev = fmaxf(qprev, V[j - 1]);
                float ua = U[i], vb = V[j], xa = PX[i], yb = PY[j];
                float dcur = 0.0f;
                {   // cost of the run the first element belongs to
                    const float q0 = fminf(ua, vb);
                    if (D0 > 0 && q0 == qprev) {  // we start inside a run: use the run's first member's ranks
                        const float cst = transport_cost(PX[lower_rank(U, n, q0)], PY[lower_rank(V, m, q0)], c.p);
                        dcur = (c.lim && q0 > 1.0f) ? 0.0f : cst;
                    }
                }
                for (int k = D0; k < D1; ++k) {
                    const bool tu = ua <= vb;
                    const float q = tu ? ua : vb;
                    if (k == 0 || q != qprev) {  // a new run starts here: its cost uses the running counts
                        const float cst = transport_cost(xa, yb, c.p);
                        dcur = (c.lim && q > 1.0f) ? 0.0f : cst;
                    }
                    qprev = q;
                    const int slot = tu ? i : (c.L.nU + j);  // GU[i] or GV[j]  (GV = GU + nU)
                    const int idx = tu ? ++i : (c.L.nU + ++j);
                    const float nv = c.base[idx];
                    const float np = c.base[idx + c.L.poff];
                    ua = tu ? nv : ua;
                    xa = tu ? np : xa;
                    vb = tu ? vb : nv;
                    yb = tu ? yb : np;
                    float g = 0.0f;
                    if (k + 1 == K) {
                        g = dcur;
                    } else {
                        const float qn = fminf(ua, vb);
                        if (qn != q) {
                            const float cn = transport_cost(xa, yb, c.p);
                            g = dcur - ((c.lim && qn > 1.0f) ? 0.0f : cn);
                        }
                    }
                    c.GU[slot] = g;
                }
            }
        }
        __syncthreads();

        // ---- reverse cumsums (fp64), normalisation terms, scatter to the original columns -------------
        const int ex0 = t * c.cptn, ey0 = t * c.cptm;
        double ga[CPT], gb[CPT];
        double runx = 0.0, runy = 0.0;
#pragma unroll
        for (int k = CPT - 1; k >= 0; --k) {
            const int ex = ex0 + k, ey = ey0 + k;
            const bool okx = (k < c.cptn) && (ex < n), oky = (k < c.cptm) && (ey < m);
            runx += okx ? (double)c.GU[ex] : 0.0;
            runy += oky ? (double)c.GV[ey] : 0.0;
            ga[k] = runx;
            gb[k] = runy;
        }
        const double inx = wave_suffix_incl_scan(runx), iny = wave_suffix_incl_scan(runy);
        double exx = __shfl_down(inx, 1), exy = __shfl_down(iny, 1);
        if (c.lane == kWave - 1) { exx = 0.0; exy = 0.0; }
        if (NW > 1) {
            if (c.lane == 0) { c.wtot[c.wv] = inx; c.wtot[NW + c.wv] = iny; }
            __syncthreads();
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
            const int ex = ex0 + k, ey = ey0 + k;
            const bool okx = (k < c.cptn) && (ex < n), oky = (k < c.cptm) && (ey < m);
            const float sx = c.sq ? wx[k] * wx[k] : wx[k];
            const float sy = c.sq ? wy[k] * wy[k] : wy[k];
            dotx += okx ? ga[k] * (double)sx : 0.0;
            doty += oky ? gb[k] * (double)sy : 0.0;
        }
        dotx = wave_sum(dotx);
        doty = wave_sum(doty);
        if (NW > 1) __syncthreads();  // wtot is reused below
        if (c.lane == 0) { c.wtot[c.wv] = dotx; c.wtot[NW + c.wv] = doty; }
        __syncthreads();
        double totx = 0.0, toty = 0.0;
        for (int w = 0; w < NW; ++w) { totx += c.wtot[w]; toty += c.wtot[NW + w]; }
        const float Sx = c.Sv[0], Sy = c.Sv[1];
        const double dx = (double)guard_mass(Sx), dy = (double)guard_mass(Sy);
        double gSx = -totx, gSy = -toty;
        if (c.dn) { gSx += gSy; gSy = 0.0; }
        gSx = (!c.prenorm && Sx > kMassEps) ? gSx / (dx * dx) : 0.0;
        gSy = (!c.prenorm && Sy > kMassEps) ? gSy / (dy * dy) : 0.0;
        const double gr = (double)b.grad_row[rowc];
        if (valid) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int ex = ex0 + k, ey = ey0 + k;
                if (b.gx && (k < c.cptn) && (ex < n)) {
                    double g = ga[k] / dx + gSx;
                    if (c.sq) g *= 2.0 * (double)wx[k];
                    b.gx[row * (int64_t)n + sidx[k]] = (float)(g * gr);
                }
                if (b.gy && (k < c.cptm) && (ey < m)) {
                    double g = gb[k] / dy + gSy;
                    if (c.sq) g *= 2.0 * (double)wy[k];
                    b.gy[row * (int64_t)m + sidy[k]] = (float)(g * gr);
                }
            }
        }
        __syncthreads();  // GU/GV/wtot reads done before the next row reuses LDS
    }
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
    for (int64_t r = t; r < B; r += blockDim.x) {
        float v = row_loss[r];
        if (apply_hinge) v = fmaxf(v - hinge, 0.0f);
        acc += (double)v;
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

// ---------------------------------------------------------------------------------------------
// Standalone segmented sort (torch.sort(keys, 1) of losses.py:287-288): one workgroup per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sot_segmented_sort_kernel(const float* __restrict__ keys, int64_t B, int n, int64_t stride,
                                                                 float* __restrict__ out_keys, int64_t* __restrict__ out_idx)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int npad = next_pow2(n);
    float* key = smem;
    int* idx = reinterpret_cast<int*>(smem + npad);
    const int t = threadIdx.x, T = blockDim.x;
    for (int64_t row = blockIdx.x; row < B; row += gridDim.x) {
        const float* src = keys + row * stride;
        for (int i = t; i < npad; i += T) { key[i] = (i < n) ? src[i] : INFINITY; idx[i] = (i < n) ? i : INT_MAX; }
        __syncthreads();
        bitonic_sort_kv(key, idx, npad, t, T, [] { __syncthreads(); });
        for (int i = t; i < n; i += T) {
            if (out_keys) out_keys[row * (int64_t)n + i] = key[i];
            if (out_idx) out_idx[row * (int64_t)n + i] = (int64_t)idx[i];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
constexpr size_t kLdsLimit = 160 * 1024;

// Allow a kernel to use up to the CU's full 160 KiB of dynamic LDS; leaves no sticky error behind.
static void allow_full_lds(const void* kernel)
{
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit) != hipSuccess)
        (void)hipGetLastError();
}

static int device_cu_count()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        else
            cus = 256;
    }
    return cus;
}

struct LaunchCfg { int G, CPT; };

static bool pick_cfg(int n, int m, bool rowpos, bool with_grad, LaunchCfg* cfg, size_t* lds_bytes, int* block, int* rpw)
{
    const int N = n > m ? n : m;
    static const int Gs[] = {64, 128, 256, 512, 1024};
    for (int cpt = 8; cpt <= 16; cpt <<= 1) {
        for (int gi = 0; gi < 5; ++gi) {
            const int G = Gs[gi];
            if ((int64_t)G * cpt < N) continue;
            const int blk = G < 256 ? 256 : G;
            const int r = blk / G;
            const RowLayout L = make_layout(n, m, G, rowpos, with_grad);
            const size_t bytes = (size_t)r * L.row_floats * sizeof(float);
            if (bytes > kLdsLimit) continue;
            cfg->G = G; cfg->CPT = cpt; *lds_bytes = bytes; *block = blk; *rpw = r;
            return true;
        }
    }
    return false;
}

template <int G, int CPT, bool ROWPOS, bool QUANT>
static hipError_t launch_forward(const FwdArgs& a, size_t lds, int grid, int block, hipStream_t s)
{
    auto kern = sot_forward_kernel<G, CPT, ROWPOS, QUANT>;
    static bool attr_set = false;
    if (!attr_set) {
        allow_full_lds(reinterpret_cast<const void*>(kern));
        attr_set = true;
    }
    (void)hipGetLastError();  // do not inherit a stale error from earlier runtime calls
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, a);
    return hipGetLastError();
}

template <bool ROWPOS, bool QUANT>
static hipError_t dispatch_forward(const LaunchCfg& c, const FwdArgs& a, size_t lds, int grid, int block, hipStream_t s)
{
    if (c.CPT == 8) {
        switch (c.G) {
            case 64: return launch_forward<64, 8, ROWPOS, QUANT>(a, lds, grid, block, s);
            case 128: return launch_forward<128, 8, ROWPOS, QUANT>(a, lds, grid, block, s);
            case 256: return launch_forward<256, 8, ROWPOS, QUANT>(a, lds, grid, block, s);
            case 512: return launch_forward<512, 8, ROWPOS, QUANT>(a, lds, grid, block, s);
            default: return launch_forward<1024, 8, ROWPOS, QUANT>(a, lds, grid, block, s);
        }
    }
    return launch_forward<1024, 16, ROWPOS, QUANT>(a, lds, grid, block, s);
}

static int validate(const sot_problem* pr)
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

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WsLayout { size_t sx, sy, px, py, ident, total; };
static WsLayout ws_layout(int n, int m)
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

static int launch_prepare(const float* xpos, const float* ypos, int n, int m, float* sx, float* sy, int* px, int* py, int* ident,
                          hipStream_t s)
{
    const int npad = next_pow2(n > m ? n : m);
    const size_t prep_lds = (size_t)npad * 8 + 16;
    if (prep_lds > kLdsLimit) return SOT_ERR_UNSUPPORTED_SIZE;
    static bool attr_set = false;
    if (!attr_set) {
        allow_full_lds(reinterpret_cast<const void*>(sot_prepare_positions_kernel));
        attr_set = true;
    }
    (void)hipGetLastError();  // do not inherit a stale error from earlier runtime calls
    hipLaunchKernelGGL(sot_prepare_positions_kernel, dim3(2), dim3(1024), prep_lds, s, xpos, ypos, n, m, sx, sy, px, py, ident);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

struct Launch {
    FwdArgs a;
    LaunchCfg cfg;
    size_t lds;
    int block, grid;
    bool rowpos;
    hipStream_t s;
};

// validation, config choice, optional position preparation, persistent-grid sizing
static int setup_launch(const sot_problem* pr, bool with_grad, void* workspace, size_t workspace_bytes, void* stream, Launch* out)
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

    // persistent grid: as many workgroups as the chip holds at this LDS footprint
    int per_cu = (int)(kLdsLimit / l.lds);
    const int wave_cap = 2048 / l.block;
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu < 1) per_cu = 1;
    const int64_t want = (pr->B + rpw - 1) / rpw;
    const int64_t cap = (int64_t)device_cu_count() * per_cu;
    l.grid = (int)(want < cap ? want : cap);
    return SOT_OK;
}

static int run_forward(const sot_problem* pr, float* row_loss, float* uq, float* vq, float* Q, float* U, float* V,
                       bool quant, void* workspace, size_t workspace_bytes, void* stream)
{
    Launch l;
    int rc = setup_launch(pr, false, workspace, workspace_bytes, stream, &l);
    if (rc != SOT_OK) return rc;
    if (pr->B == 0) return SOT_OK;
    l.a.row_loss = row_loss;
    l.a.oUq = uq; l.a.oVq = vq; l.a.oQ = Q; l.a.oU = U; l.a.oV = V;
    hipError_t e;
    if (l.rowpos) e = quant ? dispatch_forward<true, true>(l.cfg, l.a, l.lds, l.grid, l.block, l.s)
                            : dispatch_forward<true, false>(l.cfg, l.a, l.lds, l.grid, l.block, l.s);
    else e = quant ? dispatch_forward<false, true>(l.cfg, l.a, l.lds, l.grid, l.block, l.s)
                   : dispatch_forward<false, false>(l.cfg, l.a, l.lds, l.grid, l.block, l.s);
    return e == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

template <int G, int CPT, bool ROWPOS>
static hipError_t launch_backward(const BwdArgs& b, size_t lds, int grid, int block, hipStream_t s)
{
    auto kern = sot_backward_kernel<G, CPT, ROWPOS>;
    static bool attr_set = false;
    if (!attr_set) {
        allow_full_lds(reinterpret_cast<const void*>(kern));
        attr_set = true;
    }
    (void)hipGetLastError();  // do not inherit a stale error from earlier runtime calls
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, b);
    return hipGetLastError();
}

template <bool ROWPOS>
static hipError_t dispatch_backward(const LaunchCfg& c, const BwdArgs& b, size_t lds, int grid, int block, hipStream_t s)
{
    if (c.CPT == 8) {
        switch (c.G) {
            case 64: return launch_backward<64, 8, ROWPOS>(b, lds, grid, block, s);
            case 128: return launch_backward<128, 8, ROWPOS>(b, lds, grid, block, s);
            case 256: return launch_backward<256, 8, ROWPOS>(b, lds, grid, block, s);
            case 512: return launch_backward<512, 8, ROWPOS>(b, lds, grid, block, s);
            default: return launch_backward<1024, 8, ROWPOS>(b, lds, grid, block, s);
        }
    }
    return launch_backward<1024, 16, ROWPOS>(b, lds, grid, block, s);
}

static int run_backward(const sot_problem* pr, const float* grad_row, float* gx, float* gy, void* workspace,
                        size_t workspace_bytes, void* stream)
{
    Launch l;
    int rc = setup_launch(pr, true, workspace, workspace_bytes, stream, &l);
    if (rc != SOT_OK) return rc;
    if (pr->B == 0 || (gx == nullptr && gy == nullptr)) return SOT_OK;
    if (grad_row == nullptr) return SOT_ERR_NULL_POINTER;
    BwdArgs b{};
    b.f = l.a; b.grad_row = grad_row; b.gx = gx; b.gy = gy;
    const hipError_t e = l.rowpos ? dispatch_backward<true>(l.cfg, b, l.lds, l.grid, l.block, l.s)
                                  : dispatch_backward<false>(l.cfg, b, l.lds, l.grid, l.block, l.s);
    return e == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

}  // namespace sot

// =============================================================================================
// C ABI (include/sot_hip.h)
// =============================================================================================
extern "C" {

int sot_abi_version(void) { return SOT_ABI_VERSION; }

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
    return sot::ws_layout(prob->n, prob->m).total;
}

int sot_prepare_positions(const float* xpos, const float* ypos, int32_t n, int32_t m, float* xpos_sorted, float* ypos_sorted,
                          int32_t* xperm, int32_t* yperm, int32_t* perm_is_identity, void* stream)
{
    if (n < 1 || m < 1) return SOT_ERR_BAD_SHAPE;
    if (!xpos || !ypos || !xpos_sorted || !ypos_sorted || !xperm || !yperm || !perm_is_identity) return SOT_ERR_NULL_POINTER;
    return sot::launch_prepare(xpos, ypos, n, m, xpos_sorted, ypos_sorted, xperm, yperm, perm_is_identity,
                               reinterpret_cast<hipStream_t>(stream));
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

int sot_w1d_backward(const sot_problem* prob, const float* grad_row, float* grad_x, float* grad_y, void* workspace,
                     size_t workspace_bytes, void* stream)
{
    return sot::run_backward(prob, grad_row, grad_x, grad_y, workspace, workspace_bytes, stream);
}

int sot_segmented_sort(const float* keys, int64_t B, int32_t n, int64_t row_stride, float* sorted_keys, int64_t* indices,
                       void* stream)
{
    if (B < 0 || n < 1 || row_stride < n) return SOT_ERR_BAD_SHAPE;
    if (B == 0) return SOT_OK;
    if (keys == nullptr) return SOT_ERR_NULL_POINTER;
    const size_t lds = (size_t)sot::next_pow2(n) * 8;
    if (lds > sot::kLdsLimit) return SOT_ERR_UNSUPPORTED_SIZE;
    static bool attr_set = false;
    if (!attr_set) {
        sot::allow_full_lds(reinterpret_cast<const void*>(sot::sot_segmented_sort_kernel));
        attr_set = true;
    }
    int per_cu = (int)(sot::kLdsLimit / lds);
    if (per_cu > 8) per_cu = 8;
    int64_t cap = (int64_t)sot::device_cu_count() * per_cu;
    const int grid = (int)(B < cap ? B : cap);
    (void)hipGetLastError();  // do not inherit a stale error from earlier runtime calls
    hipLaunchKernelGGL(sot::sot_segmented_sort_kernel, dim3(grid), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), keys, B,
                       (int)n, row_stride, sorted_keys, indices);
    return hipGetLastError() == hipSuccess ? SOT_OK : SOT_ERR_LAUNCH;
}

}  // extern "C"
