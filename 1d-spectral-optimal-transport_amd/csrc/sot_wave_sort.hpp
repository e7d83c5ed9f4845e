// sot_wave_sort.hpp -- ONE wavefront sorts a whole array of up to 64 KPL keys in its registers (round 6; gfx950, wave64).
//
// torch.sort(positions, 1) of losses.py:286-290 with the indices bit-exact, on a structurally cheaper footing than the in-LDS merge
// sort of sot_device.hpp (merge_sort16_kv2: ~170 lane instructions per key, every compare-exchange on a 64-bit (key, index) pair,
// seven barrier-separated merge rounds):
//
//  * every key becomes ONE 32-bit word -- a row-adaptive, MONOTONE quantisation q = trunc((x - min) C / (max - min)) in the high
//    32 - IDXBITS bits, the element's index in the low IDXBITS = 6 + log2(KPL) bits -- so ordering the words orders (q, index) and a
//    compare-exchange is v_min_u32 + v_max_u32, no payload.  (Quantising the RANGE of the row, not the float's bit pattern: the
//    order bits of a float spend 9 of their top 21 bits on sign and exponent, which would put ~128 colliding pairs into a row of
//    2048 uniform positions; 2^21 equal bins over [min, max] leave ~1.)
//  * a bitonic network in "flip" form (every exchange ascending) on the blocked layout position = lane KPL + register: exchanges
//    between registers are two full-rate VALU instructions per pair; exchanges between lanes are ONE cross-lane move (DPP quad_perm /
//    row_mirror / row_half_mirror, ds_swizzle, one ds_bpermute: the source is lane ^ m) plus ONE v_med3_u32 against a per-lane bound
//    (0 keeps the minimum, ~0 the maximum).  No barrier, no bisection, no LDS traffic but the swizzles: 45 + 21 VALU + 21 moves per
//    key for 2048 keys (KPL = 32).
//  * the words leave the registers through a skewed LDS image (position p at p + p / 32: both the blocked store and the striped
//    load are bank-conflict-free with immediate offsets) so that everything after it addresses position r 64 + lane;
//  * EXACTNESS: neighbours that share q (x ^ y < 2^IDXBITS, x != y) form a run; the lane whose block holds the run's first position
//    insertion-sorts it in LDS by the full key (strict ">": stable, the network left the run in index order).  A run longer than
//    kWaveSortRunLimit, NaN / infinite keys or a degenerate range make the function return false with key[] untouched: the caller
//    takes the merge sort (clustered positions pay the old price, nothing else changes).
//
// tests/wave_sort_model.py is the lane-by-lane CPU model of this file (network, bounds, skewed image, run repair).
#pragma once
#include "sot_device.hpp"

#ifndef SOT_WSORT_DPP
#define SOT_WSORT_DPP 0x808E   /* bit m: the move from lane ^ m is a DPP (a VALU instruction) -- possible for m = 1, 2, 3, 7, 15; else a ds_swizzle (LDS crossbar) */
#endif

// Between two stages of the network the instruction scheduler may not move anything (1): a stage is 32 independent exchanges -- all the
// parallelism a wave can use -- while a scheduler left free overlaps stages until it has used every register the occupancy target
// allows (256 in the stand-alone kernel), and inlined into the row kernels that pushed THEIR long-lived values into scratch.
#ifndef SOT_WSORT_GROUP
#define SOT_WSORT_GROUP 8   /* registers per cross-lane group (moves in flight) */
#endif
#ifndef SOT_WSORT_TRANSPOSE_MIN
#define SOT_WSORT_TRANSPOSE_MIN 4   /* merges whose largest lane distance S / 4 is at least this run their lane stages on the transposed layout (64: never) */
#endif
#ifndef SOT_WSORT_BUCKETS
#define SOT_WSORT_BUCKETS 1   /* 32 keys per lane: the distribution form first, the network when a bucket overflows (0: the network always) */
#endif
#ifndef SOT_WSORT_ONE_REGION
#define SOT_WSORT_ONE_REGION 1   /* the distribution form's counters and its image share their 8.25 KB of LDS (every base is read before a word is stored) */
#endif
#ifndef SOT_WSORT_FENCED
#define SOT_WSORT_FENCED 1
#endif
#if SOT_WSORT_FENCED
#define SOT_WSORT_STAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SOT_WSORT_STAGE_FENCE() do { } while (0)
#endif

namespace sot {

constexpr int kWaveSortRunLimit = 8;

__host__ __device__ constexpr int wsort_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
// LDS dwords the index / scratch array of a wave sort needs (the key array needs 64 KPL)
// (64 KPL keys: the skewed image of 66 KPL dwords; with `buckets` KPL = 32 also holds the 2048 bucket counters of the distribution form behind it, skewed alike)
__host__ __device__ constexpr int wave_sort_scratch(int kpl, bool buckets = false) { return ((kpl == 32 && buckets && !SOT_WSORT_ONE_REGION) ? 2 : 1) * (64 * kpl + 2 * kpl) + 2; }

// ---------------------------------------------------------------------------------------------
// Register <-> element maps.  VEC = false: register r of lane l is element (position) r 64 + l.  VEC = true (KPL % 4 == 0): element
// (r / 4) 256 + 4 l + r % 4 -- four consecutive elements per lane, i.e. 16-byte loads / stores in global memory and LDS.
// ---------------------------------------------------------------------------------------------
template <bool VEC>
__device__ __forceinline__ int wsort_elem(int r, int lane) { return VEC ? ((r >> 2) << 8) + 4 * lane + (r & 3) : r * 64 + lane; }

// value of lane ^ M
template <int M>
__device__ __forceinline__ uint32_t wsort_lane_xor(uint32_t v, uint32_t addr63)
{
    (void)addr63;
    if constexpr (((SOT_WSORT_DPP >> 1) & 1) && M == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (((SOT_WSORT_DPP >> 2) & 1) && M == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (((SOT_WSORT_DPP >> 3) & 1) && M == 3) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x1B, 0xF, 0xF, true);   // quad_perm [3,2,1,0]
    else if constexpr (((SOT_WSORT_DPP >> 7) & 1) && M == 7) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    else if constexpr (((SOT_WSORT_DPP >> 15) & 1) && M == 15) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true); // row_mirror
    else if constexpr (M < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (M << 10));                    // bit-mask mode: lane ^ M within 32
    else return (uint32_t)__builtin_amdgcn_ds_bpermute((int)addr63, (int)v);                                               // M == 63: addr63 = 4 (lane ^ 63)
}

// v_med3_u32 (the backend matches this shape; inline asm would make the hazard recogniser pad every DPP move behind it with s_nop)
__device__ __forceinline__ uint32_t wsort_med3(uint32_t a, uint32_t b, uint32_t c)
{
    return max(min(a, b), min(max(a, b), c));
}

// half cleaners between registers at distances J, J / 2, ..., 1
template <int KPL, int J>
__device__ __forceinline__ void wsort_reg_tail(uint32_t (&w)[KPL])
{
    if constexpr (J >= 1) {
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
            if ((r & J) == 0) {
                const uint32_t lo = min(w[r], w[r | J]), hi = max(w[r], w[r | J]);
                w[r] = lo; w[r | J] = hi;
            }
        }
        SOT_WSORT_STAGE_FENCE();
        wsort_reg_tail<KPL, J / 2>(w);
    }
}

// the lane's own KPL registers, sorted ascending: Batcher's odd-even merge sort (191 exchanges for 32 registers where the bitonic
// merges of the rest of the network would take 240; Knuth's iterative form, every loop bound a compile-time constant)
template <int KPL>
__device__ __forceinline__ void wsort_reg_sort(uint32_t (&w)[KPL])
{
#pragma unroll
    for (int p = 1; p < KPL; p *= 2) {
#pragma unroll
        for (int k = p; k >= 1; k /= 2) {
#pragma unroll
            for (int j = k % p; j + k < KPL; j += 2 * k) {
#pragma unroll
                for (int i = 0; i < k; ++i) {
                    const int a = i + j, b = i + j + k;
                    if (b < KPL && a / (2 * p) == b / (2 * p)) {
                        const uint32_t lo = min(w[a], w[b]), hi = max(w[a], w[b]);
                        w[a] = lo; w[b] = hi;
                    }
                }
            }
            SOT_WSORT_STAGE_FENCE();
        }
    }
}

// half cleaners between lanes at distances D, D / 2, ..., 1 (groups of SOT_WSORT_GROUP registers: that many moves in flight, no more
// temporaries than that)
template <int KPL, int D>
__device__ __forceinline__ void wsort_lane_tail(uint32_t (&w)[KPL], int lane, uint32_t addr63)
{
    if constexpr (D >= 1) {
        constexpr int GRP = KPL < SOT_WSORT_GROUP ? KPL : SOT_WSORT_GROUP;
        const uint32_t bound = (lane & D) ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int g = 0; g < KPL; g += GRP) {
            uint32_t t[GRP];
#pragma unroll
            for (int r = 0; r < GRP; ++r) t[r] = wsort_lane_xor<D>(w[g + r], addr63);
#pragma unroll
            for (int r = 0; r < GRP; ++r) w[g + r] = wsort_med3(w[g + r], t[r], bound);
            SOT_WSORT_STAGE_FENCE();
        }
        wsort_lane_tail<KPL, D / 2>(w, lane, addr63);
    }
}

// 32 registers x 64 lanes: register r of lane l <-> register (l & 31) of lane (l & 32) + r, through `scratch` (LDS byte address of
// 2112 dwords owned by this wavefront).  An involution: blocked layout (position = 32 l + r) -> transposed layout (lane L, register R
// hold position 32 (32 (L >> 5) + R) + (L & 31)) and back.  Position p lives at dword p + p / 32: the store (lanes 33 dwords apart) and
// the load (consecutive lanes on consecutive dwords) both touch 32 banks per half wave.
__device__ __forceinline__ void wsort_transpose32(uint32_t (&w)[32], int lane, uint32_t scratch)
{
    const uint32_t blocked = scratch + 4u * 33u * (uint32_t)lane;                                  // position 32 l + r     at 33 l + r
    const uint32_t crossed = scratch + 4u * (33u * 32u * (uint32_t)(lane >> 5) + (uint32_t)(lane & 31));   // position 32 (32 h + R) + c  at 33 (32 h + R) + c
    // (the caller's layout decides which map stores and which loads; both are bijections onto the same image, so a wave that stored with one and
    //  loads with the other has exchanged the two index fields -- twice is the identity)
#pragma unroll
    for (int r = 0; r < 32; ++r) lds_st_u32(blocked + 4u * (uint32_t)r, w[r]);
    row_sync<1>();
#pragma unroll
    for (int r = 0; r < 32; ++r) w[r] = lds_ld_u32(crossed + 4u * 33u * (uint32_t)r);
    row_sync<1>();
    SOT_WSORT_STAGE_FENCE();
}

// merges across lanes: runs of (S / 2) KPL -> S KPL, S = 2 ... 64
template <int KPL, int S>
__device__ __forceinline__ void wsort_lane_merges(uint32_t (&w)[KPL], int lane, uint32_t addr63, uint32_t scratch)
{
    if constexpr (S <= 64) {
        // flip: partner (lane ^ (S - 1), KPL - 1 - r); registers r and KPL - 1 - r only need each other: groups of pairs
        constexpr int GRP = (KPL / 2 < SOT_WSORT_GROUP / 2) ? (KPL / 2 > 0 ? KPL / 2 : 1) : SOT_WSORT_GROUP / 2;
        const uint32_t bound = (lane & (S / 2)) ? 0xFFFFFFFFu : 0u;
        if constexpr (KPL == 1) {
            w[0] = wsort_med3(w[0], wsort_lane_xor<S - 1>(w[0], addr63), bound);
        } else {
#pragma unroll
            for (int g = 0; g < KPL / 2; g += GRP) {
                uint32_t ta[GRP], tb[GRP];
#pragma unroll
                for (int r = 0; r < GRP; ++r) { ta[r] = wsort_lane_xor<S - 1>(w[KPL - 1 - (g + r)], addr63); tb[r] = wsort_lane_xor<S - 1>(w[g + r], addr63); }
#pragma unroll
                for (int r = 0; r < GRP; ++r) { w[g + r] = wsort_med3(w[g + r], ta[r], bound); w[KPL - 1 - (g + r)] = wsort_med3(w[KPL - 1 - (g + r)], tb[r], bound); }
                SOT_WSORT_STAGE_FENCE();
            }
        }
        if constexpr (KPL == 32 && S / 4 >= SOT_WSORT_TRANSPOSE_MIN) {
            // The half cleaners between lanes at distances S / 4 ... 1 (lane bits 4 ... 0) as exchanges between REGISTERS: the lane's low five
            // bits and the register index change places through the skewed LDS image (one store + one load per key each way, immediate
            // offsets, bank-conflict-free both ways), the stages run as one VALU instruction per key instead of a move + a v_med3, and the
            // image is read back in the blocked layout.  Pays from three stages on: 2 x 3 ... 5 VALU per key against 4 LDS operations.
            wsort_transpose32(w, lane, scratch);
            wsort_reg_tail<KPL, S / 4>(w);
            wsort_transpose32(w, lane, scratch);
        } else {
            wsort_lane_tail<KPL, S / 4>(w, lane, addr63);
        }
        wsort_reg_tail<KPL, KPL / 2>(w);
        wsort_lane_merges<KPL, 2 * S>(w, lane, addr63, scratch);
    }
}

// the whole network: w[r] of lane l is position l KPL + r afterwards
template <int KPL>
__device__ __forceinline__ void wsort_network(uint32_t (&w)[KPL], int lane, uint32_t scratch)
{
    const uint32_t addr63 = (uint32_t)(lane ^ 63) << 2;
    wsort_reg_sort<KPL>(w);
    wsort_lane_merges<KPL, 2>(w, lane, addr63, scratch);
}

// ---------------------------------------------------------------------------------------------
// The DISTRIBUTION form for 32 keys per lane (round 6, second form): the network above is 72 VALU instructions per key whatever the keys
// are; packed words of generic positions are almost uniform in their top bits (that is what the row-adaptive quantisation is for), so ONE
// counting pass puts every word within a few positions of its place:
//   bucket = word >> 21 (2048 equal bins over the row's range) -> histogram by LDS atomics (the value an atomic returns is the word's rank
//   inside its bucket, in arrival order) -> exclusive scan (lane l owns buckets 32 l ... 32 l + 31: one read, 32 adds, one DPP wave scan)
//   -> scatter to base + rank in the skewed image.
// Then two passes of the 32-register Batcher network on windows of the image -- [32 l, 32 l + 32) and [32 l + 16, 32 l + 48) -- order every
// bucket of at most kWaveSortBucketLimit = 16 words completely (such a bucket lies inside a window of one of the passes, and a window sort
// keeps the buckets it holds only partly in their places), whatever order the atomics arrived in: the result does not depend on it.
// ~45 VALU + 13 LDS operations per key.  A bucket over the limit (clustered positions): returns false with w[] untouched -- the caller runs
// the network.  image / counters: LDS byte addresses of 2112 dwords each, owned by this wavefront.
// ---------------------------------------------------------------------------------------------
constexpr int kWaveSortBucketLimit = 16;

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t wsort_dpp_zero(uint32_t v)   // the DPP source lane's value; 0 where there is none
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}

typedef __attribute__((address_space(3))) uint32_t wsort_lds_u32;

template <bool FULL, bool VEC>
__device__ __forceinline__ bool wsort_bucket_sort32(uint32_t (&w)[32], int lane, int n, uint32_t image, uint32_t counters)
{
    // (Measured, 4096 x 2 x 2048 keys: this form 45.6 us at two waves per SIMD = 8 waves per CU, 60.4 us at 4 per CU -- the pass lives on
    //  occupancy; 16-bit counters packed in pairs (12.7 instead of 16.9 KB per wave) cost 3.5 us of unpacking at the same occupancy, 48.8-49.1 us,
    //  and their twelve waves per CU need <= 168 registers, which this function does not fit: 65.7 us with 76 dwords spilled.)
    const uint32_t own = counters + 4u * 33u * (uint32_t)lane;          // this lane's 32 buckets (bucket b at dword b + b / 32)
#pragma unroll
    for (int j = 0; j < 32; ++j) lds_st_u32(own + 4u * (uint32_t)j, 0u);
    row_sync<1>();
    auto slot_of = [counters](uint32_t word) { const uint32_t b = word >> 21; return counters + 4u * (b + (b >> 5)); };   // (computed twice per word rather than kept)
    uint32_t rank[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        rank[r] = 0u;
        if (FULL || w[r] != 0xFFFFFFFFu)        // (pads are not counted: they keep the positions behind the data)
            rank[r] = __hip_atomic_fetch_add(reinterpret_cast<wsort_lds_u32*>((uintptr_t)slot_of(w[r])), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    row_sync<1>();
    uint32_t c[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) c[j] = lds_ld_u32(own + 4u * (uint32_t)j);
    uint32_t most = 0u, run = 0u;
#pragma unroll
    for (int j = 0; j < 32; ++j) { most = max(most, c[j]); const uint32_t t = c[j]; c[j] = run; run += t; }
    if (__builtin_amdgcn_ballot_w64(most > (uint32_t)kWaveSortBucketLimit) != 0ull) return false;   // wave-uniform; w[] untouched
    uint32_t incl = run;                                                  // inclusive scan of the lanes' totals
    incl += wsort_dpp_zero<kRowShr1>(incl); incl += wsort_dpp_zero<kRowShr2>(incl);
    incl += wsort_dpp_zero<kRowShr4>(incl); incl += wsort_dpp_zero<kRowShr8>(incl);
    incl += wsort_dpp_zero<kRowBcast15, 0xA>(incl); incl += wsort_dpp_zero<kRowBcast31, 0xC>(incl);
    const uint32_t before = incl - run;
#pragma unroll
    for (int j = 0; j < 32; ++j) lds_st_u32(own + 4u * (uint32_t)j, c[j] + before);
    row_sync<1>();
    // every word's position first (the bases are read before anything is stored): image and counters may then be ONE region (SOT_WSORT_ONE_REGION)
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const uint32_t base = lds_ld_u32(slot_of(w[r]));
        rank[r] = (FULL || w[r] != 0xFFFFFFFFu) ? base + rank[r] : (uint32_t)wsort_elem<VEC>(r, lane);   // a pad: its own element number (>= n)
    }
    row_sync<1>();
#pragma unroll
    for (int r = 0; r < 32; ++r) lds_st_u32(image + 4u * (rank[r] + (rank[r] >> 5)), w[r]);
    row_sync<1>();
    (void)n;
    // two window passes: [32 l, 32 l + 32), then [32 l + 16, 32 l + 48) (lane 63 has no second window)
    const uint32_t blocked = image + 4u * 33u * (uint32_t)lane;
    uint32_t v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = lds_ld_u32(blocked + 4u * (uint32_t)j);
    wsort_reg_sort<32>(v);
#pragma unroll
    for (int j = 0; j < 32; ++j) lds_st_u32(blocked + 4u * (uint32_t)j, v[j]);
    row_sync<1>();
    const uint32_t shifted = blocked + 4u * 16u;                        // position 32 l + 16 + j at 33 l + 16 + j (+ 1 from j = 16 on: the next block's skew)
    if (lane < 63) {
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = lds_ld_u32(shifted + 4u * (uint32_t)(j + (j >= 16 ? 1 : 0)));
    }
    wsort_reg_sort<32>(v);
    if (lane < 63) {
#pragma unroll
        for (int j = 0; j < 32; ++j) lds_st_u32(shifted + 4u * (uint32_t)(j + (j >= 16 ? 1 : 0)), v[j]);
    }
    row_sync<1>();
#pragma unroll
    for (int j = 0; j < 32; ++j) w[j] = lds_ld_u32(blocked + 4u * (uint32_t)j);
    return true;
}

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float wsort_dpp_keep(float v)   // the DPP source lane's value; a lane without one reads its own
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wsort_wave_min(float v)
{
    v = fminf(v, wsort_dpp_keep<kRowShr1>(v)); v = fminf(v, wsort_dpp_keep<kRowShr2>(v));
    v = fminf(v, wsort_dpp_keep<kRowShr4>(v)); v = fminf(v, wsort_dpp_keep<kRowShr8>(v));
    v = fminf(v, wsort_dpp_keep<kRowBcast15, 0xA>(v)); v = fminf(v, wsort_dpp_keep<kRowBcast31, 0xC>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wsort_wave_max(float v)
{
    v = fmaxf(v, wsort_dpp_keep<kRowShr1>(v)); v = fmaxf(v, wsort_dpp_keep<kRowShr2>(v));
    v = fmaxf(v, wsort_dpp_keep<kRowShr4>(v)); v = fmaxf(v, wsort_dpp_keep<kRowShr8>(v));
    v = fmaxf(v, wsort_dpp_keep<kRowBcast15, 0xA>(v)); v = fmaxf(v, wsort_dpp_keep<kRowBcast31, 0xC>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---------------------------------------------------------------------------------------------
// x[r]: the key of element wsort_elem<VEC>(r, lane) (any value for e >= n); key[0 .. 64 KPL): the same keys in LDS in natural order
// with +inf behind the n real ones; idx: LDS scratch of wave_sort_scratch(KPL) dwords.  All 64 lanes of ONE wavefront call it; nobody
// else touches key / idx meanwhile.  FULL: n == 64 KPL is known at compile time (no validity tests).  Returns (wave-uniform) true:
// ok[r] / oi[r] are the sorted key and its original index at position wsort_elem<VEC>(r, lane) (pads: +inf / 64 KPL - 1) and, with
// STORE_LDS, key[] / idx[] hold the same in natural order on [0, 64 KPL); false: declined, key[] is untouched (idx[] is not).
// ---------------------------------------------------------------------------------------------
// wave_sort_core: the same with the full keys behind a functor (keyof(i): the key of original element i -- LDS, or global memory when no
// natural copy is kept) and WANT_KEYS = false for callers that only want the permutation (ok[] is then not written).
template <int KPL, bool STORE_LDS, bool FULL, bool VEC, bool WANT_KEYS, bool BUCKETS = false, typename KeyOf>
__device__ __forceinline__ bool wave_sort_core(const float (&x)[KPL], KeyOf keyof, float* key, uint32_t* idx, int n, int lane, float (&ok)[KPL], uint32_t (&oi)[KPL])
{
    static_assert(!STORE_LDS || WANT_KEYS, "the natural-order stores need the sorted keys");
    static_assert(!VEC || KPL % 4 == 0, "VEC needs four registers per 16-byte group");
    constexpr int NPAD = 64 * KPL, IDXBITS = 6 + wsort_ilog2(KPL), QBITS = 32 - IDXBITS;
    constexpr uint32_t QMAX = (1u << QBITS) - 1u, MASK = (1u << IDXBITS) - 1u;
    // ---- pre-pass: range of the real keys; NaN / infinite keys decline (0 x = NaN for both)
    // (pads without compare masks -- 32 SGPR pairs the allocator would have to keep: pm = all ones where element >= n, by sign extension of
    // n - 1 - element; a pad's key is replaced by the wave's first key for the range and its word is OR-ed to all ones)
    float mn = INFINITY, mx = -INFINITY, det = 0.0f;
    const bool full = FULL || (n == NPAD);    // wave-uniform: no pads, no validity tests
    uint32_t low[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) low[c] = VEC ? (uint32_t)(4 * lane + c) : (uint32_t)lane;
    auto pad_mask = [&](int r) -> uint32_t {   // element(r) = base(r) + low: base is a compile-time constant
        const int base = VEC ? ((r >> 2) << 8) : r * 64;
        return (uint32_t)((n - 1 - base - (int)low[VEC ? (r & 3) : 0]) >> 31);
    };
    if (full) {
#pragma unroll
        for (int r = 0; r < KPL; ++r) { mn = fminf(mn, x[r]); mx = fmaxf(mx, x[r]); det = fmaf(x[r], 0.0f, det); }
    } else {
        const uint32_t x0 = (uint32_t)__builtin_amdgcn_readfirstlane(__float_as_int(x[0]));   // element 0 (n >= 1): a real key
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
            const uint32_t pm = pad_mask(r);
            const float xe = __uint_as_float((__float_as_uint(x[r]) & ~pm) | (x0 & pm));   // v_bfi_b32
            mn = fminf(mn, xe); mx = fmaxf(mx, xe); det = fmaf(xe, 0.0f, det);
        }
    }
    mn = wsort_wave_min(mn); mx = wsort_wave_max(mx);
    const float range = mx - mn;
    const float scale = (float)(QMAX - 8u) / range;
    const bool bad = __builtin_amdgcn_ballot_w64(det != det) != 0ull;
    if (bad || !(range > 0.0f) || !(range < INFINITY) || !(scale < INFINITY)) return false;
    // ---- one word per key: (q << IDXBITS) + element, in two shift-adds with inline constants
    uint32_t w[KPL];
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        const uint32_t q = (uint32_t)((x[r] - mn) * scale);   // (a pad's q is garbage: its word is overwritten below)
        const uint32_t word = VEC ? (((q << (IDXBITS - 8)) + (uint32_t)(r >> 2)) << 8) + low[r & 3]
                                  : (((q << (IDXBITS - 6)) + (uint32_t)r) << 6) + low[0];
        w[r] = full ? word : (word | pad_mask(r));   // every pad is the same word 0xFFFFFFFF: behind the data, never a run
    }
    bool placed = false;
#if !defined(SOT_WSORT_DIAG_SKIP_NETWORK)   /* diagnostic (timing only, results are wrong): everything but the network */
    if constexpr (KPL == 32 && BUCKETS && SOT_WSORT_BUCKETS) placed = wsort_bucket_sort32<FULL, VEC>(w, lane, n, lds_addr(idx), lds_addr(idx) + (SOT_WSORT_ONE_REGION ? 0u : 4u * 2112u));
    if (!placed) wsort_network<KPL>(w, lane, lds_addr(idx));
#endif
    // ---- neighbours that share q: (a ^ b) - 1 < MASK (a == b: two pads)
    uint32_t cmin = 0xFFFFFFFFu;
#pragma unroll
    for (int r = 0; r + 1 < KPL; ++r) cmin = min(cmin, (w[r] ^ w[r + 1]) - 1u);
    // the first word of the next lane (lane 63 reads its own last word: xor 0, no collision)
    const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp((int)w[KPL - 1], (int)w[0], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
    cmin = min(cmin, (w[KPL - 1] ^ nxt) - 1u);
    const bool any_run = __builtin_amdgcn_ballot_w64(cmin < MASK) != 0ull;
    uint32_t cm = 0;                           // bit r: positions lane KPL + r and + r + 1 share q (only built when the wave has a run)
    if (any_run) {
#pragma unroll
        for (int r = 0; r + 1 < KPL; ++r) cm |= (((w[r] ^ w[r + 1]) - 1u) < MASK) ? (1u << r) : 0u;
        cm |= (((w[KPL - 1] ^ nxt) - 1u) < MASK) ? (1u << (KPL - 1)) : 0u;
    }
    // ---- blocked -> striped through the skewed image: position p lives at p + p / 32
    const uint32_t sbase = lds_addr(idx);
    if (!placed) {   // (the distribution form has left exactly this image behind)
        const uint32_t p0 = (uint32_t)(lane * KPL);
        const uint32_t wa = sbase + 4u * (p0 + (p0 >> 5));          // KPL <= 32: p0 + r never crosses a multiple of 32 inside one lane's block
#pragma unroll
        for (int r = 0; r < KPL; ++r) lds_st_u32(wa + 4u * r, w[r]);
    }
    row_sync<1>();
    if (any_run) {
        // the exact order inside runs of equal q, in LDS on the skewed image: the lane whose block holds a run's first position sorts it
        bool over = false;
        auto W = [&](int p) -> uint32_t& { return idx[p + (p >> 5)]; };
        while (cm != 0u) {
            const int r = __builtin_ctz(cm);
            const int p = lane * KPL + r;
            const uint32_t wp = W(p);
            const bool start = (p == 0) || !(((W(max(p - 1, 0)) ^ wp) - 1u) < MASK);
            int e = p + 1;
            while (e + 1 < NPAD && e - p < kWaveSortRunLimit && (((W(e + 1) ^ wp) - 1u) < MASK)) ++e;
            const int len = e - p + 1;
            if (start) {
                if (len > kWaveSortRunLimit) {
                    over = true;
                } else {
                    for (int i = 1; i < len; ++i) {        // insertion sort, strict ">": stable
                        const uint32_t wi = W(p + i);
                        const uint32_t ki = float_order_bits(keyof(wi & MASK));
                        int j = i;
                        while (j > 0) {
                            const uint32_t wj = W(p + j - 1);
                            if (float_order_bits(keyof(wj & MASK)) > ki) { W(p + j) = wj; --j; } else break;
                        }
                        W(p + j) = wi;
                    }
                }
            }
            cm &= ~(((1u << (len - 1)) - 1u) << r);          // the pairs of this run inside this lane's block are done (len - 1 <= 8)
        }
        row_sync<1>();
        if (__builtin_amdgcn_ballot_w64(over) != 0ull) return false;
    }
    // position p = wsort_elem<VEC>(r, lane) at p + p / 32: one base per lane, immediate offsets, every 32-lane half on 32 banks
    const uint32_t ra = sbase + 4u * (uint32_t)(VEC ? 4 * lane + (lane >> 3) : lane + (lane >> 5));
#pragma unroll
    for (int r = 0; r < KPL; ++r) w[r] = lds_ld_u32(ra + 4u * (uint32_t)(VEC ? 264 * (r >> 2) + (r & 3) : 66 * r));
    // ---- indices and sorted keys (every gather is issued before the first store: one wave, in-order LDS)
#pragma unroll
    for (int r = 0; r < KPL; ++r) { oi[r] = w[r] & MASK; if constexpr (WANT_KEYS) ok[r] = keyof(oi[r]); }
    if constexpr (STORE_LDS) {
        row_sync<1>();
        if constexpr (VEC) {
#pragma unroll
            for (int r = 0; r < KPL; r += 4) {
                *reinterpret_cast<float4*>(key + wsort_elem<true>(r, lane)) = make_float4(ok[r], ok[r + 1], ok[r + 2], ok[r + 3]);
                *reinterpret_cast<uint4*>(idx + wsort_elem<true>(r, lane)) = make_uint4(oi[r], oi[r + 1], oi[r + 2], oi[r + 3]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < KPL; ++r) { key[r * 64 + lane] = ok[r]; idx[r * 64 + lane] = oi[r]; }
        }
    }
    return true;
}

template <int KPL, bool STORE_LDS = true, bool FULL = false, bool VEC = false>
__device__ __forceinline__ bool wave_sort_kv(const float (&x)[KPL], float* key, uint32_t* idx, int n, int lane, float (&ok)[KPL], uint32_t (&oi)[KPL])
{
    return wave_sort_core<KPL, STORE_LDS, FULL, VEC, true, SOT_WSORT_ONE_REGION != 0>(x, [key](uint32_t i) { return key[i]; }, key, idx, n, lane, ok, oi);   // (one region: the scratch is no larger with the distribution form)
}

}  // namespace sot
