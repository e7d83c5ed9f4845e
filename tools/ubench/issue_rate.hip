// issue_rate.hip -- gfx950 micro-benchmarks behind DESIGN.md's cost model (not part of the product):
//   (1) cycles per wave64 VALU instruction on one SIMD at 1 / 2 / 4 / 8 resident waves per SIMD, for the instruction kinds the
//       SOT kernels are made of (fp32 fma, fp64 add, v_cndmask, DPP move, fp32<->fp64 converts, packed fp32 fma);
//   (2) LDS wave-instruction cost of ds_read_b32 / ds_read_b64 / ds_bpermute_b32 with lane addresses that are sequential,
//       random (the merge walk's pattern), or strided.
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITER = 256;      // loop trips
constexpr int UNROLL = 32;     // instructions per trip and per accumulator group

template <int KIND>
__global__ __launch_bounds__(256) void valu_kernel(float* out, unsigned long long* cyc, const int* idx_tab, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    const float m = 1.0000001f, c = 1e-9f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const f2 pm = {m, m}, pc = {c, c};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL / 8; ++u) {
            if (KIND == 0) {  // v_fma_f32, 8 independent chains
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (KIND == 1) {  // v_add_f64, 4 chains x 2
                d0 += 1e-9; d1 += 1e-9; d2 += 1e-9; d3 += 1e-9; d0 += 1e-9; d1 += 1e-9; d2 += 1e-9; d3 += 1e-9;
            } else if (KIND == 2) {  // v_cndmask (select on a compare kept outside)
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n"
                             "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %4, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
            } else if (KIND == 3) {  // DPP row_shr:1 move
                asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 4) {  // v_cvt_f64_f32 + v_cvt_f32_f64 pairs
                d0 = (double)a0; a1 = (float)d1; d2 = (double)a2; a3 = (float)d3; d1 = (double)a4; a5 = (float)d0; d3 = (double)a6; a7 = (float)d2;
            } else if (KIND == 5) {  // v_pk_fma_f32: 4 chains x 2
                p0 = __builtin_elementwise_fma(p0, pm, pc); p1 = __builtin_elementwise_fma(p1, pm, pc);
                p2 = __builtin_elementwise_fma(p2, pm, pc); p3 = __builtin_elementwise_fma(p3, pm, pc);
                p0 = __builtin_elementwise_fma(p0, pm, pc); p1 = __builtin_elementwise_fma(p1, pm, pc);
                p2 = __builtin_elementwise_fma(p2, pm, pc); p3 = __builtin_elementwise_fma(p3, pm, pc);
            } else if (KIND == 6) {  // v_add_u32
                asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
                             "v_add_u32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_add_u32 %7, %7, %4\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 7) {  // v_cmp_le_f32 + v_cndmask pairs (the walk's pattern)
                asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_le_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
                             "v_cmp_le_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n v_cmp_le_f32 vcc, %5, %4\n v_cndmask_b32 %7, %7, %6, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (r == 12345.678f) out[0] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// LDS access patterns: every lane reads ITER*UNROLL times from a per-lane address sequence
template <int KIND, int PATTERN>
__global__ __launch_bounds__(256) void lds_kernel(float* out, unsigned long long* cyc, const int* idx_tab, float seed)
{
    __shared__ __attribute__((aligned(16))) float buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = (float)i;
    const int lane = threadIdx.x & 63;
    int idx[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (PATTERN == 0) idx[k] = (lane + 64 * k) & 4095;                         // sequential
        else if (PATTERN == 1) idx[k] = idx_tab[(threadIdx.x * 8 + k) & 4095] & 4095; // random
        else if (PATTERN == 2) idx[k] = (lane * 17 / 2 + k * 3) & 4095;              // the walk's nominal stride 8.5
        else idx[k] = (lane * 8 + k) & 4095;                                       // stride 8 (owner pattern)
        if (KIND == 1) idx[k] &= ~1;                                               // b64: 8-B aligned
    }
    __syncthreads();
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL / 8; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (KIND == 0) acc += buf[idx[k]];
                else if (KIND == 1) { const float2 v = reinterpret_cast<const float2*>(buf)[idx[k] >> 1]; acc += v.x + v.y; }
                else acc += __int_as_float(__builtin_amdgcn_ds_bpermute((idx[k] & 63) << 2, __float_as_int(acc + (float)k)));
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) idx[k] = (idx[k] + ((KIND == 1) ? 2 : 1) * (PATTERN == 1 ? ((it + k) & 1) : 0)) & 4095;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 12345.678f) out[0] = acc;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <typename K>
static void run(const char* name, K kern, int blocks_per_cu, float* out, unsigned long long* cyc, const int* tab, bool lds)
{
    const int grid = 256 * blocks_per_cu;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, cyc, tab, 1.0f);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(grid * 4);
    CK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= h.size();
    const double n_inst = (double)ITER * UNROLL;
    // per-SIMD cycles per wave-instruction = wave-lifetime cycles * (1 / instructions) / waves per SIMD
    printf("%-28s waves/SIMD %d: %8.1f cycles per wave (memtime), %6.2f cyc/instr/wave, %6.2f cyc/instr per SIMD; wall %.1f us\n", name,
           blocks_per_cu, avg, avg / n_inst, avg / n_inst / blocks_per_cu, ms * 1e3);
}

template <typename T> struct Wrap { T k; };

int main()
{
    float* out; unsigned long long* cyc; int* tab;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 8 * 4)); CK(hipMalloc(&tab, 4096 * 4));
    std::vector<int> h(4096);
    srand(1);
    for (auto& v : h) v = rand();
    CK(hipMemcpy(tab, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    const char* vn[] = {"v_fma_f32", "v_add_f64", "v_cndmask_b32", "v_mov_dpp row_shr", "v_cvt f32<->f64", "v_pk_fma_f32", "v_add_u32", "v_cmp+v_cndmask"};
    for (int bpc : {1, 2, 4, 8}) {
        run(vn[0], valu_kernel<0>, bpc, out, cyc, tab, false);
        run(vn[1], valu_kernel<1>, bpc, out, cyc, tab, false);
        run(vn[2], valu_kernel<2>, bpc, out, cyc, tab, false);
        run(vn[3], valu_kernel<3>, bpc, out, cyc, tab, false);
        run(vn[4], valu_kernel<4>, bpc, out, cyc, tab, false);
        run(vn[5], valu_kernel<5>, bpc, out, cyc, tab, false);
        run(vn[6], valu_kernel<6>, bpc, out, cyc, tab, false);
        run(vn[7], valu_kernel<7>, bpc, out, cyc, tab, false);
    }
    for (int bpc : {1, 4}) {
        run("ds_read_b32 sequential", lds_kernel<0, 0>, bpc, out, cyc, tab, true);
        run("ds_read_b32 random", lds_kernel<0, 1>, bpc, out, cyc, tab, true);
        run("ds_read_b32 stride 8.5", lds_kernel<0, 2>, bpc, out, cyc, tab, true);
        run("ds_read_b32 stride 8", lds_kernel<0, 3>, bpc, out, cyc, tab, true);
        run("ds_read_b64 sequential", lds_kernel<1, 0>, bpc, out, cyc, tab, true);
        run("ds_read_b64 random", lds_kernel<1, 1>, bpc, out, cyc, tab, true);
        run("ds_bpermute sequential", lds_kernel<2, 0>, bpc, out, cyc, tab, true);
        run("ds_bpermute random", lds_kernel<2, 1>, bpc, out, cyc, tab, true);
    }
    return 0;
}
