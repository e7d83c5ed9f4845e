// row_stream.hip -- how fast can the SOT kernels' ACCESS SHAPE stream from HBM?  A persistent grid of 256-thread workgroups,
// each fetching one [x row | y row] pair of 2 x 8 KB per iteration with four 16-B loads per thread (the staging loads of
// sot_forward_full.inc), summing them and writing 4 bytes per row -- no LDS, no other arithmetic.  Variants: loads of the next row
// issued before the current one is consumed (depth 1, what the kernels do) or two rows ahead (depth 2); 4 ... 8 workgroups per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o row_stream row_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(256) void stream_rows(const float4* __restrict__ x, const float4* __restrict__ y, float* __restrict__ out, int B)
{
    const int t = threadIdx.x;
    float4 bx[DEPTH][2], by[DEPTH][2];
    const int step = gridDim.x;
    int row = blockIdx.x;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const int r = min(row + d * step, B - 1);
        bx[d][0] = x[(size_t)r * 512 + t]; bx[d][1] = x[(size_t)r * 512 + 256 + t];
        by[d][0] = y[(size_t)r * 512 + t]; by[d][1] = y[(size_t)r * 512 + 256 + t];
    }
    for (; row < B; row += DEPTH * step) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int cur = row + d * step;
            float s = bx[d][0].x + bx[d][0].y + bx[d][0].z + bx[d][0].w + bx[d][1].x + bx[d][1].y + bx[d][1].z + bx[d][1].w +
                      by[d][0].x + by[d][0].y + by[d][0].z + by[d][0].w + by[d][1].x + by[d][1].y + by[d][1].z + by[d][1].w;
            const int nxt = min(cur + DEPTH * step, B - 1);
            bx[d][0] = x[(size_t)nxt * 512 + t]; bx[d][1] = x[(size_t)nxt * 512 + 256 + t];
            by[d][0] = y[(size_t)nxt * 512 + t]; by[d][1] = y[(size_t)nxt * 512 + 256 + t];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
            if (cur < B && (t & 63) == 0) atomicAdd(&out[cur], s);
        }
    }
}

int main()
{
    const int B = 8192, SETS = 6;
    float4 *x[SETS], *y[SETS];
    float* out;
    for (int i = 0; i < SETS; ++i) { CK(hipMalloc(&x[i], (size_t)B * 8192)); CK(hipMalloc(&y[i], (size_t)B * 8192)); CK(hipMemset(x[i], 0, (size_t)B * 8192)); CK(hipMemset(y[i], 0, (size_t)B * 8192)); }
    CK(hipMalloc(&out, B * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int per_cu : {4, 5, 6, 8}) {
        for (int depth : {1, 2}) {
            const int grid = 256 * per_cu;
            auto launch = [&](int i) {
                if (depth == 1) hipLaunchKernelGGL(stream_rows<1>, dim3(grid), dim3(256), 0, 0, x[i % SETS], y[i % SETS], out, B);
                else hipLaunchKernelGGL(stream_rows<2>, dim3(grid), dim3(256), 0, 0, x[i % SETS], y[i % SETS], out, B);
            };
            for (int i = 0; i < 300; ++i) launch(i);
            CK(hipEventRecord(a));
            for (int i = 0; i < 300; ++i) launch(i);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            const double us = ms / 300 * 1e3;
            printf("workgroups per CU %d, rows prefetched ahead %d: %.2f us per 8192 x 2048 pair set = %.2f TB/s\n", per_cu, depth, us, 134.25e6 / us / 1e6);
        }
    }
    return 0;
}
