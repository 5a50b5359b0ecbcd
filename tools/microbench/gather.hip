// Microbenchmark: cost of dependent random record fetches per lane on gfx950, for several record layouts and
// lane-cooperation patterns. Guides the node layout of the BVH traversal kernel.
//   hipcc -O3 --offload-arch=gfx950 gather.hip -o gather && ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__device__ inline uint32_t quadBroadcast(uint32_t v, int j) {
    switch (j) {
        case 0: return (uint32_t) __builtin_amdgcn_mov_dpp((int) v, 0x00, 0xf, 0xf, true);   // quad_perm [0,0,0,0]
        case 1: return (uint32_t) __builtin_amdgcn_mov_dpp((int) v, 0x55, 0xf, 0xf, true);   // [1,1,1,1]
        case 2: return (uint32_t) __builtin_amdgcn_mov_dpp((int) v, 0xaa, 0xf, 0xf, true);
        default: return (uint32_t) __builtin_amdgcn_mov_dpp((int) v, 0xff, 0xf, 0xf, true);
    }
}

// MODE 0: own record, NF float4 loads per lane (stride STRIDE float4)
// MODE 1: quad-cooperative: step j, lane k loads float4 k of quad-lane j's record (first 4 float4), then own float4 #4.. if NF > 4
// MODE 2: 8-lane cooperative: step j (8 steps), lane k loads float4 k of group-lane j's record (8 float4 = 128 B)
template <int MODE, int NF, int STRIDE>
__global__ void __launch_bounds__(256) k_gather(const float4 *table, uint32_t nRecords, uint32_t iters, float *out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t state = hash32(tid * 2654435761u + 12345u);
    float acc = 0;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t idx = state % nRecords;
        float4 v[8];
        if (MODE == 0) {
            const float4 *p = table + (size_t) idx * STRIDE;
#pragma unroll
            for (int k = 0; k < NF; ++k) v[k] = p[k];
#pragma unroll
            for (int k = 0; k < NF; ++k) acc += v[k].x + v[k].w;
            state = hash32(state + __float_as_uint(v[0].y));
        } else if (MODE == 1) {
            const uint32_t k = lane & 3u;
            float4 w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t other = quadBroadcast(idx, j);
                w[j] = table[(size_t) other * STRIDE + k];
            }
#pragma unroll
            for (int k2 = 4; k2 < NF; ++k2) v[k2] = table[(size_t) idx * STRIDE + k2];
            // hand the pieces back: lane i needs w[i] of lanes 0..3 (sum stands in for the transpose; the dependency is what matters)
            float s = 0, first = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = __uint_as_float(quadBroadcast(__float_as_uint(w[k == (uint32_t) j ? 0 : 0].x), j));    // placeholder use of DPP traffic
                s += a + w[j].x + w[j].w;
                if ((lane & 3u) == (uint32_t) j) first = w[j].y;
            }
#pragma unroll
            for (int k2 = 4; k2 < NF; ++k2) s += v[k2].x;
            acc += s;
            state = hash32(state + __float_as_uint(first));
        } else {
            const uint32_t k = lane & 7u;
            float s = 0, first = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t other = (uint32_t) __shfl((int) idx, (int) ((lane & ~7u) + j));
                const float4 w = table[(size_t) other * STRIDE + k];
                s += w.x + w.w;
                if (k == (uint32_t) j) first = w.y;
            }
            acc += s;
            state = hash32(state + __float_as_uint(first));
        }
    }
    out[tid] = acc;
}

template <int MODE, int NF, int STRIDE>
void run(const char *name, const float4 *table, size_t tableBytes, float *out, int blocks) {
    const uint32_t nRecords = (uint32_t) (tableBytes / (STRIDE * 16));
    const uint32_t iters = 200;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_gather<MODE, NF, STRIDE>), dim3(blocks), dim3(256), 0, 0, table, nRecords, 20u, out);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_gather<MODE, NF, STRIDE>), dim3(blocks), dim3(256), 0, 0, table, nRecords, iters, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double fetches = (double) blocks * 256 * iters;
    printf("%-44s table %6.1f MB: %7.3f ms  %6.2f Gfetch/s  %6.1f GB/s useful\n", name, tableBytes / 1e6, ms, fetches / ms / 1e6, fetches * NF * 16 / ms / 1e6);
}

int main() {
    const size_t maxBytes = 256u << 20;
    float4 *table; float *out;
    CHECK(hipMalloc(&table, maxBytes));
    std::vector<float> h(maxBytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float) (i * 2654435761u >> 8) * 1e-7f;
    CHECK(hipMemcpy(table, h.data(), maxBytes, hipMemcpyHostToDevice));
    const int blocks = 256 * 6 * 4;
    CHECK(hipMalloc(&out, (size_t) blocks * 256 * 4));
    for (size_t mb : { (size_t) 2, (size_t) 8, (size_t) 56 }) {
        const size_t bytes = mb << 20;
        run<0, 5, 5>("own record 5xfloat4, stride 80 B", table, bytes, out, blocks);
        run<0, 5, 8>("own record 5xfloat4, stride 128 B", table, bytes, out, blocks);
        run<0, 4, 4>("own record 4xfloat4, stride 64 B", table, bytes, out, blocks);
        run<0, 3, 3>("own record 3xfloat4, stride 48 B", table, bytes, out, blocks);
        run<0, 3, 4>("own record 3xfloat4, stride 64 B", table, bytes, out, blocks);
        run<0, 1, 1>("own record 1xfloat4, stride 16 B", table, bytes, out, blocks);
        run<0, 6, 6>("own record 6xfloat4, stride 96 B", table, bytes, out, blocks);
        run<1, 5, 5>("quad coop 4 + own 1, stride 80 B", table, bytes, out, blocks);
        run<1, 5, 8>("quad coop 4 + own 1, stride 128 B", table, bytes, out, blocks);
        run<1, 4, 4>("quad coop 4, stride 64 B", table, bytes, out, blocks);
        run<2, 8, 8>("8-lane coop 8xfloat4, stride 128 B", table, bytes, out, blocks);
    }
    return 0;
}
