// Does reading only the first 64 bytes of random 128-byte-aligned records cost half of reading all
// 128 bytes (the shade step's surface records: 37 MB at the headline scene, far beyond an XCD's L2)?
// Every lane gathers random records with two gathers in flight; 32 waves per compute unit.
//   A: all 128 bytes of a record (8 loads of 16 bytes)        B: the first 64 bytes (4 loads), stride 128
//   C: 64-byte records at stride 64 (a table of half the size)
//   hipcc -O2 --offload-arch=gfx950 tools/gather_half_line.hip -o gather_half_line && ./gather_half_line
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int QUARTERS, int STRIDE_Q>
__global__ void __launch_bounds__(256) k(const float4* table, uint32_t mask, int iters, float* out) {
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        const float4* p = table + static_cast<size_t>((s >> 8) & mask) * STRIDE_Q;
        s = s * 1664525u + 1013904223u;
        const float4* q = table + static_cast<size_t>((s >> 8) & mask) * STRIDE_Q;
        float4 v[QUARTERS], w[QUARTERS];
#pragma unroll
        for (int k2 = 0; k2 < QUARTERS; ++k2) v[k2] = p[k2];
#pragma unroll
        for (int k2 = 0; k2 < QUARTERS; ++k2) w[k2] = q[k2];
#pragma unroll
        for (int k2 = 0; k2 < QUARTERS; ++k2) acc += v[k2].x + w[k2].w;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int QUARTERS, int STRIDE_Q>
void run(const char* name, const float4* table, uint32_t records, float* out, int grid) {
    const int iters = 400;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k<QUARTERS, STRIDE_Q>), dim3(grid), dim3(256), 0, 0, table, records - 1, 20, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<QUARTERS, STRIDE_Q>), dim3(grid), dim3(256), 0, 0, table, records - 1, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double n = 2.0 * iters * grid * 256.0;
    std::printf("%-44s %8.3f ms  %7.2f G records/s  %7.2f TB/s of record bytes read\n", name, ms, n / ms * 1e-6, n * QUARTERS * 16 / ms * 1e-9);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount * 8;
    for (uint32_t log_records : {18u, 21u}) {  // 32 MiB and 256 MiB of 128-byte records
        const uint32_t records = 1u << log_records;
        float4* table = nullptr;
        float* out = nullptr;
        if (hipMalloc(&table, static_cast<size_t>(records) * 128) != hipSuccess || hipMalloc(&out, static_cast<size_t>(grid) * 256 * 4) != hipSuccess) return 1;
        hipMemset(table, 0, static_cast<size_t>(records) * 128);
        std::printf("table of %u records of 128 bytes (%u MiB)\n", records, records >> 13);
        run<8, 8>("A  whole 128-byte records", table, records, out, grid);
        run<4, 8>("B  first 64 bytes of 128-byte records", table, records, out, grid);
        run<4, 4>("C  64-byte records at stride 64", table, records, out, grid);
        run<2, 8>("D  first 32 bytes of 128-byte records", table, records, out, grid);
        hipFree(table);
        hipFree(out);
    }
    return 0;
}
