// Calibrates the memory-side byte counters (FETCH_SIZE / WRITE_SIZE) on the
// access patterns the render kernels use, with known byte counts:
//   k_stream16   every lane reads 16 B, consecutive lanes consecutive addresses
//   k_gather64   every lane reads one random 64-B record (four 16-B loads)
//   k_gather128  every lane reads one random 128-B record (eight 16-B loads)
//   k_store16    every lane writes 16 B, consecutive
//   k_scatter8   every lane writes 8 B at a random 8-B slot of a large table
// Tables are 2 GiB so that neither L2 nor the 256 MiB last-level cache can hold
// them.  Run under `rocprofv3 --pmc FETCH_SIZE` (and again with WRITE_SIZE) and
// compare with the byte counts this program prints.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)

__device__ inline uint64_t mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

__global__ void k_stream16(const float4* __restrict__ src, float* sink, size_t n) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = src[i];
    if (v.x + v.y + v.z + v.w == 12345.f) sink[0] = v.x;
}

template <int VEC>
__global__ void k_gather(const float4* __restrict__ src, float* sink, size_t n_lanes, size_t n_records) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_lanes) return;
    size_t r = mix(i) % n_records;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        float4 v = src[r * VEC + k];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.f) sink[0] = acc;
}

__global__ void k_store16(float4* dst, size_t n) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dst[i] = make_float4(1.f, 2.f, 3.f, float(i));
}

__global__ void k_scatter8(uint2* dst, size_t n_lanes, size_t n_slots) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_lanes) return;
    dst[mix(i) % n_slots] = make_uint2(uint32_t(i), 7u);
}

int main() {
    const size_t bytes = size_t(2) << 30;
    void* table = nullptr;
    float* sink = nullptr;
    CHECK(hipMalloc(&table, bytes));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(table, 0, bytes));
    CHECK(hipDeviceSynchronize());

    const size_t n16 = bytes / 16;
    const size_t lanes = size_t(1) << 24;
    const int block = 256;
    hipLaunchKernelGGL(k_stream16, dim3(unsigned((n16 + block - 1) / block)), dim3(block), 0, 0,
                       static_cast<const float4*>(table), sink, n16);
    hipLaunchKernelGGL(k_gather<4>, dim3(unsigned(lanes / block)), dim3(block), 0, 0,
                       static_cast<const float4*>(table), sink, lanes, bytes / 64);
    hipLaunchKernelGGL(k_gather<8>, dim3(unsigned(lanes / block)), dim3(block), 0, 0,
                       static_cast<const float4*>(table), sink, lanes, bytes / 128);
    hipLaunchKernelGGL(k_store16, dim3(unsigned((n16 + block - 1) / block)), dim3(block), 0, 0,
                       static_cast<float4*>(table), n16);
    hipLaunchKernelGGL(k_scatter8, dim3(unsigned(lanes / block)), dim3(block), 0, 0,
                       static_cast<uint2*>(table), lanes, bytes / 8);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());

    std::printf("k_stream16  reads  %zu bytes\n", bytes);
    std::printf("k_gather<4> reads  %zu bytes (%zu records of 64 B)\n", lanes * 64, lanes);
    std::printf("k_gather<8> reads  %zu bytes (%zu records of 128 B)\n", lanes * 128, lanes);
    std::printf("k_store16   writes %zu bytes\n", bytes);
    std::printf("k_scatter8  writes %zu bytes (%zu slots of 8 B; a 64-B line holds 8)\n", lanes * 8, lanes);
    CHECK(hipFree(table));
    CHECK(hipFree(sink));
    return 0;
}
