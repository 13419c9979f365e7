// How fast a compute unit's vector L1 serves gathers: cycles per wave-instruction of
// `global_load_dwordx4` for the address patterns the traversal kernel produces, with
// every SIMD holding 8 waves (4 workgroups of 512 threads per CU) and the table resident
// in L2 (2 MiB) or in L1 (16 KiB).  Output: ns and cycles (at the reported clock) per
// wave-instruction and CU, and the 16-byte lane accesses served per cycle and CU.
//
//   hipcc -O2 --offload-arch=gfx950 tools/gather_rate.hip -o gather_rate && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

// PATTERN: 0 every lane a random 64-byte record, one quarter of it (16 B)
//          1 every lane a random 64-byte record, all four quarters (4 loads: a branch record)
//          2 groups of 4 lanes read the four quarters of one random record (64 contiguous bytes per group)
//          3 runs of 5 lanes read consecutive 48-byte records (three loads each: a leaf's triangles)
//          4 runs of 5 lanes read consecutive 16-byte entries of three separate arrays (the same triangles, structure of arrays)
//          5 runs of 5 lanes read consecutive 32-byte entries (2 loads) + consecutive 4-byte entries (1 dword load)
template <int PATTERN>
__global__ void __launch_bounds__(512) k_gather(const float4* table, uint32_t mask_records, int iters, float* out) {
    uint32_t s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
    const uint32_t lane = threadIdx.x & 63u;
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        uint32_t r = s >> 8;
        if (PATTERN == 2) r = __shfl(r, lane & ~3u);
        if (PATTERN >= 3) r = __shfl(r, lane - lane % 5u);
        if (PATTERN == 0) {
            const float4 v = table[(r & mask_records) * 4u + (r >> 30)];
            acc += v.x + v.w;
        } else if (PATTERN >= 6) {
            // pattern 1 with a cache policy on the loads: 6 nt, 7 sc1, 8 sc0 sc1, 9 sc0 sc1 nt, 10 sc0
            const float4* p = table + (r & mask_records) * 4u;
            float4 a, b, c, d;
#define RT_GATHER4(POLICY)                                                                                                 \
    asm volatile("global_load_dwordx4 %0, %4, off " POLICY "\n\tglobal_load_dwordx4 %1, %4, off offset:16 " POLICY               \
                 "\n\tglobal_load_dwordx4 %2, %4, off offset:32 " POLICY "\n\tglobal_load_dwordx4 %3, %4, off offset:48 " POLICY \
                 "\n\ts_waitcnt vmcnt(0)"                                                                                  \
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)                                                                  \
                 : "v"(p)                                                                                                  \
                 : "memory")
            if (PATTERN == 6) RT_GATHER4("nt");
            else if (PATTERN == 7) RT_GATHER4("sc1");
            else if (PATTERN == 8) RT_GATHER4("sc0 sc1");
            else if (PATTERN == 9) RT_GATHER4("sc0 sc1 nt");
            else if (PATTERN == 10) RT_GATHER4("sc0");
            else RT_GATHER4("");
            acc += a.x + b.y + c.z + d.w;
        } else if (PATTERN == 1) {
            const float4* p = table + (r & mask_records) * 4u;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else if (PATTERN == 2) {
            const float4 v = table[(r & mask_records) * 4u + (lane & 3u)];
            acc += v.x + v.w;
        } else if (PATTERN == 3) {
            const uint32_t tri = ((r & mask_records) * 4u / 3u & ~7u) + lane % 5u;  // 48-byte records
            const float4* p = table + tri * 3u;
            const float4 a = p[0], b = p[1], c = p[2];
            acc += a.x + b.y + c.z;
        } else if (PATTERN == 4) {
            const uint32_t n = (mask_records + 1u) * 4u / 3u;  // entries per array
            const uint32_t tri = ((r & mask_records) * 4u / 3u & ~7u) + lane % 5u;
            const float4 a = table[tri], b = table[n + tri], c = table[2u * n + tri];
            acc += a.x + b.y + c.z;
        } else {
            const uint32_t n = (mask_records + 1u) * 4u / 3u;
            const uint32_t tri = ((r & mask_records) * 4u / 3u & ~7u) + lane % 5u;
            const float4 a = table[2u * tri], b = table[2u * tri + 1u];
            const float c = reinterpret_cast<const float*>(table + 2u * n)[tri];
            acc += a.x + b.y + c;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int PATTERN>
void run(const char* name, const float4* table, uint32_t records, int loads_per_iter, int n_cus, double ghz, float* out) {
    const int iters = 4000;
    const int grid = n_cus * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_gather<PATTERN>, dim3(grid), dim3(512), 0, 0, table, records - 1, 50, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_gather<PATTERN>, dim3(grid), dim3(512), 0, 0, table, records - 1, iters, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_cu = 32.0 * iters * loads_per_iter;  // 32 waves per CU
    const double ns = ms * 1e6 / wave_instr_per_cu;
    std::printf("%-44s %8.3f ms  %7.2f ns = %6.1f cycles per wave-instruction and CU, %5.2f lane accesses per cycle and CU\n", name,
                ms, ns, ns * ghz, 64.0 / (ns * ghz));
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate / 1e6;
    const int n_cus = prop.multiProcessorCount;
    std::printf("device clock %.2f GHz, %d CUs\n", ghz, n_cus);
    float* out;
    hipMalloc(&out, 4);
    for (uint32_t records : {256u, 32768u, 4194304u}) {  // 16 KiB (L1), 2 MiB (L2), 256 MiB (Infinity Cache / HBM)
        float4* table;
        const size_t bytes = static_cast<size_t>(records) * 64 * 2 + 4096;
        hipMalloc(&table, bytes);
        hipMemset(table, 0, bytes);
        std::printf("table of %u 64-byte records (%.1f KiB)\n", records, records * 64 / 1024.0);
        run<0>("  random record, one 16-B quarter", table, records, 1, n_cus, ghz, out);
        run<1>("  random record, four quarters (4 loads)", table, records, 4, n_cus, ghz, out);
        run<2>("  4 lanes share a record's 64 bytes", table, records, 1, n_cus, ghz, out);
        run<3>("  runs of 5 lanes, 48-B records (3 loads)", table, records, 3, n_cus, ghz, out);
        run<4>("  runs of 5 lanes, 3 arrays of 16 B", table, records, 3, n_cus, ghz, out);
        run<5>("  runs of 5 lanes, 32 B (2 loads) + 4 B", table, records, 3, n_cus, ghz, out);
        run<11>("  four quarters, asm, default policy", table, records, 4, n_cus, ghz, out);
        run<6>("  four quarters, nt", table, records, 4, n_cus, ghz, out);
        run<10>("  four quarters, sc0", table, records, 4, n_cus, ghz, out);
        run<7>("  four quarters, sc1", table, records, 4, n_cus, ghz, out);
        run<8>("  four quarters, sc0 sc1", table, records, 4, n_cus, ghz, out);
        run<9>("  four quarters, sc0 sc1 nt", table, records, 4, n_cus, ghz, out);
        hipFree(table);
    }
    return 0;
}
