// Development aid (not product code): what a lane's 4 / 8 / 12 / 16-byte global load costs on gfx950 when the data is cache-resident and every
// lane reads its own 16-byte-aligned element of a small table at a random index -- the access pattern of the walks' node rows and records.
// Round 5 found the compiler narrowing float4 reads whose last word is unused to global_load_dwordx3 and the kernels slower for it
// (profiles/r5_ab_x4_loads.txt); this measures the instructions in isolation.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_loads.hip -o tools/bin/ubench_loads && tools/bin/ubench_loads
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// every lane chases `iters` dependent-address loads through a table of `n` 16-byte elements (n a power of two): the next index comes out of
// the word loaded, so the loads of one lane are serial and the throughput is set by how many lanes' loads the memory pipeline takes per clock
#define DEFINE_KERNEL(NAME, ASM, REGS)                                                                      \
    __global__ __launch_bounds__(256) void k_##NAME(const char *table, uint32_t mask, int iters, uint32_t *out) \
    {                                                                                                           \
        uint32_t idx = (blockIdx.x*256u + threadIdx.x)*2654435761u;                                             \
        uint32_t acc = 0;                                                                                       \
        for (int i = 0; i < iters; ++i) {                                                                       \
            const uint32_t off = (idx & mask) << 4;                                                             \
            uint32_t r0, r1 = 0, r2 = 0, r3 = 0;                                                                \
            ASM                                                                                                 \
            acc ^= r1 ^ r2 ^ r3;                                                                                \
            idx = r0;                                                                                           \
        }                                                                                                       \
        if ((acc ^ idx) == 0x12345u) out[0] = acc;                                                              \
    }
DEFINE_KERNEL(x1, asm volatile("global_load_dword %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(r0) : "v"(off), "s"(table) : "memory");, 1)
DEFINE_KERNEL(x2, { uint64_t v; asm volatile("global_load_dwordx2 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(table) : "memory"); r0 = (uint32_t)v; r1 = (uint32_t)(v >> 32); }, 2)
typedef uint32_t U3 __attribute__((ext_vector_type(3)));
typedef uint32_t U4 __attribute__((ext_vector_type(4)));
DEFINE_KERNEL(x3, { U3 v; asm volatile("global_load_dwordx3 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(table) : "memory"); r0 = v.x; r1 = v.y; r2 = v.z; }, 3)
DEFINE_KERNEL(x4, { U4 v; asm volatile("global_load_dwordx4 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(table) : "memory"); r0 = v.x; r1 = v.y; r2 = v.z; r3 = v.w; }, 4)

int main()
{
    const int iters = 2048;
    uint32_t *out = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void **>(&out), 64));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount*8;
    std::printf("%d CUs, %d blocks of 256 threads, %d dependent loads per lane; every lane reads a 16-byte-aligned element at a pseudo-random index\n", prop.multiProcessorCount, grid, iters);
    std::printf("%-14s %-12s %10s %16s\n", "table", "load", "ms", "G lane-loads/s");
    for (uint32_t logn : {12u, 16u, 21u}) {             // 64 KB (L1-sized), 1 MB (L2), 32 MB (Infinity Cache)
        const size_t n = size_t(1) << logn;
        std::vector<uint32_t> host(n*4);
        uint32_t x = 0x9E3779B9u;
        for (size_t i = 0; i < n*4; ++i) { x = x*1664525u + 1013904223u; host[i] = x >> 3; }
        char *table = nullptr;
        CHECK(hipMalloc(reinterpret_cast<void **>(&table), n*16));
        CHECK(hipMemcpy(table, host.data(), n*16, hipMemcpyHostToDevice));
#define RUN(NAME) do { \
            hipLaunchKernelGGL(k_##NAME, dim3(grid), dim3(256), 0, 0, table, uint32_t(n - 1), 64, out); \
            CHECK(hipDeviceSynchronize()); \
            CHECK(hipEventRecord(a)); \
            hipLaunchKernelGGL(k_##NAME, dim3(grid), dim3(256), 0, 0, table, uint32_t(n - 1), iters, out); \
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); \
            float ms = 0.0f; CHECK(hipEventElapsedTime(&ms, a, b)); \
            std::printf("%6zu KB      %-12s %10.3f %16.1f\n", n*16/1024, "dword" #NAME, ms, double(grid)*256.0*iters/(ms*1e6)); } while (0)
        RUN(x1); RUN(x2); RUN(x3); RUN(x4);
        CHECK(hipFree(table));
    }
    return 0;
}
