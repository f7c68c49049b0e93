// Development aid (not product code): does a NON-TEMPORAL 16-byte store (global_store_dwordx4 ... nt, what PT_NT_STATE puts on the path pool's
// stores) leave a stale copy of its line in the CU's vector L1, so that a later PLAIN load of the same launch -- by the same lane, or by another
// wave of the workgroup after a barrier -- reads the old value?  That would explain round 5's k_trace_shadow_wide failure (a slot's radiance
// read-modify-written with the hint inside the walk's loop; profiles/r5_nt_hazard.txt) as a memory-system rule instead of a miscompile, and it
// decides where the hint may stay: only on stores no load of the same launch reads back.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_nt_coherence.hip -o tools/bin/ubench_nt_coherence && tools/bin/ubench_nt_coherence
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t U4v __attribute__((ext_vector_type(4)));

// a plain load the compiler can neither forward a store into nor move (what the product's ordinary slot loads are: no sc bits, no nt)
__device__ __forceinline__ uint32_t plainLoad(const uint32_t *p)
{
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template<bool NT>
__device__ __forceinline__ void store16(uint32_t *p, uint32_t v)
{
    const U4v t = {v, v, v, v};
    if (NT) __builtin_nontemporal_store(t, reinterpret_cast<U4v *>(p));
    else *reinterpret_cast<U4v *>(p) = t;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// A: the same lane reads its slot (line now in L1), stores a new value, reads again
template<bool NT>
__global__ __launch_bounds__(256) void k_same_lane(uint32_t *buf, uint32_t *bad, int iters)
{
    uint32_t *p = buf + (size_t)(blockIdx.x*blockDim.x + threadIdx.x)*4u;
    uint32_t wrong = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t v0 = plainLoad(p);
        store16<NT>(p, v0 + 1u);
        const uint32_t v1 = plainLoad(p);
        wrong += v1 != v0 + 1u ? 1u : 0u;
    }
    if (wrong) atomicAdd(bad, wrong);
}

// B: another wave of the workgroup has the line in L1; the owner stores; after a barrier the other wave reads it with a plain load
// (the fused k_finish_trace_closest_wide: finishBody's regenerated slots, __syncthreads, the walk's refill loads them)
template<bool NT, bool PRELOAD>
__global__ __launch_bounds__(256) void k_other_wave(uint32_t *buf, uint32_t *bad, int iters)
{
    uint32_t *own = buf + (size_t)(blockIdx.x*blockDim.x + threadIdx.x)*4u;
    uint32_t *partner = buf + (size_t)(blockIdx.x*blockDim.x + (threadIdx.x ^ 64u))*4u;
    uint32_t wrong = 0;
    for (int it = 0; it < iters; ++it) {
        if (PRELOAD) (void)plainLoad(partner);       // the partner's line is in this CU's L1 before the store
        __syncthreads();
        store16<NT>(own, (uint32_t)it*2654435761u + threadIdx.x);
        __syncthreads();
        const uint32_t v1 = plainLoad(partner);
        wrong += v1 != (uint32_t)it*2654435761u + (threadIdx.x ^ 64u) ? 1u : 0u;
        __syncthreads();
    }
    if (wrong) atomicAdd(bad, wrong);
}

// C: read-modify-write of a NEIGHBOURING slot's line: lane L loads slot L (plain), lane L+1's nt store to slot L+1 shares the 128-byte line;
// then every lane re-reads its own slot after its own store (A) -- covered by A since 8 slots share a line.

template<typename K>
static int run(const char *name, K kernel, uint32_t *buf, uint32_t *bad, size_t n)
{
    CHECK(hipMemset(buf, 0, n*16));
    CHECK(hipMemset(bad, 0, 4));
    const int iters = 200;
    hipLaunchKernelGGL(kernel, dim3((unsigned)(n/256)), dim3(256), 0, 0, buf, bad, iters);
    CHECK(hipDeviceSynchronize());
    uint32_t h = 0;
    CHECK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
    std::printf("%-58s stale reads %10u of %llu\n", name, h, (unsigned long long)n*iters);
    return 0;
}

int main()
{
    const size_t n = 256u*2048u;            // 8 workgroups per CU, 16 B per lane: 8 MB (L2-resident) -- and a pool-sized run below
    uint32_t *buf = nullptr, *bad = nullptr;
    CHECK(hipMalloc(&buf, n*16*32));
    CHECK(hipMalloc(&bad, 4));
    for (int big = 0; big < 2; ++big) {
        const size_t m = big ? n*32 : n;
        std::printf("-- %zu slots (%zu MB)\n", m, m*16 >> 20);
        if (run("A same lane: load, plain store, load", k_same_lane<false>, buf, bad, m)) return 1;
        if (run("A same lane: load, NT store, load", k_same_lane<true>, buf, bad, m)) return 1;
        if (run("B other wave, line preloaded: plain store, barrier, load", k_other_wave<false, true>, buf, bad, m)) return 1;
        if (run("B other wave, line preloaded: NT store, barrier, load", k_other_wave<true, true>, buf, bad, m)) return 1;
        if (run("B other wave, line NOT preloaded: NT store, barrier, load", k_other_wave<true, false>, buf, bad, m)) return 1;
    }
    CHECK(hipFree(buf)); CHECK(hipFree(bad));
    return 0;
}
