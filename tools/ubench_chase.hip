// Development aid (not product code): what does one step of a divergent, dependent node walk cost on MI355X?
// Every lane chases its own random cycle through an array of `stride`-byte nodes, reading L x 16 bytes of each node
// (global_load_dwordx4 with 64 different cache lines per wave instruction -- the access pattern of a per-lane BVH
// walk), optionally from LDS.  Reports ns per wave step and lane-loads per clock per CU for several footprints,
// node sizes and occupancies: is the walk bound by latency (time/step falls with occupancy) or by the vector L1's
// tag rate (lane-loads/clk/CU saturates)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_chase.hip -o tools/bin/ubench_chase && tools/bin/ubench_chase
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template<int L, int CHAINS>
__global__ __launch_bounds__(512) void k_chase(const float4 *__restrict__ nodes, uint32_t strideQ, uint32_t numNodes, int steps, float *out)
{
    uint32_t tid = blockIdx.x*blockDim.x + threadIdx.x;
    uint32_t idx[CHAINS];
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
        idx[c] = (tid*2654435761u + c*40503u) % numNodes;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            const float4 *n = nodes + (size_t)idx[c]*strideQ;
            float4 v[L];
#pragma unroll
            for (int l = 0; l < L; ++l) v[l] = n[l];
#pragma unroll
            for (int l = 1; l < L; ++l) acc += v[l].x + v[l].w;
            idx[c] = __float_as_uint(v[0].x);
            acc += v[0].y;
        }
    }
    if (acc == 12345.678f) out[tid] = acc;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) if (idx[c] == 0xFFFFFFFFu) out[tid] = 1.0f;
}

// the same walk over a table staged in LDS (ds_read_b128, divergent addresses)
template<int L>
__global__ __launch_bounds__(512) void k_chase_lds(const float4 *__restrict__ nodes, uint32_t strideQ, uint32_t numNodes, int steps, float *out)
{
    extern __shared__ float4 lds[];
    for (uint32_t i = threadIdx.x; i < numNodes*strideQ; i += blockDim.x) lds[i] = nodes[i];
    __syncthreads();
    uint32_t tid = blockIdx.x*blockDim.x + threadIdx.x;
    uint32_t idx = (tid*2654435761u) % numNodes;
    float acc = 0.0f;
    for (int s = 0; s < steps; ++s) {
        const float4 *n = lds + idx*strideQ;
        float4 v[L];
#pragma unroll
        for (int l = 0; l < L; ++l) v[l] = n[l];
#pragma unroll
        for (int l = 1; l < L; ++l) acc += v[l].x + v[l].w;
        idx = __float_as_uint(v[0].x);
        acc += v[0].y;
    }
    if (acc == 12345.678f || idx == 0xFFFFFFFFu) out[tid] = acc;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate*1e-6;
    std::printf("device %s, %d CUs, %.2f GHz\n", prop.name, cus, ghz);
    float *out = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void **>(&out), size_t(cus)*32*512*sizeof(float)));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    std::mt19937 rng(1);
    const int steps = 512;

    struct Cfg { int strideB; double mb; };
    const Cfg cfgs[] = {{64, 2.6}, {64, 33.0}, {64, 150.0}, {128, 5.2}, {128, 66.0}, {80, 3.0}, {80, 40.0}};
    for (const Cfg &cf : cfgs) {
        const uint32_t strideQ = uint32_t(cf.strideB/16);
        const uint32_t n = uint32_t(cf.mb*1e6/cf.strideB);
        std::vector<uint32_t> perm(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<float> host(size_t(n)*strideQ*4, 1.0f);
        for (uint32_t i = 0; i < n; ++i) {          // one random cycle: perm[i] -> perm[i + 1]
            uint32_t next = perm[(i + 1) % n];
            std::memcpy(&host[size_t(perm[i])*strideQ*4], &next, 4);
        }
        float4 *dev = nullptr;
        CHECK(hipMalloc(reinterpret_cast<void **>(&dev), host.size()*sizeof(float)));
        CHECK(hipMemcpy(dev, host.data(), host.size()*sizeof(float), hipMemcpyHostToDevice));
        for (int wavesPerCu : {4, 8, 16, 20, 24, 32}) {
            for (int variant = 0; variant < 4; ++variant) {
                // variant 0: all 16-byte words of the node; 1: first 4 words only (when the node is larger); 2: two chains per lane; 3: one word
                const int Lfull = cf.strideB/16;
                if (variant == 1 && Lfull <= 4) continue;
                const int threads = 256, blocksPerCu = wavesPerCu*64/threads;
                if (blocksPerCu < 1) continue;
                dim3 grid(cus*blocksPerCu), block(threads);
                auto launch = [&](int st) {
#define RUN(LL, CC) hipLaunchKernelGGL((k_chase<LL, CC>), grid, block, 0, 0, dev, strideQ, n, st, out)
                    if (variant == 3) RUN(1, 1);
                    else if (variant == 2) { if (Lfull == 4) RUN(4, 2); else if (Lfull == 5) RUN(5, 2); else RUN(8, 2); }
                    else if (variant == 1) RUN(4, 1);
                    else { if (Lfull == 4) RUN(4, 1); else if (Lfull == 5) RUN(5, 1); else RUN(8, 1); }
#undef RUN
                };
                launch(16);
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(a));
                launch(steps);
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                float ms = 0.0f;
                CHECK(hipEventElapsedTime(&ms, a, b));
                const int chains = variant == 2 ? 2 : 1;
                const int L = variant == 3 ? 1 : variant == 1 ? 4 : Lfull;
                const double laneSteps = double(cus)*blocksPerCu*threads*steps*chains;
                const double laneLoads = laneSteps*L;
                std::printf("stride %3d B  footprint %6.1f MB  waves/CU %2d  L=%d chains=%d : %7.1f ns per wave step, %6.1f G lane-steps/s, %5.2f lane-loads/clk/CU, %6.1f GB/s\n",
                            cf.strideB, cf.mb, wavesPerCu, L, chains, ms*1e6/(steps*chains), laneSteps/ms*1e-6, laneLoads/(ms*1e-3)/(cus*ghz*1e9), laneLoads*16/ms*1e-6);
            }
        }
        CHECK(hipFree(dev));
    }
    // LDS-resident table: 48 KB of 80-byte / 64-byte nodes
    for (int strideB : {64, 80}) {
        const uint32_t strideQ = uint32_t(strideB/16), n = 48*1024/strideB;
        std::vector<uint32_t> perm(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<float> host(size_t(n)*strideQ*4, 1.0f);
        for (uint32_t i = 0; i < n; ++i) { uint32_t next = perm[(i + 1) % n]; std::memcpy(&host[size_t(perm[i])*strideQ*4], &next, 4); }
        float4 *dev = nullptr;
        CHECK(hipMalloc(reinterpret_cast<void **>(&dev), host.size()*sizeof(float)));
        CHECK(hipMemcpy(dev, host.data(), host.size()*sizeof(float), hipMemcpyHostToDevice));
        for (int threads : {256, 512}) {
            for (int blocksPerCu : {1, 2, 3}) {
                dim3 grid(cus*blocksPerCu), block(threads);
                const size_t ldsBytes = size_t(n)*strideB;
                auto launch = [&](int st) {
                    if (strideB == 64) hipLaunchKernelGGL((k_chase_lds<4>), grid, block, ldsBytes, 0, dev, strideQ, n, st, out);
                    else               hipLaunchKernelGGL((k_chase_lds<5>), grid, block, ldsBytes, 0, dev, strideQ, n, st, out);
                };
                launch(16);
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(a));
                launch(steps*4);
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                float ms = 0.0f;
                CHECK(hipEventElapsedTime(&ms, a, b));
                const double laneSteps = double(cus)*blocksPerCu*threads*steps*4;
                std::printf("LDS stride %3d B  waves/CU %2d : %7.1f ns per wave step, %6.1f G lane-steps/s, %5.2f lane-loads/clk/CU\n",
                            strideB, blocksPerCu*threads/64, ms*1e6/(steps*4), laneSteps/ms*1e-6, laneSteps*strideQ/(ms*1e-3)/(cus*ghz*1e9));
            }
        }
        CHECK(hipFree(dev));
    }
    return 0;
}
