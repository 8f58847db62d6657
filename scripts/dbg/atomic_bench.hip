// Microbenchmark (GPU box): cost of the work-queue operations of occ_raster2_kernel on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -o build/dbg2/atomic_bench scripts/dbg/atomic_bench.hip && build/dbg2/atomic_bench
// One wave64 per block like the raster kernel, 12 per CU; lane 0 does the memory operation and the wave waits for it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

enum { kRmwOne = 0, kRmwEight = 1, kPeekHot = 2, kPeekCold = 3, kRmwOwnXcd = 4, kPlainLoad = 5 };

__global__ __launch_bounds__(64) void k(uint32_t* ctr, uint32_t* flags, unsigned long long* cyc, int mode, int rounds, int active_mod) {
    const int lane = threadIdx.x;
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
    if ((int)(blockIdx.x % active_mod) != 0) return;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        uint32_t v = 0;
        if (lane == 0) {
            switch (mode) {
                case kRmwOne: v = atomicAdd(ctr, 1u); break;
                case kRmwEight: v = atomicAdd(ctr + ((blockIdx.x + r) & 7) * 16, 1u); break;
                case kRmwOwnXcd: v = atomicAdd(ctr + xcc * 16, 1u); break;
                case kPeekHot: v = __hip_atomic_load(ctr + ((blockIdx.x + r) & 7) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                               if ((blockIdx.x & 7) == 0) v += atomicAdd(ctr + ((blockIdx.x + r) & 7) * 16, 1u);
                               break;
                case kPeekCold: v = __hip_atomic_load(flags + ((blockIdx.x + r) & 7) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
                case kPlainLoad: v = *(volatile uint32_t*)(flags + ((blockIdx.x + r) & 7) * 16); break;
            }
        }
        acc += __builtin_amdgcn_readfirstlane(v);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        atomicAdd(&cyc[0], t1 - t0);
        atomicAdd(&cyc[1], 1ull);
        if (acc == 0xFFFFFFFFu) flags[200] = acc;
    }
}

int main() {
    uint32_t *ctr, *flags;
    unsigned long long* cyc;
    hipMalloc(&ctr, 4096); hipMalloc(&flags, 4096); hipMalloc(&cyc, 64);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int waves = p.multiProcessorCount * 12;
    const char* names[] = {"RMW, one address", "RMW, eight addresses (rotating)", "agent-scope load of a hot line (1/8 of the waves RMW it)",
                           "agent-scope load of a quiet line", "RMW, one address per XCD", "volatile load of a quiet line"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int active_mod : {1, 8, 3072}) {
        for (int mode = 0; mode < 6; ++mode) {
            const int rounds = 64;
            hipMemset(ctr, 0, 4096); hipMemset(flags, 0, 4096); hipMemset(cyc, 0, 64);
            k<<<waves, 64>>>(ctr, flags, cyc, mode, 4, active_mod);  // warm-up
            hipMemset(cyc, 0, 64);
            hipEventRecord(e0);
            k<<<waves, 64>>>(ctr, flags, cyc, mode, rounds, active_mod);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
            const double nw = (double)h[1];
            printf("waves %5.0f  %-58s launch %8.1f us  per op and wave %7.2f us  aggregate %7.1f ns/op\n", nw, names[mode],
                   ms * 1e3, (double)h[0] / nw / rounds / 100.0, ms * 1e6 / (nw * rounds));
        }
    }
    return 0;
}
