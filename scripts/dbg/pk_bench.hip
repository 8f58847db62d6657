// Microbenchmark (GPU box): is packed f32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) worth anything on gfx950 at
// the raster kernel's residency?  One wave64 per block, W waves per SIMD resident, every wave runs
// a register-only loop of independent multiply-adds: the same flops either as scalar v_fma_f32 or as v_pk_fma_f32, and a
// mul + add pair (what bit-identity with unfused scalar code would need) as v_pk_mul_f32 + v_pk_add_f32.
//   hipcc --offload-arch=gfx950 -O3 -o build/dbg2/pk_bench scripts/dbg/pk_bench.hip && build/dbg2/pk_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kAcc = 16;  // independent accumulators per lane (scalar) = 8 packed pairs

template <int MODE>  // 0: scalar fma, 1: packed fma, 2: scalar mul + add (no contraction), 3: packed mul + add
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
    float r[kAcc];
#pragma unroll
    for (int i = 0; i < kAcc; ++i) r[i] = (float)threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < kAcc; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < kAcc; i += 2) {
                v2f x = {r[i], r[i + 1]}, aa = {a, a}, bb = {b, b};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(aa), "v"(bb));
                r[i] = x.x; r[i + 1] = x.y;
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < kAcc; ++i) {
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
            }
        } else {
#pragma unroll
            for (int i = 0; i < kAcc; i += 2) {
                v2f x = {r[i], r[i + 1]}, aa = {a, a}, bb = {b, b};
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(aa));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(bb));
                r[i] = x.x; r[i + 1] = x.y;
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kAcc; ++i) s += r[i];
    if (s == 123.456f) out[0] = s;
}

template <int MODE>
static double run(int waves_per_simd, int iters) {
    int dev = 0;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, dev);
    const int blocks = p.multiProcessorCount * 4 * waves_per_simd;
    float* out;
    hipMalloc(&out, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(out, iters, 0.999f, 0.001f);  // warm-up of the same length: the clocks have ramped when the timed launch starts
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(out, iters, 0.999f, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    // lane-operations (one multiply-add of one lane) per second, chip-wide
    return (double)blocks * 64.0 * kAcc * iters / (ms * 1e-3);
}

int main() {
    const int iters = 20000;
    const char* names[4] = {"v_fma_f32 x16", "v_pk_fma_f32 x8", "v_mul_f32 + v_add_f32 x16", "v_pk_mul_f32 + v_pk_add_f32 x8"};
    printf("{\"unit\": \"10^12 lane multiply-adds per second, whole chip\", \"rows\": [\n");
    for (int w = 1; w <= 4; ++w) {
        double v[4] = {run<0>(w, iters), run<1>(w, iters), run<2>(w, iters), run<3>(w, iters)};
        for (int m = 0; m < 4; ++m)
            printf("  {\"waves_per_simd\": %d, \"mode\": \"%s\", \"tera_madds\": %.2f, \"vs_scalar\": %.2f}%s\n", w, names[m], v[m] / 1e12,
                   v[m] / v[m < 2 ? 0 : 2], (w == 4 && m == 3) ? "" : ",");
    }
    printf("]}\n");
    return 0;
}
