// occ_common.hpp -- small device helpers shared by every kernel.
// Part of the single translation unit occ_kernels.hip (included inside namespace occ; not a stand-alone header).

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fmin3(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float fmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float clamp01(float t) { return fminf(fmaxf(t, 0.0f), 1.0f); }

// Read-only data produced by an EARLIER launch (face records, bboxes, rects) is read through the
// constant address space: with a wave-uniform address hipcc then emits s_load_dwordx8/x16 into
// SGPRs (scalar cache) instead of 64 redundant vector loads.
typedef const __attribute__((address_space(4))) float* cfptr;
typedef const __attribute__((address_space(4))) int* ciptr;
__device__ __forceinline__ cfptr as_const(const float* p) { return (cfptr)(uintptr_t)p; }
__device__ __forceinline__ ciptr as_const(const int* p) { return (ciptr)(uintptr_t)p; }

// Layout of OccWorkspace.order (u32 words, see include/occlusionenv_amd.h): the work-item order of occ_raster2_kernel
constexpr int kOrdCounts = 16;    // [16 + 32 q + c]: tiles of cost class c in XCD queue q
constexpr int kOrdClasses = 32;
constexpr int kOrdBlk = 512;      // [512 + 32 eo + c]: start of object eo's class-c tiles inside the class
__host__ __device__ __forceinline__ size_t ord_tiles_word(int n_env) { return (size_t)kOrdBlk + (size_t)n_env * 3 * kOrdClasses; }
// the item list is read as uint2: its first word is even (one pad word when the tile table ends on an odd word)
__host__ __device__ __forceinline__ size_t ord_items_word(int n_env, int img) {
    return (ord_tiles_word(n_env) + (size_t)n_env * 3 * (img / 8) * (img / 8) + 1) & ~(size_t)1;
}
// cost class of a tile that c faces touch: two classes per octave, 0 = empty
__device__ __forceinline__ int ord_class(uint32_t c) {
    if (c == 0u) return 0;
    const int fl = 31 - __builtin_clz(c);
    const int half = fl > 0 ? (int)((c >> (fl - 1)) & 1u) : 0;
    return min(kOrdClasses - 1, 1 + 2 * fl + half);
}

// faces that touch a tile of class cls: strictly fewer than this
__device__ __forceinline__ uint32_t ord_class_bound(int cls) {
    if (cls <= 0) return 1u;
    if (cls >= kOrdClasses - 1) return 0xFFFFFFFFu;
    const int fl = (cls - 1) >> 1, half = (cls - 1) & 1;
    return fl == 0 ? 2u : (half ? 2u << fl : 3u << (fl - 1));
}

// Work-queue heads (OccWorkspace.queue): one per XCD group, kQueueStride words apart - a 128-byte L2 line each (at 64 bytes
// two heads shared a line: 1.894 -> 1.873 ms for the raster kernel of the bench; 256 B or 4 KB apart: the same).
#ifndef OCC_QUEUE_STRIDE
#define OCC_QUEUE_STRIDE 32
#endif
constexpr int kQueueStride = OCC_QUEUE_STRIDE;

// a * b for operands below 2^24 as ONE full-rate instruction.  Opaque on purpose: left to itself hipcc folds the
// surrounding subtraction into a multiply by a negative constant, which needs the quarter-rate v_mul_lo_u32.
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Cross-lane hand-over through LDS inside ONE wave (the raster kernel's workgroup is a single wave64).  The LDS unit
// executes a wave's operations in issue order, so a lane reading what another lane of the same wave stored earlier in
// program order needs no hardware wait at all - only the compiler has to keep that program order.  __syncthreads()
// would also drain every outstanding global load and store (s_waitcnt vmcnt(0)): the candidate-log stores of the round
// just evaluated and the record loads that are meant to stay in flight.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() also waits for every outstanding global store of the
// wave (s_waitcnt vmcnt(0)): in a loop that streams records out and hands small things over through LDS, each barrier
// then costs a store round trip to memory.  Use where nothing that crosses the barrier went through global memory.
__device__ __forceinline__ void block_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// Sum over the 64 lanes with DPP row operations (VALU, a few cycles each) instead of six ds_bpermute round trips through
// the LDS crossbar (~100 cycles each, and dependent on one another): quad swaps, half-row and row mirrors give every lane
// its 16-lane row's sum, row_bcast:15 / row_bcast:31 fold the four rows into lane 63, whose value every lane returns.
// (A fixed summation order, different from the xor butterfly's.)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    auto dpp = [](float x, auto ctrl, auto row_mask) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(row_mask)::value, 0xF, true));
    };
    using std::integral_constant;
    v += dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xF>{});   // quad_perm [1,0,3,2]
    v += dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xF>{});   // quad_perm [2,3,0,1]
    v += dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xF>{});  // row_half_mirror
    v += dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xF>{});  // row_mirror
    v += dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xA>{});  // row_bcast:15 into rows 1 and 3
    v += dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xC>{});  // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// Unsigned minimum / maximum over the 64 lanes the same way (lanes a DPP step does not reach keep the identity).
template <bool MAX>
__device__ __forceinline__ uint32_t wave_minmax_u32_dpp(uint32_t v) {
    auto step = [](uint32_t x, auto ctrl, auto row_mask) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(MAX ? 0 : -1 /* the identity: 0 / 0xFFFFFFFF */, (int)x, decltype(ctrl)::value, decltype(row_mask)::value, 0xF, false);
        return MAX ? max(x, o) : min(x, o);
    };
    using std::integral_constant;
    v = step(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xF>{});
    v = step(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xF>{});
    v = step(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xF>{});
    v = step(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xF>{});
    v = step(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xA>{});
    v = step(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xC>{});
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
    return v;
}
