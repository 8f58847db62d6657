// occ_raster.hpp -- raster kernel: per-block soft/hard rasterisation, K-buffer, exact top-K selection.
// Part of the single translation unit occ_kernels.hip (included inside namespace occ; not a stand-alone header).

// ------------------------------------------------------------------------------------------
// tile rasteriser
// ------------------------------------------------------------------------------------------
struct Cand {
    bool cand;    // soft candidate (inside, or within blur)
    bool inside;  // pixel centre strictly inside (hard candidate)
    float z;      // soft depth (clipped barycentrics)
    float zh;     // hard depth (unclipped barycentrics)
    float ad;     // |squared distance|
    int amin;     // closest edge: 0 = (v0,v1), 1 = (v0,v2), 2 = (v1,v2)
    float q;      // 1 - sigmoid(-d/sigma)
    float ge, ga; // p * d(d)/d el, p * d(d)/d az
};

// Evaluate one projected face (wave-uniform record r -> SGPRs) at this lane's pixel centre.
// Restates [P3D] CheckPixelInsideFace and, for GRAD, the dists part of
// RasterizeMeshesBackward (A.5) pushed forward along the two vertex tangents.
// One staged record pulled out of LDS with 16-byte broadcast reads (every lane reads the same address).
// Slot map as in occ_constants.h:
//   a = x0 y0 z0 x1 | b = y1 z1 x2 y2 | c = z2 id flags inv_area | d = bbox | e = il01 il02 il12 - |
//   g, h, i = tangents of v0, v1, v2 (dx/del dy/del dx/daz dy/daz)      -- 8 parts = 128 bytes = one cache line
// The eight float4 parts travel as SSA values (by value, never through a struct in memory: a select between two
// loads of one stack object gets folded into a dynamically indexed load, which pins the object in scratch).
#define OCC_REC_PARAMS float4 ra, float4 rb, float4 rc, float4 rd, float4 re, float4 rg, float4 rh, float4 ri
#define OCC_REC_LOAD(src, PARTS)                                                                   \
    (src)[0], (src)[1], (src)[2], (src)[3], ((PARTS) > 4 ? (src)[4] : make_float4(0, 0, 0, 0)),      \
        ((PARTS) > 5 ? (src)[5] : make_float4(0, 0, 0, 0)), ((PARTS) > 5 ? (src)[6] : make_float4(0, 0, 0, 0)), \
        ((PARTS) > 5 ? (src)[7] : make_float4(0, 0, 0, 0))

// Evaluate one projected face at this lane's pixel centre.  EARLY: lanes outside the face's bbox leave at once (the
// block kernel evaluates a face at all 16 pixels of a block, most of them outside); the pair kernel only visits
// pixels of the face's pixel bbox, where a divergent early exit costs more than it saves: branch-free, masked at the end.
template <bool SOFT, bool GRAD, bool EARLY = true>
__device__ __forceinline__ void eval_face(OCC_REC_PARAMS, float xf, float yf, Cand& c) {
    c.cand = false;
    c.inside = false;
    c.z = c.zh = c.ad = 0.f;
    c.q = 1.f;
    c.ge = c.ga = 0.f;
    c.amin = 0;
    const bool inb = (rd.x <= xf) && (xf <= rd.y) && (rd.z <= yf) && (yf <= rd.w);
    if (EARLY && !inb) return;
    const float x0 = ra.x, y0 = ra.y, z0 = ra.z;
    const float x1 = ra.w, y1 = rb.x, z1 = rb.y;
    const float x2 = rb.z, y2 = rb.w, z2 = rc.x;
    const float dx0 = xf - x0, dy0 = yf - y0, dx1 = xf - x1, dy1 = yf - y1, dx2 = xf - x2, dy2 = yf - y2;
    const float ex01 = x1 - x0, ey01 = y1 - y0, ex02 = x2 - x0, ey02 = y2 - y0, ex12 = x2 - x1, ey12 = y2 - y1;
    const float inv_area = rc.w;
    // barycentrics: E(p;v1,v2), E(p;v2,v0), E(p;v0,v1) over area
    const float b0 = (dx1 * ey12 - dy1 * ex12) * inv_area;
    const float b1 = (dy2 * ex02 - dx2 * ey02) * inv_area;
    const float b2 = (dx0 * ey01 - dy0 * ex01) * inv_area;
    // perspective correction
    const float w0 = b0 * z1 * z2, w1 = z0 * b1 * z2, w2 = z0 * z1 * b2;
    const float rden = frcp(fmaxf(w0 + w1 + w2, kEpsilon));
    const float p0 = w0 * rden, p1 = w1 * rden, p2 = w2 * rden;
    const bool inside = (p0 > 0.0f) && (p1 > 0.0f) && (p2 > 0.0f);
    c.zh = p0 * z0 + p1 * z1 + p2 * z2;
    c.inside = inb && inside && !(c.zh < 0.0f);
    if (!SOFT) return;
    // clipped barycentrics -> soft depth
    float c0 = fmaxf(p0, 0.f), c1 = fmaxf(p1, 0.f), c2 = fmaxf(p2, 0.f);
    const float rs = frcp(fmaxf(c0 + c1 + c2, kBaryClipMin));
    c0 *= rs;
    c1 *= rs;
    c2 *= rs;
    const float pz = c0 * z0 + c1 * z1 + c2 * z2;
    // squared distance to the three edges (v0,v1), (v0,v2), (v1,v2)
    const float il01 = re.x, il02 = re.y, il12 = re.z;
    const float dot01 = ex01 * dx0 + ey01 * dy0;
    const float dot02 = ex02 * dx0 + ey02 * dy0;
    const float dot12 = ex12 * dx1 + ey12 * dy1;
    const float t01 = il01 < 0.f ? 1.0f : clamp01(dot01 * il01);
    const float t02 = il02 < 0.f ? 1.0f : clamp01(dot02 * il02);
    const float t12 = il12 < 0.f ? 1.0f : clamp01(dot12 * il12);
    const float qx01 = t01 * ex01 - dx0, qy01 = t01 * ey01 - dy0;
    const float qx02 = t02 * ex02 - dx0, qy02 = t02 * ey02 - dy0;
    const float qx12 = t12 * ex12 - dx1, qy12 = t12 * ey12 - dy1;
    const float d01 = qx01 * qx01 + qy01 * qy01;
    const float d02 = qx02 * qx02 + qy02 * qy02;
    const float d12 = qx12 * qx12 + qy12 * qy12;
    const float dist = fmin3(d01, d02, d12);
    // closest edge with [P3D] tie order e01, e02, e12
    const bool s01 = (d01 <= d02) && (d01 <= d12);
    const bool s02 = !s01 && (d02 <= d01) && (d02 <= d12);
    const bool s12 = !s01 && !s02 && (d12 <= d01) && (d12 <= d02);
    c.amin = s01 ? 0 : (s02 ? 1 : 2);
    const bool cand = inb && !(pz < 0.0f) && (inside || dist < kBlurRadius);
    c.cand = cand;
    c.z = pz;
    c.ad = dist;
    const float sd = inside ? -dist : dist;
    // [P3D] sigmoid_alpha_blend: p = sigmoid(-d/sigma) = 1/(1+exp(d/sigma))  
    const float e = __expf(sd * kInvSigma);
    const float p = frcp(1.0f + e);
    c.q = 1.0f - p;
    if (GRAD) {
        // gradient through the closest edge; t recomputed with (l2 + eps) like [P3D]'s backward
        const float bax = s01 ? ex01 : (s02 ? ex02 : ex12);
        const float bay = s01 ? ey01 : (s02 ? ey02 : ey12);
        const float dotv = s01 ? dot01 : (s02 ? dot02 : dot12);
        // 1 / (|b-a|^2 + eps) from the stored 1 / |b-a|^2 (a degenerate edge is flagged -1: its |b-a|^2 <= eps)
        const float il = s01 ? il01 : (s02 ? il02 : il12);
        const float ile = il < 0.f ? 0.5f / kEpsilon : il * frcp(1.0f + kEpsilon * il);
        const float pax = s12 ? dx1 : dx0, pay = s12 ? dy1 : dy0;
        const float tb = clamp01(dotv * ile);
        const float gx = 2.0f * (tb * bax - pax), gy = 2.0f * (tb * bay - pay);  // 2 (proj - p)
        // tangent of the projected point: (1-t) a' + t b'
        const float a_xe = s12 ? rh.x : rg.x, a_ye = s12 ? rh.y : rg.y;
        const float a_xa = s12 ? rh.z : rg.z, a_ya = s12 ? rh.w : rg.w;
        const float b_xe = s01 ? rh.x : ri.x, b_ye = s01 ? rh.y : ri.y;
        const float b_xa = s01 ? rh.z : ri.z, b_ya = s01 ? rh.w : ri.w;
        const float mxe = a_xe + tb * (b_xe - a_xe), mye = a_ye + tb * (b_ye - a_ye);
        const float mxa = a_xa + tb * (b_xa - a_xa), mya = a_ya + tb * (b_ya - a_ya);
        const float any = (s01 || s02 || s12) ? 1.0f : 0.0f;
        const float sp = (inside ? -p : p) * any;
        c.ge = sp * (gx * mxe + gy * mye);
        c.ga = sp * (gx * mxa + gy * mya);
    }
}

__device__ __forceinline__ uint32_t zkey(float z) {
    const uint32_t b = __float_as_uint(z);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// combine a per-lane value over the four lanes that share a pixel (lanes l, l+16, l+32, l+48)
__device__ __forceinline__ int px_sum_i(int v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float px_sum_f(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float px_prod_f(float v) { v *= __shfl_xor(v, 16, 64); v *= __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ uint32_t px_min_u(uint32_t v) { v = min(v, (uint32_t)__shfl_xor((int)v, 16, 64)); v = min(v, (uint32_t)__shfl_xor((int)v, 32, 64)); return v; }
__device__ __forceinline__ uint32_t px_max_u(uint32_t v) { v = max(v, (uint32_t)__shfl_xor((int)v, 16, 64)); v = max(v, (uint32_t)__shfl_xor((int)v, 32, 64)); return v; }
__device__ __forceinline__ bool px_any(bool v) { return px_sum_i(v ? 1 : 0) != 0; }

#ifdef OCC_DBG_BOUNDS  // diagnostic build only: index checks at every memory access of the raster kernel
__device__ int g_dbg_fault[8];
#define OCC_BOUND(cond, code, v0, v1)                                          \
    ((cond) ? true                                                             \
            : ((atomicCAS(&g_dbg_fault[0], 0, (code)) == 0                     \
                    ? (g_dbg_fault[1] = (int)blockIdx.x, g_dbg_fault[2] = (int)threadIdx.x, g_dbg_fault[3] = (int)(v0), \
                       g_dbg_fault[4] = (int)(v1), 0)                          \
                    : 0),                                                      \
               false))
#define OCC_WATCHDOG(code, v0, v1)                                              \
    do {                                                                        \
        if (++wd_iters > 4000000) {                                             \
            (void)OCC_BOUND(false, (code), (v0), (v1));                         \
            return;                                                             \
        }                                                                       \
    } while (0)
#else
#define OCC_BOUND(cond, code, v0, v1) true
#define OCC_WATCHDOG(code, v0, v1) do { } while (0)
#endif

// Exact top-K-by-z for one PIXEL whose candidates sit in the lists of its four lanes (lane g = lane >> 4 holds
// the candidates of faces g, g+4, ... in face order; entry e = (key(z), 1-p, g_el, g_az) at list[e*64 + lane],
// the key being the order-preserving integer image of z).  Keeps the K smallest z like [P3D]'s (pz, face) ordering
// (SURVEY A.4); exact-z ties at the boundary are granted to lane 0 first, then 1, 2, 3, each in face order.
//
// The lists live in HBM/L2 (they do not fit LDS at 11 waves/CU), so the selection touches them as little as
// possible: a most-significant-digit radix select, kHistBits bits per level.  Every lane histograms its OWN key
// rows into a private u16 histogram in LDS (lane stride kHistStride dwords = conflict-free when lanes agree); the
// four histograms of a pixel are summed with two cross-lane adds while they are scanned, so the four lanes
// take identical decisions.  The window [L, L + (1 << kHistBits) << sh) starts at the pixel's own [kmin, kmax] key
// range, so a few levels resolve the bits below the first differing one.  Key sweeps are pipelined 16 rows
// deep (every level is one latency-bound pass over the lists).  The last sweep reads the payload rows once and takes every key below the boundary bucket plus this
// lane's share of the keys inside it.  Pixels with active == false idle.
// COMPACT: also moves the kept entries to the front of each list (stable) and returns the new own count.
// radix-select digit: 4 bits -> 16 u16 buckets = 8 dwords per lane (+1 pad: conflict-free when lanes agree)
constexpr int kHistBits = 4;       // in-loop compaction (the staging buffer is live): own small LDS area
constexpr int kHistBitsFinal = 5;  // final selection: histograms in the idle staging buffer
constexpr int kHistDwords = (1 << kHistBits) / 2;
constexpr int kHistStride = kHistDwords + 1;

template <bool COMPACT, int kBits>
__device__ __forceinline__ void topk_select4(float4* __restrict__ list, uint32_t* __restrict__ hist, int lane,
                                             int cnt, int K, bool active,
                                             uint32_t kmin_own, uint32_t kmax_own, float& pr, float& se, float& sa,
                                             uint32_t& Tmax, int& kept) {
#ifdef OCC_DBG_BOUNDS
    (void)OCC_BOUND(!active || (cnt >= 0 && cnt <= OCC_LIST_CAP), 24, cnt, K);
    const int maxc = min(wave_max_i(active ? cnt : 0), OCC_LIST_CAP);
#else
    const int maxc = wave_max_i(active ? cnt : 0);
#endif
    constexpr int kDwords = (1 << kBits) / 2, kStride = kDwords + 1;  // u16 buckets, one pad dword per lane
    uint32_t* __restrict__ h = hist + lane * kStride;
    // the key is component x of the 16-byte row entry: row e of this lane sits 256 dwords further on
    const uint32_t* __restrict__ keyp = reinterpret_cast<const uint32_t*>(list) + lane * 4;
    const uint32_t kmin = px_min_u(kmin_own), kmax = px_max_u(kmax_own);
    uint32_t L = kmin;
    const uint32_t range = kmax >= kmin ? kmax - kmin : 0u;
    int sh = range ? max(0, (32 - __builtin_clz(range)) - kBits) : 0;
    int need = K;
    int m_own = 0, mstar_px = 0;
    bool done = !active;
    while (__ballot(!done)) {
#pragma unroll
        for (int i = 0; i < kDwords; ++i) h[i] = 0u;
        for (int e0 = 0; e0 < maxc; e0 += 16) {
            uint32_t kk[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = e0 + i;
                kk[i] = (!done && e < cnt && OCC_BOUND(e < OCC_LIST_CAP, 21, e, cnt)) ? keyp[(size_t)e * 256] : 0u;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = e0 + i;
                if (!done && e < cnt) {
                    const uint32_t k = kk[i];
                    const uint32_t d = (k - L) >> sh;
                    if (k >= L && d < (1u << kBits)) h[d >> 1] += 1u << (16 * (d & 1u));
                }
            }
        }
        int cum = 0, bstar = (1 << kBits) - 1, mstar = 0, cumb = 0, mown = 0;
        bool found = false;
#pragma unroll
        for (int i = 0; i < kDwords; ++i) {
            const uint32_t wo = h[i];
            uint32_t w = wo;  // joint histogram of the pixel: the four private ones added up
            w += (uint32_t)__shfl_xor((int)w, 16, 64);
            w += (uint32_t)__shfl_xor((int)w, 32, 64);
            const int c0 = (int)(w & 0xFFFFu), c1 = (int)(w >> 16);
            if (!found && cum + c0 >= need) { found = true; bstar = 2 * i; mstar = c0; cumb = cum; mown = (int)(wo & 0xFFFFu); }
            cum += c0;
            if (!found && cum + c1 >= need) { found = true; bstar = 2 * i + 1; mstar = c1; cumb = cum; mown = (int)(wo >> 16); }
            cum += c1;
        }
        if (!done) {
            need -= cumb;
            L += (uint32_t)bstar << sh;
            m_own = mown;
            mstar_px = mstar;
            if (mstar == need || sh == 0 || !found) {
                done = true;
            } else {
                sh = max(0, sh - kBits);
            }
        }
    }
    // this lane's share of the boundary bucket [L, L + 2^sh): all of it when the whole bucket is taken, else
    // (exact ties) lanes are served in order 0, 1, 2, 3
    const int g = lane >> 4, base = lane & 15;
    int before = 0;
#pragma unroll
    for (int gg = 0; gg < 3; ++gg) {
        const int mo = __shfl(m_own, base + 16 * gg, 64);
        if (gg < g) before += mo;
    }
    int take = (mstar_px == need) ? m_own : min(m_own, max(0, need - before));
    int w = 0;
    pr = 1.0f;
    se = 0.f;
    sa = 0.f;
    uint32_t tmax = 0;
    // (payload sweep: 4 rows in flight - 8 would set the kernel's register peak and cost a wave per SIMD)
    for (int e0 = 0; e0 < maxc; e0 += 4) {
        uint32_t kk[4];
        bool inc[4];
        float4 vv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + i;
            kk[i] = (active && e < cnt && OCC_BOUND(e < OCC_LIST_CAP, 22, e, cnt)) ? keyp[(size_t)e * 256] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + i;
            const uint32_t k = kk[i];
            bool in_ = false;
            if (active && e < cnt) {
                in_ = k < L;
                if (k >= L && ((k - L) >> sh) == 0u && take > 0) {
                    in_ = true;
                    take -= 1;
                }
            }
            inc[i] = in_;
            vv[i] = make_float4(0.f, 1.f, 0.f, 0.f);
            if (in_) vv[i] = list[(size_t)e * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (inc[i]) {
                pr *= vv[i].y;
                se += vv[i].z;
                sa += vv[i].w;
                tmax = max(tmax, kk[i]);
                if (COMPACT && OCC_BOUND(w < OCC_LIST_CAP, 23, w, cnt)) list[(size_t)w * 64 + lane] = vv[i];
                w += 1;
            }
        }
    }
    Tmax = px_max_u(tmax);
    kept = w;
}

#ifdef OCC_DBG_STATS  // diagnostic build only: loop trip counts of the raster kernel
__device__ unsigned long long g_dbg_stats[8];
#define OCC_STAT(i, v) do { const unsigned long long v_ = (unsigned long long)(v); /* all lanes: v may hold a ballot */ if (lane == 0) atomicAdd(&g_dbg_stats[i], v_); } while (0)
#else
#define OCC_STAT(i, v) do { } while (0)
#endif

struct RasterParams {
    OccScene sc;
    OccWorkspace ws;
    OccRenderOut out;
    const float* cam;
    int K;
    int ntx;  // tiles per image side
};

// XCD-major order of the (env, object) pairs: env e belongs to XCD group e % 8; group g holds MQ = 3*ceil(N/8) slots.
// All blocks of an env are then dequeued by waves of ONE XCD (when placement follows XCC_ID), so the face
// records of an object, staged again by every block they touch, are fetched into one L2 instead of eight.
__host__ __device__ __forceinline__ int xcd_slots(int n_env) { return 3 * ((n_env + 7) / 8); }
__device__ __forceinline__ int perm_to_eo(int p, int mq, int n_env) {
    const int g = p / mq, slot = p - g * mq;
    const int e = (slot / 3) * 8 + g;
    return e < n_env ? e * 3 + slot % 3 : -1;
}

// One block: exclusive prefix sum of the block counts of every (env, object) rect, in XCD-major order
// -> work-item offsets (8*MQ + 1 entries).
// shift = 0: work item = one OCC_BLOCK x OCC_BLOCK block of the rect (occ_raster_kernel); shift = 1: one 2 x 2 group of
// blocks = 8 x 8-pixel tile (occ_raster2_kernel).
__global__ __launch_bounds__(1024) void occ_scan_kernel(const int* __restrict__ objrect, const int* __restrict__ nrec,
                                                        int* __restrict__ offsets, int n_env, int shift) {
    __shared__ int s_part[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mq = xcd_slots(n_env), M = 8 * mq;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < M; base += 1024) {
        const int i = base + tid;
        int c = 0;
        const int eo = i < M ? perm_to_eo(i, mq, n_env) : -1;
        if (eo >= 0 && nrec[eo] > 0) {
            const int x0 = objrect[4 * eo], y0 = objrect[4 * eo + 1], x1 = objrect[4 * eo + 2], y1 = objrect[4 * eo + 3];
            if (x1 >= x0 && y1 >= y0 && x0 >= 0 && y0 >= 0) c = ((x1 >> shift) - (x0 >> shift) + 1) * ((y1 >> shift) - (y0 >> shift) + 1);
        }
        int incl = c;  // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int v = s_part[w];
            if (w < wave) woff += v;
            tot += v;
        }
        const int carry = s_carry;
        if (i < M) offsets[i] = carry + woff + incl - c;
        __syncthreads();
        if (tid == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (tid == 0) offsets[M] = s_carry;
}

// Work-item list of occ_raster2_kernel in cost order (OccWorkspace.order): one wave per (env, object).  The setup
// kernel has left, per tile, its cost class and rank among the object's tiles of that class, per object where its share
// of every class starts, and per XCD queue the class totals.  Queue q's items are laid out heaviest class first; item
// position = queue start + tiles of heavier classes + object's start in the class + rank.  Block 0 also publishes the
// queue boundaries.  (Positions inside a class depend on the order in which the setup blocks reserved their share:
// the ORDER of items may differ between runs, the results cannot - every item writes only its own pixels.)
__global__ __launch_bounds__(64) void occ_order_kernel(const int* __restrict__ objrect, const int* __restrict__ nrec,
                                                       uint32_t* __restrict__ order, int n_env, int img) {
    __shared__ uint32_t s_base[kOrdClasses];
    const int eo = blockIdx.x, env = eo / 3, q = env & 7, lane = threadIdx.x;
    const uint32_t* counts = order + kOrdCounts;
    // queue totals: lane l < 8 sums queue l (32 loads; 256 words, L2-resident)
    uint32_t qt = 0u;
    if (lane < 8)
        for (int c = 0; c < kOrdClasses; ++c) qt += counts[lane * kOrdClasses + c];
    uint32_t qincl = qt;  // inclusive prefix over the queues
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)qincl, d, 64);
        if (lane >= d) qincl += t;
    }
    if (eo == 0 && lane < 8) {
        order[lane + 1] = qincl;
        if (lane == 0) order[0] = 0u;
    }
    const uint32_t qstart = (uint32_t)__shfl((int)(qincl - qt), q, 64);
    // classes of my queue, heaviest first: tiles of the classes above c
    const uint32_t v = lane < kOrdClasses ? counts[q * kOrdClasses + lane] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < kOrdClasses; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += t;
    }
    const uint32_t tot = (uint32_t)__shfl((int)incl, kOrdClasses - 1, 64);
    if (lane < kOrdClasses) s_base[lane] = qstart + (tot - incl) + order[kOrdBlk + (size_t)eo * kOrdClasses + lane];
    __syncthreads();
    if (nrec[eo] <= 0) return;
    const int x0 = objrect[4 * eo], y0 = objrect[4 * eo + 1], x1 = objrect[4 * eo + 2], y1 = objrect[4 * eo + 3];
    if (x1 < x0 || y1 < y0 || x0 < 0 || y0 < 0) return;
    const int ntile = ((x1 >> 1) - (x0 >> 1) + 1) * ((y1 >> 1) - (y0 >> 1) + 1);
    const int T = (img / 8) * (img / 8);
    if (ntile > T) return;  // never true for a sane rect
    const uint32_t* __restrict__ tord = order + ord_tiles_word(n_env) + (size_t)eo * T;
    uint2* __restrict__ items = reinterpret_cast<uint2*>(order + ord_items_word(n_env, img));
    const uint32_t cap = (uint32_t)n_env * 3u * (uint32_t)T;
    for (int local = lane; local < ntile; local += 64) {
        const uint32_t w = tord[local];
        const uint32_t pos = s_base[w & 31u] + (w >> 5);
        if (pos < cap) items[pos] = make_uint2((uint32_t)eo, (uint32_t)local | (w & 31u) << 24);  // tile | cost class
    }
}

// ------------------------------------------------------------------------------------------
// raster kernel: one persistent wave64 per work item (env, object, 4x4-pixel block inside the object's rect)
// ------------------------------------------------------------------------------------------
// Lane layout: lane = 16 g + l.  l = pixel of the block (x = l & 3, y = l >> 2); g = FACE SLOT: one loop
// iteration evaluates four different faces (slots 4 it + g of the staged hit list) at the 16 pixels of the
// block, so a face whose footprint (~5x5 px with the blur margin) is about the size of the block no longer
// costs a 64-lane pass.  Every pixel's candidates are therefore spread over four lanes (each in face order):
// counts, products, tangent sums and the nearest hard face are folded across the four lanes at the end of the
// item, and the exact top-K selection works on the four lists jointly (topk_select4).
#ifndef OCC_RASTER_WAVES_PER_SIMD
#define OCC_RASTER_WAVES_PER_SIMD 5
#endif
template <bool SOFT, bool HARD, bool GRAD>
__global__ __launch_bounds__(64, OCC_RASTER_WAVES_PER_SIMD) void occ_raster_kernel(RasterParams P) {
    const int lane = threadIdx.x;
    const int g = lane >> 4, l = lane & 15;
    const int px = l & 3, py = l >> 2;
    const int S = P.sc.img;
    const float fS = (float)S;
    const int cap = P.sc.rec_cap;
    const int K = P.K;
    // per-wave K-buffer: OCC_LIST_CAP rows of 64 lane entries (key(z), 1-p, g_el, g_az), 16 B each
    float4* __restrict__ mylist = reinterpret_cast<float4*>(P.ws.lists) + (size_t)blockIdx.x * OCC_LIST_CAP * 64;
    __shared__ uint32_t s_hist[64 * kHistStride];
    // records of the faces that touch this block, gathered over as many 64-face chunks as fit, staged by
    // cooperative 16-B loads (one memory round trip per <= 64 staged faces)
    constexpr int kParts = GRAD ? kRecParts : (SOFT ? 5 : 4);  // float4 parts of a record that this variant reads
    constexpr int kStage = 40;  // 40 x 128 B = 5 KiB: the wave stays below 8 KiB of LDS -> 20 waves per CU
    __shared__ float4 s_stage[kStage * kRecParts];
    __shared__ int s_hit[kStage];  // record index of every staged face
    ciptr offs = as_const(P.ws.offsets);
    const int mq = xcd_slots(P.sc.n_env), MP = 8 * mq;
    // this wave's XCD (HW_REG_XCC_ID, bits 3:0); only steers WHICH queue is drained first - any value is correct
    const int my_xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
    int qround = 0;  // queues visited so far: own XCD's first, then the others (work stealing)
#ifdef OCC_DBG_BOUNDS
    int wd_iters = 0;
#endif

    for (;;) {
        OCC_WATCHDOG(35, qround, 0);
        int item = -1;
        while (qround < 8) {
            const int qq = (my_xcc + qround) & 7;
            const int qbeg = offs[qq * mq], qend = offs[(qq + 1) * mq];
            int t = qend;
            if (lane == 0 && qbeg < qend) t = qbeg + (int)atomicAdd(P.ws.queue + qq * 16, 1u);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t < qend) {
                item = t;
                break;
            }
            qround += 1;
        }
        if (item < 0) break;
        // (env, object) of this item: largest permuted index p with offsets[p] <= item
        int lo = 0, hi = MP;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (offs[mid] <= item) lo = mid; else hi = mid;
        }
        const int eo = perm_to_eo(lo, mq, P.sc.n_env);
        if (!OCC_BOUND(eo >= 0 && eo < 3 * P.sc.n_env, 1, eo, item)) continue;
        const int local = item - offs[lo];
        ciptr rect = as_const(P.ws.objrect + eo * 4);
        const int rw = rect[2] - rect[0] + 1;
        const int by = rect[1] + local / rw, bx = rect[0] + local % rw;
        const int x0b = bx * OCC_BLOCK, y0b = by * OCC_BLOCK;
        if (x0b < 0 || y0b < 0 || x0b + OCC_BLOCK > S || y0b + OCC_BLOCK > S) continue;  // never true for a sane rect
        const int xi = x0b + px, yi = y0b + py;
        // [P3D] pixel centre in NDC, +X left, +Y up
        const float xf = -1.0f + (2.0f * (float)(S - 1 - xi) + 1.0f) / fS;
        const float yf = -1.0f + (2.0f * (float)(S - 1 - yi) + 1.0f) / fS;
        const int n = as_const(P.ws.nrec + eo)[0];
        if (!OCC_BOUND(xi >= 0 && xi < S && yi >= 0 && yi < S && n >= 0 && n <= rec_span(P.ws, cap, eo).cap, 2, xi | (yi << 16), n)) continue;
        OCC_STAT(0, 1);              // work items
        const RecSpan span = rec_span(P.ws, cap, eo);
        const float* __restrict__ recs = P.ws.rec + span.base * OCC_REC_STRIDE;
        const uint4* __restrict__ bbs = reinterpret_cast<const uint4*>(P.ws.rec_bbox) + span.base;
        const uint4* __restrict__ scan = reinterpret_cast<const uint4*>(P.ws.scan) + span.base;

        float hz = 3.0e38f;
        int hrec = 0x7FFFFFFF;
        int count = 0;          // candidates in THIS lane's list
        float prod = 1.0f, sge = 0.f, sga = 0.f;
        bool thr_on = false;    // set once the pixel's lists have been compacted to its K nearest
        bool lim_on = false;    // pixel already holds >= K candidates
        // key bound of the pixel (equal in its four lanes): a later candidate needs key < bnd to matter.  Lowered to
        // the largest stored key once the pixel holds >= K candidates (that key bounds the K-th nearest from
        // above), and to the K-th nearest key itself whenever the lists are compacted
        uint32_t bnd = 0xFFFFFFFFu;
        uint32_t thrB = 0xFFFFFFFFu;  // block-wide skip key (wave-uniform): faces whose nearest vertex is not
                                      // nearer than this can change neither a pixel's K nearest nor its hard face
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;  // key range of this lane's stored candidates

        auto touches = [&](uint4 bb) {
            const int rx0 = bb.x & 0xFFFF, ry0 = bb.x >> 16, rx1 = bb.y & 0xFFFF, ry1 = bb.y >> 16;
            return (rx0 <= x0b + OCC_BLOCK - 1) && (rx1 >= x0b) && (ry0 <= y0b + OCC_BLOCK - 1) && (ry1 >= y0b);
        };

        auto commit = [&](bool cnd, float z, float qv, float ge, float ga) {
            const uint32_t key = zkey(z);
            bool acc = cnd && key < bnd;
            if (__ballot(acc && count >= OCC_LIST_CAP)) {
                // rare: a lane's list is full -> keep the pixel's K nearest (over its four lists), go on
                const bool full = px_any(count >= OCC_LIST_CAP);
                float pr, se, sa;
                uint32_t T;
                int kept;
                topk_select4<true, kHistBits>(mylist, s_hist, lane, count, K, full, kmin, kmax, pr, se, sa, T, kept);
                if (full) {
                    count = kept;
                    thr_on = true;
                    bnd = min(bnd, T);
                    kmax = T;
                    acc = cnd && (key < bnd);
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): leave no load pending across the hot loop
            }
            if (acc && OCC_BOUND(count < OCC_LIST_CAP, 3, count, item)) {
#ifndef OCC_DBG_NO_STORE  // timing experiment only
                // one 16-byte store per candidate; 32-bit offset from the wave-uniform base
                *reinterpret_cast<float4*>(reinterpret_cast<char*>(mylist) + (uint32_t)(count * 1024 + lane * 16)) =
                    make_float4(__uint_as_float(key), qv, ge, ga);
#endif
                kmin = min(kmin, key);
                kmax = max(kmax, key);
                count += 1;
                prod *= qv;
                sge += ge;
                sga += ga;
            }
        };

        int nst = 0;  // staged faces (wave-uniform)
        auto process_staged = [&]() {
            __syncthreads();
#ifndef OCC_DBG_NO_STAGE  // timing experiment only
            for (int idx = lane; idx < nst * kParts; idx += 64) {
                const int k = idx / kParts, part = idx - k * kParts;
                if (OCC_BOUND(s_hit[k] >= 0 && s_hit[k] < n, 4, s_hit[k], n))
                    s_stage[k * kRecParts + part] = reinterpret_cast<const float4*>(recs + (size_t)s_hit[k] * OCC_REC_STRIDE)[part];
            }
#endif
            __syncthreads();
            int niter = (nst + 3) >> 2;
            OCC_STAT(1, 1);      // staging rounds
            OCC_STAT(2, nst);    // staged records = (face, block) pairs
            OCC_STAT(3, niter);  // eval iterations
#ifdef OCC_DBG_NO_EVAL  // timing experiment only
            niter = 0;
#endif
            for (int it = 0; it < niter; ++it) {
                OCC_WATCHDOG(31, nst, n);
                const int slot = 4 * it + g;
                bool active = slot < nst;
                const int sidx = active ? slot : 0;
                const int j = s_hit[sidx];
                const float4* rs = &s_stage[sidx * kRecParts];
                Cand c1;
                eval_face<SOFT, GRAD>(OCC_REC_LOAD(rs, kParts), xf, yf, c1);
                const int flags = active ? __float_as_int(rs[2].z) : 0;
                // Clipped quad split in two: the pair is resolved where its SECOND half is visited.
                // A FIRST half whose partner also touches the block is skipped here; a SECOND half whose partner
                // touches the block evaluates the partner too and keeps one of them.
                if (__ballot(flags & (FLAG_PAIR_FIRST | FLAG_PAIR_SECOND))) {
                    const bool is_first = (flags & FLAG_PAIR_FIRST) != 0, is_second = (flags & FLAG_PAIR_SECOND) != 0;
                    bool partner = false;
                    if (is_first && j + 1 < n && OCC_BOUND(j >= 0, 5, j, n)) partner = touches(bbs[j + 1]);
                    if (is_second && j >= 1 && OCC_BOUND(j < n, 6, j, n)) partner = touches(bbs[j - 1]);
                    if (is_first && partner) active = false;
                    if (__ballot(is_second && partner)) {
                        const float4* r1 = reinterpret_cast<const float4*>(recs + (size_t)(is_second && partner ? j - 1 : j) *
                                                                          OCC_REC_STRIDE);
                        Cand cf;
                        eval_face<SOFT, GRAD>(OCC_REC_LOAD(r1, kParts), xf, yf, cf);
                        if (is_second && partner) {
                            if (HARD) {
                                if (cf.inside && (cf.zh < hz || (cf.zh == hz && j - 1 < hrec))) {
                                    hz = cf.zh;
                                    hrec = j - 1;
                                }
                            }
                            // [P3D]: the second half replaces the first iff its |d| is strictly smaller.  If both
                            // are closest to the diagonal they share (t1: edge (v1,v2), t2: edge (v0,v1)) the
                            // distances are equal in exact arithmetic: keep the first.
                            const bool shared_tie = (cf.amin == 2) && (c1.amin == 0);
                            const bool take2 = c1.cand && (!cf.cand || (!shared_tie && c1.ad < cf.ad));
                            if (!take2) {
                                const bool ins = c1.inside;
                                const float zh1 = c1.zh;
                                c1 = cf;
                                c1.inside = ins;  // the hard pass still sees the second half itself
                                c1.zh = zh1;
                            }
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): leave no load pending across the hot loop
                }
                c1.cand = c1.cand && active;
                c1.inside = c1.inside && active;
                if (HARD) {
                    if (c1.inside && (c1.zh < hz || (c1.zh == hz && j < hrec))) {
                        hz = c1.zh;
                        hrec = j;
                    }
                }
                if (SOFT) commit(c1.cand, c1.z, c1.q, c1.ge, c1.ga);
            }
            __syncthreads();
            nst = 0;
            // Front-to-back pruning (the scan order is ascending in the faces' nearest vertex depth and a
            // candidate's depth is never below it).  Once a pixel holds >= K candidates, the largest stored key
            // bounds its K-th nearest from above -> later candidates at or beyond it are dropped unseen; once this
            // holds for all 16 pixels, and every pixel has a hard face, faces starting beyond both bounds are
            // skipped altogether and the item ends at the first such chunk.
            if (SOFT) {
                const int ctot = px_sum_i(count);
                if (!lim_on && ctot >= K) {
                    lim_on = true;
                    bnd = min(bnd, px_max_u(kmax));
                }
            }
            uint32_t bound = 0xFFFFFFFFu;
            if (SOFT) bound = bnd;  // 0xFFFFFFFF until the pixel holds >= K candidates
            if (HARD) {
                const uint32_t hk = hz < 3.0e38f ? zkey(hz) : 0xFFFFFFFFu;  // every lane keeps its own nearest so far
                bound = SOFT ? max(bound, px_min_u(hk)) : px_min_u(hk);
            }
            // wave max over the 16 pixels (each pixel's four lanes agree)
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) bound = max(bound, (uint32_t)__shfl_xor((int)bound, m, 64));
            thrB = (uint32_t)__builtin_amdgcn_readfirstlane((int)bound);
        };

        // two-level scan: chunk boxes (one lane per 64-record chunk) -> candidate chunks -> their record boxes,
        // the next candidate chunk's row of boxes being fetched while the current one is processed
        const int nch = (n + 63) >> 6;
        const uint4* __restrict__ cbx = reinterpret_cast<const uint4*>(P.ws.rec_cbox) + span.cbox;
        const uint4 kEmptyBox = make_uint4(0xFFFFu, 0u, 0xFFFFFFFFu, 0u);  // x0 = 65535 > any pixel: never overlaps
        int cwin = -64;
        unsigned long long cmask = 0;
        auto next_chunk = [&]() -> int {
            while (!cmask) {
#ifdef OCC_DBG_BOUNDS
                if (++wd_iters > 4000000) { (void)OCC_BOUND(false, 34, cwin, nch); return -1; }
#endif
                cwin += 64;
                if (cwin >= nch) return -1;
                uint4 cb = kEmptyBox;
                if (cwin + lane < nch && OCC_BOUND(nch <= ((span.cap + 63) >> 6), 7, nch, span.cap)) cb = cbx[cwin + lane];
                cmask = __ballot(touches(cb) && cb.z < thrB);
            }
            const int bit = __builtin_ctzll(cmask);
            cmask &= cmask - 1;
            return cwin + bit;
        };
        int c = next_chunk();
        uint4 bb_cur = kEmptyBox;
        if (c >= 0 && c * 64 + lane < n && OCC_BOUND(c * 64 + lane < span.cap, 8, c, n)) bb_cur = scan[c * 64 + lane];
        while (c >= 0) {
            OCC_WATCHDOG(33, c, n);
            const int cn = next_chunk();
            uint4 bb_nxt = kEmptyBox;
            if (cn >= 0 && cn * 64 + lane < n && OCC_BOUND(cn * 64 + lane < span.cap, 9, cn, n)) bb_nxt = scan[cn * 64 + lane];
            const bool hit = touches(bb_cur) && bb_cur.z < thrB;
            const unsigned long long mask = __ballot(hit);
            OCC_STAT(5, 1);  // chunk rows scanned
            unsigned long long m = mask;
            const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            while (m) {  // a chunk may hold more hits than the staging buffer has room for
                OCC_WATCHDOG(32, nst, n);
                const int room = kStage - nst;
                const int cnt = __popcll(m);
                const int rank = __popcll(m & lt);
                const bool mine = (m >> lane) & 1ull;
                if (mine && rank < room) s_hit[nst + rank] = (int)bb_cur.w;
                if (cnt <= room) {
                    nst += cnt;
                    m = 0;
                } else {
                    nst = kStage;
                    m = __ballot(mine && rank >= room);
                    process_staged();
                }
            }
            c = cn;
            bb_cur = bb_nxt;
        }
        if (nst > 0) process_staged();

        // ---- fold the four lanes of every pixel ---------------------------------------------------------
        const size_t opix = ((size_t)eo * S + yi) * S + xi;
        if (SOFT) {
            const int ctot = px_sum_i(count);
#ifdef OCC_DBG_NO_TOPK  // timing experiment only: skip the exact selection (results wrong where count > K)
            const bool ovf = false;
#else
            const bool ovf = (ctot > K) || px_any(thr_on);
#endif
#ifdef OCC_DBG_STATS
            {
                const int cw = (int)wave_sum((float)count);
                const int co = (int)wave_sum(ovf ? (float)count : 0.f);
                OCC_STAT(4, cw);                                   // candidates stored
                OCC_STAT(6, co);                                   // ... of which in pixels that need selection
                OCC_STAT(7, __ballot(ovf) ? 1 : 0);                // items with at least one such pixel
            }
#endif
            if (__ballot(ovf)) {
                // more than K candidates: keep the K nearest in z, SURVEY A.4
                float pr, se, sa;
                uint32_t T;
                int kept;
                // the staging buffer is idle now: its LDS holds the wider (5-bit digit) histograms of the final
                // selection - one level less on average, each level being a pass over the lists in memory
                static_assert(sizeof(float4) * kStage * kRecParts >= 64 * ((1 << kHistBitsFinal) / 2 + 1) * 4, "hist");
                __syncthreads();
                topk_select4<false, kHistBitsFinal>(mylist, reinterpret_cast<uint32_t*>(s_stage), lane, count, K, ovf, kmin,
                                                    kmax, pr, se, sa, T, kept);
                __syncthreads();
                if (ovf) {
                    prod = pr;
                    sge = se;
                    sga = sa;
                }
            }
            prod = px_prod_f(prod);
            if (GRAD) {
                sge = px_sum_f(sge);
                sga = px_sum_f(sga);
            }
            if (g == 0) {
                P.ws.obj_alpha[opix] = 1.0f - prod;
                if (GRAD) {
                    // d alpha/d theta = -(A/sigma) * sum_k p_k d(d_k)/d theta   
                    const float coef = -prod * kInvSigma;
                    reinterpret_cast<float2*>(P.ws.obj_grad)[opix] = make_float2(coef * sge, coef * sga);
                }
            }
        }
        if (HARD) {
            // nearest face over the four lanes; equal depth -> smaller record (= face) index, like (pz, f) order
#pragma unroll
            for (int m = 16; m <= 32; m <<= 1) {
                const float oz = __shfl_xor(hz, m, 64);
                const int orr = __shfl_xor(hrec, m, 64);
                if (oz < hz || (oz == hz && orr < hrec)) {
                    hz = oz;
                    hrec = orr;
                }
            }
            if (g == 0) {
                P.ws.obj_hz[opix] = hz;
                P.ws.obj_hrec[opix] = (hrec == 0x7FFFFFFF) ? -1 : hrec;
            }
        }
    }
}
