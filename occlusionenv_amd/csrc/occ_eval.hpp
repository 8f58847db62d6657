// occ_eval.hpp -- what every raster kernel shares: the per-(face, pixel) evaluation (eval_face), depth keys, the launch
// parameters, the XCD-major work-item order and its two small kernels (occ_scan_kernel, occ_order_kernel).
// Part of the single translation unit occ_kernels.hip (included inside namespace occ; not a stand-alone header).

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

// Evaluate one projected face at this lane's pixel centre.
// Restates [P3D] CheckPixelInsideFace and, for GRAD, the dists part of
// RasterizeMeshesBackward (A.5) pushed forward along the two vertex tangents.
// Record slot map (occ_constants.h):
//   a = x0 y0 z0 x1 | b = y1 z1 x2 y2 | c = z2 id flags inv_area | d = bbox | e = il01 il02 il12 spec |
//   parts 5, 6, 7 = tangents of v0, v1, v2 (dx/del dy/del dx/daz dy/daz)      -- 8 parts = 128 bytes = one cache line
// The first five parts travel as SSA values (by value, never through a struct in memory: a select between two loads of
// one stack object gets folded into a dynamically indexed load, which pins the object in scratch).  The tangents are
// FETCHED once the closest edge is known - tan(v) returns the tangent part of vertex v (0, 1, 2; per lane) - two
// 16-byte reads at a computed address instead of three up front and sixteen selects among their components.
// Part d (the float bbox +- sqrt(blur) of [P3D]'s early reject) is NOT read here: a pixel centre outside that box is
// farther than sqrt(blur) from the triangle, so its squared distance fails `dist < blur` by itself, and a centre inside
// the triangle is inside the box - the reject never changes an outcome (up to a pixel on the blur boundary to within
// rounding, which the parity checker classes as the tie it is).  Four compares, one LDS read and four registers less.
#define OCC_REC_PARAMS float4 ra, float4 rb, float4 rc, float4 re
#define OCC_REC_LOAD(src, PARTS) (src)[0], (src)[1], (src)[2], ((PARTS) > 4 ? (src)[4] : make_float4(0, 0, 0, 0))

// Branch-free (the pair kernel only visits pixels of the face's pixel bbox, where a divergent early exit costs more
// than it saves): everything is evaluated, candidates are masked at the end.
template <bool SOFT, bool GRAD, class TanFn>
__device__ __forceinline__ void eval_face(OCC_REC_PARAMS, float xf, float yf, Cand& c, TanFn tan) {
    c.cand = false;
    c.inside = false;
    c.z = c.zh = c.ad = 0.f;
    c.q = 1.f;
    c.ge = c.ga = 0.f;
    c.amin = 0;
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
    c.inside = inside && !(c.zh < 0.0f);
    if (!SOFT) return;
    // clipped barycentrics -> soft depth
    // max(x, 0) as med3(x, 0, +inf): one instruction (fmaxf canonicalises its operand first: a second v_max each)
    float c0 = __builtin_amdgcn_fmed3f(p0, 0.f, __builtin_inff()), c1 = __builtin_amdgcn_fmed3f(p1, 0.f, __builtin_inff()),
          c2 = __builtin_amdgcn_fmed3f(p2, 0.f, __builtin_inff());
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
    const bool cand = !(pz < 0.0f) && (inside || dist < kBlurRadius);
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
        // tangent of the projected point: (1-t) a' + t b'; a = v1 for edge (v1,v2) else v0, b = v1 for edge (v0,v1) else v2
        const float4 ta = tan(s12 ? 1 : 0), tb4 = tan(s01 ? 1 : 2);
        const float mxe = ta.x + tb * (tb4.x - ta.x), mye = ta.y + tb * (tb4.y - ta.y);
        const float mxa = ta.z + tb * (tb4.z - ta.z), mya = ta.w + tb * (tb4.w - ta.w);
        const float sp = inside ? -p : p;  // (exactly one of s01 / s02 / s12 holds for finite distances)
        c.ge = sp * (gx * mxe + gy * mye);
        c.ga = sp * (gx * mxa + gy * mya);
    }
}

// Order-preserving integer image of a depth that is known not to be negative (every candidate: pz < 0 is rejected): the
// float's bits with the sign bit set (the general image, used by the setup kernel for vertex depths, complements
// negative floats; the two agree on z >= 0).  unzkey() in occ_raster2.hpp inverts it.
__device__ __forceinline__ uint32_t zkey_pos(float z) { return __float_as_uint(z) | 0x80000000u; }

#ifdef OCC_DBG_BOUNDS  // diagnostic build only: index checks at every memory access of the raster kernel
__device__ int g_dbg_fault[8];
#define OCC_BOUND(cond, code, v0, v1)                                          \
    ((cond) ? true                                                             \
            : ((atomicCAS(&g_dbg_fault[0], 0, (code)) == 0                     \
                    ? (g_dbg_fault[1] = (int)blockIdx.x, g_dbg_fault[2] = (int)threadIdx.x, g_dbg_fault[3] = (int)(v0), \
                       g_dbg_fault[4] = (int)(v1), 0)                          \
                    : 0),                                                      \
               false))
#else
#define OCC_BOUND(cond, code, v0, v1) true
#endif

#ifdef OCC_DBG_STATS  // diagnostic build only: loop trip counts of the raster kernel
__device__ unsigned long long g_dbg_stats[32];
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
    // Small launches (a single env, a handful of reset candidates): every 8x8 tile is handed out as 1 << split_log2
    // work items of 8 >> split_log2 pixel rows each, so that one env's render is not bounded by the few waves its tiles
    // would occupy (a tile that needs the exact top-K keeps one wave busy for up to a millisecond).  All sub-items
    // stage the SAME faces in the SAME batches as the whole tile would - only the pairs of their own rows are
    // evaluated - so every pixel sees the same candidates in the same order with the same accumulator copies:
    // results do not depend on the split.
    int split_log2;
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
// shift = 0: work item = one OCC_BLOCK x OCC_BLOCK block of the rect; shift = 1 (what occ_render passes): one 2 x 2 group of
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
// (two halves around a block barrier: order_bases fills s_base - by the 64 lanes of ONE wave - order_items uses it)
__device__ __forceinline__ void order_bases(uint32_t* __restrict__ order, uint32_t* s_base, int eo, int lane) {
    const int env = eo / 3, q = env & 7;
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
}
__device__ __forceinline__ void order_items(const int* __restrict__ objrect, const int* __restrict__ nrec,
                                            uint32_t* __restrict__ order, const uint32_t* s_base, int n_env, int img, int eo, int lane) {
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
__global__ __launch_bounds__(64) void occ_order_kernel(const int* __restrict__ objrect, const int* __restrict__ nrec,
                                                       uint32_t* __restrict__ order, int n_env, int img) {
    __shared__ uint32_t s_base[kOrdClasses];
    order_bases(order, s_base, blockIdx.x, threadIdx.x);
    __syncthreads();
    order_items(objrect, nrec, order, s_base, n_env, img, blockIdx.x, threadIdx.x);
}
// Round 5: ONE launch for the two small per-object passes between the setup and the raster kernel - the work-item order
// (wave 0 of the block) and, for dense objects, the front-to-back sort of the scan rows (all 256 threads; occ_sort_kernel's
// body) - instead of two dependent launches of ~5 us each.
__global__ __launch_bounds__(256) void occ_sort_order_kernel(OccScene sc, OccWorkspace ws, int sort_cap) {
    __shared__ uint32_t s_base[kOrdClasses];
    const int eo = blockIdx.x, tid = threadIdx.x;
    if (tid < 64) order_bases(ws.order, s_base, eo, tid);
    __syncthreads();
    if (tid < 64) order_items(ws.objrect, ws.nrec, ws.order, s_base, sc.n_env, sc.img, eo, tid);
    if (sort_cap > 0) sort_object(sc, ws, sort_cap, eo);  // (a launch with sort_cap = 0 has 64 threads per block and no dynamic LDS)
}
