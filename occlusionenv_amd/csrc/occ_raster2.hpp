// occ_raster2.hpp -- pair-enumerating raster kernel (round 2): one persistent wave64 per (env, object, 8x8-pixel tile).
// Part of the single translation unit occ_kernels.hip (included inside namespace occ; not a stand-alone header).
//
// What changed against occ_raster.hpp (4x4-pixel block x 4 face slots, every staged face evaluated at all 16 pixels,
// one lane-strided 16-byte K-buffer store per candidate into per-lane lists):
//
//   * LANE = (face, pixel) PAIR.  For every staged face the wave enumerates exactly the pixels of the face's pixel
//     bbox (setup kernel, +-sqrt(blur)) that fall into the tile - about 20 per ShapeNet-size face, of which ~60 % turn
//     out to be candidates (was: 64 lane-evaluations per (face, block), 26 % useful).  Pair descriptors
//     (staged slot, pixel) are expanded into LDS from a wave prefix sum over the faces' pair counts; a round of the
//     loop evaluates 64 consecutive pairs, each lane reading ITS face's record from an LDS image of the staged records
//     (structure-of-parts, stride 65: conflict-free staging writes, near-broadcast reads - consecutive lanes
//     share a face).
//   * PER-PIXEL STATE IN LDS, updated with LDS atomics by whichever lane evaluated the candidate: count, largest key,
//     sum of log2(1 - p_k) (the sigmoid-alpha PRODUCT in the log domain: prod = exp2(sum); one v_log_f32 per
//     candidate, no cross-lane routing of products), the two tangent sums, nearest hard face as one 64-bit
//     atomic min of (depth key << 32 | record).  A single wave owns the tile and LDS atomics of one instruction are
//     applied in lane order, so results are reproducible and independent of the batch.
//   * K-BUFFER = WAVE-COMPACTED LOG.  Accepted candidates of a round are appended contiguously (ballot rank) to the
//     wave's log in HBM/L2: key (4 B), owning pixel (1 B), payload (log2 q, g_el, g_az) - full-line coalesced stores
//     instead of 64 partial lines.  The log is only read when some pixel of the tile collected more than K candidates.
//   * COOPERATIVE EXACT TOP-K.  Radix select (5 bits per level) over the log with all 64 lanes sweeping it
//     contiguously; every entry bumps the LDS histogram of ITS pixel; the pixel's owner lane (lane = pixel) scans its
//     32 buckets and narrows its window.  Afterwards one more sweep re-accumulates the kept entries of the
//     overflowing pixels.  When the log fills up (thousands of candidates per pixel: far cameras, dense meshes) the
//     same machinery keeps each overflowing pixel's K nearest and compacts the log in place; pruning bounds as before.
//
// Semantics are those of occ_raster.hpp: same eval_face, same candidate rule, same K-nearest-by-z
// truncation, same clipped-pair rule, same hard nearest-face rule.

#ifndef OCC_LOG_CAP
#define OCC_LOG_CAP 12288  // log entries per wave; must exceed 64 * OCC_MAX_K + 64 (a compacted log plus one round)
#endif
static_assert(OCC_LOG_CAP >= 64 * OCC_MAX_K + 128, "OCC_LOG_CAP too small");
#define OCC_LOG_BYTES ((size_t)OCC_LOG_CAP * 24)  // 16 B payload + 4 B key + 1 B pixel tag (padded to 4)

constexpr int kT2 = 8;         // tile side in pixels (== OCC_TILE)
constexpr int kStg2 = 64;      // faces staged per batch: one per lane for the pair-count prefix sum
constexpr int kStgPad = 65;    // LDS stride (float4) between the parts of the staged records
constexpr int kPairCap = 1024; // pair descriptors per expansion window
constexpr int kSelBits = 5;    // radix-select digit: 32 u16 buckets = 16 dwords per pixel
constexpr int kSelDw = (1 << kSelBits) / 2;

struct WaveLog {
    float4* __restrict__ pay;    // (log2(1-p), g_el, g_az, -)
    uint32_t* __restrict__ key;  // order-preserving depth key
    uint8_t* __restrict__ tag;   // pixel of the tile (0..63)
};

__device__ __forceinline__ float unzkey(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__device__ __forceinline__ int lane_rank(unsigned long long m) {  // set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

#ifndef OCC_RASTER2_WAVES_PER_SIMD
#define OCC_RASTER2_WAVES_PER_SIMD 3
#endif

template <bool SOFT, bool HARD, bool GRAD>
__global__ __launch_bounds__(64, OCC_RASTER2_WAVES_PER_SIMD) void occ_raster2_kernel(RasterParams P) {
    const int lane = threadIdx.x;
    const int S = P.sc.img;
    const float fS = (float)S;
    const int cap = P.sc.rec_cap;
    const int K = P.K;
    constexpr int kParts = GRAD ? kRecParts : (SOFT ? 5 : 4);  // float4 parts of a record that this variant reads

    __shared__ float4 s_rec[kRecParts * kStgPad];  // staged records, part-major; idle during selection: histograms
    __shared__ int s_hit[kStg2];                   // record index of every staged face
    __shared__ uint2 s_box[kStg2];                 // its pixel bbox (xl | yl << 16, xh | yh << 16)
    __shared__ unsigned short s_desc[kPairCap];    // (staged slot << 6) | pixel of the tile
    __shared__ uint32_t s_cnt[64], s_kmax[64], s_bnd[64], s_selL[64], s_selSh[64], s_take[64];
    __shared__ float s_slog[64], s_sge[64], s_sga[64];
    __shared__ unsigned long long s_hard[64];
    __shared__ float s_xf[kT2], s_yf[kT2];

    WaveLog lg;
    {
        char* base = reinterpret_cast<char*>(P.ws.lists) + (size_t)blockIdx.x * OCC_LOG_BYTES;
        lg.pay = reinterpret_cast<float4*>(base);
        lg.key = reinterpret_cast<uint32_t*>(base + (size_t)OCC_LOG_CAP * 16);
        lg.tag = reinterpret_cast<uint8_t*>(base + (size_t)OCC_LOG_CAP * 20);
    }
    ciptr offs = as_const(P.ws.offsets);
    const int mq = xcd_slots(P.sc.n_env), MP = 8 * mq;
    const int my_xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;  // steers which queue is drained first only
    int qround = 0;

    for (;;) {
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
        int lo = 0, hi = MP;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (offs[mid] <= item) lo = mid; else hi = mid;
        }
        const int eo = perm_to_eo(lo, mq, P.sc.n_env);
        if (eo < 0 || eo >= 3 * P.sc.n_env) continue;
        const int local = item - offs[lo];
        ciptr rect = as_const(P.ws.objrect + eo * 4);  // in OCC_BLOCK (4-pixel) units; tiles are 2 x 2 blocks
        const int tx0 = rect[0] >> 1, ty0 = rect[1] >> 1, tw = (rect[2] >> 1) - tx0 + 1;
        const int x0t = (tx0 + local % tw) * kT2, y0t = (ty0 + local / tw) * kT2;
        if (x0t < 0 || y0t < 0 || x0t + kT2 > S || y0t + kT2 > S) continue;  // never true for a sane rect
        const int xi = x0t + (lane & 7), yi = y0t + (lane >> 3);  // the pixel this lane OWNS (lane = pixel)
        const int n = as_const(P.ws.nrec + eo)[0];
        const RecSpan span = rec_span(P.ws, cap, eo);
        if (n < 0 || n > span.cap) continue;
        OCC_STAT(0, 1);  // work items
        const float4* __restrict__ recs4 = reinterpret_cast<const float4*>(P.ws.rec + span.base * OCC_REC_STRIDE);
        const uint4* __restrict__ bbs = reinterpret_cast<const uint4*>(P.ws.rec_bbox) + span.base;
        const uint4* __restrict__ scan = reinterpret_cast<const uint4*>(P.ws.scan) + span.base;

        __syncthreads();  // the previous item's readers of the LDS state are done
        if (lane < kT2) {
            // [P3D] pixel centres in NDC, +X left, +Y up: same expression as the oracle
            s_xf[lane] = -1.0f + (2.0f * (float)(S - 1 - (x0t + lane)) + 1.0f) / fS;
            s_yf[lane] = -1.0f + (2.0f * (float)(S - 1 - (y0t + lane)) + 1.0f) / fS;
        }
        s_cnt[lane] = 0u;
        s_kmax[lane] = 0u;
        s_bnd[lane] = 0xFFFFFFFFu;
        s_slog[lane] = 0.f;
        s_sge[lane] = 0.f;
        s_sga[lane] = 0.f;
        s_hard[lane] = ~0ull;
        int nlog = 0;               // entries in the wave's log (wave-uniform)
        bool lim_on = false;        // this lane's pixel already holds >= K candidates
        uint32_t bnd = 0xFFFFFFFFu; // key bound of this lane's pixel (copy of s_bnd[lane])
        uint32_t thrB = 0xFFFFFFFFu;      // tile-wide skip key (wave-uniform)
        uint32_t kmin_tile = 0xFFFFFFFFu; // smallest depth key any candidate of this tile can have (chunk boxes)

        auto touches = [&](uint4 bb) {
            const int rx0 = bb.x & 0xFFFF, ry0 = bb.x >> 16, rx1 = bb.y & 0xFFFF, ry1 = bb.y >> 16;
            return (rx0 <= x0t + kT2 - 1) && (rx1 >= x0t) && (ry0 <= y0t + kT2 - 1) && (ry1 >= y0t);
        };

        // ---- exact top-K over the log for every pixel holding more than K entries ---------------------------
        // Leaves, for those pixels, count / kmax / log-product / tangent sums of their K nearest in the LDS state;
        // COMPACT also rewrites the log so that it holds exactly the entries still accounted for.
        auto select_topk = [&](const bool compact) {
            __syncthreads();
            uint32_t* hist = reinterpret_cast<uint32_t*>(s_rec);  // 64 pixels x kSelDw dwords (u16 buckets)
            const int cnt = (int)s_cnt[lane];
            const bool ovf = cnt > K;
            // window start: a candidate's depth is a convex combination of its face's vertex depths, so no key lies
            // below the smallest chunk-box key of the tile - up to rounding, hence the margin of 4096 ulp
            uint32_t L = kmin_tile > 4096u ? kmin_tile - 4096u : 0u;
            int need = K, sh = 0;
            {
                const uint32_t kmx = s_kmax[lane];
                const uint32_t range = kmx >= L ? kmx - L : 0u;
                sh = range ? max(0, (32 - __builtin_clz(range)) - kSelBits) : 0;
            }
            bool done = !ovf, takeall = false;
            s_selSh[lane] = done ? 255u : (uint32_t)sh;
            s_selL[lane] = L;
            while (__ballot(!done)) {
#pragma unroll
                for (int i = 0; i < kSelDw; i += 4) reinterpret_cast<uint4*>(hist + lane * kSelDw)[i >> 2] = make_uint4(0u, 0u, 0u, 0u);
                __syncthreads();
                for (int e0 = 0; e0 < nlog; e0 += 64) {
                    const int e = e0 + lane;
                    if (e < nlog) {
                        const uint32_t k = lg.key[e];
                        const int t = lg.tag[e];
                        const uint32_t shp = s_selSh[t];
                        if (shp < 32u) {
                            const uint32_t Lp = s_selL[t];
                            const uint32_t d = (k - Lp) >> shp;
                            if (k >= Lp && d < (1u << kSelBits)) atomicAdd(&hist[t * kSelDw + (d >> 1)], 1u << (16 * (d & 1u)));
                        }
                    }
                }
                __syncthreads();
                if (!done) {
                    int cum = 0, bstar = (1 << kSelBits) - 1, mstar = 0, cumb = 0;
                    bool found = false;
#pragma unroll
                    for (int i = 0; i < kSelDw; ++i) {
                        const uint32_t w = hist[lane * kSelDw + i];
                        const int c0 = (int)(w & 0xFFFFu), c1 = (int)(w >> 16);
                        if (!found && cum + c0 >= need) { found = true; bstar = 2 * i; mstar = c0; cumb = cum; }
                        cum += c0;
                        if (!found && cum + c1 >= need) { found = true; bstar = 2 * i + 1; mstar = c1; cumb = cum; }
                        cum += c1;
                    }
                    need -= cumb;
                    L += (uint32_t)bstar << sh;
                    if (mstar == need || sh == 0 || !found) {
                        done = true;
                        takeall = (mstar == need) || !found;
                    } else {
                        sh = max(0, sh - kSelBits);
                    }
                    s_selSh[lane] = done ? 255u : (uint32_t)sh;
                    s_selL[lane] = L;
                }
                __syncthreads();
            }
            // final window of an overflowing pixel: keys < L are kept, of the bucket [L, L + 2^sh) `need` more
            // (all of it when takeall; exact-key ties otherwise, served in log order = scan order)
            s_selL[lane] = L;
            s_selSh[lane] = ovf ? (uint32_t)sh : 255u;
            s_take[lane] = takeall ? 0x7FFFFFFFu : (uint32_t)max(need, 0);
            if (ovf) {
                s_cnt[lane] = 0u;
                s_kmax[lane] = 0u;
                s_slog[lane] = 0.f;
                s_sge[lane] = 0.f;
                s_sga[lane] = 0.f;
            }
            __syncthreads();
            int wr = 0;
            for (int e0 = 0; e0 < nlog; e0 += 64) {
                const int e = e0 + lane;
                bool keep = false, readd = false;
                uint32_t k = 0u;
                int t = 0;
                float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < nlog) {
                    k = lg.key[e];
                    t = lg.tag[e];
                    const uint32_t shp = s_selSh[t];
                    if (shp >= 32u) {
                        keep = true;  // pixel not overflowing: its entries stay, its sums are already right
                    } else {
                        const uint32_t Lp = s_selL[t];
                        if (k < Lp) {
                            readd = true;
                        } else if (((k - Lp) >> shp) == 0u) {
                            // atomicSub returns the old value: the first `take` arrivals are kept
                            readd = (int)atomicSub(&s_take[t], 1u) > 0;
                        }
                        keep = readd;
                    }
                    if (readd || (compact && keep)) pv = lg.pay[e];
                }
                if (readd) {
                    atomicAdd(&s_cnt[t], 1u);
                    atomicMax(&s_kmax[t], k);
                    atomicAdd(&s_slog[t], pv.x);
                    if (GRAD) {
                        atomicAdd(&s_sge[t], pv.y);
                        atomicAdd(&s_sga[t], pv.z);
                    }
                }
                if (compact) {
                    const unsigned long long m = __ballot(keep);
                    if (keep) {
                        const int w = wr + lane_rank(m);  // w <= e: never overtakes the reads of a later iteration
                        lg.key[w] = k;
                        lg.tag[w] = (uint8_t)t;
                        lg.pay[w] = pv;
                    }
                    wr += __popcll(m);
                }
            }
            if (compact) {
                nlog = wr;
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the rewritten log is in place before it grows again
            }
            __syncthreads();
        };

        // ---- one staged batch: records -> LDS, pair expansion, evaluation rounds ------------------------------
        int nst = 0;  // staged faces (wave-uniform)
        auto process_batch = [&]() {
            __syncthreads();
            for (int idx = lane; idx < nst * kRecParts; idx += 64) {
                const int k = idx >> 3, part = idx & 7;
                if (part < kParts) s_rec[part * kStgPad + k] = recs4[(size_t)s_hit[k] * kRecParts + part];
            }
            // pixels of this lane's face inside the tile: pair count, prefix sum over the staged faces
            int c = 0, cx0 = 0, cy0 = 0, cw = 1;
            if (lane < nst) {
                const uint2 bb = s_box[lane];
                cx0 = max((int)(bb.x & 0xFFFFu), x0t);
                cy0 = max((int)(bb.x >> 16), y0t);
                const int cx1 = min((int)(bb.y & 0xFFFFu), x0t + kT2 - 1), cy1 = min((int)(bb.y >> 16), y0t + kT2 - 1);
                cw = cx1 - cx0 + 1;
                c = max(cw, 0) * max(cy1 - cy0 + 1, 0);
                cw = max(cw, 1);
            }
            int incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(incl, d, 64);
                if (lane >= d) incl += t;
            }
            const int pre = incl - c;
            const int ptot = __shfl(incl, 63, 64);
            const int cmax = wave_max_i(c);
            OCC_STAT(1, 1);      // staged batches
            OCC_STAT(2, nst);    // staged records = (face, tile) pairs
            OCC_STAT(3, (ptot + 63) >> 6);  // evaluation rounds
            for (int wbase = 0; wbase < ptot; wbase += kPairCap) {
                __syncthreads();  // previous window's descriptors consumed; records staged
                {
                    int xx = cx0 - x0t, yy = cy0 - y0t;
                    const int xb = xx, xe = xx + cw;
                    for (int j = 0; j < cmax; ++j) {
                        if (j < c) {
                            const int p = pre + j - wbase;
                            if ((unsigned)p < (unsigned)kPairCap) s_desc[p] = (unsigned short)((lane << 6) | (yy << 3) | xx);
                            xx += 1;
                            if (xx == xe) {
                                xx = xb;
                                yy += 1;
                            }
                        }
                    }
                }
                __syncthreads();
                const int wend = min(ptot - wbase, kPairCap);
                for (int p0 = 0; p0 < wend; p0 += 64) {
                    if (SOFT && nlog + 64 > OCC_LOG_CAP) {
                        // rare: the log is full -> keep every overflowing pixel's K nearest, go on with tighter bounds
                        select_topk(true);
                        const int cn = (int)s_cnt[lane];
                        if (cn >= K) {
                            lim_on = true;
                            bnd = min(bnd, s_kmax[lane]);
                            s_bnd[lane] = bnd;
                        }
                        // the histograms lived in s_rec: stage this batch's records again
                        __syncthreads();
                        for (int idx = lane; idx < nst * kRecParts; idx += 64) {
                            const int k = idx >> 3, part = idx & 7;
                            if (part < kParts) s_rec[part * kStgPad + k] = recs4[(size_t)s_hit[k] * kRecParts + part];
                        }
                        __syncthreads();
                    }
                    const bool live = p0 + lane < wend;
                    const uint32_t d = live ? s_desc[p0 + lane] : 0u;
                    const int f = d >> 6, pix = d & 63;
                    const int j = s_hit[f];
                    const float xf = s_xf[d & 7], yf = s_yf[(d >> 3) & 7];
                    const float4* rs = &s_rec[f];
                    Cand c1;
                    eval_face<SOFT, GRAD>(rs[0], rs[kStgPad], rs[2 * kStgPad], rs[3 * kStgPad],
                                          kParts > 4 ? rs[4 * kStgPad] : make_float4(0, 0, 0, 0),
                                          kParts > 5 ? rs[5 * kStgPad] : make_float4(0, 0, 0, 0),
                                          kParts > 5 ? rs[6 * kStgPad] : make_float4(0, 0, 0, 0),
                                          kParts > 5 ? rs[7 * kStgPad] : make_float4(0, 0, 0, 0), xf, yf, c1);
                    const int flags = live ? __float_as_int(rs[2 * kStgPad].z) : 0;
                    bool emit = live;
                    // Clipped quad split in two: only one half may enter a pixel's list.  Both halves'
                    // lanes look at both halves; the SECOND half's lane emits the winner when both are candidates,
                    // a half whose partner is no candidate at this pixel emits itself.
                    if (__ballot(flags & (FLAG_PAIR_FIRST | FLAG_PAIR_SECOND))) {
                        const bool is_first = (flags & FLAG_PAIR_FIRST) != 0, is_second = (flags & FLAG_PAIR_SECOND) != 0;
                        if ((is_first && j + 1 < n) || (is_second && j >= 1)) {
                            const float4* r1 = recs4 + (size_t)(is_first ? j + 1 : j - 1) * kRecParts;
                            Cand cp;
                            eval_face<SOFT, GRAD>(OCC_REC_LOAD(r1, kParts), xf, yf, cp);
                            if (is_first) {
                                if (cp.cand) c1.cand = false;  // the second half's lane decides
                            } else if (cp.cand && c1.cand) {
                                // [P3D]: the second half replaces the first iff its |d| is strictly smaller; both
                                // closest to the shared diagonal (first: edge (v1,v2), second: edge (v0,v1)) = equal
                                // in exact arithmetic: keep the first
                                const bool shared_tie = (cp.amin == 2) && (c1.amin == 0);
                                const bool take2 = !shared_tie && c1.ad < cp.ad;
                                if (!take2) {
                                    const bool ins = c1.inside;
                                    const float zh1 = c1.zh;
                                    c1 = cp;
                                    c1.inside = ins;  // the hard pass still sees the second half itself
                                    c1.zh = zh1;
                                }
                            }
                        }
                        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): leave no load pending across the hot loop
                    }
                    if (HARD) {
                        if (emit && c1.inside)
                            atomicMin(&s_hard[pix], ((unsigned long long)zkey(c1.zh) << 32) | (unsigned)j);
                    }
                    if (SOFT) {
                        const uint32_t key = zkey(c1.z);
                        const bool acc = emit && c1.cand && key < s_bnd[pix];
                        const unsigned long long m = __ballot(acc);
                        if (m) {
                            if (acc) {
                                const int e = nlog + lane_rank(m);
                                // log-domain product: prod(1 - p_k) = exp2(sum log2(1 - p_k)); log2(0) = -inf -> prod 0
                                const float lq = __builtin_amdgcn_logf(c1.q);
                                lg.key[e] = key;
                                lg.tag[e] = (uint8_t)pix;
                                lg.pay[e] = make_float4(lq, c1.ge, c1.ga, 0.f);
                                atomicAdd(&s_cnt[pix], 1u);
                                atomicMax(&s_kmax[pix], key);
                                atomicAdd(&s_slog[pix], lq);
                                if (GRAD) {
                                    atomicAdd(&s_sge[pix], c1.ge);
                                    atomicAdd(&s_sga[pix], c1.ga);
                                }
                            }
                            nlog += __popcll(m);
                        }
                    }
                }
            }
            __syncthreads();
            nst = 0;
            // Front-to-back pruning (SURVEY A.4 keeps the K smallest depths): once a pixel holds >= K candidates its
            // largest stored key bounds its K-th nearest from above, later candidates at or beyond it are dropped
            // unseen; once that holds for all 64 pixels (and every pixel has a hard face) whole faces / chunks whose
            // nearest vertex lies beyond every bound are skipped.
            uint32_t bound = 0xFFFFFFFFu;
            if (SOFT) {
                if (!lim_on && (int)s_cnt[lane] >= K) {
                    lim_on = true;
                    bnd = min(bnd, s_kmax[lane]);
                    s_bnd[lane] = bnd;
                }
                bound = bnd;
            }
            if (HARD) {
                const uint32_t hk = (uint32_t)(s_hard[lane] >> 32);  // 0xFFFFFFFF while the pixel has no face
                bound = SOFT ? max(bound, hk) : hk;
            }
#pragma unroll
            for (int mm = 32; mm >= 1; mm >>= 1) bound = max(bound, (uint32_t)__shfl_xor((int)bound, mm, 64));
            thrB = (uint32_t)__builtin_amdgcn_readfirstlane((int)bound);
        };

        // ---- two-level scan: chunk boxes -> candidate chunks -> their rows (as in occ_raster.hpp) ----------------
        const int nch = (n + 63) >> 6;
        const uint4* __restrict__ cbx = reinterpret_cast<const uint4*>(P.ws.rec_cbox) + span.cbox;
        const uint4 kEmptyBox = make_uint4(0xFFFFu, 0u, 0xFFFFFFFFu, 0u);  // x0 = 65535 > any pixel: never overlaps
        int cwin = -64;
        unsigned long long cmask = 0;
        auto next_chunk = [&]() -> int {
            while (!cmask) {
                cwin += 64;
                if (cwin >= nch) return -1;
                uint4 cb = kEmptyBox;
                if (cwin + lane < nch) cb = cbx[cwin + lane];
                const bool hit = touches(cb);
                uint32_t km = hit ? cb.z : 0xFFFFFFFFu;
#pragma unroll
                for (int mm = 32; mm >= 1; mm >>= 1) km = min(km, (uint32_t)__shfl_xor((int)km, mm, 64));
                kmin_tile = min(kmin_tile, (uint32_t)__builtin_amdgcn_readfirstlane((int)km));
                cmask = __ballot(hit && cb.z < thrB);
            }
            const int bit = __builtin_ctzll(cmask);
            cmask &= cmask - 1;
            return cwin + bit;
        };
        int c = next_chunk();
        uint4 bb_cur = kEmptyBox;
        if (c >= 0 && c * 64 + lane < n) bb_cur = scan[c * 64 + lane];
        while (c >= 0) {
            const int cn = next_chunk();
            uint4 bb_nxt = kEmptyBox;
            if (cn >= 0 && cn * 64 + lane < n) bb_nxt = scan[cn * 64 + lane];
            const bool hit = touches(bb_cur) && bb_cur.z < thrB;
            unsigned long long m = __ballot(hit);
            OCC_STAT(5, 1);  // chunk rows scanned
            while (m) {  // a chunk may hold more hits than the staging buffer has room for
                const int room = kStg2 - nst;
                const int cnt = __popcll(m);
                const int rank = lane_rank(m);
                const bool mine = (m >> lane) & 1ull;
                if (mine && rank < room) {
                    s_hit[nst + rank] = (int)bb_cur.w;
                    s_box[nst + rank] = make_uint2(bb_cur.x, bb_cur.y);
                }
                if (cnt <= room) {
                    nst += cnt;
                    m = 0;
                } else {
                    nst = kStg2;
                    m = __ballot(mine && rank >= room);
                    process_batch();
                }
            }
            c = cn;
            bb_cur = bb_nxt;
        }
        if (nst > 0) process_batch();

        // ---- per-pixel results (lane = pixel) ---------------------------------------------------------------------
        const size_t opix = ((size_t)eo * S + yi) * S + xi;
        if (SOFT) {
            __syncthreads();
            const bool ovf = (int)s_cnt[lane] > K;
#ifdef OCC_DBG_STATS
            {
                const int cw_ = (int)wave_sum((float)s_cnt[lane]);
                const int co_ = (int)wave_sum(ovf ? (float)s_cnt[lane] : 0.f);
                OCC_STAT(4, cw_);                    // candidates accounted for
                OCC_STAT(6, co_);                    // ... of which in pixels that need selection
                OCC_STAT(7, __ballot(ovf) ? 1 : 0);  // items with at least one such pixel
            }
#endif
            if (__ballot(ovf)) select_topk(false);  // more than K candidates: keep the K nearest in z, SURVEY A.4
            const float prod = __builtin_amdgcn_exp2f(s_slog[lane]);
            P.ws.obj_alpha[opix] = 1.0f - prod;
            if (GRAD) {
                // d alpha/d theta = -(A/sigma) * sum_k p_k d(d_k)/d theta   
                const float coef = -prod * kInvSigma;
                reinterpret_cast<float2*>(P.ws.obj_grad)[opix] = make_float2(coef * s_sge[lane], coef * s_sga[lane]);
            }
        }
        if (HARD) {
            const unsigned long long h = s_hard[lane];
            const bool any = h != ~0ull;
            P.ws.obj_hz[opix] = any ? unzkey((uint32_t)(h >> 32)) : 3.0e38f;
            P.ws.obj_hrec[opix] = any ? (int)(uint32_t)h : -1;
        }
    }
}
