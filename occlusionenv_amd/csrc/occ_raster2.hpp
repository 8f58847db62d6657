// occ_raster2.hpp -- the raster kernel: persistent wave64s, work item = (env, object, 8x8-pixel tile) - or a group of its
// pixel rows in a small launch - taken heaviest first from the XCD's queue (occ_order_kernel).
// Part of the single translation unit occ_kernels.hip (included inside namespace occ; not a stand-alone header).
//
// How it works (round 2; the round-1 kernel - 4x4-pixel block x 4 face slots, every staged face evaluated at all 16
// pixels, one lane-strided 16-byte K-buffer store per candidate into per-lane lists - was retired in round 3):
//
//   * LANE = (face, pixel) PAIR.  For every staged face the wave enumerates exactly the pixels of the face's pixel
//     bbox (setup kernel, +-sqrt(blur)) that fall into the tile - about 10 per ShapeNet-size face and tile, of which
//     68 % turn out to be candidates (round 1: 64 lane-evaluations per (face, block), 26 % useful) - minus the box corners
//     that the setup kernel marked as beyond the blur disc (finish_tri: corner cut; 76 % candidates).  Every face marks
//     its first pair in a byte map built from a wave prefix sum over the faces' pair counts; a round of the loop
//     evaluates 64 consecutive pairs (face = running count + marks at or below the lane, pixel = first + jj +
//     row (8 - width) from one multiply by 2^15 / width), each lane reading ITS face's record from an LDS image of the
//     staged records (part-major, stride 33: conflict-free staging writes, near-broadcast reads - consecutive lanes
//     share a face).  Lanes past the batch's last pair run along unmasked; every side effect is guarded.
//   * PER-PIXEL STATE IN LDS, updated by whichever lane evaluated the candidate with PLAIN read-modify-write (LDS
//     atomics retire about one lane per cycle on gfx950: five of them per candidate cost more than the evaluation,
//     measured 2.1 of 4.9 ms): (product of (1 - p_k), tangent sums, count) as one float4 and the largest key, in FOUR
//     copies selected by (staged face & 3).  Lanes of one face hit distinct pixels, consecutive faces distinct
//     copies, so a round is applied in sub-passes of four consecutive faces (usually two) and no two lanes of an
//     instruction ever share an address; LDS operations of a wave retire in order.  The owner lane (lane = pixel)
//     folds the four copies in a fixed order: results are reproducible and independent of the batch.  Nearest hard
//     face: one 64-bit LDS atomic min of (depth key << 32 | record) per INSIDE pair (few).
//   * K-BUFFER = WAVE-COMPACTED LOG.  Accepted candidates of a round are appended contiguously (ballot rank) to the
//     wave's log in HBM/L2: (key, owning pixel | face sequence number << 6) 8 B + payload (1 - p, g_el, g_az) 12 B -
//     full-line coalesced stores instead of 64 partial lines (measured: free next to the evaluation).  The log is only
//     read when some pixel of the tile collected more than K candidates, and not even written for a tile that at most
//     K faces touch (its cost class, carried by the work item, says so).
//   * BATCHES ARE PIPELINED.  While a batch of <= 32 faces is evaluated, the scan has already produced the next
//     batch's hit list (second list in LDS, two rows of the scan in flight) and its records are in flight into
//     registers; they are committed to LDS when the next batch starts, behind an explicit s_waitcnt + sched_barrier
//     (left to itself hipcc hoists the next loads above the wait for the current ones and then waits for both).
//     Without the pipelining: +7 % (round 3: 2.45 vs 2.29 ms).
//   * COOPERATIVE EXACT TOP-K.  Radix select (5 bits per level) over the log with all 64 lanes sweeping it
//     contiguously; every entry bumps the LDS histogram of ITS pixel (odd stride per pixel: a face's pixels bump the same
//     bucket); the pixel's owner lane (lane = pixel) scans its 32 buckets and narrows its window; sweeps keep two groups
//     of four 512-byte loads in flight (a sweep with one dependent load per iteration is pure memory latency).
//     Afterwards one more sweep re-accumulates the kept entries of the overflowing pixels into the pixel's four
//     accumulator copies by plain read-modify-write.  Round 5: the copy is chosen by the pixel's ARRIVAL INDEX i among
//     its kept entries (a per-pixel LDS counter, bumped in log order): copy i & 3, sub-pass (i - first index of the
//     pixel in this row) >> 2 - four entries of a pixel per sub-pass (2.7 sub-passes per 64-entry row on the bench: a
//     row's entries crowd on a dozen pixels; one sub-pass per group of four FACES present in the row was 3.5, and 6 %
//     of the kernel slower).  (Three LDS float atomics per entry, colliding on the pixel, cost 0.28 of 2.39 ms in
//     round 2; ranking the lanes of a row by pixel with six ballots was slower still.)  When the log
//     fills up (thousands of candidates per pixel: far cameras, dense meshes) the same machinery keeps each overflowing
//     pixel's K nearest and compacts the log in place; pruning bounds for front-to-back-sorted (dense) objects.
//   * TWELVE RESIDENT WAVES PER CU: three per SIMD is what <= 168 VGPRs allow (the evaluation alone holds 132-147, DESIGN
//     section 5), and twelve need <= 12.5 KB of LDS each (12 752 B still fits, 13 008 B does not; 12 240 B here): one
//     accumulator slot per pixel with a rotated column, 16-bit key bounds, the per-pixel bound of dense objects read from
//     its owner lane.  10 / 11 / 12 waves per CU = 2.66 / 2.52 / 2.42 ms (round 2).
//   * ONE WAVE PER WORKGROUP: cross-lane hand-over through LDS needs no __syncthreads() (occ_common.hpp: wave_lds_sync).
//   * SMALL LAUNCHES SPLIT THEIR TILES into 1 << split_log2 row groups (RasterParams): same staging, same batches, same
//     accumulator copies - bit-identical results - so that one env's render is not bounded by one wave per tile.
//
// Semantics: eval_face of occ_eval.hpp, the candidate rule, the K-nearest-by-z truncation, the
// clipped-pair rule and the hard nearest-face rule of [P3D].

// OCC_LOG_CAP (include/occlusionenv_amd.h): log entries per wave; must hold a compacted log (64 * OCC_MAX_K) plus the
// pairs of one batch
static_assert(OCC_LOG_CAP >= 64 * OCC_MAX_K + 2048 + 64, "OCC_LOG_CAP too small");
static_assert(OCC_LOG_ENTRY_BYTES == 30, "12 B payload + 8 B (key, tag) + 8 B (key, tag) + 2 B (log index) of the selection's own compacted copy");
static_assert(OCC_LOG_CAP % 8 == 0, "the regions of a wave's log stay 16-byte aligned");
#define OCC_LOG_BYTES ((size_t)OCC_LOG_CAP * OCC_LOG_ENTRY_BYTES)

constexpr int kT2 = 8;         // tile side in pixels (== OCC_TILE)
constexpr int kStg2 = 32;      // faces staged per batch (LDS budget: 12 resident waves per CU need <= 12.5 KB each)
constexpr int kStgPad = 33;    // LDS stride (float4) between the parts of the staged records (odd: conflict-free staging)
constexpr int kPairCap = 2048; // pairs per batch at most (kStg2 faces x 64 pixels): one byte each in the pair map
constexpr int kSelBits = 5;    // radix-select digit: 32 u16 buckets = 16 dwords per pixel (4 KB, aliasing records + descriptors)
constexpr int kSelDw = (1 << kSelBits) / 2;
constexpr int kSelStride = kSelDw + 1;  // dwords per pixel in LDS: odd, so that lanes bumping the SAME bucket of different
                                        // pixels (a face's pixels lie at one depth) fall on different banks; with the
                                        // stride of 16 they shared two banks (SQ_LDS_BANK_CONFLICT / ACTIVE_INST_LDS 0.6-0.7)
#ifndef OCC_SWEEP_U
#define OCC_SWEEP_U 4
#endif
constexpr int kSweepU = OCC_SWEEP_U;     // 64-entry rows of the log per sweep group; two groups are in flight (wider groups /
                               // 16-byte double rows were measured slower: they push the kernel into spilling)
#ifndef OCC_SWEEP_RING
#define OCC_SWEEP_RING 2
#endif
// Round 5: the final sweep issues the payload loads of a group's kept entries right after the group's decisions and applies
// them one group later (1.947 -> 1.899 ms).  The sweeps keep kRingGroups groups of kSweepU rows of (key, tag) in flight;
// more than two buys nothing (three: 1.902 ms, four: 1.939 ms - the registers cost more than the latency hidden), and rows
// are still handled kSweepU at a time: the window look-ups of a group are independent LDS reads - one row at a time,
// twelve rows deep in flight, was measured 9 % SLOWER (2.13 - 2.19 ms): the LDS round trips then lie end to end.
constexpr int kRingGroups = OCC_SWEEP_RING;  // groups of kSweepU rows of (key, tag) in flight per sweep
constexpr int kListCap = 8;    // boundary-bucket entries per pixel that the owner lane resolves itself
#ifndef OCC_ACC_COPY_BITS
#define OCC_ACC_COPY_BITS 2
#endif
constexpr int kCopyBits = OCC_ACC_COPY_BITS;  // accumulator copies = 1 << kCopyBits, chosen by (face sequence number & (kCopies - 1))
constexpr int kCopies = 1 << kCopyBits;
// (Round 5, measured: SIX copies of 12 bytes - product and tangent sums in arrays of their own, candidate count and key bound
// as one word per pixel bumped by non-returning LDS atomics, the same 12.7 KB - bring a round from 2.21 to 1.54 sub-passes and
// the kernel from 1.896 to 1.867 ms against FOUR copies in that format; this format, with its single 16-byte access per
// sub-pass and the count riding along, runs at 1.869: nothing gained, not kept.  The lanes of a round crowd on about ten
// pixels - seven tiny faces covering the same ones - so even an exact conflict schedule needs 1.9 sub-passes with four copies.
// Final sweep: sub-passes over a whole group of four rows at once (four reads in flight per sub-pass): 1.930 vs 1.887, worse.)
constexpr int kAccStride = 65; // accumulator slots per copy: one per pixel, +1 so that the copies of a pixel differ mod 16 (acc_slot)

constexpr uint32_t kNoEntry = 0xFFFFFFFFu;  // tag of a slot past the end of the log (a tag is pixel | face sequence number << 6)
struct LogPay {  // 12 bytes, moved with one dwordx3 access
    float q, ge, ga;  // 1 - p, g_el, g_az
};
struct WaveLog {
    LogPay* __restrict__ pay;
    uint2* __restrict__ kt;    // (order-preserving depth key, pixel of the tile 0..63 | face sequence number << 6)
    // Round 5: what the selection's LATER sweeps read.  Its first sweep (the first histogram pass) copies the entries of
    // the pixels that hold more than K candidates - 63 % of the log of a tile that needs selection on the bench - into
    // this region, in log order and unchanged, with every entry's index in the log (where its payload is) beside it in
    // ix2.  The other histogram passes and the final sweep then visit those rows only.
    // (A first version packed the index and twelve bits of the face sequence number into the tag: a dense object's
    // pruned log can be so sparse that a 64-entry row of the copy spans thousands of faces - test_sorted_scan_order_build
    // caught two pixels - and guarding the packing with a span check in the copying loop cost 8 % of the kernel.)
    uint2* __restrict__ kt2;
    uint16_t* __restrict__ ix2;
};
static_assert(OCC_LOG_CAP <= 65536, "the compacted copy keeps the log index in sixteen bits");

// Accumulator slot of pixel pix (= 8 py + px) in copy cpy.  A 16-byte LDS access is served 16 lanes at a time, one per
// residue of the slot index mod 16: the column is rotated by 3 every second row and the copy stride is 1 mod 16, so that
// the rows of a face's footprint and the four copies of one pixel (four faces evaluated side by side) land on
// different residues.
__device__ __forceinline__ int acc_slot(int cpy, int pix) {
    const int r2 = pix >> 4;  // the column rotates by 3 every second row (shifts and adds: no 64-bit mad)
    return (int)__umul24((uint32_t)cpy, (uint32_t)kAccStride) + (pix & 56) + ((pix + r2 + (r2 << 1)) & 7);
}
// Largest stored key of a (copy, pixel) as an upper bound in 16 bits: the key's high half + 1.  Only ever used as a bound
// from above (a looser bound prunes a little less, never wrongly); 0 = nothing stored.
__device__ __forceinline__ uint16_t akm_enc(uint32_t key) { return (uint16_t)min(0xFFFFu, (key >> 16) + 1u); }
__device__ __forceinline__ uint32_t akm_dec(uint32_t v) { return v >= 0xFFFFu ? 0xFFFFFFFFu : v << 16; }

__device__ __forceinline__ float unzkey(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__device__ __forceinline__ int lane_rank(unsigned long long m) {  // set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

#ifdef OCC_DBG_TIME  // diagnostic build only: where the wave-time of the raster kernel goes
// [0..15] shader cycles per phase, summed over waves.  In s_memrealtime ticks (100 MHz): [16] earliest wave start,
// [17] latest wave end, [18] sum of wave ends, [19] waves, [20] sum / [21] count / [22] max of the item times with an
// overflowing pixel, [23] sum / [24] count of the others, [25] sum / [26] max of (last item end -> exit), [27] waves
// that started > 50 us after the first, [32 + b] waves ending in the b-th 100 us after the first start, [64 + b] sum
// of their last item's time, [96 + x] earliest wave start on XCD x, [104 + x] latest wave end there.
// The end time is taken BEFORE the flush (3072 waves x ~30 atomics on a few lines take ~0.5 ms by themselves), and the
// flush of the early finishers still slows the queue atomics of the others: tails read from this build are upper bounds.
__device__ unsigned long long g_dbg_time[112];
#define OCC_T_DECL unsigned long long t_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long t_last = __builtin_amdgcn_s_memtime(); \
    const unsigned long long t_w0 = __builtin_amdgcn_s_memrealtime(); unsigned long long t_item = t_w0, t_os = 0, t_on = 0, t_om = 0, t_ns = 0, t_nn = 0, t_lastd = 0
#define OCC_T_ITEM(heavy) do { const unsigned long long t_n = __builtin_amdgcn_s_memrealtime(), d_ = t_n - t_item; t_item = t_n; t_lastd = d_; \
        if (heavy) { t_os += d_; t_on += 1; t_om = d_ > t_om ? d_ : t_om; } else { t_ns += d_; t_nn += 1; } } while (0)
#define OCC_T(i) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_acc[i] += t_now - t_last; t_last = t_now; } while (0)
#define OCC_T_FLUSH do { if (lane == 0) { const unsigned long long t_e = __builtin_amdgcn_s_memrealtime(); \
        for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_dbg_time[i_], t_acc[i_]); \
        atomicMin(&g_dbg_time[16], t_w0); atomicMax(&g_dbg_time[17], t_e); atomicAdd(&g_dbg_time[18], t_e); atomicAdd(&g_dbg_time[19], 1ull); \
        atomicAdd(&g_dbg_time[20], t_os); atomicAdd(&g_dbg_time[21], t_on); atomicMax(&g_dbg_time[22], t_om); \
        atomicAdd(&g_dbg_time[23], t_ns); atomicAdd(&g_dbg_time[24], t_nn); atomicAdd(&g_dbg_time[25], t_e - t_item); atomicMax(&g_dbg_time[26], t_e - t_item); \
        const unsigned long long first_ = *(volatile unsigned long long*)&g_dbg_time[16], rel_ = t_e - first_; \
        const int b_ = (int)(rel_ / 10000ull > 31ull ? 31ull : rel_ / 10000ull); \
        atomicAdd(&g_dbg_time[32 + b_], 1ull); atomicAdd(&g_dbg_time[64 + b_], t_lastd); \
        if (t_w0 - first_ > 5000ull) atomicAdd(&g_dbg_time[27], 1ull); \
        atomicMin(&g_dbg_time[96 + my_xcc], t_w0); atomicMax(&g_dbg_time[104 + my_xcc], t_e); } } while (0)
#elif defined(OCC_DBG_ENDS)  // light diagnostic build: per wave (start, end of its last item, exit, items), plain stores - no atomics
__device__ unsigned long long g_dbg_ends[4 * 4096];
#define OCC_T_DECL const unsigned long long t_w0 = __builtin_amdgcn_s_memrealtime(); unsigned long long t_item = t_w0, t_items = 0
#define OCC_T(i) do { } while (0)
#define OCC_T_ITEM(heavy) do { t_item = __builtin_amdgcn_s_memrealtime(); t_items += 1; } while (0)
#define OCC_T_FLUSH do { if (lane == 0 && blockIdx.x < 4096) { g_dbg_ends[4 * blockIdx.x] = t_w0; g_dbg_ends[4 * blockIdx.x + 1] = t_item; \
        g_dbg_ends[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime(); g_dbg_ends[4 * blockIdx.x + 3] = t_items | ((unsigned long long)my_xcc << 32); } } while (0)
#else
#define OCC_T_DECL do { } while (0)
#define OCC_T(i) do { } while (0)
#define OCC_T_ITEM(heavy) do { } while (0)
#define OCC_T_FLUSH do { } while (0)
#endif

#ifndef OCC_RASTER2_WAVES_PER_SIMD
#define OCC_RASTER2_WAVES_PER_SIMD 3
#endif

// (Launch parameters behind a constant-address-space pointer instead of by value - so that they are s_load-ed where
// they are used rather than kept or spilled - were measured in round 4: SGPR spills 83 -> 58, kernel 1.949 -> 1.941 ms,
// inside the noise; none of the spill code sits in the round loop.  Not kept.)
template <bool SOFT, bool HARD, bool GRAD>
__global__ __launch_bounds__(64, OCC_RASTER2_WAVES_PER_SIMD) void occ_raster2_kernel(RasterParams P) {
    const int lane = threadIdx.x;
    const int S = P.sc.img;
    const float fS = (float)S;
    const int cap = P.sc.rec_cap;
    const int K = P.K;
    constexpr int kParts = GRAD ? kRecParts : (SOFT ? 5 : 3);  // float4 parts of a record that this variant reads (never part 3)

    // staged records (part-major) + pair descriptors; both idle during a selection, whose histograms alias them
    constexpr int kRecF4 = kRecParts * kStgPad;
    static_assert(kPairCap == kStg2 * 64 && kPairCap == 2048, "pair map: two 16-byte stores per lane clear it");
    __shared__ float4 s_pool[kRecF4 + kPairCap / 16];
    float4* const s_rec = s_pool;
    uint8_t* const s_flag = reinterpret_cast<uint8_t*>(s_pool + kRecF4);  // pair map: 1 = first pair of a face
    // selection scratch on top of the same pool: 64 x kSelStride histogram dwords, then five 64-entry arrays
    static_assert(sizeof(float4) * (kRecF4 + kPairCap / 16) >= 64 * kSelStride * 4 + 64 * 5 * 4, "selection scratch does not fit");
    static_assert((64 * kSelStride) % 4 == 0 && 64 * kSelStride * 4 >= 64 * kListCap * 8, "alignment of the arrays behind the histograms / boundary lists on top of them");
    uint32_t* const s_selbase = reinterpret_cast<uint32_t*>(s_pool) + 64 * kSelStride;
    uint2* const s_sel = reinterpret_cast<uint2*>(s_selbase);            // selection window (low key, shift | 255 = idle)
    uint32_t* const s_take = s_selbase + 128;
    uint32_t* const s_lcnt = s_selbase + 192;   // boundary-list fill counts (final selection) ...
    uint32_t* const s_kmax2 = s_selbase + 192;  // ... or the kept entries' largest key (in-loop compaction): never both
    uint32_t* const s_arrive = s_selbase + 256; // kept entries per pixel so far (final sweep: which copy takes the next one)
    // two hit lists: while one batch is evaluated the next one is already scanned and its records are in flight
    __shared__ int s_hit[2 * kStg2];               // record index of every staged face
    __shared__ uint2 s_box[2 * kStg2];             // its pixel bbox (xl | yl << 16, xh | yh << 16), then (pre, geometry)
    __shared__ float4 s_acc[kCopies * kAccStride]; // (prod (1 - p_k), sum g_el, sum g_az, count) per copy and pixel
    __shared__ uint16_t s_akm[kCopies * kAccStride];  // bound on the largest stored key per copy and pixel (akm_enc)
    __shared__ unsigned long long s_hard[64];

    WaveLog lg;
    {
        char* base = reinterpret_cast<char*>(P.ws.lists) + (size_t)blockIdx.x * OCC_LOG_BYTES;
        lg.pay = reinterpret_cast<LogPay*>(base);
        lg.kt = reinterpret_cast<uint2*>(base + (size_t)OCC_LOG_CAP * 12);
        lg.kt2 = reinterpret_cast<uint2*>(base + (size_t)OCC_LOG_CAP * 20);
        lg.ix2 = reinterpret_cast<uint16_t*>(base + (size_t)OCC_LOG_CAP * 28);
    }
    ciptr offs = as_const(P.ws.offsets);
    ciptr ord = as_const(reinterpret_cast<const int*>(P.ws.order));  // null: rect order through offs
    const uint2* __restrict__ ord_items =
        reinterpret_cast<const uint2*>(P.ws.order + (P.ws.order ? ord_items_word(P.sc.n_env, S) : 0));
    const int mq = xcd_slots(P.sc.n_env), MP = 8 * mq;
    const int my_xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;  // steers which queue is drained first only
    int qround = 0;
    const int sl = P.split_log2;  // a tile = 1 << sl work items of 8 >> sl pixel rows (RasterParams)
    const int myslot = acc_slot(0, lane);  // accumulator slot (copy 0) of the pixel this lane owns
    OCC_T_DECL;

    // (Round 5, -DOCC_DBG_ENDS: the waves' last items end within 2.7 % of the launch; then every wave walks the eight heads with
    // a device-scope RMW each, 35-40 us from its last item to its exit.  A line of "exhausted" flags - byte stores, one load -
    // brings that to 12 us and the launch not forward: the last items just end later.  Not kept.  The heads 128 bytes apart
    // instead of 64 - one L2 line each - is what did count: 1.894 -> 1.873 ms.)
    // (Claiming the NEXT item when an item starts, so that the queue head's atomic round trip flies during the item's work:
    // measured in round 4, 1.944 -> 1.995 ms.  A wave that holds two items at a time undoes the point of the cost-ordered
    // queues near the end of the launch.  Not kept.)
    for (;;) {
        OCC_T(9);  // previous item's result stores
        int item = -1, sub = 0;
        while (qround < 8) {
            const int qq = (my_xcc + qround) & 7;
            const int qbeg = (ord ? ord[qq] : offs[qq * mq]) << sl, qend = (ord ? ord[qq + 1] : offs[(qq + 1) * mq]) << sl;
            int t = qend;
            if (lane == 0 && qbeg < qend) t = qbeg + (int)atomicAdd(P.ws.queue + qq * kQueueStride, 1u);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t < qend) {
                item = t >> sl;
                sub = t & ((1 << sl) - 1);
                break;
            }
            qround += 1;
        }
        if (item < 0) break;
        OCC_T(0);  // dequeue
        int eo, local;
        // A tile that fewer than K + 1 faces touch cannot hold a pixel with more than K candidates: its candidates
        // need no log (nothing will ever be selected from it).  The item's cost class bounds the faces from above.
        bool nolog = false;
        if (ord) {  // cost order (occ_order_kernel): the item list names the tile
            const uint2 it = ord_items[item];
            eo = __builtin_amdgcn_readfirstlane((int)it.x);
            const int w = __builtin_amdgcn_readfirstlane((int)it.y);
            local = w & 0xFFFFFF;
            nolog = ord_class_bound(w >> 24) <= (uint32_t)K + 1u;
        } else {    // rect order: find the (env, object) whose item range holds this one
            int lo = 0, hi = MP;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (offs[mid] <= item) lo = mid; else hi = mid;
            }
            eo = perm_to_eo(lo, mq, P.sc.n_env);
            local = item - offs[lo];
        }
        if (eo < 0 || eo >= 3 * P.sc.n_env || local < 0) continue;
        ciptr rect = as_const(P.ws.objrect + eo * 4);  // in OCC_BLOCK (4-pixel) units; tiles are 2 x 2 blocks
        const int tx0 = rect[0] >> 1, ty0 = rect[1] >> 1, tw = (rect[2] >> 1) - tx0 + 1;
        const int x0t = (tx0 + local % tw) * kT2, y0t = (ty0 + local / tw) * kT2;
        if (!OCC_BOUND(!(x0t < 0 || y0t < 0 || x0t + kT2 > S || y0t + kT2 > S), 46, eo, local)) continue;  // never true for a sane rect
        const int xi = x0t + (lane & 7), yi = y0t + (lane >> 3);  // the pixel this lane OWNS (lane = pixel)
        // pixel rows of the tile that THIS item evaluates and writes (all eight unless the launch splits its tiles)
        const int row_lo = sub << (3 - sl), row_hi = row_lo + (8 >> sl) - 1;
        const bool my_rows = (lane >> 3) >= row_lo && (lane >> 3) <= row_hi;
        const int n = as_const(P.ws.nrec + eo)[0];
        const RecSpan span = rec_span(P.ws, cap, eo);
        if (!OCC_BOUND(!(n < 0 || n > span.cap), 47, n, eo)) continue;
        OCC_STAT(0, 1);  // work items
        const float4* __restrict__ recs4 = reinterpret_cast<const float4*>(P.ws.rec + span.base * OCC_REC_STRIDE);
        // scan order: face order, or front to back for an object that occ_sort_kernel re-sorted (rows in rec_bbox)
        const uint4* __restrict__ scan =
            reinterpret_cast<const uint4*>(scan_is_sorted(n) ? (const uint32_t*)P.ws.rec_bbox : (const uint32_t*)P.ws.scan) + span.base;

        wave_lds_sync();  // the previous item's readers of the LDS state are done
        // [P3D] pixel centre of the pixel this lane owns, in NDC, +X left, +Y up: same expression as the
        // oracle; a pair lane fetches the centre of ITS pixel from the owning lanes with two cross-lane reads
        const float own_xf = -1.0f + (2.0f * (float)(S - 1 - xi) + 1.0f) / fS;
        const float own_yf = -1.0f + (2.0f * (float)(S - 1 - yi) + 1.0f) / fS;
#pragma unroll
        for (int cpy = 0; cpy < kCopies; ++cpy) {
            s_acc[cpy * kAccStride + myslot] = make_float4(1.f, 0.f, 0.f, 0.f);
            s_akm[cpy * kAccStride + myslot] = 0;
        }
        s_hard[lane] = ~0ull;
        OCC_T(1);  // item decode + state init
        int nlog = 0;               // entries in the wave's log (wave-uniform)
        bool lim_on = false;        // this lane's pixel already holds >= K candidates
        uint32_t bnd = 0xFFFFFFFFu; // key bound of this lane's pixel (pair lanes read it across lanes)
        uint32_t thrB = 0xFFFFFFFFu;      // tile-wide skip key (wave-uniform)
        uint32_t kmin_tile = 0xFFFFFFFFu; // smallest depth key any candidate of this tile can have (chunk boxes)
        uint32_t kmx_lane = 0u;           // largest key this lane has appended to the log (selection window)
        // pruning bounds need every pixel's largest stored key: only kept for objects whose scan order is front to
        // back (occ_sort_kernel, >= kSortMin records) - in mesh order the bounds hardly ever bite
        const bool dense = n >= kSortMin;

        auto touches = [&](uint4 bb) {
            const int rx0 = bb.x & 0xFFFF, ry0 = bb.x >> 16, rx1 = bb.y & 0xFFFF, ry1 = (bb.y >> 16) & 0x0FFF;  // (corner-cut bits above)
            return (rx0 <= x0t + kT2 - 1) && (rx1 >= x0t) && (ry0 <= y0t + kT2 - 1) && (ry1 >= y0t);
        };
        // this lane's pixel: candidates held and their largest key (four copies folded)
        auto own_count = [&]() {
            float c = 0.f;  // small integers: any summation order is exact
#pragma unroll
            for (int cpy = 0; cpy < kCopies; ++cpy) c += s_acc[cpy * kAccStride + myslot].w;
            return (int)c;
        };
        auto own_kmax = [&]() {
            uint32_t m = 0u;
#pragma unroll
            for (int cpy = 0; cpy < kCopies; ++cpy) m = max(m, (uint32_t)s_akm[cpy * kAccStride + myslot]);
            return akm_dec(m);
        };
        // this lane's pixel: the copies folded in a FIXED order (pairwise: (0 * 1) * (2 * 3)) - reproducible results
        auto own_fold = [&]() {
            float4 a[kCopies];
#pragma unroll
            for (int cpy = 0; cpy < kCopies; ++cpy) a[cpy] = s_acc[cpy * kAccStride + myslot];
#pragma unroll
            for (int st = 1; st < kCopies; st <<= 1) {
#pragma unroll
                for (int cpy = 0; cpy + st < kCopies; cpy += 2 * st)
                    a[cpy] = make_float4(a[cpy].x * a[cpy + st].x, a[cpy].y + a[cpy + st].y, a[cpy].z + a[cpy + st].z, a[cpy].w + a[cpy + st].w);
            }
            return a[0];
        };

        int fseq_base = 0;  // faces staged before this batch: sequence number of a face in the tile = fseq_base + its slot
        // re-accumulated (prod (1 - p), sum g_el, sum g_az, count) of THIS lane's pixel when it went through selection
        float4 acc2 = make_float4(1.f, 0.f, 0.f, 0.f);
        // ---- exact top-K over the log for every pixel holding more than K entries ---------------------------
        // Leaves the sums of the lane's own pixel's K nearest in acc2 (and the largest kept key in s_kmax2 when compacting) and returns whether
        // this lane's pixel was one of them.  COMPACT also rewrites the log so that it holds exactly the entries still
        // accounted for, and folds the selected sums back into the accumulator copies.
        auto select_topk = [&](auto compact_c) __attribute__((always_inline)) -> bool {
            constexpr bool compact = decltype(compact_c)::value;
            // rows of (key, tag) in flight: the in-loop compaction runs with the next batch's records in registers
            constexpr int kRingG = compact ? 2 : kRingGroups;
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): every log entry written so far is in memory before it is swept
            wave_lds_sync();
            (void)OCC_BOUND(nlog >= 0 && nlog <= OCC_LOG_CAP, 44, nlog, compact);
            uint32_t* hist = reinterpret_cast<uint32_t*>(s_pool);  // 64 pixels x kSelStride dwords (32 u16 buckets + 1 pad)
            const int cnt = own_count();
            const bool ovf = cnt > K;
            // window start: a candidate's depth is a convex combination of its face's vertex depths, so no key lies
            // below the smallest chunk-box key of the tile - up to rounding, hence the margin of 4096 ulp
            uint32_t L = kmin_tile > 4096u ? kmin_tile - 4096u : 0u;
            int need = K, sh = 0;
            {
                const uint32_t kmx = wave_minmax_u32_dpp<true>(kmx_lane);  // largest key in the log: wave maximum of the lanes' own
                const uint32_t range = kmx >= L ? kmx - L : 0u;
                sh = range ? max(0, (32 - __builtin_clz(range)) - kSelBits) : 0;
            }
            // A sweep visits the log in groups of kSweepU rows of 64 entries; the next group's loads are issued before
            // the current one is processed (two groups = 8 KB in flight per wave: a sweep is pure memory latency).
            // src / nsrc: what is swept - the log, or (final selection, after its first pass) the compacted copy
            const uint2* __restrict__ src = lg.kt;
            int nsrc = nlog;
            bool fmt_b = false;  // the sweeps read the compacted copy (WaveLog.kt2) and its entry format
            auto load_group = [&](const int e0, auto& kt) __attribute__((always_inline)) {
                constexpr int U = (int)(sizeof(kt) / sizeof(kt[0]));
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = e0 + u * 64 + lane;
                    kt[u] = e < nsrc ? src[e] : make_uint2(0u, kNoEntry);
                }
            };
            constexpr int kGroup = 64 * kSweepU;
            // how the boundary bucket [L, L + 2^sh) of an overflowing pixel is resolved in the final sweep
            enum { kAll = 0, kTies = 1, kList = 2 };
            int mode = kAll;
            bool done = !ovf;
            s_sel[lane] = make_uint2(L, done ? 255u : (uint32_t)sh);
            OCC_T(10);  // selection: set-up
            OCC_STAT(9, nlog);                       // log entries a selection sweeps (per pass)
            OCC_STAT(10, __popcll(__ballot(ovf)));   // pixels that go through selection
            int pass = 0;
            while (__ballot(!done)) {
                OCC_STAT(8, 1);                      // histogram passes
#pragma unroll
                for (int i = 0; i < kSelStride; ++i) hist[lane + 64 * i] = 0u;  // the whole array, lane-contiguous
                wave_lds_sync();
                auto bump = [&](const uint2 (&kt)[kSweepU]) __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < kSweepU; ++u) {
                        if (kt[u].y != kNoEntry) {
                            const uint32_t px = kt[u].y & 63u;
                            const uint2 w = s_sel[px];
                            if (w.y < 32u) {
                                const uint32_t d = (kt[u].x - w.x) >> w.y;
                                if (kt[u].x >= w.x && d < (1u << kSelBits)) atomicAdd(&hist[px * kSelStride + (d >> 1)], 1u << (16 * (d & 1u)));
                            }
                        }
                    }
                };
                // FINAL selection, first pass: the entries of the pixels under selection are also copied, in log order, to the
                // compacted region that every later sweep reads (WaveLog.kt2)
                int n2 = 0;
                const unsigned long long ovf_m = __ballot(ovf);
                const uint32_t ovf_lo = (uint32_t)ovf_m, ovf_hi = (uint32_t)(ovf_m >> 32);
                const uint32_t L0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)L);  // (pass 0: L and sh are still the tile's, in every lane)
                const uint32_t sh0 = (uint32_t)__builtin_amdgcn_readfirstlane(sh);
                auto bump_copy = [&](const int e0, const uint2 (&kt)[kSweepU]) __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < kSweepU; ++u) {
                        // (first pass: every pixel under selection has the SAME window - the tile's - and which pixels those are is a
                        // 64-bit mask: no window look-up in LDS, one dependent round trip less per group of rows)
                        bool ov = false;
                        if (kt[u].y != kNoEntry) {
                            const uint32_t px = kt[u].y & 63u;
#ifdef OCC_EXP_NO_P1S  // A/B build: the window looked up in LDS as in the later passes
                            const uint2 w = s_sel[px];
                            ov = w.y < 32u;
                            if (ov) {
                                const uint32_t d = (kt[u].x - w.x) >> w.y;
                                if (kt[u].x >= w.x && d < (1u << kSelBits)) atomicAdd(&hist[px * kSelStride + (d >> 1)], 1u << (16 * (d & 1u)));
                            }
#else
                            ov = (((px & 32u) ? ovf_hi : ovf_lo) >> (px & 31u)) & 1u;
                            if (ov) {
                                const uint32_t d = (kt[u].x - L0) >> sh0;
                                if (kt[u].x >= L0 && d < (1u << kSelBits)) atomicAdd(&hist[px * kSelStride + (d >> 1)], 1u << (16 * (d & 1u)));
                            }
#endif
                        }
                        const unsigned long long m = __ballot(ov);
                        if (ov) {
                            const int wpos = n2 + lane_rank(m);
                            lg.kt2[wpos] = kt[u];
                            lg.ix2[wpos] = (uint16_t)(e0 + u * 64 + lane);
                        }
                        n2 += __popcll(m);
                    }
                };
                {
                    // kRingG groups of kSweepU rows in flight: the group that was just consumed is requested again at once
                    uint2 ring[kRingG][kSweepU];
#pragma unroll
                    for (int gi = 0; gi < kRingG; ++gi) load_group(gi * kGroup, ring[gi]);
                    for (int e0 = 0; e0 < nsrc; e0 += kRingG * kGroup) {
#pragma unroll
                        for (int gi = 0; gi < kRingG; ++gi) {
#ifdef OCC_EXP_NO_CLOG  // A/B build: no compacted copy, every sweep reads the log itself
                            bump(ring[gi]);
#else
                            if (!compact && pass == 0) bump_copy(e0 + gi * kGroup, ring[gi]);
                            else bump(ring[gi]);
#endif
                            load_group(e0 + (kRingG + gi) * kGroup, ring[gi]);
                        }
                    }
                }
#ifndef OCC_EXP_NO_CLOG
                if (!compact && pass == 0) {
                    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the copy is in memory before it is swept
                    src = lg.kt2;
                    nsrc = n2;
                    fmt_b = true;
                    OCC_STAT(13, n2);  // entries in the compacted copy
                }
#endif
                pass += 1;
                wave_lds_sync();
                OCC_T(11);  // selection: histogram sweeps
                if (!done) {
                    int cum = 0, bstar = (1 << kSelBits) - 1, mstar = 0, cumb = 0;
                    bool found = false;
#pragma unroll
                    for (int i = 0; i < kSelDw; ++i) {
                        const uint32_t w = hist[lane * kSelStride + i];
                        const int c0 = (int)(w & 0xFFFFu), c1 = (int)(w >> 16);
                        if (!found && cum + c0 >= need) { found = true; bstar = 2 * i; mstar = c0; cumb = cum; }
                        cum += c0;
                        if (!found && cum + c1 >= need) { found = true; bstar = 2 * i + 1; mstar = c1; cumb = cum; }
                        cum += c1;
                    }
                    need -= cumb;
                    L += (uint32_t)bstar << sh;
                    if (mstar == need || !found) {
                        done = true;
                        mode = kAll;  // the whole bucket is kept
                    } else if (sh == 0) {
                        done = true;
                        mode = kTies;  // one key value, more holders than places: first come (log order = scan order)
                    } else if (!compact && mstar <= kListCap) {
                        done = true;
                        mode = kList;  // a handful of entries: the owner lane sorts them out after the final sweep
                    } else {
                        sh = max(0, sh - kSelBits);
                    }
#ifdef OCC_EXP_HIST_ONCE  // timing experiment only (results void): what the histogram passes after the first cost
                    done = true;
                    mode = kAll;
#endif
                    s_sel[lane] = make_uint2(L, done ? 255u : (uint32_t)sh);
                }
                wave_lds_sync();
                OCC_T(12);  // selection: bucket scans
            }
            // final window of an overflowing pixel: keys < L are kept, of the bucket [L, L + 2^sh) `need` more
            s_sel[lane] = make_uint2(L, ovf ? (uint32_t)sh | ((uint32_t)mode << 8) : 255u);
            s_take[lane] = mode == kAll ? 0x7FFFFFFFu : (uint32_t)max(need, 0);
            s_lcnt[lane] = 0u;  // kList: entries collected so far / compaction: largest kept key
            s_arrive[lane] = 0u;
            acc2 = make_float4(1.f, 0.f, 0.f, 0.f);
            if (ovf) {  // its accumulated state is void: the copies now collect the kept entries (product, sums, count)
#pragma unroll
                for (int cpy = 0; cpy < kCopies; ++cpy) s_acc[cpy * kAccStride + myslot] = make_float4(1.f, 0.f, 0.f, 0.f);
            }
            uint2* blist = reinterpret_cast<uint2*>(s_pool);  // kList: (key, log index) x kListCap per pixel (histograms are done)
            wave_lds_sync();
            int wr = 0;
            auto settle = [&](const int e0, const auto& kt) __attribute__((always_inline)) {
                constexpr int kU = (int)(sizeof(kt) / sizeof(kt[0]));
                uint32_t keepm = 0u, readdm = 0u;  // per-row decisions of this lane, bit u
                LogPay pv[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {  // decisions in log order (ties are served first come)
                    if (kt[u].y != kNoEntry) {
                        const uint32_t px = kt[u].y & 63u;
                        const uint2 w = s_sel[px];
                        const uint32_t shp = w.y & 255u;
                        if (shp >= 32u) {
                            keepm |= 1u << u;  // pixel not overflowing: its entries stay, its sums are already right
                        } else {
                            bool r = false;
                            if (kt[u].x < w.x) {
                                r = true;
                            } else if (((kt[u].x - w.x) >> shp) == 0u) {
                                if ((w.y >> 8) == (uint32_t)kList) {
                                    const uint32_t sl = atomicAdd(&s_lcnt[px], 1u);
                                    if (sl < (uint32_t)kListCap) blist[px * kListCap + sl] = make_uint2(kt[u].x, (uint32_t)(e0 + u * 64 + lane));
                                } else {
                                    // atomicSub returns the old value: the first `take` arrivals are kept
                                    r = (int)atomicSub(&s_take[px], 1u) > 0;
                                }
                            }
                            if (r) {
                                readdm |= 1u << u;
                                keepm |= 1u << u;
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int e = e0 + u * 64 + lane;
#ifdef OCC_EXP_FS_NO_PAY  // timing experiment only (results void): what the final sweep's dependent payload loads cost
                    pv[u] = LogPay{__uint_as_float(0x3f800000u | ((readdm >> u) & 1u)), 0.f, (float)e};
#else
                    pv[u] = (((readdm | (compact ? keepm : 0u)) >> u) & 1u) ? lg.pay[e] : LogPay{1.f, 0.f, 0.f};
#endif
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    {
                        // Re-accumulate the kept entries - the product of the (1 - p_k) and the tangent sums, exactly as
                        // the evaluation rounds do - into the pixel's four accumulator copies by PLAIN read-modify-write:
                        // the tag carries the face's sequence number, faces 4 g .. 4 g + 3
                        // use copies 0..3, and one sub-pass applies the lanes of one group g, so no two lanes of an
                        // instruction share an address (three LDS float atomics per entry, colliding on the pixel,
                        // cost 0.28 of the launch's 2.39 ms).
                        const bool act = (readdm >> u) & 1u;
                        const uint32_t tag = kt[u].y;
                        const uint32_t grp = tag >> (6 + kCopyBits);
                        const int slot = acc_slot((int)((tag >> 6) & (uint32_t)(kCopies - 1)), (int)(tag & 63u));
                        unsigned long long rem = __ballot(act);
                        OCC_STAT(11, __popcll(rem));  // entries re-accumulated
#ifdef OCC_EXP_FS_NO_RMW  // timing experiment only (results void): what the final sweep's RMW sub-passes cost
                        asm volatile("" ::"v"(pv[u].q), "v"(pv[u].ge), "v"(pv[u].ga), "v"(slot), "v"(grp));
                        rem = 0ull;
#endif
                        while (rem) {
                            OCC_STAT(12, 1);          // RMW sub-passes of the final sweep
                            const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)grp, __ffsll(rem) - 1);
#ifdef OCC_EXP_ONE_SUBPASS
                            const bool mine = act;
#else
                            const bool mine = act && grp == g0;
#endif
                            if (mine) {
                                float4 a = s_acc[slot];
                                a.x *= pv[u].q;
                                if (GRAD) {
                                    a.y += pv[u].ge;
                                    a.z += pv[u].ga;
                                }
                                a.w += 1.0f;
                                s_acc[slot] = a;
                            }
                            rem &= ~__ballot(mine);
                        }
                        if (compact && act) atomicMax(&s_kmax2[tag & 63u], kt[u].x);  // largest kept key (the loop goes on)
                    }
                    if (compact) {
                        const bool kp = (keepm >> u) & 1u;
                        const unsigned long long m = __ballot(kp);
                        if (kp) {
                            const int w = wr + lane_rank(m);  // w <= e: never overtakes the reads of a later group
                            lg.kt[w] = kt[u];
                            lg.pay[w] = pv[u];
                        }
                        wr += __popcll(m);
                    }
                }
            };
            if constexpr (compact) {  // in-place rewrite: a group's loads must not run ahead of the previous group's stores
                uint2 ka[kSweepU];
                for (int e0 = 0; e0 < nlog; e0 += kGroup) {
                    load_group(e0, ka);
                    settle(e0, ka);
                }
            } else {
                // Group pipeline: decide(group g) - the window tests, the tie counters and boundary lists, in log order -
                // issues the payload loads of its kept entries at once; apply(group g - 1) re-accumulates a group whose
                // payloads have had a group's work to arrive.  Groups are applied in log order, sub-passes as in the evaluation
                // rounds: the sums come out bit for bit as before.
                uint2 ring[kRingG][kSweepU];
                uint32_t ringx[kRingG][kSweepU];  // the entries' indices in the log (their payloads' places)
                auto load_index = [&](const int e0, uint32_t (&ix)[kSweepU]) __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < kSweepU; ++u) {
                        const int e = e0 + u * 64 + lane;
                        ix[u] = fmt_b ? (e < nsrc ? (uint32_t)lg.ix2[e] : 0u) : (uint32_t)e;
                    }
                };
                uint32_t ptag[kSweepU];  // tag | 1 << 31 of an entry that is re-accumulated, else 0
                LogPay ppay[kSweepU];
#pragma unroll
                for (int gi = 0; gi < kRingG; ++gi) {
                    load_group(gi * kGroup, ring[gi]);
                    load_index(gi * kGroup, ringx[gi]);
                }
#pragma unroll
                for (int u = 0; u < kSweepU; ++u) {
                    ptag[u] = 0u;
                    ppay[u] = LogPay{1.f, 0.f, 0.f};
                }
                auto apply = [&]() __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < kSweepU; ++u) {
                        const bool act = (ptag[u] >> 31) != 0u;
                        const uint32_t tag = ptag[u] & 0x7FFFFFFFu;  // pixel | copy << 6 | sub-pass << 8 (decide)
                        const uint32_t grp = tag >> (6 + kCopyBits);
                        const int slot = acc_slot((int)((tag >> 6) & (uint32_t)(kCopies - 1)), (int)(tag & 63u));
                        unsigned long long rem = __ballot(act);
                        OCC_STAT(11, __popcll(rem));  // entries re-accumulated
#ifdef OCC_EXP_FS_NO_RMW  // timing experiment only (results void): what the final sweep's RMW sub-passes cost
                        asm volatile("" ::"v"(ppay[u].q), "v"(ppay[u].ge), "v"(ppay[u].ga), "v"(slot), "v"(grp));
                        rem = 0ull;
#endif
                        while (rem) {
                            OCC_STAT(12, 1);  // RMW sub-passes of the final sweep
                            const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)grp, __ffsll(rem) - 1);
#ifdef OCC_EXP_ONE_SUBPASS
                            const bool mine = act;
#else
                            const bool mine = act && grp == g0;
#endif
                            if (mine) {
                                float4 a = s_acc[slot];
                                a.x *= ppay[u].q;
                                if (GRAD) {
                                    a.y += ppay[u].ge;
                                    a.z += ppay[u].ga;
                                }
                                a.w += 1.0f;
                                s_acc[slot] = a;
                            }
                            rem &= ~__ballot(mine);
                        }
                    }
                };
                auto decide = [&](const uint2 (&kt)[kSweepU], const uint32_t (&ix)[kSweepU]) __attribute__((always_inline)) {
                    uint32_t readdm = 0u;
                    uint32_t rtag[kSweepU];  // pixel | copy << 6 | sub-pass << 8 of a kept entry
#pragma unroll
                    for (int u = 0; u < kSweepU; ++u) {  // decisions in log order (ties are served first come)
                        rtag[u] = 0u;
                        if (kt[u].y != kNoEntry) {
                            const uint32_t px = kt[u].y & 63u;
                            const uint2 w = s_sel[px];
                            const uint32_t shp = w.y & 255u;
                            if (shp < 32u) {  // (a pixel that does not overflow keeps its accumulated sums)
                                bool r = false;
                                if (kt[u].x < w.x) {
                                    r = true;
                                } else if (((kt[u].x - w.x) >> shp) == 0u) {
                                    if ((w.y >> 8) == (uint32_t)kList) {
                                        const uint32_t sl = atomicAdd(&s_lcnt[px], 1u);
                                        if (sl < (uint32_t)kListCap) blist[px * kListCap + sl] = make_uint2(kt[u].x, ix[u]);  // (key, log index)
                                    } else {
                                        r = (int)atomicSub(&s_take[px], 1u) > 0;  // the first `take` arrivals are kept
                                    }
                                }
                                if (r) {
                                    readdm |= 1u << u;
#ifndef OCC_EXP_FS_SEQ_COPIES
                                    // Which copy takes the entry and in which sub-pass: by the pixel's ARRIVAL INDEX i among its
                                    // kept entries (one counter per pixel; entries of one pixel in one row get consecutive
                                    // indices, in lane = log order - the rule the tie counters above already rest on).  Copy
                                    // i & 3; an entry waits for the sub-pass (i - index of the pixel's first entry in this row)
                                    // >> 2: four entries of a pixel go at once, and a row holds more than four only rarely -
                                    // against one sub-pass per group of four FACES present in the row (5.3 per row on the bench:
                                    // a row of kept entries spans dozens of faces).  The copy of an entry depends on nothing but
                                    // the pixel's own sequence of kept entries: results do not depend on how a launch splits its
                                    // tiles or on which other pixels are under selection.
                                    const uint32_t base = s_arrive[px];                 // (before any lane's add of this row)
                                    const uint32_t i = atomicAdd(&s_arrive[px], 1u);
                                    rtag[u] = px | ((i & 3u) << 6) | (((i - base) >> 2) << 8);
#else
                                    rtag[u] = kt[u].y;
#endif
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kSweepU; ++u) {
                        const bool r = (readdm >> u) & 1u;
                        const uint32_t e = ix[u];  // index of the entry in the log itself: where its payload is
                        ptag[u] = r ? (rtag[u] | 0x80000000u) : 0u;
#ifdef OCC_EXP_FS_NO_PAY  // timing experiment only (results void): what the final sweep's dependent payload loads cost
                        ppay[u] = LogPay{__uint_as_float(0x3f800000u | (r ? 1u : 0u)), 0.f, (float)e};
#else
                        ppay[u] = r ? lg.pay[e] : LogPay{1.f, 0.f, 0.f};
#endif
                    }
                };
                for (int e0 = 0; e0 < nsrc; e0 += kRingG * kGroup) {
#pragma unroll
                    for (int gi = 0; gi < kRingG; ++gi) {
                        apply();  // the group before this one
                        decide(ring[gi], ringx[gi]);
                        load_group(e0 + (kRingG + gi) * kGroup, ring[gi]);
                        load_index(e0 + (kRingG + gi) * kGroup, ringx[gi]);
                    }
                }
                apply();  // the last group
            }
            wave_lds_sync();
            if (ovf) acc2 = own_fold();
            OCC_T(13);  // selection: final sweep
            if (!compact && __ballot(mode == kList)) {
                // the owner lane picks the `need` nearest of its <= kListCap boundary entries by (key, log index) and
                // adds them to what the sweep accumulated
                const int nb = (mode == kList) ? min((int)s_lcnt[lane], kListCap) : 0;
                uint2 be[kListCap];
#pragma unroll
                for (int i = 0; i < kListCap; ++i) be[i] = i < nb ? blist[lane * kListCap + i] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                uint32_t takem = 0u;
#pragma unroll
                for (int i = 0; i < kListCap; ++i) {
                    int rank = 0;
#pragma unroll
                    for (int jj = 0; jj < kListCap; ++jj)
                        rank += (be[jj].x < be[i].x || (be[jj].x == be[i].x && be[jj].y < be[i].y)) ? 1 : 0;
                    if (i < nb && rank < need) takem |= 1u << i;
                }
                LogPay pv[kListCap];
#pragma unroll
                for (int i = 0; i < kListCap; ++i) pv[i] = ((takem >> i) & 1u) ? lg.pay[be[i].y] : LogPay{1.f, 0.f, 0.f};
                float sl = acc2.x, se = acc2.y, sa = acc2.z;
#pragma unroll
                for (int i = 0; i < kListCap; ++i) {
                    if ((takem >> i) & 1u) {
                        sl *= pv[i].q;
                        se += pv[i].ge;
                        sa += pv[i].ga;
                    }
                }
                acc2 = make_float4(sl, se, sa, acc2.w);
            }
            OCC_T(14);  // selection: boundary lists
            if (compact) {
                nlog = wr;
                if (ovf) {  // the selected sums become the pixel's accumulated state (copy 0; the others empty)
                    s_acc[myslot] = acc2;
                    s_akm[myslot] = akm_enc(s_kmax2[lane]);
#pragma unroll
                    for (int cpy = 1; cpy < kCopies; ++cpy) {
                        s_acc[cpy * kAccStride + myslot] = make_float4(1.f, 0.f, 0.f, 0.f);
                        s_akm[cpy * kAccStride + myslot] = 0;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the rewritten log is in place before it grows again
                wave_lds_sync();
            }
            return ovf;
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
                // (DPP reductions, not xor butterflies through the LDS crossbar: six dependent ds_bpermute round trips each)
                kmin_tile = min(kmin_tile, wave_minmax_u32_dpp<false>(hit ? cb.z : 0xFFFFFFFFu));
                cmask = __ballot(hit && cb.z < thrB);
            }
            const int bit = __builtin_ctzll(cmask);
            cmask &= cmask - 1;
            return cwin + bit;
        };
        // The rows of the next TWO candidate chunks are in flight while one is worked on: a batch of 32 hits usually
        // spans two or three chunk rows, and with one row ahead the second of them was waited for (~1 us) in every batch.
        auto load_row = [&](const int ch) -> uint4 {
            return (ch >= 0 && ch * 64 + lane < n) ? scan[ch * 64 + lane] : kEmptyBox;
        };
        int c = next_chunk();
        int cn = c >= 0 ? next_chunk() : -1, cn2 = -1;
        uint4 bb_cur = load_row(c), bb_nxt = load_row(cn), bb_nx2 = kEmptyBox;
        unsigned long long m = 0;  // hits of row c not yet staged
        bool opened = false;       // row c has been balloted (and the row after next requested)
        // next batch of the scan: up to kStg2 hits into hit list `boff` (0 or kStg2); returns how many
        auto fill = [&](const int boff) __attribute__((always_inline)) -> int {
            int cntf = 0;
            while (cntf < kStg2 && c >= 0) {
                if (!opened) {
                    cn2 = cn >= 0 ? next_chunk() : -1;
                    bb_nx2 = load_row(cn2);
                    m = __ballot(touches(bb_cur) && bb_cur.z < thrB);
                    opened = true;
                    OCC_STAT(5, 1);  // chunk rows scanned
                }
                if (m) {  // a row may hold more hits than the hit list has room for
                    const int room = kStg2 - cntf;
                    const int cnt = __popcll(m);
                    const int rank = lane_rank(m);
                    const bool mine = (m >> lane) & 1ull;
                    if (mine && rank < room) {
                        s_hit[boff + cntf + rank] = (int)bb_cur.w;
                        s_box[boff + cntf + rank] = make_uint2(bb_cur.x, bb_cur.y);
                    }
                    cntf += min(cnt, room);
                    m = cnt <= room ? 0ull : __ballot(mine && rank >= room);
                }
                if (!m) {
                    c = cn;
                    bb_cur = bb_nxt;
                    cn = cn2;
                    bb_nxt = bb_nx2;
                    opened = false;
                }
            }
            return cntf;
        };

        // ---- one staged batch: records -> LDS, pair expansion, evaluation rounds ------------------------------
        int nst = 0;  // staged faces (wave-uniform)
        // records of the staged faces -> LDS (part-major).  Lane i of a group of 8 fetches part i of one record: the 8
        // loads of a record are one 128-byte line.  kStg2 * 8 / 64 = 4 loads per lane, issued together.
        constexpr int kStageLoads = kStg2 * kRecParts / 64;
        auto stage_issue = [&](float4 (&r)[kStageLoads], const int boff, const int cntf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < kStageLoads; ++i) {
                const int idx = lane + 64 * i, k = idx >> 3, part = idx & 7;
                const bool on = idx < cntf * kRecParts && part < kParts && part != 3;
                const int hj = on ? s_hit[boff + k] : 0;
                r[i] = (on && OCC_BOUND(hj >= 0 && hj < n, 42, hj, n)) ? recs4[(size_t)hj * kRecParts + part] : make_float4(0, 0, 0, 0);
            }
        };
        auto stage_commit = [&](const float4 (&r)[kStageLoads], const int cntf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < kStageLoads; ++i) {
                const int idx = lane + 64 * i, k = idx >> 3, part = idx & 7;
                if (idx < cntf * kRecParts && part < kParts && part != 3) s_rec[part * kStgPad + k] = r[i];
            }
        };
        float4 rstage[kStageLoads];  // records of the NEXT batch in flight
        int boff = 0;                // hit list of the CURRENT batch
        auto stage_records = [&]() __attribute__((always_inline)) {  // (rare) put the current batch's records back
            float4 r[kStageLoads];
            stage_issue(r, boff, nst);
            stage_commit(r, nst);
        };
        // One batch: its records (requested a batch ago) go to LDS, pair counts, prefix sum, pair map; then the NEXT
        // batch is scanned and its record loads are issued, to fly during this batch's evaluation rounds.
        auto process_batch = [&]() __attribute__((always_inline)) -> int {
            OCC_T(2);  // scan (chunk boxes, rows, hit lists)
            wave_lds_sync();
            // pixels of this lane's face inside the tile: pair count, prefix sum over the staged faces
            int c = 0, cx0 = 0, cy0 = 0, cw = 1;
            // corner cut (finish_tri): which corners of the face's pixel box fall into this item's part of it and are
            // marked as beyond the blur disc.  The pairs of the box are numbered row by row; e0 drops number 0, thrA /
            // thrB are the numbers (counted with e0 already skipped) from which one / two more are skipped (127: never).
            uint32_t e0 = 0u, thrA = 127u, thrB2 = 127u;
            if (lane < nst) {
                const uint2 bb = s_box[boff + lane];
                const int bxl = (int)(bb.x & 0xFFFFu), byl = (int)(bb.x >> 16), bxh = (int)(bb.y & 0xFFFFu), byh = (int)((bb.y >> 16) & 0x0FFFu);
                cx0 = max(bxl, x0t);
                cy0 = max(byl, y0t + row_lo);
                const int cx1 = min(bxh, x0t + kT2 - 1), cy1 = min(byh, y0t + row_hi);
                cw = cx1 - cx0 + 1;
                const int ch = cy1 - cy0 + 1;
                c = max(cw, 0) * max(ch, 0);
                if (c > 0) {
                    const uint32_t cb = bb.y >> 28;
                    const bool left = cx0 == bxl, right = cx1 == bxh, top = cy0 == byl, bot = cy1 == byh;
                    e0 = ((cb & 1u) && left && top) ? 1u : 0u;
                    const uint32_t e1 = ((cb & 2u) && right && top && cw > 1) ? 1u : 0u;
                    const uint32_t e2 = ((cb & 4u) && left && bot && ch > 1) ? 1u : 0u;
                    const uint32_t e3 = ((cb & 8u) && right && bot && cw > 1 && ch > 1) ? 1u : 0u;
                    if (e1) thrA = (uint32_t)(cw - 1);
                    if (e2) thrB2 = (uint32_t)((ch - 1) * cw) - e1;
                    c -= (int)(e0 + e1 + e2 + e3);
                }
                cw = max(cw, 1);
            }
            // exclusive prefix sum of the pair counts (<= 64 each: 7 bits) from one ballot per bit - no LDS round trips
            int pre = 0, ptot = 0;
#pragma unroll
            for (int b = 0; b < 7; ++b) {
                const unsigned long long mb = __ballot((c >> b) & 1);
                pre += lane_rank(mb) << b;
                ptot += __popcll(mb) << b;
            }
            // Pair p of the batch belongs to the face with the largest pre <= p.  Instead of writing one descriptor per
            // pair, every face that HAS pairs marks its FIRST pair in a byte map; a round ballots its 64 flags and a
            // lane's face is the running count of such faces plus the marks at or below it.  (pre, clip origin, width,
            // staged slot, 2^15 / width) of the r-th face with pairs replace the bbox in s_box[r].  (A face always has a
            // pixel in the tile; it can be without pairs only in a row group of a split tile, RasterParams.split_log2.)
            auto mark_pairs = [&]() __attribute__((always_inline)) {
                const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
                reinterpret_cast<uint4*>(s_flag)[lane] = z4;
                reinterpret_cast<uint4*>(s_flag)[lane + 64] = z4;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                const bool has = lane < nst && c > 0;
                const int r = lane_rank(__ballot(has));
                if (has) {
                    const uint32_t inv15 = (32768u + (uint32_t)cw - 1u) / (uint32_t)cw;
                    // geometry word: pixel of the face's first pair (6 bits), 8 - width (3 bits, 0..7), staged slot, 2^15 / width
                    // .x: number of the face's first pair in the batch (minus e0: a lane's pair number minus this, mod 2^16, is
                    // its number in the face's box with the first corner skipped) | thrA << 16 | thrB << 24
                    s_box[boff + r] = make_uint2(((uint32_t)(pre - (int)e0) & 0xFFFFu) | (thrA << 16) | (thrB2 << 24),
                                                 (uint32_t)(cx0 - x0t) | ((uint32_t)(cy0 - y0t) << 3) | ((uint32_t)(8 - cw) << 6) |
                                                     ((uint32_t)lane << 9) | (inv15 << 16));
                    if (r > 0) s_flag[pre] = 1;
                }
            };
            OCC_STAT(1, 1);      // staged batches
            OCC_STAT(2, nst);    // staged records = (face, tile) pairs
            OCC_STAT(3, (ptot + 63) >> 6);  // evaluation rounds
            {
                OCC_T(3);  // pair counts
                mark_pairs();
                // the records of THIS batch were requested a whole batch ago: wait for them here, explicitly, and keep
                // the compiler from moving the next batch's loads above this wait (it would then wait for those too)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                stage_commit(rstage, nst);
                wave_lds_sync();
                __builtin_amdgcn_sched_barrier(0);
                OCC_T(4);  // pair map + record commit
            }
            // scan ahead: the next batch's hit list, its record loads fly while this batch is evaluated
            const int nst_next = fill(boff ^ kStg2);
            if (nst_next) stage_issue(rstage, boff ^ kStg2, nst_next);
            OCC_T(2);
            {
#ifndef OCC_DBG2_NO_COMPACT  // register-pressure experiment only (scripts/dbg)
                if (SOFT && nlog + ptot > OCC_LOG_CAP) {
                    // rare: the log could fill up inside this batch -> keep every overflowing pixel's K nearest,
                    // compact the log, go on with tighter bounds
                    select_topk(std::true_type{});
                    if (dense && own_count() >= K) {
                        lim_on = true;
                        bnd = min(bnd, own_kmax());
                    }
                    // the histograms lived on top of the records and the pair map: put both back
                    stage_records();
                    mark_pairs();
                    wave_lds_sync();
                    OCC_T(7);  // in-loop compaction
                }
#endif
                int fbase = 0;  // face of the pair just before this round
#ifdef OCC_DBG2_NO_EVAL  // timing experiment only
                if (ptot > 0) fbase = -1;
#endif
                // pair -> (staged slot, pixel, pixel centre) of one round.  (Issuing the decode of round r + 1 at the top of
                // round r, so that its three dependent LDS round trips overlap the arithmetic: +1.5 % - the loop is bound by
                // its VALU work, and carrying the decoded values costs five instructions; reading only the next round's mark bytes ahead: no change.)
                struct Dec {
                    bool live;
                    int f, pix, nlive;
                    float xf, yf;
                };
                auto decode = [&](const int p0) __attribute__((always_inline)) -> Dec {
                    Dec dc;
                    dc.nlive = min(64, ptot - p0);
                    const bool live = lane < dc.nlive;
                    // Lanes past the batch's last pair run along UNMASKED (no exec-mask branches around the LDS reads): the
                    // pair map is zero behind the last pair, so they decode to the last face with a pair index beyond its
                    // count - in-range LDS addresses, a meaningless pixel - and every side effect below is guarded by `live`.
                    const bool mark = s_flag[min(p0 + lane, kPairCap - 1)] != 0;
                    const unsigned long long mk = __ballot(mark);
                    const int fr = fbase + lane_rank(mk) + (mark ? 1 : 0);  // rank among the faces with pairs
                    fbase += __popcll(mk);
                    const uint2 fg = s_box[boff + fr];
                    dc.f = (int)((fg.y >> 9) & 31u);  // its staged slot
                    // number of the pair in the face's box: skipped corners (corner cut) are stepped over
                    const uint32_t j1 = ((uint32_t)(p0 + lane) - fg.x) & 0xFFFFu;
                    const uint32_t jj = j1 + (j1 >= ((fg.x >> 16) & 0xFFu) ? 1u : 0u) + (j1 >= (fg.x >> 24) ? 1u : 0u);
                    // row = jj / width, column = jj - row * width: pixel = first + 8 row + column = first + jj + row (8 - width)
                    // (exact for jj < 64, width <= 8; lanes past the last pair may hold more - masked below); 24-bit multiplies: full rate (v_mul_lo_u32 is quarter rate)
                    const uint32_t wq = mul24(jj, fg.y >> 16) >> 15;
                    const uint32_t d = ((fg.y & 63u) + jj + mul24(wq, (fg.y >> 6) & 7u)) & 63u;
                    dc.pix = (int)d;
                    dc.live = live;
                    dc.xf = __shfl(own_xf, (int)(d & 7u), 64);
                    dc.yf = __shfl(own_yf, (int)(d & 56u), 64);
                    return dc;
                };
                for (int p0 = 0; p0 < ptot && fbase >= 0; p0 += 64) {
                    const Dec cur = decode(p0);
                    const bool live = cur.live;
                    const int f = cur.f, pix = cur.pix, nlive = cur.nlive;
                    const float xf = cur.xf, yf = cur.yf;
                    (void)OCC_BOUND(!live || (f >= 0 && f < nst && (unsigned)pix < 64u), 43, f, pix);
                    const int j = s_hit[boff + f];
                    const float4* rs = &s_rec[f];
                    Cand c1;
                    eval_face<SOFT, GRAD>(rs[0], rs[kStgPad], rs[2 * kStgPad],
                                          kParts > 4 ? rs[4 * kStgPad] : make_float4(0, 0, 0, 0), xf, yf, c1,
                                          [&](int v) { return rs[(5 + v) * kStgPad]; });
                    const int flags = live ? __float_as_int(rs[2 * kStgPad].z) : 0;
                    // Clipped quad split in two: only one half may enter a pixel's list.  Both halves'
                    // lanes look at both halves; the SECOND half's lane emits the winner when both are candidates,
                    // a half whose partner is no candidate at this pixel emits itself.  (A partner that the pruning
                    // bounds kept out of the batch has no lane: it lies beyond every pixel's K nearest anyway.)
#ifdef OCC_DBG2_NO_PAIR  // static instruction count of the main path only (scripts/dbg/loop_count.sh)
                    if (false) {
#else
                    if (__ballot(flags & (FLAG_PAIR_FIRST | FLAG_PAIR_SECOND))) {
#endif
                        const bool is_first = (flags & FLAG_PAIR_FIRST) != 0, is_second = (flags & FLAG_PAIR_SECOND) != 0;
                        if ((is_first && j + 1 < n) || (is_second && j >= 1)) {
                            const float4* r1 = recs4 + (size_t)(is_first ? j + 1 : j - 1) * kRecParts;
                            Cand cp;
                            eval_face<SOFT, GRAD>(OCC_REC_LOAD(r1, kParts), xf, yf, cp, [&](int v) { return r1[5 + v]; });
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
                        if (live && c1.inside)
                            atomicMin(&s_hard[pix], ((unsigned long long)zkey_pos(c1.zh) << 32) | (unsigned)j);
                    }
                    if (SOFT) {
                        const uint32_t key = zkey_pos(c1.z);  // only read for candidates (pz >= 0)
                        // dense objects: the pixel's bound sits in its owner lane (bnd; only changes between batches)
                        uint32_t pbnd = 0xFFFFFFFFu;
                        if (dense) pbnd = (uint32_t)__shfl((int)bnd, pix, 64);
                        const bool acc = live && c1.cand && (!dense || key < pbnd);
                        const unsigned long long m = __ballot(acc);
                        if (m) {
#ifndef OCC_DBG2_NO_LOG  // timing experiment only
                            if (acc && !nolog) {
                                // 32-bit byte offsets from the wave-uniform bases (e < OCC_LOG_CAP): no 64-bit mad per lane
                                const uint32_t e = (uint32_t)(nlog + lane_rank(m));
                                if (OCC_BOUND(e < (uint32_t)OCC_LOG_CAP, 41, e, nlog)) {
                                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(lg.kt) + (e << 3)) =
                                        make_uint2(key, (uint32_t)pix | (uint32_t)(fseq_base + f) << 6);
                                    // (the payload is read back for the kept entries only; writing it with the non-temporal hint:
                                    // 1.990 vs 1.992 ms, nothing - round 4)
#ifndef OCC_DBG2_NO_PAY  // timing experiment only: what the payload's 12 bytes per candidate cost (results are void)
                                    *reinterpret_cast<LogPay*>(reinterpret_cast<char*>(lg.pay) + __umul24(e, 12u)) = LogPay{c1.q, c1.ge, c1.ga};
#endif
                                }
                            }
#endif
                            if (!nolog) nlog += __popcll(m);
#ifndef OCC_DBG2_NO_ATOM  // timing experiment only
                            // accumulate: plain read-modify-write in sub-passes of four consecutive staged faces
                            const int f_first = __builtin_amdgcn_readfirstlane(f);
                            const int f_last = __builtin_amdgcn_readlane(f, nlive - 1);
                            // slot of (copy f & 3, pixel): the pixel's copy-0 slot is what its owner lane holds in myslot
                            // (one cross-lane read instead of redoing acc_slot's arithmetic), the copy adds a constant
                            const int slot = __shfl(myslot, pix, 64) + (int)__umul24((uint32_t)(f & (kCopies - 1)), (uint32_t)kAccStride);
                            const int grp = (f - f_first) >> kCopyBits;
#ifdef OCC_EXP_ONE_SUBPASS  // timing experiment only (results void): every round applied in ONE sub-pass - the ceiling for more copies
                            const int nsub = 1;
#else
                            const int nsub = ((f_last - f_first) >> kCopyBits) + 1;
#endif
#ifdef OCC_DBG_STATS  // what more accumulator copies would buy: sub-passes per round with 4 / 5 / 6 / 8 copies
                            {
                                const int nf_ = f_last - f_first + 1;
                                OCC_STAT(14, 1);  // rounds with at least one accepted pair
                                OCC_STAT(15, (nf_ + 3) / 4);
                                OCC_STAT(16, (nf_ + 4) / 5);
                                OCC_STAT(17, (nf_ + 5) / 6);
                                OCC_STAT(18, (nf_ + 7) / 8);
                                OCC_STAT(19, nf_);
                                // ... and if only the faces of ACCEPTED pairs counted (first to last accepted lane)
                                const int fa_ = __builtin_amdgcn_readlane(f, __ffsll(m) - 1), fb_ = __builtin_amdgcn_readlane(f, 63 - __builtin_clzll(m));
                                OCC_STAT(20, (fb_ - fa_ + 4) / 4);
                                OCC_STAT(21, (fb_ - fa_ + 6) / 6);
                                OCC_STAT(22, __popcll(m));  // accepted pairs
                            }
#endif
#ifdef OCC_DBG_STATS  // exact conflict schedules (what a per-round rank by LDS atomics would give)
                            {
                                __shared__ uint32_t s_dbgrk[64];
                                auto maxrank = [&](const bool on, const uint32_t k) {  // k < 256: counter (byte) index
                                    s_dbgrk[lane] = 0u;
                                    wave_lds_sync();
                                    const uint32_t old = atomicAdd(&s_dbgrk[k >> 2], on ? 1u << (8u * (k & 3u)) : 0u);
                                    wave_lds_sync();
                                    const uint32_t rk = on ? (old >> (8u * (k & 3u))) & 255u : 0u;
                                    return (int)wave_minmax_u32_dpp<true>(rk);
                                };
                                const uint32_t kc = (uint32_t)((f & 3) * 64 + pix), kp = (uint32_t)pix;
                                OCC_STAT(23, maxrank(live, kc) + 1);          // copy = face & 3, schedule by rank among LIVE lanes of (pixel, copy)
                                OCC_STAT(24, maxrank(acc, kc) + 1);           // ... among ACCEPTED lanes
                                OCC_STAT(25, maxrank(live, kp) / 4 + 1);      // copy = rank & 3 among live lanes of the pixel
                                OCC_STAT(26, maxrank(acc, kp) / 4 + 1);       // ... among accepted lanes (= arrival index)
                            }
#endif
                            auto subpasses = [&](auto with_bound) __attribute__((always_inline)) {
                                for (int sp = 0; sp < nsub; ++sp) {
#ifdef OCC_EXP_ONE_SUBPASS
                                    if (acc) {
#else
                                    if (acc && grp == sp) {
#endif
                                        float4 a = s_acc[slot];
                                        a.x *= c1.q;
                                        if (GRAD) {
                                            a.y += c1.ge;
                                            a.z += c1.ga;
                                        }
                                        a.w += 1.0f;
                                        s_acc[slot] = a;
                                        if (decltype(with_bound)::value) s_akm[slot] = max(s_akm[slot], akm_enc(key));
                                    }
                                }
                            };
                            // (two loops: with one, the bound's encoding and address are computed for every object)
                            if (dense) subpasses(std::true_type{});
                            else subpasses(std::false_type{});
                            if (acc) kmx_lane = max(kmx_lane, key);
#else
                            asm volatile("" ::"v"(c1.q), "v"(c1.ge), "v"(c1.ga));
#endif
                        }
                    }
                }
            }
            wave_lds_sync();
            OCC_T(5);  // evaluation rounds
            // Front-to-back pruning (SURVEY A.4 keeps the K smallest depths): once a pixel holds >= K candidates its
            // largest stored key bounds its K-th nearest from above, later candidates at or beyond it are dropped
            // unseen; once that holds for all 64 pixels (and every pixel has a hard face) whole faces / chunks whose
            // nearest vertex lies beyond every bound are skipped.
            // (In a soft launch an object in mesh order never gets per-pixel bounds - bnd stays at its maximum - so the
            // tile-wide skip key cannot move either: the wave-wide reduction is only run where it can.)
            if (!SOFT || dense) {
                uint32_t bound = 0xFFFFFFFFu;
                if (SOFT) {
                    if (!lim_on && own_count() >= K) {
                        lim_on = true;
                        bnd = min(bnd, own_kmax());
                    }
                    bound = bnd;
                }
                if (HARD) {
                    const uint32_t hk = (uint32_t)(s_hard[lane] >> 32);  // 0xFFFFFFFF while the pixel has no face
                    bound = SOFT ? max(bound, hk) : hk;
                }
                thrB = wave_minmax_u32_dpp<true>(bound);
            }
            OCC_T(6);  // pruning bounds
            fseq_base += nst;
            return nst_next;
        };

        nst = fill(0);
        if (nst) stage_issue(rstage, 0, nst);
        while (nst) {
            const int nst_next = process_batch();
            boff ^= kStg2;
            nst = nst_next;
        }


        OCC_T(2);

        // ---- per-pixel results (lane = pixel) ---------------------------------------------------------------------
        const size_t opix = ((size_t)eo * S + yi) * S + xi;
        (void)OCC_BOUND(opix < (size_t)P.sc.n_env * 3 * S * S && xi < S && yi < S, 45, eo, yi * S + xi);
        if (SOFT) {
            wave_lds_sync();
            const bool ovf = own_count() > K;
#ifdef OCC_DBG_STATS
            {
                const int cw_ = (int)wave_sum((float)own_count());
                const int co_ = (int)wave_sum(ovf ? (float)own_count() : 0.f);
                OCC_STAT(4, cw_);                    // candidates accounted for
                OCC_STAT(6, co_);                    // ... of which in pixels that need selection
                OCC_STAT(7, __ballot(ovf) ? 1 : 0);  // items with at least one such pixel
            }
#endif
            bool selected = false;
#ifndef OCC_DBG2_NO_SEL  // timing experiment only
            if (__ballot(ovf)) {
                if (nolog) {  // cannot happen (the class bound counts every face that can reach the tile): say so loudly
                    if (lane == 0) atomicOr(&P.ws.status[eo / 3], OCC_STATUS_LIST_OVERFLOW);
                } else {
                    selected = select_topk(std::false_type{});  // more than K candidates: keep the K nearest in z, A.4
                }
            }
#endif
            OCC_T(8);  // final selection
            OCC_T_ITEM(__ballot(ovf) != 0ull);
            float prod, sge, sga;
            if (selected) {
                prod = acc2.x;
                sge = acc2.y;
                sga = acc2.z;
            } else {
                const float4 a = own_fold();
                prod = a.x;
                sge = a.y;
                sga = a.z;
            }
            if (my_rows) {
                P.ws.obj_alpha[opix] = 1.0f - prod;
                if (GRAD) {
                    // d alpha/d theta = -(A/sigma) * sum_k p_k d(d_k)/d theta   
                    const float coef = -prod * kInvSigma;
                    reinterpret_cast<float2*>(P.ws.obj_grad)[opix] = make_float2(coef * sge, coef * sga);
                }
            }
        }
        if (HARD && my_rows) {
            const unsigned long long h = s_hard[lane];
            const bool any = h != ~0ull;
            P.ws.obj_hz[opix] = any ? unzkey((uint32_t)(h >> 32)) : 3.0e38f;
            P.ws.obj_hrec[opix] = any ? (int)(uint32_t)h : -1;
        }
    }
    OCC_T_FLUSH;
}
