// occ_oplevel.hpp -- operator-level naive rasteriser (K-buffer outputs in PyTorch3D layout) and its dists backward.
// Part of the single translation unit occ_kernels.hip (included inside namespace occ; not a stand-alone header).

// ------------------------------------------------------------------------------------------
// Operator-level replacement of PyTorch3D's _C.rasterize_meshes / _C.rasterize_meshes_backward
// (naive path, bin_size = 0): K-buffer outputs in PyTorch3D's layout.  The fused step() above never
// materialises these; this pair exists for callers of the rasteriser itself (SURVEY.md §8b lower surface)
// and is written for exactness, not speed: one thread per pixel, all faces of its mesh, replace-the-farthest
// K list kept directly in the output arrays, bubble sort at the end - the structure of upstream's naive
// CUDA kernel.  No FMA contraction / reciprocal shortcuts: the arithmetic order is the one of SURVEY A.4.
// ------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__device__ __forceinline__ float k_edge(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}
__device__ __forceinline__ float k_seg(float px, float py, float ax, float ay, float bx, float by) {
    const float bax = bx - ax, bay = by - ay;
    const float l2 = bax * bax + bay * bay;
    if (l2 <= kEpsilon) return (px - bx) * (px - bx) + (py - by) * (py - by);
    float t = (bax * (px - ax) + bay * (py - ay)) / l2;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float qx = ax + t * bax - px, qy = ay + t * bay - py;
    return qx * qx + qy * qy;
}

struct KbufArgs {
    const float* face_verts;      // (F,3,3)
    const int64_t* first_idx;     // (N)
    const int64_t* num_faces;     // (N)
    const int64_t* neighbor;      // (F) or null
    int N, H, W, K;
    float blur;
    int persp, clipb, cull;
    int64_t* p2f;  // (N,H,W,K)
    float* zbuf;   // (N,H,W,K)
    float* bary;   // (N,H,W,K,3)
    float* dists;  // (N,H,W,K)
};

__global__ __launch_bounds__(64) void occ_rast_naive_fwd_kernel(KbufArgs a) {
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long npix = (long)a.N * a.H * a.W;
    if (pix >= npix) return;
    const int n = (int)(pix / ((long)a.H * a.W));
    const int rem = (int)(pix - (long)n * a.H * a.W);
    const int yi = rem / a.W, xi = rem - yi * a.W;
    const float yf = -1.0f + (2.0f * (float)(a.H - 1 - yi) + 1.0f) / (float)a.H;
    const float xf = -1.0f + (2.0f * (float)(a.W - 1 - xi) + 1.0f) / (float)a.W;
    const float sqb = sqrtf(a.blur);
    const int K = a.K;
    int64_t* qf = a.p2f + pix * K;
    float* qz = a.zbuf + pix * K;
    float* qd = a.dists + pix * K;
    float* qb = a.bary + pix * K * 3;
    int qn = 0;
    const int64_t f0 = a.first_idx[n], f1 = f0 + a.num_faces[n];
    for (int64_t f = f0; f < f1; ++f) {
        const float* v = a.face_verts + f * 9;
        const float x0 = v[0], y0 = v[1], z0 = v[2], x1 = v[3], y1 = v[4], z1 = v[5], x2 = v[6], y2 = v[7], z2 = v[8];
        const float area = k_edge(x0, y0, x1, y1, x2, y2);
        if (a.cull && area < 0.0f) continue;
        if (area <= kEpsilon && area >= -kEpsilon) continue;
        if (fmaxf(fmaxf(z0, z1), z2) < 0.0f) continue;
        const float xmin = fminf(fminf(x0, x1), x2) - sqb, xmax = fmaxf(fmaxf(x0, x1), x2) + sqb;
        const float ymin = fminf(fminf(y0, y1), y2) - sqb, ymax = fmaxf(fmaxf(y0, y1), y2) + sqb;
        if (!((xmin <= xf && xf <= xmax) && (ymin <= yf && yf <= ymax))) continue;
        const float ar = k_edge(x2, y2, x0, y0, x1, y1) + kEpsilon;
        const float b0 = k_edge(xf, yf, x1, y1, x2, y2) / ar;
        const float b1 = k_edge(xf, yf, x2, y2, x0, y0) / ar;
        const float b2 = k_edge(xf, yf, x0, y0, x1, y1) / ar;
        float p0 = b0, p1 = b1, p2 = b2;
        if (a.persp) {
            const float w0 = b0 * z1 * z2, w1 = z0 * b1 * z2, w2 = z0 * z1 * b2;
            const float den = fmaxf(w0 + w1 + w2, kEpsilon);
            p0 = w0 / den; p1 = w1 / den; p2 = w2 / den;
        }
        float c0 = p0, c1 = p1, c2 = p2;
        if (a.clipb) {
            c0 = fmaxf(p0, 0.0f); c1 = fmaxf(p1, 0.0f); c2 = fmaxf(p2, 0.0f);
            const float sm = fmaxf(c0 + c1 + c2, kBaryClipMin);
            c0 /= sm; c1 /= sm; c2 /= sm;
        }
        const float pz = c0 * z0 + c1 * z1 + c2 * z2;
        if (pz < 0.0f) continue;
        const float e01 = k_seg(xf, yf, x0, y0, x1, y1), e02 = k_seg(xf, yf, x0, y0, x2, y2), e12 = k_seg(xf, yf, x1, y1, x2, y2);
        const float dist = fminf(fminf(e01, e02), e12);
        const int amin = (e01 <= e02 && e01 <= e12) ? 0 : ((e02 <= e01 && e02 <= e12) ? 1 : 2);
        const bool inside = p0 > 0.0f && p1 > 0.0f && p2 > 0.0f;
        if (!inside && dist >= a.blur) continue;
        const float sd = inside ? -dist : dist;
        // clipped-pair rule, incl. the shared-diagonal tie definition of DESIGN.md §2
        int itop = -1;
        const int64_t nb = a.neighbor ? a.neighbor[f] : -1;
        if (nb != -1) {
            for (int i = 0; i < qn; ++i)
                if (qf[i] == nb) { itop = i; break; }
        }
        int slot = -1;
        if (itop != -1) {
            // closest edge of the entry already in the list: recompute from its face
            const float* u = a.face_verts + nb * 9;
            const float g01 = k_seg(xf, yf, u[0], u[1], u[3], u[4]), g02 = k_seg(xf, yf, u[0], u[1], u[6], u[7]),
                        g12 = k_seg(xf, yf, u[3], u[4], u[6], u[7]);
            const int amin_nb = (g01 <= g02 && g01 <= g12) ? 0 : ((g02 <= g01 && g02 <= g12) ? 1 : 2);
            const bool shared_tie = (nb == f - 1 && amin_nb == 2 && amin == 0) || (nb == f + 1 && amin_nb == 0 && amin == 2);
            if (!shared_tie && dist < fabsf(qd[itop])) slot = itop;
        } else if (qn < K) {
            slot = qn++;
        } else {
            // full: the candidate displaces the largest (z, f) entry if it is smaller
            int im = 0;
            for (int i = 1; i < K; ++i)
                if (qz[i] > qz[im] || (qz[i] == qz[im] && qf[i] > qf[im])) im = i;
            if (pz < qz[im] || (pz == qz[im] && f < qf[im])) slot = im;
        }
        if (slot >= 0) {
            qf[slot] = f; qz[slot] = pz; qd[slot] = sd;
            qb[slot * 3] = c0; qb[slot * 3 + 1] = c1; qb[slot * 3 + 2] = c2;
        }
    }
    // ascending (z, f); empty slots = -1
    for (int i = 0; i < qn - 1; ++i)
        for (int j = 0; j < qn - 1 - i; ++j)
            if (qz[j] > qz[j + 1] || (qz[j] == qz[j + 1] && qf[j] > qf[j + 1])) {
                const int64_t tf = qf[j]; qf[j] = qf[j + 1]; qf[j + 1] = tf;
                float t = qz[j]; qz[j] = qz[j + 1]; qz[j + 1] = t;
                t = qd[j]; qd[j] = qd[j + 1]; qd[j + 1] = t;
#pragma unroll
                for (int c = 0; c < 3; ++c) { t = qb[j * 3 + c]; qb[j * 3 + c] = qb[(j + 1) * 3 + c]; qb[(j + 1) * 3 + c] = t; }
            }
    for (int i = qn; i < K; ++i) {
        qf[i] = -1; qz[i] = -1.0f; qd[i] = -1.0f;
        qb[i * 3] = qb[i * 3 + 1] = qb[i * 3 + 2] = -1.0f;
    }
}

// ------------------------------------------------------------------------------------------
// Tiled producer of the same K-buffers (occ_rasterize_meshes_tiled): one wave per (mesh, 8x8-pixel tile), lane = pixel.
// The per-(face, pixel) arithmetic and the list rules are those of occ_rast_naive_fwd_kernel, statement for statement
// (kbuf_eval below is that kernel's loop body): what changes is WHICH faces a pixel looks at and WHERE its list lives.
//   * Faces are taken 64 at a time, one per lane; a lane applies the per-face rejects (back face, degenerate, behind the
//     camera) and asks whether the face's bbox +- sqrt(blur) can hold ANY pixel centre of the tile.  A face that fails
//     fails the naive kernel's own bbox test at every pixel of the tile, so skipping it changes nothing; the survivors
//     are then visited in ascending face order by all 64 pixels - the order the naive kernel sees them in.
//   * A pixel's list keeps (depth, face) only, in LDS, entry i of pixel l at [i * 64 + l] (lane-contiguous: no bank
//     conflicts); signed distance and barycentrics are NOT carried along but re-derived from the face at the end -
//     the same expressions on the same inputs give the same bits.  The farthest entry is found by scanning the LDS list
//     (the naive kernel scans its list in the output arrays: two dependent global loads per entry and candidate).
//   * At the end every lane sorts its <= K entries by (depth, face) in LDS and writes its K output slots once.
// Bit-identical to the naive kernel on all four outputs (tests/test_gpu_rasterize_op.py), ~F / (faces per tile) times
// less evaluation work.  Needs 64 * K * 8 bytes of LDS: K <= 1024 (the C entry point falls back to the naive kernel above that).
// ------------------------------------------------------------------------------------------
struct KbufFace {
    float x0, y0, z0, x1, y1, z1, x2, y2, z2;
};
struct KbufHit {
    bool ok;      // the face is a candidate at this pixel
    bool inside;
    float pz, dist, c0, c1, c2;
    int amin;
};
// the naive kernel's per-(face, pixel) evaluation after the per-face rejects: bbox test ... blur test
__device__ __forceinline__ KbufHit kbuf_eval(const KbufFace& v, float xf, float yf, float sqb, float blur, int persp, int clipb) {
    KbufHit h;
    h.ok = false;
    h.inside = false;
    h.pz = h.dist = h.c0 = h.c1 = h.c2 = 0.f;
    h.amin = 0;
    const float xmin = fminf(fminf(v.x0, v.x1), v.x2) - sqb, xmax = fmaxf(fmaxf(v.x0, v.x1), v.x2) + sqb;
    const float ymin = fminf(fminf(v.y0, v.y1), v.y2) - sqb, ymax = fmaxf(fmaxf(v.y0, v.y1), v.y2) + sqb;
    if (!((xmin <= xf && xf <= xmax) && (ymin <= yf && yf <= ymax))) return h;
    const float ar = k_edge(v.x2, v.y2, v.x0, v.y0, v.x1, v.y1) + kEpsilon;
    const float b0 = k_edge(xf, yf, v.x1, v.y1, v.x2, v.y2) / ar;
    const float b1 = k_edge(xf, yf, v.x2, v.y2, v.x0, v.y0) / ar;
    const float b2 = k_edge(xf, yf, v.x0, v.y0, v.x1, v.y1) / ar;
    float p0 = b0, p1 = b1, p2 = b2;
    if (persp) {
        const float w0 = b0 * v.z1 * v.z2, w1 = v.z0 * b1 * v.z2, w2 = v.z0 * v.z1 * b2;
        const float den = fmaxf(w0 + w1 + w2, kEpsilon);
        p0 = w0 / den; p1 = w1 / den; p2 = w2 / den;
    }
    float c0 = p0, c1 = p1, c2 = p2;
    if (clipb) {
        c0 = fmaxf(p0, 0.0f); c1 = fmaxf(p1, 0.0f); c2 = fmaxf(p2, 0.0f);
        const float sm = fmaxf(c0 + c1 + c2, kBaryClipMin);
        c0 /= sm; c1 /= sm; c2 /= sm;
    }
    const float pz = c0 * v.z0 + c1 * v.z1 + c2 * v.z2;
    if (pz < 0.0f) return h;
    const float e01 = k_seg(xf, yf, v.x0, v.y0, v.x1, v.y1), e02 = k_seg(xf, yf, v.x0, v.y0, v.x2, v.y2),
                e12 = k_seg(xf, yf, v.x1, v.y1, v.x2, v.y2);
    const float dist = fminf(fminf(e01, e02), e12);
    h.amin = (e01 <= e02 && e01 <= e12) ? 0 : ((e02 <= e01 && e02 <= e12) ? 1 : 2);
    h.inside = p0 > 0.0f && p1 > 0.0f && p2 > 0.0f;
    if (!h.inside && dist >= blur) return h;
    h.ok = true;
    h.pz = pz; h.dist = dist; h.c0 = c0; h.c1 = c1; h.c2 = c2;
    return h;
}
__device__ __forceinline__ KbufFace kbuf_load(const float* __restrict__ face_verts, int64_t f) {
    const float* v = face_verts + f * 9;
    return KbufFace{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]};
}

// Does mesh n hold a clipped-face pair?  (PyTorch3D's clip_faces hands the rasteriser a neighbour array whenever clipping is
// enabled - all -1 unless a face straddled the clip plane - so "no array" is the rare case and "an array without pairs" the
// common one.)  Every wave of the two tiled kernels asks this for its mesh first: a coalesced pass over the mesh's slice of
// the array, a few microseconds; the kernel whose mode the mesh is not in returns at once.
__device__ __forceinline__ bool mesh_has_pairs(const KbufArgs& a, int n, int lane) {
    if (!a.neighbor) return false;
    const int64_t f0 = a.first_idx[n], f1 = f0 + a.num_faces[n];
    bool any = false;
    for (int64_t fc = f0; fc < f1; fc += 64) {
        const int64_t fl = fc + lane;
        any = any || (fl < f1 && a.neighbor[fl] != -1);
    }
    return __ballot(any) != 0ull;
}

// (ordered_only_with_pairs: the order-free kernel below was launched as well and takes the meshes without pairs)
__global__ __launch_bounds__(64) void occ_rast_tiled_fwd_kernel(KbufArgs a, int tiles_x, int tiles_y, int ordered_only_with_pairs) {
    extern __shared__ unsigned char kb_smem[];
    const int K = a.K, lane = threadIdx.x;
    float* sz = reinterpret_cast<float*>(kb_smem);            // [K][64] depths
    int* sf = reinterpret_cast<int*>(kb_smem) + 64 * K;       // [K][64] packed face indices
    const int tpm = tiles_x * tiles_y;
    const int n = blockIdx.x / tpm, t = blockIdx.x - n * tpm;
    if (ordered_only_with_pairs && !mesh_has_pairs(a, n, lane)) return;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int xi = tx * 8 + (lane & 7), yi = ty * 8 + (lane >> 3);
    const bool valid = xi < a.W && yi < a.H;
    const float yf = -1.0f + (2.0f * (float)(a.H - 1 - yi) + 1.0f) / (float)a.H;
    const float xf = -1.0f + (2.0f * (float)(a.W - 1 - xi) + 1.0f) / (float)a.W;
    // pixel centres of the tile: xf decreases with xi (NDC +X is left), yf with yi
    const int xi1 = min(tx * 8 + 7, a.W - 1), yi1 = min(ty * 8 + 7, a.H - 1);
    const float txmax = -1.0f + (2.0f * (float)(a.W - 1 - tx * 8) + 1.0f) / (float)a.W;
    const float txmin = -1.0f + (2.0f * (float)(a.W - 1 - xi1) + 1.0f) / (float)a.W;
    const float tymax = -1.0f + (2.0f * (float)(a.H - 1 - ty * 8) + 1.0f) / (float)a.H;
    const float tymin = -1.0f + (2.0f * (float)(a.H - 1 - yi1) + 1.0f) / (float)a.H;
    const float sqb = sqrtf(a.blur);
    int qn = 0;
    const int64_t f0 = a.first_idx[n], f1 = f0 + a.num_faces[n];
    for (int64_t fc = f0; fc < f1; fc += 64) {
        const int64_t fl = fc + lane;
        bool hit = false;
        if (fl < f1) {
            const KbufFace v = kbuf_load(a.face_verts, fl);
            const float area = k_edge(v.x0, v.y0, v.x1, v.y1, v.x2, v.y2);
            hit = !(a.cull && area < 0.0f) && !(area <= kEpsilon && area >= -kEpsilon) && !(fmaxf(fmaxf(v.z0, v.z1), v.z2) < 0.0f);
            const float xmin = fminf(fminf(v.x0, v.x1), v.x2) - sqb, xmax = fmaxf(fmaxf(v.x0, v.x1), v.x2) + sqb;
            const float ymin = fminf(fminf(v.y0, v.y1), v.y2) - sqb, ymax = fmaxf(fmaxf(v.y0, v.y1), v.y2) + sqb;
            // some centre of the tile inside [xmin, xmax] x [ymin, ymax]?  necessary: the ranges overlap
            hit = hit && xmin <= txmax && txmin <= xmax && ymin <= tymax && tymin <= ymax;
        }
        unsigned long long m = __ballot(hit);
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            const int64_t f = fc + b;
            const KbufFace v = kbuf_load(a.face_verts, f);  // wave-uniform address
            const KbufHit h = kbuf_eval(v, xf, yf, sqb, a.blur, a.persp, a.clipb);
            if (!(valid && h.ok)) continue;
            // clipped-pair rule, incl. the shared-diagonal tie definition of DESIGN.md §2
            int itop = -1;
            const int64_t nb = a.neighbor ? a.neighbor[f] : -1;
            if (nb != -1) {
                for (int i = 0; i < qn; ++i)
                    if ((int64_t)sf[i * 64 + lane] == nb) { itop = i; break; }
            }
            int slot = -1;
            if (itop != -1) {
                // closest edge and distance of the entry already in the list: recomputed from its face (what the naive
                // kernel reads back from its dists slot is this very minimum)
                const KbufFace u = kbuf_load(a.face_verts, nb);
                const float g01 = k_seg(xf, yf, u.x0, u.y0, u.x1, u.y1), g02 = k_seg(xf, yf, u.x0, u.y0, u.x2, u.y2),
                            g12 = k_seg(xf, yf, u.x1, u.y1, u.x2, u.y2);
                const int amin_nb = (g01 <= g02 && g01 <= g12) ? 0 : ((g02 <= g01 && g02 <= g12) ? 1 : 2);
                const float dist_nb = fminf(fminf(g01, g02), g12);
                const bool shared_tie = (nb == f - 1 && amin_nb == 2 && h.amin == 0) || (nb == f + 1 && amin_nb == 0 && h.amin == 2);
                if (!shared_tie && h.dist < dist_nb) slot = itop;
            } else if (qn < K) {
                slot = qn++;
            } else {
                // full: the candidate displaces the largest (z, f) entry if it is smaller
                int im = 0;
                float zm = sz[lane];
                int fm = sf[lane];
                for (int i = 1; i < K; ++i) {
                    const float zi = sz[i * 64 + lane];
                    const int fi = sf[i * 64 + lane];
                    if (zi > zm || (zi == zm && fi > fm)) { im = i; zm = zi; fm = fi; }
                }
                if (h.pz < zm || (h.pz == zm && f < (int64_t)fm)) slot = im;
            }
            if (slot >= 0) {
                sz[slot * 64 + lane] = h.pz;
                sf[slot * 64 + lane] = (int)f;
            }
        }
    }
    if (!valid) return;
    // ascending (z, f): insertion sort of this lane's column
    for (int i = 1; i < qn; ++i) {
        const float zi = sz[i * 64 + lane];
        const int fi = sf[i * 64 + lane];
        int j = i - 1;
        while (j >= 0) {
            const float zj = sz[j * 64 + lane];
            const int fj = sf[j * 64 + lane];
            if (!(zj > zi || (zj == zi && fj > fi))) break;
            sz[(j + 1) * 64 + lane] = zj;
            sf[(j + 1) * 64 + lane] = fj;
            --j;
        }
        sz[(j + 1) * 64 + lane] = zi;
        sf[(j + 1) * 64 + lane] = fi;
    }
    const long pix = ((long)n * a.H + yi) * a.W + xi;
    int64_t* qf = a.p2f + pix * K;
    float* qz = a.zbuf + pix * K;
    float* qd = a.dists + pix * K;
    float* qb = a.bary + pix * K * 3;
    for (int i = 0; i < qn; ++i) {
        const int f = sf[i * 64 + lane];
        const KbufHit h = kbuf_eval(kbuf_load(a.face_verts, f), xf, yf, sqb, a.blur, a.persp, a.clipb);
        qf[i] = f;
        qz[i] = sz[i * 64 + lane];
        qd[i] = h.inside ? -h.dist : h.dist;
        qb[i * 3] = h.c0; qb[i * 3 + 1] = h.c1; qb[i * 3 + 2] = h.c2;
    }
    for (int i = qn; i < K; ++i) {
        qf[i] = -1; qz[i] = -1.0f; qd[i] = -1.0f;
        qb[i * 3] = qb[i * 3 + 1] = qb[i * 3 + 2] = -1.0f;
    }
}

// ------------------------------------------------------------------------------------------
// Sub-pixel meshes (round 4): occ_rast_quad_fwd_kernel - one wave per (mesh, 4x4-pixel tile), FOUR faces in flight.
// Without clipped-face pairs (no neighbour array, or one that is -1 for every face of the mesh - mesh_has_pairs above: the
// case of every mesh the camera is not inside) the
// K-buffer of a pixel is simply the K smallest (depth, face) among its candidates - independent of the order the faces
// arrive in - so the arrival order may be given up for parallelism:
//   * lane = (pixel of the tile, quarter): the 16 pixels x 4 lanes each; the four lanes of a pixel take the tile's faces
//     in turn and keep a K-list each (same LDS as the 8x8 kernel's 64 lists).  A wave's evaluation pass covers four
//     faces, a face's bbox +- sqrt(blur) covers most of a 4x4 tile (it covered a third of an 8x8 one), and a tile's
//     margin ring holds 2.25x fewer faces: ~9x fewer passes per tile, four times as many tiles to fill the chip with.
//   * a face's nine floats come from the lane that loaded them in the tile test (cross-lane read) instead of a second,
//     dependent memory read per surviving face.
//   * the farthest entry of a full list is cached (depth, face, slot): a candidate that does not displace it costs one
//     compare instead of a scan of the list; the scan runs only after a replacement.
//   * at the end every lane sorts its list, finds the rank of each of its entries in the union of the pixel's four lists
//     (three monotone cursors into the siblings' sorted lists) and writes the entries ranked below K - distance and
//     barycentrics re-derived from the face, the same expressions on the same inputs as at arrival.
// Bit-identical to occ_rast_naive_fwd_kernel on all four outputs (tests/test_gpu_rasterize_op.py).  Meshes with pairs
// keep the 8x8 kernel above: the pair rule asks whether the sibling is in the list AT THE TIME the face arrives.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void occ_rast_quad_fwd_kernel(KbufArgs a, int tiles_x, int tiles_y) {
    extern __shared__ unsigned char kb_smem[];
    const int K = a.K, lane = threadIdx.x;
    float* sz = reinterpret_cast<float*>(kb_smem);       // [K][64] depths, entry i of lane l at i * 64 + l
    int* sf = reinterpret_cast<int*>(kb_smem) + 64 * K;  // [K][64] packed face indices
    const int tpm = tiles_x * tiles_y;
    const int n = blockIdx.x / tpm, t = blockIdx.x - n * tpm;
    if (mesh_has_pairs(a, n, lane)) return;  // the ordered 8x8 kernel takes this mesh
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int pl = lane & 15, sub = lane >> 4;  // pixel of the tile, quarter
    const int xi = tx * 4 + (pl & 3), yi = ty * 4 + (pl >> 2);
    const bool valid = xi < a.W && yi < a.H;
    const float yf = -1.0f + (2.0f * (float)(a.H - 1 - yi) + 1.0f) / (float)a.H;
    const float xf = -1.0f + (2.0f * (float)(a.W - 1 - xi) + 1.0f) / (float)a.W;
    const int xi1 = min(tx * 4 + 3, a.W - 1), yi1 = min(ty * 4 + 3, a.H - 1);
    const float txmax = -1.0f + (2.0f * (float)(a.W - 1 - tx * 4) + 1.0f) / (float)a.W;
    const float txmin = -1.0f + (2.0f * (float)(a.W - 1 - xi1) + 1.0f) / (float)a.W;
    const float tymax = -1.0f + (2.0f * (float)(a.H - 1 - ty * 4) + 1.0f) / (float)a.H;
    const float tymin = -1.0f + (2.0f * (float)(a.H - 1 - yi1) + 1.0f) / (float)a.H;
    const float sqb = sqrtf(a.blur);
    int qn = 0;
    float zm = 0.f;  // farthest entry of the list once it is full: (depth, face) and its slot
    int fm = 0, im = 0;
    const int64_t f0 = a.first_idx[n], f1 = f0 + a.num_faces[n];
    for (int64_t fc = f0; fc < f1; fc += 64) {
        const int64_t fl = fc + lane;
        bool hit = false;
        KbufFace v{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (fl < f1) {
            v = kbuf_load(a.face_verts, fl);
            const float area = k_edge(v.x0, v.y0, v.x1, v.y1, v.x2, v.y2);
            hit = !(a.cull && area < 0.0f) && !(area <= kEpsilon && area >= -kEpsilon) && !(fmaxf(fmaxf(v.z0, v.z1), v.z2) < 0.0f);
            const float xmin = fminf(fminf(v.x0, v.x1), v.x2) - sqb, xmax = fmaxf(fmaxf(v.x0, v.x1), v.x2) + sqb;
            const float ymin = fminf(fminf(v.y0, v.y1), v.y2) - sqb, ymax = fmaxf(fmaxf(v.y0, v.y1), v.y2) + sqb;
            hit = hit && xmin <= txmax && txmin <= xmax && ymin <= tymax && tymin <= ymax;
        }
        unsigned long long m = __ballot(hit);
        while (m) {  // four surviving faces per pass, one per quarter
            int bsel = -1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (m) {
                    const int b = __builtin_ctzll(m);
                    m &= m - 1;
                    if (sub == q) bsel = b;
                }
            }
            const int src = bsel < 0 ? lane : bsel;
            KbufFace u;
            u.x0 = __shfl(v.x0, src, 64); u.y0 = __shfl(v.y0, src, 64); u.z0 = __shfl(v.z0, src, 64);
            u.x1 = __shfl(v.x1, src, 64); u.y1 = __shfl(v.y1, src, 64); u.z1 = __shfl(v.z1, src, 64);
            u.x2 = __shfl(v.x2, src, 64); u.y2 = __shfl(v.y2, src, 64); u.z2 = __shfl(v.z2, src, 64);
            if (bsel < 0 || !valid) continue;
            const KbufHit h = kbuf_eval(u, xf, yf, sqb, a.blur, a.persp, a.clipb);
            if (!h.ok) continue;
            const int f = (int)(fc + bsel);
            int slot = -1;
            if (qn < K) {
                slot = qn++;
                if (qn == K) {  // the list has just filled up: find its farthest entry (this one included)
                    sz[slot * 64 + lane] = h.pz;
                    sf[slot * 64 + lane] = f;
                    slot = -1;
                    im = 0; zm = sz[lane]; fm = sf[lane];
                    for (int i = 1; i < K; ++i) {
                        const float zi = sz[i * 64 + lane];
                        const int fi = sf[i * 64 + lane];
                        if (zi > zm || (zi == zm && fi > fm)) { im = i; zm = zi; fm = fi; }
                    }
                }
            } else if (h.pz < zm || (h.pz == zm && f < fm)) {
                // full: the candidate displaces the farthest entry; the new farthest is found by one scan
                sz[im * 64 + lane] = h.pz;
                sf[im * 64 + lane] = f;
                im = 0; zm = sz[lane]; fm = sf[lane];
                for (int i = 1; i < K; ++i) {
                    const float zi = sz[i * 64 + lane];
                    const int fi = sf[i * 64 + lane];
                    if (zi > zm || (zi == zm && fi > fm)) { im = i; zm = zi; fm = fi; }
                }
            }
            if (slot >= 0) {
                sz[slot * 64 + lane] = h.pz;
                sf[slot * 64 + lane] = f;
            }
        }
    }
    // ascending (z, f): insertion sort of this lane's column
    for (int i = 1; i < qn; ++i) {
        const float zi = sz[i * 64 + lane];
        const int fi = sf[i * 64 + lane];
        int j = i - 1;
        while (j >= 0) {
            const float zj = sz[j * 64 + lane];
            const int fj = sf[j * 64 + lane];
            if (!(zj > zi || (zj == zi && fj > fi))) break;
            sz[(j + 1) * 64 + lane] = zj;
            sf[(j + 1) * 64 + lane] = fj;
            --j;
        }
        sz[(j + 1) * 64 + lane] = zi;
        sf[(j + 1) * 64 + lane] = fi;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // the pixel's four sorted lists -> one: rank of each own entry in the union (faces are distinct, so no ties)
    const int l1 = pl + 16 * ((sub + 1) & 3), l2 = pl + 16 * ((sub + 2) & 3), l3 = pl + 16 * ((sub + 3) & 3);
    const int n1 = __shfl(qn, l1, 64), n2 = __shfl(qn, l2, 64), n3 = __shfl(qn, l3, 64);
    if (!valid) return;
    const int total = min(K, qn + n1 + n2 + n3);
    const long pix = ((long)n * a.H + yi) * a.W + xi;
    int64_t* qf = a.p2f + pix * K;
    float* qz = a.zbuf + pix * K;
    float* qd = a.dists + pix * K;
    float* qb = a.bary + pix * K * 3;
    int p1 = 0, p2 = 0, p3 = 0;
    auto less_than = [&](int l, int i, float z, int f) {  // entry i of lane l < (z, f)
        const float zi = sz[i * 64 + l];
        return zi < z || (zi == z && sf[i * 64 + l] < f);
    };
    for (int i = 0; i < qn; ++i) {
        const float z = sz[i * 64 + lane];
        const int f = sf[i * 64 + lane];
        while (p1 < n1 && less_than(l1, p1, z, f)) ++p1;
        while (p2 < n2 && less_than(l2, p2, z, f)) ++p2;
        while (p3 < n3 && less_than(l3, p3, z, f)) ++p3;
        const int rank = i + p1 + p2 + p3;
        if (rank >= K) break;
        const KbufHit h = kbuf_eval(kbuf_load(a.face_verts, f), xf, yf, sqb, a.blur, a.persp, a.clipb);
        qf[rank] = f;
        qz[rank] = z;
        qd[rank] = h.inside ? -h.dist : h.dist;
        qb[rank * 3] = h.c0; qb[rank * 3 + 1] = h.c1; qb[rank * 3 + 2] = h.c2;
    }
    for (int i = total + sub; i < K; i += 4) {
        qf[i] = -1; qz[i] = -1.0f; qd[i] = -1.0f;
        qb[i * 3] = qb[i * 3 + 1] = qb[i * 3 + 2] = -1.0f;
    }
}

// dists part of RasterizeMeshesBackward: one thread per (pixel, k), atomicAdd into grad_face_verts
__global__ __launch_bounds__(256) void occ_rast_naive_bwd_kernel(const float* __restrict__ face_verts,
                                                                 const int64_t* __restrict__ p2f,
                                                                 const float* __restrict__ grad_dists, int N, int H, int W,
                                                                 int K, int persp, int clipb,
                                                                 float* __restrict__ grad_face_verts) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tot = (long)N * H * W * K;
    if (t >= tot) return;
    const int64_t f = p2f[t];
    if (f < 0) return;
    const long pix = t / K;
    const int rem = (int)(pix % ((long)H * W));
    const int yi = rem / W, xi = rem - yi * W;
    const float yf = -1.0f + (2.0f * (float)(H - 1 - yi) + 1.0f) / (float)H;
    const float xf = -1.0f + (2.0f * (float)(W - 1 - xi) + 1.0f) / (float)W;
    const float* v = face_verts + f * 9;
    const float x0 = v[0], y0 = v[1], z0 = v[2], x1 = v[3], y1 = v[4], z1 = v[5], x2 = v[6], y2 = v[7], z2 = v[8];
    const float ar = k_edge(x2, y2, x0, y0, x1, y1) + kEpsilon;
    float p0 = k_edge(xf, yf, x1, y1, x2, y2) / ar, p1 = k_edge(xf, yf, x2, y2, x0, y0) / ar, p2 = k_edge(xf, yf, x0, y0, x1, y1) / ar;
    if (persp) {
        const float w0 = p0 * z1 * z2, w1 = z0 * p1 * z2, w2 = z0 * z1 * p2;
        const float den = fmaxf(w0 + w1 + w2, kEpsilon);
        p0 = w0 / den; p1 = w1 / den; p2 = w2 / den;
    }
    if (clipb) { p0 = fmaxf(p0, 0.f); p1 = fmaxf(p1, 0.f); p2 = fmaxf(p2, 0.f); }  // the sign test below is all that matters
    const bool inside = p0 > 0.0f && p1 > 0.0f && p2 > 0.0f;
    const float g = (inside ? -1.0f : 1.0f) * grad_dists[t];
    const float e01 = k_seg(xf, yf, x0, y0, x1, y1), e02 = k_seg(xf, yf, x0, y0, x2, y2), e12 = k_seg(xf, yf, x1, y1, x2, y2);
    int ia, ib;
    if (e01 <= e02 && e01 <= e12) { ia = 0; ib = 1; }
    else if (e02 <= e01 && e02 <= e12) { ia = 0; ib = 2; }
    else if (e12 <= e01 && e12 <= e02) { ia = 1; ib = 2; }
    else return;
    const float ax = v[ia * 3], ay = v[ia * 3 + 1], bx = v[ib * 3], by = v[ib * 3 + 1];
    const float bax = bx - ax, bay = by - ay;
    float tt = (bax * (xf - ax) + bay * (yf - ay)) / (bax * bax + bay * bay + kEpsilon);
    tt = fminf(fmaxf(tt, 0.0f), 1.0f);
    const float dx = (1.0f - tt) * ax + tt * bx - xf, dy = (1.0f - tt) * ay + tt * by - yf;
    float* gf = grad_face_verts + f * 9;
    atomicAdd(gf + ia * 3, g * (1.0f - tt) * 2.0f * dx);
    atomicAdd(gf + ia * 3 + 1, g * (1.0f - tt) * 2.0f * dy);
    atomicAdd(gf + ib * 3, g * tt * 2.0f * dx);
    atomicAdd(gf + ib * 3 + 1, g * tt * 2.0f * dy);
}

// ------------------------------------------------------------------------------------------
// zbuf / bary part of RasterizeMeshesBackward.  Zero on the OcclusionEnv path (the silhouette shader reads
// dists only), built for callers of the rasteriser that shade with barycentrics or depth.  One thread per (pixel, k):
// the forward formulas of A.4 - area-normalised edge functions, perspective correction, lower-bound clip with
// renormalisation, depth - are re-evaluated in FORWARD mode, once per coordinate of the face (nine passes of a
// value + one-derivative pair), and each pass's directional derivative of (bary, zbuf) is contracted with the
// incoming gradients and added into grad_face_verts.  Kinks (max with a floor, clip at 0) take the derivative of the
// active branch, 0 on the floor - the convention of [P3D]'s Barycentric*Backward.
// ------------------------------------------------------------------------------------------
struct D1 {
    float v, d;
};
__device__ __forceinline__ D1 operator+(D1 a, D1 b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ D1 operator-(D1 a, D1 b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ D1 operator*(D1 a, D1 b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ D1 operator/(D1 a, D1 b) { const float q = a.v / b.v; return {q, (a.d - q * b.d) / b.v}; }
__device__ __forceinline__ D1 d1_const(float c) { return {c, 0.f}; }
__device__ __forceinline__ D1 d1_max(D1 a, float floor_) { return a.v > floor_ ? a : D1{floor_, 0.f}; }
__device__ __forceinline__ D1 d1_edge(D1 px, D1 py, D1 ax, D1 ay, D1 bx, D1 by) { return (px - ax) * (by - ay) - (py - ay) * (bx - ax); }

__global__ __launch_bounds__(256) void occ_rast_bwd_zbary_kernel(const float* __restrict__ face_verts,
                                                                 const int64_t* __restrict__ p2f,
                                                                 const float* __restrict__ grad_zbuf,
                                                                 const float* __restrict__ grad_bary, int N, int H, int W,
                                                                 int K, int persp, int clipb,
                                                                 float* __restrict__ grad_face_verts) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tot = (long)N * H * W * K;
    if (t >= tot) return;
    const int64_t f = p2f[t];
    if (f < 0) return;
    const float gz = grad_zbuf ? grad_zbuf[t] : 0.f;
    const float g0 = grad_bary ? grad_bary[t * 3] : 0.f, g1 = grad_bary ? grad_bary[t * 3 + 1] : 0.f,
                g2 = grad_bary ? grad_bary[t * 3 + 2] : 0.f;
    if (gz == 0.f && g0 == 0.f && g1 == 0.f && g2 == 0.f) return;
    const long pix = t / K;
    const int rem = (int)(pix % ((long)H * W));
    const int yi = rem / W, xi = rem - yi * W;
    const float yf = -1.0f + (2.0f * (float)(H - 1 - yi) + 1.0f) / (float)H;
    const float xf = -1.0f + (2.0f * (float)(W - 1 - xi) + 1.0f) / (float)W;
    const float* v = face_verts + f * 9;
    float* gf = grad_face_verts + f * 9;
    const D1 px = d1_const(xf), py = d1_const(yf);
    for (int j = 0; j < 9; ++j) {  // directional derivative along coordinate j of the face
        D1 c[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) c[i] = D1{v[i], i == j ? 1.f : 0.f};
        const D1 x0 = c[0], y0 = c[1], z0 = c[2], x1 = c[3], y1 = c[4], z1 = c[5], x2 = c[6], y2 = c[7], z2 = c[8];
        const D1 ar = d1_edge(x2, y2, x0, y0, x1, y1) + d1_const(kEpsilon);
        D1 p0 = d1_edge(px, py, x1, y1, x2, y2) / ar, p1 = d1_edge(px, py, x2, y2, x0, y0) / ar, p2 = d1_edge(px, py, x0, y0, x1, y1) / ar;
        if (persp) {
            const D1 w0 = p0 * z1 * z2, w1 = z0 * p1 * z2, w2 = z0 * z1 * p2;
            const D1 den = d1_max(w0 + w1 + w2, kEpsilon);
            p0 = w0 / den; p1 = w1 / den; p2 = w2 / den;
        }
        if (clipb) {
            p0 = d1_max(p0, 0.0f); p1 = d1_max(p1, 0.0f); p2 = d1_max(p2, 0.0f);
            const D1 sm = d1_max(p0 + p1 + p2, kBaryClipMin);
            p0 = p0 / sm; p1 = p1 / sm; p2 = p2 / sm;
        }
        const D1 pz = p0 * z0 + p1 * z1 + p2 * z2;
        const float g = g0 * p0.d + g1 * p1.d + g2 * p2.d + gz * pz.d;
        if (g != 0.f) atomicAdd(gf + j, g);
    }
}
#pragma clang fp contract(fast)
