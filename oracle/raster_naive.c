/*
 * oracle/raster_naive.c  --  TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * CPU restatement of the rasterisation arithmetic that OcclusionEnv reaches through
 * PyTorch3D (pinned pytorch3d==0.6.2 in /root/reference/requirements.txt:50, 0.7.0 in
 * conda_environment.yml:61).  PyTorch3D is NOT vendored in the reference and is not installed
 * here, so this file restates its *published* algorithm (pytorch3d/csrc/rasterize_meshes/
 * rasterize_meshes_cpu.cpp + csrc/utils/geometry_utils.h, upstream v0.6.2/0.7.0) from the
 * description in SURVEY.md Appendix A.4/A.5.  The reference call sites it serves are
 * /root/reference/environment.py:249-255 (soft, K=100, blur>0, cull_backfaces) and :267-273
 * (hard, K=1, blur=0), called at :310,316-318,370-372,375.
 *
 * PARITY UNPINNED: the reference has no tests/golden vectors for this path
 * and PyTorch3D cannot be executed in this container, so this restatement is pinned only by
 * the build's own closed-form / finite-difference tests (tests/test_oracle_*.py).
 *
 * The file is compiled twice (-DREAL=float / -DREAL=double): the f32 build is the oracle the
 * HIP path is compared with; the f64 build exists to finite-difference the analytic backward.
 * Single-threaded, one image per call, no FMA contraction (-ffp-contract=off) to mimic the
 * generic x86 wheels of PyTorch3D.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
#ifndef SUFFIX
#define SUFFIX f32
#endif
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

static const REAL kEpsilon = (REAL)1e-8; /* SURVEY A.0 */

typedef struct { REAL x, y; } v2;
typedef struct { REAL x, y, z; } v3;

static inline REAL rmin(REAL a, REAL b) { return a < b ? a : b; }
static inline REAL rmax(REAL a, REAL b) { return a > b ? a : b; }
static inline REAL dot2(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
static inline v2 sub2(v2 a, v2 b) { v2 r = {a.x - b.x, a.y - b.y}; return r; }

/* E(p; a, b) = (p-a) x (b-a)    */
static inline REAL edge_fn(v2 p, v2 a, v2 b) {
    return (p.x - a.x) * (b.y - a.y) - (p.y - a.y) * (b.x - a.x);
}

static inline v3 bary_coords(v2 p, v2 v0, v2 v1, v2 v2_) {
    const REAL area = edge_fn(v2_, v0, v1) + kEpsilon;
    v3 w;
    w.x = edge_fn(p, v1, v2_) / area;
    w.y = edge_fn(p, v2_, v0) / area;
    w.z = edge_fn(p, v0, v1) / area;
    return w;
}

static inline v3 bary_persp(v3 b, REAL z0, REAL z1, REAL z2) {
    const REAL w0 = b.x * z1 * z2;
    const REAL w1 = z0 * b.y * z2;
    const REAL w2 = z0 * z1 * b.z;
    const REAL denom = rmax(w0 + w1 + w2, kEpsilon);
    v3 r = {w0 / denom, w1 / denom, w2 / denom};
    return r;
}

/* lower-bound-only clamp then renormalise (upstream "_clip_barycentric_coordinates") */
static inline v3 bary_clip(v3 b) {
    v3 w = {rmax(b.x, (REAL)0), rmax(b.y, (REAL)0), rmax(b.z, (REAL)0)};
    const REAL s = rmax(w.x + w.y + w.z, (REAL)1e-5);
    w.x /= s; w.y /= s; w.z /= s;
    return w;
}

/* squared distance of p to segment a-b */
static inline REAL point_line_dist(v2 p, v2 a, v2 b) {
    const v2 ba = sub2(b, a);
    const REAL l2 = dot2(ba, ba);
    if (l2 <= kEpsilon) {
        const v2 d = sub2(p, b);
        return dot2(d, d);
    }
    REAL t = dot2(ba, sub2(p, a)) / l2;
    t = rmin(rmax(t, (REAL)0), (REAL)1);
    const v2 proj = {a.x + t * ba.x, a.y + t * ba.y};
    const v2 d = sub2(proj, p);
    return dot2(d, d);
}

static inline REAL point_tri_dist(v2 p, v2 v0, v2 v1, v2 v2_, int* amin) {
    const REAL e01 = point_line_dist(p, v0, v1);
    const REAL e02 = point_line_dist(p, v0, v2_);
    const REAL e12 = point_line_dist(p, v1, v2_);
    *amin = (e01 <= e02 && e01 <= e12) ? 0 : ((e02 <= e01 && e02 <= e12) ? 1 : 2);
    return rmin(rmin(e01, e02), e12);
}

/* gradient of the squared point-segment distance wrt a and b (t treated as a constant,
 * which is exact: interior t is the minimiser, clamped t is constant)  -- SURVEY A.5 */
static inline void point_line_dist_bwd(v2 p, v2 a, v2 b, REAL g, v2* ga, v2* gb) {
    const v2 ba = sub2(b, a);
    const v2 pa = sub2(p, a);
    REAL t = dot2(ba, pa) / (dot2(ba, ba) + kEpsilon);
    t = rmin(rmax(t, (REAL)0), (REAL)1);
    const v2 proj = {((REAL)1 - t) * a.x + t * b.x, ((REAL)1 - t) * a.y + t * b.y};
    const v2 d = sub2(proj, p);
    ga->x = g * ((REAL)1 - t) * (REAL)2 * d.x;
    ga->y = g * ((REAL)1 - t) * (REAL)2 * d.y;
    gb->x = g * t * (REAL)2 * d.x;
    gb->y = g * t * (REAL)2 * d.y;
}

typedef struct {
    REAL z; int64_t f; REAL d; REAL b0, b1, b2;
    int amin; /* closest edge: 0 = (v0,v1), 1 = (v0,v2), 2 = (v1,v2); not part of the ordering */
} qent;

/* lexicographic (z, f, d, b0, b1, b2) like std::tuple operator<  */
static inline int qless(const qent* a, const qent* b) {
    if (a->z != b->z) return a->z < b->z;
    if (a->f != b->f) return a->f < b->f;
    if (a->d != b->d) return a->d < b->d;
    if (a->b0 != b->b0) return a->b0 < b->b0;
    if (a->b1 != b->b1) return a->b1 < b->b1;
    return a->b2 < b->b2;
}

/*
 * Naive rasteriser for ONE image (upstream RasterizeMeshesNaiveCpu).
 *   face_verts : (F,3,3) rows (x_ndc, y_ndc, z_view)
 *   neighbor   : (F,) index of the clipped-pair partner or -1 (may be NULL)
 * Outputs (H,W,K[,3]); empty slots hold -1.
 */
int FN(orc_rasterize_naive)(const REAL* face_verts, const int64_t* neighbor, int64_t F,
                            int H, int W, REAL blur_radius, int K,
                            int perspective_correct, int clip_barycentric, int cull_backfaces,
                            int64_t* pix_to_face, REAL* zbuf, REAL* bary, REAL* dists) {
    qent* q = (qent*)malloc(sizeof(qent) * (size_t)(K + 1));
    if (!q) return 1;
    const REAL sq_blur = (REAL)sqrt((double)blur_radius);
    const size_t npix = (size_t)H * (size_t)W;
    for (size_t i = 0; i < npix * (size_t)K; ++i) { pix_to_face[i] = -1; zbuf[i] = -1; dists[i] = -1; }
    for (size_t i = 0; i < npix * (size_t)K * 3; ++i) bary[i] = -1;

    for (int yi = 0; yi < H; ++yi) {
        const int yidx = H - 1 - yi; /* +Y up */
        const REAL yf = (REAL)-1 + ((REAL)2 * (REAL)yidx + (REAL)1) / (REAL)H;
        for (int xi = 0; xi < W; ++xi) {
            const int xidx = W - 1 - xi; /* +X left */
            const REAL xf = (REAL)-1 + ((REAL)2 * (REAL)xidx + (REAL)1) / (REAL)W;
            const v2 pxy = {xf, yf};
            int qn = 0;
            for (int64_t f = 0; f < F; ++f) {
                const REAL* fv = face_verts + f * 9;
                const v2 v0 = {fv[0], fv[1]}, v1 = {fv[3], fv[4]}, v2_ = {fv[6], fv[7]};
                const REAL z0 = fv[2], z1 = fv[5], z2 = fv[8];
                const REAL area = edge_fn(v0, v1, v2_);
                const int back_face = area < 0;
                if (cull_backfaces && back_face) continue;
                if (area <= kEpsilon && area >= -kEpsilon) continue;
                if (rmax(rmax(z0, z1), z2) < 0) continue;
                const REAL xmin = rmin(rmin(v0.x, v1.x), v2_.x) - sq_blur;
                const REAL xmax = rmax(rmax(v0.x, v1.x), v2_.x) + sq_blur;
                const REAL ymin = rmin(rmin(v0.y, v1.y), v2_.y) - sq_blur;
                const REAL ymax = rmax(rmax(v0.y, v1.y), v2_.y) + sq_blur;
                const int x_in = xmin <= xf && xf <= xmax;
                const int y_in = ymin <= yf && yf <= ymax;
                if (!(x_in && y_in)) continue;

                const v3 b0 = bary_coords(pxy, v0, v1, v2_);
                const v3 b1 = perspective_correct ? bary_persp(b0, z0, z1, z2) : b0;
                const v3 bc = clip_barycentric ? bary_clip(b1) : b1;
                const REAL pz = bc.x * z0 + bc.y * z1 + bc.z * z2;
                if (pz < 0) continue;
                int amin;
                const REAL dist = point_tri_dist(pxy, v0, v1, v2_, &amin);
                const int inside = b1.x > 0 && b1.y > 0 && b1.z > 0;
                const REAL sdist = inside ? -dist : dist;
                if (!inside && dist >= blur_radius) continue;

                const qent e = {pz, f, sdist, bc.x, bc.y, bc.z, amin};
                int idx_top = -1;
                const int64_t nb = neighbor ? neighbor[f] : -1;
                if (nb != -1) {
                    for (int i = 0; i < qn; ++i) if (q[i].f == nb) { idx_top = i; break; }
                }
                if (idx_top != -1) {
                    /* Upstream: "if (dist < neighbor_dist) overwrite".  When the closest edge of BOTH halves is
                     * the diagonal they share -- t1 = (p4,p2,p5): edge (v1,v2); t2 = (p5,p2,p3): edge (v0,v1),
                     * SURVEY A.3 -- the two distances are the same number in exact arithmetic and the strict '<'
                     * keeps the first half; in floating point the outcome is rounding noise (upstream's CPU and
                     * CUDA paths disagree with each other there).  This restatement fixes that case to the
                     * exact-arithmetic answer (keep the first half) so that it is well defined. */
                    const int shared_tie = (nb == f - 1 && q[idx_top].amin == 2 && amin == 0) ||
                                           (nb == f + 1 && q[idx_top].amin == 0 && amin == 2);
                    const REAL dn = (REAL)fabs((double)q[idx_top].d);
                    if (!shared_tie && dist < dn) q[idx_top] = e;
                } else {
                    q[qn++] = e;
                }
                /* keep sorted ascending (insertion sort == std::sort result, keys are unique in f) */
                for (int i = 1; i < qn; ++i) {
                    qent t = q[i]; int j = i - 1;
                    while (j >= 0 && qless(&t, &q[j])) { q[j + 1] = q[j]; --j; }
                    q[j + 1] = t;
                }
                if (qn > K) qn = K; /* drop the largest */
            }
            const size_t base = ((size_t)yi * (size_t)W + (size_t)xi) * (size_t)K;
            for (int i = 0; i < qn; ++i) {
                pix_to_face[base + i] = q[i].f;
                zbuf[base + i] = q[i].z;
                dists[base + i] = q[i].d;
                bary[(base + i) * 3 + 0] = q[i].b0;
                bary[(base + i) * 3 + 1] = q[i].b1;
                bary[(base + i) * 3 + 2] = q[i].b2;
            }
        }
    }
    free(q);
    return 0;
}

/*
 * Backward of the dists output only (upstream RasterizeMeshesBackwardCpu).  On the
 * OcclusionEnv path the shaders ignore zbuf and bary of the soft render (SoftSilhouetteShader,
 * /root/reference/environment.py:263) and nothing differentiates the hard render, so grad_zbuf
 * and grad_bary are identically zero there; they are not restated.
 *   grad_face_verts (F,3,3) is overwritten.
 */
int FN(orc_rasterize_backward_dists)(const REAL* face_verts, const int64_t* pix_to_face,
                                     const REAL* grad_dists, int64_t F, int H, int W, int K,
                                     int perspective_correct, int clip_barycentric,
                                     REAL* grad_face_verts) {
    memset(grad_face_verts, 0, sizeof(REAL) * (size_t)F * 9);
    for (int yi = 0; yi < H; ++yi) {
        const int yidx = H - 1 - yi;
        const REAL yf = (REAL)-1 + ((REAL)2 * (REAL)yidx + (REAL)1) / (REAL)H;
        for (int xi = 0; xi < W; ++xi) {
            const int xidx = W - 1 - xi;
            const REAL xf = (REAL)-1 + ((REAL)2 * (REAL)xidx + (REAL)1) / (REAL)W;
            const v2 pxy = {xf, yf};
            const size_t base = ((size_t)yi * (size_t)W + (size_t)xi) * (size_t)K;
            for (int k = 0; k < K; ++k) {
                const int64_t f = pix_to_face[base + k];
                if (f < 0) break; /* slots are sorted: first empty ends the list */
                const REAL* fv = face_verts + f * 9;
                const v2 v0 = {fv[0], fv[1]}, v1 = {fv[3], fv[4]}, v2_ = {fv[6], fv[7]};
                const REAL z0 = fv[2], z1 = fv[5], z2 = fv[8];
                const v3 b0 = bary_coords(pxy, v0, v1, v2_);
                const v3 b1 = perspective_correct ? bary_persp(b0, z0, z1, z2) : b0;
                const v3 bc = clip_barycentric ? bary_clip(b1) : b1;
                const int inside = bc.x > 0 && bc.y > 0 && bc.z > 0;
                const REAL sign = inside ? (REAL)-1 : (REAL)1;
                const REAL g = sign * grad_dists[base + k];
                const REAL e01 = point_line_dist(pxy, v0, v1);
                const REAL e02 = point_line_dist(pxy, v0, v2_);
                const REAL e12 = point_line_dist(pxy, v1, v2_);
                REAL* gf = grad_face_verts + f * 9;
                v2 ga, gb;
                if (e01 <= e02 && e01 <= e12) {
                    point_line_dist_bwd(pxy, v0, v1, g, &ga, &gb);
                    gf[0] += ga.x; gf[1] += ga.y; gf[3] += gb.x; gf[4] += gb.y;
                } else if (e02 <= e01 && e02 <= e12) {
                    point_line_dist_bwd(pxy, v0, v2_, g, &ga, &gb);
                    gf[0] += ga.x; gf[1] += ga.y; gf[6] += gb.x; gf[7] += gb.y;
                } else if (e12 <= e01 && e12 <= e02) {
                    point_line_dist_bwd(pxy, v1, v2_, g, &ga, &gb);
                    gf[3] += ga.x; gf[4] += ga.y; gf[6] += gb.x; gf[7] += gb.y;
                }
            }
        }
    }
    return 0;
}

/*
 * Diagnostic for the parity tests' tie classifier (tests/parity_utils.py): everything the naive rasteriser
 * computes for ONE pixel, for every face that is a candidate there OR misses being one by a hair
 * (dist within blur*(1+band), bbox widened by the same relative band, signed area within area_band + vert_band *
 * perimeter of the kEpsilon visibility threshold: moving each vertex by <= vert_band moves the area by at most that).  No clipped-pair rule here (the classifier only asks how close a decision was).
 * Rows: f, z (pz), dist (unsigned), minb = smallest of the perspective-corrected barycentrics (the `inside` test is
 * minb > 0), flags: 1 = inside, 2 = candidate under the exact rule of orc_rasterize_naive, 4 = pz < 0,
 * 8 = the face's area is within that band of kEpsilon (visible / culled by a hair; such a face is reported
 * whichever side it fell).  Returns the number of rows (<= max_out), -1 on bad args.
 */
int FN(orc_pixel_candidates)(const REAL* face_verts, int64_t F, int H, int W, int yi, int xi, REAL blur_radius,
                             int perspective_correct, int clip_barycentric, int cull_backfaces, REAL band,
                             REAL area_band, REAL vert_band, int64_t* out_f, REAL* out_z, REAL* out_dist,
                             REAL* out_minb, int32_t* out_flags, int max_out) {
    if (!face_verts || yi < 0 || yi >= H || xi < 0 || xi >= W || max_out <= 0) return -1;
    const REAL sq_blur = (REAL)sqrt((double)blur_radius);
    const REAL sq_wide = (REAL)sqrt((double)(blur_radius * ((REAL)1 + band))) + (REAL)1e-6;
    const REAL yf = (REAL)-1 + ((REAL)2 * (REAL)(H - 1 - yi) + (REAL)1) / (REAL)H;
    const REAL xf = (REAL)-1 + ((REAL)2 * (REAL)(W - 1 - xi) + (REAL)1) / (REAL)W;
    const v2 pxy = {xf, yf};
    int n = 0;
    for (int64_t f = 0; f < F && n < max_out; ++f) {
        const REAL* fv = face_verts + f * 9;
        const v2 v0 = {fv[0], fv[1]}, v1 = {fv[3], fv[4]}, v2_ = {fv[6], fv[7]};
        const REAL z0 = fv[2], z1 = fv[5], z2 = fv[8];
        const REAL area = edge_fn(v0, v1, v2_);
        const double perim = sqrt((double)((v1.x - v0.x) * (v1.x - v0.x) + (v1.y - v0.y) * (v1.y - v0.y))) +
                             sqrt((double)((v2_.x - v1.x) * (v2_.x - v1.x) + (v2_.y - v1.y) * (v2_.y - v1.y))) +
                             sqrt((double)((v0.x - v2_.x) * (v0.x - v2_.x) + (v0.y - v2_.y) * (v0.y - v2_.y)));
        const REAL ab = area_band + vert_band * (REAL)perim;
        const int hair = cull_backfaces ? (area > kEpsilon - ab && area < kEpsilon + ab)
                                        : ((area > kEpsilon - ab && area < kEpsilon + ab) ||
                                           (-area > kEpsilon - ab && -area < kEpsilon + ab));
        const int visible = !(cull_backfaces && area < 0) && !(area <= kEpsilon && area >= -kEpsilon);
        if (!visible && !hair) continue;
        if (rmax(rmax(z0, z1), z2) < 0) continue;
        const REAL bx0 = rmin(rmin(v0.x, v1.x), v2_.x), bx1 = rmax(rmax(v0.x, v1.x), v2_.x);
        const REAL by0 = rmin(rmin(v0.y, v1.y), v2_.y), by1 = rmax(rmax(v0.y, v1.y), v2_.y);
        if (!(bx0 - sq_wide <= xf && xf <= bx1 + sq_wide && by0 - sq_wide <= yf && yf <= by1 + sq_wide)) continue;
        const int in_box = (bx0 - sq_blur <= xf && xf <= bx1 + sq_blur && by0 - sq_blur <= yf && yf <= by1 + sq_blur);
        const v3 b0 = bary_coords(pxy, v0, v1, v2_);
        const v3 b1 = perspective_correct ? bary_persp(b0, z0, z1, z2) : b0;
        const v3 bc = clip_barycentric ? bary_clip(b1) : b1;
        const REAL pz = bc.x * z0 + bc.y * z1 + bc.z * z2;
        int amin;
        const REAL dist = point_tri_dist(pxy, v0, v1, v2_, &amin);
        const int inside = b1.x > 0 && b1.y > 0 && b1.z > 0;
        const REAL minb = rmin(rmin(b1.x, b1.y), b1.z);
        /* on (or a hair off) an edge: within 1e-6 NDC of the triangle's boundary */
        const int near_inside = dist <= (REAL)1e-12;
        if (!inside && !near_inside && !(dist < blur_radius * ((REAL)1 + band))) continue;
        const int cand = visible && in_box && !(pz < 0) && (inside || dist < blur_radius);
        out_f[n] = f;
        out_z[n] = pz;
        out_dist[n] = dist;
        out_minb[n] = minb;
        out_flags[n] = (inside ? 1 : 0) | (cand ? 2 : 0) | (pz < 0 ? 4 : 0) | (hair ? 8 : 0);
        ++n;
    }
    return n;
}
