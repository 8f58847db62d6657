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
#pragma clang fp contract(fast)
