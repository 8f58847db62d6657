// occ_combine.hpp -- combine / reduce / finish kernels and the host-driven reset hand-off.
// Part of the single translation unit occ_kernels.hip (included inside namespace occ; not a stand-alone header).

// ------------------------------------------------------------------------------------------
// combine kernel: one thread per pixel - occlusion image, loss / gradient partials, shading, outputs
// ------------------------------------------------------------------------------------------
// this step's footprint of env: the union of its object rects, in pixels (empty: x1 < x0)
__device__ __forceinline__ int4 env_footprint(const OccWorkspace& ws, int env, int S) {
    int ux0 = S, uy0 = S, ux1 = -1, uy1 = -1;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const int eo = env * 3 + o;
        ciptr rect = as_const(ws.objrect + eo * 4);
        if (as_const(ws.nrec + eo)[0] > 0 && rect[2] >= rect[0] && rect[3] >= rect[1]) {
            ux0 = min(ux0, rect[0] * OCC_BLOCK);
            uy0 = min(uy0, rect[1] * OCC_BLOCK);
            ux1 = max(ux1, rect[2] * OCC_BLOCK + OCC_BLOCK - 1);
            uy1 = max(uy1, rect[3] * OCC_BLOCK + OCC_BLOCK - 1);
        }
    }
    return make_int4(ux0, uy0, ux1, uy1);
}
// the 256-pixel blocks of an S x S frame that meet pixel rows [y0, y1] (empty: first > last)
__device__ __forceinline__ void rows_to_blocks(int y0, int y1, int S, int bpe, int& b0, int& b1) {
    if (y1 < y0) { b0 = 1; b1 = 0; return; }
    b0 = max((y0 * S) >> 8, 0);
    b1 = min((y1 * S + S - 1) >> 8, bpe - 1);
}

template <bool SOFT, bool HARD, bool GRAD>
__device__ __forceinline__ void combine_block(const RasterParams& P, int bpe, int env, int blk, float (*s_red)[3],
                                              const int32_t* __restrict__ rprev, const int32_t* __restrict__ aprev);

// Grid = n_env x G thread blocks; the blocks of an env share its 256-pixel blocks round robin.  With every written
// output under REGION TRACKING (OccRenderOut.rect_prev / arect_prev: outside those rects the persistent outputs already
// hold their background values) only the pixel blocks between the first and the last row that this step's footprint or
// one of those rects touches are visited at all: a launch of one thread block per 256 pixels spent most of its time
// starting 57 000 blocks that read six rects and exit (three quarters of a 128 x 128 frame are background).
// (__launch_bounds__(256, 6): 74 registers instead of 83, six waves per SIMD instead of five - the kernel lives on memory
// latency: -15 us.  Eight waves (64 registers, 48 B of scratch): the same.  Issuing the next pixel block's plane loads
// before working on the current one - software pipelining at 94 - 102 registers - gained nothing over that.)
template <bool SOFT, bool HARD, bool GRAD>
__global__ __launch_bounds__(256, 6) void occ_combine_kernel(RasterParams P, int bpe, int G) {
    __shared__ float s_red[4][3];
    const int tid = threadIdx.x;
    const int env = blockIdx.x / G, g = blockIdx.x - env * G;
    const int S = P.sc.img;
    const int32_t* __restrict__ rprev = (HARD || SOFT) ? P.out.rect_prev : nullptr;
    const int32_t* __restrict__ aprev = SOFT ? P.out.arect_prev : nullptr;
    if (P.sc.skip && P.sc.skip[env]) {  // outputs of a skipped scene row stay untouched - and so do their rects
        if (g == 0 && tid < 4) {
            if (rprev && P.out.rect_next) P.out.rect_next[env * 4 + tid] = rprev[env * 4 + tid];
            if (aprev && P.out.arect_next) P.out.arect_next[env * 4 + tid] = aprev[env * 4 + tid];
        }
        return;
    }
    const int4 fp = env_footprint(P.ws, env, S);
    if (g == 0 && tid == 0) {
        if (P.out.rect_next) reinterpret_cast<int4*>(P.out.rect_next)[env] = fp;
        if (P.out.arect_next) reinterpret_cast<int4*>(P.out.arect_next)[env] = fp;
    }
    // pixel blocks to visit: all of them unless every output this variant writes is tracked
    int b0 = 0, b1 = bpe - 1;
    const bool o_tracked = rprev != nullptr || !((HARD && P.out.obs) || (SOFT && P.out.full_state));
    const bool a_tracked = aprev != nullptr || !(SOFT && P.out.alphas);
    if (o_tracked && a_tracked) {
        int y0 = fp.y, y1 = fp.w;
        if (rprev) {
            const int4 r = reinterpret_cast<const int4*>(rprev)[env];
            if (r.z >= r.x && r.w >= r.y) { y0 = min(y0, r.y); y1 = max(y1, r.w); }
        }
        if (aprev) {
            const int4 r = reinterpret_cast<const int4*>(aprev)[env];
            if (r.z >= r.x && r.w >= r.y) { y0 = min(y0, r.y); y1 = max(y1, r.w); }
        }
        rows_to_blocks(y0, y1, S, bpe, b0, b1);
    }
    for (int blk = b0 + g; blk <= b1; blk += G) {
        combine_block<SOFT, HARD, GRAD>(P, bpe, env, blk, s_red, rprev, aprev);
        __syncthreads();  // s_red is reused by the next pixel block
    }
}

template <bool SOFT, bool HARD, bool GRAD>
__device__ __forceinline__ void combine_block(const RasterParams& P, int bpe, int env, int blk, float (*s_red)[3],
                                              const int32_t* __restrict__ rprev, const int32_t* __restrict__ aprev) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = P.sc.img;
    const int pidx = env * bpe + blk;  // this pixel block's slot in ws.partials
    const int pix = blk * 256 + tid;
    const bool live = pix < S * S;
    const int yi = live ? pix / S : 0, xi = live ? pix - (pix / S) * S : 0;
    const int tx = xi / OCC_BLOCK, ty = yi / OCC_BLOCK;
    const int cap = P.sc.rec_cap;
#ifndef OCC_NO_BG_FASTPATH
    {
        // BACKGROUND BLOCKS (three quarters of them in the bench workload): none of the block's 256 consecutive pixels
        // lies in any object's rect - nothing to read, the outputs are constants.  Written with one 16-byte store per
        // thread and array (a pixel-per-thread block issues eight 4-byte plane stores); same values as the path below.
        const int p0 = blk * 256, p1 = p0 + 255;
        bool bg = p1 < S * S;
        const int r0 = p0 / S, r1 = p1 / S, c0 = p0 - r0 * S, c1 = c0 + 255;
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const int eo = env * 3 + o;
            ciptr rect = as_const(P.ws.objrect + eo * 4);
            const bool hit = as_const(P.ws.nrec + eo)[0] > 0 && r1 >= rect[1] * OCC_BLOCK && r0 <= rect[3] * OCC_BLOCK + OCC_BLOCK - 1 &&
                             (r0 != r1 || (c1 >= rect[0] * OCC_BLOCK && c0 <= rect[2] * OCC_BLOCK + OCC_BLOCK - 1));
            bg = bg && !hit;
        }
        if (bg) {  // block-uniform
            const size_t S2 = (size_t)S * S;
            const int plane = tid >> 6, quad = tid & 63;
            // a tracked buffer is background here already unless the block meets what the buffer last held
            auto meets = [&](const int32_t* __restrict__ rp) {
                if (!rp) return true;
                const int4 r = reinterpret_cast<const int4*>(rp)[env];  // x0, y0, x1, y1
                return r1 >= r.y && r0 <= r.w && (r0 != r1 || (c1 >= r.x && c0 <= r.z));
            };
            const bool wo = meets(rprev), wa = meets(aprev);
            if (SOFT) {
                if (tid == 0) reinterpret_cast<float4*>(P.ws.partials)[pidx] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (P.out.full_state && wo) reinterpret_cast<float4*>(P.out.full_state)[(size_t)env * S2 + pix] = make_float4(3.f, 3.f, 3.f, 0.f);
                if (P.out.alphas && plane < 3 && wa)
                    reinterpret_cast<float4*>(P.out.alphas + ((size_t)env * 3 + plane) * S2 + p0)[quad] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (HARD && wo) {
                const float v = plane < 3 ? 1.f : -1.f;  // white background, depth -1 (environment.py:378)
                reinterpret_cast<float4*>(P.out.obs + ((size_t)env * 4 + plane) * S2 + p0)[quad] = make_float4(v, v, v, v);
            }
            return;
        }
    }
#endif
    float alpha[3] = {0.f, 0.f, 0.f}, dae[3] = {0.f, 0.f, 0.f}, daa[3] = {0.f, 0.f, 0.f};
    float hz = 3.0e38f;
    int hrec = -1, hobj = 0;
    bool touched = false;  // the pixel lies in some object's rect of this step
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const int eo = env * 3 + o;
        ciptr rect = as_const(P.ws.objrect + eo * 4);
        const bool in = live && as_const(P.ws.nrec + eo)[0] > 0 && tx >= rect[0] && ty >= rect[1] && tx <= rect[2] &&
                        ty <= rect[3];
        touched = touched || in;
        if (in) {
            const size_t opix = ((size_t)eo * S + yi) * S + xi;
            if (SOFT) {
                alpha[o] = P.ws.obj_alpha[opix];
                if (GRAD) {
                    const float2 g = reinterpret_cast<const float2*>(P.ws.obj_grad)[opix];
                    dae[o] = g.x;
                    daa[o] = g.y;
                }
            }
            if (HARD) {
                // (the record index is fetched WITH the depth, not after the comparison: the kernel lives on memory latency,
                // and a load that waits for another load's result is a round trip more per object)
                const float z = P.ws.obj_hz[opix];
                const int hr = P.ws.obj_hrec[opix];
                if (z < hz) {  // strict: on equal depth the earlier object of the joined scene wins
                    hz = z;
                    hrec = hr;
                    hobj = o;
                }
            }
        }
    }
    const size_t gp = (size_t)yi * S + xi;
    // region tracking, per pixel: a tracked buffer already holds the background value at a pixel that lies neither in
    // this step's object rects nor in the rect of what the buffer held before
    auto in_prev = [&](const int32_t* __restrict__ rp) {
        if (!rp) return true;
        const int4 r = reinterpret_cast<const int4*>(rp)[env];
        return xi >= r.x && xi <= r.z && yi >= r.y && yi <= r.w;
    };
    const bool wr_o = touched || in_prev(rprev), wr_a = touched || in_prev(aprev);
    if (SOFT) {
        // environment.py:373: image = i1*i2 + i2*i3 + i1*i3 ; RGB of every silhouette is 1
        const float I = alpha[0] * alpha[1] + alpha[1] * alpha[2] + alpha[0] * alpha[2];
        // optional per-pixel weight of the loss term (OccScene.pix_weight)
        const float wpx = (live && P.sc.pix_weight) ? P.sc.pix_weight[(size_t)env * S * S + gp] : 1.0f;
        float lsum = live ? wpx * (I * I) : 0.f, ge = 0.f, ga = 0.f;
        if (GRAD && live) {
            const float g0 = alpha[1] + alpha[2], g1 = alpha[0] + alpha[2], g2 = alpha[0] + alpha[1];
            ge = wpx * (2.0f * I * (g0 * dae[0] + g1 * dae[1] + g2 * dae[2]));
            ga = wpx * (2.0f * I * (g0 * daa[0] + g1 * daa[1] + g2 * daa[2]));
        }
        lsum = wave_sum_dpp(lsum);  // (DPP row operations: a fixed order, no LDS-crossbar round trips)
        if (GRAD) {
            ge = wave_sum_dpp(ge);
            ga = wave_sum_dpp(ga);
        }
        if (lane == 0) {
            s_red[wave][0] = lsum;
            s_red[wave][1] = ge;
            s_red[wave][2] = ga;
        }
        __syncthreads();
        if (tid == 0) {
            float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                a += s_red[w][0];
                b += s_red[w][1];
                c += s_red[w][2];
            }
            reinterpret_cast<float4*>(P.ws.partials)[pidx] = make_float4(a, b, c, 0.f);
        }
        if (live) {
            if (P.out.full_state && wr_o)
                reinterpret_cast<float4*>(P.out.full_state)[(size_t)env * S * S + gp] = make_float4(3.f, 3.f, 3.f, I);
            if (P.out.alphas && wr_a) {
                float* __restrict__ al = P.out.alphas + (size_t)env * 3 * S * S + gp;
                al[0] = alpha[0];
                al[(size_t)S * S] = alpha[1];
                al[(size_t)2 * S * S] = alpha[2];
            }
        }
    }
    if (HARD && live && wr_o) {
        // [P3D] HardFlatShader + hard_rgb_blend; depth in channel 3 (environment.py:378)
        float cr = 1.f, cg = 1.f, cb = 1.f, depth = -1.f;
        // where the three objects' records start: block-uniform (scalar loads), fetched before anything depends on the
        // nearest object - as a per-pixel load it was one more dependent round trip between the depth and the record
        size_t rb[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) rb[o] = rec_span(P.ws, cap, env * 3 + o).base;
        if (hrec >= 0) {
            const int eo = env * 3 + hobj;
            const size_t rbase = hobj == 0 ? rb[0] : (hobj == 1 ? rb[1] : rb[2]);
            const float* __restrict__ r = P.ws.rec + (rbase + hrec) * OCC_REC_STRIDE;
            const int fid = __float_as_int(r[R_ID]);
            const int mesh = P.sc.scene_mesh[eo];
            const int vo = P.sc.mesh_vert_off[mesh], fo = P.sc.mesh_face_off[mesh];
            const float ox = P.sc.scene_offset[eo * 3], oy = P.sc.scene_offset[eo * 3 + 1], oz = P.sc.scene_offset[eo * 3 + 2];
            // per-face shading terms from the setup kernel (flat_shade): one gather instead of face -> 3 vertices
            const float amb_diff = r[R_AMB];
            const float spec = r[R_SPEC];
            const float* __restrict__ cm = P.cam + (size_t)env * OCC_CAM_STRIDE;
            // texel: white TexturesVertex interpolated with the (unclipped) barycentrics, or the face's atlas
            const float fS = (float)S;
            const float xf = -1.0f + (2.0f * (float)(S - 1 - xi) + 1.0f) / fS;
            const float yf = -1.0f + (2.0f * (float)(S - 1 - yi) + 1.0f) / fS;
            const float x0 = r[R_X0], y0 = r[R_Y0], z0 = r[R_Z0], x1 = r[R_X1], y1 = r[R_Y1], z1 = r[R_Z1];
            const float x2 = r[R_X2], y2 = r[R_Y2], z2 = r[R_Z2];
            const float ia = r[R_INV_AREA];
            const float b0 = ((xf - x1) * (y2 - y1) - (yf - y1) * (x2 - x1)) * ia;
            const float b1 = ((yf - y2) * (x2 - x0) - (xf - x2) * (y2 - y0)) * ia;
            const float b2 = ((xf - x0) * (y1 - y0) - (yf - y0) * (x1 - x0)) * ia;
            const float w0 = b0 * z1 * z2, w1 = z0 * b1 * z2, w2 = z0 * z1 * b2;
            const float den = fmaxf(w0 + w1 + w2, kEpsilon);
            float q0 = w0 / den, q1 = w1 / den, q2 = w2 / den;
            float tr = q0 + q1 + q2, tg = tr, tb = tr;
            const int64_t aoff = P.sc.pool_atlas ? P.sc.mesh_atlas_off[mesh] : -1;
            const int shader = P.sc.pool_vnormals ? P.sc.shader : OCC_SHADER_FLAT;
            if ((aoff >= 0 || shader != OCC_SHADER_FLAT) && (__float_as_int(r[R_FLAGS]) & FLAG_CLIPPED)) {
                // [P3D] convert_clipped_rasterization_to_original_faces: barycentrics w.r.t. the ORIGINAL face.
                // Perspective-correct barycentrics are the 3-D ones: beta_i ~ d . (V_j x V_k) with d the pixel ray
                // and V the face's view-space vertices (valid for vertices behind the clip plane too).
                float V[3][3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int vi = P.sc.pool_faces[(size_t)(fo + fid) * 3 + k];
                    const float* pv = P.sc.pool_verts + (size_t)(vo + vi) * 3;
                    const float wx_ = pv[0] + ox, wy_ = pv[1] + oy, wz_ = pv[2] + oz;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        V[k][j] = wx_ * cm[C_R + j] + wy_ * cm[C_R + 3 + j] + wz_ * cm[C_R + 6 + j] + cm[C_T + j];
                }
                const float dx = xf / kProjScale, dy = yf / kProjScale, dz = 1.0f;
                auto tri = [&](const float* a, const float* b) {
                    return dx * (a[1] * b[2] - a[2] * b[1]) + dy * (a[2] * b[0] - a[0] * b[2]) + dz * (a[0] * b[1] - a[1] * b[0]);
                };
                const float e0 = tri(V[1], V[2]), e1 = tri(V[2], V[0]), e2 = tri(V[0], V[1]);
                const float es = e0 + e1 + e2;
                q0 = e0 / es; q1 = e1 / es; q2 = e2 / es;
                tr = tg = tb = q0 + q1 + q2;
            }
            if (aoff >= 0) {
                // [P3D] TexturesAtlas.sample_textures: (w0, w1) -> texel of the R x R grid, upper triangle mirrored
                const int Rr = P.sc.atlas_res;
                int wx = min((int)(q0 * (float)Rr), Rr - 1), wy = min((int)(q1 * (float)Rr), Rr - 1);
                const bool below = ((q0 + q1) * (float)Rr - ((float)wx + (float)wy)) <= 1.0f;
                if (!below) { wx = Rr - 1 - wx; wy = Rr - 1 - wy; }
                const float* tx = P.sc.pool_atlas + aoff + (((size_t)fid * Rr + wy) * Rr + wx) * 3;
                tr = tx[0]; tg = tx[1]; tb = tx[2];
            }
            float ad = amb_diff, sp = spec;
            if (shader != OCC_SHADER_FLAT) {
                // [P3D] phong_shading: position and normal of the PIXEL = barycentric interpolation of the face's
                // vertex positions / vertex normals (Meshes.verts_normals_packed), lighting as for the flat shader
                float px_ = 0.f, py_ = 0.f, pz_ = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
                const float qk[3] = {q0, q1, q2};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int vi = P.sc.pool_faces[(size_t)(fo + fid) * 3 + k];
                    const float* pv = P.sc.pool_verts + (size_t)(vo + vi) * 3;
                    const float* pn = P.sc.pool_vnormals + (size_t)(vo + vi) * 3;
                    px_ += qk[k] * (pv[0] + ox); py_ += qk[k] * (pv[1] + oy); pz_ += qk[k] * (pv[2] + oz);
                    nx += qk[k] * pn[0]; ny += qk[k] * pn[1]; nz += qk[k] * pn[2];
                }
                auto inv_len = [](float x, float y, float z) { return frcp(fmaxf(__builtin_amdgcn_sqrtf(x * x + y * y + z * z), kShadeEps)); };
                const float in_ = inv_len(nx, ny, nz);
                nx *= in_; ny *= in_; nz *= in_;
                float lx = kLightX - px_, ly = kLightY - py_, lz = kLightZ - pz_;
                const float il = inv_len(lx, ly, lz);
                lx *= il; ly *= il; lz *= il;
                const float cosang = nx * lx + ny * ly + nz * lz;
                float vx = cm[C_C] - px_, vy = cm[C_C + 1] - py_, vz = cm[C_C + 2] - pz_;
                const float iv = inv_len(vx, vy, vz);
                vx *= iv; vy *= iv; vz *= iv;
                const float rx = -lx + 2.f * (cosang * nx), ry = -ly + 2.f * (cosang * ny), rz = -lz + 2.f * (cosang * nz);
                float sa = fmaxf(vx * rx + vy * ry + vz * rz, 0.f) * (cosang > 0.f ? 1.f : 0.f);
                sa *= sa; sa *= sa; sa *= sa; sa *= sa; sa *= sa; sa *= sa;  // ^64
                ad = kAmbient + kDiffuse * fmaxf(cosang, 0.f);
                sp = kSpecular * sa;
            }
            cr = ad * tr + sp;
            cg = ad * tg + sp;
            cb = ad * tb + sp;
            if (shader == OCC_SHADER_SOFT_PHONG) {
                // [P3D] softmax_rgb_blend with K = 1, BlendParams() defaults (sigma = gamma = 1e-4, white background),
                // znear 1, zfar 100: weight = sigmoid(|d| / sigma) of the one face against delta = exp((eps - z_inv)/gamma)
                auto seg = [&](float ax, float ay, float bx, float by) {
                    const float ex = bx - ax, ey = by - ay, l2 = ex * ex + ey * ey;
                    if (l2 <= kEpsilon) return (xf - bx) * (xf - bx) + (yf - by) * (yf - by);
                    const float t = clamp01((ex * (xf - ax) + ey * (yf - ay)) / l2);
                    const float qx = ax + t * ex - xf, qy = ay + t * ey - yf;
                    return qx * qx + qy * qy;
                };
                const float dist = fmin3(seg(x0, y0, x1, y1), seg(x0, y0, x2, y2), seg(x1, y1, x2, y2));
                const float prob = 1.0f / (1.0f + __expf(-dist * kInvSigma));
                const float z_inv = fmaxf((100.0f - hz) / 99.0f, 1e-10f);
                const float delta = fmaxf(__expf((1e-10f - z_inv) * kInvSigma), 1e-10f);
                const float id = 1.0f / (prob + delta);
                cr = (prob * cr + delta) * id;
                cg = (prob * cg + delta) * id;
                cb = (prob * cb + delta) * id;
            }
            depth = hz;
        }
        float* __restrict__ ob = P.out.obs + (size_t)env * 4 * S * S + gp;
        ob[0] = cr;
        ob[(size_t)S * S] = cg;
        ob[(size_t)2 * S * S] = cb;
        ob[(size_t)3 * S * S] = depth;
    }
}

// ------------------------------------------------------------------------------------------
// per-env fixed-order reduction of the tile partials
// ------------------------------------------------------------------------------------------
// reward rule of one env (environment.py:381-392) + action Jacobian (:356-361): shared by occ_finish_kernel and the
// fused tail of occ_reduce_kernel
__device__ __forceinline__ void finish_one(int n, float l, float gl_e, float gl_a, bool have_grad, const float* __restrict__ cam,
                                           float* __restrict__ full_reward, const float* __restrict__ object_mass,
                                           float* __restrict__ reward, uint8_t* __restrict__ done, float* __restrict__ grad_action) {
    const float om = object_mass[n];
    float rw = full_reward[n] - l;
    full_reward[n] = l;
    const bool fin = l < kDoneThreshold;
    rw = rw / om;
    rw = fin ? rw + kDoneBonus : rw - kStepPenalty;
    reward[n] = rw;
    done[n] = fin ? 1 : 0;
    if (grad_action) {
        float ga0 = 0.f, ga1 = 0.f;
        if (have_grad) {
            const float* __restrict__ J = cam + (size_t)n * OCC_CAM_STRIDE + C_J;
            // d reward/d action_j = -(1/objectMass) * (dL/del * del/da_j + dL/daz * daz/da_j)
            ga0 = -(gl_e * J[0] + gl_a * J[2]) / om;
            ga1 = -(gl_e * J[1] + gl_a * J[3]) / om;
        }
        grad_action[2 * n] = ga0;
        grad_action[2 * n + 1] = ga1;
    }
}

struct FinishArgs {  // OccStepFinish by value (n_step = 0: no fused tail)
    float* full_reward; const float* object_mass; float* reward; uint8_t* done; float* grad_action; const float* cam;
    int n_step;
};

// Only the pixel blocks inside this step's footprint carry a partial (the combine kernel does not even visit the others
// when its outputs are tracked; where it does, they hold exact zeros): lane l adds its blocks l, l + 64, ... of the
// footprint in that order, which is the order - and the value - of the sum over all blocks.
__global__ __launch_bounds__(64) void occ_reduce_kernel(const float* __restrict__ partials, int ntiles,
                                                        float* __restrict__ loss, float* __restrict__ grad_elaz,
                                                        const int* __restrict__ skip, FinishArgs fin, OccWorkspace ws, int S) {
    const int env = blockIdx.x, lane = threadIdx.x;
    if (skip && skip[env]) return;
    const float4* __restrict__ p = reinterpret_cast<const float4*>(partials) + (size_t)env * ntiles;
    float l = 0.f, ge = 0.f, ga = 0.f;
    const int4 fp = env_footprint(ws, env, S);
    int tb0, tb1;
    rows_to_blocks(fp.y, fp.w, S, ntiles, tb0, tb1);
    for (int t = lane + (tb0 & ~63); t <= tb1; t += 64) {
        if (t < tb0) continue;
        const float4 v = p[t];
        l += v.x;
        ge += v.y;
        ga += v.z;
    }
    l = wave_sum_dpp(l);
    ge = wave_sum_dpp(ge);
    ga = wave_sum_dpp(ga);
    if (lane == 0) {
        if (loss) loss[env] = l;
        if (grad_elaz) {
            grad_elaz[2 * env] = ge;
            grad_elaz[2 * env + 1] = ga;
        }
        if (env < fin.n_step)
            finish_one(env, l, ge, ga, grad_elaz != nullptr, fin.cam, fin.full_reward, fin.object_mass, fin.reward, fin.done,
                       fin.grad_action);
    }
}

// environment.py:381-392 + action Jacobian (:356-361)
__global__ __launch_bounds__(64) void occ_finish_kernel(const float* __restrict__ loss, const float* __restrict__ grad_elaz,
                                                        const float* __restrict__ cam, float* __restrict__ full_reward,
                                                        const float* __restrict__ object_mass, float* __restrict__ reward,
                                                        uint8_t* __restrict__ done, float* __restrict__ grad_action,
                                                        int n_env) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_env) return;
    finish_one(n, loss[n], grad_elaz ? grad_elaz[2 * n] : 0.f, grad_elaz ? grad_elaz[2 * n + 1] : 0.f, grad_elaz != nullptr, cam,
               full_reward, object_mass, reward, done, grad_action);
}

// One small int32 buffer per step for the host: [0..n) done, [n..n+r) reserve scene passes the reset test
// (loss > 0.1, environment.py:327), [n+r] any kernel status bit set -> ONE device-to-host copy per step.
__global__ __launch_bounds__(256) void occ_flags_kernel(const uint8_t* __restrict__ done, const float* __restrict__ loss_all,
                                                        const int* __restrict__ status, int n, int r,
                                                        int* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) flags[n + r] = 0;
    if (i < n) flags[i] = done[i];
    else if (i < n + r) flags[i] = loss_all[i] > kDoneThreshold ? 1 : 0;
}
__global__ __launch_bounds__(256) void occ_status_any_kernel(const int* __restrict__ status, int nt, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nt && status[i] != 0) atomicOr(out, 1);
}

// Auto-reset commit: env dst[k] takes over reserve row src[k] (all per-env state + the freshly rendered
// observation) in one launch.  block = one (k, array) pair chunk.
struct CommitArgs {
    const int* pairs;  // (n,2): dst env row, src row
    int n;
    float* el; float* az; float* radius; float* campos; float* cam; float* alphas; float* full_reward; float* object_mass;
    int* scene_mesh; float* scene_offset; float* obs; const float* obs_all; const float* loss_all;
    int img;
};
__global__ __launch_bounds__(256) void occ_commit_kernel(CommitArgs a) {
    const int k = blockIdx.x;
    const int dst = a.pairs[2 * k], src = a.pairs[2 * k + 1];
    const int tid = threadIdx.x;
    const size_t S2 = (size_t)a.img * a.img;
    if (blockIdx.y == 0) {
        if (tid == 0) {
            a.el[dst] = a.el[src];
            a.az[dst] = a.az[src];
            a.radius[dst] = a.radius[src];
            const float l = a.loss_all[src];
            a.full_reward[dst] = l;
            a.object_mass[dst] = l + 1.0f;
        }
        if (tid < 3) {
            a.campos[dst * 3 + tid] = 0.f;
            a.scene_mesh[dst * 3 + tid] = a.scene_mesh[src * 3 + tid];
        }
        if (tid < 9) a.scene_offset[dst * 9 + tid] = a.scene_offset[src * 9 + tid];
        if (tid < OCC_CAM_STRIDE) a.cam[(size_t)dst * OCC_CAM_STRIDE + tid] = a.cam[(size_t)src * OCC_CAM_STRIDE + tid];
    } else if (blockIdx.y == 1) {
        const float4* s4 = reinterpret_cast<const float4*>(a.obs_all + (size_t)src * 4 * S2);
        float4* d4 = reinterpret_cast<float4*>(a.obs + (size_t)dst * 4 * S2);
        for (size_t i = tid; i < S2; i += 256) d4[i] = s4[i];
    } else {
        const float* s1 = a.alphas + (size_t)src * 3 * S2;
        float* d1 = a.alphas + (size_t)dst * 3 * S2;
        for (size_t i = tid; i < 3 * S2; i += 256) d1[i] = s1[i];
    }
}
