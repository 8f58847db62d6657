// occ_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the batched OcclusionEnv step().
//
// What the reference does per environment through PyTorch3D (4 renders, ~200 autograd nodes,
// /root/reference/environment.py:352-396) is done here for N environments in a fixed handful of launches:
//
//   occ_camera_kernel   one thread per env : action -> (el, az) -> C -> look_at R,T, carrying
//                       forward-mode tangents d/d(el,az) of R and T (environment.py:356-368)
//   occ_setup_kernel    one block per (env, object): gather pool verts, world->view->NDC
//                       (MeshRasterizer.transform), z-clip (clip_faces), cull, per-face record
//                       with NDC verts + tangents + invariants, ORDERED compaction (face order is
//                       PyTorch3D's tie-break order), packed tile bbox, object tile rect
//   occ_order_kernel    work items (env, object, 8x8-pixel tile) in cost order, heaviest first per XCD queue
//                       (occ_scan_kernel: plain rect order when the workspace has no order buffer)
//   occ_raster2_kernel  persistent wave64 per work item: lane = (face, pixel) PAIR over the pixels of every face's
//                       pixel bbox inside the tile (on-the-fly binning by ballot over chunk boxes and pixel bboxes,
//                       records staged in LDS by cooperative 16-B loads): soft silhouette (K nearest-z sigmoid
//                       product) + hard nearest face of ONE object in one sweep, per-pixel state in LDS, candidate
//                       K-buffer = wave-compacted log streamed to HBM/L2 in full lines, exact top-K-by-z selection
//                       (cooperative LDS-histogram radix select) when a pixel has more than K candidates
//   occ_combine_kernel  one thread per pixel: occlusion image of the three silhouettes, loss and
//                       d loss/d(el,az) partials, nearest-object pick + flat shading, outputs
//   occ_reduce_kernel   one wave per env: fixed-order sum of the per-block partials
//   occ_finish_kernel   reward bookkeeping + action Jacobian (environment.py:381-392,356-361)
//
// The gradient is carried in FORWARD mode (two tangent directions, el and az) through the very
// same sweep that renders: d alpha/d theta = -(A/sigma) * sum_k p_k * d dist_k/d theta needs no
// second pass over the K-buffer, no atomics into grad_face_verts and no saved fragments.  It
// equals what autograd + _C.rasterize_meshes_backward produce.
//
// No MFMA: the path is rasterisation.  fp32 throughout.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "occ_constants.h"
#include "occlusionenv_amd.h"

namespace occ {

#include "occ_common.hpp"
#include "occ_camera.hpp"
#include "occ_setup.hpp"
#include "occ_eval.hpp"
#include "occ_raster2.hpp"
#include "occ_combine.hpp"
#include "occ_blend.hpp"
#include "occ_reset.hpp"
#include "occ_oplevel.hpp"
#include "occ_ppo.hpp"

}  // namespace occ

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
using namespace occ;

extern "C" int occ_abi_version(void) { return OCC_ABI_VERSION; }

// ---- measurement hooks ---------------------------------------------------------------------
namespace {
constexpr int kProfMax = 4096;
bool g_prof_on = false;
int g_prof_n = 0;
hipEvent_t g_prof_ev[2 * kProfMax];
int g_prof_nenv[kProfMax];
bool g_prof_created = false;
}  // namespace

extern "C" int occ_profile_enable(int on) {
    if (on && !g_prof_created) {
        for (int i = 0; i < 2 * kProfMax; ++i)
            if (hipEventCreate(&g_prof_ev[i]) != hipSuccess) return OCC_ERR_LAUNCH;
        g_prof_created = true;
    }
    g_prof_on = on != 0;
    g_prof_n = 0;
    return OCC_OK;
}

extern "C" int occ_profile_read(double* ms_sum, int* launches) {
    if (!ms_sum || !launches) return OCC_ERR_ARG;
    double tot = 0.0;
    int big = 0, cnt = 0;
    for (int i = 0; i < g_prof_n; ++i) big = g_prof_nenv[i] > big ? g_prof_nenv[i] : big;
    for (int i = 0; i < g_prof_n; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof_ev[2 * i + 1]) != hipSuccess) return OCC_ERR_LAUNCH;
        if (hipEventElapsedTime(&ms, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]) != hipSuccess) return OCC_ERR_LAUNCH;
        if (g_prof_nenv[i] != big) continue;  // full-batch launches only (auto-resets render tiny batches)
        tot += ms;
        cnt += 1;
    }
    *ms_sum = tot;
    *launches = cnt;
    g_prof_n = 0;
    return OCC_OK;
}

#ifdef OCC_DBG_STATS
extern "C" int occ_debug_stats(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(occ::g_dbg_stats), 32 * sizeof(unsigned long long)) != hipSuccess) return 2;  // (32 slots)
    unsigned long long z[32] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(occ::g_dbg_stats), z, sizeof(z)) == hipSuccess ? 0 : 2;
}
#endif

#ifdef OCC_DBG_TIME
extern "C" int occ_debug_time(unsigned long long* out112) {
    if (hipMemcpyFromSymbol(out112, HIP_SYMBOL(occ::g_dbg_time), 112 * sizeof(unsigned long long)) != hipSuccess) return 2;
    unsigned long long z[112] = {0};
    z[16] = ~0ull;
    for (int i = 96; i < 104; ++i) z[i] = ~0ull;
    return hipMemcpyToSymbol(HIP_SYMBOL(occ::g_dbg_time), z, sizeof(z)) == hipSuccess ? 0 : 2;
}
#endif

#ifdef OCC_DBG_ENDS
extern "C" int occ_debug_ends(unsigned long long* out16384) {
    return hipMemcpyFromSymbol(out16384, HIP_SYMBOL(occ::g_dbg_ends), 4 * 4096 * sizeof(unsigned long long)) == hipSuccess ? 0 : 2;
}
#endif

#ifdef OCC_DBG_BOUNDS
extern "C" int occ_debug_fault(int* out8) {
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(occ::g_dbg_fault), 8 * sizeof(int)) == hipSuccess ? 0 : 2;
}
#endif

extern "C" int occ_device_cu_count(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    return prop.multiProcessorCount;
}

static bool scene_ok(const OccScene* s) {
    return s && s->pool_verts && s->pool_faces && s->mesh_vert_off && s->mesh_face_off && s->scene_mesh &&
           s->scene_offset && s->n_env > 0 && s->img >= OCC_TILE && s->img % OCC_TILE == 0 && s->img <= 2048 &&
           s->rec_cap > 0 && s->shader >= OCC_SHADER_FLAT && s->shader <= OCC_SHADER_SOFT_PHONG &&
           (s->shader == OCC_SHADER_FLAT || s->pool_vnormals);
}

extern "C" int occ_workspace_query(const OccScene* scene, int n_slots, OccWorkspaceSizes* out) {
    if (!scene || !out || scene->n_env <= 0 || scene->img < OCC_TILE || scene->img % OCC_TILE || scene->rec_cap <= 0)
        return OCC_ERR_ARG;
    if (n_slots <= 0) {
        const int cus = occ_device_cu_count();
        n_slots = (cus > 0 ? cus : 256) * 8;
    }
    const size_t N = (size_t)scene->n_env, cap = (size_t)scene->rec_cap;
    out->rec_bytes = N * 3 * cap * OCC_REC_STRIDE * sizeof(float);
    out->rec_bbox_bytes = N * 3 * cap * 4 * sizeof(uint32_t);
    out->scan_bytes = N * 3 * cap * 4 * sizeof(uint32_t);
    out->rec_cbox_bytes = N * 3 * ((cap + 63) / 64) * 4 * sizeof(uint32_t);
    out->nrec_bytes = N * 3 * sizeof(int32_t);
    out->objrect_bytes = N * 3 * 4 * sizeof(int32_t);
    out->queue_bytes = 8 * kQueueStride * sizeof(uint32_t);  // eight queue heads, one 128-B line each
    out->lists_bytes = (size_t)n_slots * OCC_LOG_BYTES;  // per-wave K-buffer: the compacted candidate log
    const size_t S2 = (size_t)scene->img * scene->img;
    out->partials_bytes = N * ((S2 + 255) / 256) * 4 * sizeof(float);
    out->offsets_bytes = (size_t)(8 * xcd_slots(scene->n_env) + 1) * sizeof(int32_t);
    out->obj_alpha_bytes = N * 3 * S2 * sizeof(float);
    out->obj_grad_bytes = N * 3 * S2 * 2 * sizeof(float);
    out->obj_hz_bytes = N * 3 * S2 * sizeof(float);
    out->obj_hrec_bytes = N * 3 * S2 * sizeof(int32_t);
    out->status_bytes = N * sizeof(int32_t);
    out->n_slots = n_slots;
    out->rec_off_bytes = (N * 3 + 1) * sizeof(int64_t);
    {   // work-item order: header + per-object class offsets + per-tile (rank, class) + the item list
        const size_t T = (size_t)(scene->img / 8) * (scene->img / 8);
        out->order_bytes = (ord_items_word(scene->n_env, scene->img) + N * 3 * T * 2) * sizeof(uint32_t);
    }
    return OCC_OK;
}

extern "C" int occ_record_sizes(int64_t rec_total, int n_env, OccWorkspaceSizes* io) {
    if (!io || rec_total <= 0 || n_env <= 0 || (rec_total & 63)) return OCC_ERR_ARG;
    const size_t T = (size_t)rec_total;
    io->rec_bytes = T * OCC_REC_STRIDE * sizeof(float);
    io->rec_bbox_bytes = T * 4 * sizeof(uint32_t);
    io->scan_bytes = T * 4 * sizeof(uint32_t);
    io->rec_cbox_bytes = (T / 64) * 4 * sizeof(uint32_t);
    io->rec_off_bytes = ((size_t)n_env * 3 + 1) * sizeof(int64_t);
    return OCC_OK;
}

extern "C" int occ_camera(int mode, const float* action, float* el, float* az, const float* radius, float* cam,
                          float* cam_pos_out, int n_env, void* stream) {
    if (!cam || n_env <= 0) return OCC_ERR_ARG;
    if (mode == OCC_CAM_STEP && (!action || !el || !az || !radius)) return OCC_ERR_ARG;
    if (mode == OCC_CAM_LOOKAT && (!el || !az || !radius)) return OCC_ERR_ARG;
    if (mode == OCC_CAM_POSITION && !action) return OCC_ERR_ARG;
    if (mode < 0 || mode > 2) return OCC_ERR_ARG;
    hipLaunchKernelGGL(occ_camera_kernel, dim3((n_env + 63) / 64), dim3(64), 0, (hipStream_t)stream, mode, action, el, az,
                       radius, cam, cam_pos_out, n_env);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

// OCC_DEBUG_SYNC=1 in the environment: every launch of occ_render is announced on stderr and waited for, so that a
// faulting kernel is the last one named (diagnostics only; serialises the stream).  The only getenv of the library.
static bool dbg_sync_on() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("OCC_DEBUG_SYNC");
        on = (e && e[0] && e[0] != '0') ? 1 : 0;
    }
    return on == 1;
}
#define OCC_DBG_SYNC(name)                                                          \
    do {                                                                            \
        if (dbg_sync_on()) {                                                        \
            fprintf(stderr, "[occ] %s launched (n_env %d) ...", name, scene->n_env); \
            fflush(stderr);                                                         \
            const hipError_t e_ = hipStreamSynchronize(st);                         \
            fprintf(stderr, " %s\n", e_ == hipSuccess ? "ok" : hipGetErrorString(e_)); \
            fflush(stderr);                                                         \
        }                                                                           \
    } while (0)

extern "C" int occ_render(const OccScene* scene, const float* cam, const OccWorkspace* ws, const OccRenderOut* out,
                          int flags, int faces_per_pixel, void* stream) {
    return occ_step(scene, nullptr, const_cast<float*>(cam), ws, out, flags, faces_per_pixel, stream);
}

extern "C" int occ_step(const OccScene* scene, const OccCameraArgs* camera, float* cam, const OccWorkspace* ws,
                        const OccRenderOut* out, int flags, int faces_per_pixel, void* stream) {
    if (!scene_ok(scene) || !cam || !ws || !out) return OCC_ERR_ARG;
    OccCameraArgs ca{};
    if (camera) {
        ca = *camera;
        if (ca.n <= 0 || ca.n > scene->n_env || ca.mode < 0 || ca.mode > 2) return OCC_ERR_ARG;
        if (ca.mode == OCC_CAM_STEP && (!ca.action || !ca.el || !ca.az || !ca.radius)) return OCC_ERR_ARG;
        if (ca.mode == OCC_CAM_LOOKAT && (!ca.el || !ca.az || !ca.radius)) return OCC_ERR_ARG;
        if (ca.mode == OCC_CAM_POSITION && !ca.action) return OCC_ERR_ARG;
    }
    if ((out->rect_prev != nullptr) != (out->rect_next != nullptr) || (out->rect_prev && out->rect_prev == out->rect_next)) return OCC_ERR_ARG;
    if ((out->arect_prev != nullptr) != (out->arect_next != nullptr) || (out->arect_prev && out->arect_prev == out->arect_next)) return OCC_ERR_ARG;
    FinishArgs fin{};
    if (out->finish) {
        const OccStepFinish& f = *out->finish;
        if (!(flags & OCC_RENDER_SOFT) || !out->loss || !f.full_reward || !f.object_mass || !f.reward || !f.done || f.n_step <= 0 ||
            f.n_step > scene->n_env || (f.grad_action && (flags & OCC_RENDER_GRAD) && !out->grad_elaz))
            return OCC_ERR_ARG;
        fin = FinishArgs{f.full_reward, f.object_mass, f.reward, f.done, f.grad_action, cam, f.n_step};
    }
    if (!ws->rec || !ws->rec_bbox || !ws->nrec || !ws->objrect || !ws->queue || !ws->lists || !ws->partials ||
        !ws->status || !ws->rec_cbox || !ws->scan || !ws->offsets || !ws->obj_alpha || !ws->obj_grad || !ws->obj_hz || !ws->obj_hrec ||
        ws->n_slots <= 0)
        return OCC_ERR_ARG;
    const bool soft = flags & OCC_RENDER_SOFT, hard = flags & OCC_RENDER_HARD, grad = flags & OCC_RENDER_GRAD;
    if (!soft && !hard) return OCC_ERR_ARG;
    if (grad && !soft) return OCC_ERR_ARG;
    if (hard && !out->obs) return OCC_ERR_ARG;
    if (soft && (faces_per_pixel <= 0 || faces_per_pixel > OCC_MAX_K)) return OCC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int N = scene->n_env;
    const OccWorkspace& wsv = *ws;
    if (ws->rec_off && (ws->rec_total <= 0 || (ws->rec_total & 63))) return OCC_ERR_ARG;
    static_assert(kOrdBlk <= 1024, "occ_recoff_kernel zeroes the order header with one block");
    // prologue: zero the queue heads and the order header, lay out the variable record spans
    hipLaunchKernelGGL(occ_recoff_kernel, dim3(1 + (ca.n + 1023) / 1024), dim3(1024), 0, st, *scene, (long long*)ws->rec_off,
                       (long long)ws->rec_total, ws->status, ws->queue, wsv.order, ca, cam);
    OCC_DBG_SYNC("recoff");
    // world-space vertices of an object staged in LDS when they fit (OccScene.max_mesh_verts; 0 = gather from global memory)
    int vcap = scene->max_mesh_verts > 0 ? ((scene->max_mesh_verts + 63) & ~63) : 0;
    if (vcap > kSetupVcapMax) vcap = kSetupVcapMax;
    {   // the staged vertices come on top of the kernel's static LDS: never ask for more than the device gives a block
        static int lds_room = -1;  // bytes of dynamic LDS occ_setup_kernel can have (queried once)
        if (lds_room < 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            hipFuncAttributes fa;
            lds_room = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&occ_setup_kernel<true, kSetupTB>)) == hipSuccess &&
                prop.sharedMemPerBlock > fa.sharedSizeBytes)
                lds_room = (int)(prop.sharedMemPerBlock - fa.sharedSizeBytes);
        }
        const int fit = (lds_room / 12) & ~63;
        if (vcap > fit) vcap = fit;  // (0: gather from global memory, same results)
    }
#ifdef OCC_EXP_SETUP_DUMMY_LDS  // occupancy experiment only: the LDS is reserved but the vertices are still gathered from memory
    size_t vlds = (size_t)vcap * 3 * sizeof(float);
    vcap = 0;
#else
    const size_t vlds = (size_t)vcap * 3 * sizeof(float);
#endif
    if (grad)
        hipLaunchKernelGGL((occ_setup_kernel<true, kSetupTB>), dim3(N * 3), dim3(kSetupTB), vlds, st, *scene, cam, wsv, vcap);
    else
        hipLaunchKernelGGL((occ_setup_kernel<false, kSetupTB>), dim3(N * 3), dim3(kSetupTB), vlds, st, *scene, cam, wsv, vcap);
    OCC_DBG_SYNC("setup");
    const bool may_sort = scene->rec_cap >= kSortMin;  // dense objects only: front-to-back scan order (LDS sort buffer: 8192 keys = 64 KiB)
    const int sort_cap = kSortCap;
    const size_t sort_lds = may_sort ? (size_t)sort_cap * sizeof(unsigned long long) : 0;
    if (may_sort && !wsv.order) {
        hipLaunchKernelGGL(occ_sort_kernel, dim3(N * 3), dim3(256), sort_lds, st, *scene, *ws, sort_cap);
        OCC_DBG_SYNC("sort");
    }
    if (hipGetLastError() != hipSuccess) return OCC_ERR_LAUNCH;
    RasterParams P;
    P.sc = *scene;
    P.ws = wsv;
    P.out = *out;
    P.cam = cam;
    P.K = faces_per_pixel;
    P.ntx = scene->img / OCC_TILE;
    {   // small launches: split every tile into row groups until the expected work items fill about half the grid
        // (an object's rect covers a few per cent of the image: 0.04 x tiles is the typical count at radius 4)
        const double est = (double)N * 3.0 * P.ntx * P.ntx * 0.04;
        int sl = 0;
        while (sl < 3 && est * (double)(2 << sl) <= (double)ws->n_slots) ++sl;
        P.split_log2 = sl;
    }
    if (wsv.order) {
        // (One launch for sort + order - occ_sort_order_kernel - was measured in round 5: 18.9 us against 5 + 5 for the two:
        // the sort's 64 KB of LDS lets two blocks share a CU, and the order pass then runs at that occupancy.  It is used
        // where no object can be dense - nothing to sort, no LDS.)
        if (may_sort) {
            hipLaunchKernelGGL(occ_sort_kernel, dim3(N * 3), dim3(256), sort_lds, st, *scene, *ws, sort_cap);
            OCC_DBG_SYNC("sort");
            hipLaunchKernelGGL(occ_order_kernel, dim3(N * 3), dim3(64), 0, st, ws->objrect, ws->nrec, wsv.order, N, scene->img);
        } else {
            hipLaunchKernelGGL(occ_sort_order_kernel, dim3(N * 3), dim3(64), 0, st, *scene, wsv, 0);
        }
        OCC_DBG_SYNC("order");
    } else {
        hipLaunchKernelGGL(occ_scan_kernel, dim3(1), dim3(1024), 0, st, ws->objrect, ws->nrec, ws->offsets, N, 1);
        OCC_DBG_SYNC("scan");
    }
    if (hipGetLastError() != hipSuccess) return OCC_ERR_LAUNCH;
    const dim3 grid(ws->n_slots), block(64);
    const bool prof = g_prof_on && g_prof_n < kProfMax;
    if (prof) (void)hipEventRecord(g_prof_ev[2 * g_prof_n], st);
    const int bpe = (scene->img * scene->img + 255) / 256;
    // thread blocks per env of the combine kernel: enough of them for a small launch, few enough that a big one does
    // not start tens of thousands of blocks that find nothing to do (occ_combine_kernel)
    int cG = bpe / 16 > (4096 + N - 1) / N ? bpe / 16 : (4096 + N - 1) / N;
    if (cG > bpe) cG = bpe;
    if (cG < 1) cG = 1;
    const dim3 cgrid(N * cG), cblock(256);
#define OCC_LAUNCH(SOFT_, HARD_, GRAD_)                                                             \
    do {                                                                                            \
        hipLaunchKernelGGL((occ_raster2_kernel<SOFT_, HARD_, GRAD_>), grid, block, 0, st, P);       \
        OCC_DBG_SYNC("raster");                                                                     \
        if (prof) {                                                                                 \
            (void)hipEventRecord(g_prof_ev[2 * g_prof_n + 1], st);                                  \
            g_prof_nenv[g_prof_n] = N;                                                              \
            g_prof_n += 1;                                                                          \
        }                                                                                           \
        hipLaunchKernelGGL((occ_combine_kernel<SOFT_, HARD_, GRAD_>), cgrid, cblock, 0, st, P, bpe, cG); \
        OCC_DBG_SYNC("combine");                                                                    \
    } while (0)
    if (soft && hard && grad)
        OCC_LAUNCH(true, true, true);
    else if (soft && hard)
        OCC_LAUNCH(true, true, false);
    else if (soft && grad)
        OCC_LAUNCH(true, false, true);
    else if (soft)
        OCC_LAUNCH(true, false, false);
    else
        OCC_LAUNCH(false, true, false);
#undef OCC_LAUNCH
    if (hipGetLastError() != hipSuccess) return OCC_ERR_LAUNCH;
    if (soft && (out->loss || out->grad_elaz)) {
        hipLaunchKernelGGL(occ_reduce_kernel, dim3(N), dim3(64), 0, st, ws->partials, bpe, out->loss,
                           grad ? out->grad_elaz : nullptr, scene->skip, fin, wsv, scene->img);
        if (hipGetLastError() != hipSuccess) return OCC_ERR_LAUNCH;
    }
    return OCC_OK;
}

extern "C" int occ_rasterize_meshes_naive(const float* face_verts, const int64_t* mesh_to_face_first_idx,
                                          const int64_t* num_faces_per_mesh, const int64_t* clipped_faces_neighbor_idx,
                                          int n_meshes, int H, int W, float blur_radius, int faces_per_pixel,
                                          int perspective_correct, int clip_barycentric_coords, int cull_backfaces,
                                          int64_t* pix_to_face, float* zbuf, float* bary, float* dists, void* stream) {
    if (!face_verts || !mesh_to_face_first_idx || !num_faces_per_mesh || !pix_to_face || !zbuf || !bary || !dists ||
        n_meshes <= 0 || H <= 0 || W <= 0 || faces_per_pixel <= 0 || blur_radius < 0.f)
        return OCC_ERR_ARG;
    KbufArgs a{face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx, n_meshes, H, W,
               faces_per_pixel, blur_radius, perspective_correct, clip_barycentric_coords, cull_backfaces, pix_to_face,
               zbuf, bary, dists};
    const long npix = (long)n_meshes * H * W;
    hipLaunchKernelGGL(occ_rast_naive_fwd_kernel, dim3((unsigned)((npix + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_rasterize_meshes_tiled(const float* face_verts, const int64_t* mesh_to_face_first_idx,
                                          const int64_t* num_faces_per_mesh, const int64_t* clipped_faces_neighbor_idx,
                                          int n_meshes, int H, int W, float blur_radius, int faces_per_pixel,
                                          int perspective_correct, int clip_barycentric_coords, int cull_backfaces,
                                          int64_t* pix_to_face, float* zbuf, float* bary, float* dists, void* stream) {
    if (!face_verts || !mesh_to_face_first_idx || !num_faces_per_mesh || !pix_to_face || !zbuf || !bary || !dists ||
        n_meshes <= 0 || H <= 0 || W <= 0 || faces_per_pixel <= 0 || blur_radius < 0.f)
        return OCC_ERR_ARG;
    const size_t lds = (size_t)64 * faces_per_pixel * 8;
    if (lds > 64 * 1024)  // beyond the (depth, face) lists the kernel keeps in LDS: same results from the naive kernel
        return occ_rasterize_meshes_naive(face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx,
                                          n_meshes, H, W, blur_radius, faces_per_pixel, perspective_correct,
                                          clip_barycentric_coords, cull_backfaces, pix_to_face, zbuf, bary, dists, stream);
    KbufArgs a{face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx, n_meshes, H, W,
               faces_per_pixel, blur_radius, perspective_correct, clip_barycentric_coords, cull_backfaces, pix_to_face,
               zbuf, bary, dists};
    // meshes without clipped-face pairs (no neighbour array, or all -1 for the mesh): the K-buffer does not depend on the
    // arrival order - 4x4 tiles, four faces in flight; meshes with pairs: 8x8 tiles, faces in order.  With an array both
    // kernels are launched and every wave finds out first which of the two its mesh belongs to.
    const int qx = (W + 3) / 4, qy = (H + 3) / 4;
    hipLaunchKernelGGL(occ_rast_quad_fwd_kernel, dim3((unsigned)(n_meshes * qx * qy)), dim3(64), lds, (hipStream_t)stream, a, qx, qy);
    if (clipped_faces_neighbor_idx) {
        const int tiles_x = (W + 7) / 8, tiles_y = (H + 7) / 8;
        hipLaunchKernelGGL(occ_rast_tiled_fwd_kernel, dim3((unsigned)(n_meshes * tiles_x * tiles_y)), dim3(64), lds,
                           (hipStream_t)stream, a, tiles_x, tiles_y, 1);
    }
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_rasterize_meshes_backward_dists(const float* face_verts, const int64_t* pix_to_face,
                                                   const float* grad_dists, int64_t n_faces, int n_meshes, int H, int W,
                                                   int faces_per_pixel, int perspective_correct,
                                                   int clip_barycentric_coords, float* grad_face_verts, void* stream) {
    if (!face_verts || !pix_to_face || !grad_dists || !grad_face_verts || n_faces <= 0 || n_meshes <= 0 || H <= 0 ||
        W <= 0 || faces_per_pixel <= 0)
        return OCC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(grad_face_verts, 0, (size_t)n_faces * 9 * sizeof(float), st) != hipSuccess) return OCC_ERR_LAUNCH;
    const long tot = (long)n_meshes * H * W * faces_per_pixel;
    hipLaunchKernelGGL(occ_rast_naive_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, face_verts,
                       pix_to_face, grad_dists, n_meshes, H, W, faces_per_pixel, perspective_correct,
                       clip_barycentric_coords, grad_face_verts);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_rasterize_meshes_backward(const float* face_verts, const int64_t* pix_to_face, const float* grad_zbuf,
                                             const float* grad_bary, const float* grad_dists, int64_t n_faces, int n_meshes,
                                             int H, int W, int faces_per_pixel, int perspective_correct,
                                             int clip_barycentric_coords, float* grad_face_verts, void* stream) {
    if (!face_verts || !pix_to_face || !grad_face_verts || n_faces <= 0 || n_meshes <= 0 || H <= 0 || W <= 0 ||
        faces_per_pixel <= 0)
        return OCC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(grad_face_verts, 0, (size_t)n_faces * 9 * sizeof(float), st) != hipSuccess) return OCC_ERR_LAUNCH;
    const long tot = (long)n_meshes * H * W * faces_per_pixel;
    const dim3 grid((unsigned)((tot + 255) / 256)), block(256);
    if (grad_dists)
        hipLaunchKernelGGL(occ_rast_naive_bwd_kernel, grid, block, 0, st, face_verts, pix_to_face, grad_dists, n_meshes, H, W,
                           faces_per_pixel, perspective_correct, clip_barycentric_coords, grad_face_verts);
    if (grad_zbuf || grad_bary)
        hipLaunchKernelGGL(occ_rast_bwd_zbary_kernel, grid, block, 0, st, face_verts, pix_to_face, grad_zbuf, grad_bary,
                           n_meshes, H, W, faces_per_pixel, perspective_correct, clip_barycentric_coords, grad_face_verts);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_sigmoid_alpha_blend_fwd(const float* dists, const int64_t* pix_to_face, int64_t n_pix, int faces_per_pixel,
                                           float sigma, float* images, void* stream) {
    if (!dists || !pix_to_face || !images || n_pix <= 0 || faces_per_pixel <= 0 || !(sigma > 0.f)) return OCC_ERR_ARG;
    hipLaunchKernelGGL(occ_blend_fwd_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dists,
                       pix_to_face, (long)n_pix, faces_per_pixel, sigma, images);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_sigmoid_alpha_blend_bwd(const float* dists, const int64_t* pix_to_face, const float* grad_images,
                                           int64_t n_pix, int faces_per_pixel, float sigma, float* grad_dists, void* stream) {
    if (!dists || !pix_to_face || !grad_images || !grad_dists || n_pix <= 0 || faces_per_pixel <= 0 || !(sigma > 0.f))
        return OCC_ERR_ARG;
    hipLaunchKernelGGL(occ_blend_bwd_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dists,
                       pix_to_face, grad_images, (long)n_pix, faces_per_pixel, sigma, grad_dists);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_ppo_max_blocks(void) { return OCC_PPO_MAX_BLOCKS; }

extern "C" int occ_ppo_update(const float* feats, const float* actions, const float* old_logprob, const float* returns,
                              int64_t M, float action_var, float eps_clip, float lr_actor, float lr_critic, float beta1,
                              float beta2, float adam_eps, const OccPpoState* state, int n_epochs, float* losses,
                              float* scratch, uint32_t* counter, void* stream) {
    static_assert(kPpoFeat == OCC_PPO_FEATURES && kPpoParams == OCC_PPO_PARAMS, "occ_ppo.hpp and the header disagree");
    if (!feats || !actions || !old_logprob || !returns || M <= 0 || !(action_var > 0.f) || !state || n_epochs < 0 || !losses ||
        !scratch || !counter || !state->w_a || !state->b_a || !state->w_v || !state->b_v || !state->adam_m || !state->adam_v ||
        !state->adam_step)
        return OCC_ERR_ARG;
    PpoArgs A;
    A.feats = feats; A.actions = actions; A.old_lp = old_logprob; A.returns = returns; A.M = (long long)M;
    A.inv_var = 1.0f / action_var;
    // MultivariateNormal(mean, var I) in two dimensions: log-density constant and entropy (PPO.py:62-104)
    const float log2pi = 1.8378770664093453f, logdet = 2.0f * logf(action_var);
    A.lp_const = -0.5f * (2.0f * log2pi + logdet);
    A.ent_term = 0.01f * 0.5f * (2.0f * (1.0f + log2pi) + logdet);
    A.eps_clip = eps_clip;
    A.lr_actor = lr_actor; A.lr_critic = lr_critic; A.beta1 = beta1; A.beta2 = beta2; A.adam_eps = adam_eps;
    A.w_a = state->w_a; A.b_a = state->b_a; A.w_v = state->w_v; A.b_v = state->b_v;
    A.m = state->adam_m; A.v = state->adam_v; A.step = state->adam_step;
    A.partials = scratch; A.counter = counter;
    // A wave takes ~12 samples of a 10^4-sample rollout: 64 blocks of 16 waves, few rows for the last block's sum.  The
    // replicated learner of an 8-GPU node sees 8 x the samples: from 2^16 samples on, twice the blocks (measured, us per
    // epoch at 12 800 / 51 200 / 102 400 samples: 64 blocks 40 / 67 / 106, 128 blocks 49 / 69 / 88, 256 blocks 50 / 103 / 115:
    // more rows cost the last block more than the extra waves save until the sample loop dominates).
    const long long cap = M >= 65536 ? OCC_PPO_MAX_BLOCKS : (OCC_PPO_MAX_BLOCKS < 64 ? OCC_PPO_MAX_BLOCKS : 64);
    const long long want = (M + 16 * 8 - 1) / (16 * 8);
    const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > cap ? cap : want));
    for (int e = 0; e < n_epochs; ++e) {
        A.losses = losses + 2 * e;
        hipLaunchKernelGGL(occ_ppo_epoch_kernel, dim3(blocks), dim3(kPpoThreads), 0, (hipStream_t)stream, A);
    }
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_pool8(const float* obs, int64_t n, int img, float* feats, void* stream) {
    if (!obs || !feats || n <= 0 || img < 8 || img % 8) return OCC_ERR_ARG;
    const dim3 grid((unsigned)(n * 4 * 8));
    if ((img / 8) % 4 == 0)
        hipLaunchKernelGGL(occ_pool8_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, obs, feats, img);
    else
        hipLaunchKernelGGL(occ_pool8_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, obs, feats, img);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_step_flags(const uint8_t* done, const float* loss_all, const int32_t* status, int n_env, int n_reserve,
                              int32_t* flags, void* stream) {
    if (!done || !status || !flags || n_env <= 0 || n_reserve < 0 || (n_reserve > 0 && !loss_all)) return OCC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int tot = n_env + n_reserve;
    hipLaunchKernelGGL(occ_flags_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, done, loss_all, status, n_env, n_reserve,
                       flags);
    hipLaunchKernelGGL(occ_status_any_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, status, tot, flags + tot);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_reset_commit(const int32_t* pairs, int n, float* el, float* az, float* radius, float* campos, float* cam,
                                float* alphas, float* full_reward, float* object_mass, int32_t* scene_mesh,
                                float* scene_offset, float* obs, const float* obs_all, const float* loss_all, int img,
                                void* stream) {
    if (n == 0) return OCC_OK;
    if (!pairs || n < 0 || !el || !az || !radius || !campos || !cam || !alphas || !full_reward || !object_mass ||
        !scene_mesh || !scene_offset || !obs || !obs_all || !loss_all || img <= 0)
        return OCC_ERR_ARG;
    CommitArgs a{pairs, n, el, az, radius, campos, cam, alphas, full_reward, object_mass, scene_mesh, scene_offset,
                 obs, obs_all, loss_all, img};
    hipLaunchKernelGGL(occ_commit_kernel, dim3(n, 3), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_auto_reset(const uint8_t* done, const float* loss_all, const int32_t* status, int n_env, int n_reserve,
                              int32_t* rs_state, int32_t* rs_tries, const OccEnvState* st, float* obs_all,
                              const float* full_state_all, const OccReserveStore* store, float* term_obs, int img,
                              int32_t* pairs, int32_t* report, const OccAutoResetOpts* opts, void* stream) {
    if (!done || !loss_all || !status || n_env <= 0 || n_reserve <= 0 || n_reserve > 512 || !rs_state || !rs_tries || !st ||
        !obs_all || !full_state_all || !store || !term_obs || img < OCC_TILE || img % OCC_TILE || !pairs || !report)
        return OCC_ERR_ARG;
    if (!st->el || !st->az || !st->radius || !st->campos || !st->cam || !st->alphas || !st->full_reward || !st->object_mass ||
        !st->scene_mesh || !st->scene_offset || !store->obs || !store->full_state || !store->loss || !store->skip)
        return OCC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    OccAutoResetOpts o{};
    if (opts) o = *opts;
    if (o.max_ep_len > 0 && !o.age) return OCC_ERR_ARG;
    if (o.norm_flags && !o.slot_objsum) return OCC_ERR_ARG;
    int32_t* was_pending = pairs + 2 + 2 * n_reserve;  // third section of the scratch
    PairArgs pa{done, loss_all, status, n_env, n_reserve, rs_state, rs_tries, pairs, report, store->skip, o.age, o.max_ep_len,
                was_pending, o.report_host};
    hipLaunchKernelGGL(occ_pair_kernel, dim3(1), dim3(1024), 0, s, pa);
    StashCommitArgs ca{was_pending, report + n_env + n_reserve, *st, obs_all, full_state_all, loss_all, *store, term_obs, img, n_env,
                       o.age, o.rect, o.arect, o.reset_full_state, o.norm_flags, o.slot_objsum};
    hipLaunchKernelGGL(occ_stash_commit_kernel, dim3(n_reserve, 1 + commit_obs_blocks(img) + commit_alpha_blocks(img)), dim3(256), 0, s, ca);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_object_mass(const float* alphas, int n_rows, int img, const int32_t* gate, int gate_value, float* out,
                               void* stream) {
    if (!alphas || !out || n_rows <= 0 || img <= 0) return OCC_ERR_ARG;
    hipLaunchKernelGGL(occ_object_mass_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, alphas, img, gate, gate_value, out);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_reserve_refill(const int32_t* packed, int n, int n_env, int n_reserve, int32_t* scene_mesh,
                                  float* scene_offset, int32_t* rs_state, int32_t* skip, void* stream) {
    if (n == 0) return OCC_OK;
    if (!packed || n < 0 || n_env <= 0 || n_reserve <= 0 || !scene_mesh || !scene_offset || !rs_state || !skip) return OCC_ERR_ARG;
    hipLaunchKernelGGL(occ_refill_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, packed, n, n_env, n_reserve, scene_mesh,
                       scene_offset, rs_state, skip);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_step_finish(const float* loss, const float* grad_elaz, const float* cam, float* full_reward,
                               const float* object_mass, float* reward, uint8_t* done, float* grad_action, int n_env,
                               void* stream) {
    if (!loss || !cam || !full_reward || !object_mass || !reward || !done || n_env <= 0) return OCC_ERR_ARG;
    hipLaunchKernelGGL(occ_finish_kernel, dim3((n_env + 63) / 64), dim3(64), 0, (hipStream_t)stream, loss, grad_elaz, cam,
                       full_reward, object_mass, reward, done, grad_action, n_env);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}
