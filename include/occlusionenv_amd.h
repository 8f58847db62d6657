/*
 * occlusionenv_amd.h -- C ABI of the MI355X (gfx950) OcclusionEnv step() hot path.
 *
 * Drop-in boundary: these entry points replace, for a whole batch of N
 * environments at once, what /root/reference/environment.py does per environment through
 * PyTorch3D:
 *   - camera:   environment.py:356-368 (step), :308 (reset, look_at_view_transform),
 *               :334-335 (render)                                   -> occ_camera
 *   - renders:  silhouette_renderer(...) x3 + phong_renderer(...)   (environment.py:310,
 *               316-318, 370-372, 375) incl. MeshRasterizer.transform, clip_faces,
 *               _C.rasterize_meshes (K=100 soft / K=1 hard), SoftSilhouetteShader,
 *               HardFlatShader, observation packing (:376-378), occlusion image (:319,:373)
 *               and loss (:322,:381), plus d loss / d(elevation, azimuth) that the
 *               reference obtains from autograd + _C.rasterize_meshes_backward
 *                                                                   -> occ_render
 *   - reward bookkeeping environment.py:382-392 and the action Jacobian :356-361
 *                                                                   -> occ_step_finish
 *   - the operator-level replacement of _C.rasterize_meshes / _C.rasterize_meshes_backward (K-buffer in
 *     PyTorch3D layout, naive path)                                 -> occ_rasterize_meshes_naive,
 *                                                                      occ_rasterize_meshes_backward_dists
 *   - PyTorch3D sigmoid_alpha_blend on those K-buffers (SoftSilhouetteShader, environment.py:263)
 *                                                                   -> occ_sigmoid_alpha_blend_fwd / _bwd
 *   - SimpleVecEnv.step_wait's per-step host hand-off and auto-reset (SubProcVecEnv.py:209-218)
 *                                                                   -> occ_step_flags, occ_reset_commit,
 *                                                                      occ_auto_reset, occ_reserve_refill
 *
 * Conventions: plain pointers and sizes only; every pointer is DEVICE memory owned by the
 * caller (PyTorch's ROCm allocator in the Python host); calls are asynchronous on `stream`
 * (a hipStream_t passed as void*; NULL = default stream); nothing is allocated, freed or
 * synchronised inside; return value 0 = launched OK, nonzero = bad arguments / launch failure
 * (see OCC_ERR_*).  Per-environment device-side status words report data-dependent failures
 * (OCC_STATUS_*).
 */
#ifndef OCCLUSIONENV_AMD_H
#define OCCLUSIONENV_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCC_ABI_VERSION 9

/* return codes */
#define OCC_OK 0
#define OCC_ERR_ARG 1    /* null pointer / bad size */
#define OCC_ERR_LAUNCH 2 /* hipLaunch failed (hipGetLastError != success) */

/* per-env status bits written by the kernels (0 = fine) */
#define OCC_STATUS_LIST_OVERFLOW 1 /* internal error: a pixel of a tile whose cost class rules it out held more than K candidates */
#define OCC_STATUS_REC_OVERFLOW 2  /* more visible (clipped) faces than record capacity */

/* layout constants shared with the host */
#define OCC_CAM_STRIDE 48  /* floats per env in the camera buffer */
#define OCC_REC_STRIDE 32  /* floats per projected-face record (eight 16-byte parts = one 128-byte line) */
#define OCC_TILE 8         /* image sides must be a multiple of this */
#define OCC_BLOCK 4        /* unit of OccWorkspace.objrect: 4x4-pixel blocks (a raster work item = 2x2 blocks = one 8x8 tile) */
#ifndef OCC_LOG_CAP
#define OCC_LOG_CAP 12288  /* candidate-log entries per persistent wave (20 B each); a log about to fill up is compacted
                              in place to every overflowing pixel's K nearest.  Must be >= 64*OCC_MAX_K + 2048 + 64 */
#endif
#define OCC_LOG_ENTRY_BYTES 30 /* (depth key u32, pixel | face sequence << 6 u32) + payload (1-p, p dd/del, p dd/daz) f32 x3
                                  + 8 + 2 B of the selection's compacted copy of the entries of the pixels that hold more than K
                                  (the entry again, and its index in the log) */
#define OCC_MAX_K 128      /* largest faces_per_pixel the fused path accepts */

/* occ_camera modes */
#define OCC_CAM_STEP 0     /* environment.py:356-368: action -> el/az += 0.05*n -> C -> look_at */
#define OCC_CAM_LOOKAT 1   /* environment.py:308: look_at_view_transform(radius, el, az)        */
#define OCC_CAM_POSITION 2 /* environment.py:334-335: look_at_rotation(camera_position)         */

/* OccScene.shader */
#define OCC_SHADER_FLAT 0
#define OCC_SHADER_HARD_PHONG 1
#define OCC_SHADER_SOFT_PHONG 2

/* occ_render flags */
#define OCC_RENDER_SOFT 1  /* three soft silhouettes + occlusion image + loss */
#define OCC_RENDER_HARD 2  /* flat-shaded RGB-D observation of the joined scene */
#define OCC_RENDER_GRAD 4  /* also d loss / d(el, az) (needs OCC_CAM_STEP tangents) */

/* GPU-resident mesh pool + per-env scene description (all device pointers). */
typedef struct OccScene {
    const float* pool_verts;      /* (sumV,3) */
    const int32_t* pool_faces;    /* (sumF,3) vertex ids local to their mesh */
    const int32_t* mesh_vert_off; /* (n_meshes+1) */
    const int32_t* mesh_face_off; /* (n_meshes+1) */
    const int32_t* scene_mesh;    /* (n_env,3) pool mesh id of object 1..3 (environment.py:193) */
    const float* scene_offset;    /* (n_env,3,3) world offset of each object (environment.py:148,171) */
    int32_t n_meshes;
    int32_t n_env;
    int32_t img;       /* S: image side in pixels, multiple of 8, <= 2048 */
    int32_t rec_cap;   /* fixed layout: record capacity of every (env, object), >= 2 x faces of the largest mesh;
                          variable layout (OccWorkspace.rec_off): upper bound of any object's span */
    /* optional per-face texture atlases (PyTorch3D TexturesAtlas, environment.py:127,152,175); NULL = all white */
    const float* pool_atlas;        /* packed (sum over textured meshes of F*R*R*3) */
    const int64_t* mesh_atlas_off;  /* (n_meshes) float offset of each mesh's atlas in pool_atlas, -1 = white vertices */
    int32_t atlas_res;              /* R */
    /* optional (n_env) int32: nonzero = do not render this scene row in this launch (its outputs are left
     * untouched); used for reserve slots that are not under test (occ_auto_reset keeps it up to date) */
    const int32_t* skip;
    /* optional (n_env,S,S) f32 weight of every pixel's term of the loss: loss = sum_p w_p * full_state[p,3]^2 and
     * its gradient likewise (NULL = 1 everywhere = environment.py:381).  Images are not affected.  Used to score a
     * region of interest, and by the parity tests to leave out pixels classified as exact ties. */
    const float* pix_weight;
    /* Shader of the RGB-D observation (environment.py:281-283): OCC_SHADER_FLAT = HardFlatShader (what the reference
     * runs), OCC_SHADER_HARD_PHONG / OCC_SHADER_SOFT_PHONG = the two alternatives it keeps commented out (per-pixel
     * Phong shading with interpolated vertex normals; soft = softmax_rgb_blend with default BlendParams).  The Phong
     * shaders need pool_vnormals (sumV,3): [P3D] Meshes.verts_normals_packed() of every pool mesh. */
    int32_t shader;
    const float* pool_vnormals;
    /* Largest vertex count of any pool mesh, a sizing hint: the setup kernel stages an object's (<= 4 096) world-space
     * vertices in LDS once instead of gathering three corners per face from global memory; larger objects, or 0 here,
     * keep the global gathers.  Results are identical either way. */
    int32_t max_mesh_verts;
} OccScene;

/* Caller-allocated scratch; sizes from occ_workspace_query(). */
typedef struct OccWorkspace {
    float* rec;         /* (n_env,3,rec_cap,OCC_REC_STRIDE) projected face records */
    uint32_t* rec_bbox; /* (n_env,3,rec_cap,4) scan rows of the objects occ_sort_kernel re-sorted front to back (4 096..8 192 records); untouched otherwise (until ABI v7: a per-record bbox row, written for every record) */
    uint32_t* scan;     /* (n_env,3,rec_cap,4) per record, in face order: pixel bbox xl|yl<<16, xh|yh<<16|corner-cut bits<<28 (img <= 2048), key of its nearest vertex depth, record index (the raster scan order unless re-sorted, see rec_bbox) */
    int32_t* nrec;      /* (n_env,3) */
    int32_t* objrect;   /* (n_env,3,4) block rect bx0,by0,bx1,by1 (inclusive, OCC_BLOCK-pixel units) */
    uint32_t* queue;    /* (8,32) one work-queue head per XCD group, a 128-B line each (zeroed by occ_render) */
    float* lists;       /* (n_slots, OCC_LOG_CAP*OCC_LOG_ENTRY_BYTES) per-wave K-buffer = wave-compacted candidate log:
                           OCC_LOG_CAP payloads of 12 B, then OCC_LOG_CAP (key, tag) pairs of 8 B (structure of arrays), then
                           OCC_LOG_CAP x (8 + 2) B for the exact top-K's compacted copy of the entries it has to rank */
    float* partials;    /* (n_env,ceil(S*S/256),4) per-block loss / gradient partial sums */
    int32_t* status;    /* (n_env) OCC_STATUS_* bits, OR-ed in; caller clears */
    int32_t* offsets;   /* (8*3*ceil(n_env/8)+1) rect order only (order == NULL): first work item of every (env, object), XCD-major */
    uint32_t* rec_cbox; /* (n_env,3,ceil(rec_cap/64),4) union pixel bbox + nearest depth key of every 64-entry scan chunk */
    float* obj_alpha;   /* (n_env,3,S,S) per-object silhouette alpha, valid inside the object's tile rect */
    float* obj_grad;    /* (n_env,3,S,S,2) d alpha / d(el, az) */
    float* obj_hz;      /* (n_env,3,S,S) depth of the nearest face of the object (3e38 = none) */
    int32_t* obj_hrec;  /* (n_env,3,S,S) its record index, -1 = none */
    int32_t n_slots;    /* persistent waves = blocks occ_raster2_kernel is launched with */
    /* Optional variable record layout: rec / rec_bbox / scan hold rec_total records in all and every (env, object)
     * gets room for ITS mesh (2 x faces, rounded up to 64) at record offset rec_off[env*3+obj]; rec_off
     * (3*n_env+1 int64) is filled by occ_render.  NULL = fixed stride rec_cap per (env, object). */
    int64_t* rec_off;
    int64_t rec_total;
    /* Optional work-item ORDER of occ_raster2_kernel (NULL = rect order): the tiles of all objects sorted by an
     * estimate of their cost (faces whose pixel bbox touches the tile), heaviest first inside every XCD queue, so
     * that the longest tiles start first and the launch does not end on a few stragglers.  u32 words:
     *   [0..8] first item of every XCD queue and the total, [16 + 32 q + c] tiles of cost class c in queue q,
     *   [512 + 32 (env*3+obj) + c] where the object's class-c tiles start inside the class,
     *   then (n_env,3,T) per tile: rank inside the object's class << 5 | class   (T = (S/8)^2 tiles per image),
     *   then (one pad word if needed for 8-byte alignment and) (n_env*3*T) items
     *   (env*3+obj, tile index inside the object's rect | class << 24), 8 B each.
     * occ_render zeroes the first 512 words. */
    uint32_t* order;
} OccWorkspace;

typedef struct OccWorkspaceSizes {
    size_t rec_bytes, rec_bbox_bytes, nrec_bytes, objrect_bytes, queue_bytes, lists_bytes,
        partials_bytes, status_bytes, offsets_bytes, obj_alpha_bytes, obj_grad_bytes, obj_hz_bytes,
        obj_hrec_bytes, rec_cbox_bytes, scan_bytes;
    int32_t n_slots; /* recommended persistent-wave count for this device */
    size_t rec_off_bytes;
    size_t order_bytes;
} OccWorkspaceSizes;

/*
 * Optional tail of a STEP launch: the reward bookkeeping of occ_step_finish (environment.py:381-392) done by the very
 * launch that reduces the loss (one dependent launch less per step).  Rows [0, n_step) are stepping envs; rows beyond
 * (reserve scenes riding along) only get their loss.
 */
typedef struct OccStepFinish {
    float* full_reward;       /* (n_step) in/out */
    const float* object_mass; /* (n_step) */
    float* reward;            /* (n_step) out */
    uint8_t* done;            /* (n_step) out, 0 / 1 */
    float* grad_action;       /* (n_step,2) out or NULL; needs OccRenderOut.grad_elaz */
    int32_t n_step;
} OccStepFinish;

/* Outputs of one batched render (device pointers; any may be NULL if the flag is off). */
typedef struct OccRenderOut {
    float* obs;        /* (n_env,4,S,S)  RGB + view-space depth, -1 background (environment.py:376-378) */
    float* full_state; /* (n_env,S,S,4)  i1*i2+i2*i3+i1*i3, RGB == 3 (environment.py:373) */
    float* alphas;     /* (n_env,3,S,S)  alpha channel of the three silhouettes */
    float* loss;       /* (n_env)        sum(full_state[...,3]^2) (environment.py:381) */
    float* grad_elaz;  /* (n_env,2)      d loss / d(elevation, azimuth) */
    /*
     * Optional REGION TRACKING of persistent output buffers (an output ring, the alphas state): three quarters of a
     * 128 x 128 frame are background, and a freshly allocated output has every one of those pixels written every step.
     * rect_prev (n_env,4) int32 pixel rects x0,y0,x1,y1 (inclusive; x1 < x0 = empty): the caller GUARANTEES that obs and
     * full_state of row e hold their background values (1,1,1,-1 / 3,3,3,0) everywhere outside rect_prev[e].  The
     * launch then writes only the 256-pixel blocks that meet this step's object rects or rect_prev[e] and stores this
     * step's union rect in rect_next[e] (skipped rows: rect_next = rect_prev).  NULL = every pixel is written.
     * arect_prev / arect_next: the same for alphas (background 0).  prev and next must be different arrays.
     */
    const int32_t* rect_prev;
    int32_t* rect_next;
    const int32_t* arect_prev;
    int32_t* arect_next;
    const OccStepFinish* finish; /* host pointer, read during the call; NULL = loss / gradient only */
} OccRenderOut;

/* Optional head of a STEP launch: the camera update of occ_camera done by the launch's prologue kernel (one dependent
 * launch less per step).  Arguments as occ_camera; cam_pos_out2 (n,3): a second copy of C (the step's info["position"]
 * snapshot) or NULL. */
typedef struct OccCameraArgs {
    int32_t mode;
    const float* action;
    float* el;
    float* az;
    const float* radius;
    float* cam_pos_out;
    float* cam_pos_out2;
    int32_t n;
} OccCameraArgs;

int occ_abi_version(void);

/* number of CUs of the current device (used to size n_slots); <=0 on failure. Host-side query only. */
int occ_device_cu_count(void);

int occ_workspace_query(const OccScene* scene, int n_slots, OccWorkspaceSizes* out);

/* Sizes of the record arrays (rec, rec_bbox, scan, rec_cbox, rec_off) of a variable-layout workspace holding
 * rec_total records for n_env scenes; the other fields of *inout are left as occ_workspace_query set them. */
int occ_record_sizes(int64_t rec_total, int n_env, OccWorkspaceSizes* inout);

/*
 * Camera for N envs.  cam: (N,OCC_CAM_STRIDE) floats =
 *   [0..8] R row-major, [9..11] T, [12..14] C, [15..23] dR/d el, [24..26] dT/d el,
 *   [27..35] dR/d az, [36..38] dT/d az, [39..42] d(el,az)/d(action) row-major, [43] el, [44] az.
 * OCC_CAM_STEP:     action (N,2); el, az (N) updated IN PLACE (environment.py:360-361); radius (N).
 * OCC_CAM_LOOKAT:   el, az, radius (N) read; action ignored.
 * OCC_CAM_POSITION: action points to camera positions (N,3); el, az, radius ignored.
 * cam_pos_out (N,3) receives C when non-NULL (environment.py:363-365).
 */
int occ_camera(int mode, const float* action, float* el, float* az, const float* radius,
               float* cam, float* cam_pos_out, int n_env, void* stream);

/* Projection + z-clipping + culling + ordered face-record build, per-(env, object, tile) rasterisation
 * (soft silhouette + nearest hard face), per-pixel combine (occlusion image, shading, loss and forward-mode
 * gradient) and the per-env reduction, for all envs. */
int occ_render(const OccScene* scene, const float* cam, const OccWorkspace* ws,
               const OccRenderOut* out, int flags, int faces_per_pixel, void* stream);

/* occ_render with the step's camera update folded into its prologue launch: rows [0, camera->n) of cam are written
 * from (action, el, az, radius) first (OccCameraArgs; NULL = cam is ready, exactly occ_render).  Together with
 * OccRenderOut.finish this is the whole of OcclusionEnv.step() (environment.py:352-396) for N envs in one call. */
int occ_step(const OccScene* scene, const OccCameraArgs* camera, float* cam, const OccWorkspace* ws,
             const OccRenderOut* out, int flags, int faces_per_pixel, void* stream);

/*
 * Reward bookkeeping of step() (environment.py:381-392) for N envs:
 *   reward = (full_reward - loss)/object_mass + (5 if loss < 0.1 else -0.2); full_reward <- loss;
 *   done = loss < 0.1;  grad_action = -(1/object_mass) * J^T grad_elaz  (d reward / d action).
 */
int occ_step_finish(const float* loss, const float* grad_elaz, const float* cam,
                    float* full_reward, const float* object_mass, float* reward, uint8_t* done,
                    float* grad_action, int n_env, void* stream);

/*
 * Operator-level drop-in for PyTorch3D's `_C.rasterize_meshes` (naive path, bin_size = 0) and the `dists` part of
 * `_C.rasterize_meshes_backward` - the ops environment.py reaches through MeshRasterizer (:258-262, :276-280;
 * signatures in SURVEY.md §8b).  face_verts (F,3,3) = (x_ndc, y_ndc, z_view) AFTER clip_faces; outputs in
 * PyTorch3D's layout: pix_to_face (N,H,W,K) int64 (packed face index, -1 empty), zbuf, dists (N,H,W,K),
 * bary (N,H,W,K,3), slots ascending in (z, face).  Built for exactness (reference arithmetic order, no FMA
 * contraction), not speed: the fused occ_render never materialises these buffers.
 * grad_face_verts (F,3,3) is overwritten; the dists entry point gives x,y their gradient (zbuf / bary gradients - zero on the
 * OcclusionEnv path - come from occ_rasterize_meshes_backward below).
 */
int occ_rasterize_meshes_naive(const float* face_verts, const int64_t* mesh_to_face_first_idx,
                               const int64_t* num_faces_per_mesh, const int64_t* clipped_faces_neighbor_idx,
                               int n_meshes, int H, int W, float blur_radius, int faces_per_pixel,
                               int perspective_correct, int clip_barycentric_coords, int cull_backfaces,
                               int64_t* pix_to_face, float* zbuf, float* bary, float* dists, void* stream);
/*
 * The same K-buffers, bit for bit, from a tiled kernel: one wave per (mesh, tile) looks only at the faces whose
 * bbox +- sqrt(blur) can reach a pixel centre of the tile and keeps the (depth, face) lists in LDS (F / faces-per-tile
 * times less evaluation work than the naive kernel; the producer callers of MeshRasterizer should use).  For a mesh
 * without clipped-face pairs (clipped_faces_neighbor_idx == NULL, or -1 for every face of the mesh: found out on the
 * device, mesh by mesh) a pixel's K-buffer is the K smallest (depth, face) of its candidates whatever their arrival order:
 * 4x4-pixel tiles, four faces in flight per wave, four partial lists per pixel merged by rank at the end (round 4:
 * 10.6 -> 1.2 ms for a 5 120-face mesh at 128x128, K = 100; the teapot at 256x256 1.06 -> 0.22 ms).  A mesh with pairs:
 * 8x8 tiles, faces in order.  Same arguments; faces_per_pixel above 128 (64 KiB of lists) is served by the naive kernel.
 */
int occ_rasterize_meshes_tiled(const float* face_verts, const int64_t* mesh_to_face_first_idx,
                               const int64_t* num_faces_per_mesh, const int64_t* clipped_faces_neighbor_idx,
                               int n_meshes, int H, int W, float blur_radius, int faces_per_pixel,
                               int perspective_correct, int clip_barycentric_coords, int cull_backfaces,
                               int64_t* pix_to_face, float* zbuf, float* bary, float* dists, void* stream);
int occ_rasterize_meshes_backward_dists(const float* face_verts, const int64_t* pix_to_face, const float* grad_dists,
                                        int64_t n_faces, int n_meshes, int H, int W, int faces_per_pixel,
                                        int perspective_correct, int clip_barycentric_coords, float* grad_face_verts,
                                        void* stream);

/*
 * The whole of `_C.rasterize_meshes_backward`: grad_face_verts (F,3,3), overwritten, from any of grad_zbuf (N,H,W,K),
 * grad_bary (N,H,W,K,3), grad_dists (N,H,W,K) - NULL = that output carries no gradient.  The dists part is
 * occ_rasterize_meshes_backward_dists; zbuf / bary gradients (zero on the OcclusionEnv path) differentiate the
 * area-normalised, perspective-corrected, clipped barycentrics and the depth they interpolate.
 */
int occ_rasterize_meshes_backward(const float* face_verts, const int64_t* pix_to_face, const float* grad_zbuf,
                                  const float* grad_bary, const float* grad_dists, int64_t n_faces, int n_meshes, int H, int W,
                                  int faces_per_pixel, int perspective_correct, int clip_barycentric_coords,
                                  float* grad_face_verts, void* stream);

/*
 * [P3D] sigmoid_alpha_blend (SoftSilhouetteShader, environment.py:263) on K-buffers in PyTorch3D layout:
 *   alpha[p] = 1 - prod_k (1 - sigmoid(-dists[p,k] / sigma) * [pix_to_face[p,k] >= 0]),   images (n_pix,4) RGBA, RGB = 1.
 * Backward: grad_dists[p,k] from grad_images (only the alpha channel carries gradient).
 */
int occ_sigmoid_alpha_blend_fwd(const float* dists, const int64_t* pix_to_face, int64_t n_pix, int faces_per_pixel,
                                float sigma, float* images, void* stream);
int occ_sigmoid_alpha_blend_bwd(const float* dists, const int64_t* pix_to_face, const float* grad_images, int64_t n_pix,
                                int faces_per_pixel, float sigma, float* grad_dists, void* stream);

/*
 * The K epochs of PPO.update (PPO.py:196-217) for the heads-only learner of the vectorised env: the reference optimises
 * action_head (256 -> 2) and value_head (256 -> 1) only (PPO.py:113-116; model.py:153-154,168-171 detaches the features),
 * with the fixed diagonal Gaussian of ActorCritic (PPO.py:62-104) and torch.optim.Adam.  ONE launch per epoch: forward,
 * clipped-surrogate loss, backward and the Adam step over the 771 parameters; parameters, moments and step count are
 * updated in place.  feats (M,256), actions (M,2), old_logprob (M), returns (M) [already normalised, PPO.py:187-188];
 * losses (n_epochs,2) receives (total loss, value loss) of every epoch; scratch: OCC_PPO_SCRATCH_FLOATS floats;
 * counter: one zeroed uint32 (left at zero).  Adam moments m, v: 771 floats each in the order W_a | b_a | W_v | b_v.
 */
#define OCC_PPO_FEATURES 256
#define OCC_PPO_PARAMS (3 * OCC_PPO_FEATURES + 3)
#ifndef OCC_PPO_MAX_BLOCKS
#define OCC_PPO_MAX_BLOCKS 128
#endif
#define OCC_PPO_SCRATCH_FLOATS (OCC_PPO_MAX_BLOCKS * (OCC_PPO_PARAMS + 2))
typedef struct OccPpoState {
    float *w_a, *b_a, *w_v, *b_v; /* action_head.weight (2,256), .bias (2), value_head.weight (1,256), .bias (1) */
    float *adam_m, *adam_v;       /* OCC_PPO_PARAMS each */
    float* adam_step;             /* (1) step count as float (torch's capturable Adam keeps it on the device too) */
} OccPpoState;
/* The block cap this library was built with (OCC_PPO_MAX_BLOCKS): scratch must hold occ_ppo_max_blocks() * (OCC_PPO_PARAMS + 2) floats. */
int occ_ppo_max_blocks(void);
int occ_ppo_update(const float* feats, const float* actions, const float* old_logprob, const float* returns, int64_t M,
                   float action_var, float eps_clip, float lr_actor, float lr_critic, float beta1, float beta2,
                   float adam_eps, const OccPpoState* state, int n_epochs, float* losses, float* scratch,
                   uint32_t* counter, void* stream);

/*
 * The rollout record's 256 features (PPO.py:155-158 stores the frozen encoder's pooled feature of the acting state; the
 * network itself is outside this path): 4 channels x 8 x 8 average pooling of the observation, obs (n,4,img,img) ->
 * feats (n,256) in the order of adaptive_avg_pool2d(obs, 8).reshape(n, 256).  img: a multiple of 8.
 */
int occ_pool8(const float* obs, int64_t n, int img, float* feats, void* stream);

/*
 * Host hand-off of SimpleVecEnv.step_wait (SubProcVecEnv.py:209-218): one int32 buffer
 *   flags[0..n_env) = done, flags[n_env..n_env+n_reserve) = reserve scene passes reset()'s acceptance test
 *   (loss > 0.1, environment.py:327), flags[n_env+n_reserve] = some status word is non-zero
 * so that the host needs ONE device-to-host copy per batched step.  status has n_env+n_reserve words.
 */
int occ_step_flags(const uint8_t* done, const float* loss_all, const int32_t* status, int n_env, int n_reserve,
                   int32_t* flags, void* stream);

/*
 * Auto-reset commit ("obs = self.envs[env_idx].reset()", SubProcVecEnv.py:214, from a pre-rendered candidate):
 * for k < n, env row pairs[2k] takes over row pairs[2k+1] of every per-env state array (el, az, radius, cam,
 * alphas, scene) with camera_position = 0, full_reward = loss, object_mass = loss + 1 (environment.py:302-324),
 * and obs[dst] = obs_all[src].  All arrays hold n_env + n_reserve rows except campos/full_reward/object_mass/obs
 * (n_env rows).
 */
int occ_reset_commit(const int32_t* pairs, int n, float* el, float* az, float* radius, float* campos, float* cam,
                     float* alphas, float* full_reward, float* object_mass, int32_t* scene_mesh, float* scene_offset,
                     float* obs, const float* obs_all, const float* loss_all, int img, void* stream);

/*
 * Device-side auto-reset (the whole of "if buf_done: obs = self.envs[env_idx].reset()", SubProcVecEnv.py:209-218,
 * incl. reset()'s rejection loop environment.py:288-327) for a batched step whose launch also rendered n_reserve
 * speculative reset scenes (rows n_env .. n_env+n_reserve-1 of every *_all array, reset camera).
 *
 * Reserve slot life cycle (rs_state): OCC_RS_EMPTY -- host uploads a scene (occ_reserve_refill) --> OCC_RS_PENDING
 * -- rendered by a step; loss > 0.1 or 10th try --> OCC_RS_READY (else back to EMPTY, try count kept) -- taken by
 * a finished env --> EMPTY (tries = 0).  Only the host leaves EMPTY, only the device leaves PENDING / READY.
 *
 * One call = two launches (three until ABI v8): (1) a single block ages the PENDING slots, lists finished envs and READY
 * slots in index order, pairs them and refreshes the skip mask (rendered next step = PENDING only); (2) per slot: the rows
 * this step rendered for a slot that was PENDING and is not taken now (observation, full_state, loss) are copied into the
 * persistent OccReserveStore - READY slots are not rendered again (OccScene.skip), their last render stays valid - and
 * every pair copies the slot's last render (this step's rows, or the store) and state into the env's rows (as
 * occ_reset_commit), saving the env's final observation to term_obs[slot] first (info["terminal_observation"]).
 * report (n_env + 2*n_reserve + 2 int32): [0,n_env) 1 = done, 2 = time limit (OccAutoResetOpts) | [n_env, +n_reserve) slot state AFTER the call |
 * [.., +n_reserve) env that took the slot this call or -1 | any status bit | finished envs left without a slot.
 * pairs: scratch, 2 + 3*n_reserve int32.  Arrays of OccEnvState hold n_env + n_reserve rows except
 * campos / full_reward / object_mass (n_env rows); obs_all has n_env + n_reserve rows.
 */
#define OCC_RS_EMPTY 0
#define OCC_RS_PENDING 1
#define OCC_RS_READY 2

typedef struct OccEnvState {
    float* el;
    float* az;
    float* radius;
    float* campos;
    float* cam;
    float* alphas;
    float* full_reward;
    float* object_mass;
    int32_t* scene_mesh;
    float* scene_offset;
} OccEnvState;

/* Persistent copy of what the last render of every reserve slot produced (a slot is only rendered while PENDING). */
typedef struct OccReserveStore {
    float* obs;        /* (n_reserve,4,S,S) */
    float* full_state; /* (n_reserve,S,S,4) */
    float* loss;       /* (n_reserve) */
    int32_t* skip;     /* (n_env + n_reserve) the OccScene.skip mask; rows >= n_env maintained here */
} OccReserveStore;

/* Optional extras of occ_auto_reset (NULL = none of them). */
typedef struct OccAutoResetOpts {
    /* Episode time limit (trainRL.py:22,191-229: `for t in range(1, max_ep_len + 1)` then env.reset(), is_terminal stays
     * False): age (n_env) = steps since the env's last reset, incremented by this call; an env whose age reaches
     * max_ep_len (> 0) is reset from the reserve like a finished one, with done left 0 (report value 2); a committed
     * env's age returns to 0.  age == NULL: no counting. */
    int32_t* age;
    int32_t max_ep_len;
    /* Region-tracking rects of obs_all and of the alphas state (OccRenderOut.rect_next / arect_next of the launch that
     * produced this step): a committed row holds a whole stored frame and is marked full-frame. */
    int32_t* rect;
    int32_t* arect;
    /* (n_reserve,S,S,4): receives the stored occlusion image of every slot taken in this call (what the reference's
     * env.image holds after reset(), environment.py:319). */
    float* reset_full_state;
    /* normWithObjectSize (environment.py:208,320,324): norm_flags (n_env) int32, nonzero = that env divides its reward by
     * objectMass = sum_px (a1 + a2 + a3)^2 + 1 of its reset render instead of loss + 1; slot_objsum (n_reserve) holds that
     * sum for the stored render of every reserve slot (occ_object_mass).  Both NULL: loss + 1 for every env. */
    const int32_t* norm_flags;
    const float* slot_objsum;
    /* A second copy of `report`, written by the pairing launch itself into PINNED HOST memory that the device can address
     * (hipHostMalloc / torch pin_memory): the host reads it after an event recorded behind this call - no copy launch.  Its
     * first n_env words must be ZERO on entry: only the entries of envs that are reset (1 / 2) are written. */
    int32_t* report_host;
} OccAutoResetOpts;

int occ_auto_reset(const uint8_t* done, const float* loss_all, const int32_t* status, int n_env, int n_reserve,
                   int32_t* rs_state, int32_t* rs_tries, const OccEnvState* st, float* obs_all, const float* full_state_all,
                   const OccReserveStore* store, float* term_obs, int img, int32_t* pairs, int32_t* report,
                   const OccAutoResetOpts* opts, void* stream);

/*
 * objectMass of reset() with normWithObjectSize (environment.py:320,324: self.objects = image1 + image2 + image3;
 * objectMass = sum(objects[..., 3] ** 2) + 1): out[r] = sum over the pixels of (a1 + a2 + a3)^2 for rows r of
 * alphas (n_rows,3,S,S), in a fixed summation order.  gate (n_rows) int32 or NULL: only rows with gate[r] == gate_value
 * are computed, the others keep what out holds (the reserve: rows rendered in this step are the OCC_RS_PENDING ones).
 */
int occ_object_mass(const float* alphas, int n_rows, int img, const int32_t* gate, int gate_value, float* out, void* stream);

/*
 * Host -> reserve: n packed rows of 13 words (slot, mesh id x3, offset x9 as float bits) in device memory or in
 * pinned host memory the device can address (then no copy launch is needed: the rows are read when the kernel runs, so
 * the host must leave them alone until it has synchronised with something later on the stream); scatters them into scene_mesh / scene_offset rows n_env + slot and marks the slots OCC_RS_PENDING.
 */
int occ_reserve_refill(const int32_t* packed, int n, int n_env, int n_reserve, int32_t* scene_mesh, float* scene_offset,
                       int32_t* rs_state, int32_t* skip, void* stream);

/*
 * Measurement hooks (bench.py only; not part of the reference surface).  While enabled, occ_render
 * brackets its dominant kernel (occ_raster2_kernel) with HIP events on the launch stream.
 * occ_profile_read synchronises the recorded events (host sync!), returns the summed duration in
 * milliseconds and the number of launches since the last read -- counting only the launches with the
 * largest n_env seen (full batches; auto-resets render tiny ones) -- and resets the ring (max 4096 launches).
 */
int occ_profile_enable(int on);
int occ_profile_read(double* ms_sum, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* OCCLUSIONENV_AMD_H */
