// occ_reset.hpp -- device-side auto-reset: stash, pairing, commit, refill.
// Part of the single translation unit occ_kernels.hip (included inside namespace occ; not a stand-alone header).

// ------------------------------------------------------------------------------------------
// device-side auto-reset: pairing (one block) + commit (one block group per pair)
// ------------------------------------------------------------------------------------------
struct PairArgs {
    const uint8_t* done; const float* loss_all; const int* status;
    int n_env, n_res;
    int* rs_state; int* rs_tries; int* pairs; int* report; int* skip;
    int* age; int max_ep_len;  // episode time limit (trainRL.py:22,191-229): age (n_env) counts the steps since the last reset
    int* was_pending;          // (n_res) out: the slot was under test, i.e. rendered by this step's launch (occ_stash_commit_kernel)
    int* report_host;          // second copy of the report in pinned host memory (no copy launch), or null
};

// ordered compaction helper: exclusive prefix of flag over a 1024-thread block (16 waves)
__device__ __forceinline__ int block_prefix_1024(bool flag, int* s_w, int tid, int& total) {
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned long long m = __ballot(flag);
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    __syncthreads();  // s_w reuse
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int v = s_w[w];
        if (w < wave) off += v;
        tot += v;
    }
    total = tot;
    return off + __popcll(m & lt);
}

__global__ __launch_bounds__(1024) void occ_pair_kernel(PairArgs a) {
    __shared__ int s_w[16];
    __shared__ int s_fin[512], s_ready[512];
    __shared__ int s_any;
    const int tid = threadIdx.x;
    const int N = a.n_env, R = a.n_res;
    if (tid == 0) s_any = 0;
    // (1) age the PENDING slots: every slot was rendered by this step's launch with its current scene
    int st = OCC_RS_EMPTY;
    if (tid < R) {
        st = a.rs_state[tid];
        a.was_pending[tid] = st == OCC_RS_PENDING ? 1 : 0;
        if (st == OCC_RS_PENDING) {
            const int t = a.rs_tries[tid] + 1;
            // accept, or keep the 10th try regardless (environment.py:288,327)
            st = (a.loss_all[N + tid] > kDoneThreshold || t >= 10) ? OCC_RS_READY : OCC_RS_EMPTY;
            a.rs_tries[tid] = t;
        }
        a.report[N + R + tid] = -1;
        if (a.report_host) a.report_host[N + R + tid] = -1;
    }
    int nready;
    const int rpos = block_prefix_1024(tid < R && st == OCC_RS_READY, s_w, tid, nready);
    if (tid < R && st == OCC_RS_READY) s_ready[rpos] = tid;
    // (2) finished envs in index order (the first 512 are kept: no more slots than that exist)
    int nfin = 0, any = 0;
    for (int base = 0; base < N; base += 1024) {
        const int i = base + tid;
        const bool fin = i < N && a.done[i] != 0;
        // an env that has run max_ep_len steps since its reset is reset like a finished one, but NOT done (the reference's
        // loop leaves `for t in range(1, max_ep_len + 1)` and calls env.reset(); is_terminal stays False)
        bool expired = false;
        if (i < N && a.age) {
            const int ag = a.age[i] + 1;
            a.age[i] = ag;
            expired = a.max_ep_len > 0 && ag >= a.max_ep_len;
        }
        const bool f = fin || expired;
        if (i < N) {
            a.report[i] = fin ? 1 : (expired ? 2 : 0);
            // (the host hands over its copy with this section zeroed: only the few envs that were reset are written across the bus)
            if (a.report_host && f) a.report_host[i] = fin ? 1 : 2;
            any |= a.status[i];
        }
        int tot;
        const int pos = nfin + block_prefix_1024(f, s_w, tid, tot);
        if (f && pos < 512) s_fin[pos] = i;
        nfin += tot;
    }
    if (tid < R) any |= a.status[N + tid];
    if (any) atomicOr(&s_any, 1);
    __syncthreads();
    // (3) pair them
    const int npair = min(min(nfin, nready), R);
    if (tid < npair) {
        const int r = s_ready[tid], i = s_fin[tid];
        a.pairs[2 + 2 * tid] = i;
        a.pairs[3 + 2 * tid] = N + r;
        a.report[N + R + r] = i;
        if (a.report_host) a.report_host[N + R + r] = i;
    }
    // a READY slot that was taken goes back to EMPTY with a fresh try count
    const bool taken = tid < R && st == OCC_RS_READY && rpos < npair;
    if (tid < R) {
        if (taken) {
            st = OCC_RS_EMPTY;
            a.rs_tries[tid] = 0;
        }
        a.rs_state[tid] = st;
        a.report[N + tid] = st;
        if (a.report_host) a.report_host[N + tid] = st;
        a.skip[N + tid] = (st != OCC_RS_PENDING) ? 1 : 0;  // only slots under test are rendered by the next step
    }
    if (tid == 0) {
        a.pairs[0] = npair;
        a.report[N + 2 * R] = s_any;
        a.report[N + 2 * R + 1] = nfin - npair;
        if (a.report_host) {
            a.report_host[N + 2 * R] = s_any;
            a.report_host[N + 2 * R + 1] = nfin - npair;
        }
    }
    // (the host reads report_host after the event that follows this call's last launch: the kernel boundary makes the stores visible)
}

// One launch after the pairing (round 5; until then occ_stash_kernel BEFORE and occ_auto_commit_kernel AFTER it), one block
// group per reserve slot r:
//   * the slot was under test in this step (was_pending) and nobody takes it: its rendered rows (observation, occlusion
//     image, loss) go to the persistent store - a READY slot is not rendered again;
//   * an env takes the slot (slot_env[r] >= 0): the env's final observation goes to term_obs[r]
//     (info["terminal_observation"]), the slot's render - straight from this step's rows if it was rendered just now,
//     else from the store - and state become the env's (environment.py:302-324), the stored occlusion image goes to
//     reset_fs[r].
struct StashCommitArgs {
    const int* was_pending;  // (n_res)
    const int* slot_env;     // (n_res) env that takes the slot in this call or -1 (the report's third section)
    OccEnvState st;
    float* obs_all; const float* fs_all; const float* loss_all;
    OccReserveStore store;
    float* term_obs;
    int img, n_env;
    int* age;                  // (n_env) or null: zeroed for the env that takes a slot
    int* rect; int* arect;     // region-tracking rects of obs_all / the alphas state or null: the committed row is a full frame
    float* reset_fs;           // (n_res,S,S,4) or null
    const int* norm_flags; const float* slot_objsum;  // normWithObjectSize (environment.py:324) or null
};
// grid.y = 1 (state) + obs_blocks + alpha_blocks: the copies are sized by the image - a block moves ~4 x 256 float4 per
// plane it touches
__host__ __device__ inline int commit_obs_blocks(int img) { return max(1, (img * img) / 256); }   // one float4 per thread and array
__host__ __device__ inline int commit_alpha_blocks(int img) { return max(1, (3 * img * img) / 1024); }
__global__ __launch_bounds__(256) void occ_stash_commit_kernel(StashCommitArgs a) {
    const int r = blockIdx.x;
    const bool pend = a.was_pending[r] != 0;
    const int dst = a.slot_env[r];
    if (!pend && dst < 0) return;
    const int src = a.n_env + r;
    const int tid = threadIdx.x, y = blockIdx.y;
    const size_t S2 = (size_t)a.img * a.img;
    const int obs_blocks = commit_obs_blocks(a.img), alpha_blocks = commit_alpha_blocks(a.img);
    if (y == 0) {
        const float l = pend ? a.loss_all[src] : a.store.loss[r];
        if (dst < 0) {
            if (tid == 0) a.store.loss[r] = l;
            return;
        }
        if (tid == 0) {
            a.st.el[dst] = a.st.el[src];
            a.st.az[dst] = a.st.az[src];
            a.st.radius[dst] = a.st.radius[src];
            a.st.full_reward[dst] = l;
            // environment.py:324: objectMass = sum(objects^2) + 1 if normWithObjectSize else loss + 1
            const bool norm = a.norm_flags && a.norm_flags[dst] != 0;
            a.st.object_mass[dst] = (norm ? a.slot_objsum[r] : l) + 1.0f;
        }
        if (tid < 3) {
            a.st.campos[dst * 3 + tid] = 0.f;
            a.st.scene_mesh[dst * 3 + tid] = a.st.scene_mesh[src * 3 + tid];
        }
        if (tid < 9) a.st.scene_offset[dst * 9 + tid] = a.st.scene_offset[src * 9 + tid];
        if (tid < OCC_CAM_STRIDE) a.st.cam[(size_t)dst * OCC_CAM_STRIDE + tid] = a.st.cam[(size_t)src * OCC_CAM_STRIDE + tid];
        if (tid == 0 && a.age) a.age[dst] = 0;
        if (tid < 4) {  // the row now holds a whole stored frame, not "background outside this step's footprint"
            const int full = tid < 2 ? 0 : a.img - 1;
            if (a.rect) a.rect[dst * 4 + tid] = full;
            if (a.arect) a.arect[dst * 4 + tid] = full;
        }
    } else if (y <= obs_blocks) {
        // what the slot's last render produced: this step's rows if it was rendered just now, else the store
        const float4* o4 = reinterpret_cast<const float4*>(pend ? a.obs_all + (size_t)src * 4 * S2 : a.store.obs + (size_t)r * 4 * S2);
        const float4* f4 = reinterpret_cast<const float4*>(pend ? a.fs_all + (size_t)src * 4 * S2 : a.store.full_state + (size_t)r * 4 * S2);
        if (dst < 0) {  // stash
            float4* od = reinterpret_cast<float4*>(a.store.obs + (size_t)r * 4 * S2);
            float4* fd = reinterpret_cast<float4*>(a.store.full_state + (size_t)r * 4 * S2);
            for (size_t i = (size_t)(y - 1) * 256 + tid; i < S2; i += (size_t)obs_blocks * 256) {
                od[i] = o4[i];
                fd[i] = f4[i];
            }
        } else {  // commit: final observation -> term_obs[slot], reset observation -> obs[env] (same element range, same thread)
            float4* d4 = reinterpret_cast<float4*>(a.obs_all + (size_t)dst * 4 * S2);
            float4* t4 = reinterpret_cast<float4*>(a.term_obs + (size_t)r * 4 * S2);
            float4* g4 = a.reset_fs ? reinterpret_cast<float4*>(a.reset_fs + (size_t)r * 4 * S2) : nullptr;
            for (size_t i = (size_t)(y - 1) * 256 + tid; i < S2; i += (size_t)obs_blocks * 256) {
                t4[i] = d4[i];
                d4[i] = o4[i];
                if (g4) g4[i] = f4[i];
            }
        }
    } else if (dst >= 0) {
        // (img is a multiple of 8: the three alpha planes are a whole number of float4)
        const float4* s4 = reinterpret_cast<const float4*>(a.st.alphas + (size_t)src * 3 * S2);
        float4* d4 = reinterpret_cast<float4*>(a.st.alphas + (size_t)dst * 3 * S2);
        for (size_t i = (size_t)(y - 1 - obs_blocks) * 256 + tid; i < 3 * S2 / 4; i += (size_t)alpha_blocks * 256) d4[i] = s4[i];
    }
}

// sum_px (a1 + a2 + a3)^2 of one row of alphas (3,S,S) per block: thread partials in pixel order, DPP wave sums, the four
// waves' sums added in wave order (environment.py:320,324)
__global__ __launch_bounds__(256) void occ_object_mass_kernel(const float* __restrict__ alphas, int img, const int* __restrict__ gate,
                                                              int gate_value, float* __restrict__ out) {
    __shared__ float s_w[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    if (gate && gate[r] != gate_value) return;
    const size_t S2 = (size_t)img * img;
    const float* a = alphas + (size_t)r * 3 * S2;
    float acc = 0.f;
    for (size_t i = tid; i < S2; i += 256) {
        const float o = a[i] + a[S2 + i] + a[2 * S2 + i];
        acc += o * o;
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) s_w[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) out[r] = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
}

__global__ __launch_bounds__(64) void occ_refill_kernel(const int* __restrict__ packed, int n, int n_env, int n_res,
                                                        int* __restrict__ scene_mesh, float* __restrict__ scene_offset,
                                                        int* __restrict__ rs_state, int* __restrict__ skip) {
    const int k = blockIdx.x, tid = threadIdx.x;
    if (k >= n) return;
    const int* row = packed + 13 * k;
    const int slot = row[0];
    if (slot < 0 || slot >= n_res) return;
    const int e = n_env + slot;
    if (tid < 3) scene_mesh[e * 3 + tid] = row[1 + tid];
    if (tid < 9) scene_offset[e * 9 + tid] = __int_as_float(row[4 + tid]);
    if (tid == 0) {
        rs_state[slot] = OCC_RS_PENDING;
        skip[e] = 0;  // rendered from the next launch on
    }
}
