// occ_ppo.hpp -- one PPO epoch of the heads-only learner as ONE kernel (config 5: the update of the rollout loop).
// Part of the single translation unit occ_kernels.hip (included inside namespace occ; not a stand-alone header).
//
// The reference optimises the ACTION and VALUE heads only (PPO.py:113-116; FullNetwork.act detaches the features,
// model.py:168-171): per epoch (PPO.py:196-217) two 256 -> 2 / 256 -> 1 linear layers over the M = T * N samples of the
// rollout, the clipped surrogate, its backward and an Adam step over 771 parameters.  As framework ops that is ~35
// launches of a few microseconds each per epoch (80 epochs: 35 ms per update even as replays of a captured graph, a fifth
// of config 5's step time); as arithmetic it is one pass over 13 MB of features.  Here: every wave takes samples in a
// strided loop - the lane holds 4 of the 256 features (one 16-byte load per sample, the three weight rows in registers),
// three wave reductions give (mean_0, mean_1, value), every lane then forms the sample's loss terms and output
// gradients and adds its share of the weight gradients in registers.  Blocks write their partial sums to scratch; the
// LAST block to arrive (device-scope counter) adds the partials in block order - a fixed summation order, results are
// reproducible - applies Adam and writes the epoch's two loss values.  The next epoch is the next launch on the stream.
//
// Formulas (fixed diagonal Gaussian of ActorCritic, PPO.py:62-104; loss PPO.py:199-212; torch.optim.Adam):
//   lp = -0.5 sum_j (a_j - mean_j)^2 / var - 0.5 (2 log 2 pi + 2 log var);  ratio = exp(lp - lp_old);  adv = ret - value
//   loss = mean(-min(ratio adv, clamp(ratio, 1 - c, 1 + c) adv)) + 0.5 mean((value - ret)^2) - 0.01 entropy
//   d loss / d mean_j = -[unclipped or surr1 < surr2] adv ratio (a_j - mean_j) / (var M)     (torch.min splits a tie evenly
//   between two equal branches with equal derivatives);  d loss / d value = (value - ret) / M;  the entropy is constant.
//   Adam: m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2; p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).

constexpr int kPpoFeat = 256;                      // features per sample (FullNetwork's pooled encoder output)
constexpr int kPpoParams = 3 * kPpoFeat + 3;       // W_a (2,256) | b_a (2) | W_v (1,256) | b_v (1): the layout of OccPpoState
constexpr int kPpoPartial = kPpoParams + 2;        // + (sum of -min(surr1, surr2), sum of (value - ret)^2)
constexpr int kPpoThreads = 1024;                 // 16 waves per block: few blocks = few rows for the last block's sum
constexpr int kPpoRow = (kPpoPartial + 3) & ~3;  // LDS row of a wave's sums: 16-byte aligned for the float4 stores

struct PpoArgs {
    const float* feats;     // (M, 256)
    const float* actions;   // (M, 2)
    const float* old_lp;    // (M)
    const float* returns;   // (M) normalised returns
    long long M;
    float inv_var, lp_const, eps_clip, ent_term;  // 1 / action variance; -0.5 (2 log 2 pi + log det); clip; 0.01 * entropy
    float lr_actor, lr_critic, beta1, beta2, adam_eps;
    float* w_a;  float* b_a;  float* w_v;  float* b_v;       // parameters (updated in place)
    float* m;    float* v;                                   // Adam moments, kPpoParams each, parameter layout above
    float* step;                                             // (1) Adam step count (float, like torch's capturable Adam)
    float* partials;                                         // (gridDim.x, kPpoPartial) scratch
    unsigned int* counter;                                   // (1) zero before the first launch; left at zero
    float* losses;                                           // (2) this epoch's (loss, value loss)
};

__global__ __launch_bounds__(kPpoThreads) void occ_ppo_epoch_kernel(PpoArgs A) {
    __shared__ __attribute__((aligned(16))) float s_part[kPpoThreads / 64][kPpoRow];
    __shared__ bool s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4 wa0 = reinterpret_cast<const float4*>(A.w_a)[lane];
    const float4 wa1 = reinterpret_cast<const float4*>(A.w_a + kPpoFeat)[lane];
    const float4 wv = reinterpret_cast<const float4*>(A.w_v)[lane];
    const float ba0 = A.b_a[0], ba1 = A.b_a[1], bv = A.b_v[0];
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, gv = g0;
    float gb0 = 0.f, gb1 = 0.f, gbv = 0.f, lsum = 0.f, vsum = 0.f;
    const float invM = 1.0f / (float)A.M;
    const long long stride = (long long)gridDim.x * (kPpoThreads / 64);
    // One sample = one 16-byte load per lane and a chain of three wave reductions: a wave that waits for each sample's
    // load before it asks for the next runs at memory LATENCY (0.9 TB/s over 102 400 samples, the 8-GPU learner's load).
    // The loads of kPpoAhead samples are issued together; the samples are then consumed in the same order as before, so
    // every sum is formed in the same order (results unchanged).
    auto consume = [&](const long long i, const float4 f) {
        float d0 = f.x * wa0.x + f.y * wa0.y + f.z * wa0.z + f.w * wa0.w;
        float d1 = f.x * wa1.x + f.y * wa1.y + f.z * wa1.z + f.w * wa1.w;
        float dv = f.x * wv.x + f.y * wv.y + f.z * wv.z + f.w * wv.w;
        // (three DPP wave sums: as xor butterflies - eighteen dependent ds_bpermute round trips per sample - the
        // reductions were the kernel: 0.8 us per sample and wave whatever the grid, 9.0 ms per 80-epoch update of the
        // 102 400 samples an 8-GPU config-5 learner sees)
        d0 = wave_sum_dpp(d0);
        d1 = wave_sum_dpp(d1);
        dv = wave_sum_dpp(dv);
        const float mean0 = d0 + ba0, mean1 = d1 + ba1, value = dv + bv;
        const float2 act = reinterpret_cast<const float2*>(A.actions)[i];
        const float e0 = act.x - mean0, e1 = act.y - mean1;
        const float lp = -0.5f * (e0 * e0 + e1 * e1) * A.inv_var + A.lp_const;  // log N(a; mean, var I)
        const float ret = A.returns[i];
        const float ratio = __expf(lp - A.old_lp[i]);
        const float adv = ret - value;
        const float lo = 1.0f - A.eps_clip, hi = 1.0f + A.eps_clip;
        const float clipped = fminf(fmaxf(ratio, lo), hi);
        const float s1 = ratio * adv, s2 = clipped * adv;
        const bool through = (ratio >= lo && ratio <= hi) || s1 < s2;  // the branch of min() that depends on the policy
        const float dlp = through ? -adv * ratio * invM : 0.0f;        // d loss / d lp
        const float gm0 = dlp * e0 * A.inv_var, gm1 = dlp * e1 * A.inv_var, gval = (value - ret) * invM;
        g0.x += gm0 * f.x; g0.y += gm0 * f.y; g0.z += gm0 * f.z; g0.w += gm0 * f.w;
        g1.x += gm1 * f.x; g1.y += gm1 * f.y; g1.z += gm1 * f.z; g1.w += gm1 * f.w;
        gv.x += gval * f.x; gv.y += gval * f.y; gv.z += gval * f.z; gv.w += gval * f.w;
        gb0 += gm0; gb1 += gm1; gbv += gval;  // (identical in every lane: lane 0's copy is used)
        lsum += -fminf(s1, s2);
        vsum += (value - ret) * (value - ret);
    };
    constexpr int kPpoAhead = 4;
    for (long long i = (long long)blockIdx.x * (kPpoThreads / 64) + wave; i < A.M; i += stride * kPpoAhead) {
        float4 f[kPpoAhead];
#pragma unroll
        for (int u = 0; u < kPpoAhead; ++u) {
            const long long iu = i + u * stride;
            f[u] = iu < A.M ? reinterpret_cast<const float4*>(A.feats + iu * kPpoFeat)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < kPpoAhead; ++u) {
            const long long iu = i + u * stride;
            if (iu < A.M) consume(iu, f[u]);
        }
    }
    // waves -> block (fixed order) -> this block's row of the scratch
    float* sp = s_part[wave];
    reinterpret_cast<float4*>(sp)[lane] = g0;
    reinterpret_cast<float4*>(sp + kPpoFeat)[lane] = g1;
    if (lane == 0) {
        sp[2 * kPpoFeat] = gb0;
        sp[2 * kPpoFeat + 1] = gb1;
        sp[3 * kPpoFeat + 2] = gbv;
        sp[kPpoParams] = lsum;
        sp[kPpoParams + 1] = vsum;
    }
    // (W_v starts at 2 * 256 + 2: not 16-byte aligned in this layout - scalar stores)
    sp[2 * kPpoFeat + 2 + 4 * lane] = gv.x;
    sp[2 * kPpoFeat + 2 + 4 * lane + 1] = gv.y;
    sp[2 * kPpoFeat + 2 + 4 * lane + 2] = gv.z;
    sp[2 * kPpoFeat + 2 + 4 * lane + 3] = gv.w;
    __syncthreads();
    float* mine = A.partials + (size_t)blockIdx.x * kPpoPartial;
    for (int k = tid; k < kPpoPartial; k += kPpoThreads) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < kPpoThreads / 64; ++w) a += s_part[w][k];
        mine[k] = a;
    }
    // last block to arrive reduces the rows and takes the optimiser step
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(A.counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const float t = A.step[0] + 1.0f;
    const float bc1 = 1.0f - powf(A.beta1, t), bc2s = sqrtf(1.0f - powf(A.beta2, t));
    __shared__ float s_loss[2];
    for (int k = tid; k < kPpoPartial; k += kPpoThreads) {
        float g = 0.f;
#pragma unroll 16
        for (unsigned b = 0; b < gridDim.x; ++b) g += A.partials[(size_t)b * kPpoPartial + k];  // (<= OCC_PPO_MAX_BLOCKS rows, sixteen loads in flight together)
        if (k >= kPpoParams) {
            s_loss[k - kPpoParams] = g;
            continue;
        }
        float* p;
        float lr;
        if (k < 2 * kPpoFeat) { p = A.w_a + k; lr = A.lr_actor; }
        else if (k < 2 * kPpoFeat + 2) { p = A.b_a + (k - 2 * kPpoFeat); lr = A.lr_actor; }
        else if (k < 3 * kPpoFeat + 2) { p = A.w_v + (k - 2 * kPpoFeat - 2); lr = A.lr_critic; }
        else { p = A.b_v; lr = A.lr_critic; }
        const float mk = A.beta1 * A.m[k] + (1.0f - A.beta1) * g;
        const float vk = A.beta2 * A.v[k] + (1.0f - A.beta2) * g * g;
        A.m[k] = mk;
        A.v[k] = vk;
        *p -= (lr / bc1) * mk / (sqrtf(vk) / bc2s + A.adam_eps);
    }
    __syncthreads();
    if (tid == 0) {
        const float vloss = s_loss[1] * invM;
        A.losses[0] = s_loss[0] * invM + 0.5f * vloss - A.ent_term;
        A.losses[1] = vloss;
        A.step[0] = t;
        *A.counter = 0u;
    }
}

// ------------------------------------------------------------------------------------------
// 4 channels x 8 x 8 average pooling of the observation: the stand-in for the frozen encoder's 256-d pooled feature
// (PPO.py:155-157; the network itself is out of scope).  obs (n,4,S,S), S a multiple of 8 -> feats (n,256) in the order
// of adaptive_avg_pool2d(obs, 8).reshape(n, 256).  One block per (env, channel, row of 8 cells): every thread sums its
// columns over the S/8 rows of the band (row-contiguous, 16-byte loads when a cell is a multiple of 4 pixels wide) into
// LDS, eight threads add the column sums of their cell in column order (no atomics: reproducible).  One pass over the
// observation at memory speed (the framework's generic pooling kernel took 0.22 ms of config 5's step for 268 MB).
// ------------------------------------------------------------------------------------------
template <bool VEC4>
__global__ __launch_bounds__(256) void occ_pool8_kernel(const float* __restrict__ obs, float* __restrict__ feats, int S) {
    // column sums of the band, one row of the array per wave (the waves share the band's rows: wave w takes rows w, w + 4, ...);
    // added up per cell in (wave, column) order: reproducible.  S <= 2048.
    __shared__ float s_col[4][2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int oy = blockIdx.x & 7, nc = blockIdx.x >> 3;  // nc = env * 4 + channel
    const int cell = S >> 3;
    const float* band = obs + ((size_t)nc * S + (size_t)oy * cell) * S;
    const int ncol = VEC4 ? (S >> 2) : S;  // VEC4: a float4 column j = pixels 4j .. 4j+3, all of one cell
    for (int j = lane; j < ncol; j += 64) {
        float a = 0.f;
        if (VEC4) {
            for (int r = wave; r < cell; r += 4) {
                const float4 v = reinterpret_cast<const float4*>(band + (size_t)r * S)[j];
                a += (v.x + v.y) + (v.z + v.w);
            }
        } else {
            for (int r = wave; r < cell; r += 4) a += band[(size_t)r * S + j];
        }
        s_col[wave][j] = a;
    }
    __syncthreads();
    if (tid < 8) {
        const int per = ncol >> 3;  // columns per cell
        float a = 0.f;
        for (int w = 0; w < 4; ++w)
            for (int j = 0; j < per; ++j) a += s_col[w][tid * per + j];
        feats[(size_t)nc * 64 + oy * 8 + tid] = a / (float)(cell * cell);
    }
}
