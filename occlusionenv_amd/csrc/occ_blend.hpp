// occ_blend.hpp -- operator-level sigmoid_alpha_blend kernels.
// Part of the single translation unit occ_kernels.hip (included inside namespace occ; not a stand-alone header).

// [P3D] sigmoid_alpha_blend on K-buffers (operator level; the fused path never materialises them).  One thread per
// pixel, plain IEEE arithmetic in PyTorch3D's order: prob = sigmoid(-d / sigma) * mask, alpha = 1 - prod(1 - prob).
__global__ __launch_bounds__(256) void occ_blend_fwd_kernel(const float* __restrict__ dists, const int64_t* __restrict__ p2f,
                                                            long n_pix, int K, float sigma, float* __restrict__ images) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pix) return;
    float prod = 1.0f;
    for (int k = 0; k < K; ++k) {
        const float m = p2f[p * K + k] >= 0 ? 1.0f : 0.0f;
        const float prob = m / (1.0f + expf(dists[p * K + k] / sigma));
        prod *= 1.0f - prob;
    }
    reinterpret_cast<float4*>(images)[p] = make_float4(1.0f, 1.0f, 1.0f, 1.0f - prod);
}

// d alpha / d d_k = (prod_{j != k} (1 - prob_j)) * prob_k (1 - prob_k) / sigma   (masked entries carry no gradient);
// the leave-one-out products come from prefix / suffix passes, so a factor (1 - prob_k) = 0 needs no division
__global__ __launch_bounds__(256) void occ_blend_bwd_kernel(const float* __restrict__ dists, const int64_t* __restrict__ p2f,
                                                            const float* __restrict__ grad_images, long n_pix, int K,
                                                            float sigma, float* __restrict__ grad_dists) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pix) return;
    const float g = grad_images[p * 4 + 3];
    // suffix products into the output buffer first, then a forward sweep with the running prefix
    float suf = 1.0f;
    for (int k = K - 1; k >= 0; --k) {
        grad_dists[p * K + k] = suf;
        const float m = p2f[p * K + k] >= 0 ? 1.0f : 0.0f;
        suf *= 1.0f - m / (1.0f + expf(dists[p * K + k] / sigma));
    }
    float pre = 1.0f;
    for (int k = 0; k < K; ++k) {
        const float m = p2f[p * K + k] >= 0 ? 1.0f : 0.0f;
        const float prob = m / (1.0f + expf(dists[p * K + k] / sigma));
        // alpha = 1 - prod(q): d alpha / d prob_k = prod_{j != k} q_j ; d prob_k / d d_k = -prob_k (1 - prob_k) / sigma
        grad_dists[p * K + k] = -g * (pre * grad_dists[p * K + k]) * prob * (1.0f - prob) / sigma * m;
        pre *= 1.0f - prob;
    }
}
