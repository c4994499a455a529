// Either side of the model in every training step:
//  * TransformerTTSLoss (reference loss.py:15-55): masked MSE on pred / post mels + stop-gate BCE-with-logits with
//    pos_weight, as ONE streaming reduction + a tiny finalize, and ONE element-wise backward -- instead of the
//    reference's boolean-index gathers (loss.py:34-36,44) that allocate data-dependent shapes and synchronise.
//  * block-wise scheduled-sampling mix (reference utils/util.py:103-120): Bernoulli seed per frame from a uniform
//    draw, dilation by max_pool1d(kernel 8, stride 1, padding 4)[:T], select pred / ground truth, zero the padding.
#include "ttts_common.h"

namespace ttts {

constexpr int LOSS_BLOCKS = 1024;

__device__ __forceinline__ float softplus_neg(float x) {   // log(1 + exp(-x)), stable
    return fmaxf(-x, 0.f) + log1pf(__expf(-fabsf(x)));
}

// partial sums per block: [0] sum (pred-mel)^2 over valid frames, [1] same for post, [2] sum of weighted BCE terms
__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ pred, const float* __restrict__ post,
                                                           const float* __restrict__ stop, const float* __restrict__ mel,
                                                           const int64_t* __restrict__ lens, float* __restrict__ ws, int B,
                                                           int T, int C, float pos_weight) {
    __shared__ float red[3][4];
    const int c4n = C >> 2;
    const long n4 = (long)B * T * c4n;
    float s_pred = 0.f, s_post = 0.f, s_stop = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long frame = i / c4n;
        const int b = (int)(frame / T), t = (int)(frame - (long)b * T);
        if (t < lens[b]) {
            const float4 p = reinterpret_cast<const float4*>(pred)[i];
            const float4 q = reinterpret_cast<const float4*>(post)[i];
            const float4 m = reinterpret_cast<const float4*>(mel)[i];
            float d;
            d = p.x - m.x; s_pred += d * d; d = p.y - m.y; s_pred += d * d;
            d = p.z - m.z; s_pred += d * d; d = p.w - m.w; s_pred += d * d;
            d = q.x - m.x; s_post += d * d; d = q.y - m.y; s_post += d * d;
            d = q.z - m.z; s_post += d * d; d = q.w - m.w; s_post += d * d;
        }
    }
    const long nf = (long)B * T;
    for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += (long)gridDim.x * blockDim.x) {
        const int b = (int)(f / T), t = (int)(f - (long)b * T);
        const long len = lens[b];
        if (t < len) {
            const float x = stop[f];
            const float y = (t == len - 1) ? 1.f : 0.f;
            s_stop += (1.f - y) * x + (1.f + (pos_weight - 1.f) * y) * softplus_neg(x);
        }
    }
    s_pred = wave_sum(s_pred); s_post = wave_sum(s_post); s_stop = wave_sum(s_stop);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s_pred; red[1][wave] = s_post; red[2][wave] = s_stop; }
    __syncthreads();
    if (threadIdx.x < 3)
        ws[(long)blockIdx.x * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// out = [total, pred_mel, post_mel, stop], aux = [1/(n_frames*C), 1/n_frames]
__global__ void loss_final_kernel(const float* __restrict__ ws, const int64_t* __restrict__ lens, float* __restrict__ out,
                                  float* __restrict__ aux, int nblk, int B, int T, int C) {
    __shared__ float red[3];
    float s = 0.f;
    const int which = threadIdx.x >> 6, lane = threadIdx.x & 63;   // 3 waves, one per sum
    if (which < 3) {
        for (int i = lane; i < nblk; i += 64) s += ws[(long)i * 3 + which];
        s = wave_sum(s);
        if (lane == 0) red[which] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long n = 0;
        for (int b = 0; b < B; ++b) { long l = lens[b]; n += (l < T ? (l > 0 ? l : 0) : T); }
        const float inv_nc = 1.f / ((float)n * (float)C), inv_n = 1.f / (float)n;
        const float lp = red[0] * inv_nc, lq = red[1] * inv_nc, ls = red[2] * inv_n;
        out[0] = lp + 0.5f * lq + ls; out[1] = lp; out[2] = lq; out[3] = ls;
        aux[0] = inv_nc; aux[1] = inv_n;
    }
}

// g0..g3 = upstream gradients of total, pred_mel, post_mel, stop: device scalars, NULL = that output has no gradient (0)
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ post,
                                                       const float* __restrict__ stop, const float* __restrict__ mel,
                                                       const int64_t* __restrict__ lens, const float* __restrict__ aux,
                                                       const float* __restrict__ g0, const float* __restrict__ g1,
                                                       const float* __restrict__ g2, const float* __restrict__ g3,
                                                       float* __restrict__ dpred, float* __restrict__ dpost,
                                                       float* __restrict__ dstop, int B, int T, int C, float pos_weight) {
    const float inv_nc = aux[0], inv_n = aux[1];
    const float gt = g0 ? g0[0] : 0.f, gp_ = g1 ? g1[0] : 0.f, gq_ = g2 ? g2[0] : 0.f, gs_ = g3 ? g3[0] : 0.f;
    const float kp = (gt + gp_) * 2.f * inv_nc, kq = (0.5f * gt + gq_) * 2.f * inv_nc, ks = (gt + gs_) * inv_n;
    const int c4n = C >> 2;
    const long n4 = (long)B * T * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long frame = i / c4n;
        const int b = (int)(frame / T), t = (int)(frame - (long)b * T);
        float4 gp = make_float4(0.f, 0.f, 0.f, 0.f), gq = gp;
        if (t < lens[b]) {
            const float4 p = reinterpret_cast<const float4*>(pred)[i];
            const float4 q = reinterpret_cast<const float4*>(post)[i];
            const float4 m = reinterpret_cast<const float4*>(mel)[i];
            gp = make_float4(kp * (p.x - m.x), kp * (p.y - m.y), kp * (p.z - m.z), kp * (p.w - m.w));
            gq = make_float4(kq * (q.x - m.x), kq * (q.y - m.y), kq * (q.z - m.z), kq * (q.w - m.w));
        }
        reinterpret_cast<float4*>(dpred)[i] = gp;
        reinterpret_cast<float4*>(dpost)[i] = gq;
    }
    const long nf = (long)B * T;
    for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += (long)gridDim.x * blockDim.x) {
        const int b = (int)(f / T), t = (int)(f - (long)b * T);
        const long len = lens[b];
        float gs = 0.f;
        if (t < len) {
            const float x = stop[f];
            const float y = (t == len - 1) ? 1.f : 0.f;
            const float sig = 1.f / (1.f + __expf(-x));
            gs = ks * (sig * (1.f + (pos_weight - 1.f) * y) - pos_weight * y);
        }
        dstop[f] = gs;
    }
}

// mixed[b,t,:] = (t < len) ? (any(u[b, t-4 .. t+3] < 1-p_tf) ? pred : mel) : 0
// u == NULL: the uniform draw of frame (b,s) is generated here, 24 bits of hash(seed, b*T + s) -- one value per
// frame whichever thread asks for it, so the 8-frame windows of neighbouring outputs see the same draws.
__global__ __launch_bounds__(256) void sched_mix_kernel(const float* __restrict__ pred, const float* __restrict__ mel,
                                                        const float* __restrict__ u, const int64_t* __restrict__ lens,
                                                        float* __restrict__ out, int B, int T, int C, float thresh,
                                                        int l_bar, uint64_t seed, const ttts_step_state* __restrict__ st,
                                                        float* __restrict__ amax_out) {
    float vmax = 0.f;
    if (st != nullptr) {                 // captured-graph form: this step's ratio and seed word live in device memory
        thresh = 1.0f - st->p_tf;
        seed ^= st->seed;
    }
    const int c4n = C >> 2;
    const long n4 = (long)B * T * c4n;
    const int pad = l_bar / 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long frame = i / c4n;
        const int b = (int)(frame / T), t = (int)(frame - (long)b * T);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < lens[b]) {
            bool take_pred = false;
            for (int k = 0; k < l_bar; ++k) {                 // window of output t: inputs t-pad .. t-pad+l_bar-1
                int s = t - pad + k;
                if (s >= 0 && s < T) {
                    const long f = (long)b * T + s;
                    const float us = u != nullptr ? u[f]
                                                  : (float)(hash_pair(seed, (uint32_t)f, (uint32_t)(f >> 32) + 0x51ED27u) >> 8) *
                                                        (1.0f / 16777216.0f);
                    take_pred = take_pred || (us < thresh);
                }
            }
            v = take_pred ? reinterpret_cast<const float4*>(pred)[i] : reinterpret_cast<const float4*>(mel)[i];
        }
        reinterpret_cast<float4*>(out)[i] = v;
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (amax_out != nullptr) amax_publish(vmax, amax_out, blockIdx.x * 4 + (threadIdx.x >> 6));
}

static inline int grid_for(long n_items, int cap) {
    long g = (n_items + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace ttts

using namespace ttts;

extern "C" {

size_t ttts_loss_workspace_bytes(void) { return (size_t)(LOSS_BLOCKS * 3 + 2) * sizeof(float); }

int ttts_loss_fwd(const float* pred, const float* post, const float* stop, const float* mel, const int64_t* lens,
                  float* out4, float* ws, size_t ws_bytes, int B, int T, int C, float pos_weight, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TTTS_REQUIRE(pred && post && stop && mel && lens && out4 && ws, "loss_fwd: null pointer");
    TTTS_REQUIRE(B > 0 && T > 0 && C > 0 && C % 4 == 0, "loss_fwd: C=%d must be a multiple of 4", C);
    TTTS_REQUIRE(ws_bytes >= ttts_loss_workspace_bytes(), "loss_fwd: workspace too small");
    int nblk = grid_for((long)B * T * (C / 4), LOSS_BLOCKS);
    hipLaunchKernelGGL(loss_partial_kernel, dim3(nblk), dim3(256), 0, stream, pred, post, stop, mel, lens, ws, B, T, C,
                       pos_weight);
    TTTS_LAUNCH_CHECK("loss_partial_kernel");
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, stream, ws, lens, out4, ws + LOSS_BLOCKS * 3, nblk, B, T, C);
    TTTS_LAUNCH_CHECK("loss_final_kernel");
    return TTTS_OK;
}

int ttts_loss_bwd(const float* pred, const float* post, const float* stop, const float* mel, const int64_t* lens,
                  const float* ws, const float* g_total, const float* g_pred_mel, const float* g_post_mel, const float* g_stop,
                  float* dpred, float* dpost, float* dstop, int B, int T, int C, float pos_weight, void* stream) {
    TTTS_REQUIRE(pred && post && stop && mel && lens && ws && dpred && dpost && dstop, "loss_bwd: null pointer");
    TTTS_REQUIRE(B > 0 && T > 0 && C > 0 && C % 4 == 0, "loss_bwd: C=%d must be a multiple of 4", C);
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(grid_for((long)B * T * (C / 4), 2048)), dim3(256), 0, (hipStream_t)stream, pred,
                       post, stop, mel, lens, ws + LOSS_BLOCKS * 3, g_total, g_pred_mel, g_post_mel, g_stop, dpred, dpost, dstop, B, T, C,
                       pos_weight);
    TTTS_LAUNCH_CHECK("loss_bwd_kernel");
    return TTTS_OK;
}

int ttts_sched_sampling_mix(const float* pred, const float* mel, const float* u, const int64_t* lens, float* out, int B,
                            int T, int C, float p_tf, int l_bar, uint64_t seed, const ttts_step_state* st,
                            float* out_amax_out, void* stream) {
    TTTS_REQUIRE(pred && mel && lens && out, "sched_sampling_mix: null pointer");
    TTTS_REQUIRE(B > 0 && T > 0 && C > 0 && C % 4 == 0 && l_bar > 0, "sched_sampling_mix: bad dims (C %% 4 must be 0)");
    hipLaunchKernelGGL(sched_mix_kernel, dim3(grid_for((long)B * T * (C / 4), 2048)), dim3(256), 0, (hipStream_t)stream, pred,
                       mel, u, lens, out, B, T, C, 1.0f - p_tf, l_bar, seed, st, out_amax_out);
    TTTS_LAUNCH_CHECK("sched_mix_kernel");
    return TTTS_OK;
}

}  // extern "C"
