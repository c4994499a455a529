// Optimizer step over FLAT buffers: global-norm gradient clipping (train.py:41) and
// Adam with the reference's hyper-parameters (lightning_module.py:160-163; torch.optim.Adam arithmetic, no amsgrad, no
// weight decay) as two streaming kernels over the same flat fp32 gradient bucket RCCL all-reduces -- instead of
// ~134 per-tensor foreach launches.  The Noam factor is applied by the host as `lr`.
#include "ttts_common.h"

namespace ttts {

constexpr int OPT_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, float* __restrict__ ws, long n4) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void norm_final_kernel(const float* __restrict__ ws, float* __restrict__ norm_out, int n) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += ws[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) norm_out[0] = sqrtf(s);
}

__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        const float* __restrict__ gnorm, long n4, float lr,
                                                        long step, const ttts_step_state* __restrict__ st, float beta1,
                                                        float beta2, float eps, float max_norm) {
    if (st != nullptr) {                 // captured-graph form: this step's lr and step count live in device memory
        lr = st->lr;
        step = st->step;
    }
    // bias corrections in double, exactly as torch.optim.Adam's host arithmetic (a few thousand cycles per thread, once)
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    float scale = 1.f;
    if (gnorm != nullptr && max_norm > 0.f) {
        const float c = max_norm / (gnorm[0] + 1e-6f);      // torch.nn.utils.clip_grad_norm_: coef clamped to 1
        scale = c < 1.f ? c : 1.f;
    }
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
        float mm[4] = {mv.x, mv.y, mv.z, mv.w}, vq[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gs = gg[j] * scale;
            mm[j] = beta1 * mm[j] + omb1 * gs;                 // exp_avg.lerp_(grad, 1 - beta1)
            vq[j] = beta2 * vq[j] + omb2 * gs * gs;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
            const float denom = sqrtf(vq[j]) * inv_bc2_sqrt + eps;
            pp[j] -= step_size * (mm[j] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(vq[0], vq[1], vq[2], vq[3]);
    }
}

}  // namespace ttts

using namespace ttts;

extern "C" {

size_t ttts_grad_norm_workspace_bytes(void) { return (size_t)OPT_BLOCKS * sizeof(float); }

int ttts_grad_norm(const float* g, float* norm_out, float* ws, size_t ws_bytes, int64_t n, void* stream_) {
    // norm_out[0] = || g ||_2 over the flat buffer (n % 4 == 0), fixed summation order
    hipStream_t stream = (hipStream_t)stream_;
    TTTS_REQUIRE(g && norm_out && ws && n > 0 && n % 4 == 0, "grad_norm: bad arguments (n %% 4 must be 0)");
    TTTS_REQUIRE(ws_bytes >= ttts_grad_norm_workspace_bytes(), "grad_norm: workspace too small");
    long n4 = n / 4;
    long gl = (n4 + 255) / 256;
    int grid = (int)(gl > OPT_BLOCKS ? OPT_BLOCKS : (gl < 1 ? 1 : gl));
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(grid), dim3(256), 0, stream, g, ws, n4);
    TTTS_LAUNCH_CHECK("sumsq_partial_kernel");
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(64), 0, stream, ws, norm_out, grid);
    TTTS_LAUNCH_CHECK("norm_final_kernel");
    return TTTS_OK;
}

int ttts_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, const float* grad_norm, int64_t n, float lr,
                   float beta1, float beta2, float eps, int64_t step, float max_grad_norm, const ttts_step_state* st,
                   void* stream) {
    // one torch.optim.Adam step (step >= 1 is the 1-based step count) on flat buffers; gradients are scaled by
    // min(1, max_grad_norm / (grad_norm + 1e-6)) first when grad_norm != NULL and max_grad_norm > 0.
    // st != NULL: lr and step are read from st (device memory) when the kernel runs and the by-value ones are ignored.
    TTTS_REQUIRE(p && g && exp_avg && exp_avg_sq && n > 0 && n % 4 == 0, "adam_step: bad arguments (n %% 4 must be 0)");
    TTTS_REQUIRE((st != nullptr || step >= 1) && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f,
                 "adam_step: bad hyper-parameters");
    long n4 = n / 4;
    long gl = (n4 + 255) / 256;
    int grid = (int)(gl > 2048 ? 2048 : (gl < 1 ? 1 : gl));
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg, exp_avg_sq, grad_norm,
                       n4, lr, (long)step, st, beta1, beta2, eps, max_grad_norm);
    TTTS_LAUNCH_CHECK("adam_step_kernel");
    return TTTS_OK;
}

}  // extern "C"
