/* ttts_hip.h -- C ABI of libttts_hip.so: hand-written gfx950 (MI355X / CDNA4) HIP kernels for the
 * teacher-forced Transformer-TTS forward/backward hot path.
 *
 * The reference (Orca0917/TransformerTTS, /root/reference) has no FFI of its own: it is pure Python
 * and delegates its arithmetic to stock torch ops.  Each entry point below therefore names the torch
 * call site in the reference that it replaces (file:line relative to /root/reference;
 * "torch/..." = the un-vendored PyTorch the reference runs on).
 *
 * Conventions:
 *   - plain pointers + sizes only; activations are row-major contiguous (B,T,C) fp32, weights keep the
 *     state-dict layouts ((out,in) linear, (Cout,Cin,K) conv, packed (3d,d) in-proj); lengths are int64.
 *   - every function only ENQUEUES work on `stream` (a hipStream_t passed as void*); no device allocation, no
 *     synchronisation, no state that outlives the call: workspaces are owned by the caller, and so is the one host-side
 *     object of the ABI, the deferred-reduction queue (ttts_reduce_queue).  The library reads no environment variables.
 *     Calls are re-entrant from several host threads on different streams (different queues, different workspaces).
 *   - return 0 on success, a negative TTTS_ERR_* otherwise (never throws / exits);
 *     ttts_last_error() returns a thread-local message for the last failure.
 *   - dropout masks are a pure function of (seed, flat element index), so backward entry points
 *     regenerate the forward mask from the same seed; p = 0 disables dropout.
 *   - HIP-graph replay: kernel arguments are frozen when a graph is captured, so everything that changes from step to
 *     step can ALSO be read from a small block of device memory (ttts_step_state): every `seed` argument is
 *     accompanied by `step_seed` (NULL, or a device pointer whose 64-bit word is XORed into the seed when the kernel
 *     runs -- &state->seed), and the optimizer / scheduled-sampling entry points take the state block itself.
 *     The caller refreshes the block with one small host-to-device copy per step, outside the graph.
 */
#ifndef TTTS_HIP_H
#define TTTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTTS_OK 0
#define TTTS_ERR_INVALID (-1) /* bad argument (shape, alignment, null pointer) */
#define TTTS_ERR_LAUNCH (-2)  /* hipLaunchKernel reported an error */

/* length of every partial-maxima array (`*_amax`, `*_amax_out`) of this header, in floats */
#define TTTS_AMAX_SLOTS 1024

#define TTTS_ACT_NONE 0
#define TTTS_ACT_RELU 1
#define TTTS_ACT_TANH 2

/* Per-step scalars a captured graph reads from device memory (32 bytes, fields at fixed offsets). */
typedef struct ttts_step_state {
    uint64_t seed;    /*  0: XORed into every dropout / sampling site seed of the step */
    float lr;         /*  8: Adam learning rate of this step (base lr x Noam factor) */
    float p_tf;       /* 12: teacher-forcing ratio of this step (utils/util.py:54-92) */
    int64_t step;     /* 16: 1-based optimizer step count (Adam bias correction) */
    int64_t reserved; /* 24 */
} ttts_step_state;

const char* ttts_last_error(void);
int ttts_abi_version(void);

/* ---- deferred second-stage reductions ---------------------------------------------------------------------------
 * The weight-gradient (linear / conv1d / rowdot) and LayerNorm-backward entry points end in a small reduction of
 * their split-K or per-block partials into the parameter gradient -- ~90 launches of a few microseconds each per training
 * step.  They take a `queue` argument: NULL launches the reduction at once; a queue created by the caller makes the
 * entry point append a descriptor to it instead, and ttts_reduce_queue_flush runs everything appended so far in one
 * launch per 48 entries (same summation order as the immediate form, bit for bit).  Until the flush has been enqueued on
 * the same stream the caller must keep every workspace (`ws`) it passed alive and must not read those gradients.
 * The queue is host memory owned by its creator; the library keeps no queue of its own.  Appending (from whichever
 * thread runs the backward node) and flushing are safe against each other; two independent backward passes use two
 * queues.  clear drops what is queued (after a failed pass). */
typedef struct ttts_reduce_queue ttts_reduce_queue;
ttts_reduce_queue* ttts_reduce_queue_create(void);
void ttts_reduce_queue_destroy(ttts_reduce_queue* queue);
int64_t ttts_reduce_queue_pending(ttts_reduce_queue* queue);
int ttts_reduce_queue_flush(ttts_reduce_queue* queue, void* stream);
int ttts_reduce_queue_clear(ttts_reduce_queue* queue);

/* ------------------------------------------------------------------ linear (nn.Linear / LinearNorm)
 * y[M,N] = drop(act(x[M,K] . w[N,K]^T + bias)) + residual          K % 16 == 0
 * Replaces: LinearNorm.forward (model/module.py:52-53); MHA in/out projections
 * (torch/nn/functional.py:6206+ in_proj, out_proj); FFN `_ff_block`
 * (torch/nn/modules/transformer.py:980-982,1197-1199) incl. its ReLU, Dropout and the residual add
 * of model/layers.py:47-50; DecoderPreNet (model/model.py:65-66).
 * row_shift = -1 with T = frames per utterance folds the go-frame shift of model/model.py:278-279
 * into the loader (row (b,t) reads x[b,t-1], zeros at t = 0). */
int ttts_linear_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y, int64_t M,
                    int N, int K, int act, float drop_p, uint64_t seed, const uint64_t* step_seed, int row_shift, int T,
                    void* stream);
/* dx[M,K] = gate * (dy[M,N] . w[N,K]) + residual[M,K]               N % 16 == 0, K % 4 == 0
 * residual may be NULL.  relu_out (NULL or (M,K)): the forward value of the activation dx is the gradient of, when that
 * activation is drop(relu(.)) of a preceding Linear (model/module.py:76-80 prenet, the torch FFN block): gate =
 * relu_out > 0 ? relu_scale : 0 with relu_scale = 1/(1-p) -- the relu / dropout backward mask in the epilogue. */
int ttts_linear_bwd_data(const float* dy, const float* w, const float* residual, float* dx, int64_t M, int N, int K,
                         const float* relu_out, float relu_scale, void* stream);
/* dw[N,K] = dy[M,N]^T . x[M,K] (x rows shifted as in forward); dbias[N] = column sums of dy (optional).
 * Every parameter-gradient output in this header takes `accumulate`: 0 stores, 1 adds to what is there, so that
 * gradients can land directly in slices of one flat, pre-zeroed gradient buffer (the data-parallel bucket). */
size_t ttts_wgrad_workspace_bytes(int64_t M, int N, int K, int taps);
int ttts_linear_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, float* ws, size_t ws_bytes,
                           int64_t M, int N, int K, int row_shift, int T, int accumulate, ttts_reduce_queue* queue,
                           void* stream);

/* ---- split-precision ("bf16x6") forms of the forward / data-gradient GEMMs.  Same arithmetic contract (fp32 in,
 * fp32 out, fp32 accumulate) with every product formed as six bf16 x bf16 MFMA terms of a 3-way hi/mid/lo split:
 * measured error vs fp64 1.1e-7 (a plain fp32 fma chain: 2.9e-7) at 6/16 of the fp32 MFMA cycles.
 * The weight operand is split once per call by ttts_weight_split into three bf16 planes (hi, mid, lo) of a rows x cols
 * matrix, stored k-tile-major as [cols/16][plane][rows][16] (cols must be a multiple of 16):
 *   mode 0  linear forward     planes of w (N,K)               rows = N,    cols = K
 *   mode 1  linear data-grad   planes of w^T                   rows = K,    cols = N
 *   mode 2  conv forward       [co][tap*cin + ci]              rows = cout, cols = taps*cin,  channels_per_tap = cin
 *   mode 3  conv data-grad     [ci][tap*cout + co]             rows = cin,  cols = taps*cout, channels_per_tap = cout
 *
 * fp16x3 ("h3") form: modes 4-7 of ttts_weight_split are the same four matrices as modes 0-3 written as TWO f16 planes
 * (hi, lo) of w * s, stored [cols/32][plane][rows][32] (cols, and channels_per_tap for the conv modes, must be multiples
 * of 32), followed by a 16-byte tail whose first float is max|w|: s is the power of two that puts max|w| in [2^11, 2^12),
 * measured by the split itself and re-derived from the tail by every kernel that takes the planes.
 * ttts_linear_fwd_h3 / ttts_conv1d_fwd_h3 take such planes and form each product from three f16 x f16 MFMA terms
 * (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi): the same fp32-grade result (error vs fp64 a few 1e-7) for half the matrix-pipe
 * work.  f16 has no bf16-like exponent range, so the ACTIVATION operand is pre-scaled while it is staged, by the power
 * of two that puts ITS measured maximum in [2^11, 2^12): `x_amax` = TTTS_AMAX_SLOTS partial maxima of |x| (their maximum must be
 * >= max|x|; ttts_amax_partials computes them with one read of x, and every kernel of this header that PRODUCES an
 * activation or a gradient can leave them behind itself: the `*_amax_out` arguments).  Every element within 2^-15 of the
 * largest keeps 22 significant bits, smaller ones an absolute error of 2^-37 * max|x| -- no assumption about the operands'
 * magnitudes is left.
 * Every such array has TTTS_AMAX_SLOTS floats.  Arrays named `*_amax_out` are filled with slot-wise atomic maxima and must
 * be zeroed by the caller first (ttts_zero); only ttts_amax_partials writes its array in full. */
size_t ttts_split_bytes(int64_t rows, int64_t cols);
/* bytes of the image ttts_weight_split writes in `mode`.  Modes 4-7 (fp16x3) pad the channels of a tap -- all columns of a
 * linear weight -- with zeros to the next multiple of 32, so that the 32-deep k-tiles of the GEMM never straddle a tap; a
 * channel count that is no multiple of 32 (the 80-channel mel side) therefore needs no other kernel: the loader reads the
 * activation row's neighbour against zero weight columns.  Channel counts must be multiples of 4. */
size_t ttts_split_image_bytes(int64_t rows, int64_t cols, int mode, int channels_per_tap, int taps);
/* tile shape the forward / data-gradient dispatch uses for an M x N output with reduction length K (1: 64x64, 2: 128x128,
 * 3: 64x128, 4: 128x96; fp16x3 only: 6: 256x256 / 8 waves, 7: 256x128 / 8 waves, 8: 256x128 / 4 waves, two workgroups per
 * CU); x6 = 0: fp32-MFMA kernel, 1: bf16x6 kernel, 2: fp16x3 kernel.  A profiling aid (bench.py attributes launches). */
int ttts_gemm_tile_choice(int64_t M, int N, int K, int x6);
int ttts_weight_split(const float* w, void* planes, int rows, int cols, int mode, int channels_per_tap, int taps,
                      void* stream);
/* all weights in one launch: descs (device memory) = n x 8 int64 {w, planes, rows, cols, mode, channels_per_tap, taps,
 * first_block}, first_block = running sum of ttts_weight_split_units(rows, cols, mode, channels_per_tap) over the entries
 * before this one, total_blocks = the final sum.  (A unit is one workgroup: 256 elements of a bf16x6 image; a 32-row x
 * 32-channel tile, all taps of it, of an fp16x3 image -- read in the order the source is contiguous in and written as whole
 * 2 KB runs of the planes.)  A LINEAR fp16x3 entry (modes 4, 5, 8, 9) may be a WINDOW of a STACKED image that several
 * weights share (the fused K/V projection of all decoder layers): channels_per_tap = (image rows << 32) | image columns and
 * taps = (first image row << 32) | first image column of the window (a multiple of 32); `planes` then is the shared image,
 * whose one tail -- one scale -- receives the maximum over every window's weight.  0 / 0: the entry is its own image. */
int64_t ttts_weight_split_units(int64_t rows, int64_t cols, int mode, int channels_per_tap);
int ttts_weight_split_batched(const int64_t* descs, int n, int64_t total_blocks, void* stream);
int ttts_linear_fwd_x6(const float* x, const void* w_planes, const float* bias, const float* residual, float* y,
                       int64_t M, int N, int K, int act, float drop_p, uint64_t seed, const uint64_t* step_seed, int row_shift, int T,
                    void* stream);
/* y_amax_out: NULL, or a caller-zeroed TTTS_AMAX_SLOTS-float array receiving max|y| (y feeds another fp16x3 GEMM / attention) */
int ttts_linear_fwd_h3(const float* x, const void* w_planes, const float* bias, const float* residual, float* y,
                       int64_t M, int N, int K, int act, float drop_p, uint64_t seed, const uint64_t* step_seed, int row_shift,
                       int T, const float* x_amax, float* y_amax_out, void* stream);
/* bn_partials: NULL, or the workspace of ttts_bn_train_stats (ttts_bn_workspace_bytes): the epilogue then leaves BatchNorm's
 * row-chunk partials (count, mean, M2 per output channel and chunk, [nblk][3][cout]) behind and
 * ttts_bn_train_stats_from_partials finishes the statistics without a pass over y.  nblk = ttts_conv1d_fwd_h3_bn_blocks(...);
 * 0 means this shape's tile cannot emit them (pass NULL and use ttts_bn_train_stats). */
int ttts_conv1d_fwd_h3_bn_blocks(int B, int T, int cin, int cout, int taps);
/* rows of (b, t) one partial covers (partial i: rows i * chunk ..; 0: this shape emits none): the statistics of a row range that is
 * a whole number of chunks can be finished from its own partials, bn_partials + first_chunk * 3 * cout (ABI v12) */
int ttts_conv1d_fwd_h3_bn_chunk_rows(int B, int T, int cin, int cout, int taps);
int ttts_conv1d_fwd_h3(const float* x, const void* planes_fwd, const float* bias, float* y, int B, int T, int cin, int cout,
                       int taps, const float* x_amax, float* bn_partials, void* stream);
/* fp16x3 data gradients: as the forward, with the gradient as the activation operand (dy_amax: its partial maxima;
 * 1e-7 .. 1e-5 behind a mean-reduced loss, any magnitude in general).  planes: modes 5 / 7. */
int ttts_amax_partials(const float* x, int64_t n, float* partials /* TTTS_AMAX_SLOTS floats, fully written */, void* stream);
int ttts_linear_bwd_data_h3(const float* dy, const void* wt_planes, const float* residual, float* dx, int64_t M, int N,
                            int K, const float* relu_out, float relu_scale, const float* dy_amax, float* dx_amax_out,
                            void* stream);
/* ---- fp16x3 with an IMAGE activation operand (ABI v10; transformertts_amd/csrc/gemm_h3i.hip).  The kernels above split
 * the fp32 activation while they stage it (registers, 3.5 VALU instructions per MFMA, one k-tile of requests in flight per
 * CU).  Here the producer of the activation writes it already split: an IMAGE is row-major, a row of K values is K/16
 * groups of 64 bytes = {16 f16 "hi", 16 f16 "lo"} of x * 2^e_row (4 bytes per element, as fp32) with a PER-ROW power of two
 * that puts the row's maximum in [2^11, 2^12); row_inv[row] = 2^-e_row.  A row scale of the left operand factors out of the
 * output row, so the GEMM undoes it in its epilogue.  Both operands are then staged by LDS-DMA (no staging registers), on a
 * 128 x 256 tile with two workgroups per CU.  ttts_act_image makes an image from fp32 (K % 16 == 0, K <= 1024);
 * ttts_layernorm_fwd / _bwd emit one next to their fp32 output (their *_image_out arguments).  Replaces the same call
 * sites as ttts_linear_fwd_h3 / ttts_linear_bwd_data_h3 (torch F.linear inside nn.MultiheadAttention / _ff_block,
 * torch/nn/modules/transformer.py:951-982,1158-1199, reached from model/model.py:189-213 and model/layers.py:29-74);
 * no row shift, K % 32 == 0, N % 4 == 0. */
int ttts_act_image(const float* x, void* image, float* row_inv, int64_t M, int K, void* stream);
int ttts_linear_fwd_h3i(const void* x_image, const float* x_row_inv, const void* w_planes, const float* bias,
                        const float* residual, float* y, int64_t M, int N, int K, int act, float drop_p, uint64_t seed,
                        const uint64_t* step_seed, float* y_amax_out, void* stream);
int ttts_linear_bwd_data_h3i(const void* dy_image, const float* dy_row_inv, const void* wt_planes, const float* residual,
                             float* dx, int64_t M, int N, int K, const float* relu_out, float relu_scale,
                             float* dx_amax_out, void* stream);
/* the same kernel on a plain fp32 activation (per-tensor scale from x_amax, as ttts_linear_fwd_h3): the raw k-tile is staged by
 * LDS-DMA and split in place in LDS, so no producer has to write an image; weight planes = modes 8 / 9.  Arguments as
 * ttts_linear_fwd_h3 / ttts_linear_bwd_data_h3 (no row shift; K resp. N a multiple of 32). */
int ttts_linear_fwd_h3d(const float* x, const void* w_planes, const float* bias, const float* residual, float* y,
                        int64_t M, int N, int K, int act, float drop_p, uint64_t seed, const uint64_t* step_seed,
                        const float* x_amax, float* y_amax_out, void* stream);
int ttts_linear_bwd_data_h3d(const float* dy, const void* wt_planes, const float* residual, float* dx, int64_t M, int N,
                             int K, const float* relu_out, float relu_scale, const float* dy_amax, float* dx_amax_out,
                             void* stream);
/* ---- HEAD-IMAGE outputs (ABI v11; gemm_h3i.hip, attention_img.hip).  The output of an attention in-projection (the packed
 * q/k/v of torch `_sa_block` / F.multi_head_attention_forward, torch/nn/functional.py:6206+, or the q and k/v row slices of
 * model/layers.py:54-74) is read by nothing but the attention kernels, which form every product from f16 hi / lo pieces.  So the
 * in-projection writes it in that form INSTEAD of fp32: y_image has the geometry of the fp32 output (M rows of N 4-byte cells),
 * and the 256 bytes of (row, 64-column head) hold {64 f16 hi, 64 f16 lo} of (x W^T + b) * 2^e(row, head) with the power of
 * two that puts the head row's maximum in [2^11, 2^12); y_row_inv[head * M + row] = 2^-e.  The attention kernels stage K / V
 * tiles of it by LDS-DMA (no split arithmetic, two tiles deep).  y_amax_out: NULL, or N / amax_section_cols caller-zeroed
 * TTTS_AMAX_SLOTS-float arrays (one per section of amax_section_cols columns: q / k / v) receiving max|y| of the section
 * (amax_section_cols = 0: one array).  K % 32 == 0, N % 64 == 0, bias-only epilogue; weight planes = mode 8. */
int ttts_linear_fwd_h3d_img(const float* x, const void* w_planes, const float* bias, void* y_image, float* y_row_inv,
                            int64_t M, int N, int K, const float* x_amax, float* y_amax_out, int amax_section_cols,
                            void* stream);
/* the stand-alone conversion: fp32 rows (row stride ld_x floats) -> head image (row stride ld_image cells) + inverse scales
 * [N / 64][M] (tests; operands no GEMM of this library produced) */
int ttts_head_image(const float* x, int64_t ld_x, void* image, int64_t ld_image, float* row_inv, int64_t M, int N, void* stream);
int ttts_conv1d_bwd_data_h3(const float* dy, const void* planes_bwd, float* dx, int B, int T, int cin, int cout, int taps,
                            const float* dy_amax, void* stream);
int ttts_linear_bwd_data_x6(const float* dy, const void* wt_planes, const float* residual, float* dx, int64_t M, int N,
                            int K, const float* relu_out, float relu_scale, void* stream);
int ttts_conv1d_fwd_x6(const float* x, const void* planes_fwd, const float* bias, float* y, int B, int T, int cin, int cout,
                       int taps, void* stream);
int ttts_conv1d_bwd_data_x6(const float* dy, const void* planes_bwd, float* dx, int B, int T, int cin, int cout, int taps,
                            void* stream);
/* Weight gradients in the same split-precision form (both operands are activations, split while they are staged);
 * arguments, workspace (ttts_wgrad_workspace_bytes) and results as ttts_linear_bwd_weight / ttts_conv1d_bwd_weight. */
int ttts_linear_bwd_weight_x6(const float* dy, const float* x, float* dw, float* dbias, float* ws, size_t ws_bytes,
                              int64_t M, int N, int K, int row_shift, int T, int accumulate, ttts_reduce_queue* queue,
                              void* stream);
int ttts_conv1d_bwd_weight_x6(const float* dy, const float* x, float* dw, float* dbias, float* ws, size_t ws_bytes, int B,
                              int T, int cin, int cout, int taps, int accumulate, ttts_reduce_queue* queue, void* stream);
/* fp16x3 weight gradients: both operands are activations, pre-scaled from their partial maxima (dy_amax is shared with the
 * data gradient of the same dy, x_amax with the forward GEMM that read x); shapes the split kernels do not cover fall
 * back to the fp32-MFMA kernel exactly as the _x6 entry points do. */
int ttts_linear_bwd_weight_h3(const float* dy, const float* x, float* dw, float* dbias, float* ws, size_t ws_bytes,
                              int64_t M, int N, int K, int row_shift, int T, int accumulate, const float* dy_amax,
                              const float* x_amax, ttts_reduce_queue* queue, void* stream);
int ttts_conv1d_bwd_weight_h3(const float* dy, const float* x, float* dw, float* dbias, float* ws, size_t ws_bytes, int B,
                              int T, int cin, int cout, int taps, int accumulate, const float* dy_amax, const float* x_amax,
                              ttts_reduce_queue* queue, void* stream);
/* ONE weight-gradient GEMM dy^T x (N x K) whose N rows belong to `nparts` weights (equal row blocks): block i is reduced into
 * dw_parts[i] ((N / nparts) x K floats) and its column sums into dbias_parts[i] (NULL array: no bias gradients); dw_parts /
 * dbias_parts are HOST arrays of device pointers.  The backward of the cross-attention K/V projection of every decoder layer
 * run as one GEMM on the encoder memory they all read (the reference runs it per layer: model/layers.py:54-74, rows d..3d of
 * each layer's multihead_attn.in_proj_weight). */
/* ---- ABI v12 (round 6): grouped weight gradients, the multi-destination weight gradient, stacked weight images
 * (ttts_weight_split_batched).
 * GROUPED weight gradients: n <= 4 independent problems as ONE launch of an fp16x3 weight-gradient kernel -- the weights of a layer,
 * whose operands are all at hand when backward leaves the layer and whose results nobody reads before the optimizer.  Member i
 * with taps_i = 1: dw_i[N_i,K_i] (+)= dy_i[M_i,N_i]^T x_i[M_i,K_i] (ttts_linear_bwd_weight_h3; row_shift_i / T_i as its row_shift /
 * T); taps_i > 1: the weight gradient
 * of a same-padded convolution over utterances of T_i rows, dw_i[N_i = cout, K_i = cin, taps_i], M_i = B T_i
 * (ttts_conv1d_bwd_weight_h3).  dbias_i: column sums of dy_i, or NULL.  Every array argument is a HOST array of n entries; ws_i /
 * ws_bytes_i as ttts_wgrad_workspace_bytes(M_i, N_i, K_i, taps_i).  The row splits are planned for the group, so each member writes
 * 1/n of the partial sums of a launch of its own.  ttts_wgrad_group_ok: the CLASS of a problem -- 0: no member of any group; 1: the
 * 4-wave 128 x 128 tile; 2: whole 256 x 256 tiles on the 8-wave LDS-DMA kernel (the FFN, packed in-projection and post-net
 * convolution weights over long row ranges); 3 / 4: the 128 x 96 / 96 x 128 tiles of weights with 80 input / output channels
 * (the mel side) -- the members of one launch share a class.  (The autograd of torch.nn.Linear /
 * nn.Conv1d runs these one by one: model/module.py:4-53 and the torch layers of model/model.py:189-213.) */
int ttts_wgrad_group_ok(int64_t M, int N, int K, int taps);
int ttts_wgrad_group(int n, const float* const* dy, const float* const* x, float* const* dw, float* const* dbias,
                     float* const* ws, const size_t* ws_bytes, const int64_t* M, const int* N, const int* K, const int* taps,
                     const int* T, const int* row_shift, int accumulate, const float* const* dy_amax, const float* const* x_amax,
                     ttts_reduce_queue* queue, void* stream);
int ttts_linear_bwd_weight_h3_parts(const float* dy, const float* x, float* const* dw_parts, float* const* dbias_parts, int nparts,
                                    float* ws, size_t ws_bytes, int64_t M, int N, int K, int accumulate, const float* dy_amax,
                                    const float* x_amax, ttts_reduce_queue* queue, void* stream);

/* ------------------------------------------------------------------ Conv1d (k taps, same padding) on (B,T,C)
 * Replaces ConvNormBN's permute -> nn.Conv1d(pad=(k-1)//2) -> permute (model/module.py:28-33) as an
 * implicit GEMM directly on the (B,T,C) layout.  Weights are re-laid once per call by
 * ttts_conv1d_pack_weight: w[co][ci][tap] -> w_fwd[co][tap][ci] and/or w_bwd[ci][tap][co]. */
size_t ttts_conv1d_pack_bytes(int cout, int cin, int taps);
int ttts_conv1d_pack_weight(const float* w, float* w_fwd, float* w_bwd, int cout, int cin, int taps, void* stream);
int ttts_conv1d_fwd(const float* x, const float* w_fwd, const float* bias, float* y, int B, int T, int cin, int cout,
                    int taps, void* stream);
int ttts_conv1d_bwd_data(const float* dy, const float* w_bwd, float* dx, int B, int T, int cin, int cout, int taps,
                         void* stream);
int ttts_conv1d_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, float* ws, size_t ws_bytes, int B,
                           int T, int cin, int cout, int taps, int accumulate, ttts_reduce_queue* queue, void* stream);

/* ------------------------------------------------------------------ BatchNorm1d over (M = B*T rows, C channels)
 * Replaces nn.BatchNorm1d inside ConvNormBN (model/module.py:19,31) plus the Tanh / Dropout entries that
 * follow it in EncoderPreNet / PostNet (model/model.py:29-33,113-126).
 * train: batch statistics over all rows (padding included), running stats updated with momentum and the
 * unbiased variance, num_batches_tracked += 1.  eval: normalise with the running statistics. */
size_t ttts_bn_workspace_bytes(int64_t M, int C);
int ttts_bn_train_stats(const float* x, float* mean, float* invstd, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float* ws, size_t ws_bytes, int64_t M, int C, float momentum,
                        float eps, void* stream);
int ttts_bn_train_stats_from_partials(const float* partials, int nblk, float* mean, float* invstd, float* running_mean,
                                      float* running_var, int64_t* num_batches_tracked, int C, float momentum, float eps,
                                      void* stream);
/* as above, merged with n_rows (0 .. 256) rows of the matrix itself (row stride C; `rows` their first) that none of the nblk (>= 0)
 * partials covers: a row range that does not end on a chunk boundary takes its whole chunks from the epilogue's partials and the rest
 * from y (the halves of a twin batch).  (ABI v13) */
int ttts_bn_train_stats_from_partials_rows(const float* partials, int nblk, const float* rows, int n_rows, float* mean,
                                           float* invstd, float* running_mean, float* running_var,
                                           int64_t* num_batches_tracked, int C, float momentum, float eps, void* stream);
/* ttts_bn_train_stats_from_partials_rows for TWO row sets of one matrix in one launch (the halves of a twin batch); the running
 * statistics and the batch counter receive set a's update first, then set b's -- two calls in that order, one launch (ABI v14) */
int ttts_bn_train_stats_twin(const float* partials_a, int nblk_a, const float* rows_a, int n_rows_a, float* mean_a, float* invstd_a,
                             const float* partials_b, int nblk_b, const float* rows_b, int n_rows_b, float* mean_b, float* invstd_b,
                             float* running_mean, float* running_var, int64_t* num_batches_tracked, int C, float momentum,
                             float eps, void* stream);
int ttts_bn_eval_stats(const float* running_mean, const float* running_var, float* mean, float* invstd, int C, float eps,
                       void* stream);
/* z = drop(act((x - mean) * invstd * gamma + beta)); z_amax_out: NULL, or a caller-zeroed TTTS_AMAX_SLOTS-float array receiving max|z| */
int ttts_bn_apply_fwd(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                      float* z, int64_t M, int C, int act, float drop_p, uint64_t seed, const uint64_t* step_seed,
                      float* z_amax_out, void* stream);
/* backward through drop/act/BN: dx, dgamma, dbeta from dz and the saved x, mean, invstd.  batch_stats = 1: train mode (mean /
 * invstd are the batch statistics of x, whose derivative dx carries); 0: eval mode (running statistics, constants). */
int ttts_bn_bwd(const float* dz, const float* x, const float* mean, const float* invstd, const float* gamma,
                const float* beta, float* dx, float* dgamma, float* dbeta, float* ws, size_t ws_bytes, int64_t M, int C,
                int act, float drop_p, uint64_t seed, const uint64_t* step_seed, int accumulate, float* dx_amax_out,
                int batch_stats, void* stream);

/* ------------------------------------------------------------------ LayerNorm over the last dim (1 <= d <= 1024)
 * Replaces nn.LayerNorm norm1/2/3 of the encoder/decoder layers (torch/nn/modules/transformer.py:951-956,
 * model/layers.py:47-50).  The residual sum is produced by the preceding GEMM's epilogue. */
/* y_image_out / y_row_inv_out (ABI v10): NULL, or the activation image of y and its per-row inverse scales for the GEMMs that read
 * y (ttts_linear_fwd_h3i; M * d * 4 bytes and M floats; d in {256, 512, 1024}): the row is complete in the kernel's registers,
 * so the split costs one extra write and no pass. */
int ttts_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       int64_t M, int d, float eps, float* y_amax_out /* NULL, or zeroed TTTS_AMAX_SLOTS floats: max|y| */,
                       void* y_image_out, float* y_row_inv_out, void* stream);
size_t ttts_layernorm_bwd_workspace_bytes(int d);
int ttts_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       float* dx, float* dgamma, float* dbeta, float* ws, size_t ws_bytes, int64_t M, int d,
                       int accumulate, void* dx_image_out /* NULL, or the image of dx (ttts_linear_bwd_data_h3i) */,
                       float* dx_row_inv_out, ttts_reduce_queue* queue, void* stream);
/* the same, and in the same pass dacc = dx * keep(seed, element) / (1 - drop_p): the gradient behind the residual dropout
 * of the sublayer whose output this LayerNorm normalised (ttts_dropout_bwd(dx) without a pass of its own; same mask as
 * the forward epilogue of that sublayer's last Linear).  dacc_amax: NULL, or a caller-zeroed TTTS_AMAX_SLOTS-float array that receives
 * partial maxima of |dacc|.  d in {256, 512, 1024}, operands 16-byte aligned. */
int ttts_layernorm_bwd_drop(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                            float* dx, float* dgamma, float* dbeta, float* ws, size_t ws_bytes, int64_t M, int d,
                            int accumulate, float* dacc, float drop_p, uint64_t seed, const uint64_t* step_seed,
                            float* dacc_amax, void* dacc_image_out /* NULL, or the image of dacc */, float* dacc_row_inv_out,
                            ttts_reduce_queue* queue, void* stream);

/* ------------------------------------------------------------------ attention (64 columns per head)
 * Scaled dot-product attention with masks computed from lengths in-kernel (no mask tensors):
 * key j of batch b is dead when j >= key_lens[b], or (causal) j > i.  q is multiplied by q_scale before q.k^T exactly as
 * torch does (q * sqrt(1 / head_dim), torch/nn/functional.py:6578): 0.125 for head_dim 64.  q/k/v/o are addressed as
 * ptr[(b*T + t)*ld + h*64 + c], so packed in-proj outputs of 64-wide heads are consumed in place; narrower heads
 * (head_dim < 64) are zero-padded to 64 columns by ttts_heads_pad (zeros add nothing to q.k^T and produce zero output
 * columns) and passed with q_scale = sqrt(1 / head_dim).
 * Replaces: encoder self-attention and decoder `_sa_block` (torch/nn/modules/transformer.py:961-978,
 * 1158-1175 -> F.scaled_dot_product_attention, torch/nn/functional.py:6629) and the decoder's
 * `_mha_block` cross-attention with need_weights=True, average_attn_weights=False
 * (model/layers.py:54-74; torch/nn/functional.py:6576-6610).
 * attn (optional) receives the per-head, post-dropout weights (B,H,Tq,Tk); lse (B,H,Tq) is saved for backward. */
/* head_dim < 64: dst (rows, H*64) = the H heads of src (rows x ld_src, head h at column h*head_dim) zero-padded to 64
 * columns each; ttts_heads_unpad writes the first head_dim columns of every 64-wide head of src (rows, H*64) back to
 * dst (rows x ld_dst, head h at column h*head_dim), leaving the other columns of dst untouched. */
int ttts_heads_pad(const float* src, int64_t ld_src, float* dst, int64_t rows, int H, int head_dim, void* stream);
int ttts_heads_unpad(const float* src, float* dst, int64_t ld_dst, int64_t rows, int H, int head_dim, void* stream);
int ttts_attention_fwd(const float* q, const float* k, const float* v, float* o, float* lse, float* attn,
                       const int64_t* key_lens, int B, int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo,
                       int causal, float q_scale, float drop_p, uint64_t seed, const uint64_t* step_seed, void* stream);
/* dq, dk, dv from do_ (recomputes the probabilities from q, k and lse); delta (B,H,Tq) is scratch */
int ttts_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* do_,
                       const float* lse, float* delta, float* dq, float* dk, float* dv, const int64_t* key_lens, int B,
                       int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv,
                       int causal, float q_scale, float drop_p, uint64_t seed, const uint64_t* step_seed, void* stream);
/* Split-precision forms of the two entry points above (bf16x6 MFMA products, fp32-grade results; same arguments). */
int ttts_attention_fwd_x6(const float* q, const float* k, const float* v, float* o, float* lse, float* attn,
                          const int64_t* key_lens, int B, int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo,
                          int causal, float q_scale, float drop_p, uint64_t seed, const uint64_t* step_seed, void* stream);
/* fp16x3 form of the forward (three f16 MFMA terms; Q/8, K and V pre-scaled from their partial maxima q_amax / k_amax /
 * v_amax -- TTTS_AMAX_SLOTS floats each, the same array three times for a packed projection output -- and probabilities x 2^10);
 * lse in natural units, so either backward form can follow it.  o_amax_out: NULL, or zeroed TTTS_AMAX_SLOTS floats: max|o|.
 * rowstat_out: NULL, or (3, B, H, Tq) floats (ABI v10: a third plane): per query row the subtrahend of the weights' exponents
 * exactly as the kernel used it with the final row maximum (one rounded product, in the kernel's own pre-scaled units), log2
 * of the row sum of those weights -- what ttts_attention_bwd_h3 needs to form the very probabilities the forward formed (it
 * re-creates bit-identical score accumulators and the same exponents, whatever the scores' magnitude; from lse alone the
 * probabilities are only good to ulp(lse), i.e. 6 % at scores of 1e6, where torch -- which keeps the probabilities -- is
 * still accurate) -- and 1.0 / 0.0: the row is ONE-HOT in fp32 (its sum is its largest term).  torch's softmax backward of
 * such a row is exactly zero; a backward that recomputes the probabilities would return the rounding of delta against dP
 * instead, so ttts_attention_bwd_h3 takes the row's score gradient as zero. */
int ttts_attention_fwd_h3(const float* q, const float* k, const float* v, float* o, float* lse, float* attn,
                          const int64_t* key_lens, int B, int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo,
                          int causal, float q_scale, float drop_p, uint64_t seed, const uint64_t* step_seed, const float* q_amax,
                          const float* k_amax, const float* v_amax, float* o_amax_out, float* rowstat_out, void* stream);
int ttts_attention_bwd_x6(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                          const float* lse, float* delta, float* dq, float* dk, float* dv, const int64_t* key_lens, int B,
                          int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv,
                          int causal, float q_scale, float drop_p, uint64_t seed, const uint64_t* step_seed, void* stream);
/* fp16x3 form of the backward.  d_o is pre-scaled by the power of two that puts max|d_o| in [2^11, 2^12) (do_amax = the
 * TTTS_AMAX_SLOTS partial maxima of ttts_amax_partials(d_o)); dS = P (dP - delta) lives in registers as a lane-local accumulator
 * column and gets a lane-local pre-scale that is lowered on the fly together with its accumulator. */
int ttts_attention_bwd_h3(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                          const float* lse, float* delta, float* dq, float* dk, float* dv, const int64_t* key_lens, int B,
                          int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv,
                          int causal, float q_scale, float drop_p, uint64_t seed, const uint64_t* step_seed, const float* do_amax,
                          float* dq_amax_out, float* dkv_amax_out, const float* q_amax, const float* k_amax,
                          const float* v_amax, const float* rowstat /* of the h3 forward, or NULL: use lse */, void* stream);
/* ---- attention on head-image operands (ABI v11; attention_img.hip): the same call sites as ttts_attention_fwd_h3 /
 * ttts_attention_bwd_h3 (encoder self-attention and decoder `_sa_block`, torch/nn/modules/transformer.py:961-978,1158-1175 ->
 * torch/nn/functional.py:6629; the decoder's `_mha_block`, model/layers.py:54-74), head_dim 64.  q / k / v point at head 0 of
 * their section inside a head image (row strides ld* in 4-byte cells; a packed q/k/v image is passed as three pointers d cells
 * apart), *_inv at the [head][rows] inverse scales of that section (B * Tq rows per head plane on the q side, B * Tk on the key
 * side), v_amax at the TTTS_AMAX_SLOTS partial maxima of |v| (the section array ttts_linear_fwd_h3d_img filled).  o, lse, attn,
 * o_amax_out as ttts_attention_fwd_h3.  rowstat_out: (5, B, H, Tq) floats -- the three planes of ttts_attention_fwd_h3 plus
 * the query row's exponent multiplier 2^-e_q * q_scale * log2(e) and score multiplier 2^-e_q * q_scale; the backward
 * requires it (it forms the forward's own probabilities from bit-identical accumulators).  q_inv_rows / k_inv_rows / stat_plane (ABI v12): rows per head plane of the inverse scales and
 * floats per plane of the row statistics -- 0: B * Tq, B * Tk, B * H * Tq; larger when the batch is the FIRST part of a bigger
 * image (the encoder of both forwards of a training step run as one batch: ops.twin). */
int ttts_attention_fwd_img(const void* q, const void* k, const void* v, const float* q_inv, const float* k_inv,
                           const float* v_inv, float* o, float* lse, float* attn, const int64_t* key_lens, int B, int H, int Tq,
                           int Tk, int ldq, int ldk, int ldv, int ldo, int causal, float q_scale, float drop_p, uint64_t seed,
                           const uint64_t* step_seed, const float* v_amax, float* o_amax_out, float* rowstat_out, int64_t q_inv_rows, int64_t k_inv_rows, int64_t stat_plane,
                           void* stream);
/* dq, dk, dv in fp32 (strides ldd*) from d_o; o / d_o fp32; do_amax = partial maxima of |d_o|; delta (B,H,Tq) is scratch.
 * q_splits > 1 (dk / dv the two halves of one packed (B, Tk, 2 H 64) gradient; causal too since round 6): the dK / dV kernel splits the query
 * range over q_splits workgroups per key block, partial sums in dkv_partials, one fixed-order reduction at the end. */
int ttts_attention_bwd_img(const void* q, const void* k, const void* v, const float* q_inv, const float* k_inv,
                           const float* v_inv, const float* o, const float* d_o, const float* rowstat, float* delta, float* dq,
                           float* dk, float* dv, const int64_t* key_lens, int B, int H, int Tq, int Tk, int ldq, int ldk, int ldv,
                           int ldo, int lddq, int lddk, int lddv, int causal, float q_scale, float drop_p, uint64_t seed,
                           const uint64_t* step_seed, const float* do_amax, float* dq_amax_out, float* dkv_amax_out,
                           float* dkv_partials /* NULL, or q_splits x B x Tk x lddk floats */, int q_splits /* 1: no split */,
                           int64_t q_inv_rows, int64_t k_inv_rows, int64_t stat_plane, void* stream);

/* ------------------------------------------------------------------ small row / element-wise pieces
 * nn.Embedding gather / scatter-add (model/model.py:168,288; no padding_idx) */
int ttts_embedding_fwd(const int64_t* ids, const float* table, float* out, int64_t n, int vocab, int d,
                       float* out_amax_out /* NULL, or zeroed TTTS_AMAX_SLOTS floats: max|out| */, void* stream);
int ttts_embedding_bwd(const int64_t* ids, const float* dout, float* dtable, int64_t n, int vocab, int d, int accumulate,
                       void* stream);
/* PositionalEncoding.forward (model/model.py:91-97): y = drop(x + alpha * pe[t]) */
int ttts_posenc_fwd(const float* x, const float* pe, const float* alpha, float* y, int B, int T, int d, float drop_p,
                    uint64_t seed, const uint64_t* step_seed, float* y_amax_out /* NULL, or zeroed TTTS_AMAX_SLOTS floats */, void* stream);
size_t ttts_posenc_bwd_workspace_bytes(void);
int ttts_posenc_bwd(const float* dy, const float* pe, float* dx, float* dalpha, float* ws, size_t ws_bytes, int B, int T,
                    int d, float drop_p, uint64_t seed, const uint64_t* step_seed, int accumulate, void* stream);
/* dx = dy * 1[out > 0] / (1-p): backward of drop(relu(.)) given the forward output.  dx_amax_out (NULL, or zeroed
 * TTTS_AMAX_SLOTS floats): also leave the partial maxima of |dx| behind -- the dynamic pre-scale input of
 * the fp16x3 gradient GEMMs that consume dx -- without a second pass over it. */
int ttts_relu_dropout_bwd(const float* dy, const float* out, float* dx, int64_t n, float drop_p, float* dx_amax_out,
                          void* stream);
/* dx = dy * keep(seed, i) / (1-p); dx_amax_out as above */
int ttts_dropout_bwd(const float* dy, float* dx, int64_t n, float drop_p, uint64_t seed, const uint64_t* step_seed,
                     float* dx_amax_out, void* stream);
/* p[0 .. nbytes) = 0 by a fill kernel on the stream (the flat gradient bucket and the partial-maxima arrays at the start of
 * a step; p and nbytes multiples of 4).  Deliberately not hipMemsetAsync: see csrc/elementwise.hip. */
int ttts_zero(void* p, size_t nbytes, void* stream);
/* z = x + y */
int ttts_add(const float* x, const float* y, float* z, int64_t n, void* stream);
/* z = (x + y) + w: the gradient of a tensor with three consumers in one pass (autograd's own accumulation is one add per
 * extra consumer); n % 4 == 0 */
int ttts_add3(const float* x, const float* y, const float* w, float* z, int64_t n, void* stream);
/* ---- input side: device-side padding of a ragged batch --------------------------------------
 * Replaces the host loop of collate_fn (dataset.py:76-91) and the per-sample transpose of __getitem__
 * (dataset.py:64).  `ragged` holds the B utterances back to back, each in its on-disk (n_mels, len_b) row-major
 * layout (preprocess.py:36-42); frame_offsets (B+1, int64, device) are prefix sums of the lengths in frames.
 * out (B, Tmax, n_mels) fp32 is fully written (frames >= len_b are zero).  1 <= n_mels <= 128. */
int ttts_collate_melspec(const float* ragged, const int64_t* frame_offsets, float* out, int B, int Tmax, int n_mels,
                         void* stream);
/* phoneme ids: ragged int64 ids back to back, offsets (B+1); out (B, Pmax) int64, padded with 0 (dataset.py:79,88) */
int ttts_collate_phoneme(const int64_t* ragged, const int64_t* offsets, int64_t* out, int B, int Pmax, void* stream);

/* stop-token head, LinearNorm(d_model, 1) (model/model.py:226,313): y[m] = x[m,:].w + b, and its backward */
int ttts_rowdot_fwd(const float* x, const float* w, const float* b, float* y, int64_t M, int d, void* stream);
size_t ttts_rowdot_bwd_workspace_bytes(int d);
int ttts_rowdot_bwd(const float* dy, const float* x, const float* w, float* dx_accum, float* dw, float* db, float* ws,
                    size_t ws_bytes, int64_t M, int d, int accumulate, ttts_reduce_queue* queue, void* stream);

/* ------------------------------------------------------------------ loss and scheduled-sampling mix
 * TransformerTTSLoss.forward (loss.py:15-55): out4 = [total, pred_mel, post_mel, stop]; masked MSE over frames
 * t < lens[b] (x2, post weighted 0.5 in the total) + BCE-with-logits on the stop gate (1 at t = lens[b]-1) with
 * pos_weight, mean over valid frames.  ws keeps the normalisers for ttts_loss_bwd, which takes the upstream gradients
 * of the four outputs as device scalars (NULL = no gradient for that output) and writes the gradients of pred / post /
 * stop. */
size_t ttts_loss_workspace_bytes(void);
int ttts_loss_fwd(const float* pred, const float* post, const float* stop, const float* mel, const int64_t* lens,
                  float* out4, float* ws, size_t ws_bytes, int B, int T, int C, float pos_weight, void* stream);
int ttts_loss_bwd(const float* pred, const float* post, const float* stop, const float* mel, const int64_t* lens,
                  const float* ws, const float* g_total, const float* g_pred_mel, const float* g_post_mel, const float* g_stop,
                  float* dpred, float* dpost, float* dstop, int B, int T, int C, float pos_weight, void* stream);
/* block_mask + apply_teacher_forcing (utils/util.py:103-120): u is the (B,T) uniform draw; frame t takes the model's
 * prediction when any u[t-l_bar/2 .. t-l_bar/2+l_bar-1] < 1-p_tf (max_pool1d(k=l_bar, s=1, pad=l_bar/2)[:T]), the
 * ground truth otherwise, and zero beyond lens[b].  u == NULL: the draw (torch.rand of utils/util.py:108) is generated
 * in the kernel from `seed` (a counter-based uniform per frame).  st != NULL: p_tf and the seed word of the step are
 * read from device memory when the kernel runs (p_tf by value is ignored).
 * out_amax_out: NULL, or a caller-zeroed TTTS_AMAX_SLOTS-float array receiving max|out|. */
int ttts_sched_sampling_mix(const float* pred, const float* mel, const float* u, const int64_t* lens, float* out, int B,
                            int T, int C, float p_tf, int l_bar, uint64_t seed, const ttts_step_state* st,
                            float* out_amax_out, void* stream);

/* ------------------------------------------------------------------ optimizer over flat buffers
 * Global L2 norm of the flat gradient bucket (clip_grad_norm_, train.py:41) and one torch.optim.Adam step
 * (lightning_module.py:160-163: betas (0.9, 0.98), eps 1e-9, lr = Noam factor set by the host) on flat parameter /
 * gradient / moment buffers; the clip factor min(1, max_grad_norm / (norm + 1e-6)) is applied inside the step.
 * st != NULL: lr and step are read from the device-side step state when the kernel runs (graph replay). */
size_t ttts_grad_norm_workspace_bytes(void);
int ttts_grad_norm(const float* g, float* norm_out, float* ws, size_t ws_bytes, int64_t n, void* stream);
int ttts_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, const float* grad_norm, int64_t n, float lr,
                   float beta1, float beta2, float eps, int64_t step, float max_grad_norm, const ttts_step_state* st,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TTTS_HIP_H */
