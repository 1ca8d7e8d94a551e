/* libembnet_hip.so — C ABI of the MI355X (gfx950) metric-learning hot path.
 *
 * The reference (RocketFlash/EmbeddingNet) is pure Python on Keras: it has no FFI
 * or operator-plugin interface.  Its boundary for this path is the Python call
 * surface that tools/train.py uses; embeddingnet_amd/ mirrors that
 * surface and lowers it onto the entry points below.  Each entry point names the
 * reference code it stands in for (paths relative to the reference root).
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer (HBM) unless marked "host";
 *     tensors are dense fp32, images/feature maps NHWC, conv kernels [R,S,Cin,Cout]
 *     (Keras HWIO), dense kernels [in,out];
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     nothing synchronises, nothing allocates: the caller owns outputs and workspaces
 *     (sizes from the matching *_workspace_bytes call);
 *   - return 0 on success; <0 on error (EMBNET_E*), message via embnet_last_error()
 *     (thread-local).  Invalid arguments are rejected before anything is launched;
 *   - safe to call from several host threads on distinct streams/workspaces.
 */
#ifndef EMBNET_H
#define EMBNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMBNET_ABI_VERSION 22

enum {
  EMBNET_MINE_SEMIHARD = 0,    /* datagenerators.py:196-199 */
  EMBNET_MINE_HARDEST = 1,     /* datagenerators.py:188-190 */
  EMBNET_MINE_RANDOM_HARD = 2, /* datagenerators.py:192-194 */
  EMBNET_MINE_BATCH_HARD = 3   /* Hermans et al. (README.md:112 cites it; not in the reference's code): fused entry point only */
};

int embnet_abi_version(void);
const char* embnet_last_error(void);

/* Optional per-kernel timing (bench.py's roofline leg; build-defined, nothing in the reference to stand in for).
 * While enabled, every kernel launch of the entry points below is bracketed by two HIP events on the launch stream
 * and logged with the kernel's name and its ALGORITHMIC work: unit 0 = FLOP (MFMA-bound kernels: 2*M*N*K of the
 * GEMM the kernel computes), unit 1 = bytes (HBM-bound kernels: the bytes the operation must read and write once,
 * DESIGN.md §3.2); `bytes` is that byte count for every kernel (for a conv: input + kernel + output, once each).
 * embnet_trace_get waits for record i's end event and returns its duration; name/ms/work/unit/bytes are HOST
 * pointers.  Disabled (the default) a launch pays one flag test. */
int embnet_trace_enable(int on);          /* returns the previous state; the log is kept until embnet_trace_reset */
int embnet_trace_reset(void);
int embnet_trace_count(void);
int embnet_trace_get(int i, char* name, int name_cap, float* ms, double* work, int* unit, double* bytes);

/* ------------------------------------------------------------------ loss path */

/* datagenerators.py:219 `pairwise_distances(all_embeddings)` (scikit-learn euclidean):
 * dist[n,n] = sqrt(max(|x_i|^2 + |x_j|^2 - 2 x_i.x_j, 0)), diagonal 0; squared!=0 skips the sqrt.
 * x[n,e].  workspace >= embnet_pairwise_workspace_bytes(n, e) (row norms + the partial Gram slabs of a K-split launch: a
 * batch of n < 512 rows at a long e, e.g. the reference's default encodings_len = 4096, is too few tiles to fill the chip
 * otherwise; a workspace of only n floats is accepted and runs unsplit). */
size_t embnet_pairwise_workspace_bytes(int n, int e);
int embnet_pairwise_dist_f32(const float* x, int n, int e, float* dist, int squared,
                             void* workspace, size_t workspace_bytes, void* stream);

/* datagenerators.py:225-250 (+ selection rules :188-199): mine (anchor,positive,negative) row
 * indices from dist[n,n], n = p*k, class c = rows [c*k,(c+1)*k).  Pairs are visited in the
 * reference's order; triplets[T,3] is written in that order, T -> *count (>=1: the reference's
 * fallback triplet (n-2,n-1,0) when nothing qualifies).  triplets must hold
 * embnet_mine_max_triplets(p,k) rows; selected[pairs] receives the chosen negative row or -1;
 * cand_mask (optional, [pairs][ceil((n-k)/32)] words) receives the candidate set of each pair as
 * a bitmask over the ascending out-of-class columns.  seed drives the two random rules. */
int embnet_mine_max_triplets(int p, int k);
int embnet_mine_triplets(const float* dist, int p, int k, float margin, int mode, uint64_t seed,
                         int32_t* triplets, int32_t* count, int32_t* selected, uint32_t* cand_mask,
                         void* stream);

/* Hermans batch-hard selection (README.md:112 cites it; no reference code): per anchor the
 * farthest same-class and the closest other-class row.  triplets[n,3]; *count = n (optional). */
int embnet_batch_hard(const float* dist, int p, int k, int32_t* triplets, int32_t* count, void* stream);

/* losses_and_accuracies.py:26-42 loss_function(y_true, y_pred): y_pred[t,3e] = concat(a,p,n);
 * loss[t] = max(|a-p|^2 - |a-n|^2 + margin, 0).  bwd: dy[t,3e] from dloss[t]. */
int embnet_triplet_hinge_fwd(const float* y_pred, int t, int e, float margin, float* loss, void* stream);
int embnet_triplet_hinge_bwd(const float* y_pred, const float* dloss, int t, int e, float margin,
                             float* dy, void* stream);

/* Same hinge on rows gathered from emb[n,e] by triplets[.,3] (fused train step; replaces the
 * three-branch concat of models.py:181-185 + the loss).  *count triplets are live (device
 * scalar, no host sync); loss[max_t], active[max_t] (1 where the hinge passes gradient);
 * *mean_loss = mean over the live triplets (Keras' reduction).  bwd: demb[n,e] =
 * (*upstream / count) * d(sum loss)/d emb, upstream NULL = 1. */
int embnet_triplet_gather_fwd(const float* emb, int n, int e, const int32_t* triplets,
                              const int32_t* count, int max_t, float margin, float* loss,
                              float* active, float* mean_loss, void* stream);
int embnet_triplet_gather_bwd(const float* emb, int n, int e, const int32_t* triplets,
                              const int32_t* count, int max_t, const float* active,
                              const float* upstream, float* demb, void* stream);

/* losses_and_accuracies.py:4-11 contrastive_loss (margin 1, y=1 same class): *loss = mean(...).
 * bwd: ddist[b] = *upstream * d loss / d dist. */
int embnet_contrastive_fwd(const float* y_true, const float* dist, int b, float* loss, void* stream);
int embnet_contrastive_bwd(const float* y_true, const float* dist, int b, const float* upstream,
                           float* ddist, void* stream);

/* losses_and_accuracies.py:47-50 accuracy: *acc = mean((dist < 0.5) == y_true). */
int embnet_accuracy(const float* y_true, const float* dist, int b, float* acc, void* stream);

/* backbones.py:38/:77/:118 K.l2_normalize(x, axis=1): y = x * rsqrt(max(sum x^2, 1e-12)).
 * rnorm[n] is saved for backward. */
int embnet_l2norm_fwd(const float* x, int n, int e, float* y, float* rnorm, void* stream);
int embnet_l2norm_bwd(const float* y, const float* rnorm, const float* dy, int n, int e, float* dx,
                      void* stream);

/* models.py:225 siamese 'l2' head: dist[b] = sqrt(max(sum (e1-e2)^2, 1e-7)). */
int embnet_pair_distance_fwd(const float* e1, const float* e2, int b, int e, float* dist, void* stream);
int embnet_pair_distance_bwd(const float* e1, const float* e2, const float* dist, const float* ddist,
                             int b, int e, float* de1, float* de2, void* stream);

/* Softmax pre-training head (backbones.py:128-204: Dense(n_classes, softmax), loss 'categorical_crossentropy',
 * metric 'accuracy'): logits/targets[b,c] (targets one-hot or soft); prob[b,c] kept for backward;
 * *mean_loss = mean_b(-sum_c t log softmax), *accuracy = mean(argmax z == argmax t). */
int embnet_softmax_xent_fwd(const float* logits, const float* targets, int b, int c, float* prob, float* row_loss,
                            float* row_correct, float* mean_loss, float* accuracy, void* stream);
int embnet_softmax_xent_bwd(const float* prob, const float* targets, int b, int c, const float* upstream,
                            float* dlogits, void* stream);

/* ---- evaluation right after training (models.py:128-161 predict_knn / calculate_prediction_accuracy, which
 * go through sklearn KNeighborsClassifier, brute-force Euclidean) ---- */
/* dist[nq,n] = Euclidean distance of every query row q[nq,e] to every gallery row x[n,e] (sklearn
 * euclidean_distances(Q, X) semantics; squared!=0 skips the sqrt). */
size_t embnet_cross_dist_workspace_bytes(int nq, int n);
int embnet_cross_dist_f32(const float* q, int nq, const float* x, int n, int e, float* dist, int squared,
                          void* workspace, size_t workspace_bytes, void* stream);
/* idx/val[rows,k] = the k smallest entries of each row of dist[rows,n], ascending, ties to the smaller column. */
int embnet_topk_smallest(const float* dist, int rows, int n, int k, int32_t* idx, float* val, void* stream);
/* pred[rows] = majority label among labels[idx[row,:]], ties to the smallest label (KNeighborsClassifier.predict). */
int embnet_knn_vote(const int32_t* idx, const int32_t* labels, int rows, int k, int32_t* pred, void* stream);

/* ------------------------------------------------------------------ backbone layers
 * Stand-ins for the Keras layers that backbones.py:19-121 instantiates (TensorFlow kernels in the
 * reference).  All NHWC fp32. */

/* Conv2D (backbones.py:21-31, :44-68, zoo ResNet/EfficientNet convs): implicit GEMM.  fp32 in, fp32 out, fp32
 * accumulation; each fp32 product is formed on the bf16 matrix instruction from an EXACT three-way split of both
 * operands by truncation (x = x1 + x2 + x3, 8 significant bits each) as six of the nine cross terms.
 * ERROR BOUND.  The three dropped terms x2*y3 + x3*y2 + x3*y3 are <= 2^-20 |x||y| in the worst case (the low 16 mantissa
 * bits of both operands all ones; typically 2^-22) and have the product's sign, so on top of fp32 accumulation a result is
 * LOW by at most 2^-20 * sum|x||y| — a bias, not zero-mean noise.  Measured on that adversarial input with all-positive
 * operands (tests/test_round3_gpu.py::test_conv_split_worst_case, profiles/r03_split_worst_case_*.json), relative to
 * sum|x||y|:  K = 576: 1.5e-6 max / -4.4e-7 mean (a float32 CPU convolution on the same input: 2.1e-6 / -9.4e-7);
 * K = 4608 forward: 3.3e-5 / -2.2e-5 (float32 CPU: 1.9e-5 / -1.4e-5: fp32 accumulation dominates there), data gradient
 * 2.3e-6 (6.7e-6), weight gradient 5.8e-7 (1.1e-7).  On operands with random mantissas the error is that of the fp32
 * MFMA's k-ordered fma chain (test_conv2d_products_are_fp32_accurate); on integer-valued operands results are exact.
 * The three bf16 pieces of THIS (six-term) form keep fp32's exponent range (no overflow / underflow beyond fp32's own); the
 * two-piece fp16 form of the *_ex entry points and of the planes kernels does not — its bound is stated under PRECISION below.  INFINITY: an infinite operand yields
 * NaN where an fp32 product would yield inf (inf - inf inside the split) — training has diverged by then either way.
 * embnet_conv_mfma_terms() = bf16 MFMA terms per product in this build (6), or 1 for a build on v_mfma_f32_32x32x2_f32.
 * x[n,h,w,c], w[r,s,c,k], y[n,oh,ow,k]; taps outside the image read 0 (pad_t/pad_l = top/left
 * padding; bottom/right follow from oh/ow, which the caller computes: Keras 'valid', 'same' incl. its
 * bottom/right asymmetry, or an explicit ZeroPadding2D).  bias may be NULL; relu!=0 fuses the activation;
 * residual (NULL or [n,oh,ow,k]) is added last — the Add layer that closes a residual unit.
 * in_scale/in_shift [c] (NULL or both) + in_act (0 none, 1 relu, 2 swish): the conv reads
 * act(x*in_scale + in_shift) instead of x — the BatchNormalization(+activation) in front of it, applied in
 * registers between the gather and LDS so the normalised tensor is never written; padding stays 0.
 * Needs c % 4 == 0 and k % 4 == 0.  conv2d_wgrad takes the same triple (it re-derives the conv input).
 * stats (NULL or [2][k][P], P = embnet_conv2d_fwd_stats_rows): per-channel sum and sum of squares of y, by row
 * band, written by the epilogue while the tile is in registers — the statistics pass of the BatchNormalization
 * that follows (hand them to embnet_bn_train_fwd as partial_in).
 * workspace (optional, may be NULL/0): >= embnet_conv2d_fwd_workspace_bytes lets the launcher cut the
 * `tiles mod 256` left-over output tiles along K so the last round of workgroups fills every CU
 * (partial tiles + fixed-order fix-up; results differ from the unsplit launch only in fp32 summation order). */
int embnet_conv_mfma_terms(void);
/* ... per fp32 product in the kernels that read pre-split planes (embnet_conv2d_patch_f32, embnet_conv2d_wgrad_planes_f32):
 * 3 in the default planes format (two fp16 pieces + a power-of-two scale per tensor), 6 with EMBNET_PLANES_F16=0 (three bf16 pieces). */
int embnet_conv_planes_mfma_terms(void);
size_t embnet_conv2d_fwd_workspace_bytes(int n, int c, int r, int s, int k, int oh, int ow);
int embnet_conv2d_fwd_stats_rows(int n, int c, int r, int s, int k, int oh, int ow);
int embnet_conv2d_fwd_f32(const float* x, const float* w, const float* bias, float* y, int n, int h, int wd, int c,
                          int r, int s, int k, int stride, int pad_t, int pad_l, int oh, int ow, int relu,
                          const float* residual, const float* in_scale, const float* in_shift, int in_act,
                          float* stats, void* workspace, size_t workspace_bytes, void* stream);
/* dx[n,h,w,c] from dy[n,oh,ow,k] (gradient w.r.t. the conv input).  workspace as for fwd (stride 1 only).
 * accumulate != 0: dx += instead of dx = (the input feeds two convs, e.g. a residual unit's 3x3 and its 1x1
 * projection shortcut: the second dgrad adds in its epilogue and skips pixels no tap reaches).
 * dx_add (NULL or [n,h,w,c]): dx = result + dx_add — the gradient reaching x through a skip connection. */
size_t embnet_conv2d_dgrad_workspace_bytes(int n, int h, int wd, int c, int r, int s, int k, int stride);
int embnet_conv2d_dgrad_f32(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int r, int s,
                            int k, int stride, int pad_t, int pad_l, int oh, int ow, int accumulate, const float* dx_add,
                            void* workspace, size_t workspace_bytes, void* stream);
/* Data gradient that also emits the BatchNorm-backward sums of the layer that produced this conv's input: the conv read
 * a = act(bn_x*bn_scale + bn_shift) (a BatchNormalization with a fused activation, /root/reference/embedding_net/backbones.py:99-104:
 * the zoo ResNets' pre-activation units), so  dz = dx * act'(..)  is that layer's output gradient; bn_partial [2][c][bn_rows]
 * (bn_rows = embnet_conv2d_dgrad_bnsums_rows(...), 0 = geometry not supported: stride 1, c % 4 == 0, k % 4 == 0) receives the
 * per-row-band sums of dz and dz * (bn_x - bn_mean) * bn_rstd — what embnet_bn_bwd's reduction pass would compute from two
 * tensors — for embnet_bn_bwd_partials.  Everything else as embnet_conv2d_dgrad_f32 without accumulate / dx_add. */
int embnet_conv2d_dgrad_bnsums_rows(int n, int h, int wd, int c, int r, int s, int k, int stride);
int embnet_conv2d_dgrad_bnsums_f32(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int r, int s,
                                   int k, int stride, int pad_t, int pad_l, int oh, int ow, const float* bn_x,
                                   const float* bn_scale, const float* bn_shift, const float* bn_mean, const float* bn_rstd,
                                   int bn_act, float* bn_partial, int bn_rows, void* workspace, size_t workspace_bytes,
                                   void* stream);
/* dw[r,s,c,k]; split-K slabs live in `workspace` (>= embnet_conv2d_wgrad_workspace_bytes). */
size_t embnet_conv2d_wgrad_workspace_bytes(int n, int c, int r, int s, int k, int oh, int ow);
int embnet_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                            int n, int h, int wd, int c, int r, int s, int k, int stride, int pad_t, int pad_l,
                            int oh, int ow, const float* in_scale, const float* in_shift, int in_act, void* stream);

/* conv2d_wgrad in two calls (split-K MFMA kernel into the slabs; then the fixed-order slab sum), same
 * arguments: lets a caller time the MFMA kernel alone.  conv2d_wgrad_f32 == slabs + reduce. */
int embnet_conv2d_wgrad_slabs_f32(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                                  int n, int h, int wd, int c, int r, int s, int k, int stride, int pad_t, int pad_l,
                                  int oh, int ow, const float* in_scale, const float* in_shift, int in_act,
                                  void* stream);
int embnet_conv2d_wgrad_reduce_f32(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                                   int n, int h, int wd, int c, int r, int s, int k, int stride, int pad_t, int pad_l,
                                   int oh, int ow, const float* in_scale, const float* in_shift, int in_act,
                                   void* stream);
/* Split count of conv2d_wgrad's plan for a geometry (1 = the kernel writes dw itself, no slabs). */
int embnet_conv2d_wgrad_splits(int n, int c, int r, int s, int k, int oh, int ow);
/* The slab sums of MANY weight gradients in one launch (a step's backward runs embnet_conv2d_wgrad_slabs_f32 into
 * per-layer slab buffers and adds them up once, before the optimizer / a bucket's all-reduce).
 * host_table: HOST array of n_tensors descriptors { const float* slabs; float* out; int64 n; int32 splits; int32 reserved }
 * (32 bytes; device pointers inside): out[i] = sum_s slabs[s*n + i], in embnet_conv2d_wgrad_reduce_f32's order
 * (bit-identical).  The descriptors travel as kernel arguments (112 per launch), so nothing has to stay alive for a
 * captured graph.  slabs and out 16-byte aligned when n % 4 == 0. */
int embnet_slab_reduce_multi(const void* host_table, int n_tensors, void* stream);

/* ---- three products per fp32 product for the implicit-GEMM ("gather") convolutions above (ABI 21 / 22; csrc/conv.hip "Ranges") -------
 * The layers of backbones.py:99-104 the patch kernels do not take — the 7x7 stem, the stride-2 3x3 convs, every 1x1 conv — split
 * their fp32 operands inside the kernel.  Given the RANGE of BOTH operands they use the planes kernels' arithmetic (below: two fp16
 * pieces of x s, three products, the sums x 1 / s x 1 / s') instead of three bf16 pieces and six products: half the matrix
 * instructions.  Without either range a call computes the six-term products: there is NO default scale (ABI 20 ran activations at
 * s = 1, which loses precision for tensors of amplitude << 2^-3 and clamps above 65504; see PRECISION under the planes format).
 * A RANGE SLOT is one uint32 in device memory holding the bit pattern of a float B >= max |element| of a tensor (0: an all-zero
 * tensor).  Who writes it:
 *   kernels      embnet_range_multi: the exact maxima of many tensors (a model's kernels, once per optimizer step) in two launches;
 *                device table of 24-byte rows { const float* x; int64 n; uint32* slot; }, chunk list int32 [n_chunks][2] = (row,
 *                chunk of embnet_range_chunk_elems() elements);
 *   gradients    `dx_range` of embnet_bn_bwd_ex / embnet_bn_bwd_partials_ex / embnet_bn_act_maxpool_bwd_ex: the exact max |dx| of the
 *                fp32 dx the call writes.  The slot has embnet_range_slot_words() words: the workgroups of the apply pass join their
 *                maxima into the words behind the first (zeroed by the finalize kernel; an order-independent unsigned maximum, spread
 *                over many words because same-address atomics serialise) and a one-workgroup launch folds them into word 0.  The
 *                call fails if it cannot emit (planes-only dx, c % 4 != 0, no saved statistics);
 *   activations  `y_range` of embnet_bn_train_fwd_ex / embnet_affine_act_planes_ex: an UPPER BOUND of max |act(BN(x))|, known before
 *                the apply pass runs — per channel |scale_c| sqrt(q_c) + |shift_c| with q_c the largest per-band sum of squares
 *                among the statistics partials (>= max x^2), folded over the channels; above the true maximum by at most
 *                sqrt(rows per band) (3.3 binades for a conv epilogue's bands), never below it.
 *   without a BatchNormalization (ABI 22; the `simple` / `simple2` backbones, reference backbones.py:19-81): the exact maximum of what
 *                the pass writes, a slot of embnet_range_slot_words() words as for gradients — `y_range` of embnet_pad_channels_ex
 *                (the image batch) and embnet_maxpool_fwd_ex (pooled activations), `dz_range` of embnet_maxpool_relu_bwd_colsum_ex,
 *                embnet_relu_bwd_colsum_ex and embnet_bn_bwd_inrelu[_dropout]_ex (the gradient behind a fused ReLU's mask).  These
 *                passes zero the slot with a KERNEL: a hipMemsetAsync node did not zero it when a captured HIP graph was replayed.
 * Who reads it: the last two pointer arguments (in front of `stream`) of embnet_conv2d_{fwd, dgrad, dgrad_bnsums, wgrad,
 * wgrad_slabs}_f32_ex — the ranges of the call's first and second tensor argument (x, w | dy, w | x, dy); NULL = unknown.  The
 * scale of a tensor puts B into [2^14, 2^15).  The scalar-load kernels (c or k % 4 != 0), a fused input transform and the thin
 * 1x1 streams compute as if no range had been given.  Results differ from the six-term kernels' in the last bits
 * (tests/test_conv_ranges_gpu.py: both against float64, activation amplitudes 1e-4 ... 1e5).
 * DEPRECATED, kept for one round: embnet_range_emit(slot) and embnet_conv2d_ranges(a, b) arm the same slots for the NEXT non-_ex
 * call of the calling thread (ABI 20) — hidden per-thread state: a binding in another language, a second stream on one thread or
 * an exception between the two calls gets the wrong arithmetic silently.  Every conv / bn_bwd entry point clears the request. */
/* out = the range slot of a tensor y with |y| <= factor * max_c bound[c] (+ the range in add_range, or NULL): for the outputs of
 * the fused BatchNorm passes that are not plain applies (act(BN(x)) * gate; skip + drop_factor * BN(x)); bound from embnet_bn_train_fwd_ex. */
int embnet_range_from_bound(const float* bound, int c, float factor, const uint32_t* add_range, uint32_t* out, void* stream);
int embnet_range_slot_words(void);
int embnet_range_emit(uint32_t* slot);
int embnet_range_chunk_elems(void);
int embnet_range_multi(const void* table, int n_tensors, const int32_t* chunks, int n_chunks, void* stream);
int embnet_conv2d_ranges(const uint32_t* a, const uint32_t* b);
int embnet_conv2d_fwd_f32_ex(const float* x, const float* w, const float* bias, float* y, int n, int h, int wd, int c,
                             int r, int s, int k, int stride, int pad_t, int pad_l, int oh, int ow, int relu,
                             const float* residual, const float* in_scale, const float* in_shift, int in_act,
                             float* stats, void* workspace, size_t workspace_bytes, const uint32_t* x_range,
                             const uint32_t* w_range, void* stream);
int embnet_conv2d_dgrad_f32_ex(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int r, int s,
                               int k, int stride, int pad_t, int pad_l, int oh, int ow, int accumulate, const float* dx_add,
                               void* workspace, size_t workspace_bytes, const uint32_t* dy_range, const uint32_t* w_range,
                               void* stream);
/* (the _ex form of dgrad_bnsums writes bn_partial as [3][c][bn_rows]: the two sums and max |dz| per row band) */
int embnet_conv2d_dgrad_bnsums_f32_ex(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int r, int s,
                                      int k, int stride, int pad_t, int pad_l, int oh, int ow, const float* bn_x,
                                      const float* bn_scale, const float* bn_shift, const float* bn_mean, const float* bn_rstd,
                                      int bn_act, float* bn_partial, int bn_rows, void* workspace, size_t workspace_bytes,
                                      const uint32_t* dy_range, const uint32_t* w_range, void* stream);
int embnet_conv2d_wgrad_f32_ex(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                               int n, int h, int wd, int c, int r, int s, int k, int stride, int pad_t, int pad_l,
                               int oh, int ow, const float* in_scale, const float* in_shift, int in_act,
                               const uint32_t* x_range, const uint32_t* dy_range, void* stream);
int embnet_conv2d_wgrad_slabs_f32_ex(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                                     int n, int h, int wd, int c, int r, int s, int k, int stride, int pad_t, int pad_l,
                                     int oh, int ow, const float* in_scale, const float* in_shift, int in_act,
                                     const uint32_t* x_range, const uint32_t* dy_range, void* stream);

/* Which kernel symbol (as rocprofv3 names it) the conv entry points launch for a geometry:
 * kind 0 = fwd, 1 = dgrad, 2 = wgrad.  Host-only helper for attributing timings. */
const char* embnet_conv2d_kernel_name(int kind, int n, int h, int wd, int c, int r, int s, int k, int oh, int ow);

/* 1x1 convolutions with a thin reduction (csrc/conv_thin.hip): embnet_conv2d_fwd_f32 with r = s = 1, pad 0, c <= 40 (EfficientNet's
 * expand convs) and embnet_conv2d_dgrad_f32 with r = s = 1, stride 1, k <= 40 (its project convs' data gradient) run as an HBM
 * stream — exact fp32 FMA chains over the reduction, the dense output written as one contiguous stream, statistics rows as
 * embnet_conv2d_fwd_stats_rows reports — instead of an implicit GEMM with a one-tile K loop; embnet_conv2d_wgrad_f32 with
 * r = s = 1, pad 0 and min(c, k) in {4, 8, ..., 20, 24, 32, 40} likewise (one slab per workgroup; embnet_conv2d_wgrad_splits /
 * _workspace_bytes report this path's counts).  1 when (reduction, output columns) takes the forward / data-gradient path
 * (EMBNET_CONV_THIN=0 turns all of it off, EMBNET_CONV_THIN_WGRAD=0 the weight gradient). */
int embnet_conv1x1_thin_supported(int red, int ncols);

/* ---- stride-1 3x3 convolution on pre-split operands ("patch" kernel, csrc/conv_patch.hip) --------------------------------
 * The zoo ResNets' 3x3 stride-1 layers (backbones.py:99-104), forward and data gradient, 1.1-1.25x faster than the kernels
 * above: every fp32 operand value is split into its three bf16 pieces ONCE, by the kernel that produces the tensor, and the
 * convolution keeps a patch of the padded input in LDS so that each input value is fetched once per output tile instead of
 * once per tap.  Summation order (chunk, r, s, channel).
 * FORMAT OF THE PLANES (one per process, csrc/common.h planes_f16()).  Default (ABI 19, end of round 5): TWO fp16 pieces of x s,
 * s a power of two per tensor — x = (h1 + h2) / s, h1 = fp16(x s) rounded to nearest, h2 = fp16(x s - h1): 22 mantissa bits —
 * and THREE products per fp32 product (h1 h1' + h1 h2' + h2 h1', v_mfma_f32_32x32x16_f16), the result x 1 / (s s') (exact).
 * Error on top of fp32 accumulation <= 2^-21 sum|x||y| (each operand kept to 2^-23 of itself, the dropped product <= 2^-22);
 * measured on the adversarial input above 4.1e-7 / 3.0e-7 / 1.7e-7 (forward / data gradient / weight gradient, K = 4608;
 * tests/test_round3_gpu.py::test_planes_split_worst_case, profiles/r05_split_worst_case_planes_*.json).
 * PRECISION.  s puts a bound B >= max |x| of the tensor into [2^14, 2^15).  Elements with |x s| >= 2^-3 (within 2^17 of B) are
 * kept to 2^-22 of themselves; below, h2 is an fp16 subnormal and the element carries an ABSOLUTE error <= 2^-25 / s <= 2^-39 B
 * — not a relative one: a tensor is only as precise as its largest element allows.  |x s| > 65504 cannot happen with a sound B and is
 * not clamped: it overflows to inf / NaN, loudly.  Where B comes from: the tensor's own largest element for kernels
 * (embnet_conv_weight_planes: a max pass) and gradients (embnet_bn_bwd's dx_planes: a dry run of the pass; embnet_planes_from_f32:
 * an abs-max pass); for activations the BatchNormalization's output bound (embnet_affine_act_planes_ex `y_bound`: RANGE SLOT above)
 * or, without one, a dry run of the pass.  (ABI 20 and earlier fixed s = 1 for activations: tensors of amplitude << 2^-3 sat on
 * the subnormal floor — 1.8e-4 relative error at amplitude 1e-4)
 * The buffers keep the three-plane size below: planes 0 and 1 hold the pieces, the first two floats of the third plane's space
 * hold (s, 1 / s) (and a scratch word).  EMBNET_PLANES_F16=0: three bf16 pieces (exact split by truncation) and the six-term products documented above.
 *   planes of an activation / gradient x[pixels, c] (c % 16 == 0):  16-bit [3][c/16][pixels][16]  (piece, 16-channel chunk,
 *     pixel, channel in chunk) — written by embnet_affine_act_planes, by embnet_bn_bwd(dx_planes), or from an fp32 tensor by
 *     embnet_planes_from_f32;
 *   planes of a kernel w[r,s,c,k]:  16-bit [3][r][red/16][s][rows][16] — flip 0 (forward: rows = k, reduction channels = c) or
 *     flip 1 (stride-1 data gradient: rows = c, reduction = k, taps flipped) — written for any number of kernels by ONE launch
 *     of embnet_conv_weight_planes over a device table of 40-byte descriptors
 *       { const float* w; void* out; int32 r, s, c, k, flip, pad; }   and a chunk list int32 [n_chunks][2] =
 *     (tensor index, chunk of embnet_conv_weight_planes_chunk_elems() elements), typically once per optimizer step.
 * embnet_conv2d_patch_f32: y[n,oh,ow,k] = conv(x planes, w planes) with the epilogue options of embnet_conv2d_fwd_f32 (bias,
 * relu, residual, stats with P = embnet_conv2d_patch_stats_rows) — and, with dy planes, flip-1 kernel planes, c and k
 * swapped and pad = kernel - 1 - pad, the data gradient (residual = the gradient of the tensor's other consumer).
 * embnet_conv2d_patch_supported: 1 for 3x3, stride 1, c % 16 == 0, k % 4 == 0 and a patch that fits LDS. */
int embnet_conv2d_patch_supported(int n, int c, int r, int s, int k, int stride, int oh, int ow);
size_t embnet_conv2d_patch_workspace_bytes(int n, int c, int r, int s, int k, int oh, int ow);
int embnet_conv2d_patch_stats_rows(int n, int oh, int ow);
int embnet_conv2d_patch_f32(const void* x_planes, const void* w_planes, const float* bias, float* y, int n, int h, int wd, int c,
                            int r, int s, int k, int pad_t, int pad_l, int oh, int ow, int relu, const float* residual,
                            float* stats, void* workspace, size_t workspace_bytes, void* stream);
/* ABI 22 — the zoo ResNets' stem (7x7, stride 2, 4 -> 64 channels; reference backbones.py:99-104 via image-classifiers) forward as a
 * kernel of its own (csrc/conv_stem.hip): x fp32 [n,h,wd,4] (the image widened to four channels), w fp32 [7,7,4,64], y [n,oh,ow,64];
 * taps outside the image read zeros (pad_t / pad_l: where output (0, 0)'s window starts above / left of the image).  Every input
 * pixel reaches LDS once per 16 x 16 output tile (LDS-DMA), the kernel is split into its fp16 pieces by each workgroup, the products
 * are the three-term form: BOTH range slots are required.  stats (NULL or [2][64][embnet_conv2d_stem_stats_rows(n, oh, ow)]): the
 * BatchNorm statistics partials of the layer behind, as embnet_conv2d_fwd_f32's.  The gather kernel's result to fp32 rounding. */
int embnet_conv2d_stem_supported(int n, int h, int wd, int c, int r, int s, int k, int stride, int pad_t, int pad_l, int oh, int ow);
int embnet_conv2d_stem_stats_rows(int n, int oh, int ow);
int embnet_conv2d_stem_f32(const float* x, const float* w, float* y, int n, int h, int wd, int pad_t, int pad_l, int oh, int ow,
                           float* stats, const uint32_t* x_range, const uint32_t* w_range, void* stream);
/* 1x1 convolutions on the planes (ABI 21; csrc/conv_patch.hip conv1x1_planes_kernel): the bottleneck units' conv1 / conv3 and the
 * projection shortcuts (backbones.py:99-104) as a GEMM whose operands arrive by LDS-DMA from the planes — x planes of [n,h,wd,c],
 * kernel planes of a [1,1,c,k] kernel (embnet_conv_weight_planes; flip 1 and dy planes give the stride-1 data gradient) —
 * y[n,oh,ow,k] = sum_c x[n, oh*stride, ow*stride, c] w[c,k], epilogue options and workspace as embnet_conv2d_patch_f32
 * (embnet_conv2d_patch_supported / _workspace_bytes / _stats_rows with r = s = 1: c % 32 == 0, k % 4 == 0, stride 1 or 2, the
 * two-piece planes format). */
int embnet_conv2d_planes1x1_f32(const void* x_planes, const void* w_planes, const float* bias, float* y, int n, int h, int wd, int c,
                                int k, int stride, int oh, int ow, int relu, const float* residual, float* stats,
                                void* workspace, size_t workspace_bytes, void* stream);
/* ABI 22 — the same 1x1 product with the ACTIVATION operand read as fp32 (csrc/conv_patch.hip conv1x1_a32_kernel): no planes of x
 * have to exist.  x fp32 [n,h,wd,c] reaches LDS by DMA (16-byte pieces, one 128-byte line per pixel and 32-channel step, three stages
 * in flight) and is split into its two fp16 pieces by the matrix waves; x_range: x's RANGE SLOT (required — its scale); wp: the kernel
 * planes of embnet_conv_weight_planes (r = s = 1; flip 0).  With dy, dy's range slot, the flip-1 planes and c, k swapped (stride 1) the
 * call is the DATA GRADIENT (residual = a gradient to add).  c % 32 == 0, k % 4 == 0, stride 1 or 2, the tensor below 4 GiB:
 * embnet_conv2d_dma1x1_supported.  Workspace: embnet_conv2d_patch_workspace_bytes(n, c, 1, 1, k, oh, ow).  The Python layers leave it
 * off by default (layers.CONV1X1_DMA; DESIGN 3.14: 1.1 - 1.3 x the gather kernel back to back, no gain inside the HBM-bound C3 step). */
int embnet_conv2d_dma1x1_supported(int n, int h, int wd, int c, int k, int stride, int oh, int ow);
int embnet_conv2d_dma1x1_f32(const float* x, const void* wp, const float* bias, float* y, int n, int h, int wd, int c, int k,
                             int stride, int oh, int ow, int relu, const float* residual, float* stats, const uint32_t* x_range,
                             void* workspace, size_t workspace_bytes, void* stream);
/* The patch kernel as a stride-1 data gradient (dy planes, flip-1 kernel planes, c and k swapped as above) whose epilogue ALSO
 * emits the BatchNorm-backward sums of the BatchNormalization in front of the conv — the conv's input was
 * act(bn_scale * bn_x + bn_shift): embnet_conv2d_dgrad_bnsums_f32's contract on this kernel, bn_partial [2][k][bn_rows] with
 * bn_rows = embnet_conv2d_patch_stats_rows(n, oh, ow); embnet_bn_bwd_partials consumes it (ABI 19). */
int embnet_conv2d_patch_bnsums_f32(const void* x_planes, const void* w_planes, float* y, int n, int h, int wd, int c, int r, int s,
                                   int k, int pad_t, int pad_l, int oh, int ow, const float* bn_x, const float* bn_scale,
                                   const float* bn_shift, const float* bn_mean, const float* bn_rstd, int bn_act,
                                   float* bn_partial, int bn_rows, void* workspace, size_t workspace_bytes, void* stream);
/* The squeeze-and-excite gate of an MBConv block, gate = sigmoid(swish(pooled w1 + b1) w2 + b2) (pooled [n,c], w1 [c,s], w2 [s,c]),
 * as one forward launch (z1 [n,s]: the first layer's pre-activation, kept for backward) and two backward launches (dz1 [n,s]:
 * scratch; writes dpooled [n,c], dw1 [c,s], db1 [s], dw2 [s,c], db2 [c]; sums over the samples in sample order) instead of
 * twelve dense / activation / column-sum launches.  embnet_se_mlp_supported: 1 where they apply (s <= 160, LDS). */
int embnet_se_mlp_supported(int n, int c, int s);
int embnet_se_mlp_fwd(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, int n, int c, int s,
                      float* z1, float* gate, void* stream);
int embnet_se_mlp_bwd(const float* dgate, const float* gate, const float* z1, const float* pooled, const float* w1, const float* w2,
                      int n, int c, int s, float* dz1, float* dpooled, float* dw1, float* db1, float* dw2, float* db2, void* stream);
int embnet_planes_from_f32(const float* x, long pixels, int c, void* planes, void* stream);
int embnet_conv_weight_planes_chunk_elems(void);
int embnet_conv_weight_planes(const void* table, int n_tensors, const int32_t* chunks, int n_chunks, void* stream);
/* y (NULL or [m,c]) = act(x*scale + shift) as embnet_affine_act, AND the same values as planes: the BatchNormalization in
 * front of a patch convolution writes the convolution's operand in its final form.
 * _ex: y_bound (NULL or [c]: the per-channel bounds embnet_bn_train_fwd_ex left) gives the planes' scale without another look at
 * the data; NULL (and the non-_ex form): a dry run of the pass finds the exact maximum first (+ 4 B per element read).
 * y_range (NULL or a range slot, needs y): the bound (or the exact maximum) as the range slot of the fp32 y. */
int embnet_affine_act_planes(const float* x, long m, int c, const float* scale, const float* shift, int act, float* y,
                             void* planes, void* stream);
int embnet_affine_act_planes_ex(const float* x, long m, int c, const float* scale, const float* shift, int act, float* y,
                                void* planes, const float* y_bound, uint32_t* y_range, void* stream);
/* Weight gradient of a 3x3, stride-1, pad-1 ('same') convolution FROM THE PLANES (csrc/conv_wgrad_planes.hip):
 * dw[3,3,c,k] = sum over pixels of x[n,h,wd,c] (shifted by the tap) * dy[n,h,wd,k], both operands as planes (layout above).
 * Every operand byte is fetched once per (64 channels x 64 filters) tile and all nine taps are accumulated from one window
 * of positions in LDS; split over pixel ranges into fp32 slabs [splits][9*c*k] in `workspace` and summed in fixed order
 * (bitwise reproducible).  reduce = 0 leaves the slabs for embnet_slab_reduce_multi (descriptor: workspace, dw, 9*c*k,
 * embnet_conv2d_wgrad_planes_splits).  Products in the planes' format (above); against embnet_conv2d_wgrad_f32 the order of the
 * pixel sum differs too, so results agree within fp32 rounding, not bitwise.  supported: 1 for r = s = 3, stride 1, pad 1, oh = h,
 * ow = wd <= 62, c % 64 == 0, k % 64 == 0. */
int embnet_conv2d_wgrad_planes_supported(int n, int h, int wd, int c, int r, int s, int k, int stride, int pad_t, int pad_l,
                                         int oh, int ow);
int embnet_conv2d_wgrad_planes_splits(int n, int h, int wd, int c, int k);
size_t embnet_conv2d_wgrad_planes_workspace_bytes(int n, int h, int wd, int c, int k);
int embnet_conv2d_wgrad_planes_f32(const void* x_planes, const void* dy_planes, float* dw, void* workspace,
                                   size_t workspace_bytes, int n, int h, int wd, int c, int k, int reduce, void* stream);

/* Input pipeline, last step (datagenerators.py:145-156: cv2 decode -> resize -> `/ 255.` on the host, per batch): uint8
 * images [n_src, pixels, c_in] (BGR as decoded) -> float32 batch dst[n, pixels, c_out]:
 *   dst[i][p][j] = j < c_in ? src[index ? index[i] : i][p][j] / denom : 0      (float32 division: the reference's values bit for bit)
 * index (NULL = the first n images in order) gathers a batch out of a dataset that is resident in HBM as uint8; c_out > c_in
 * pads channels with zeros (the 4-channel image the stem convolution gathers 16 bytes at a time).  n <= 65535. */
int embnet_u8_to_f32(const void* src, const int32_t* index, int n, long pixels, int c_in, int c_out, float denom, float* dst,
                     void* stream);

/* Dense (backbones.py:35,72,75,114,116; models.py:44): x[m,in], w[in,out], y[m,out].
 * workspace (optional, may be NULL/0): >= embnet_dense_fwd_workspace_bytes lets a forward with few output tiles and a long
 * reduction (simple2's Flatten -> Dense(512): 12 800 x 512 at batch 32) cut K over workgroups (partial slabs + fixed-order
 * sum with bias / ReLU); 0 bytes = this shape runs unsplit. */
size_t embnet_dense_fwd_workspace_bytes(int m, int in, int out);
int embnet_dense_fwd_f32(const float* x, const float* w, const float* bias, float* y, int m, int in, int out,
                         int relu, void* workspace, size_t workspace_bytes, void* stream);
int embnet_dense_dgrad_f32(const float* dy, const float* w, float* dx, int m, int in, int out, void* stream);
int embnet_dense_wgrad_f32(const float* x, const float* dy, float* dw, int m, int in, int out, void* stream);

/* BatchNormalization over the last axis of x[m,c] (backbones.py:46-69; zoo ResNet BN with eps 2e-5).
 * train: biased batch variance; moving <- momentum*moving + (1-momentum)*batch (updated in place, may
 * be NULL); gamma NULL = scale=False; relu = fused activation after the affine: 0 none, 1 ReLU, 2 swish
 * (backward recomputes it from x, nothing extra is stored).  save_mean/save_rstd/scale/shift
 * ([c] each) are kept by the caller for backward.  y may be NULL (statistics and scale/shift only, for a
 * fused consumer).  partial_in (NULL, or [2][c][partial_rows] sums / sums of squares of x over disjoint row
 * sets, e.g. from conv2d_fwd's epilogue) replaces the kernel's own read of x for the statistics. */
size_t embnet_bn_workspace_bytes(long m, int c);
int embnet_bn_train_fwd(const float* x, long m, int c, const float* gamma, const float* beta, float eps,
                        float momentum, int relu, float* y, float* save_mean, float* save_rstd, float* scale,
                        float* shift, float* moving_mean, float* moving_var, const float* partial_in,
                        int partial_rows, void* workspace, size_t workspace_bytes, void* stream);
/* _ex: y_bound (NULL or [c]) receives, per channel, an upper bound of |act(scale_c x + shift_c)| over the batch (RANGE SLOT above);
 * y_range (NULL or a range slot; needs y_bound and y): their maximum, as the range slot of y;  xhat_bound (NULL or [c]): an upper
 * bound of |x - mean_c| rstd_c — what embnet_bn_bwd_ex needs to bound its dx without a dry run. */
int embnet_bn_train_fwd_ex(const float* x, long m, int c, const float* gamma, const float* beta, float eps,
                           float momentum, int relu, float* y, float* save_mean, float* save_rstd, float* scale,
                           float* shift, float* moving_mean, float* moving_var, const float* partial_in,
                           int partial_rows, void* workspace, size_t workspace_bytes, float* y_bound, uint32_t* y_range,
                           float* xhat_bound, void* stream);
int embnet_bn_infer_fwd(const float* x, long m, int c, const float* gamma, const float* beta,
                        const float* moving_mean, const float* moving_var, float eps, int relu, float* y,
                        float* scale, float* shift, void* stream);
/* dx_add (NULL or [m,c]) is added to dx: the gradient that reaches x through its other consumer (the identity
 * shortcut of a residual unit), so autograd needs no separate accumulation pass. */
int embnet_bn_bwd(const float* dy, const float* x, long m, int c, const float* save_mean, const float* save_rstd,
                  const float* scale, const float* shift, int relu, int training, const float* dx_add, float* dx,
                  float* dgamma, float* dbeta, void* dx_planes, void* workspace, size_t workspace_bytes, void* stream);
/* _ex: dx_range (NULL or a range slot of embnet_range_slot_words() words) receives max |dx| of the fp32 dx (RANGE SLOT) — exact, or,
 * where dx is also written as planes from a bound, that bound.
 * xhat_bound ([c] from embnet_bn_train_fwd_ex, or NULL), dx_add_range (the range slot of dx_add, or NULL): with dx_planes in the
 * two-piece format the planes' scale then comes from an upper bound of |dx| —
 *   |scale_c| (max |dz| + |dbeta_c| / m + xhat_bound_c |dgamma_c| / m)  [+ the range of dx_add],
 * max |dz| per channel being a third output of the reduction pass — instead of from a DRY RUN of the apply pass (8 B per element
 * read once more: round 5's form, still taken when either pointer is missing; EMBNET_BN_BWD_BOUND=0 forces it). */
int embnet_bn_bwd_ex(const float* dy, const float* x, long m, int c, const float* save_mean, const float* save_rstd,
                     const float* scale, const float* shift, int relu, int training, const float* dx_add, float* dx,
                     float* dgamma, float* dbeta, void* dx_planes, void* workspace, size_t workspace_bytes, uint32_t* dx_range,
                     const float* xhat_bound, const uint32_t* dx_add_range, void* stream);
/* BatchNorm backward of a layer whose output is also globally average-pooled (the squeeze-and-excite block of the EfficientNet
 * MBConv, reference backbones.py:84-98): the output gradient is dy[n,p,c] + dpool[n,c] / hw; both passes form it on the fly with
 * embnet_gap_bwd's arithmetic (the result of embnet_gap_bwd(dx_add = dy) followed by embnet_bn_bwd to the last bits), so the
 * summed tensor is never written.  gate (NULL or [n,c]): dy is the gradient of the GATED tensor y * gate (embnet_channel_scale_fwd)
 * and the scaling's backward multiply is applied here too — its caller then only needs embnet_channel_scale_dgate.
 * Training statistics, c % 4 == 0; workspace as embnet_bn_bwd. */
int embnet_bn_bwd_gap(const float* dy, const float* dpool, const float* gate, int n, int hw, const float* x, int c, const float* save_mean,
                      const float* save_rstd, const float* scale, const float* shift, int relu, float* dx, float* dgamma,
                      float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
/* Squeeze-and-excite backward without a BatchNorm reduction pass.  embnet_se_bn_sums: ONE pass over dg = d(gated tensor) and the
 * BatchNormalization's input x gives sums[n][5][c]: row 0 = the gate's gradient sum_p dg * act(BN(x)) (what
 * embnet_channel_scale_dgate computes from the stored activation), rows 1..4 = sum_p of a' dg, a', a' dg xhat, a' xhat
 * (a' = act'(BN(x))).  embnet_bn_bwd_gap_sums: dbeta / dgamma from those rows, the gate and the pooled gradient (the layer's
 * output gradient is a' (dg gate + dpool / hw), linear in the two per-(n,c) factors), then embnet_bn_bwd_gap's apply pass. */
int embnet_se_bn_sums(const float* dg, const float* x, int n, int hw, int c, const float* save_mean, const float* save_rstd,
                      const float* scale, const float* shift, int act, float* sums, void* stream);
int embnet_bn_bwd_gap_sums(const float* dy, const float* dpool, const float* gate, const float* sums, int n, int hw, const float* x,
                           int c, const float* save_mean, const float* save_rstd, const float* scale, const float* shift, int relu,
                           float* dx, float* dgamma, float* dbeta, void* stream);
/* BatchNorm backward whose column sums came from the data gradient of the conv that consumed this layer's output
 * (embnet_conv2d_dgrad_bnsums_f32 below): partials [2][c][rows] -> dbeta / dgamma (added in double), then the apply pass of
 * embnet_bn_bwd.  Training statistics, c % 4 == 0. */
int embnet_bn_bwd_partials(const float* dy, const float* x, long m, int c, const float* save_mean, const float* save_rstd,
                           const float* scale, const float* shift, int relu, const float* partials, int rows,
                           const float* dx_add, float* dx, float* dgamma, float* dbeta, void* dx_planes, void* stream);
/* _ex: partial_kinds = 3: partials is [3][c][rows], the third plane holding max |dz| per row band (what
 * embnet_conv2d_dgrad_bnsums_f32_ex writes): with xhat_bound / dx_add_range as for embnet_bn_bwd_ex, no dry run. */
int embnet_bn_bwd_partials_ex(const float* dy, const float* x, long m, int c, const float* save_mean, const float* save_rstd,
                              const float* scale, const float* shift, int relu, const float* partials, int rows,
                              const float* dx_add, float* dx, float* dgamma, float* dbeta, void* dx_planes, uint32_t* dx_range,
                              int partial_kinds, const float* xhat_bound, const uint32_t* dx_add_range, void* stream);
/* BatchNorm backward for a BN whose input x is the output of a layer with a fused ReLU (conv -> ReLU -> BN, the small
 * backbones' block): dz = d(x) * [x > 0] (the gradient the producer's data / weight gradients consume, ReLU backward
 * included) and dbias[c] = column sums of dz (the producer's bias gradient), in the pass that computes d(x).  c % 4 == 0. */
int embnet_bn_bwd_inrelu(const float* dy, const float* x, long m, int c, const float* save_mean, const float* save_rstd,
                         const float* scale, const float* shift, int relu, int training, float* dz, float* dgamma,
                         float* dbeta, float* dbias, void* workspace, size_t workspace_bytes, void* stream);
/* ABI 22 — the same with `dz_range` (NULL or a RANGE SLOT): the exact max |dz| in its first word, for the producer's data and
 * weight gradient on three products */
int embnet_bn_bwd_inrelu_ex(const float* dy, const float* x, long m, int c, const float* save_mean, const float* save_rstd,
                            const float* scale, const float* shift, int relu, int training, float* dz, float* dgamma,
                            float* dbeta, float* dbias, void* workspace, size_t workspace_bytes, uint32_t* dz_range, void* stream);
/* dx_planes (NULL, or 3*m*c bf16, c % 16 == 0): dx ALSO as the pre-split planes embnet_conv2d_patch_f32 takes (below) —
 * the gradient of the convolution output in front of this BatchNormalization, i.e. that convolution's data-gradient operand.
 * With dx_planes given, dx may be NULL (embnet_bn_bwd, embnet_bn_bwd_partials; c % 4 == 0): the fp32 tensor is not written —
 * for a convolution that takes both its data gradient (embnet_conv2d_patch_f32) and its weight gradient
 * (embnet_conv2d_wgrad_planes_f32) from the planes. */

/* y = act(x*scale[c] + shift[c]) on x[m,c]: the apply half of BatchNormalization on its own (scale/shift from
 * bn_train_fwd / bn_infer_fwd with y = NULL), for a deferred BN output whose consumer cannot fuse it. */
int embnet_affine_act(const float* x, long m, int c, const float* scale, const float* shift, int act, float* y,
                      void* stream);

/* MaxPool2D (backbones.py:23,26,29: 2x2/2; zoo ResNet: ZeroPadding2D(1) + 3x3/2).  Taps outside the
 * image read 0 and take no gradient.  argmax[n,oh,ow,c] (uint8) is kept for backward. */
int embnet_maxpool_fwd(const float* x, int n, int h, int w, int c, int k, int stride, int pad, int oh, int ow,
                       float* y, uint8_t* argmax, void* stream);
/* ABI 22 — the same with `y_range` (NULL, or a RANGE SLOT of embnet_range_slot_words() words, c % 4 == 0): the exact max |y| of the
 * pooled tensor is left in its first word, for the three-product conv that reads y (the `simple` backbone's conv -> ReLU -> pool
 * blocks have no BatchNormalization whose statistics could bound the activation; reference backbones.py:21-31). */
int embnet_maxpool_fwd_ex(const float* x, int n, int h, int w, int c, int k, int stride, int pad, int oh, int ow,
                          float* y, uint8_t* argmax, uint32_t* y_range, void* stream);
int embnet_maxpool_bwd(const float* dy, const uint8_t* argmax, int n, int h, int w, int c, int k, int stride,
                       int pad, int oh, int ow, float* dx, void* stream);
/* embnet_maxpool_bwd followed by embnet_relu_bwd_colsum in one pass, for the conv -> ReLU -> MaxPool blocks of the
 * 'simple' backbone (reference backbones.py:21-31): y is the pool's input (the Conv2D's ReLU output); dz[n,h,w,c] is the
 * gradient w.r.t. the Conv2D's pre-activation, dbias[c] its column sums.  The scattered full-size gradient never exists
 * (8 + 5/stride^2 bytes per element of y instead of 16 + 5/stride^2).  dz equals the two-call result bit for bit; dbias is
 * the same sum in another fp32 order.  c % 4 == 0, n*h*w < 2^31; workspace >= embnet_bn_workspace_bytes(n*h*w, c). */
int embnet_maxpool_relu_bwd_colsum(const float* dy, const uint8_t* argmax, const float* y, int n, int h, int w, int c,
                                   int k, int stride, int pad, int oh, int ow, float* dz, float* dbias, void* workspace,
                                   size_t workspace_bytes, void* stream);
/* ABI 22 — the same with `dz_range` (NULL or a RANGE SLOT): the exact max |dz| in its first word, for the conv's data and
 * weight gradient on three products. */
int embnet_maxpool_relu_bwd_colsum_ex(const float* dy, const uint8_t* argmax, const float* y, int n, int h, int w, int c,
                                      int k, int stride, int pad, int oh, int ow, float* dz, float* dbias, void* workspace,
                                      size_t workspace_bytes, uint32_t* dz_range, void* stream);

/* BatchNorm-apply + activation + ZeroPadding2D(pad) + MaxPool(k,stride) in one pass (the zoo ResNet stem
 * bn0 -> relu -> pad -> pool, reference backbones.py:99-104 via image-classifiers).  scale/shift come from
 * embnet_bn_train_fwd / embnet_bn_infer_fwd called with y = NULL.  Backward takes the POOLED gradient and
 * returns dx (w.r.t. the BN input), dgamma, dbeta; the activation tensor and its gradient never exist.
 * c % 4 == 0.  Same results as bn -> maxpool (sums in a different fp32 order).
 * xwin (optional, [n,oh,ow,c]): forward also stores the BN input at each window's winning tap; handing it to backward
 * turns the dgamma/dbeta reduction's gather through the arg-max into a streaming read (188 -> ~60 us on the ResNet stem). */
int embnet_bn_act_maxpool_fwd(const float* x, int n, int h, int w, int c, const float* scale, const float* shift, int act,
                              int k, int stride, int pad, int oh, int ow, float* y, uint8_t* argmax, float* xwin,
                              void* stream);
size_t embnet_bn_act_maxpool_bwd_workspace_bytes(int n, int oh, int ow, int c);
int embnet_bn_act_maxpool_bwd(const float* dy, const uint8_t* argmax, const float* x, int n, int h, int w, int c, int k,
                              int stride, int pad, int oh, int ow, const float* save_mean, const float* save_rstd,
                              const float* scale, const float* shift, int act, int training, const float* xwin, float* dx,
                              float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
int embnet_bn_act_maxpool_bwd_ex(const float* dy, const uint8_t* argmax, const float* x, int n, int h, int w, int c, int k,
                                 int stride, int pad, int oh, int ow, const float* save_mean, const float* save_rstd,
                                 const float* scale, const float* shift, int act, int training, const float* xwin, float* dx,
                                 float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, uint32_t* dx_range,
                                 void* stream);

/* GlobalAveragePooling2D (backbones.py:111): x[n,hw,c] -> y[n,c]. */
/* act(x*scale + shift) -> y AND its per-image channel means -> gap[n,c], one pass (the BatchNormalization + swish in front
 * of a squeeze-and-excite block, whose pooling reads what the BN writes); scale/shift from embnet_bn_train_fwd(y = NULL). */
/* y = drop_n(x*scale + shift) + skip: the BatchNormalization apply, the per-sample drop-connect (embnet_sample_dropout's mask) and the
 * residual Add of an MBConv block (reference backbones.py:84-98) in one pass; factor [n,c] receives 1/(1-rate) or 0 per sample —
 * the gate of embnet_bn_bwd_gap for the backward (dpool = zeros).  rate = 0: a plain BatchNorm apply + Add. */
int embnet_affine_drop_add(const float* x, int n, int hw, int c, const float* scale, const float* shift, float rate, uint64_t seed,
                           const uint64_t* seed_add_dev, const float* skip, float* y, float* factor, void* stream);
/* y = act(x*scale + shift) * gate[n,c] in one pass (the BatchNormalization apply and the squeeze-and-excite multiply; the pooled
 * means the gate was computed from come from embnet_affine_act_gap with y = NULL, so the activated tensor is never written). */
int embnet_affine_act_scale(const float* x, int n, int hw, int c, const float* scale, const float* shift, int act,
                            const float* gate, float* y, void* stream);
int embnet_affine_act_gap(const float* x, int n, int hw, int c, const float* scale, const float* shift, int act, float* y,
                          float* gap, void* stream);
int embnet_gap_fwd(const float* x, int n, int hw, int c, float* y, void* stream);
int embnet_gap_bwd(const float* dy, int n, int hw, int c, const float* dx_add, float* dx, void* stream);   /* dx_add (NULL or
    [n,hw,c], c % 4 == 0): gradient of x's other consumer, summed in the same pass */

/* Elementwise helpers of the backward pass and the residual blocks. */
int embnet_relu_bwd(const float* dy, const float* y, long total, float* dz, void* stream);   /* dz = dy*[y>0] */
/* the same on dy/y[m,c] plus dbias[c] = column sums of dz, one pass (Conv2D / Dense with bias and a fused ReLU);
 * workspace >= embnet_colsum_workspace_bytes(m, c) */
int embnet_relu_bwd_colsum(const float* dy, const float* y, long m, int c, float* dz, float* dbias, void* workspace,
                           size_t workspace_bytes, void* stream);
/* ABI 22 — the same with `dz_range` (NULL or a RANGE SLOT): the exact max |dz| in its first word */
int embnet_relu_bwd_colsum_ex(const float* dy, const float* y, long m, int c, float* dz, float* dbias, void* workspace,
                              size_t workspace_bytes, uint32_t* dz_range, void* stream);
size_t embnet_colsum_workspace_bytes(long m, int c);
int embnet_colsum(const float* x, long m, int c, float* out, void* workspace, size_t workspace_bytes,
                  void* stream);                                                                /* bias grads */
int embnet_add(const float* a, const float* b, long total, float* y, void* stream);           /* Add() */
int embnet_scale(const float* x, long total, float alpha, const float* alpha_dev, float* y, void* stream);
/* out[c] = sum_{tap,k} w[tap,c,k] * tap_sums[tap,k]: gradient of a per-channel offset added to a conv
 * input (the zoo ResNet's bn_data beta) from per-tap sums of dy — avoids a full 3-channel dgrad. */
int embnet_tap_contract(const float* w, const float* tap_sums, int taps, int c, int k, float* out, void* stream);
/* tap_sums[r,s,k] for embnet_tap_contract when dy[n,oh,ow,k] sums to zero over the pixels of every channel (dy is the
 * data gradient of a training-mode BatchNormalization: the zoo ResNet's bn0 behind conv0): the sum over the pixels
 * whose tap (r,s) is inside the image = minus the sum over those whose tap falls into the padding — border strips
 * only (row sums + column sums - corners), 6 % of the tensor for the 7x7/2 stem.  The caller vouches for the zero-sum
 * property. */
size_t embnet_tap_border_sums_workspace_bytes(int n, int oh, int ow, int k, int r, int s, int stride, int pad_t, int pad_l,
                                              int h, int w);
int embnet_tap_border_sums(const float* dy, int n, int oh, int ow, int k, int r, int s, int stride, int pad_t, int pad_l,
                           int h, int w, float* tap_sums, void* workspace, size_t workspace_bytes, void* stream);
/* y[pixels,cout] = [x[pixels,cin] | 0]: widens 3-channel images to 4 channels for 16-byte stem gathers. */
int embnet_pad_channels(const float* x, long pixels, int cin, int cout, float* y, void* stream);
/* ABI 22 — the same with `y_range` (NULL or a RANGE SLOT): the exact max |x| of the image batch in its first word */
int embnet_pad_channels_ex(const float* x, long pixels, int cin, int cout, float* y, uint32_t* y_range, void* stream);
/* Dropout (backbones.py:53,64,73): inverted scaling, counter-based mask from (seed, index). */
int embnet_dropout(const float* x, long total, float rate, uint64_t seed, const uint64_t* seed_add_dev, float* y,
                   void* stream);   /* seed_add_dev (NULL or device uint64): added to seed — a replayed HIP graph draws a new mask per step */
/* A Dropout layer directly behind a BatchNormalization (simple2's bn3 -> drop1, bn6 -> drop2: backbones.py:52-55,63-66)
 * riding on the BatchNormalization's passes, with embnet_dropout's mask and arithmetic (bit-identical to the two
 * layers run separately):  forward  y = dropout(act(x*scale + shift));  backward  embnet_bn_bwd_inrelu with
 * dy := dropout_backward(dy) applied as dy is read. */
int embnet_affine_act_dropout(const float* x, long m, int c, const float* scale, const float* shift, int act, float rate,
                              uint64_t seed, const uint64_t* seed_add_dev, float* y, void* stream);
int embnet_bn_bwd_inrelu_dropout(const float* dy, const float* x, long m, int c, const float* save_mean,
                                 const float* save_rstd, const float* scale, const float* shift, int relu, int training,
                                 float rate, uint64_t seed, const uint64_t* seed_add_dev, float* dz, float* dgamma,
                                 float* dbeta, float* dbias, void* workspace, size_t workspace_bytes, void* stream);
int embnet_bn_bwd_inrelu_dropout_ex(const float* dy, const float* x, long m, int c, const float* save_mean,
                                    const float* save_rstd, const float* scale, const float* shift, int relu, int training,
                                    float rate, uint64_t seed, const uint64_t* seed_add_dev, float* dz, float* dgamma,
                                    float* dbeta, float* dbias, void* workspace, size_t workspace_bytes, uint32_t* dz_range,
                                    void* stream);                                         /* ABI 22: + dz_range, as above */
/* ---- EfficientNet MBConv pieces (backbones.py:84-98, `efficientnet` zoo package) and the siamese 'l1' head ---- */
/* DepthwiseConv2D: x[n,h,w,c], w[r,s,c] (Keras depthwise_kernel [r,s,c,1]), y[n,oh,ow,c]; padding as conv2d. */
int embnet_dwconv2d_fwd_f32(const float* x, const float* w, float* y, int n, int h, int wd, int c, int r, int s,
                            int stride, int pad_t, int pad_l, int oh, int ow, void* stream);
/* DepthwiseConv2D forward that also writes the statistics partials [2][c][P] (sum, sum of squares of y per channel, one row per
 * workgroup; P = embnet_dwconv2d_fwd_stats_rows(...), 0 = not available for the geometry) of the BatchNormalization that follows
 * (embnet_bn_train_fwd's `partials`): that layer then does not read y for its statistics.  Every (channel, row) is written
 * (since ABI 19; earlier, wide layers — c / 4 > 256 — wanted `stats` zeroed first, which remains harmless). */
int embnet_dwconv2d_fwd_stats_rows(int n, int c, int r, int s, int stride, int oh, int ow);
int embnet_dwconv2d_fwd_stats_f32(const float* x, const float* w, float* y, int n, int h, int wd, int c, int r, int s, int stride,
                                  int pad_t, int pad_l, int oh, int ow, float* stats, void* stream);
int embnet_dwconv2d_dgrad_f32(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int r, int s,
                              int stride, int pad_t, int pad_l, int oh, int ow, void* stream);
/* Depthwise data gradient (stride 1 or 2) that also emits the BatchNorm-backward sums of the layer in front of the depthwise conv (its
 * input was act(bn_x*bn_scale + bn_shift): an MBConv block's expand BatchNormalization), as embnet_conv2d_dgrad_bnsums_f32 does
 * for the gather convs: bn_partial [2][c][bn_rows], bn_rows = embnet_dwconv2d_dgrad_bnsums_rows(...) (0: not available);
 * every (channel, row) of bn_partial is written (ABI 19; zeroing it first, as wide layers once required, stays harmless).  For
 * embnet_bn_bwd_partials. */
int embnet_dwconv2d_dgrad_bnsums_rows(int n, int h, int wd, int c, int r, int s, int stride);
int embnet_dwconv2d_dgrad_bnsums_f32(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int r, int s,
                                     int stride, int pad_t, int pad_l, int oh, int ow, const float* bn_x, const float* bn_scale,
                                     const float* bn_shift, const float* bn_mean, const float* bn_rstd, int bn_act,
                                     float* bn_partial, int bn_rows, void* stream);
size_t embnet_dwconv2d_wgrad_workspace_bytes(int n, int c, int r, int s, int oh, int ow);
int embnet_dwconv2d_wgrad_f32(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                              int n, int h, int wd, int c, int r, int s, int stride, int pad_t, int pad_l, int oh,
                              int ow, void* stream);
/* kind 0 = sigmoid, 1 = swish (x*sigmoid(x)); backward recomputes from x. */
int embnet_activation_fwd(const float* x, long total, int kind, float* y, void* stream);
int embnet_activation_bwd(const float* x, const float* dy, long total, int kind, float* dx, void* stream);
/* squeeze-excite multiply: y[n,p,c] = x[n,p,c]*s[n,c]; bwd gives dx and ds[n,c]. */
int embnet_channel_scale_fwd(const float* x, const float* s, int n, int hw, int c, float* y, void* stream);
/* ds[n,c] = sum_p dy * x only (embnet_channel_scale_bwd without dx: see embnet_bn_bwd_gap's gate).  c % 4 == 0. */
int embnet_channel_scale_dgate(const float* x, const float* dy, int n, int hw, int c, float* ds, void* stream);
int embnet_channel_scale_bwd(const float* x, const float* s, const float* dy, int n, int hw, int c, float* dx,
                             float* ds, void* stream);
/* drop-connect: Dropout with noise_shape (None,1,1,1): one keep/drop decision per sample, inverted scaling. */
int embnet_sample_dropout(const float* x, long total, long per_sample, float rate, uint64_t seed,
                          const uint64_t* seed_add_dev, float* y, void* stream);
/* models.py:218 L1 layer: y = |a - b|. */
int embnet_absdiff_fwd(const float* a, const float* b, long total, float* y, void* stream);
int embnet_absdiff_bwd(const float* a, const float* b, const float* dy, long total, float* da, float* db,
                       void* stream);

/* kernel_regularizer=l2(lambda) (backbones.py:22-36): *out = alpha * sum x^2. */
/* All regularisers of a model in one launch pair: *out = sum_t alpha_t * sum(x_t^2).
 *   table   device array of n_tensors descriptors, 24 bytes each: { const float* x; int64 n; float alpha; int32 pad; }
 *   chunks  device int32 [n_chunks][2] = (tensor index, chunk index), chunk = embnet_sumsq_chunk_elems() elements
 *   workspace >= n_chunks floats. */
int embnet_sumsq_chunk_elems(void);
int embnet_sumsq_multi(const void* table, int n_tensors, const int32_t* chunks, int n_chunks, float* out, void* workspace,
                       size_t workspace_bytes, void* stream);
size_t embnet_sumsq_workspace_bytes(void);
int embnet_sumsq(const float* x, long total, float alpha, float* out, void* workspace, size_t workspace_bytes,
                 void* stream);

/* The same loss path in ONE launch for small batches (the sizes the reference trains at): distances of each class's K
 * anchors to all N = p*k rows (sklearn arithmetic, kept in LDS), mine-and-select (mode as embnet_mine_triplets, or
 * EMBNET_MINE_BATCH_HARD), squared-L2 hinge of each selected triplet, and — by the last workgroup to finish — compaction
 * in pair order, the reference's fallback triplet and the mean.  Stands in for datagenerators.py:219,225-250 +
 * losses_and_accuracies.py:26-42 (+ Keras' mean), i.e. for embnet_pairwise_dist_f32 + embnet_mine_triplets +
 * embnet_triplet_gather_fwd, and writes the same outputs (triplets[T,3], count, selected[pairs], loss[max_t],
 * active[max_t], mean_loss), so embnet_triplet_gather_bwd is the backward of both.  batch-hard: T = N, one triplet per
 * anchor, `selected` unused; triplets/loss/active must hold N rows.
 * embnet_fused_loss_supported: 1 when N <= 512, k <= 16 and k*(e+N) floats fit 64 KiB of LDS.
 * workspace (embnet_fused_loss_workspace_bytes): zero-filled ONCE by the caller; every launch leaves its counter zeroed.
 * seed_dev (NULL or a device uint64): the seed of the random rules read from device memory instead of `seed`, so that
 * a captured HIP graph of the training step draws new negatives at every replay. */
int embnet_fused_loss_supported(int p, int k, int e);
size_t embnet_fused_loss_workspace_bytes(int p, int k);
int embnet_fused_triplet_loss_fwd(const float* emb, int p, int k, int e, float margin, int mode, uint64_t seed,
                                  const uint64_t* seed_dev, int32_t* triplets, int32_t* count, int32_t* selected,
                                  float* loss, float* active, float* mean_loss, void* workspace, size_t workspace_bytes,
                                  void* stream);

/* ------------------------------------------------------------------ optimizer update
 * utils.py:143-153 get_optimizer(name, lr): `Adam(lr)`, `RMSprop(lr)`, `keras_radam.RAdam(lr)`, else `SGD(lr)`
 * with the library defaults — applied by Keras after train.py:160-177's compile/fit.  One launch updates every
 * tensor (multi-tensor apply).
 *   table   device array of n_tensors descriptors, 48 bytes each:
 *             { float* w; const float* g; float* slot1; float* slot2; int64 n; float l2x2; int32 pad; }
 *           g == NULL: the variable got no gradient this step and is skipped (Keras' behaviour);
 *           l2x2 = 2*lambda of the variable's kernel_regularizer=l2(lambda) (backbones.py:22-36), 0 for none: the kernel
 *           uses g + l2x2*w as the gradient (what Keras gets from differentiating loss + regularisers);
 *           slot1/slot2: Adam/RAdam m and v, RMSprop rms (slot2 unused), SGD none — zero-initialised by the caller;
 *   chunks  device int32 [n_chunks][2] = (tensor index, chunk index within it), chunk = embnet_optimizer_chunk_elems()
 *           consecutive elements; every element of every tensor must be covered exactly once;
 *   rule and host-computed scalars (t = 1-based step count):
 *     EMBNET_OPT_SGD         w -= lr*g
 *     EMBNET_OPT_RMSPROP     rms = b1*rms + (1-b1)*g^2;  w -= lr*g/(sqrt(rms)+eps)                 (b1 = rho)
 *     EMBNET_OPT_ADAM        m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g^2;  w -= c1*m/(sqrt(v)+eps),      c1 = lr*sqrt(1-b2^t)/(1-b1^t)
 *     EMBNET_OPT_RADAM       m, v as Adam;  w -= c1*m/(sqrt(v*c2)+eps),    c1 = lr*r_t/(1-b1^t), c2 = 1/(1-b2^t)   (sma_t >= 5)
 *     EMBNET_OPT_RADAM_WARM  m, v as Adam;  w -= c1*m,                     c1 = lr/(1-b1^t)                          (sma_t < 5)
 *   coef_dev (NULL or device float[6] = lr, b1, b2, eps, c1, c2): the scalars read from device memory instead of the
 *           arguments — a captured HIP graph of the training step replays with the step count's current coefficients.
 * HBM-bound: 12 (SGD) .. 28 (Adam/RAdam) bytes per element. */
enum { EMBNET_OPT_SGD = 0, EMBNET_OPT_RMSPROP = 1, EMBNET_OPT_ADAM = 2, EMBNET_OPT_RADAM = 3, EMBNET_OPT_RADAM_WARM = 4 };
int embnet_optimizer_chunk_elems(void);
int embnet_optimizer_step(int rule, const void* table, int n_tensors, const int32_t* chunks, int n_chunks,
                          float lr, float b1, float b2, float eps, float c1, float c2, const float* coef_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EMBNET_H */
