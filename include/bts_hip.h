/* bts_hip.h -- C ABI of the MI355X (gfx950) 3D U-Net+VAE segmentation engine (libbts_hip.so).
 *
 * Every entry point: plain pointers + sizes, returns int status (0 = ok, >0 = hipError_t, <0 = engine
 * code below), enqueues asynchronously on the caller's hipStream_t, never allocates, never synchronises.
 * Activations are fp32 NDHWC with an explicit pixel stride `ld` (floats between consecutive voxels), so
 * a tensor may be a channel slice of a wider slab (virtual Concatenate).
 * Weights are passed in the reference's Keras layouts and re-packed by bts_conv_pack().
 *
 * The reference (vliu15/3d-brain-tumor-segmentation) has no FFI of its own: its boundary is the Keras
 * layer protocol. Each function below cites the reference call site whose arithmetic it replaces.
 */
#ifndef BTS_HIP_H
#define BTS_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* bts_stream_t; /* == hipStream_t */

#define BTS_OK 0
#define BTS_ERR_SHAPE (-1)
#define BTS_ERR_ALIGN (-2)
#define BTS_ERR_UNSUPPORTED (-3)
#define BTS_ERR_WORKSPACE (-4)
#define BTS_ERR_RCCL_BASE (-100) /* bts_dp_*: an RCCL call failed; the code is BTS_ERR_RCCL_BASE - ncclResult_t (-101 ncclUnhandledCudaError,
                                    -102 ncclSystemError, -103 ncclInternalError, -104 ncclInvalidArgument, -105 ncclInvalidUsage, ...) */

/* convolution kinds */
#define BTS_CONV_K1 0    /* Conv3D k=1 s=1            layers/resnet.py:30-37, layers/decoder.py:55-63 */
#define BTS_CONV_K3S1 1  /* Conv3D k=3 s=1 'same'     layers/resnet.py:80-87,96-103, layers/vae.py:92-99 */
#define BTS_CONV_K3S2 2  /* Conv3D k=3 s=2 'same'     layers/downsample.py:28-35 (even sizes: pad (0,1)) */
#define BTS_CONV_K3S2T 3 /* Conv3DTranspose k=3 s=2   layers/upsample.py:28-33 (output 2x, tap at 2n cropped) */
#define BTS_ROLE_FWD 0
#define BTS_ROLE_BWD_DATA 1
#define BTS_CONV_FLAG_SIGMOID 1 /* fused activation='sigmoid' (decoder.py:60) */
#define BTS_CONV_FLAG_ACCUM 2   /* add into the destination instead of overwriting */

/* GroupNormalization modes (layers/group_norm.py:83-124) */
#define BTS_GN_SLAB 0    /* channels_last reference semantics: groups = contiguous 1/G chunks of (D,H,W,C) */
#define BTS_GN_CHANNEL 1 /* channels_first reference semantics (true GroupNorm) on NDHWC memory */

/* ---- declarations are appended below, grouped by reference call site ---- */

/* ===== convolutions (layers/resnet.py:30-37,80-87,96-103; downsample.py:28-35; upsample.py:28-33; decoder.py:55-63; vae.py:92-99) ===== */
/* number of floats of the packed weight image for (kind, role); Cin is the slab (folded) input-channel count.
   K3S1 images hold two parts: the implicit-GEMM layout (27 taps) and the Winograd-domain layout (3 x taps x 16 points) */
long bts_conv_packed_floats(int kind, int role, int Cin, int Cout);
/* w: reference Keras layout ((kd,kh,kw,Cin_ref,Cout) or, for K3S2T, (kd,kh,kw,Cout,Cin_ref)); wp: packed image.
 * Cin_slab + dup_shift == Cin_ref. dup_shift>0 folds the encoder's duplicated dense-connection slice
 * (encoder.py:83-87: inputs [o_{j-1}, o_0..o_{j-1}] -> slab [o_0..o_{j-1}], dup_start = (j-1)*F, dup_shift = F). */
int bts_conv_pack(int kind, int role, const float* w, float* wp, int Cin_ref, int Cout, int Cin_slab, int dup_start,
                  int dup_shift, bts_stream_t stream);
/* All weight images in one launch (every image goes stale together at the optimiser step, train.py:152): the caller
 * fills a HOST table of bts_conv_pack_desc_bytes()-sized descriptors with bts_conv_pack_desc (arguments as bts_conv_pack;
 * returns the entry's block count > 0, or a negative engine code; first_block = running sum of those counts), copies it
 * to device memory and passes both copies with the total block count (table_host may be NULL: see below). */
long bts_conv_pack_desc_bytes(void);
long bts_conv_pack_desc(void* host_table, int index, long first_block, int kind, int role, const float* w, float* wp,
                        int Cin_ref, int Cout, int Cin_slab, int dup_start, int dup_shift);
int bts_conv_pack_batch(const void* table_dev, const void* table_host, int n, long total_blocks, bts_stream_t stream);
/* A BTS_CONV_K3S1 image holds three forms of the same weights (implicit GEMM, F(2x2,3x3) x direct, F(2x2x2,3x3x3)); a layer reads the one
 * its geometry selects.  The library records which forms each image has been read in and bts_conv_pack_desc describes those only (all
 * three for an image never read); a form a launch selects although the last re-pack left it out is packed on the spot on the launch's
 * stream.  The counter below grows whenever an image is read in a form it had not been read in before: a host table built at an older
 * value should be rebuilt at the next re-pack (it stays CORRECT either way).  What counts as written is what the table actually RUN
 * wrote: bts_conv_pack_batch reads the forms of each entry from table_host (the host copy the device table was made from), so several
 * tables over one image, or a bts_conv_pack of the whole image between two runs of a narrower cached table, cannot leave a stale form
 * marked fresh.  table_host == NULL: the library cannot tell what the table covers and treats every form of every registered image as
 * stale (correct; each launch then re-packs its form once per weight change).  bts_conv_pack_forget drops the record of an image whose
 * memory is being released (returns 1 if there was one). */
long bts_conv_pack_generation(void);
int bts_conv_pack_forget(const float* wp);
/* y = act(conv(x) + bias). x (N,D,H,W,Cin) stride ldx; y (N,D',H',W',Cout) stride ldy; D' = D | D/2 | 2D. */
/* workspace (may be NULL / 0) lets grids too small to fill the chip split the contraction over workgroups (deterministic
 * two-stage reduction); size from bts_conv3d_fwd_workspace (0 when the shape does not need it). */
long bts_conv3d_fwd_workspace(int kind, int N, int D, int H, int W, int Cin, int Cout);
int bts_conv3d_fwd(int kind, const float* x, const float* wp_fwd, const float* bias, float* y, void* workspace,
                   long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldy, int flags,
                   bts_stream_t stream);
/* ResNet block pair from one pass over x: y = conv3x3x3(x)+bias (resnet.py:133-134), y2 = conv1x1x1(x)+bias2 (resnet.py:118).
 * wp2 is the BTS_CONV_K1 forward packing. bts_conv3d_fwd_can_fuse() == 0 -> call the two convolutions separately. */
int bts_conv3d_fwd_can_fuse(int N, int D, int H, int W, int Cin, int Cout);
int bts_conv3d_fwd_fused2(const float* x, const float* wp_fwd, const float* bias, float* y, const float* wp2,
                          const float* bias2, float* y2, int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldy,
                          int ldy2, bts_stream_t stream);
/* dx (+)= conv^T(dy). (D,H,W) are the forward INPUT dims. Replaces tf.GradientTape for these ops (train.py:142-151). */
long bts_conv3d_bwd_data_workspace(int kind, int N, int D, int H, int W, int Cin, int Cout);
int bts_conv3d_bwd_data(int kind, const float* dy, const float* wp_bwd, float* dx, void* workspace, long workspace_bytes,
                        int N, int D, int H, int W, int Cin, int lddx, int Cout, int lddy, int flags, bts_stream_t stream);
/* which igemm_kernel<MS,NS,WM,WN,KGS> instantiation a call resolves to (id 0..4, +8 for the 1x1x1 staging variant);
 * used by bench.py to attribute measured launch times to kernel symbols */
int bts_conv3d_fwd_config(int kind, int N, int D, int H, int W, int Cin, int Cout);
int bts_conv3d_bwd_data_config(int kind, int N, int D, int H, int W, int Cin, int Cout);
long bts_conv3d_bwd_weight_workspace(int kind, int N, int D, int H, int W, int Cin, int Cout);
/* dw in the reference layout (Cin_ref = Cin + dup_shift); db (may be NULL; not produced for K3S2T: use bts_colsum). */
int bts_conv3d_bwd_weight(int kind, const float* x, const float* dy, float* dw, float* db, void* workspace,
                          long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int lddy,
                          int dup_start, int dup_shift, int accumulate, bts_stream_t stream);

/* ===== GroupNormalization (layers/group_norm.py:83-124) ===== */
long bts_gn_workspace(int N, long V, int C, int G, int mode);
long bts_gn_bwd_workspace(int N, long V, int C, int G, int mode);
/* mean,rstd: (N*G) floats. x dense NDHWC (ld == C). */
int bts_gn_stats(const float* x, float* mean, float* rstd, void* workspace, long workspace_bytes, int N, long V, int C,
                 int G, int mode, float eps, bts_stream_t stream);
/* y = [relu](gamma_b*(x-mean)*rstd + beta_b); y may be a channel slice (ldy >= C). */
int bts_gn_apply(const float* x, float* y, const float* gamma, const float* beta, const float* mean, const float* rstd,
                 int N, long V, int C, int ldy, int G, int mode, int relu, bts_stream_t stream);
int bts_gn_bwd(const float* x, const float* dy, float* dx, const float* gamma, const float* beta, const float* mean,
               const float* rstd, float* dgamma, float* dbeta, void* workspace, long workspace_bytes, int N, long V, int C,
               int lddy, int G, int mode, int relu, int accumulate_params, bts_stream_t stream);

/* ===== squeeze-excitation gate + block epilogue (layers/resnet.py:116-138) ===== */
long bts_colsum_workspace(int N, long rows, int C);
int bts_colsum(const float* x, float* out, void* workspace, long workspace_bytes, int N, long rows, int C, int ld,
               float scale, int sum_over_n, int accumulate, bts_stream_t stream);
int bts_se_mlp_fwd(const float* gap, const float* w1, const float* w2, float* h, float* ch, int N, int F, int R,
                   bts_stream_t stream);
/* out = res*(sigmoid(res.wsp)+ch) + relu(GN2(c2)); c2 may be NULL (gate only). sp: (N*V) saved for backward. */
int bts_block_epilogue_fwd(const float* res, const float* c2, float* out, float* sp, const float* wsp, const float* ch,
                           const float* gamma, const float* beta, const float* mean, const float* rstd, int N, long V,
                           int F, int ldo, int G, int mode, bts_stream_t stream);
long bts_se_bwd_workspace(int N, long V, int F, int R);
int bts_se_bwd(const float* dout, const float* res, const float* sp, const float* gap, const float* h, const float* ch,
               const float* w1, const float* w2, const float* wsp, float* dres, float* ds, float* dgap, float* dw1,
               float* dw2, float* dwsp, void* workspace, long workspace_bytes, int N, long V, int F, int R, int lddo,
               int accumulate_params, bts_stream_t stream);

/* A ResnetBlock's gate backward (bts_se_bwd) and GroupNorm-2 backward (bts_gn_bwd, slab mode, ReLU) in one pair of passes: what
 * tf.GradientTape (train.py:142,151) derives for layers/resnet.py:121-137 from the gradient of the block output.  dout (N,V,F) rows of
 * lddo floats; res, c2 dense; outputs dres, dc2 dense, ds (N*V) and dgap (N,F) scratch.  accumulate_*: += into the parameter gradients.
 * The workspace query returns -1 and the call BTS_ERR_UNSUPPORTED outside the kernels' tiling (the caller runs the two separate calls). */
long bts_block_bwd_workspace(int N, long V, int F, int R, int G);
int bts_block_bwd(const float* dout, int lddo, const float* res, const float* c2, const float* sp, const float* gap, const float* h,
                  const float* ch, const float* w1, const float* w2, const float* wsp, const float* gamma, const float* beta,
                  const float* mean, const float* rstd, float* dres, float* dc2, float* ds, float* dgap, float* dw1, float* dw2,
                  float* dwsp, float* dgamma, float* dbeta, void* workspace, long workspace_bytes, int N, long V, int F, int R, int G,
                  int accumulate_gate_params, int accumulate_norm_params, bts_stream_t stream);

/* ===== element-wise (layers/encoder.py:39,71; layers/vae.py:9-13) ===== */
int bts_dropout_mask(uint8_t* mask, long n, float rate, uint64_t seed, bts_stream_t stream);
int bts_dropout_apply(const float* x, const uint8_t* mask, float* y, long n, float rate, bts_stream_t stream);
int bts_normal(float* out, long n, uint64_t seed, bts_stream_t stream);
int bts_vae_sample_fwd(const float* proj, const float* eps, float* z, int N, int L, bts_stream_t stream);
int bts_vae_sample_bwd(const float* proj, const float* eps, const float* dz, float* dproj, int N, int L, bts_stream_t stream);
int bts_fill(float* p, long n, float v, bts_stream_t stream);
int bts_axpy(float* y, const float* x, long n, float a, bts_stream_t stream);
int bts_add_strided(float* dst, const float* src, long rows, int C, int ldd, int lds_, int accumulate, bts_stream_t stream);
int bts_scalar_lincomb(float* out, const float* a, const float* b, float ca, float cb, bts_stream_t stream);
int bts_relu_bwd(const float* y, const float* dy, float* dx, long n, bts_stream_t stream);
/* dx (rows,C dense) = dy*y*(1-y): gradient through the fused output sigmoid (decoder.py:60) */
int bts_sigmoid_bwd(const float* y, const float* dy, float* dx, long rows, int C, int ldy, int lddy, bts_stream_t stream);

/* ===== Dense (layers/vae.py:61-64,105-109) ===== */
long bts_dense_workspace(int N, int in, int out);
int bts_dense_fwd(const float* x, const float* w, const float* bias, float* y, void* workspace, long workspace_bytes, int N,
                  int in, int out, int relu, bts_stream_t stream);
int bts_dense_bwd(const float* x, const float* w, const float* g, float* dx, float* dw, float* db, int N, int in, int out,
                  int accumulate_dx, int accumulate_params, bts_stream_t stream);

/* ===== loss / metric / regulariser (util.py:13-24,35-57; train.py:146) ===== */
long bts_loss_workspace(void);
/* sums: 3C+4 doubles = I[C],P[C],T[C], sum (x-y_vae)^2, KL sum, numel_x, numel_z -- all-reduce these for data parallel */
int bts_loss_sums(const float* y_pred, const float* y, const float* x, const float* y_vae, const float* proj, double* sums,
                  void* workspace, long workspace_bytes, int N, long V, int C, int ldp, int ldy, int Cx, int ldx, int ldv,
                  int Lz, bts_stream_t stream);
int bts_loss_value(const double* sums, float* loss, float* parts, int C, int has_vae, bts_stream_t stream);
int bts_loss_bwd(const float* y_pred, const float* y, const float* x, const float* y_vae, const float* proj,
                 const double* sums, const float* gscale, float* dlogit, float* dyvae, float* dproj, int N, long V, int C,
                 int ldp, int ldy, int Cx, int ldx, int ldv, int Lz, int through_sigmoid, bts_stream_t stream);
/* table: (cells*C*3) doubles, cells = W if channels_last_axes else 1; labels: (N*V) uint8 or NULL */
int bts_dice_metric_sums(const float* y_true, const float* y_pred, uint8_t* labels, double* table, int N, long V, int W,
                         int C, int ldt, int ldp, int channels_last_axes, bts_stream_t stream);
int bts_dice_metric_value(const double* table, float* out, int W, int C, int channels_last_axes, bts_stream_t stream);
long bts_l2_workspace(void);
int bts_l2_reg_fwd(const float* params, const long* off, const long* len, const float* coef, int nranges, float* out,
                   void* workspace, long workspace_bytes, bts_stream_t stream);
int bts_l2_reg_bwd(const float* params, float* grads, const long* off, const long* len, const float* coef, int nranges,
                   const float* gscale, bts_stream_t stream);

/* ===== optimiser (util.py:60-84; Keras Adam, epsilon un-corrected) ===== */
int bts_adam_tf_step(float* p, const float* g, float* m, float* v, long n, float lr_t, float beta1, float beta2, float eps,
                     float gmul, bts_stream_t stream);
/* Dynamic loss scaling of the fp16-storage trainer without a host round trip (no reference counterpart: the reference trains in
 * fp32, train.py:140-152): flag[0] = 1 iff any gradient element is Inf / NaN; the guarded step is bts_adam_tf_step unless *skip != 0,
 * in which case nothing is written (the step is skipped on the device; the host learns of it one step later). */
int bts_grad_nonfinite(const float* g, long n, int* flag, bts_stream_t stream);
int bts_adam_tf_step_guarded(float* p, const float* g, float* m, float* v, long n, float lr_t, float beta1, float beta2, float eps,
                             float gmul, const int* skip, bts_stream_t stream);

/* ===== the two data-gradient paths into a ResNet block's input in one pass (resnet.py:118,134: x feeds conv1 and the
 * 1x1x1 shortcut) ===== */
/* dx (+)= bwd_data(3x3x3 s1)(dy, wp_bwd) + bwd_data(1x1x1)(dy2, wp2_bwd); both weight images in BTS_ROLE_BWD_DATA packing.
 * The 1x1x1 term is evaluated at the centre tap of the 3x3x3 sweep; the library falls back to two launches where that is
 * not possible (split-K grids, unaligned dy2). */
long bts_conv3d_bwd_data_pair_workspace(int N, int D, int H, int W, int Cin, int Cout);
int bts_conv3d_bwd_data_pair(const float* dy, const float* wp_bwd, const float* dy2, const float* wp2_bwd, float* dx,
                             void* workspace, long workspace_bytes, int N, int D, int H, int W, int Cin, int lddx, int Cout,
                             int lddy, int lddy2, int flags, bts_stream_t stream);

/* ===== convolution + GroupNorm statistics of its output in one pass (resnet.py:80-93: conv -> GroupNormalization) ===== */
/* y = conv(x) + bias, y DENSE (voxel stride Cout), and (mean, rstd)[N*G] = BTS_GN_SLAB statistics of y.  The sums come out of
 * the conv epilogue when the tiled kernel takes the launch (no split-K, whole tiles per z-slab group); otherwise the library
 * runs bts_gn_stats on y itself.  Workspace (required) from bts_conv3d_fwd_gn_workspace. */
long bts_conv3d_fwd_gn_workspace(int kind, int N, int D, int H, int W, int Cin, int Cout, int G);
int bts_conv3d_fwd_gn(int kind, const float* x, const float* wp_fwd, const float* bias, float* y, void* workspace,
                      long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int G, float eps, float* mean,
                      float* rstd, bts_stream_t stream);
/* the fused shortcut pair (bts_conv3d_fwd_fused2) with the statistics of its 3x3x3 output y */
int bts_conv3d_fwd_fused2_gn(const float* x, const float* wp_fwd, const float* bias, float* y, const float* wp2,
                             const float* bias2, float* y2, void* workspace, long workspace_bytes, int N, int D, int H, int W,
                             int Cin, int ldx, int Cout, int ldy2, int G, float eps, float* mean, float* rstd,
                             bts_stream_t stream);

/* ===== non-default samplers (downsample.py:51-70, upsample.py:49-79) ===== */
/* MaxPooling3D(pool 2, stride 2) on even (D,H,W): y (N,D/2,H/2,W/2,C); idx (dense, one byte per output element) records the
 * window position dz*4+dy*2+dx of the first maximum and routes the gradient in bts_maxpool2_bwd (D,H,W = INPUT dims). */
int bts_maxpool2_fwd(const float* x, float* y, uint8_t* idx, int N, int D, int H, int W, int C, int ldx, int ldy,
                     bts_stream_t stream);
int bts_maxpool2_bwd(const float* dy, const uint8_t* idx, float* dx, int N, int D, int H, int W, int C, int lddy, int lddx,
                     int accumulate, bts_stream_t stream);
/* UpSampling3D(size 2) = nearest-neighbour repeat: y (N,2D,2H,2W,C) from x (N,D,H,W,C); backward sums the 8 children
 * (D,H,W = COARSE dims in both calls). */
int bts_upsample2_fwd(const float* x, float* y, int N, int D, int H, int W, int C, int ldx, int ldy, bts_stream_t stream);
int bts_upsample2_bwd(const float* dy, float* dx, int N, int D, int H, int W, int C, int lddy, int lddx, int accumulate,
                      bts_stream_t stream);

/* ===== full-volume inference helpers (test.py:95-151,259-261) ===== */
/* dst (+)= scale * t(flip(src)) on dense NDHWC tensors; flip_mask bits 4|2|1 reverse D|H|W (tf.reverse, test.py:139,142);
 * mean/stdv (C floats, both or neither): t(v) = (v - mean[c]) / std[c] (test.py:111), identity when NULL. src != dst
 * unless flip_mask == 0. */
int bts_flip_affine(const float* src, float* dst, const float* mean, const float* stdv, int N, int D, int H, int W, int C,
                    int flip_mask, float scale, int accumulate, bts_stream_t stream);
/* y = prob * bmask (test.py:154-155; bmask one float per voxel) and/or the label map the script intends: argmax_c + 1,
 * values >= 3 -> 4 (test.py:259-261), 0 where masked out or below `threshold`.  y and labels may each be NULL. */
int bts_tta_finish(const float* prob, const float* bmask, float* y, uint8_t* labels, long nvox, int C, float threshold,
                   bts_stream_t stream);

/* ===== training-time augmentation on the device (train.py:14-49) ===== */
/* per-channel mean / population variance of a (nvox, C) tensor with voxel stride ld (tf.nn.moments, train.py:18);
 * C <= 16; mean may be NULL; fp64 partials, fixed-order combine */
long bts_channel_moments_workspace(int C);
int bts_channel_moments(const float* x, float* mean, float* var, void* workspace, long workspace_bytes, long nvox, int C,
                        int ld, bts_stream_t stream);
/* xo = crop+flip of (x + shift[c]*sqrt(var[c])) * scale[c] (train.py:19-34), yo = one-hot of the cropped+flipped labels
 * without the background channel (train.py:38-41).  x: (S0,S1,S2,C), y: (S0,S1,S2) float labels, var: C device floats,
 * shift/scale: C HOST floats (the caller's random draws), window origin (o0,o1,o2), flip_mask bits 4|2|1 = axes 0|1|2. */
int bts_augment_crop(const float* x, const float* y, const float* var, float* xo, float* yo, int S0, int S1, int S2, int C,
                     int T0, int T1, int T2, int o0, int o1, int o2, int flip_mask, const float* shift, const float* scale,
                     int out_ch, bts_stream_t stream);

/* library identification */
const char* bts_version(void);

/* ===== optional in-library timing of the two dominant kernels (bench.py roofline) ===== */
/* HIP events recorded on the launch stream immediately around igemm_kernel / wgrad_kernel launches.
 * bts_profile_enable(1) clears and starts recording, (0) stops; read the records after synchronising the stream.
 * sym: 0..4 = igemm_kernel<2,1,4,1|2,2,4,1|1,2,2,2|1,1,2,2|1,1,4,1>, +8 = its 1x1x1 staging variant; 100/101 = wgrad_kernel<true/false> */
int bts_profile_enable(int on);
int bts_profile_count(void);
int bts_profile_get(int i, int* sym, double* flops, float* ms);

/* ===== reduced-precision STORAGE path of the forward pass (BASELINE configs[4]: fp16 full-volume inference, VAE off: model.py:67-68,
 * test.py:128-134; forward half of configs[2], bf16).  Activations and packed weights are 16-bit, every sum is fp32
 * (v_mfma_f32_32x32x16_{f16,bf16}; GroupNorm statistics, gates and the sigmoid head in fp32 from the stored values).  The reference has
 * no such mode: the parity reference of these entry points is the fp32 engine above.  `ld` arguments count ELEMENTS. ===== */
#define BTS_LP_F16 1
#define BTS_LP_BF16 2
/* packed weight image of a conv kind: [tap][16-cin step][32-cout block][k half][32 couts][8 cin]; same folding arguments as bts_conv_pack */
long bts_lp_packed_bytes(int kind, int role, int Cin_slab, int Cout);
int bts_lp_pack(int kind, int role, int dtype, const float* w, void* wp, int Cin_ref, int Cout, int Cin_slab, int dup_start,
                int dup_shift, bts_stream_t stream);
/* y = conv(x) + bias (resnet.py:30-37,80-87,96-103; downsample.py:28-35; upsample.py:28-33).  Cin % 16 == 0; views' strides % 8 == 0 */
/* workspace (may be NULL / 0): lets stride-1 grids too small to fill the chip split the input channels over workgroups (fp32 partial
 * sums, fixed-order reduce); size from bts_lp_conv3d_workspace (0 when the shape does not need it) */
long bts_lp_conv3d_workspace(int kind, int N, int D, int H, int W, int Cin, int Cout);
int bts_lp_conv3d_fwd(int kind, int dtype, const void* x, const void* wp, const float* bias, void* y, void* workspace, long workspace_bytes,
                      int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldy, bts_stream_t stream);
/* dx (+)= conv^T(dy) in the storage type (the data gradients TF autodiff derives, train.py:142-151); (D,H,W) are the forward INPUT
 * dims, wp_bwd = bts_lp_pack(kind, BTS_ROLE_BWD_DATA, ...); Cout % 16 == 0 */
long bts_lp_conv3d_bwd_data_workspace(int kind, int N, int D, int H, int W, int Cin, int Cout);
int bts_lp_conv3d_bwd_data(int kind, int dtype, const void* dy, const void* wp_bwd, void* dx, void* workspace, long workspace_bytes,
                           int N, int D, int H, int W, int Cin, int lddx, int Cout, int lddy, int accum, bts_stream_t stream);
/* dx (+)= conv3x3x3^T(dy) + conv1x1x1^T(dy2) in ONE launch: the data gradients of the two convolutions that read a ResnetBlock's input
 * (conv1 resnet.py:80-87 and the shortcut resnet.py:96-103, both applied to `inputs`: resnet.py:118,134) meet in dx under train.py:151.
 * The shortcut's contraction rides on conv1's as an extra K-segment at the centre tap (1/27 more matrix work) with dy2 read straight
 * into the matrix instruction -- instead of a second launch that read-modify-writes the Cin-wide dx.  dy / dy2: (N,D,H,W,Cout), voxel
 * strides lddy / lddy2; wp_bwd / wp2_bwd = bts_lp_pack(K3S1 / K1, BTS_ROLE_BWD_DATA, ...) with the same Cin_slab and fold.  Shapes the
 * fused kernels do not take run as the two launches (the call always completes); *fused (may be NULL) = 1 if the one-launch form ran.
 * dx_split (elements; 0 = dx is one (N,D,H,W,Cin) view): columns [32 b, 32 b + 32) go to dx + b * dx_split, every block a tensor of its
 * own with voxel stride lddx -- the gradient of the decoder's concat of 32-channel tensors (decoder.py:75) leaves as DENSE tensors whose
 * readers fetch whole 128-byte lines.  Only the fused tiled kernel writes that form: bts_lp_conv3d_bwd_data_sc_split_ok (1 / 0) says
 * whether it takes the shape; BTS_ERR_UNSUPPORTED otherwise.  BTS_LP_SC=0 in the environment: always the two launches, BTS_LP_SC_SPLIT=0:
 * _split_ok answers 0 (A/B aids) */
long bts_lp_conv3d_bwd_data_sc_workspace(int N, int D, int H, int W, int Cin, int Cout);
int bts_lp_conv3d_bwd_data_sc_split_ok(int N, int D, int H, int W, int Cin, int Cout);
int bts_lp_conv3d_bwd_data_sc(int dtype, const void* dy, const void* wp_bwd, const void* dy2, const void* wp2_bwd, void* dx, long dx_split,
                              void* workspace, long workspace_bytes, int N, int D, int H, int W, int Cin, int lddx, int Cout, int lddy,
                              int lddy2, int accum, int* fused, bts_stream_t stream);
/* The weight gradients of the TWO convolutions that read a ResnetBlock's input (resnet.py:134 conv1, 3x3x3; resnet.py:118 shortcut, 1x1x1)
 * from ONE pass over that input (train.py:151): dw3 (+)= from dy3 as bts_lp_conv3d_bwd_weight(K3S1), dw1 (Keras layout (1,1,1,Cin_ref,Cout))
 * (+)= sum_v x[v][c] dy1[v][k] -- one more accumulator per wave of the streaming kernel, fed by the centre-tap fragments of x it reads anyway;
 * the HBM-bound 1x1x1 weight-gradient launch and its own read of the Cin-wide x go away.  db3 (may be NULL; dy3 dense then) (+)= sum dy3.
 * dup_start / dup_shift fold both kernels alike.  x_split (elements; 0 = one tensor): x as a list of Cin / 32 dense 32-channel tensors x +
 * b * x_split with voxel stride ldx (the operands of a concat; no fold then).  Workspace query -1 / return value 1 (nothing launched) outside the streaming kernel's
 * shapes (W % 32, H % 8, large volumes): run bts_lp_conv3d_bwd_weight twice.  BTS_LP_WPAIR=0 in the environment: never (A/B aid) */
long bts_lp_conv3d_bwd_weight_pair_workspace(int N, int D, int H, int W, int Cin, int Cout);
int bts_lp_conv3d_bwd_weight_pair(int dtype, const void* x, long x_split, const void* dy3, const void* dy1, float* dw3, float* dw1, float* db3,
                                  void* workspace, long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int lddy3,
                                  int lddy1, int dup_start, int dup_shift, int accumulate, bts_stream_t stream);
/* dw (fp32, Keras layout (kd,kh,kw,Cin_ref,Cout)) (+)= the weight gradient of a stride-1 3x3x3 / 1x1x1 conv from 16-bit x and dy
 * (voxel contraction on the 16-bit matrix pipe, fp32 partials, fixed-order finalize); db (may be NULL; dy dense then) (+)= sum dy.
 * dup_start / dup_shift as bts_conv_pack: Cin + dup_shift == Cin_ref, both copies of the folded slice receive the gradient.
 * BTS_ERR_UNSUPPORTED for the strided kinds and channel counts that are not multiples of 8: run bts_conv3d_bwd_weight on widened copies */
long bts_lp_conv3d_bwd_weight_workspace(int kind, int N, int D, int H, int W, int Cin, int Cout);
int bts_lp_conv3d_bwd_weight(int kind, int dtype, const void* x, const void* dy, float* dw, float* db, void* workspace, long workspace_bytes,
                             int N, int D, int H, int W, int Cin, int ldx, int Cout, int lddy, int dup_start, int dup_shift, int accumulate,
                             bts_stream_t stream);
/* fp32 <-> storage type, `rows` rows of C elements with row strides (the 2-channel input block runs in fp32: K = 16 is its floor) */
int bts_lp_cast(int dtype, const float* src, long ld_src, void* dst, long ld_dst, long rows, int C, bts_stream_t stream);
/* encoder.py:39,71 (Dropout(rate) on the input volume) AND the cast of the dense (rows, C <= 4) fp32 volume into the zero-padded 16-channel
 * matrix step, in one pass: element i is kept (scaled by 1 / (1 - rate)) where bts_dropout_mask's generator says so for the same (seed, i)
 * -- bit-identical to bts_dropout_mask + bts_dropout_apply + bts_lp_cast_pad16, without the mask and the dropped volume being written */
int bts_lp_dropout_cast_pad16(int dtype, const float* src, void* dst, long rows, int C, float rate, uint64_t seed, bts_stream_t stream);
int bts_lp_uncast(int dtype, const void* src, long ld_src, float* dst, long ld_dst, long rows, int C, bts_stream_t stream);
/* fp32 rows of C <= 4 channels -> dense storage-type rows of 16 channels with a zero tail, in one pass: the 2-channel input volume
 * (model.py:58, after encoder.py:71's dropout), the 2-channel VAE-output gradient and the 1-channel VAE tensor (vae.py:110-111) as
 * whole 16-channel matrix steps */
int bts_lp_cast_pad16(int dtype, const float* src, long ld_src, void* dst, long rows, int C, bts_stream_t stream);
/* GroupNormalization (group_norm.py:83-124), both semantics; x dense */
long bts_lp_gn_workspace(int N, long V, int C, int G);
int bts_lp_gn_stats(int dtype, const void* x, float* mean, float* rstd, void* workspace, long workspace_bytes, int N, long V, int C, int G,
                    int mode, float eps, bts_stream_t stream);
int bts_lp_gn_apply(int dtype, const void* x, void* y, const float* gamma, const float* beta, const float* mean, const float* rstd, int N,
                    long V, int C, int ldy, int G, int mode, int relu, bts_stream_t stream);
/* GroupNormalization backward (slab semantics; fused ReLU mask when relu != 0): dx in the storage type and, when dx32 != NULL, the same
 * values in fp32 (for a weight gradient that still runs on the fp32 matrix pipe); dgamma / dbeta fp32 (+= when accumulate_params).
 * dbias (may be NULL, C floats, += when accumulate_params): column sums of dx over voxels and samples = the bias gradient of the conv
 * whose output this layer normalised (resnet.py:80-93: conv -> GroupNormalization), taken from the same pass.
 * BTS_ERR_UNSUPPORTED for shapes outside its tiling (group length not a multiple of 2048, C/G > 32): run bts_gn_bwd on widened copies */
long bts_lp_gn_bwd_workspace(int N, long V, int C, int G);
int bts_lp_gn_bwd(int dtype, const void* x, const void* dy, void* dx, float* dx32, const float* gamma, const float* beta, const float* mean,
                  const float* rstd, float* dgamma, float* dbeta, void* workspace, long workspace_bytes, int N, long V, int C, int lddy, int G,
                  int relu, int accumulate_params, float* dbias, bts_stream_t stream);
/* A ResnetBlock's conv2 data gradient and its GroupNorm-1 (+ReLU) backward as one entry point (resnet.py:80-93 in reverse under
 * train.py:151; replaces bts_lp_conv3d_bwd_data followed by bts_lp_gn_bwd): da = conv3x3x3^T(dy) is stored dense (N,D,H,W,Cg) in the
 * storage type, dc / dc32 / dgamma / dbeta / dbias are bts_lp_gn_bwd's outputs for x = c, dy = da.  Where the z-marching kernel takes
 * the layer the class sums GroupNorm's backward starts from leave the conv's epilogue (one read of c per stored row) instead of a
 * reduce pass over da and c; elsewhere the two run back to back -- same results up to the order of the fp32 class sums.
 * Cg = GroupNorm channels = the forward conv's input channels, Cdy = dy's channels (rows of lddy); wp_bwd = bts_lp_pack(BTS_CONV_K3S1,
 * BTS_ROLE_BWD_DATA, ...); *fused_out (may be NULL) <- 1 | 0: which form ran.  Error codes as bts_lp_gn_bwd. */
long bts_lp_conv3d_bwd_data_gn_bwd_workspace(int N, int D, int H, int W, int Cg, int Cdy, int G);
int bts_lp_conv3d_bwd_data_gn_bwd(int dtype, const void* dy, const void* wp_bwd, void* da, const void* c, void* dc, float* dc32,
                                  const float* gamma, const float* beta, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                  void* workspace, long workspace_bytes, int N, int D, int H, int W, int Cg, int Cdy, int lddy, int G, int relu,
                                  int accumulate_params, float* dbias, int* fused_out, bts_stream_t stream);
/* GlobalAveragePooling3D of the shortcut (resnet.py:45-46,121): out[n][c] = scale * sum_v x */
long bts_lp_colsum_workspace(int N, long V, int C);
int bts_lp_colsum(int dtype, const void* x, float* out, void* workspace, long workspace_bytes, int N, long V, int C, float scale,
                  bts_stream_t stream);
/* out = res * (sigmoid(res . w_sp) + ch[n]) + relu(GN2(c2))   (resnet.py:127-137) */
/* sp_out (may be NULL): the per-voxel spatial gate sigmoid(res . w_sp), fp32 [N*V], kept for the backward pass */
int bts_lp_block_epilogue(int dtype, const void* res, const void* c2, void* out, float* sp_out, const float* wsp, const float* ch, const float* gamma,
                          const float* beta, const float* mean, const float* rstd, int N, long V, int C, int ldo, int G, int mode,
                          bts_stream_t stream);
/* The FIRST ResnetBlock's two convolutions of the raw in_ch = 2 volume in a forward without a backward (resnet.py:30-37,80-87,118 at encoder
 * level 0; model.py:63-68): c1 = conv3x3x3(x) + b3 and res = conv1x1x1(x) + b1 in the storage type (dense, F channels) from ONE pass over the
 * fp32 input x (N,D,H,W,2), GroupNorm-1's BTS_GN_SLAB statistics of c1 (mean, rstd [N*G]) and gap[n][c] = mean over voxels of res (the
 * gate's squeeze, resnet.py:121).  w3 (3,3,3,2,F), w1 (1,1,1,2,F): the reference's fp32 kernels, unpacked.  Replaces bts_lp_cast_pad16 +
 * bts_lp_conv1_gap + bts_lp_conv3d_fwd_gn on the zero-padded 16-channel copy.  Workspace query -1 / BTS_ERR_UNSUPPORTED outside its shapes
 * (F in {8,16,24,32}, D % G == 0, F % G == 0). */
long bts_lp_first_block_workspace(int N, int D, int H, int W, int F, int G);
int bts_lp_first_block_fwd(int dtype, const float* x, const float* w3, const float* b3, const float* w1, const float* b1, void* c1, void* res,
                           float* mean, float* rstd, float* gap, void* workspace, long workspace_bytes, int N, int D, int H, int W, int F,
                           int G, float eps, bts_stream_t stream);
/* The last decoder block's epilogue with the output head in it (decoder.py:55-63: Conv3D 1x1x1 -> out_ch, sigmoid; model.py:63-68 in a
 * forward without a backward): y_head [N*V][K] fp32 = sigmoid?(out . head_w + head_b) with `out` as bts_lp_block_epilogue forms it, never
 * written.  head_w (C, K) fp32 row-major, head_b (K) or NULL.  BTS_ERR_UNSUPPORTED outside the fused kernel's shapes (K <= 4, C <= 64,
 * whole 2048-element chunks per unit): call bts_lp_block_epilogue and bts_lp_head. */
int bts_lp_block_epilogue_head(int dtype, const void* res, const void* c2, float* y_head, const float* wsp, const float* ch, const float* gamma,
                               const float* beta, const float* mean, const float* rstd, const float* head_w, const float* head_b, int N,
                               long V, int C, int G, int mode, int K, int sigmoid, bts_stream_t stream);
/* y = conv3x3x3(x) + bias in the storage type (dense) and the BTS_GN_SLAB statistics of y (resnet.py:80-93 conv -> GroupNormalization)
 * in one pass where the tiled kernel can emit the partial sums from its epilogue; the 16-bit counterpart of bts_conv3d_fwd_gn */
long bts_lp_conv3d_fwd_gn_workspace(int N, int D, int H, int W, int Cin, int Cout, int G);
int bts_lp_conv3d_fwd_gn(int dtype, const void* x, const void* wp, const float* bias, void* y, float* mean, float* rstd, void* workspace,
                         long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int G, float eps,
                         bts_stream_t stream);
/* bts_lp_conv3d_fwd_gn / bts_lp_conv1_gap ADDING to what y / res already hold (accumulate != 0), statistics / squeeze of the final sums: a
 * contraction split over its input channels whose parts become available at different times -- the decoder block's conv1 and shortcut over
 * [skip | up-sampled] (decoder.py:75; resnet.py:118,134): the skip part (with the bias) can run as soon as the encoder level is done, next
 * to the under-filled deep levels, the up-sampled part adds to it later (bias = NULL then).  The partial sum passes through the storage type
 * once.  Workspaces: bts_lp_conv3d_fwd_gn_workspace / bts_lp_conv1_gap_workspace */
int bts_lp_conv3d_fwd_gn_acc(int dtype, const void* x, const void* wp, const float* bias, void* y, float* mean, float* rstd, void* workspace,
                             long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int G, float eps, int accumulate,
                             bts_stream_t stream);
int bts_lp_conv1_gap_acc(int dtype, const void* x, const void* wp, const float* bias, void* res, float* gap, void* workspace,
                         long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldres, int accumulate,
                         bts_stream_t stream);
/* conv1 AND the shortcut of a ResnetBlock from ONE pass over the block input (resnet.py:118 and resnet.py:134 read the same `inputs`):
 * y = conv3x3x3(x) + bias with GroupNorm G's statistics of y (as bts_lp_conv3d_fwd_gn), res = conv1x1x1(x) + bias_pt and gap[n][c] = mean
 * over the voxels of the unrounded res (as bts_lp_conv1_gap).  The shortcut is a second set of output columns at the centre tap of the
 * z-marching kernel's input planes: x is read once.  wp / wp_pt = bts_lp_pack(K3S1 / K1, BTS_ROLE_FWD, ...) with the same Cin_slab and
 * fold; y, res dense (N,D,H,W,Cout).  Cin = 64 on volumes of >= 8 M voxels: both ride on the two z-marching passes over the channel halves.
 * x_split (elements; 0 = x is one (N,D,H,W,Cin) view): Cin = 64 as TWO 32-channel tensors x and x + x_split with voxel stride ldx each -- the
 * operands of a concat (decoder.py:75) never materialised side by side, each read in whole lines (virtual concat as a list of
 * (ptr, C) segments); the two passes at any volume size (pass ldx = 32 to the workspace query).  Workspace query -1 / return value 1 (nothing
 * launched) outside the kernel's shapes: run bts_lp_conv1_gap +
 * bts_lp_conv3d_fwd_gn.  BTS_LP_FS=0 in the environment: never, BTS_LP_FS_PAIR=0: not on the two-pass form (A/B aids) */
long bts_lp_conv3d_fwd_gn_shortcut_workspace(int N, int D, int H, int W, int Cin, int ldx, int Cout, int G);
int bts_lp_conv3d_fwd_gn_shortcut(int dtype, const void* x, long x_split, const void* wp, const float* bias, void* y, float* mean, float* rstd,
                                  const void* wp_pt, const float* bias_pt, void* res, float* gap, void* workspace, long workspace_bytes,
                                  int N, int D, int H, int W, int Cin, int ldx, int Cout, int G, float eps, bts_stream_t stream);
/* The same with the GroupNorm + ReLU of the INPUT applied on the way in (in_relu = 1): y = conv3x3x3(relu(GN_in(x))) + bias and the statistics of y.
 * conv2 of a ResnetBlock reading conv1's raw output (layers/resnet.py:133-136: conv -> GroupNormalization -> relu -> conv) in a forward
 * whose normalised tensor nobody else reads (Model.call(inference=True), model.py:58-71; test.py:128-151): the separate apply pass of
 * layers/group_norm.py:110-122 goes away.  x dense (N,D,H,W,Cin); in_* = that GroupNorm's parameters and statistics (slab mode).
 * The workspace query returns -1 and the call BTS_ERR_UNSUPPORTED where no kernel takes the shape in this form. */
long bts_lp_conv3d_gnin_fwd_gn_workspace(int N, int D, int H, int W, int Cin, int Cout, int in_G, int G);
int bts_lp_conv3d_gnin_fwd_gn(int dtype, const void* x, const float* in_gamma, const float* in_beta, const float* in_mean,
                              const float* in_rstd, int in_G, int in_relu, const void* wp, const float* bias, void* y, float* mean,
                              float* rstd, void* workspace, long workspace_bytes, int N, int D, int H, int W, int Cin, int Cout, int G,
                              float eps, bts_stream_t stream);
/* The same idea for TRAINING: conv2's weight gradient dW[t][c][k] = sum_v a[v + off_t][c] dy[v][k], a = relu(GN_in(x)), taken from the RAW x
 * (the streaming weight-gradient kernel normalises its planes in LDS) -- with bts_lp_conv3d_gnin_fwd_gn in the forward, relu(GN1(c1)) of
 * layers/resnet.py:133-134 is never written (what tf.GradientTape keeps alive for train.py:151 is c1 alone).
 * bts_lp_conv3d_gnin_train_ok: 1 when both kernels take the shape in this form (ask before the forward), 0 otherwise (not a status).
 * bts_lp_conv3d_gnin_bwd_weight: x dense (N,D,H,W,Cin) raw; dy rows of lddy; dw (3,3,3,Cin,Cout) fp32 (+)=; db (may be NULL) (+)=;
 * workspace of bts_lp_conv3d_bwd_weight_workspace(BTS_CONV_K3S1, ...). */
int bts_lp_conv3d_gnin_train_ok(int N, int D, int H, int W, int Cin, int Cout, int in_G, int G);
int bts_lp_conv3d_gnin_bwd_weight(int dtype, const void* x, const float* in_gamma, const float* in_beta, const float* in_mean,
                                  const float* in_rstd, int in_G, const void* dy, float* dw, float* db, void* workspace,
                                  long workspace_bytes, int N, int D, int H, int W, int Cin, int Cout, int lddy, int accumulate,
                                  bts_stream_t stream);
/* y = Conv3DTranspose(k3, s2, 'same')(x) + bias (dense fine tensor (N,2D,2H,2W,Cout), storage type) and the slab-mode GroupNorm
 * statistics of y in one pass: ConvUpsample (upsample.py:28-43: conv -> GroupNormalization) without a statistics pass over the fine
 * tensor.  (D,H,W): the COARSE grid; wp = bts_lp_pack(BTS_CONV_K3S2T, BTS_ROLE_FWD, ...).  Falls back to the conv + bts_lp_gn_stats
 * where the merged transposed-conv kernel declines the shape or a group is not whole fine planes. */
long bts_lp_convT3d_fwd_gn_workspace(int N, int D, int H, int W, int Cin, int Cout, int G);
int bts_lp_convT3d_fwd_gn(int dtype, const void* x, const void* wp, const float* bias, void* y, float* mean, float* rstd, void* workspace,
                          long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int G, float eps,
                          bts_stream_t stream);
/* res = conv1x1x1(x) + bias in the storage type (dense, ldres == Cout) and gap[n][c] = mean over voxels of res: the block's
 * shortcut and the squeeze of its gate (resnet.py:118-121) in one pass (column sums from the conv epilogue + a small finalize) */
long bts_lp_conv1_gap_workspace(int N, long V, int Cout);
int bts_lp_conv1_gap(int dtype, const void* x, const void* wp, const float* bias, void* res, float* gap, void* workspace,
                     long workspace_bytes, int N, int D, int H, int W, int Cin, int ldx, int Cout, int ldres, bts_stream_t stream);
/* gate backward (resnet.py:121-130 under autodiff) on 16-bit dout / res -> dres in the storage type; fp32 parameter gradients;
 * sp = the sp_out of bts_lp_block_epilogue; ds [N*V] and dgap [N*F] are fp32 scratch outputs; dbias (may be NULL, F floats, += when
 * accumulate_params): column sums of dres = the bias gradient of the block's 1x1x1 shortcut conv (resnet.py:96-103) */
long bts_lp_se_bwd_workspace(int N, long V, int F, int R);
int bts_lp_se_bwd(int dtype, const void* dout, const void* res, const float* sp, const float* gap, const float* h, const float* ch,
                  const float* w1, const float* w2, const float* wsp, void* dres, float* ds, float* dgap, float* dw1, float* dw2, float* dwsp,
                  void* workspace, long workspace_bytes, int N, long V, int F, int R, int lddo, int accumulate_params, float* dbias,
                  bts_stream_t stream);
/* A ResnetBlock's gate backward and GroupNorm-2 (+ReLU) backward in one pair of passes (resnet.py:121-137 under TF autodiff): what
 * bts_lp_se_bwd followed by bts_lp_gn_bwd compute, reading the block-output gradient dout (N,V,F; rows of lddo) twice instead of four
 * times.  res, c2 dense; dres, dc2 dense outputs in the storage type; ds (N*V) / dgap (N,F) fp32 scratch outputs; parameter gradients
 * accumulate; dbias_pt / dbias_c2 (may be NULL): bias gradients of the shortcut conv / conv2 (+=).  BTS_ERR_UNSUPPORTED outside the
 * kernels' tiling (the caller runs the two separate entry points). */
long bts_lp_block_bwd_workspace(int N, long V, int F, int R, int G);
int bts_lp_block_bwd(int dtype, const void* dout, int lddo, const void* res, const void* c2, const float* sp, const float* gap, const float* h,
                     const float* ch, const float* w1, const float* w2, const float* wsp, const float* gamma, const float* beta,
                     const float* mean, const float* rstd, void* dres, void* dc2, float* ds, float* dgap, float* dw1, float* dw2, float* dwsp,
                     float* dgamma, float* dbeta, float* dbias_pt, float* dbias_c2, void* workspace, long workspace_bytes, int N, long V, int F,
                     int R, int G, bts_stream_t stream);
/* All 16-bit weight images in one launch (they go stale together at the optimiser step, train.py:152): host table of
 * bts_lp_pack_desc_bytes()-sized descriptors filled by bts_lp_pack_desc (arguments as bts_lp_pack; returns the entry's block count > 0 or
 * a negative engine code; first_block = running sum of those counts), copied to device memory by the caller. */
long bts_lp_pack_desc_bytes(void);
long bts_lp_pack_desc(void* host_table, int index, long first_block, int kind, int role, const float* w, void* wp, int Cin_ref, int Cout,
                      int Cin_slab, int dup_start, int dup_shift);
int bts_lp_pack_batch(int dtype, const void* table_dev, int n, long total_blocks, bts_stream_t stream);
/* the non-default samplers on 16-bit tensors (args.py:136-141): MaxPooling3D(2) (downsample.py:51-70; (D,H,W) = INPUT dims, idx the
 * window position of the first maximum per output element, NULL when no gradient is wanted) and UpSampling3D(2) (upsample.py:69;
 * (D,H,W) = COARSE dims), with their gradients; C % 8 == 0, channel-slice views allowed */
int bts_lp_maxpool2_fwd(int dtype, const void* x, void* y, uint8_t* idx, int N, int D, int H, int W, int C, int ldx, int ldy,
                        bts_stream_t stream);
int bts_lp_maxpool2_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int N, int D, int H, int W, int C, int lddy, int lddx,
                        int accumulate, bts_stream_t stream);
int bts_lp_upsample2_fwd(int dtype, const void* x, void* y, int N, int D, int H, int W, int C, int ldx, int ldy, bts_stream_t stream);
int bts_lp_upsample2_bwd(int dtype, const void* dy, void* dx, int N, int D, int H, int W, int C, int lddy, int lddx, int accumulate,
                         bts_stream_t stream);
/* output head (decoder.py:55-63): y = sigmoid(x . W + b), W (C, K <= 4) fp32, y fp32 (the label map is taken from it) */
int bts_lp_head(int dtype, const void* x, const float* w, const float* bias, float* y, long nvox, int C, int ldx, int K, int sigmoid,
                bts_stream_t stream);
/* output head backward (decoder.py:55-63 under autodiff; train.py:142-151): dpre = dL/d(x . W + b) (nvox, K) fp32 (bts_sigmoid_bwd) ->
 * dx (nvox, C) in the storage type, dw (C, K) and db (K) fp32 (+= when accumulate), all from ONE pass over x.  C in {16,32,64}, K <= 4 */
long bts_lp_head_bwd_workspace(int C, int K);
int bts_lp_head_bwd(int dtype, const void* x, const float* dpre, const float* w, void* dx, float* dw, float* db, void* workspace,
                    long workspace_bytes, long nvox, int C, int ldx, int lddx, int K, int accumulate, bts_stream_t stream);

/* ---- data-parallel exchange over RCCL on device buffers (the reference is single-device, train.py:138) --------------------
 * Sharding train.py:140-152 by sample needs three collectives: C1 the sum of the flat fp32 gradient buffer over the ranks (in a few
 * large buckets: xGMI is point to point, a ring is per-link bound, few large messages win), C2 rank `root`'s parameters on every rank
 * after initialisation / load, C3 the sum of the raw loss sums (bts_loss_sums) and of the Dice table, which the reference sums over the
 * batch axis (util.py:11,18-20).  comm = the caller's ncclComm_t (one process per GPU); every call is enqueued on `stream` and returns
 * BTS_OK, a BTS_ERR_* code, or BTS_ERR_RCCL_BASE - ncclResult_t.  RCCL is bound on first use by dlopen (the copy the process already holds,
 * else the system one): the library has no link-time dependency on it; bts_dp_available says whether it was found (BTS_ERR_UNSUPPORTED
 * from the others when not).  bts_dp_comm_* are conveniences over ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy for hosts that do
 * not link RCCL themselves (unique id: bts_dp_unique_id_bytes() bytes, made on one rank and handed to the others by the host's own means).
 * bts_amd.parallel issues the same three collectives through torch.distributed, whose 'nccl' backend is RCCL. */
int bts_dp_available(void);
long bts_dp_unique_id_bytes(void);
int bts_dp_comm_unique_id(void* id_out);
int bts_dp_comm_init(void** comm_out, int nranks, const void* id, int rank);
int bts_dp_comm_destroy(void* comm);
/* C1: in-place sum over the ranks of the buckets [bucket_off[b], bucket_off[b] + bucket_len[b]) (elements; host arrays) of the flat
 * gradient buffer, one RCCL group per call.  Call it once per finished bucket from inside the backward for the overlapped form. */
int bts_dp_allreduce_buckets(void* comm, float* flat_grads, const long* bucket_off, const long* bucket_len, int nbuckets, bts_stream_t stream);
/* C2 */
int bts_dp_broadcast_params(void* comm, float* flat_params, long n, int root, bts_stream_t stream);
/* C3: in-place sum of n fp64 values */
int bts_dp_allreduce_small(void* comm, double* sums, int n, bts_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BTS_HIP_H */
