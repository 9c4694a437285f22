/* obman_hip.h - C-ABI of the MI355X (gfx950) mesh-loss kernels.
 *
 * The reference (hassony2/obman_train) has no FFI: its hot path is a chain of stock torch ops
 * inside HandNet.forward (mano_train/networks/handnet.py:198-392).  This header is the boundary a
 * maintainer binds instead (ctypes stub in INTEGRATION.md): one launcher per kernel and direction,
 * raw device pointers + sizes + a HIP stream, int status (0 = ok, >0 = hipError_t, <0 = bad
 * argument).  The caller owns every buffer; launchers are stream-ordered, stateless, allocate
 * nothing and never synchronise.  All tensors are dense row-major fp32 unless stated; indices int32.
 *
 * Each entry cites the reference op sequence it replaces.
 */
#ifndef OBMAN_HIP_H
#define OBMAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* obman_stream_t; /* hipStream_t */

/* Library / device probe.  Returns the ABI version (int) - used by the loader's self-check. */
int obman_abi_version(void);

/* ---- K1/K2/K3: brute-force pair-min -----------------------------------------------------------
 * Replaces batch_pairwise_dist + torch.min over dim 1 / dim 2
 * (atlasutils.py:11-39 ChamferLoss; contactloss.py:60-79,164-166; handnet.py:353-357).
 * x [B,Nx,3], y [B,Ny,3].  min_x[b,i] = min_j |x_i - y_j|^2, idx_x = argmin (first index on ties);
 * min_y / idx_y likewise over i.  A direction whose min pointer is NULL is skipped (idx may be
 * NULL independently).  Distances use the direct-difference form (no |x|^2+|y|^2-2xy cancellation). */
int obman_pairmin_fwd(const float* x, const float* y, int B, int Nx, int Ny,
                      float* min_x, int* idx_x, float* min_y, int* idx_y,
                      void* ws, long ws_bytes, obman_stream_t stream);

/* Optional scratch for obman_pairmin_fwd / obman_chamfer_fwd: with ws >= this many bytes a long
 * reference set facing few queries (e.g. 600 GT points vs 64 050 predicted vertices) is split over
 * blocks and merged with 64-bit atomicMin.  ws == NULL is always valid (no split). */
long obman_pairmin_ws_bytes(int B, int Nx, int Ny);

/* Backward of obman_pairmin_fwd w.r.t. both point sets.  g_min_x [B,Nx] / g_min_y [B,Ny] are the
 * upstream gradients of the minima (NULL = zeros).  grad_x [B,Nx,3] / grad_y [B,Ny,3] (NULL = not
 * wanted) are overwritten.  Deterministic: the scatter side is an owner-scan, no float atomics. */
int obman_pairmin_bwd(const float* x, const float* y, int B, int Nx, int Ny,
                      const int* idx_x, const int* idx_y, const float* g_min_x, const float* g_min_y,
                      float* grad_x, float* grad_y, obman_stream_t stream);

/* Fused ChamferLoss.forward (atlasutils.py:11-18) with the reference's argument order:
 * P = dist(gts, preds); loss_1[b] = mean_j min_i P (per pred), loss_2[b] = mean_i min_j P (per gt).
 * preds [B,Np,3], gts [B,Ng,3] -> loss_1 [B], loss_2 [B]; idx_pred [B,Np] (nearest gt of each pred),
 * idx_gt [B,Ng] (nearest pred of each gt) and the scratch minima min_pred [B,Np], min_gt [B,Ng]
 * are saved for the backward. */
int obman_chamfer_fwd(const float* preds, const float* gts, int B, int Np, int Ng,
                      float* loss_1, float* loss_2, float* min_pred, int* idx_pred,
                      float* min_gt, int* idx_gt, void* ws, long ws_bytes,
                      void* sync, long sync_bytes, obman_stream_t stream);

/* Optional single-launch mode of obman_chamfer_fwd for point sets that fit LDS (both <= 1024 points, e.g. 642 x 600): with
 * `sync` >= this many bytes the per-sample means are produced inside the pair-min launch by the last block of each sample to
 * finish (`sync` = 4096 arrival counters, then the published partial sums; B <= 4096).  CONTRACT: the caller zero-initialises `sync` ONCE
 * (hipMemset at allocation); every call leaves it zero again, so one buffer per stream serves all later calls.  It must not be
 * shared by calls that can run concurrently (different streams).  Returns 0 when the sizes take the general path (then `sync`
 * is ignored and may be NULL); sync == NULL is always valid (the means then cost a second, tiny launch). */
long obman_chamfer_sync_bytes(int B, int Np, int Ng);

/* Backward of obman_chamfer_fwd: g_loss_1 [B], g_loss_2 [B] -> grad_preds [B,Np,3], grad_gts
 * [B,Ng,3] (either may be NULL). */
int obman_chamfer_bwd(const float* preds, const float* gts, int B, int Np, int Ng,
                      const int* idx_pred, const int* idx_gt, const float* g_loss_1, const float* g_loss_2,
                      float* grad_preds, float* grad_gts, obman_stream_t stream);

/* ---- K7: MANO linear-blend skinning ------------------------------------------------------------
 * Replaces the external manopth.ManoLayer.forward the reference calls at manobranch.py:92-105,
 * 170-182 (algorithm: SURVEY App. B; MANO parity unpinned - manopth is not vendored).
 * model_right / model_left: packed fp32 model blobs of obman_mano_model_floats() floats each (layout
 * in csrc/mano_lbs.hip, built by obman_train_amd/mano_model.py).  side [B] int32: 0 -> right model,
 * 1 -> left model (NULL = all right; replaces the boolean-mask split/re-assembly of
 * manobranch.py:133-207).  use_pca selects the pose input: 1 -> pose [B, 3+ncomps] (root axis-angle + PCA coefficients),
 * 0 -> [B,48] axis-angle, 2 -> [B,16,3,3] rotation matrices used as given (the reference's mano_use_pca=False path,
 * manobranch.py:52-54,126-128); betas [B,10] or NULL (zeros).
 * center_idx in [-1,20] (-1 = no centring); verts [B,778,3] mm, joints [B,21,3] mm; state
 * [B, OBMAN_MANO_STATE_FLOATS] is saved for the backward (NULL = inference only). */
#define OBMAN_MANO_STATE_FLOATS 2768
int obman_mano_model_floats(void);
int obman_mano_state_floats(void);
int obman_mano_lbs_fwd(const float* model_right, const float* model_left, const int* side, const float* pose,
                       const float* betas, int B, int ncomps, int use_pca, int center_idx, int root_palm,
                       float* verts, float* joints, float* state, obman_stream_t stream);

/* Backward: g_verts [B,778,3] / g_joints [B,21,3] (either NULL = zeros) -> g_pose [B,npose] (npose = 3+ncomps | 48 | 144),
 * g_betas [B,10] (NULL = not wanted).  scratch: obman_mano_bwd_scratch_floats(B) floats (per-tile partial sums).
 * Deterministic (fixed reduction trees, no atomics). */
int obman_mano_bwd_scratch_floats(int B);
int obman_mano_lbs_bwd(const float* model_right, const float* model_left, const int* side, const float* state,
                       const float* g_verts, const float* g_joints, int B, int ncomps, int use_pca, int center_idx,
                       int root_palm, float* g_pose, float* g_betas, float* scratch, obman_stream_t stream);

/* ---- K4: ray-parity inside test -----------------------------------------------------------------
 * Replaces batch_mesh_contains_points (contactutils.py:62-159) + the obj_verts[:, faces] gather
 * (contactloss.py:169).  points [B,P,3], verts [B,Nv,3] (object mesh vertices), faces [F,3] int32
 * shared by the batch -> hits [B,P] int32 = number of triangles the fixed-direction ray crosses;
 * exterior <=> hits even (contactutils.py:158).  No gradient (inputs are detached, contactloss.py:170). */
int obman_mesh_contains_fwd(const float* points, const float* verts, const int* faces, int B, int P, int Nv,
                            int F, int* hits, obman_stream_t stream);
/* Multi-patch templates (extension of this build, BASELINE.json configs 3/5; no reference counterpart): faces = G <= 32
 * consecutive groups of group_faces triangles, each a closed patch surface -> parity_bits [B,P] int32, bit g = parity of
 * the crossings with patch g.  Inside the union of the patches <=> parity_bits != 0 (the OR of the per-patch tests of
 * contactutils.py:158; the parity of the TOTAL count would call a point inside two overlapping patches exterior). */
int obman_mesh_contains_groups_fwd(const float* points, const float* verts, const int* faces, int B, int P, int Nv,
                                   int F, int group_faces, int* parity_bits, obman_stream_t stream);
/* The same outputs from the ALL-PAIRS kernel (every (point, triangle) pair evaluated, the formulation of
 * contactutils.py:62-159 itself; group_faces = 0: hit counts, > 0: per-patch parity bits).  The two entry points above cull
 * pairs with a 2-D grid in the plane normal to the fixed ray (contactutils.py:65) and must return bit-identical words; this one
 * is the checker they are tested against (tests/test_contains_binned_gpu.py), not a fallback - the product never calls it. */
int obman_mesh_contains_bruteforce_fwd(const float* points, const float* verts, const int* faces, int B, int P, int Nv,
                                       int F, int group_faces, int* hits, obman_stream_t stream);

/* ---- K5: contact / penetration loss tail ---------------------------------------------------------
 * Replaces contactloss.py:173-308 (after pair-min and inside test).  hand [B,V,3], obj [B,N,3],
 * idx21/mins21 [B,V] from obman_pairmin_fwd(hand,obj), hits [B,V] from obman_mesh_contains_fwd.
 * zone_mode 0=all, 1=list (tips: ids in zone_ids[0:zone_offsets[1]]), 2=zones (per-zone arg-min of
 * mins21; zone z = zone_ids[zone_offsets[z]:zone_offsets[z+1]]).  *_mode 0=dist_sq 1=dist 2=dist_tanh.
 * Outputs: attr_mask/rep_mask [B,V] uint8, contact_points [B,V,3], partials [B,8] scratch,
 * out [8] = {missed_loss, penetr_loss, max_penetr, mean_penetr, n_missed, n_penetr, 0, 0}. */
int obman_contact_fwd(const float* hand, const float* obj, const int* idx21, const float* mins21, const int* hits,
                      int B, int V, int N, const int* zone_ids, const int* zone_offsets, int n_zones, int zone_mode,
                      int contact_mode, float contact_thresh, int collision_mode, float collision_thresh,
                      unsigned char* attr_mask, unsigned char* rep_mask, float* contact_points, float* partials,
                      float* out, obman_stream_t stream);

/* Backward: g_missed / g_penetr are DEVICE scalars (upstream grads of out[0], out[1]; NULL = 0).
 * target 0=all 1=obj 2=hand (contact_target, contactloss.py:176-203).  grad_hand [B,V,3],
 * grad_obj [B,N,3] (either NULL = not wanted) are overwritten. */
int obman_contact_bwd(const float* hand, const float* obj, const int* idx21, const unsigned char* attr_mask,
                      const unsigned char* rep_mask, const float* out, const float* g_missed, const float* g_penetr,
                      int B, int V, int N, int contact_mode, float contact_thresh, int collision_mode,
                      float collision_thresh, int target, float* grad_hand, float* grad_obj, obman_stream_t stream);

/* ---- K6: AtlasNet PointGenCon decoder (fp32 MFMA; optional bf16 MFMA) ---------------------------------------------------
 * Replaces atlasbranch.py:117-132 (grid/feature repeat + concat) + PointGenCon.forward (atlasutils.py:65-75):
 * 4 pointwise convs C1 -> C1 -> C1/2 -> C1/4 -> 3 with BatchNorm1d + ReLU after the first three, x out_factor.
 * C1 = 3 + feature size (515).  grid [N,3] is the shared sphere template (or [B,N,3], see grid_per_sample), feat [B,C1-3]; weights are the conv
 * weights [Cout,Cin] (kernel size 1 squeezed), BN gamma/beta/running stats per layer.  training != 0: batch
 * statistics (running stats updated in place with `momentum`); else running statistics.  out [B,N,3].
 * ws: obman_pointgen_ws_floats(p, 0) floats, written by fwd and read by bwd; ws2: ..._ws_floats(p, 1) scratch. */
typedef struct {
  int B, N, C1, training;
  float eps, momentum, out_factor;
  int mfma_bf16; /* 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32); 1: operands rounded to bf16, v_mfma_f32_32x32x16_bf16, fp32
                    accumulation and fp32 BatchNorm statistics (BASELINE configs[2] flavour) */
  int grid_per_sample; /* 0: grid [N,3] shared by the batch (AtlasBranch.forward_inference, atlasbranch.py:110-150);
                          1: grid [B,N,3], one point set per sample (AtlasBranch.forward's random sphere samples,
                          atlasbranch.py:78-108); fp32 contraction only */
  int reserved_;
  const float *grid, *feat;
  const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4;
  const float* bn_w[3];
  const float* bn_b[3];
  float* bn_rm[3];
  float* bn_rv[3];
} obman_pointgen_params;

typedef struct {  /* gradient outputs of obman_pointgen_bwd, same shapes as the parameters; feat may be NULL */
  float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4;
  float* bn_w[3];
  float* bn_b[3];
  float* feat;
} obman_pointgen_grads;

long obman_pointgen_ws_floats(const obman_pointgen_params* p, int backward);
int obman_pointgen_fwd(const obman_pointgen_params* p, float* out, float* ws, obman_stream_t stream);
int obman_pointgen_bwd(const obman_pointgen_params* p, const float* g_out, const float* ws, float* ws2,
                       const obman_pointgen_grads* g, obman_stream_t stream);

/* ---- K8: edge-length regulariser -----------------------------------------------------------------
 * Replaces edge_loss (atlasbranch.py:153-167).  verts [B,N,3], faces [F,3] int32 -> loss [1] =
 * mean over samples and the 3F face edges of |squared edge length - per-sample mean|.  stats [3*B] fwd -> bwd.
 * g_loss is a DEVICE scalar; grad [B,N,3] is overwritten (owner scan over the faces, deterministic). */
int obman_edge_loss_fwd(const float* verts, const int* faces, int B, int N, int F, float* loss, float* stats, obman_stream_t stream);
int obman_edge_loss_bwd(const float* verts, const int* faces, int B, int N, int F, const float* stats, const float* g_loss,
                        float* grad, obman_stream_t stream);

/* ---- K9: template-Laplacian regulariser -----------------------------------------------------------
 * Replaces LaplacianLoss / Laplacian (laplacianloss.py:24-150; CPU SciPy round trip every step in the reference).
 * CSR (row_ptr [N+1], col, val) of the fixed symmetric N x N cotangent Laplacian of the template sphere, shared by all
 * samples; verts [B,N,3] -> loss [1] = mean_{b,i} ||(L x_b)_i||_2.  Lx [B,N,3] fwd -> bwd; partial: B*ceil(N/256). */
int obman_laplacian_fwd(const int* row_ptr, const int* col, const float* val, const float* verts, int B, int N, float* Lx,
                        float* partial, float* loss, obman_stream_t stream);
int obman_laplacian_bwd(const int* row_ptr, const int* col, const float* val, const float* Lx, const float* g_loss, int B, int N,
                        float* scratch, float* grad, obman_stream_t stream);

/* ---- fused BatchNorm2d (+ skip add) (+ ReLU), NHWC fp32 --------------------------------------------
 * Replaces bn -> relu and bn -> (+residual) -> relu of the ResNet blocks (bases/resnet.py:38-52,77-96) - separate
 * memory-bound passes in eager PyTorch.  x, skip, y, dy, dx, dskip are [R, C] row-major (R = B*H*W of a
 * channels_last tensor), C % 64 == 0.  stats [4*C] = mean | rstd | scale | shift (fwd -> bwd).
 * ws: obman_bnact_ws_floats(R, C) floats of scratch. */
long obman_bnact_ws_floats(long R, int C);
int obman_bnact_fwd(const float* x, const float* skip, const float* gamma, const float* beta, float* rmean, float* rvar, long R, int C,
                    int training, float eps, float momentum, int relu, float* y, float* stats, float* ws, obman_stream_t stream);
int obman_bnact_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* stats, long R, int C, int training,
                    int relu, int has_skip, float* dx, float* dgamma, float* dbeta, float* dskip, float* ws, obman_stream_t stream);
/* The same with TWO incoming gradients (ABI 5; relu && has_skip only, dy2 may be NULL): a residual block's output feeds the next
 * block's first convolution and its skip branch (bases/resnet.py:38-54, 76-96), autograd delivers one gradient per consumer, and this entry
 * point adds them while its statistics pass reads them - the separate add over the activation (8 per ResNet-18 step) disappears. */
int obman_bnact_bwd2(const float* x, const float* y, const float* dy, const float* dy2, const float* gamma, const float* stats, long R, int C,
                     int training, int relu, int has_skip, float* dx, float* dgamma, float* dbeta, float* dskip, float* ws, obman_stream_t stream);

/* Stem variant: y_pool = MaxPool2d(3, stride 2, pad 1)(relu(bn(x))) (bases/resnet.py:156-163) without materialising the
 * full-resolution activation.  x [B,H,W,C] NHWC, y_pool / d_pool [B,(H-1)/2+1,(W-1)/2+1,C]; stats as above;
 * ws: obman_bnact_ws_floats(B*H*W, C). */
int obman_bnpool_fwd(const float* x, const float* gamma, const float* beta, float* rmean, float* rvar, int B, int H, int W, int C,
                     int training, float eps, float momentum, float* y_pool, float* stats, float* ws, obman_stream_t stream);
int obman_bnpool_bwd(const float* x, const float* y_pool, const float* d_pool, const float* gamma, const float* stats, int B, int H, int W,
                     int C, int training, float* dx, float* dgamma, float* dbeta, float* ws, obman_stream_t stream);

/* The same four entry points for bf16 ACTIVATIONS (raw bf16 bits; x, skip, y, dy, dx, dskip, y_pool, d_pool), used when the encoder
 * runs under bf16 autocast (BASELINE configs[2]).  Parameters, running statistics, stats and ws stay fp32; the arithmetic and the
 * batch statistics are fp32 / fp64 exactly as above, the result is rounded to bf16 (nearest even) once, at the store. */
int obman_bnact_fwd_bf16(const uint16_t* x, const uint16_t* skip, const float* gamma, const float* beta, float* rmean, float* rvar, long R,
                         int C, int training, float eps, float momentum, int relu, uint16_t* y, float* stats, float* ws,
                         obman_stream_t stream);
int obman_bnact_bwd_bf16(const uint16_t* x, const uint16_t* y, const uint16_t* dy, const float* gamma, const float* stats, long R, int C,
                         int training, int relu, int has_skip, uint16_t* dx, float* dgamma, float* dbeta, uint16_t* dskip, float* ws,
                         obman_stream_t stream);
int obman_bnact_bwd2_bf16(const uint16_t* x, const uint16_t* y, const uint16_t* dy, const uint16_t* dy2, const float* gamma, const float* stats,
                          long R, int C, int training, int relu, int has_skip, uint16_t* dx, float* dgamma, float* dbeta, uint16_t* dskip,
                          float* ws, obman_stream_t stream);
/* The stem's forward additionally writes amax [B,Ho,Wo,C] bytes: the 3x3 window tap (0..8, row-major; 255 = none, all taps <= 0) of
 * the FIRST maximum - torch's max_pool2d arg-max - and the backward routes d_pool by it.  (The fp32 entry points route to the taps
 * that equal the pooled value; bf16 inputs tie far too often for that.) */
int obman_bnpool_fwd_bf16(const uint16_t* x, const float* gamma, const float* beta, float* rmean, float* rvar, int B, int H, int W, int C,
                          int training, float eps, float momentum, uint16_t* y_pool, uint8_t* amax, float* stats, float* ws,
                          obman_stream_t stream);
int obman_bnpool_bwd_bf16(const uint16_t* x, const uint8_t* amax, const uint16_t* d_pool, const float* gamma, const float* stats, int B,
                          int H, int W, int C, int training, uint16_t* dx, float* dgamma, float* dbeta, float* ws, obman_stream_t stream);

/* ---- K10: GPU-side image input stream -----------------------------------------------------------
 * Replaces the CPU pixel pipeline of HandDataset.get_sample (handobjectdatasets/handataset.py:373-405): per sample
 * Gaussian blur (PIL ImageFilter.GaussianBlur = 3+3 extended-box passes) and colour jitter (imgtrans.py:31-53:
 * torchvision adjust_brightness / saturation / hue / contrast on uint8, in the sample's shuffled order) of the source
 * image, the affine crop handutils.transform_img (handutils.py:48-60: PIL AFFINE transform, NEAREST, 16.16 fixed
 * point), to_tensor (/255), the optional black frame (handataset.py:390-397) and normalize (handataset.py:399-404).
 * Byte/integer work: bit-identical to the CPU path.
 * src [B, pitch_h, pitch_w, 3] uint8 RGB (PIL raw "RGB": the host stages a plain memcpy of the decoded image), each
 * sample's image in the top-left src_h x src_w corner of its slot.  params: DEVICE array of B records, filled by the host from its RNG draws:
 *   flip      read the source mirrored left-right (Image.FLIP_LEFT_RIGHT, handataset.py:134-135)
 *   A[6]      xin = (A[2] + x*A[0] + y*A[1]) >> 16, yin = (A[5] + x*A[3] + y*A[4]) >> 16   (Geometry.c affine_fixed)
 *   blur_r/ww/fw  integer radius (-1: no blur) and 8.24 weights of the extended box filter (BoxBlur.c)
 *   op[k], factor[k], k < n_ops <= 4   1 brightness, 2 saturation, 3 hue (uses hue_shift, a uint8 increment of H),
 *             4 contrast; factor = blend factor as C float
 * max_blur_r = max over the batch of blur_r (-1 skips the blur kernel; > 8 is rejected), any_contrast = some sample has
 * op 4 (runs the whole-image luma reduction).  mean3 / std3: HOST float[3] or NULL (0.5 / 1, the reference's default).
 * black_pad: frame width in pixels (int(inp_res * 0.2) in the reference) or 0.  out [B,3,res,res] fp32, or
 * [B,res,res,3] when channels_last.  ws: obman_imgstream_ws_bytes(B, pitch_h, pitch_w) bytes of scratch. */
typedef struct obman_img_params {
  int32_t src_h, src_w, flip;
  int32_t A[6];
  int32_t blur_r;
  uint32_t blur_ww, blur_fw;
  int32_t n_ops;
  int32_t op[4];
  float factor[4];
  int32_t hue_shift;
  int32_t reserved[2];
} obman_img_params; /* 24 x 32-bit words */
long obman_imgstream_ws_bytes(int B, int pitch_h, int pitch_w);
int obman_imgstream_fwd(const uint8_t* src, int B, int pitch_h, int pitch_w, const obman_img_params* params, int max_blur_r,
                        int any_contrast, int out_res, int channels_last, int black_pad, const float* mean3, const float* std3,
                        void* ws, float* out, obman_stream_t stream);

/* ---- K11: optimizer step ------------------------------------------------------------------------
 * Replaces torch.optim.Adam(model.parameters()).step() of the reference's training loop (traineval.py:112-127,
 * epochpass3d.py:86-91; defaults nets3dopts.py:249-273): exp_avg / exp_avg_sq / bias-corrected update of every
 * parameter in ONE multi-tensor kernel per <= 64 tensors (plain Adam: L2 weight decay added to the gradient, no amsgrad).
 * `tensors` is a HOST array; every pointer in it is a device pointer to n contiguous fp32 elements in the parameter's own
 * memory order (p, g, m, v alike).  `step` is the tensor's step counter (device fp32, incremented by this call before use,
 * as torch's capturable Adam keeps it).  `shadow_bf16` (NULL = none): receives bf16(p_new) - the copy a bf16-autocast
 * convolution reads, so that no per-step cast kernel is needed (obman_bf16_shadow initialises it).  The hyper-parameters are doubles, as
 * torch holds them: 1 - beta2 is formed in double (0.001, not the 0.00099998713 of fp32 arithmetic on the rounded 0.999f). */
typedef struct obman_adam_tensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  uint16_t* shadow_bf16;
  float* step;
  long n;
} obman_adam_tensor;
int obman_adam_step(const obman_adam_tensor* tensors, int count, double lr, double beta1, double beta2, double eps,
                    double weight_decay, obman_stream_t stream);
int obman_bf16_shadow(const float* src, uint16_t* dst, long n, obman_stream_t stream);

/* ---- K12: objpoints3d = scale * verts + trans ---------------------------------------------------
 * Replaces `scaled = scale.unsqueeze(1) * verts; points = scaled + trans.unsqueeze(1)` (atlasbranch.py:133-138) and its
 * autograd backward (two broadcast products and two [B,N,3] -> [B,1,.] reductions).  verts / out / g / gverts [B,N,3],
 * scale [B] (NULL = 1), trans [B,3] (NULL = 0); gverts / gscale [B] / gtrans [B,3] may each be NULL.  ws: scratch of
 * obman_affine_points_ws_floats(B) floats (fixed-order partial sums: deterministic). */
int obman_affine_points_fwd(const float* verts, const float* scale, const float* trans, int B, int N, float* out,
                            obman_stream_t stream);
long obman_affine_points_ws_floats(int B);
int obman_affine_points_bwd(const float* g, const float* verts, const float* scale, int B, int N, float* gverts, float* gscale,
                            float* gtrans, float* ws, obman_stream_t stream);

/* ---- K13: the mean-squared-error heads ----------------------------------------------------------
 * Replaces the torch_f.mse_loss calls of ManoLoss.compute_loss (manobranch.py:251-318: vertices, joints, shape, pose
 * regulariser) and AtlasLoss.compute_loss (atlasbranch.py:211-228: translation, scale): out[i] = mean((pred_i - target_i)^2)
 * for up to 8 differently sized tensors in one launch (+ a one-block finalize); target NULL = zeros.  Backward:
 * grad_i = 2 / n_i * (pred_i - target_i) * g_out[i] for every term whose grad pointer is set.  `terms` is a HOST array of
 * device pointers; ws: obman_mse_terms_ws_floats() floats. */
typedef struct obman_mse_term {
  const float* pred;
  const float* target;
  float* grad;
  long n;
} obman_mse_term;
long obman_mse_terms_ws_floats(void);
int obman_mse_terms_fwd(const obman_mse_term* terms, int count, float* ws, float* out, obman_stream_t stream);
int obman_mse_terms_bwd(const obman_mse_term* terms, int count, const float* g_out, obman_stream_t stream);

/* ---- K14: ground-truth object statistics --------------------------------------------------------
 * Replaces `centroids = gt.mean(1); centred = gt - centroids.unsqueeze(1); torch.norm(centred, 2, 2).max(1)[0]`
 * (atlasbranch.py:211-222).  gt [B,N,3] -> centroid [B,3], centred [B,N,3], maxnorm [B].  No gradient (targets). */
int obman_gt_object_stats(const float* gt, int B, int N, float* centroid, float* centred, float* maxnorm, obman_stream_t stream);

/* ---- measurement utility (not on the product path) ----------------------------------------------
 * obman_prof_enable(1) makes the launchers bracket their main kernel with HIP events on the launch
 * stream (pool of 8192 records, reset by every enable call); obman_prof_summary synchronises and
 * returns the summed duration / launch count of one kernel id (1 pair-min fwd, 2 pair-min bwd,
 * 3 inside test, 4/5 contact fwd/bwd, 6/7 MANO fwd/bwd, 8/9 decoder fwd/bwd, 10/11 ChamferLoss fwd/bwd,
 * 12/13 the second direction's own launch of a ChamferLoss / pair-min forward at asymmetric sizes).  Used by bench.py for
 * the `roofline` object; off by default. */
int obman_prof_enable(int on);
int obman_prof_summary(int kernel_id, double* total_ms, long* launches);

#ifdef __cplusplus
}
#endif
#endif /* OBMAN_HIP_H */
