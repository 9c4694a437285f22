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
                      float* min_gt, int* idx_gt, void* ws, long ws_bytes, obman_stream_t stream);

/* Backward of obman_chamfer_fwd: g_loss_1 [B], g_loss_2 [B] -> grad_preds [B,Np,3], grad_gts
 * [B,Ng,3] (either may be NULL). */
int obman_chamfer_bwd(const float* preds, const float* gts, int B, int Np, int Ng,
                      const int* idx_pred, const int* idx_gt, const float* g_loss_1, const float* g_loss_2,
                      float* grad_preds, float* grad_gts, obman_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OBMAN_HIP_H */
