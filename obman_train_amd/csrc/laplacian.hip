// K9 - template-Laplacian regulariser, forward + backward (gfx950).
//
// Replaces LaplacianLoss / Laplacian (laplacianloss.py:24-150): the reference builds a (B*N)^2 block-diagonal SciPy CSR,
// copies the vertices to the host every step, multiplies on the CPU and copies back.  The matrix is the SAME fixed
// N x N cotangent stencil (template sphere) for every sample, so it lives on the device once as CSR and the product
// is a gather of <= 7 neighbours per vertex: loss = mean_{b,i} || sum_j L_ij x_bj ||_2.
// Backward (L symmetric, laplacianloss.py:137-150): u_bi = g * Lx_bi / (||Lx_bi|| * B*N), grad = L u.
// Deterministic fixed-order reductions; latency / L2 bound (12 KB per sample).
#include "common.h"
#include "../../include/obman_hip.h"

namespace {

__device__ __forceinline__ void spmv_row(const int* __restrict__ rp, const int* __restrict__ ci, const float* __restrict__ va,
                                         const float* __restrict__ xb, int i, float& ox, float& oy, float& oz) {
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int k = rp[i]; k < rp[i + 1]; ++k) {
    const float w = va[k];
    const float* p = xb + (size_t)ci[k] * 3;
    ax = __fmaf_rn(w, p[0], ax); ay = __fmaf_rn(w, p[1], ay); az = __fmaf_rn(w, p[2], az);
  }
  ox = ax; oy = ay; oz = az;
}

// Lx [B,N,3] and per-block partial sums of the row norms
__global__ __launch_bounds__(256) void lap_fwd_kernel(const int* __restrict__ rp, const int* __restrict__ ci, const float* __restrict__ va,
                                                      const float* __restrict__ X, int N, float* __restrict__ Lx,
                                                      float* __restrict__ partial) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float nrm = 0.f;
  if (i < N) {
    float x, y, z;
    spmv_row(rp, ci, va, X + (size_t)b * N * 3, i, x, y, z);
    float* o = Lx + ((size_t)b * N + i) * 3;
    o[0] = x; o[1] = y; o[2] = z;
    nrm = sqrtf(x * x + y * y + z * z);
  }
  nrm = obman_wave_sum(nrm);
  __shared__ float red[4];
  if (lane == 0) red[wave] = nrm;
  __syncthreads();
  if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void lap_finalize_kernel(const float* __restrict__ partial, int n, float inv_count, float* __restrict__ loss) {
  float s = 0.f;
  for (int k = threadIdx.x; k < n; k += 64) s += partial[k];
  s = obman_wave_sum(s);
  if (threadIdx.x == 0) loss[0] = s * inv_count;
}

// u = g * Lx / (|Lx| * count)  (in place over Lx's copy), then grad = L u
__global__ __launch_bounds__(256) void lap_unit_kernel(const float* __restrict__ Lx, const float* __restrict__ g, float inv_count, long n,
                                                       float* __restrict__ U) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const float x = Lx[r * 3], y = Lx[r * 3 + 1], z = Lx[r * 3 + 2];
  const float nrm = sqrtf(x * x + y * y + z * z);
  const float s = nrm > 0.f ? g[0] * inv_count / nrm : 0.f;  // torch.norm's sub-gradient at 0 is 0
  U[r * 3] = s * x; U[r * 3 + 1] = s * y; U[r * 3 + 2] = s * z;
}

__global__ __launch_bounds__(256) void lap_apply_kernel(const int* __restrict__ rp, const int* __restrict__ ci, const float* __restrict__ va,
                                                        const float* __restrict__ U, int N, float* __restrict__ G) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float x, y, z;
  spmv_row(rp, ci, va, U + (size_t)b * N * 3, i, x, y, z);
  float* o = G + ((size_t)b * N + i) * 3;
  o[0] = x; o[1] = y; o[2] = z;
}

}  // namespace

extern "C" {

/* row_ptr [N+1], col [nnz], val [nnz]: CSR of the symmetric template Laplacian.  Lx [B,N,3] is saved for the backward;
 * partial: B * ceil(N/256) floats of scratch. */
int obman_laplacian_fwd(const int* row_ptr, const int* col, const float* val, const float* verts, int B, int N, float* Lx,
                        float* partial, float* loss, obman_stream_t stream) {
  if (!row_ptr || !col || !val || !verts || !Lx || !partial || !loss || B <= 0 || N <= 0) return -1;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(obman_cdiv(N, 256), B);
  lap_fwd_kernel<<<grid, 256, 0, st>>>(row_ptr, col, val, verts, N, Lx, partial);
  OBMAN_LAUNCH_CHECK();
  lap_finalize_kernel<<<1, 64, 0, st>>>(partial, (int)(grid.x * grid.y), 1.f / ((float)B * (float)N), loss);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

/* g_loss: DEVICE scalar.  scratch [B,N,3]; grad [B,N,3] overwritten. */
int obman_laplacian_bwd(const int* row_ptr, const int* col, const float* val, const float* Lx, const float* g_loss, int B, int N,
                        float* scratch, float* grad, obman_stream_t stream) {
  if (!row_ptr || !col || !val || !Lx || !g_loss || !scratch || !grad || B <= 0 || N <= 0) return -1;
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)B * N;
  lap_unit_kernel<<<obman_cdiv(n, 256), 256, 0, st>>>(Lx, g_loss, 1.f / ((float)B * (float)N), n, scratch);
  OBMAN_LAUNCH_CHECK();
  lap_apply_kernel<<<dim3(obman_cdiv(N, 256), B), 256, 0, st>>>(row_ptr, col, val, scratch, N, grad);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
