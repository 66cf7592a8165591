// HuBERT-style NCE head (SpeechT5/speecht5/models/modules/speech_encoder_postnet.py:56-76 compute_nce):
//   logits[s] = [cos(x_s, e_{t_s}), cos(x_s, e_0), ..., cos(x_s, e_{V-1})] / temp, with -inf at every class whose code-book row
//   is IDENTICAL to the positive's row (the reference builds the [S, V, D] comparison; here a per-class canonical index).
// Pieces (fp32 throughout, as the reference computes this head): row normalisation of the projected frames and of the code
// book (+ canonical index of duplicate rows), the [S, V] cosine GEMM (st5_gemm, by the caller), the logit assembly, and the
// matching gradient kernels.  ~10 launches for what the torch formulation needs ~60 for.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "../../include/speecht5_hip.h"

#define DISPATCH(dtype, CALL_BF, CALL_F)   \
  if (dtype == ST5_BF16) { CALL_BF; }      \
  else if (dtype == ST5_F32) { CALL_F; }   \
  else return ST5_ERR_ARG;

namespace {

constexpr float NORM_EPS = 1e-8f;   // F.normalize(eps=1e-8) as torch.cosine_similarity uses it (:63)

// y[r] = x[r] / max(|x[r]|, eps) (fp32), inv[r] = 1 / max(|x[r]|, eps).  One wave per row.
template <typename T>
__global__ __launch_bounds__(256) void norm_rows_kernel(const T* __restrict__ x, float* __restrict__ y, float* __restrict__ inv, long long rows,
                                                        int cols) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float ss = 0.f;
  for (int c = lane; c < cols; c += 64) { const float v = Elem<T>::to_f(x[r * cols + c]); ss = fmaf(v, v, ss); }
  const float k = 1.f / fmaxf(sqrtf(wave_sum(ss)), NORM_EPS);
  for (int c = lane; c < cols; c += 64) y[r * cols + c] = Elem<T>::to_f(x[r * cols + c]) * k;
  if (lane == 0) inv[r] = k;
}

// dx[r] (+)= inv[r] (dy[r] - y[r] <y[r], dy[r]>)   (gradient of the normalisation; |x| <= eps rows: dx = inv dy)
template <typename T>
__global__ __launch_bounds__(256) void norm_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ inv, const float* __restrict__ dy,
                                                            T* __restrict__ dx, long long rows, int cols, int accumulate) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float dot = 0.f;
  for (int c = lane; c < cols; c += 64) dot = fmaf(y[r * cols + c], dy[r * cols + c], dot);
  dot = wave_sum(dot);
  const float k = inv[r];
  const bool clamped = k >= 1.f / NORM_EPS;
  for (int c = lane; c < cols; c += 64) {
    const float g = k * (dy[r * cols + c] - (clamped ? 0.f : y[r * cols + c] * dot));
    dx[r * cols + c] = Elem<T>::from_f(accumulate ? Elem<T>::to_f(dx[r * cols + c]) + g : g);
  }
}

// canon[c] = smallest c' with e[c'] == e[c] element-wise.  Block per class; thread j tests candidates j, j + 256, ...
__global__ __launch_bounds__(256) void canon_rows_kernel(const float* __restrict__ e, int* __restrict__ canon, int V, int D) {
  __shared__ int best;
  const int c = blockIdx.x;
  if (threadIdx.x == 0) best = c;
  __syncthreads();
  for (int j = threadIdx.x; j < c; j += 256) {
    bool same = true;
    for (int k = 0; k < D && same; ++k) same = e[(long long)j * D + k] == e[(long long)c * D + k];
    if (same) atomicMin(&best, j);
  }
  __syncthreads();
  if (threadIdx.x == 0) canon[c] = best;
}

// logits [S, 1 + V] from sim [S, V]
__global__ __launch_bounds__(256) void nce_logits_kernel(const float* __restrict__ sim, const int* __restrict__ target, const int* __restrict__ canon,
                                                         float* __restrict__ logits, long long S, int V, float inv_temp) {
  const long long s = blockIdx.x;
  const int t = target[s];
  const int ct = canon[t];
  const float* row = sim + s * V;
  float* out = logits + s * (V + 1);
  if (threadIdx.x == 0) out[0] = row[t] * inv_temp;
  for (int c = threadIdx.x; c < V; c += 256) out[1 + c] = canon[c] == ct ? -INFINITY : row[c] * inv_temp;
}

// dsim [S, V] from dlogits [S, 1 + V]
__global__ __launch_bounds__(256) void nce_logits_bwd_kernel(const float* __restrict__ dlogits, const int* __restrict__ target,
                                                             const int* __restrict__ canon, float* __restrict__ dsim, long long S, int V,
                                                             float inv_temp) {
  const long long s = blockIdx.x;
  const int t = target[s];
  const int ct = canon[t];
  const float* g = dlogits + s * (V + 1);
  float* out = dsim + s * V;
  for (int c = threadIdx.x; c < V; c += 256) {
    float v = canon[c] == ct ? 0.f : g[1 + c] * inv_temp;
    if (c == t) v += g[0] * inv_temp;
    out[c] = v;
  }
}

}  // namespace

extern "C" int st5_norm_rows(const void* x, float* y, float* inv, int64_t rows, int32_t cols, int dtype, void* stream) {
  if (!x || !y || !inv || rows < 0 || cols <= 0) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  DISPATCH(dtype, hipLaunchKernelGGL(norm_rows_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, y, inv, (long long)rows, cols),
           hipLaunchKernelGGL(norm_rows_kernel<float>, grid, dim3(256), 0, s, (const float*)x, y, inv, (long long)rows, cols));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_norm_rows_bwd(const float* y, const float* inv, const float* dy, void* dx, int64_t rows, int32_t cols, int32_t accumulate,
                                 int dtype, void* stream) {
  if (!y || !inv || !dy || !dx || rows < 0 || cols <= 0) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  DISPATCH(dtype, hipLaunchKernelGGL(norm_rows_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, y, inv, dy, (bf16_t*)dx, (long long)rows, cols, accumulate),
           hipLaunchKernelGGL(norm_rows_bwd_kernel<float>, grid, dim3(256), 0, s, y, inv, dy, (float*)dx, (long long)rows, cols, accumulate));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_canon_rows(const float* e, int32_t* canon, int32_t V, int32_t D, void* stream) {
  if (!e || !canon || V <= 0 || D <= 0) return ST5_ERR_ARG;
  hipLaunchKernelGGL(canon_rows_kernel, dim3((unsigned)V), dim3(256), 0, (hipStream_t)stream, e, canon, V, D);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_nce_logits(const float* sim, const int32_t* target, const int32_t* canon, float* logits, int64_t S, int32_t V, float temp,
                              void* stream) {
  if (!sim || !target || !canon || !logits || S < 0 || V <= 0 || !(temp > 0.f)) return ST5_ERR_ARG;
  if (S == 0) return ST5_OK;
  hipLaunchKernelGGL(nce_logits_kernel, dim3((unsigned)S), dim3(256), 0, (hipStream_t)stream, sim, target, canon, logits, (long long)S, V, 1.f / temp);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_nce_logits_bwd(const float* dlogits, const int32_t* target, const int32_t* canon, float* dsim, int64_t S, int32_t V, float temp,
                                  void* stream) {
  if (!dlogits || !target || !canon || !dsim || S < 0 || V <= 0 || !(temp > 0.f)) return ST5_ERR_ARG;
  if (S == 0) return ST5_OK;
  hipLaunchKernelGGL(nce_logits_bwd_kernel, dim3((unsigned)S), dim3(256), 0, (hipStream_t)stream, dlogits, target, canon, dsim, (long long)S, V,
                     1.f / temp);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
