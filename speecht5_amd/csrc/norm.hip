// LayerNorm forward/backward + column reductions (wave64 row reductions, 16-byte vector IO).
// Replaces fairseq LayerNorm (F.layer_norm) at encoder.py:226-227, transformer_layer.py:124,132,
// 350,378,402, speech_encoder_prenet.py:174 and the bias / affine-parameter gradient reductions
// of their autograd backward passes.  HBM-bound: one read + one write per element in forward;
// backward reads x and dy once for dx and once more for the (deterministic, two-stage)
// dgamma/dbeta column reduction.
#include "common.h"
#include "../../include/speecht5_hip.h"

namespace {

constexpr int MAXCH = 8;  // up to 8 chunks x 512 columns per wave pass => cols <= 4096 held in registers

template <typename T, int NCH>
__device__ __forceinline__ void load_row(const T* row, int cols, int lane, bool vec, float (&v)[NCH][8]) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = i * 512 + lane * 8;
    if (c < cols) {
      if (vec) load8f<T>(row + c, v[i]);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = (c + e < cols) ? Elem<T>::to_f(row[c + e]) : 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
}
template <typename T, int NCH>
__device__ __forceinline__ void store_row(T* row, int cols, int lane, bool vec, const float (&v)[NCH][8]) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = i * 512 + lane * 8;
    if (c < cols) {
      if (vec) store8f<T>(row + c, v[i]);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (c + e < cols) row[c + e] = Elem<T>::from_f(v[i][e]);
      }
    }
  }
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     long long rows, int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bool vec = (cols % 8) == 0;
  float v[NCH][8];
  load_row<T, NCH>(x + row * cols, cols, lane, vec, v);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[i][e];
  const float mu = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = i * 512 + lane * 8 + e;
      const float d = c < cols ? v[i][e] - mu : 0.f;
      q += d * d;
    }
  const float rs = rsqrtf(wave_sum(q) / (float)cols + eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = i * 512 + lane * 8 + e;
      if (c < cols) v[i][e] = (v[i][e] - mu) * rs * gamma[c] + beta[c];
    }
  store_row<T, NCH>(y + row * cols, cols, lane, vec, v);
  if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, T* __restrict__ dx,
                                                        long long rows, int cols) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bool vec = (cols % 8) == 0;
  float xv[NCH][8], gv[NCH][8];
  load_row<T, NCH>(x + row * cols, cols, lane, vec, xv);
  load_row<T, NCH>(dy + row * cols, cols, lane, vec, gv);
  const float mu = mean[row], rs = rstd[row];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = i * 512 + lane * 8 + e;
      if (c < cols) {
        const float xh = (xv[i][e] - mu) * rs;
        const float g = gv[i][e] * gamma[c];
        xv[i][e] = xh; gv[i][e] = g;
        s1 += g; s2 += g * xh;
      }
    }
  s1 = wave_sum(s1) / (float)cols;
  s2 = wave_sum(s2) / (float)cols;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) gv[i][e] = rs * (gv[i][e] - s1 - xv[i][e] * s2);
  store_row<T, NCH>(dx + row * cols, cols, lane, vec, gv);
}

// ---- column reductions: out[c] (+)= scale * sum_r f(r, c) --------------------------------------
// MODE 0: f = x            MODE 1: f = dy * (x - mean[r]) * rstd[r]   (LayerNorm dgamma)
// Stage 1: grid (ceil(cols/256), nsplit), 256 threads = 32 column-groups(8 cols) x 8 row lanes.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, float* __restrict__ part,
                                                        long long rows, int cols, long long ld) {
  __shared__ float red[8][256 + 8];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 256 + tx * 8;
  const bool vec = (ld % 8) == 0 && c0 + 8 <= cols;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (c0 < cols) {
    for (long long r = (long long)blockIdx.y * 8 + ty; r < rows; r += (long long)gridDim.y * 8) {
      float a[8], b[8];
      if (vec) load8f<T>(x + r * ld + c0, a);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (c0 + e < cols) ? Elem<T>::to_f(x[r * ld + c0 + e]) : 0.f;
      }
      if (MODE == 1) {
        if (vec) load8f<T>(dy + r * ld + c0, b);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) b[e] = (c0 + e < cols) ? Elem<T>::to_f(dy[r * ld + c0 + e]) : 0.f;
        }
        const float mu = mean[r], rs = rstd[r];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += b[e] * (a[e] - mu) * rs;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += a[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = acc[e];
  __syncthreads();
  const int c = threadIdx.x;
  if (blockIdx.x * 256 + c < cols) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += red[j][c];
    part[(long long)blockIdx.y * cols + blockIdx.x * 256 + c] = s;
  }
}
// single-launch variant: block partials go straight into out[] with fp32 atomics (cols x nsplit atomics in total)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void colreduce_atomic_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ out,
                                                               long long rows, int cols, long long ld, float scale) {
  __shared__ float red[8][256 + 8];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 256 + tx * 8;
  const bool vec = (ld % 8) == 0 && c0 + 8 <= cols;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (c0 < cols) {
    for (long long r = (long long)blockIdx.y * 8 + ty; r < rows; r += (long long)gridDim.y * 8) {
      float a[8], b[8];
      if (vec) load8f<T>(x + r * ld + c0, a);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (c0 + e < cols) ? Elem<T>::to_f(x[r * ld + c0 + e]) : 0.f;
      }
      if (MODE == 1) {
        if (vec) load8f<T>(dy + r * ld + c0, b);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) b[e] = (c0 + e < cols) ? Elem<T>::to_f(dy[r * ld + c0 + e]) : 0.f;
        }
        const float mu = mean[r], rs = rstd[r];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += b[e] * (a[e] - mu) * rs;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += a[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = acc[e];
  __syncthreads();
  const int c = threadIdx.x;
  if (blockIdx.x * 256 + c < cols) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += red[j][c];
    unsafeAtomicAdd(out + blockIdx.x * 256 + c, s * scale);
  }
}
__global__ void colreduce_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nsplit, int cols,
                                       float scale, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int j = 0; j < nsplit; ++j) s += part[(long long)j * cols + c];
  s *= scale;
  out[c] = accumulate ? out[c] + s : s;
}

int nsplit_for(long long rows) {
  long long n = (rows + 63) / 64;
  if (n < 1) n = 1;
  if (n > 64) n = 64;
  return (int)n;
}

template <typename T, int MODE>
int colreduce(const void* x, const void* dy, const float* mean, const float* rstd, float* out, float* ws,
              long long rows, int cols, long long ld, float scale, int accumulate, hipStream_t s) {
  const int ns = nsplit_for(rows);
  dim3 grid((cols + 255) / 256, ns);
  if (accumulate) {  // += : one launch, block partials combined with fp32 atomics
    hipLaunchKernelGGL((colreduce_atomic_kernel<T, MODE>), grid, dim3(256), 0, s, (const T*)x, (const T*)dy, mean, rstd,
                       out, rows, cols, ld, scale);
  } else {
    hipLaunchKernelGGL((colreduce_kernel<T, MODE>), grid, dim3(256), 0, s, (const T*)x, (const T*)dy, mean, rstd, ws,
                       rows, cols, ld);
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, ws, out, ns, cols, scale,
                       accumulate);
  }
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

}  // namespace

extern "C" int st5_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                 float* rstd, int64_t rows, int32_t cols, float eps, int dtype, void* stream) {
  if (!x || !y || !gamma || !beta || rows < 0 || cols <= 0 || cols > MAXCH * 512) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype != ST5_BF16 && dtype != ST5_F32) return ST5_ERR_ARG;
#define LNF(TT, NCH)                                                                                              \
  hipLaunchKernelGGL((ln_fwd_kernel<TT, NCH>), grid, dim3(256), 0, s, (const TT*)x, gamma, beta, (TT*)y, mean, rstd, \
                     (long long)rows, cols, eps)
#define LNF_T(TT)                                                                                \
  do {                                                                                           \
    if (cols <= 512) LNF(TT, 1); else if (cols <= 1024) LNF(TT, 2); else if (cols <= 2048) LNF(TT, 4); else LNF(TT, 8); \
  } while (0)
  if (dtype == ST5_BF16) LNF_T(bf16_t); else LNF_T(float);
#undef LNF_T
#undef LNF
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int64_t st5_layernorm_bwd_ws_bytes(int64_t rows, int32_t cols) {
  return (int64_t)nsplit_for(rows) * cols * sizeof(float);
}

extern "C" int st5_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                 const float* rstd, void* dx, float* dgamma, float* dbeta, void* ws, int64_t rows,
                                 int32_t cols, int dtype, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || rows < 0 || cols <= 0 || cols > MAXCH * 512) return ST5_ERR_ARG;
  if ((dgamma || dbeta) && !ws) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  // parameter gradients first (dx may alias dy)
  int rc = ST5_OK;
  if (dtype == ST5_BF16) {
    if (dgamma) rc |= colreduce<bf16_t, 1>(x, dy, mean, rstd, dgamma, (float*)ws, rows, cols, cols, 1.f, 1, s);
    if (dbeta) rc |= colreduce<bf16_t, 0>(dy, nullptr, nullptr, nullptr, dbeta, (float*)ws, rows, cols, cols, 1.f, 1, s);
  } else if (dtype == ST5_F32) {
    if (dgamma) rc |= colreduce<float, 1>(x, dy, mean, rstd, dgamma, (float*)ws, rows, cols, cols, 1.f, 1, s);
    if (dbeta) rc |= colreduce<float, 0>(dy, nullptr, nullptr, nullptr, dbeta, (float*)ws, rows, cols, cols, 1.f, 1, s);
  } else return ST5_ERR_ARG;
  if (rc) return rc;
  if (dx) {
    dim3 grid((unsigned)((rows + 3) / 4));
#define LNB(TT, NCH)                                                                                         \
  hipLaunchKernelGGL((ln_bwd_dx_kernel<TT, NCH>), grid, dim3(256), 0, s, (const TT*)dy, (const TT*)x, gamma, mean, \
                     rstd, (TT*)dx, (long long)rows, cols)
#define LNB_T(TT)                                                                                \
  do {                                                                                           \
    if (cols <= 512) LNB(TT, 1); else if (cols <= 1024) LNB(TT, 2); else if (cols <= 2048) LNB(TT, 4); else LNB(TT, 8); \
  } while (0)
    if (dtype == ST5_BF16) LNB_T(bf16_t); else LNB_T(float);
#undef LNB_T
#undef LNB
    HIP_CHECK_LAUNCH();
  }
  return ST5_OK;
}

// out[c] (+)= scale * sum_r x[r, c]; uses an internal static workspace-free two-stage path via `ws`
// passed through the trailing part of `out`?  No: colsum allocates nothing -- the caller provides
// ws through st5_colsum_ws (kept simple: ws is a dedicated per-stream buffer owned by the host side).
extern "C" int st5_colsum_ws(const void* x, float* out, void* ws, int64_t rows, int32_t cols, int64_t ld, float scale,
                             int32_t accumulate, int dtype, void* stream) {
  if (!x || !out || !ws || rows < 0 || cols <= 0) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == ST5_BF16)
    return colreduce<bf16_t, 0>(x, nullptr, nullptr, nullptr, out, (float*)ws, rows, cols, ld, scale, accumulate, s);
  if (dtype == ST5_F32)
    return colreduce<float, 0>(x, nullptr, nullptr, nullptr, out, (float*)ws, rows, cols, ld, scale, accumulate, s);
  return ST5_ERR_ARG;
}
extern "C" int64_t st5_colsum_ws_bytes(int64_t rows, int32_t cols) {
  return (int64_t)nsplit_for(rows) * cols * sizeof(float);
}
