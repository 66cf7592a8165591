// BatchNorm1d (training statistics) + tanh + dropout (+ residual) over channels-last rows, forward and backward:
// the non-GEMM part of espnet's Tacotron Postnet as the reference uses it (speech_decoder_postnet.py:39-51,65-70:
// 5 x [Conv1d k5 -> BatchNorm1d -> tanh -> dropout], last block without tanh; `after = before + postnet(before)`).
//
// HBM-bound and tiny ([B*L, 256] fp32 = 5 MB at cfg 2), so the design goal is few launches and fp32 where it matters:
//   * the convolution GEMM hands over its fp32 accumulators (ST5_GEMM_OUT_F32): statistics, normalisation and the whole
//     gradient path between two convolutions stay fp32 -- the BatchNorm backward subtracts the per-channel mean of the
//     incoming gradient, which amplifies any rounding of that gradient (bf16 there cost 20 % relative error in the post-net
//     weight gradients on the Base model, tests/test_fullsize_gpu.py);
//   * forward = partial sums (fp64) -> finalize (mean, rstd, running statistics) -> apply (normalise, tanh, dropout,
//     residual) writing the NEXT convolution's operand directly in its zero-haloed time layout;
//   * backward = partial sums of (g, g*xhat) -> finalize (+= dgamma, dbeta) -> apply, again into the haloed layout the
//     transposed convolution reads.  Deterministic (fixed reduction order, no atomics).
#include "common.h"
#include "../../include/speecht5_hip.h"

namespace {

constexpr int BN_MAXBLK = 64;   // row blocks of the partial-sum kernels

struct BnGeom { int C, cq, rg; };   // channels, channel quads, rows per block iteration
__device__ __forceinline__ BnGeom geom(int C) { BnGeom g; g.C = C; g.cq = C / 4; g.rg = 256 / g.cq; return g; }

__device__ __forceinline__ float act_out(int act, float z) { return act == ACT_TANH ? tanhf(z) : z; }

// partial[blk][c][0..1] (double): sum x, sum x^2 of the block's rows
__global__ __launch_bounds__(256) void bn_fwd_partial_kernel(const float* __restrict__ x, double* __restrict__ partial, long long rows, int C) {
  __shared__ double red[256][8];
  const BnGeom g = geom(C);
  const int tid = threadIdx.x, q = tid % g.cq, ro = tid / g.cq;
  double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  if (ro < g.rg) {
    for (long long r = (long long)blockIdx.x * g.rg + ro; r < rows; r += (long long)gridDim.x * g.rg) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + q * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[e] += (double)v[e]; ss[e] += (double)v[e] * (double)v[e]; }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[tid][e] = s[e]; red[tid][4 + e] = ss[e]; }
  __syncthreads();
  if (tid < g.cq) {
    for (int o = 1; o < g.rg; ++o)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[tid][e] += red[tid + o * g.cq][e];
    double* dst = partial + ((long long)blockIdx.x * C + tid * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { dst[2 * e] = red[tid][e]; dst[2 * e + 1] = red[tid][4 + e]; }
  }
}

// stats[c] = mean, stats[C + c] = rstd.  Training: from the partial sums, running statistics updated as torch does
// (momentum blend, unbiased variance); eval: from the running statistics.
__global__ void bn_fwd_finalize_kernel(const double* __restrict__ partial, int nblk, long long rows, int C, float eps, float momentum,
                                       int training, float* __restrict__ running_mean, float* __restrict__ running_var,
                                       long long* __restrict__ num_batches, float* __restrict__ stats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && training && num_batches) num_batches[0] += 1;
  if (c >= C) return;
  if (!training) {
    stats[c] = running_mean[c];
    stats[C + c] = 1.0f / sqrtf(running_var[c] + eps);
    return;
  }
  double s = 0, ss = 0;
  for (int b = 0; b < nblk; ++b) { s += partial[((long long)b * C + c) * 2]; ss += partial[((long long)b * C + c) * 2 + 1]; }
  const double mean = s / (double)rows;
  double var = ss / (double)rows - mean * mean;
  var = var > 0 ? var : 0;
  stats[c] = (float)mean;
  stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unb = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

struct RowMap { long long L, bstride, off; };   // output element offset of row r: (r / L) * bstride + (r % L) * C + off   (L = 0: r * C)
__device__ __forceinline__ long long map_row(const RowMap& m, long long r, int C) {
  return m.L ? (r / m.L) * m.bstride + (r % m.L) * (long long)C + m.off : r * (long long)C;
}

// y = dropout(act((x - mean) * rstd * gamma + beta)) [+ R]; y in T or fp32; halo rows of the mapped layout zeroed
template <typename T, bool OUTF32>
__global__ __launch_bounds__(256) void bn_fwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const T* __restrict__ R, void* __restrict__ y,
                                                           long long rows, int C, int act, float p, unsigned long long seed, RowMap om,
                                                           long long halo_rows_per_batch) {
  const BnGeom g = geom(C);
  const int tid = threadIdx.x, q = tid % g.cq, ro = tid / g.cq;
  const unsigned int thresh = p > 0.f ? dropout_thresh(p) : 0u;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  if (ro < g.rg) {
    const f32x4 mean = *reinterpret_cast<const f32x4*>(stats + q * 4), rstd = *reinterpret_cast<const f32x4*>(stats + C + q * 4);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + q * 4), be = *reinterpret_cast<const f32x4*>(beta + q * 4);
    for (long long r = (long long)blockIdx.x * g.rg + ro; r < rows; r += (long long)gridDim.x * g.rg) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + q * 4);
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = act_out(act, (v[e] - mean[e]) * rstd[e] * ga[e] + be[e]);
      if (p > 0.f) {
        float d[4];
        dropout_scale4(seed, (unsigned long long)(r * C + q * 4), thresh, inv_keep, d);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] *= d[e];
      }
      if (R) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += Elem<T>::to_f(R[r * C + q * 4 + e]);
      }
      const long long oo = map_row(om, r, C) + q * 4;
      if (OUTF32) {
        f32x4 w = {o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + oo) = w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) reinterpret_cast<T*>(y)[oo + e] = Elem<T>::from_f(o[e]);
      }
    }
  }
  // zero halo: `halo` rows in front of and behind every batch block of the mapped layout (om.off = halo * C)
  if (om.L && halo_rows_per_batch > 0 && !OUTF32) {
    const long long nb = rows / om.L, per = 2 * halo_rows_per_batch * C;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < nb * per; i += (long long)gridDim.x * 256) {
      const long long b = i / per, j = i % per;
      const long long e = j < halo_rows_per_batch * C ? j : om.off + om.L * C + (j - halo_rows_per_batch * C);
      reinterpret_cast<T*>(y)[b * om.bstride + e] = Elem<T>::from_f(0.f);
    }
  }
}

// g = dropmask * dy * act'(y),  y recomputed from x;  partial[blk][c] = (sum g, sum g * xhat)
__device__ __forceinline__ void bn_g4(const f32x4& xv, const f32x4& dyv, const f32x4& mean, const f32x4& rstd, const f32x4& ga, const f32x4& be,
                                      int act, float p, unsigned long long seed, unsigned long long idx, unsigned int thresh, float inv_keep,
                                      float (&gout)[4], float (&xh)[4]) {
  float d[4] = {1.f, 1.f, 1.f, 1.f};
  if (p > 0.f) dropout_scale4(seed, idx, thresh, inv_keep, d);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    xh[e] = (xv[e] - mean[e]) * rstd[e];
    float gg = dyv[e] * d[e];
    if (act == ACT_TANH) { const float yv = tanhf(xh[e] * ga[e] + be[e]); gg *= 1.f - yv * yv; }
    gout[e] = gg;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ stats,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, double* __restrict__ partial,
                                                             long long rows, int C, int act, float p, unsigned long long seed) {
  __shared__ double red[256][8];
  const BnGeom g = geom(C);
  const int tid = threadIdx.x, q = tid % g.cq, ro = tid / g.cq;
  const unsigned int thresh = p > 0.f ? dropout_thresh(p) : 0u;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (ro < g.rg) {
    const f32x4 mean = *reinterpret_cast<const f32x4*>(stats + q * 4), rstd = *reinterpret_cast<const f32x4*>(stats + C + q * 4);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + q * 4), be = *reinterpret_cast<const f32x4*>(beta + q * 4);
    for (long long r = (long long)blockIdx.x * g.rg + ro; r < rows; r += (long long)gridDim.x * g.rg) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * C + q * 4);
      const f32x4 dv = *reinterpret_cast<const f32x4*>(dy + r * C + q * 4);
      float gg[4], xh[4];
      bn_g4(xv, dv, mean, rstd, ga, be, act, p, seed, (unsigned long long)(r * C + q * 4), thresh, inv_keep, gg, xh);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[e] += (double)gg[e]; s2[e] += (double)gg[e] * (double)xh[e]; }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[tid][e] = s1[e]; red[tid][4 + e] = s2[e]; }
  __syncthreads();
  if (tid < g.cq) {
    for (int o = 1; o < g.rg; ++o)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[tid][e] += red[tid + o * g.cq][e];
    double* dst = partial + ((long long)blockIdx.x * C + tid * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { dst[2 * e] = red[tid][e]; dst[2 * e + 1] = red[tid][4 + e]; }
  }
}

// sums[c] = sum g / rows, sums[C + c] = sum g xhat / rows;  dbeta += sum g, dgamma += sum g xhat
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ partial, int nblk, long long rows, int C, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ sums) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0, s2 = 0;
  for (int b = 0; b < nblk; ++b) { s1 += partial[((long long)b * C + c) * 2]; s2 += partial[((long long)b * C + c) * 2 + 1]; }
  sums[c] = (float)(s1 / (double)rows);
  sums[C + c] = (float)(s2 / (double)rows);
  if (dbeta) dbeta[c] += (float)s1;
  if (dgamma) dgamma[c] += (float)s2;
}

// dx = gamma * rstd * (g - mean(g) - xhat * mean(g xhat))   (eval statistics: dx = gamma * rstd * g)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ stats,
                                                           const float* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           T* __restrict__ dx, long long rows, int C, int act, float p, unsigned long long seed,
                                                           int training, RowMap om, long long halo_rows_per_batch) {
  const BnGeom g = geom(C);
  const int tid = threadIdx.x, q = tid % g.cq, ro = tid / g.cq;
  const unsigned int thresh = p > 0.f ? dropout_thresh(p) : 0u;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  if (ro < g.rg) {
    const f32x4 mean = *reinterpret_cast<const f32x4*>(stats + q * 4), rstd = *reinterpret_cast<const f32x4*>(stats + C + q * 4);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + q * 4), be = *reinterpret_cast<const f32x4*>(beta + q * 4);
    f32x4 m1 = {0.f, 0.f, 0.f, 0.f}, m2 = {0.f, 0.f, 0.f, 0.f};
    if (training) { m1 = *reinterpret_cast<const f32x4*>(sums + q * 4); m2 = *reinterpret_cast<const f32x4*>(sums + C + q * 4); }
    for (long long r = (long long)blockIdx.x * g.rg + ro; r < rows; r += (long long)gridDim.x * g.rg) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * C + q * 4);
      const f32x4 dv = *reinterpret_cast<const f32x4*>(dy + r * C + q * 4);
      float gg[4], xh[4];
      bn_g4(xv, dv, mean, rstd, ga, be, act, p, seed, (unsigned long long)(r * C + q * 4), thresh, inv_keep, gg, xh);
      const long long oo = map_row(om, r, C) + q * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) dx[oo + e] = Elem<T>::from_f(ga[e] * rstd[e] * (gg[e] - m1[e] - xh[e] * m2[e]));
    }
  }
  if (om.L && halo_rows_per_batch > 0) {
    const long long nb = rows / om.L, per = 2 * halo_rows_per_batch * C;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < nb * per; i += (long long)gridDim.x * 256) {
      const long long b = i / per, j = i % per;
      const long long e = j < halo_rows_per_batch * C ? j : om.off + om.L * C + (j - halo_rows_per_batch * C);
      dx[b * om.bstride + e] = Elem<T>::from_f(0.f);
    }
  }
}

int nblk_for(long long rows, int C) {
  const int rg = 256 / (C / 4);
  long long b = (rows + (long long)rg * 16 - 1) / ((long long)rg * 16);
  return (int)(b < 1 ? 1 : (b > BN_MAXBLK ? BN_MAXBLK : b));
}
int apply_blocks(long long rows, int C) {
  const int rg = 256 / (C / 4);
  long long b = (rows + (long long)rg * 4 - 1) / ((long long)rg * 4);
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
bool bn_shape_ok(long long rows, int C) { return rows > 0 && C >= 4 && C % 4 == 0 && C <= 1024; }

}  // namespace

extern "C" int64_t st5_batchnorm_ws_bytes(int32_t C) { return (int64_t)BN_MAXBLK * C * 2 * sizeof(double) + (int64_t)2 * C * sizeof(float); }

extern "C" int st5_batchnorm_act_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                     int64_t* num_batches_tracked, float momentum, float eps, int32_t training, int32_t act,
                                     float dropout_p, uint64_t seed, const void* residual, void* y, int32_t y_f32, float* stats,
                                     void* ws, int64_t rows, int32_t C, int64_t out_L, int64_t out_bstride, int64_t out_off, int32_t halo,
                                     int dtype, void* stream) {
  if (!x || !gamma || !beta || !y || !stats || !ws || !bn_shape_ok(rows, C)) return ST5_ERR_ARG;
  if (act != ST5_ACT_NONE && act != ST5_ACT_TANH) return ST5_ERR_ARG;
  if (!training && (!running_mean || !running_var)) return ST5_ERR_ARG;
  if (out_L && (rows % out_L)) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  double* partial = reinterpret_cast<double*>(ws);
  const int nblk = nblk_for(rows, C);
  if (training) hipLaunchKernelGGL(bn_fwd_partial_kernel, dim3(nblk), dim3(256), 0, s, x, partial, (long long)rows, C);
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, partial, nblk, (long long)rows, C, eps, momentum, training,
                     running_mean, running_var, (long long*)num_batches_tracked, stats);
  RowMap om; om.L = out_L; om.bstride = out_bstride; om.off = out_off;
  const dim3 grid(apply_blocks(rows, C));
#define BN_APPLY(T, F32) hipLaunchKernelGGL((bn_fwd_apply_kernel<T, F32>), grid, dim3(256), 0, s, x, stats, gamma, beta, (const T*)residual, y, \
                                            (long long)rows, C, act, dropout_p, (unsigned long long)seed, om, (long long)halo)
  if (dtype == ST5_BF16) { if (y_f32) BN_APPLY(bf16_t, true); else BN_APPLY(bf16_t, false); }
  else if (dtype == ST5_F32) { if (y_f32) BN_APPLY(float, true); else BN_APPLY(float, false); }
  else return ST5_ERR_ARG;
#undef BN_APPLY
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_batchnorm_act_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta, float* dgamma,
                                     float* dbeta, int32_t training, int32_t act, float dropout_p, uint64_t seed, void* dx, void* ws,
                                     int64_t rows, int32_t C, int64_t out_L, int64_t out_bstride, int64_t out_off, int32_t halo, int dtype,
                                     void* stream) {
  if (!x || !dy || !stats || !gamma || !beta || !dx || !ws || !bn_shape_ok(rows, C)) return ST5_ERR_ARG;
  if (act != ST5_ACT_NONE && act != ST5_ACT_TANH) return ST5_ERR_ARG;
  if (out_L && (rows % out_L)) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  double* partial = reinterpret_cast<double*>(ws);
  float* sums = reinterpret_cast<float*>(partial + (size_t)BN_MAXBLK * C * 2);
  const int nblk = nblk_for(rows, C);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nblk), dim3(256), 0, s, x, dy, stats, gamma, beta, partial, (long long)rows, C, act, dropout_p,
                     (unsigned long long)seed);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, partial, nblk, (long long)rows, C, dgamma, dbeta, sums);
  RowMap om; om.L = out_L; om.bstride = out_bstride; om.off = out_off;
  const dim3 grid(apply_blocks(rows, C));
  if (dtype == ST5_BF16)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t>), grid, dim3(256), 0, s, x, dy, stats, sums, gamma, beta, (bf16_t*)dx, (long long)rows, C, act,
                       dropout_p, (unsigned long long)seed, training, om, (long long)halo);
  else if (dtype == ST5_F32)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), grid, dim3(256), 0, s, x, dy, stats, sums, gamma, beta, (float*)dx, (long long)rows, C, act,
                       dropout_p, (unsigned long long)seed, training, om, (long long)halo);
  else return ST5_ERR_ARG;
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
