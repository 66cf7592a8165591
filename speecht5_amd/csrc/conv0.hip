// Speech pre-net layer 0: Conv1d(1 -> C, k, stride, no bias) + GroupNorm(C groups, i.e. per
// (clip, channel) statistics over time) + GELU, output channels-last [B, L, C].
// Reference: SpeechT5/speecht5/models/modules/speech_encoder_prenet.py:300,323-324 (block 0 of
// ConvFeatureExtractionModel, mode "default").
//
// HBM-bound: the waveform is 0.64 MB/clip while the output is 32.8 MB/clip (bf16), so the
// convolution is recomputed instead of stored: forward = stats pass (reads wav only) + one fused
// conv+normalise+GELU pass that writes the output once.  Backward recomputes conv/x_hat from the
// waveform as well and reads dY twice (statistics of the GroupNorm backward, then dW), so nothing
// but (mean, rstd) [B,C] is saved for backward.
#include "common.h"
#include "../../include/speecht5_hip.h"

namespace {

constexpr int TCH = 256;   // output time steps per block
constexpr int MAXK = 16;   // max kernel width held in registers

__host__ __device__ inline int out_len(int S, int k, int stride) { return S < k ? 0 : (S - k) / stride + 1; }

// Each thread owns 8 consecutive channels (one 16-byte channels-last store); 256 threads =
// (C/8 channel groups) x (256*8/C time lanes).  Requires C % 8 == 0 and C <= 2048.
struct Geo { int cg, tl; };
__device__ __forceinline__ Geo geo(int C) { Geo g; g.cg = C / 8; g.tl = 256 / g.cg; return g; }


// weights of this thread's 8 channels in registers; taps >= k are zero
template <int KW>
__device__ __forceinline__ void load_w8(const float* __restrict__ w, int c0, int k, float (&wr)[8][KW]) {
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j < KW; ++j) wr[e][j] = j < k ? w[(c0 + e) * k + j] : 0.f;
}
template <int KW>
__device__ __forceinline__ void conv8(const float* __restrict__ segp, const float (&wr)[8][KW], float (&y)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) y[e] = 0.f;
#pragma unroll
  for (int j = 0; j < KW; ++j) {
    const float xv = segp[j];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = fmaf(wr[e][j], xv, y[e]);
  }
}

// ---- forward stats: partial sums of y and y^2 per (b, chunk, c) ----
__global__ __launch_bounds__(256) void conv0_stats_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                          float* __restrict__ part, int S, int L, int C, int k,
                                                          int stride, int nch) {
  extern __shared__ float seg[];  // TCH*stride + k waveform samples
  const int b = blockIdx.y, ch = blockIdx.x, t0 = ch * TCH;
  const int nt = min(TCH, L - t0);
  const int nseg = (nt - 1) * stride + k;
  for (int i = threadIdx.x; i < nseg + MAXK; i += 256)
    seg[i] = i < nseg ? wav[(long long)b * S + (long long)t0 * stride + i] : 0.f;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float wr[MAXK];
#pragma unroll
    for (int j = 0; j < MAXK; ++j) wr[j] = j < k ? w[c * k + j] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    for (int t = 0; t < nt; ++t) {
      float y = 0.f;
#pragma unroll
      for (int j = 0; j < MAXK; ++j) if (j < k) y = fmaf(wr[j], seg[t * stride + j], y);
      s1 += y; s2 = fmaf(y, y, s2);
    }
    float* o = part + (((long long)b * nch + ch) * C + c) * 2;
    o[0] = s1; o[1] = s2;
  }
}
// part: [B][nch][C][2] chunk partials -> (mean, rstd) per (b, c).  Block = 32 channels x 8 chunk lanes.
__global__ __launch_bounds__(256) void conv0_stats_final_kernel(const float* __restrict__ part, float* __restrict__ stats,
                                                                int C, int nch, int L, float eps) {
  __shared__ double red[8][32][2];
  const int b = blockIdx.y, cl = threadIdx.x & 31, j = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int ch = j; ch < nch; ch += 8) {
      const float* o = part + (((long long)b * nch + ch) * C + c) * 2;
      s1 += (double)o[0]; s2 += (double)o[1];
    }
  red[j][cl][0] = s1; red[j][cl][1] = s2;
  __syncthreads();
  if (j == 0 && c < C) {
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { s1 += red[q][cl][0]; s2 += red[q][cl][1]; }
    const double mu = s1 / L;
    double var = s2 / L - mu * mu;
    if (var < 0.0) var = 0.0;
    stats[((long long)b * C + c) * 2 + 0] = (float)mu;
    stats[((long long)b * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// ---- forward apply: conv -> normalise -> affine -> GELU -> channels-last store ----
template <typename T, int KW>
__global__ __launch_bounds__(256) void conv0_apply_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ stats, T* __restrict__ out, int S,
                                                          int L, int C, int k, int stride) {
  extern __shared__ float seg[];
  const int b = blockIdx.y, t0 = blockIdx.x * TCH;
  const int nt = min(TCH, L - t0);
  const int nseg = (nt - 1) * stride + k;
  for (int i = threadIdx.x; i < nseg + MAXK; i += 256)
    seg[i] = i < nseg ? wav[(long long)b * S + (long long)t0 * stride + i] : 0.f;
  const Geo g = geo(C);
  const int cgi = threadIdx.x % g.cg, tli = threadIdx.x / g.cg;
  const int c0 = cgi * 8;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float mu = stats[((long long)b * C + c0 + e) * 2], rs = stats[((long long)b * C + c0 + e) * 2 + 1];
    sc[e] = rs * gamma[c0 + e];
    sh[e] = beta[c0 + e] - mu * sc[e];
  }
  float wr[8][KW];
  load_w8<KW>(w, c0, k, wr);
  __syncthreads();
  if (tli >= g.tl) return;
  for (int t = tli; t < nt; t += g.tl) {
    float y[8];
    conv8<KW>(seg + t * stride, wr, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = gelu_f(fmaf(y[e], sc[e], sh[e]));
    store8f<T>(out + ((long long)b * L + t0 + t) * C + c0, y);
  }
}

// ---- backward pass A: S1 = sum_t dz, S2 = sum_t dz * x_hat per (b, chunk, c) ----
template <typename T, int KW>
__global__ __launch_bounds__(256) void conv0_bwd_stats_kernel(const float* __restrict__ wav,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ stats,
                                                              const T* __restrict__ dY, float* __restrict__ part,
                                                              int S, int L, int C, int k, int stride, int nch) {
  extern __shared__ float seg[];
  float* red = seg + (TCH * stride + k + MAXK + 3) / 4 * 4;  // [tl][C][2]
  const int b = blockIdx.y, ch = blockIdx.x, t0 = ch * TCH;
  const int nt = min(TCH, L - t0);
  const int nseg = (nt - 1) * stride + k;
  for (int i = threadIdx.x; i < nseg + MAXK; i += 256)
    seg[i] = i < nseg ? wav[(long long)b * S + (long long)t0 * stride + i] : 0.f;
  const Geo g = geo(C);
  const int cgi = threadIdx.x % g.cg, tli = threadIdx.x / g.cg;
  const int c0 = cgi * 8;
  float mu[8], rs[8], ga[8], be[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = stats[((long long)b * C + c0 + e) * 2]; rs[e] = stats[((long long)b * C + c0 + e) * 2 + 1];
    ga[e] = gamma[c0 + e]; be[e] = beta[c0 + e]; s1[e] = 0.f; s2[e] = 0.f;
  }
  float wr[8][KW];
  load_w8<KW>(w, c0, k, wr);
  __syncthreads();
  if (tli < g.tl) {
    for (int t = tli; t < nt; t += g.tl) {
      float y[8], dy[8];
      conv8<KW>(seg + t * stride, wr, y);
      load8f<T>(dY + ((long long)b * L + t0 + t) * C + c0, dy);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (y[e] - mu[e]) * rs[e];
        const float dz = dy[e] * gelu_fast<true>(fmaf(xh, ga[e], be[e]));
        s1[e] += dz; s2[e] = fmaf(dz, xh, s2[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[((long long)tli * C + c0 + e) * 2] = s1[e];
      red[((long long)tli * C + c0 + e) * 2 + 1] = s2[e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, bq = 0.f;
    for (int l = 0; l < g.tl; ++l) { a += red[((long long)l * C + c) * 2]; bq += red[((long long)l * C + c) * 2 + 1]; }
    float* o = part + (((long long)b * nch + ch) * C + c) * 2;
    o[0] = a; o[1] = bq;
  }
}
// sums[b,c] = (S1, S2); dgamma[c] += gscale * sum_b S2; dbeta[c] += gscale * sum_b S1  (grid (C/32, B); the B blocks of a
// channel combine with fp32 atomics)
__global__ __launch_bounds__(256) void conv0_bwd_stats_final_kernel(const float* __restrict__ part, float* __restrict__ sums,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                    int B, int C, int nch, float gscale) {
  __shared__ double red[8][32][2];
  const int b = blockIdx.y, cl = threadIdx.x & 31, j = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int ch = j; ch < nch; ch += 8) {
      const float* o = part + (((long long)b * nch + ch) * C + c) * 2;
      s1 += (double)o[0]; s2 += (double)o[1];
    }
  red[j][cl][0] = s1; red[j][cl][1] = s2;
  __syncthreads();
  if (j == 0 && c < C) {
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { s1 += red[q][cl][0]; s2 += red[q][cl][1]; }
    sums[((long long)b * C + c) * 2] = (float)s1;
    sums[((long long)b * C + c) * 2 + 1] = (float)s2;
    if (dgamma) unsafeAtomicAdd(dgamma + c, gscale * (float)s2);
    if (dbeta) unsafeAtomicAdd(dbeta + c, gscale * (float)s1);
  }
}

// ---- backward pass B: dconv = rstd*gamma*(dz - S1/L - x_hat*S2/L); dw[c,j] = sum dconv * wav ----
// A thread owns TWO channels (their 2 x KW weights and 2 x KW gradient accumulators live in registers) and walks the
// block's time steps sequentially; the waveform window comes from LDS as broadcast reads, dY as 4-byte (bf16 pair)
// loads that are contiguous across the wave.  Every (c, j) partial has one owner: plain stores, no atomics.
template <typename T> __device__ __forceinline__ void load2f(const T* p, float& a, float& b);
template <> __device__ __forceinline__ void load2f<float>(const float* p, float& a, float& b) {
  const float2 v = *reinterpret_cast<const float2*>(p); a = v.x; b = v.y;
}
template <> __device__ __forceinline__ void load2f<bf16_t>(const bf16_t* p, float& a, float& b) {
  const unsigned int v = *reinterpret_cast<const unsigned int*>(p);
  a = __uint_as_float(v << 16); b = __uint_as_float(v & 0xffff0000u);
}

template <typename T, int KW>
__global__ __launch_bounds__(256) void conv0_bwd_dw_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ sums, const T* __restrict__ dY,
                                                           float* __restrict__ part, int S, int L, int C, int k,
                                                           int stride, int nch) {
  extern __shared__ float seg[];
  const int b = blockIdx.y, ch = blockIdx.x, t0 = ch * TCH;
  const int nt = min(TCH, L - t0);
  const int nseg = (nt - 1) * stride + k;
  for (int i = threadIdx.x; i < nseg + MAXK; i += 256)
    seg[i] = i < nseg ? wav[(long long)b * S + (long long)t0 * stride + i] : 0.f;
  __syncthreads();
  const float invL = 1.f / (float)L;
  for (int cp = threadIdx.x; cp < C / 2; cp += 256) {
    const int c0 = cp * 2;
    float mu[2], rs[2], ga[2], be[2], m1[2], m2[2], wr[2][KW], dw[2][KW];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      mu[e] = stats[((long long)b * C + c0 + e) * 2]; rs[e] = stats[((long long)b * C + c0 + e) * 2 + 1];
      ga[e] = gamma[c0 + e]; be[e] = beta[c0 + e];
      m1[e] = sums[((long long)b * C + c0 + e) * 2] * invL; m2[e] = sums[((long long)b * C + c0 + e) * 2 + 1] * invL;
#pragma unroll
      for (int j = 0; j < KW; ++j) { wr[e][j] = j < k ? w[(c0 + e) * k + j] : 0.f; dw[e][j] = 0.f; }
    }
    const T* dyp = dY + ((long long)b * L + t0) * C + c0;
    float d0, d1;
    load2f<T>(dyp, d0, d1);
    for (int t = 0; t < nt; ++t) {
      float n0 = 0.f, n1 = 0.f;
      if (t + 1 < nt) load2f<T>(dyp + (long long)(t + 1) * C, n0, n1);   // next step's dY in flight during the math
      float xv[KW];
#pragma unroll
      for (int j = 0; j < KW; ++j) xv[j] = seg[t * stride + j];
      float y0 = 0.f, y1 = 0.f;
#pragma unroll
      for (int j = 0; j < KW; ++j) { y0 = fmaf(wr[0][j], xv[j], y0); y1 = fmaf(wr[1][j], xv[j], y1); }
      const float xh0 = (y0 - mu[0]) * rs[0], xh1 = (y1 - mu[1]) * rs[1];
      const float dz0 = d0 * gelu_fast<true>(fmaf(xh0, ga[0], be[0]));
      const float dz1 = d1 * gelu_fast<true>(fmaf(xh1, ga[1], be[1]));
      const float dc0 = rs[0] * ga[0] * (dz0 - m1[0] - xh0 * m2[0]);
      const float dc1 = rs[1] * ga[1] * (dz1 - m1[1] - xh1 * m2[1]);
#pragma unroll
      for (int j = 0; j < KW; ++j) { dw[0][j] = fmaf(dc0, xv[j], dw[0][j]); dw[1][j] = fmaf(dc1, xv[j], dw[1][j]); }
      d0 = n0; d1 = n1;
    }
    float* o = part + ((long long)b * nch + ch) * C * k;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int j = 0; j < KW; ++j) if (j < k) o[(c0 + e) * k + j] = dw[e][j];
  }
}
// dw[i] += gscale * sum_p part[p][i]   (block = 32 outputs x 8 part lanes)
__global__ __launch_bounds__(256) void conv0_bwd_dw_final_kernel(const float* __restrict__ part, float* __restrict__ dw, int n,
                                                                 int nparts, float gscale) {
  __shared__ double red[8][32];
  const int il = threadIdx.x & 31, j = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + il;
  double s = 0.0;
  if (i < n)
    for (int p = j; p < nparts; p += 8) s += (double)part[(long long)p * n + i];
  red[j][il] = s;
  __syncthreads();
  if (j == 0 && i < n) {
    s = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][il];
    dw[i] += gscale * (float)s;
  }
}

}  // namespace

extern "C" int64_t st5_conv0_ws_bytes(int32_t B, int32_t S, int32_t C, int32_t k, int32_t stride) {
  const int L = out_len(S, k, stride);
  const int64_t nch = (L + TCH - 1) / TCH;
  const int64_t a = (int64_t)B * nch * C * 2;        // stats partials
  const int64_t b = (int64_t)B * nch * C * k;        // dw partials
  const int64_t c = (int64_t)B * C * 2;              // bwd sums
  return ((a > b ? a : b) + c) * (int64_t)sizeof(float);
}

extern "C" int st5_conv0_gn_gelu_fwd(const float* wav, const float* w, const float* gamma, const float* beta,
                                     void* out, float* stats, void* ws, int32_t B, int32_t S, int32_t C, int32_t k,
                                     int32_t stride, float eps, int dtype, void* stream) {
  if (!wav || !w || !gamma || !beta || !out || !stats || !ws) return ST5_ERR_ARG;
  if (C % 8 || C > 2048 || 256 % (C / 8 > 256 ? 256 : C / 8) || k > MAXK || k < 1 || stride < 1) return ST5_ERR_ARG;
  if (C / 8 > 256) return ST5_ERR_ARG;
  const int L = out_len(S, k, stride);
  if (L <= 0 || B <= 0) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nch = (L + TCH - 1) / TCH;
  const size_t shm = (size_t)(TCH * stride + k + MAXK) * sizeof(float);
  float* part = (float*)ws;
  hipLaunchKernelGGL(conv0_stats_kernel, dim3(nch, B), dim3(256), shm, s, wav, w, part, S, L, C, k, stride, nch);
  hipLaunchKernelGGL(conv0_stats_final_kernel, dim3((C + 31) / 32, B), dim3(256), 0, s, part, stats, C, nch, L, eps);
#define APPLY(TT, KW)                                                                                           \
  hipLaunchKernelGGL((conv0_apply_kernel<TT, KW>), dim3(nch, B), dim3(256), shm, s, wav, w, gamma, beta, stats, \
                     (TT*)out, S, L, C, k, stride)
  if (dtype == ST5_BF16) { if (k <= 10) APPLY(bf16_t, 10); else APPLY(bf16_t, MAXK); }
  else if (dtype == ST5_F32) { if (k <= 10) APPLY(float, 10); else APPLY(float, MAXK); }
  else return ST5_ERR_ARG;
#undef APPLY
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_conv0_gn_gelu_bwd(const float* wav, const float* w, const float* gamma, const float* beta,
                                     const float* stats, const void* dY, float* dw, float* dgamma, float* dbeta,
                                     void* ws, int32_t B, int32_t S, int32_t C, int32_t k, int32_t stride,
                                     float gscale, int dtype, void* stream) {
  if (!wav || !w || !gamma || !beta || !stats || !dY || !ws) return ST5_ERR_ARG;
  if (C % 8 || C / 8 > 256 || 256 % (C / 8) || k > MAXK || k < 1 || stride < 1) return ST5_ERR_ARG;
  const int L = out_len(S, k, stride);
  if (L <= 0 || B <= 0) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nch = (L + TCH - 1) / TCH;
  const int tl = 256 / (C / 8);
  const size_t segf = (size_t)((TCH * stride + k + MAXK + 3) / 4 * 4);
  const int64_t a = (int64_t)B * nch * C * 2, b = (int64_t)B * nch * C * k;
  float* part = (float*)ws;
  float* sums = part + (a > b ? a : b);
  const size_t shmA = (segf + (size_t)tl * C * 2) * sizeof(float);
  const size_t shmB = segf * sizeof(float);
#define BSTATS(TT, KW)                                                                                         \
  hipLaunchKernelGGL((conv0_bwd_stats_kernel<TT, KW>), dim3(nch, B), dim3(256), shmA, s, wav, w, gamma, beta,  \
                     stats, (const TT*)dY, part, S, L, C, k, stride, nch)
  if (dtype == ST5_BF16) { if (k <= 10) BSTATS(bf16_t, 10); else BSTATS(bf16_t, MAXK); }
  else if (dtype == ST5_F32) { if (k <= 10) BSTATS(float, 10); else BSTATS(float, MAXK); }
  else return ST5_ERR_ARG;
#undef BSTATS
  hipLaunchKernelGGL(conv0_bwd_stats_final_kernel, dim3((C + 31) / 32, B), dim3(256), 0, s, part, sums, dgamma, dbeta,
                     B, C, nch, gscale);
  if (dw) {
#define BDW(TT, KW)                                                                                          \
  hipLaunchKernelGGL((conv0_bwd_dw_kernel<TT, KW>), dim3(nch, B), dim3(256), shmB, s, wav, w, gamma, beta,   \
                     stats, sums, (const TT*)dY, part, S, L, C, k, stride, nch)
    if (dtype == ST5_BF16) { if (k <= 10) BDW(bf16_t, 10); else BDW(bf16_t, MAXK); }
    else { if (k <= 10) BDW(float, 10); else BDW(float, MAXK); }
#undef BDW
    hipLaunchKernelGGL(conv0_bwd_dw_final_kernel, dim3((C * k + 31) / 32), dim3(256), 0, s, part, dw, C * k,
                       B * nch, gscale);
  }
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
