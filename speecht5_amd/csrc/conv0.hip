// Speech pre-net layer 0: Conv1d(1 -> C, k, stride, no bias) + GroupNorm(C groups, i.e. per
// (clip, channel) statistics over time) + GELU, output channels-last [B, L, C].
// Reference: SpeechT5/speecht5/models/modules/speech_encoder_prenet.py:300,323-324 (block 0 of
// ConvFeatureExtractionModel, mode "default").
//
// The waveform is 0.64 MB/clip while the output is 32.8 MB/clip (bf16), so the convolution is never stored: it is
// recomputed from the waveform wherever it is needed, and nothing but (mean, rstd) [B, C] is saved for backward.
//
// The GroupNorm statistics of y_t = sum_j w_j x[s t + j] need no pass over y at all.  With the waveform moments
//     M_j  = sum_t x[s t + j]                    (k numbers per clip)
//     R_jj'= sum_t x[s t + j] x[s t + j']        (k (k+1)/2 numbers per clip)
// one gets  sum_t y = w.M  and  sum_t y^2 = w^T R w  for every channel, so
//   forward  = moments of the waveform (reads 0.64 MB/clip) + ONE fused conv + normalise + GELU pass that writes
//              the output once;
//   backward = ONE pass over dY accumulating S1 = sum dz, S2 = sum dz x_hat and A_j = sum dz x[s t + j]
//              (dz = dy gelu'(.)); the GroupNorm backward  dconv = rstd gamma (dz - S1/L - x_hat S2/L)  is folded in
//              afterwards per (clip, channel):  dw_j = rstd gamma [A_j - S1 M_j / L - S2 rstd (R w - mean M)_j / L].
// (The previous version read dY twice and ran a statistics pass over the recomputed convolution in both directions.)
#include "common.h"
#include "../../include/speecht5_hip.h"

namespace {

constexpr int TCH = 256;   // output time steps per block
constexpr int MAXK = 16;   // max kernel width held in registers
constexpr int MAXMOM = MAXK + MAXK * (MAXK + 1) / 2;

__host__ __device__ inline int out_len(int S, int k, int stride) { return S < k ? 0 : (S - k) / stride + 1; }
__host__ __device__ inline int nmom(int k) { return k + k * (k + 1) / 2; }
// index of R_{j,j'} (j <= j') inside the moment vector (after the k first-order moments)
__host__ __device__ inline int ridx(int k, int j, int jp) { return k + j * k - j * (j - 1) / 2 + (jp - j); }

// Each thread of the apply kernel owns 8 consecutive channels (one 16-byte channels-last store); 256 threads =
// (C/8 channel groups) x (256*8/C time lanes).  Requires C % 8 == 0 and C <= 2048.
struct Geo { int cg, tl; };
__device__ __forceinline__ Geo geo(int C) { Geo g; g.cg = C / 8; g.tl = 256 / g.cg; return g; }

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

__device__ __forceinline__ void stage_wav(float* seg, const float* __restrict__ wav, int b, int S, int t0, int nt, int k,
                                          int stride) {
  const int nseg = (nt - 1) * stride + k;
  for (int i = threadIdx.x; i < nseg + MAXK; i += 256)
    seg[i] = i < nseg ? wav[(long long)b * S + (long long)t0 * stride + i] : 0.f;
}

// ---- waveform moments: part[b][chunk][nmom] (one time step per thread, wave shuffle + LDS reduction) ----
template <int KW>
__global__ __launch_bounds__(256) void conv0_moments_kernel(const float* __restrict__ wav, float* __restrict__ part, int S,
                                                            int L, int k, int stride, int nch) {
  extern __shared__ float seg[];
  // one slot per wave and moment, summed in a fixed order below: LDS atomics here would make the GroupNorm statistics --
  // and through bf16 rounding every activation after them -- depend on wave arrival order (run-to-run differences)
  __shared__ float red[4][MAXMOM];
  const int b = blockIdx.y, ch = blockIdx.x, t0 = ch * TCH;
  const int nt = min(TCH, L - t0);
  stage_wav(seg, wav, b, S, t0, nt, k, stride);
  const int nm = nmom(k);
  __syncthreads();
  const int t = threadIdx.x;
  const int wv = threadIdx.x >> 6;
  float x[KW];
#pragma unroll
  for (int j = 0; j < KW; ++j) x[j] = (t < nt && j < k) ? seg[t * stride + j] : 0.f;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < KW; ++j) {
    if (j < k) {
      const float s = wave_sum(x[j]);
      if (lane == 0) red[wv][j] = s;
#pragma unroll
      for (int jp = j; jp < KW; ++jp) {
        if (jp < k) {
          const float r = wave_sum(x[j] * x[jp]);
          if (lane == 0) red[wv][ridx(k, j, jp)] = r;
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nm; i += 256)
    part[((long long)b * nch + ch) * nm + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}
// mom[b][nmom] (double) = sum over chunks
__global__ __launch_bounds__(256) void conv0_moments_final_kernel(const float* __restrict__ part, double* __restrict__ mom,
                                                                  int nm, int nch) {
  __shared__ double red[4][MAXMOM];
  const int b = blockIdx.x;
  const int i = threadIdx.x & 63, j = threadIdx.x >> 6;   // nm <= 152: three passes of 64
  for (int base = 0; base < nm; base += 64) {
    const int m = base + i;
    double s = 0.0;
    if (m < nm)
      for (int ch = j; ch < nch; ch += 4) s += (double)part[((long long)b * nch + ch) * nm + m];
    if (m < nm) red[j][m] = s;
  }
  __syncthreads();
  for (int m = threadIdx.x; m < nm; m += 256) mom[(long long)b * nm + m] = red[0][m] + red[1][m] + red[2][m] + red[3][m];
}
// (mean, rstd) per (b, c) from the moments:  mean = w.M / L,  E[y^2] = w^T R w / L
__global__ __launch_bounds__(256) void conv0_stats_from_moments_kernel(const double* __restrict__ mom, const float* __restrict__ w,
                                                                       float* __restrict__ stats, int C, int k, int L, float eps) {
  const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double* mb = mom + (long long)b * nmom(k);
  double s1 = 0.0, s2 = 0.0;
  for (int j = 0; j < k; ++j) {
    const double wj = (double)w[c * k + j];
    s1 += wj * mb[j];
    for (int jp = j; jp < k; ++jp) {
      const double t = wj * (double)w[c * k + jp] * mb[ridx(k, j, jp)];
      s2 += (jp == j) ? t : 2.0 * t;
    }
  }
  const double mu = s1 / L;
  double var = s2 / L - mu * mu;
  if (var < 0.0) var = 0.0;
  stats[((long long)b * C + c) * 2 + 0] = (float)mu;
  stats[((long long)b * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
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
  stage_wav(seg, wav, b, S, t0, nt, k, stride);
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
  constexpr bool FAST = sizeof(T) == 2;   // bf16 output: fast erf (|error| 1.5e-7), fp32 parity mode: libm erff
  for (int t = tli; t < nt; t += g.tl) {
    float y[8];
    conv8<KW>(seg + t * stride, wr, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float z = fmaf(y[e], sc[e], sh[e]);
      y[e] = FAST ? gelu_fast<false>(z) : gelu_f(z);
    }
    store8f<T>(out + ((long long)b * L + t0 + t) * C + c0, y);
  }
}

// ---- backward: ONE pass over dY.  part[b][chunk][c][KW + 2] = (A_0..A_{k-1}, S1, S2) ----
// A thread owns TWO channels (their 2 x KW taps and 2 x (KW + 2) accumulators live in registers) and walks the block's
// time steps sequentially; the waveform window comes from LDS as broadcast reads, dY as 4-byte (bf16 pair) loads that
// are contiguous across the wave.  Every partial has one owner: plain stores, no atomics.
template <typename T> __device__ __forceinline__ void load2f(const T* p, float& a, float& b);
template <> __device__ __forceinline__ void load2f<float>(const float* p, float& a, float& b) {
  const float2 v = *reinterpret_cast<const float2*>(p); a = v.x; b = v.y;
}
template <> __device__ __forceinline__ void load2f<bf16_t>(const bf16_t* p, float& a, float& b) {
  const unsigned int v = *reinterpret_cast<const unsigned int*>(p);
  a = __uint_as_float(v << 16); b = __uint_as_float(v & 0xffff0000u);
}

template <typename T, int KW>
__global__ __launch_bounds__(256) void conv0_bwd_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ stats, const T* __restrict__ dY,
                                                        float* __restrict__ part, int S, int L, int C, int k, int stride,
                                                        int nch) {
  extern __shared__ float seg[];
  const int b = blockIdx.y, ch = blockIdx.x, t0 = ch * TCH;
  const int nt = min(TCH, L - t0);
  stage_wav(seg, wav, b, S, t0, nt, k, stride);
  __syncthreads();
  constexpr bool FAST = sizeof(T) == 2;
  for (int cp = threadIdx.x; cp < C / 2; cp += 256) {
    const int c0 = cp * 2;
    float mu[2], rs[2], ga[2], be[2], wr[2][KW], A[2][KW], s1[2], s2[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      mu[e] = stats[((long long)b * C + c0 + e) * 2]; rs[e] = stats[((long long)b * C + c0 + e) * 2 + 1];
      ga[e] = gamma[c0 + e]; be[e] = beta[c0 + e]; s1[e] = 0.f; s2[e] = 0.f;
#pragma unroll
      for (int j = 0; j < KW; ++j) { wr[e][j] = j < k ? w[(c0 + e) * k + j] : 0.f; A[e][j] = 0.f; }
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
      const float z0 = fmaf(xh0, ga[0], be[0]), z1 = fmaf(xh1, ga[1], be[1]);
      const float dz0 = d0 * (FAST ? gelu_fast<true>(z0) : gelu_grad_f(z0));
      const float dz1 = d1 * (FAST ? gelu_fast<true>(z1) : gelu_grad_f(z1));
      s1[0] += dz0; s1[1] += dz1;
      s2[0] = fmaf(dz0, xh0, s2[0]); s2[1] = fmaf(dz1, xh1, s2[1]);
#pragma unroll
      for (int j = 0; j < KW; ++j) { A[0][j] = fmaf(dz0, xv[j], A[0][j]); A[1][j] = fmaf(dz1, xv[j], A[1][j]); }
      d0 = n0; d1 = n1;
    }
    float* o = part + (((long long)b * nch + ch) * C + c0) * (KW + 2);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
      for (int j = 0; j < KW; ++j) o[e * (KW + 2) + j] = A[e][j];
      o[e * (KW + 2) + KW] = s1[e];
      o[e * (KW + 2) + KW + 1] = s2[e];
    }
  }
}
// sums[b][c][KW + 2] (double) = sum over chunks of part.  Block = 32 (c, i) entries x 8 chunk lanes.
__global__ __launch_bounds__(256) void conv0_bwd_reduce_kernel(const float* __restrict__ part, double* __restrict__ sums, int C,
                                                               int nch, int nv) {
  __shared__ double red[8][32];
  const int b = blockIdx.y, il = threadIdx.x & 31, j = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + il;      // index into [C][nv]
  const int n = C * nv;
  double s = 0.0;
  if (i < n)
    for (int ch = j; ch < nch; ch += 8) s += (double)part[((long long)b * nch + ch) * n + i];
  red[j][il] = s;
  __syncthreads();
  if (j == 0 && i < n) {
    s = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][il];
    sums[(long long)b * n + i] = s;
  }
}
// dw[c][j] += gscale sum_b rstd gamma [A_j - S1 M_j / L - S2 rstd ((R w)_j - mean M_j) / L];
// dgamma[c] += gscale sum_b S2;  dbeta[c] += gscale sum_b S1
// One thread per (channel, tap) -- 16 tap lanes per channel, lane k..15 idle, lane 0 also owns dgamma / dbeta.  (One thread
// per channel walked B x k x k fp64 terms alone: 124 us for 8 waves of dependent double arithmetic.)
__global__ __launch_bounds__(256) void conv0_bwd_final_kernel(const double* __restrict__ sums, const double* __restrict__ mom,
                                                             const float* __restrict__ w, const float* __restrict__ gamma,
                                                             const float* __restrict__ stats, float* __restrict__ dw,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C,
                                                             int k, int L, int nv, float gscale) {
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4), j = threadIdx.x & 15;
  if (c >= C || j >= k) return;
  double g1 = 0.0, g2 = 0.0, acc = 0.0;
  const double ga = (double)gamma[c], invL = 1.0 / (double)L;
  for (int b = 0; b < B; ++b) {
    const double* sb = sums + ((long long)b * C + c) * nv;
    const double* mb = mom + (long long)b * nmom(k);
    const double mu = (double)stats[((long long)b * C + c) * 2], rs = (double)stats[((long long)b * C + c) * 2 + 1];
    const double S1 = sb[nv - 2], S2 = sb[nv - 1];
    g1 += S1; g2 += S2;
    double rw = 0.0;   // (R w)_j
    for (int jp = 0; jp < k; ++jp) {
      const int a = j < jp ? j : jp, bq = j < jp ? jp : j;
      rw += (double)w[c * k + jp] * mb[ridx(k, a, bq)];
    }
    acc += rs * ga * (sb[j] - S1 * invL * mb[j] - S2 * invL * rs * (rw - mu * mb[j]));
  }
  if (dw) dw[c * k + j] += gscale * (float)acc;
  if (j == 0) {
    if (dgamma) dgamma[c] += gscale * (float)g2;
    if (dbeta) dbeta[c] += gscale * (float)g1;
  }
}

// workspace layout (floats unless noted): [chunk partials: max(B nch nmom, B nch C (KWmax + 2))] [mom: B nmom doubles]
// [sums: B C (KWmax + 2) doubles]
struct Ws { float* part; double* mom; double* sums; };
__host__ inline int64_t ws_part_floats(int B, int nch, int C, int k) {
  const int64_t a = (int64_t)B * nch * nmom(k), b = (int64_t)B * nch * C * (MAXK + 2);
  return ((a > b ? a : b) + 1) / 2 * 2;   // keep the doubles 8-byte aligned
}
__host__ inline Ws carve(void* ws, int B, int nch, int C, int k) {
  Ws r;
  r.part = (float*)ws;
  r.mom = (double*)(r.part + ws_part_floats(B, nch, C, k));
  r.sums = r.mom + (int64_t)B * nmom(k);
  return r;
}

template <int KW>
void launch_moments(const float* wav, const Ws& W, int B, int S, int L, int k, int stride, int nch, size_t shm, hipStream_t s) {
  hipLaunchKernelGGL((conv0_moments_kernel<KW>), dim3(nch, B), dim3(256), shm, s, wav, W.part, S, L, k, stride, nch);
  hipLaunchKernelGGL(conv0_moments_final_kernel, dim3(B), dim3(256), 0, s, W.part, W.mom, nmom(k), nch);
}

}  // namespace

extern "C" int64_t st5_conv0_ws_bytes(int32_t B, int32_t S, int32_t C, int32_t k, int32_t stride) {
  const int L = out_len(S, k, stride);
  const int64_t nch = (L + TCH - 1) / TCH;
  return ws_part_floats(B, (int)nch, C, k) * (int64_t)sizeof(float) +
         ((int64_t)B * nmom(k) + (int64_t)B * C * (MAXK + 2)) * (int64_t)sizeof(double);
}

extern "C" int st5_conv0_gn_gelu_fwd(const float* wav, const float* w, const float* gamma, const float* beta,
                                     void* out, float* stats, void* ws, int32_t B, int32_t S, int32_t C, int32_t k,
                                     int32_t stride, float eps, int dtype, void* stream) {
  if (!wav || !w || !gamma || !beta || !out || !stats || !ws) return ST5_ERR_ARG;
  if (C % 8 || C > 2048 || 256 % (C / 8 > 256 ? 256 : C / 8) || k > MAXK || k < 1 || stride < 1) return ST5_ERR_ARG;
  if (C / 8 > 256) return ST5_ERR_ARG;
  const int L = out_len(S, k, stride);
  if (L <= 0 || B <= 0) return ST5_ERR_ARG;
  if (dtype != ST5_BF16 && dtype != ST5_F32) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nch = (L + TCH - 1) / TCH;
  const size_t shm = (size_t)(TCH * stride + k + MAXK) * sizeof(float);
  const Ws W = carve(ws, B, nch, C, k);
  if (k <= 10) launch_moments<10>(wav, W, B, S, L, k, stride, nch, shm, s);
  else launch_moments<MAXK>(wav, W, B, S, L, k, stride, nch, shm, s);
  hipLaunchKernelGGL(conv0_stats_from_moments_kernel, dim3((C + 255) / 256, B), dim3(256), 0, s, W.mom, w, stats, C, k, L, eps);
#define APPLY(TT, KW)                                                                                           \
  hipLaunchKernelGGL((conv0_apply_kernel<TT, KW>), dim3(nch, B), dim3(256), shm, s, wav, w, gamma, beta, stats, \
                     (TT*)out, S, L, C, k, stride)
  if (dtype == ST5_BF16) { if (k <= 10) APPLY(bf16_t, 10); else APPLY(bf16_t, MAXK); }
  else { if (k <= 10) APPLY(float, 10); else APPLY(float, MAXK); }
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
  if (dtype != ST5_BF16 && dtype != ST5_F32) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nch = (L + TCH - 1) / TCH;
  const size_t shm = (size_t)(TCH * stride + k + MAXK) * sizeof(float);
  const Ws W = carve(ws, B, nch, C, k);
  // waveform moments again (0.64 MB/clip; cheaper than keeping them alive between forward and backward)
  if (k <= 10) launch_moments<10>(wav, W, B, S, L, k, stride, nch, shm, s);
  else launch_moments<MAXK>(wav, W, B, S, L, k, stride, nch, shm, s);
  const int KWv = k <= 10 ? 10 : MAXK, nv = KWv + 2;
  {   // ST5_POISON=1 (debug): the partials region is NaN before the backward kernel fills it -- a reduce that ran ahead of a
      // block of conv0_bwd_kernel, or a block that never stored, then shows as NaN instead of as last step's value
    static const bool poison = [] { const char* e = getenv("ST5_POISON"); return e && e[0] == '1'; }();
    if (poison && hipMemsetAsync(W.part, 0xFF, (size_t)B * nch * C * nv * sizeof(float), s) != hipSuccess) return ST5_ERR_LAUNCH;
  }
#define BWD(TT, KW)                                                                                               \
  hipLaunchKernelGGL((conv0_bwd_kernel<TT, KW>), dim3(nch, B), dim3(256), shm, s, wav, w, gamma, beta, stats,     \
                     (const TT*)dY, W.part, S, L, C, k, stride, nch)
  if (dtype == ST5_BF16) { if (k <= 10) BWD(bf16_t, 10); else BWD(bf16_t, MAXK); }
  else { if (k <= 10) BWD(float, 10); else BWD(float, MAXK); }
#undef BWD
  hipLaunchKernelGGL(conv0_bwd_reduce_kernel, dim3((C * nv + 31) / 32, B), dim3(256), 0, s, W.part, W.sums, C, nch, nv);
  hipLaunchKernelGGL(conv0_bwd_final_kernel, dim3((C + 15) / 16), dim3(256), 0, s, W.sums, W.mom, w, gamma, stats, dw, dgamma,
                     dbeta, B, C, k, L, nv, gscale);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
