// Gumbel vector quantizer (fairseq GumbelVectorQuantizer as SpeechT5 uses it: speecht5.py:95-107, 858-882; semantics in
// SURVEY.md App. A) + the time-wise mix of codes and encoder states (speecht5.py:870-877), as three kernels.
//
//   logits [N, G*V] fp32 (the weight_proj GEMM's output), gumbel noise [N, G*V] fp32 (drawn by the caller: -log(Exp(1))),
//   vars [G*V, Dg] fp32 code book.  Per (row n, group g):
//     k_hard = argmax_v logits                 -> code_perplexity = sum_g exp(-sum_v h log(h + 1e-7)),  h = mean_n onehot(k_hard)
//     p      = softmax_v logits                -> prob_perplexity = sum_g exp(-sum_v a log(a + 1e-7)),  a = mean_n p
//     y      = softmax_v((logits + gumbel)/tau), idx = argmax y (training) | k_hard (eval)
//     q[n, g*Dg:(g+1)*Dg] = vars[g*V + idx]    (hard one-hot, straight-through gradient through y)
//   out[n] = w[t] * q[n] + (1 - w[t]) * enc[n]   (t = n % T; w = NULL: out = q)
// The reference runs ~65 element-wise / reduction kernels forward and as many backward for this; here: forward + finalize,
// and for the backward one GEMM (dsel = dOut . vars^T, st5_gemm), this file's gradient kernel and the deterministic
// row scatter of elementwise.hip for dvars.  Deterministic: every wave owns one group and a fixed set of rows; the
// perplexity sums are reduced in wave order by the finalize kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "../../include/speecht5_hip.h"

#define DISPATCH(dtype, CALL_BF, CALL_F)   \
  if (dtype == ST5_BF16) { CALL_BF; }      \
  else if (dtype == ST5_F32) { CALL_F; }   \
  else return ST5_ERR_ARG;

namespace {

constexpr int VQ_BLOCKS = 512;          // 2048 waves: a wave walks its rows one after the other (latency-bound), so many short walks
constexpr int VP = 128;                 // padded V (two values per lane)
constexpr float PERP_EPS = 1e-7f;

struct Arg2 { float v; int i; };
__device__ __forceinline__ Arg2 wave_argmax(float v, int i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
  return {v, i};
}

// part: [nwaves][2][VP] (p sums, hard counts); wave `wid` owns group wid % G and rows wid / G, wid / G + nwg, ...
template <typename T>
__global__ __launch_bounds__(256) void vq_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ gumbel,
                                                     const float* __restrict__ vars, const T* __restrict__ enc, const float* __restrict__ w,
                                                     float tau, const float* __restrict__ tau_dev, int training, T* __restrict__ out,
                                                     int* __restrict__ idx_out, float* __restrict__ part, int N, int G, int V, int Dg, int Tn) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int nwg = nwaves / G;                       // waves per group (trailing waves idle)
  float ps0 = 0.f, ps1 = 0.f, c0 = 0.f, c1 = 0.f;
  if (wid < nwg * G) {
    const int g = wid % G;
    const float inv_tau = 1.f / (tau_dev ? tau_dev[0] : tau);
    const int v0 = lane, v1 = lane + 64;
    const int d = G * Dg;
    for (int n = wid / G; n < N; n += nwg) {
      const long long base = (long long)n * G * V + (long long)g * V;
      const float l0 = v0 < V ? logits[base + v0] : -INFINITY, l1 = v1 < V ? logits[base + v1] : -INFINITY;
      const Arg2 a0 = l1 > l0 ? Arg2{l1, v1} : Arg2{l0, v0};
      const Arg2 hard = wave_argmax(a0.v, a0.i);
      const float e0 = v0 < V ? expf(l0 - hard.v) : 0.f, e1 = v1 < V ? expf(l1 - hard.v) : 0.f;
      const float inv = 1.f / wave_sum(e0 + e1);
      ps0 += e0 * inv; ps1 += e1 * inv;
      c0 += hard.i == v0 ? 1.f : 0.f; c1 += hard.i == v1 ? 1.f : 0.f;
      int idx = hard.i;
      if (training) {
        const float z0 = v0 < V ? (l0 + gumbel[base + v0]) * inv_tau : -INFINITY, z1 = v1 < V ? (l1 + gumbel[base + v1]) * inv_tau : -INFINITY;
        const Arg2 b0 = z1 > z0 ? Arg2{z1, v1} : Arg2{z0, v0};
        idx = wave_argmax(b0.v, b0.i).i;
      }
      if (lane == 0) idx_out[(long long)n * G + g] = g * V + idx;   // code-book row
      const float* code = vars + ((long long)g * V + idx) * Dg;
      const float wt = w ? w[n % Tn] : 1.f;
      for (int c = lane; c < Dg; c += 64) {
        const long long o = (long long)n * d + (long long)g * Dg + c;
        const float q = code[c];
        out[o] = Elem<T>::from_f(w ? wt * q + (1.f - wt) * Elem<T>::to_f(enc[o]) : q);
      }
    }
  }
  float* pw = part + (long long)wid * 2 * VP;
  pw[lane] = ps0; pw[64 + lane] = ps1; pw[VP + lane] = c0; pw[VP + 64 + lane] = c1;
}

// avg [G][VP] (mean softmax), out2 = {code_perplexity, prob_perplexity}.  One block of 256 threads: thread (half, v) sums the
// partials of half of the group's waves in wave order, the halves are added in order, then wave 0 / wave 2 reduce the entropies.
__global__ __launch_bounds__(256) void vq_final_kernel(const float* __restrict__ part, int nwaves, int N, int G, int V, float* __restrict__ avg,
                                                      float* __restrict__ out2) {
  __shared__ float sh[2][2][VP];     // [half][p | count][v]
  __shared__ float ent[2];
  const int v = threadIdx.x & (VP - 1), half = threadIdx.x >> 7, lane = threadIdx.x & 63;
  const int nwg = nwaves / G;
  const int k0 = half ? nwg / 2 : 0, k1 = half ? nwg : nwg / 2;
  float code_pp = 0.f, prob_pp = 0.f;
  for (int g = 0; g < G; ++g) {
    float a = 0.f, h = 0.f;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {     // fixed order, 16 independent loads in flight
      float pa[8], ph[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const float* pw = part + (long long)((k + u) * G + g) * 2 * VP; pa[u] = pw[v]; ph[u] = pw[VP + v]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += pa[u]; h += ph[u]; }
    }
    for (; k < k1; ++k) {
      const float* pw = part + (long long)(k * G + g) * 2 * VP;
      a += pw[v]; h += pw[VP + v];
    }
    __syncthreads();
    sh[half][0][v] = a; sh[half][1][v] = h;
    __syncthreads();
    const float invn = 1.f / (float)N;
    if (threadIdx.x < VP) {
      a = (sh[0][0][v] + sh[1][0][v]) * invn;
      h = (sh[0][1][v] + sh[1][1][v]) * invn;
      avg[g * VP + v] = a;
      sh[0][0][v] = v < V ? a * logf(a + PERP_EPS) : 0.f;
      sh[0][1][v] = v < V ? h * logf(h + PERP_EPS) : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const float ha = wave_sum(sh[0][0][lane] + sh[0][0][64 + lane]);
      const float hh = wave_sum(sh[0][1][lane] + sh[0][1][64 + lane]);
      if (lane == 0) { ent[0] = ha; ent[1] = hh; }
    }
    __syncthreads();
    prob_pp += expf(-ent[0]);
    code_pp += expf(-ent[1]);
  }
  if (threadIdx.x == 0) { out2[0] = code_pp; out2[1] = prob_pp; }
}

// dlogits[n, g, v] = w_t * y (dsel - <y, dsel>) / tau  +  p (gA - <p, gA>),   gA_v = g_pp exp(H_g) (-log(a_v + eps) - a_v / (a_v + eps)) / N
// d_enc[n] = (1 - w_t) dOut[n]   (when enc took part in the mix)
template <typename T>
__global__ __launch_bounds__(256) void vq_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ gumbel, const float* __restrict__ dsel,
                                                     int dsel_ld, const float* __restrict__ avg, const float* __restrict__ g_pp,
                                                     const T* __restrict__ dout, const float* __restrict__ w, float tau,
                                                     const float* __restrict__ tau_dev, int training, float* __restrict__ dlogits,
                                                     T* __restrict__ denc, int N, int G, int V, int Dg, int Tn) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int nwg = nwaves / G;
  if (wid >= nwg * G) return;
  const int g = wid % G;
  const float inv_tau = 1.f / (tau_dev ? tau_dev[0] : tau);
  const int v0 = lane, v1 = lane + 64;
  const bool in0 = v0 < V, in1 = v1 < V;
  const int d = G * Dg;
  // gradient of prob_perplexity w.r.t. the mean softmax of this group
  float gA0 = 0.f, gA1 = 0.f;
  if (g_pp) {
    const float a0 = avg[g * VP + v0], a1 = avg[g * VP + v1];
    const float H = -wave_sum((in0 ? a0 * logf(a0 + PERP_EPS) : 0.f) + (in1 ? a1 * logf(a1 + PERP_EPS) : 0.f));
    const float k = g_pp[0] * expf(H) / (float)N;
    gA0 = in0 ? k * (-logf(a0 + PERP_EPS) - a0 / (a0 + PERP_EPS)) : 0.f;
    gA1 = in1 ? k * (-logf(a1 + PERP_EPS) - a1 / (a1 + PERP_EPS)) : 0.f;
  }
  for (int n = wid / G; n < N; n += nwg) {
    const long long base = (long long)n * G * V + (long long)g * V;
    const float wt = w ? w[n % Tn] : 1.f;
    const float l0 = in0 ? logits[base + v0] : -INFINITY, l1 = in1 ? logits[base + v1] : -INFINITY;
    float r0 = 0.f, r1 = 0.f;
    if (g_pp) {
      const float mx = wave_max(fmaxf(l0, l1));
      const float e0 = in0 ? expf(l0 - mx) : 0.f, e1 = in1 ? expf(l1 - mx) : 0.f;
      const float inv = 1.f / wave_sum(e0 + e1);
      const float p0 = e0 * inv, p1 = e1 * inv;
      const float dot = wave_sum(p0 * gA0 + p1 * gA1);
      r0 = p0 * (gA0 - dot); r1 = p1 * (gA1 - dot);
    }
    if (training && dsel) {
      const float z0 = in0 ? (l0 + gumbel[base + v0]) * inv_tau : -INFINITY, z1 = in1 ? (l1 + gumbel[base + v1]) * inv_tau : -INFINITY;
      const float mz = wave_max(fmaxf(z0, z1));
      const float e0 = in0 ? expf(z0 - mz) : 0.f, e1 = in1 ? expf(z1 - mz) : 0.f;
      const float inv = 1.f / wave_sum(e0 + e1);
      const float y0 = e0 * inv, y1 = e1 * inv;
      const float* ds = dsel + (long long)n * dsel_ld + (long long)g * VP;
      const float s0 = in0 ? ds[v0] : 0.f, s1 = in1 ? ds[v1] : 0.f;
      const float dot = wave_sum(y0 * s0 + y1 * s1);
      const float k = wt * inv_tau;
      r0 += k * y0 * (s0 - dot); r1 += k * y1 * (s1 - dot);
    }
    if (in0) dlogits[base + v0] = r0;
    if (in1) dlogits[base + v1] = r1;
    if (denc) {
      for (int c = lane; c < Dg; c += 64) {
        const long long o = (long long)n * d + (long long)g * Dg + c;
        denc[o] = Elem<T>::from_f((1.f - wt) * Elem<T>::to_f(dout[o]));
      }
    }
  }
}

}  // namespace

extern "C" int64_t st5_vq_ws_bytes(void) { return (int64_t)VQ_BLOCKS * 4 * 2 * VP * sizeof(float); }
extern "C" int32_t st5_vq_vpad(void) { return VP; }

extern "C" int st5_vq_fwd(const float* logits, const float* gumbel, const float* vars, const void* enc, const float* mix_w, float tau,
                          const float* tau_dev, int32_t training, void* out, int32_t* idx, float* avg /* [G][128] */, float* perp2,
                          void* ws, int32_t N, int32_t G, int32_t V, int32_t Dg, int32_t T, int dtype, void* stream) {
  if (!logits || !vars || !out || !idx || !avg || !perp2 || !ws || N <= 0 || G <= 0 || V <= 0 || V > VP || Dg <= 0 || T <= 0) return ST5_ERR_ARG;
  if (G > VQ_BLOCKS * 4 || (training && !gumbel) || (mix_w && !enc)) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH(dtype, hipLaunchKernelGGL(vq_fwd_kernel<bf16_t>, dim3(VQ_BLOCKS), dim3(256), 0, s, logits, gumbel, vars, (const bf16_t*)enc, mix_w, tau, tau_dev, training, (bf16_t*)out, idx, (float*)ws, N, G, V, Dg, T),
           hipLaunchKernelGGL(vq_fwd_kernel<float>, dim3(VQ_BLOCKS), dim3(256), 0, s, logits, gumbel, vars, (const float*)enc, mix_w, tau, tau_dev, training, (float*)out, idx, (float*)ws, N, G, V, Dg, T));
  hipLaunchKernelGGL(vq_final_kernel, dim3(1), dim3(256), 0, s, (const float*)ws, VQ_BLOCKS * 4, N, G, V, avg, perp2);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_vq_bwd(const float* logits, const float* gumbel, const float* dsel, int32_t dsel_ld, const float* avg, const float* g_prob_perp,
                          const void* dout, const float* mix_w, float tau, const float* tau_dev, int32_t training, float* dlogits, void* denc,
                          int32_t N, int32_t G, int32_t V, int32_t Dg, int32_t T, int dtype, void* stream) {
  if (!logits || !dlogits || !avg || N <= 0 || G <= 0 || V <= 0 || V > VP || Dg <= 0 || T <= 0) return ST5_ERR_ARG;
  if ((training && dsel && !gumbel) || (denc && !dout) || (dsel && dsel_ld < G * VP)) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH(dtype, hipLaunchKernelGGL(vq_bwd_kernel<bf16_t>, dim3(VQ_BLOCKS), dim3(256), 0, s, logits, gumbel, dsel, dsel_ld, avg, g_prob_perp, (const bf16_t*)dout, mix_w, tau, tau_dev, training, dlogits, (bf16_t*)denc, N, G, V, Dg, T),
           hipLaunchKernelGGL(vq_bwd_kernel<float>, dim3(VQ_BLOCKS), dim3(256), 0, s, logits, gumbel, dsel, dsel_ld, avg, g_prob_perp, (const float*)dout, mix_w, tau, tau_dev, training, dlogits, (float*)denc, N, G, V, Dg, T));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
