// CTC loss of the ASR fine-tuning criterion and the guided-attention loss of the TTS criterion.
//
// CTC (SpeechT5/speecht5/criterions/speech_to_text_loss.py:301-337: F.ctc_loss(lprobs [T,B,V] fp32, flat targets, input /
// target lengths, blank, reduction="sum", zero_infinity) with cuDNN off, i.e. torch's own alpha-beta recursion):
//   l' = blank, l_1, blank, l_2, ..., blank  (S = 2 L + 1 states)
//   alpha_t(s) = lp_t(l'_s) + logsumexp(alpha_{t-1}(s), alpha_{t-1}(s-1), [l'_s != l'_{s-2}] alpha_{t-1}(s-2))
//   nll = -logsumexp(alpha_{T_b-1}(S-1), alpha_{T_b-1}(S-2));  beta the mirror image from the end
//   d nll / d lp_t(c), in the form torch returns it (already combined with the log-softmax in front: p - posterior):
//       exp(lp_t(c)) - exp(logsumexp_{s: l'_s = c}(alpha_t(s) + beta_t(s)) + nll - lp_t(c)),  0 for t >= T_b
// One block per sentence walks the frames; the state row lives in LDS (double buffered), alpha rows are kept in the
// workspace for the backward.  The per-class sums of the backward run over the label positions sorted by class (built once
// per sentence in LDS), so every class is summed by one thread in a fixed order: deterministic, no atomics.
//
// Guided attention (criterions/text_to_speech_loss.py:370-427, espnet GuidedMultiHeadAttentionLoss):
//   loss = alpha * mean_{b,h,to<olen_b,ti<ilen_b} (1 - exp(-(ti/ilen_b - to/olen_b)^2 / (2 sigma^2))) * att[b,h,to,ti]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "common.h"
#include "../../include/speecht5_hip.h"

namespace {

constexpr int CTC_THREADS = 256;
constexpr int GA_BLOCKS = 256;

__device__ __forceinline__ float lse3(float a, float b, float c) {
  float m = fmaxf(a, fmaxf(b, c));
  if (m == -INFINITY) m = 0.f;
  return logf(expf(a - m) + expf(b - m) + expf(c - m)) + m;
}

// state s of the blank-extended label sequence
__device__ __forceinline__ int prime(const int* tg, int s, int blank) { return (s & 1) ? tg[s >> 1] : blank; }

struct CtcDims {
  int T, B, V, maxL, blank;
};

// LDS layout (ints / floats): tg[maxL] | row0[S] | row1[S] | (backward only) ab[S] perm[maxL] start[V + 1] | red[CTC_THREADS]
__global__ __launch_bounds__(CTC_THREADS) void ctc_alpha_kernel(const float* __restrict__ lp, const long long* __restrict__ targets,
                                                                const long long* __restrict__ offsets,
                                                                const long long* __restrict__ in_len, const long long* __restrict__ tg_len,
                                                                CtcDims d, float* __restrict__ alpha, float* __restrict__ nll) {
  extern __shared__ int smem[];
  const int b = blockIdx.x;
  const int Smax = 2 * d.maxL + 1;
  int* tg = smem;
  float* row[2] = {reinterpret_cast<float*>(smem + d.maxL), reinterpret_cast<float*>(smem + d.maxL + Smax)};
  long long Lq = tg_len[b];
  const int L = (int)(Lq < 0 ? 0 : (Lq > d.maxL ? d.maxL : Lq));
  const int S = 2 * L + 1;
  long long Tq = in_len[b];
  const int Tb = (int)(Tq < 0 ? 0 : (Tq > d.T ? d.T : Tq));
  for (int i = threadIdx.x; i < L; i += CTC_THREADS) tg[i] = (int)targets[offsets[b] + i];
  __syncthreads();
  float* ab = alpha + (long long)b * d.T * Smax;
  if (Tb == 0) {
    if (threadIdx.x == 0) nll[b] = L == 0 ? 0.f : INFINITY;
    return;
  }
  const long long tstride = (long long)d.B * d.V;
  const float* lpb = lp + (long long)b * d.V;
  for (int s = threadIdx.x; s < S; s += CTC_THREADS) {
    float a = -INFINITY;
    if (s == 0) a = lpb[d.blank];
    else if (s == 1) a = lpb[tg[0]];
    row[0][s] = a;
    ab[s] = a;
  }
  __syncthreads();
  for (int t = 1; t < Tb; ++t) {
    const float* prev = row[(t - 1) & 1];
    float* cur = row[t & 1];
    const float* lpt = lpb + t * tstride;
    for (int s = threadIdx.x; s < S; s += CTC_THREADS) {
      const int c = prime(tg, s, d.blank);
      const float a1 = prev[s];
      const float a2 = s > 0 ? prev[s - 1] : -INFINITY;
      const float a3 = (s > 1 && prime(tg, s - 2, d.blank) != c) ? prev[s - 2] : -INFINITY;
      const float a = lse3(a1, a2, a3) + lpt[c];
      cur[s] = a;
      ab[(long long)t * Smax + s] = a;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float* last = row[(Tb - 1) & 1];
    const float l1 = last[S - 1], l2 = S > 1 ? last[S - 2] : -INFINITY;
    float m = fmaxf(l1, l2);
    if (m == -INFINITY) m = 0.f;
    nll[b] = -(logf(expf(l1 - m) + expf(l2 - m)) + m);
  }
}

// loss = sum_b nll_b (inf -> 0 with zero_infinity); fixed order, fp64
__global__ void ctc_sum_kernel(const float* __restrict__ nll, int B, int zero_infinity, float* __restrict__ loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double acc = 0.0;
  for (int b = 0; b < B; ++b) {
    const float v = nll[b];
    acc += (zero_infinity && v == INFINITY) ? 0.0 : (double)v;
  }
  loss[0] = (float)acc;
}

__global__ __launch_bounds__(CTC_THREADS) void ctc_grad_kernel(const float* __restrict__ lp, const long long* __restrict__ targets,
                                                               const long long* __restrict__ offsets,
                                                               const long long* __restrict__ in_len, const long long* __restrict__ tg_len,
                                                               CtcDims d, int zero_infinity, const float* __restrict__ alpha,
                                                               const float* __restrict__ nll, const float* __restrict__ gout,
                                                               float* __restrict__ grad) {
  extern __shared__ int smem[];
  const int b = blockIdx.x;
  const int Smax = 2 * d.maxL + 1;
  int* tg = smem;
  float* row[2] = {reinterpret_cast<float*>(smem + d.maxL), reinterpret_cast<float*>(smem + d.maxL + Smax)};
  float* abs_ = reinterpret_cast<float*>(smem + d.maxL + 2 * Smax);
  int* perm = smem + d.maxL + 3 * Smax;
  int* start = perm + d.maxL;
  float* red = reinterpret_cast<float*>(start + d.V + 1);
  long long Lq = tg_len[b];
  const int L = (int)(Lq < 0 ? 0 : (Lq > d.maxL ? d.maxL : Lq));
  const int S = 2 * L + 1;
  long long Tq = in_len[b];
  const int Tb = (int)(Tq < 0 ? 0 : (Tq > d.T ? d.T : Tq));
  const long long tstride = (long long)d.B * d.V;
  const float* lpb = lp + (long long)b * d.V;
  float* gb = grad + (long long)b * d.V;
  const float nl = nll[b];
  const float gr = gout[0];
  const bool dead = zero_infinity && nl == INFINITY;
  // frames past the sentence's length (and every frame of a zeroed sentence) take no gradient
  for (int t = dead ? 0 : Tb; t < d.T; ++t)
    for (int c = threadIdx.x; c < d.V; c += CTC_THREADS) gb[t * tstride + c] = 0.f;
  if (dead || Tb == 0) return;
  for (int i = threadIdx.x; i < L; i += CTC_THREADS) tg[i] = (int)targets[offsets[b] + i];
  __syncthreads();
  // label positions grouped by class, in position order inside a class: counting sort, one thread per class
  for (int c = threadIdx.x; c <= d.V; c += CTC_THREADS) start[c] = 0;
  __syncthreads();
  for (int c = threadIdx.x; c < d.V; c += CTC_THREADS) {
    int n = 0;
    for (int i = 0; i < L; ++i) n += tg[i] == c;
    start[c + 1] = n;
  }
  __syncthreads();
  if (threadIdx.x == 0)
    for (int c = 0; c < d.V; ++c) start[c + 1] += start[c];
  __syncthreads();
  for (int c = threadIdx.x; c < d.V; c += CTC_THREADS) {
    int k = start[c];
    if (start[c + 1] > k)
      for (int i = 0; i < L; ++i)
        if (tg[i] == c) perm[k++] = i;
  }
  __syncthreads();
  const float* al = alpha + (long long)b * d.T * Smax;
  for (int t = Tb - 1; t >= 0; --t) {
    float* cur = row[t & 1];
    const float* nxt = row[(t + 1) & 1];
    const float* lpt = lpb + t * tstride;
    float mx = -INFINITY;
    for (int s = threadIdx.x; s < S; s += CTC_THREADS) {
      const int c = prime(tg, s, d.blank);
      float bt;
      if (t == Tb - 1) {
        bt = (s == S - 1 || s == S - 2) ? lpt[c] : -INFINITY;
      } else {
        const float b1 = nxt[s];
        const float b2 = s < S - 1 ? nxt[s + 1] : -INFINITY;
        const float b3 = (s < S - 2 && prime(tg, s + 2, d.blank) != c) ? nxt[s + 2] : -INFINITY;
        bt = lse3(b1, b2, b3) + lpt[c];
      }
      cur[s] = bt;
      const float v = al[(long long)t * Smax + s] + bt;
      abs_[s] = v;
      mx = fmaxf(mx, v);
    }
    // block maximum of alpha + beta, then the blank class (even states) as a fixed-order block sum
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (mx == -INFINITY) mx = 0.f;
    __syncthreads();
    float bs = 0.f;
    for (int s = 2 * threadIdx.x; s < S; s += 2 * CTC_THREADS) bs += expf(abs_[s] - mx);
    bs = wave_sum(bs);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bs;
    __syncthreads();
    const float blank_sum = (red[0] + red[1]) + (red[2] + red[3]);
    for (int c = threadIdx.x; c < d.V; c += CTC_THREADS) {
      float sum = c == d.blank ? blank_sum : 0.f;
      for (int k = start[c]; k < start[c + 1]; ++k) sum += expf(abs_[2 * perm[k] + 1] - mx);
      const float l = lpt[c];
      const float post = sum > 0.f ? expf(logf(sum) + mx + nl - l) : 0.f;
      gb[t * tstride + c] = (expf(l) - post) * gr;
    }
    __syncthreads();
  }
}

// ---- guided attention ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ga_weight(int to, int ti, float olen, float ilen, float inv_two_sigma_sq) {
  const float dlt = (float)ti / ilen - (float)to / olen;
  return 1.f - expf(-(dlt * dlt) * inv_two_sigma_sq);
}

__global__ __launch_bounds__(256) void guided_attn_partial_kernel(const float* __restrict__ att, const long long* __restrict__ ilens,
                                                                  const long long* __restrict__ olens, int B, int H, int To, int Ti,
                                                                  float inv_two_sigma_sq, float* __restrict__ part) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  const long long rows = (long long)B * H * To;
  for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
    const int to = (int)(row % To);
    const int b = (int)(row / ((long long)H * To));
    const long long ol = olens[b], il = ilens[b];
    if (to >= ol) continue;
    const int n = (int)(il < Ti ? il : Ti);
    const float* a = att + row * Ti;
    for (int ti = lane; ti < n; ti += 64) acc = fmaf(ga_weight(to, ti, (float)ol, (float)il, inv_two_sigma_sq), a[ti], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = alpha * sum / count, out[1] = alpha / count (the backward's factor);  count = H * sum_b min(olen, To) * min(ilen, Ti)
__global__ void guided_attn_final_kernel(const float* __restrict__ part, int nparts, const long long* __restrict__ ilens,
                                         const long long* __restrict__ olens, int B, int H, int To, int Ti, float alpha,
                                         float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < nparts; ++i) s += (double)part[i];
  double cnt = 0.0;
  for (int b = 0; b < B; ++b) {
    long long ol = olens[b] < To ? olens[b] : To, il = ilens[b] < Ti ? ilens[b] : Ti;
    if (ol < 0) ol = 0;
    if (il < 0) il = 0;
    cnt += (double)ol * (double)il;
  }
  cnt *= H;
  out[0] = (float)(alpha * s / cnt);     // (an empty selection gives nan, as torch.mean of nothing does)
  out[1] = (float)(alpha / cnt);
}

__global__ __launch_bounds__(256) void guided_attn_bwd_kernel(const long long* __restrict__ ilens, const long long* __restrict__ olens, int B,
                                                              int H, int To, int Ti, float inv_two_sigma_sq,
                                                              const float* __restrict__ out, const float* __restrict__ gout,
                                                              float* __restrict__ datt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float k = out[1] * gout[0];
  const long long rows = (long long)B * H * To;
  for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
    const int to = (int)(row % To);
    const int b = (int)(row / ((long long)H * To));
    const long long ol = olens[b], il = ilens[b];
    float* g = datt + row * Ti;
    for (int ti = lane; ti < Ti; ti += 64)
      g[ti] = (to < ol && ti < il) ? k * ga_weight(to, ti, (float)ol, (float)il, inv_two_sigma_sq) : 0.f;
  }
}

size_t ctc_lds_bytes(int maxL, int V, bool backward) {
  const size_t S = 2 * (size_t)maxL + 1;
  size_t words = maxL + 2 * S;
  if (backward) words += S + maxL + V + 1 + 8;
  return words * 4;
}

}  // namespace

extern "C" int64_t st5_ctc_loss_ws_bytes(int32_t T, int32_t B, int32_t max_target_len) {
  if (T < 0 || B < 0 || max_target_len < 0) return -1;
  return (int64_t)T * B * (2 * (int64_t)max_target_len + 1) * (int64_t)sizeof(float);
}

extern "C" int st5_ctc_loss_fwd(const float* lprobs, const int64_t* targets, const int64_t* target_offsets, const int64_t* input_lengths,
                                const int64_t* target_lengths, int32_t T, int32_t B, int32_t V, int32_t max_target_len, int32_t blank,
                                int32_t zero_infinity, float* nll, float* loss, void* ws, void* stream) {
  if (!lprobs || !target_offsets || !input_lengths || !target_lengths || !nll || !loss || !ws) return ST5_ERR_ARG;
  if (T <= 0 || B <= 0 || V <= 0 || max_target_len < 0 || blank < 0 || blank >= V) return ST5_ERR_ARG;
  if (max_target_len > 0 && !targets) return ST5_ERR_ARG;
  const size_t lds = ctc_lds_bytes(max_target_len, V, false);
  if (lds > 60 * 1024) return ST5_ERR_ARG;   // (targets of up to ~3000 labels)
  hipStream_t s = (hipStream_t)stream;
  const CtcDims d{T, B, V, max_target_len, blank};
  hipLaunchKernelGGL(ctc_alpha_kernel, dim3((unsigned)B), dim3(CTC_THREADS), lds, s, lprobs, reinterpret_cast<const long long*>(targets),
                     reinterpret_cast<const long long*>(target_offsets), reinterpret_cast<const long long*>(input_lengths),
                     reinterpret_cast<const long long*>(target_lengths), d, static_cast<float*>(ws), nll);
  HIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(ctc_sum_kernel, dim3(1), dim3(64), 0, s, nll, B, zero_infinity, loss);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_ctc_loss_bwd(const float* lprobs, const int64_t* targets, const int64_t* target_offsets, const int64_t* input_lengths,
                                const int64_t* target_lengths, int32_t T, int32_t B, int32_t V, int32_t max_target_len, int32_t blank,
                                int32_t zero_infinity, const float* nll, const float* grad_out, const void* ws, float* grad, void* stream) {
  if (!lprobs || !target_offsets || !input_lengths || !target_lengths || !nll || !grad_out || !ws || !grad) return ST5_ERR_ARG;
  if (T <= 0 || B <= 0 || V <= 0 || max_target_len < 0 || blank < 0 || blank >= V) return ST5_ERR_ARG;
  if (max_target_len > 0 && !targets) return ST5_ERR_ARG;
  const size_t lds = ctc_lds_bytes(max_target_len, V, true);
  if (lds > 60 * 1024) return ST5_ERR_ARG;
  const CtcDims d{T, B, V, max_target_len, blank};
  hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)B), dim3(CTC_THREADS), lds, (hipStream_t)stream, lprobs,
                     reinterpret_cast<const long long*>(targets), reinterpret_cast<const long long*>(target_offsets),
                     reinterpret_cast<const long long*>(input_lengths), reinterpret_cast<const long long*>(target_lengths), d, zero_infinity,
                     static_cast<const float*>(ws), nll, grad_out, grad);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int64_t st5_guided_attn_ws_bytes(void) { return (int64_t)GA_BLOCKS * sizeof(float); }

extern "C" int st5_guided_attn_fwd(const float* att, const int64_t* ilens, const int64_t* olens, int32_t B, int32_t H, int32_t To, int32_t Ti,
                                   float sigma, float alpha, float* out2, void* ws, void* stream) {
  if (!att || !ilens || !olens || !out2 || !ws || B <= 0 || H <= 0 || To <= 0 || Ti <= 0 || !(sigma > 0.f)) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const float k = 1.f / (2.f * sigma * sigma);
  hipLaunchKernelGGL(guided_attn_partial_kernel, dim3(GA_BLOCKS), dim3(256), 0, s, att, reinterpret_cast<const long long*>(ilens),
                     reinterpret_cast<const long long*>(olens), B, H, To, Ti, k, static_cast<float*>(ws));
  HIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(guided_attn_final_kernel, dim3(1), dim3(64), 0, s, static_cast<const float*>(ws), GA_BLOCKS,
                     reinterpret_cast<const long long*>(ilens), reinterpret_cast<const long long*>(olens), B, H, To, Ti, alpha, out2);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_guided_attn_bwd(const int64_t* ilens, const int64_t* olens, int32_t B, int32_t H, int32_t To, int32_t Ti, float sigma,
                                   const float* out2, const float* grad_out, float* datt, void* stream) {
  if (!ilens || !olens || !out2 || !grad_out || !datt || B <= 0 || H <= 0 || To <= 0 || Ti <= 0 || !(sigma > 0.f)) return ST5_ERR_ARG;
  const float k = 1.f / (2.f * sigma * sigma);
  const long long rows = (long long)B * H * To;
  const unsigned grid = (unsigned)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
  hipLaunchKernelGGL(guided_attn_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const long long*>(ilens),
                     reinterpret_cast<const long long*>(olens), B, H, To, Ti, k, out2, grad_out, datt);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
