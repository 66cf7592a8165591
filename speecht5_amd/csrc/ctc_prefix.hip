// CTC prefix scores for joint CTC / attention beam search (Watanabe et al. 2017, algorithm 2, many next labels at once).
// Follows what SpeechT5/speecht5/sequence_generator.py:273-418 calls once per hypothesis and step on the host
// (espnet CTCPrefixScore, in-tree copy Speech2C/speech2c/models/modules/ctc_prefix_score.py:10-112, numpy fp32):
//   r[t,0|1](h.c) = log r_t^n / r_t^b of prefix h extended by candidate c,   log_psi(h.c) = log prefix probability.
// Here every (hypothesis, candidate) pair of a step is one thread walking the T encoder frames; the CTC posterior x[T,V]
// and the per-hypothesis states stay on the device (the reference copies tokens and candidates to the host and loops in
// numpy per hypothesis).  The walk is a dependent chain of 3 log-add-exp per frame: latency-bound, ~T * 0.3 us per launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "../../include/speecht5_hip.h"

namespace {

constexpr float LOGZERO = -10000000000.0f;   // ctc_prefix_score.py:21

// numpy's logaddexp for float32 (npy_logaddexpf): x + log1p(exp(-|x - y|)) on the larger operand
__device__ __forceinline__ float logaddexp(float a, float b) {
  if (a == b) return a + 0.693147180559945309417232121458176568f;
  const float d = a - b;
  return d > 0.f ? a + log1pf(expf(-d)) : b + log1pf(expf(d));
}

// cumulative blank path: r[t,1] = sum_{u<=t} x[u,blank], r[t,0] = logzero   (ctc_prefix_score.py:27-39)
__global__ void ctc_initial_state_kernel(const float* __restrict__ x, int T, int V, int blank, float* __restrict__ r) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float acc = 0.f;
  for (int t = 0; t < T; ++t) {
    acc = t == 0 ? x[blank] : acc + x[(long long)t * V + blank];
    r[2 * t] = LOGZERO;
    r[2 * t + 1] = acc;
  }
}

__global__ void ctc_prefix_score_kernel(const float* __restrict__ x, int T, int V, int blank, int eos,
                                        const float* __restrict__ r_prev, const long long* __restrict__ last,
                                        int out_len, const long long* __restrict__ cs, int nc,
                                        float* __restrict__ log_psi_out, float* __restrict__ r_new) {
  const int h = blockIdx.x, c = threadIdx.x;
  if (c >= nc) return;
  const long long tok = cs[(long long)h * nc + c];
  const float* rp = r_prev + (long long)h * T * 2;
  float* r = r_new + ((long long)h * nc + c) * T * 2;
  // the label that repeats the prefix's last one may only follow a blank: log_phi = r_prev^b instead of r^n + r^b (:66-72)
  const bool repeat = out_len > 0 && tok == last[h];
  const int start = out_len > 1 ? out_len : 1;
  for (int t = 0; t < start - 1; ++t) { r[2 * t] = LOGZERO; r[2 * t + 1] = LOGZERO; }   // (never read; numpy leaves them unset)
  float rn, rb;
  if (out_len == 0) { rn = x[tok]; rb = LOGZERO; }   // (:55-57)
  else { rn = LOGZERO; rb = LOGZERO; }               // (:59)
  if (start - 1 < T) { r[2 * (start - 1)] = rn; r[2 * (start - 1) + 1] = rb; }
  float log_psi = rn;                                 // (:78)
  for (int t = start; t < T; ++t) {
    const float pn = rp[2 * (t - 1)], pb = rp[2 * (t - 1) + 1];
    const float phi = repeat ? pb : logaddexp(pn, pb);
    const float xt = x[(long long)t * V + tok];
    const float nn = logaddexp(rn, phi) + xt;                               // (:80)
    const float nb = logaddexp(rn, rb) + x[(long long)t * V + blank];       // (:81-83)
    log_psi = logaddexp(log_psi, phi + xt);                                 // (:84)
    rn = nn; rb = nb;
    r[2 * t] = rn; r[2 * t + 1] = rb;
  }
  if (tok == eos) log_psi = logaddexp(rp[2 * (T - 1)], rp[2 * (T - 1) + 1]);   // P(... eos | X): the prefix itself ends (:87-89)
  if (tok == blank) log_psi = LOGZERO;                                           // (:92-94)
  log_psi_out[(long long)h * nc + c] = log_psi;
}

}  // namespace

extern "C" int st5_ctc_initial_state(const float* x, int T, int V, int blank, float* r, void* stream) {
  if (!x || !r || T <= 0 || V <= 0 || blank < 0 || blank >= V) return ST5_ERR_ARG;
  hipLaunchKernelGGL(ctc_initial_state_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), x, T, V, blank, r);
  return hipGetLastError() == hipSuccess ? ST5_OK : ST5_ERR_LAUNCH;
}

extern "C" int st5_ctc_prefix_score(const float* x, int T, int V, int blank, int eos, const float* r_prev, const int64_t* last,
                                    int out_len, const int64_t* cs, int nh, int nc, float* log_psi, float* r_new, void* stream) {
  if (!x || !r_prev || !last || !cs || !log_psi || !r_new) return ST5_ERR_ARG;
  if (T <= 0 || V <= 0 || nh <= 0 || nc <= 0 || nc > 1024 || out_len < 0 || out_len > T) return ST5_ERR_ARG;
  if (blank < 0 || blank >= V || eos < 0 || eos >= V) return ST5_ERR_ARG;
  const int threads = (nc + 63) / 64 * 64;
  hipLaunchKernelGGL(ctc_prefix_score_kernel, dim3(nh), dim3(threads), 0, reinterpret_cast<hipStream_t>(stream), x, T, V, blank, eos,
                     r_prev, reinterpret_cast<const long long*>(last), out_len, reinterpret_cast<const long long*>(cs), nc, log_psi,
                     r_new);
  return hipGetLastError() == hipSuccess ? ST5_OK : ST5_ERR_LAUNCH;
}
