// Criterion arithmetic of the speech decoder as single passes.
//
// Tacotron2Loss with masking (SpeechT5/speecht5/criterions/text_to_speech_loss.py:263-345, called from :186-216):
//   olens' = olens - olens % r ;  labels[b, olens'_b - 1] = 1
//   l1  = (sum_valid |after - y| + sum_valid |before - y|) / n_elem        (two nn.L1Loss means over masked_select)
//   mse = (sum_valid (after - y)^2 + sum_valid (before - y)^2) / n_elem
//   bce = sum_valid BCEWithLogits(logit, label, pos_weight) / n_frames
// valid = frames t < olens'_b; n_frames = sum_b olens'_b, n_elem = n_frames * C.  The reference runs ~45 element-wise /
// reduction kernels and a masked_select (host sync) for this; here: one reduction pass + one finalize, and one pass for
// the three input gradients.  Deterministic (block partials in a fixed order, fp64 final sum).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "../../include/speecht5_hip.h"

namespace {

constexpr int TL_BLOCKS = 256;

__device__ __forceinline__ long long eff_len(const long long* olens, int b, int r, int L) {
  long long o = olens[b];
  o -= o % r;
  return o < 0 ? 0 : (o > L ? L : o);
}

// F.binary_cross_entropy_with_logits(x, y, pos_weight): (1 - y) x + (1 + (pw - 1) y) (log1p(exp(-|x|)) + max(-x, 0))
__device__ __forceinline__ float bce_logits(float x, float y, float pw) {
  const float lw = 1.f + (pw - 1.f) * y;
  return (1.f - y) * x + lw * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f));
}

__global__ __launch_bounds__(256) void tacotron_loss_partial_kernel(const float* __restrict__ after, const float* __restrict__ before,
                                                                    const float* __restrict__ logits, const float* __restrict__ ys,
                                                                    long long ys_bs, const float* __restrict__ labels, long long lab_bs,
                                                                    const long long* __restrict__ olens, int B, int L, int C, int r,
                                                                    float pw, float* __restrict__ part) {
  __shared__ float red[4][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s1 = 0.f, s2 = 0.f, sb = 0.f;
  const long long rows = (long long)B * L;
  for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
    const int b = (int)(row / L), t = (int)(row % L);
    const long long ol = eff_len(olens, b, r, L);
    if (t >= ol) continue;
    const float* a = after + row * C;
    const float* bf = before + row * C;
    const float* y = ys + (long long)b * ys_bs + (long long)t * C;
    for (int c = lane; c < C; c += 64) {
      const float da = a[c] - y[c], db = bf[c] - y[c];
      s1 += fabsf(da) + fabsf(db);
      s2 = fmaf(da, da, fmaf(db, db, s2));
    }
    if (lane == 0) {
      const float lab = (r > 1 && t == ol - 1) ? 1.f : labels[(long long)b * lab_bs + t];   // (:168: only when r > 1)
      sb += bce_logits(logits[row], lab, pw);
    }
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2); sb = wave_sum(sb);
  if (lane == 0) { red[wave][0] = s1; red[wave][1] = s2; red[wave][2] = sb; }
  __syncthreads();
  if (threadIdx.x < 3) part[blockIdx.x * 3 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out = {l1, mse, bce, n_frames}
__global__ __launch_bounds__(64) void tacotron_loss_final_kernel(const float* __restrict__ part, int nblk, const long long* __restrict__ olens,
                                                                 int B, int L, int C, int r, float* __restrict__ out) {
  const int lane = threadIdx.x;
  double s[3] = {0.0, 0.0, 0.0};
  for (int i = lane; i < nblk; i += 64)
    for (int k = 0; k < 3; ++k) s[k] += (double)part[i * 3 + k];
  double nf = 0.0;
  for (int b = lane; b < B; b += 64) nf += (double)eff_len(olens, b, r, L);
  for (int o = 32; o > 0; o >>= 1) {
    for (int k = 0; k < 3; ++k) s[k] += __shfl_xor(s[k], o, 64);
    nf += __shfl_xor(nf, o, 64);
  }
  if (lane == 0) {
    const double ne = nf * (double)C;
    out[0] = (float)(s[0] / ne);
    out[1] = (float)(s[1] / ne);
    out[2] = (float)(s[2] / nf);
    out[3] = (float)nf;
  }
}

// d_after = m (g1 sign(da) + 2 g2 da) / n_elem ; d_before likewise ; d_logit = m g3 (sigma(x) (1 - y + pw y) - pw y) / n_frames
__global__ __launch_bounds__(256) void tacotron_loss_bwd_kernel(const float* __restrict__ after, const float* __restrict__ before,
                                                                const float* __restrict__ logits, const float* __restrict__ ys, long long ys_bs,
                                                                const float* __restrict__ labels, long long lab_bs,
                                                                const long long* __restrict__ olens, int B, int L, int C, int r, float pw,
                                                                const float* __restrict__ stats, const float* __restrict__ g1p,
                                                                const float* __restrict__ g2p, const float* __restrict__ g3p,
                                                                float* __restrict__ d_after, float* __restrict__ d_before,
                                                                float* __restrict__ d_logits) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float nf = stats[3], ne = nf * (float)C;
  const float g1 = (g1p ? g1p[0] : 0.f) / ne, g2 = 2.f * (g2p ? g2p[0] : 0.f) / ne, g3 = (g3p ? g3p[0] : 0.f) / nf;
  const long long rows = (long long)B * L;
  for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
    const int b = (int)(row / L), t = (int)(row % L);
    const long long ol = eff_len(olens, b, r, L);
    const bool valid = t < ol;
    const float* y = ys + (long long)b * ys_bs + (long long)t * C;
    for (int c = lane; c < C; c += 64) {
      float ga = 0.f, gb = 0.f;
      if (valid) {
        const float da = after[row * C + c] - y[c], db = before[row * C + c] - y[c];
        ga = g1 * (da > 0.f ? 1.f : (da < 0.f ? -1.f : 0.f)) + g2 * da;
        gb = g1 * (db > 0.f ? 1.f : (db < 0.f ? -1.f : 0.f)) + g2 * db;
      }
      if (d_after) d_after[row * C + c] = ga;
      if (d_before) d_before[row * C + c] = gb;
    }
    if (lane == 0 && d_logits) {
      float gl = 0.f;
      if (valid) {
        const float lab = (r > 1 && t == ol - 1) ? 1.f : labels[(long long)b * lab_bs + t];   // (:168: only when r > 1)
        const float x = logits[row];
        const float sg = 1.f / (1.f + expf(-x));
        gl = g3 * (sg * (1.f - lab + pw * lab) - pw * lab);
      }
      d_logits[row] = gl;
    }
  }
}

}  // namespace

extern "C" int64_t st5_tacotron_loss_ws_bytes(void) { return (int64_t)TL_BLOCKS * 3 * sizeof(float); }

extern "C" int st5_tacotron_loss_fwd(const float* after, const float* before, const float* logits, const float* ys, int64_t ys_bstride,
                                     const float* labels, int64_t labels_bstride, const int64_t* olens, int32_t B, int32_t L, int32_t C,
                                     int32_t r, float pos_weight, float* out4, void* ws, void* stream) {
  if (!after || !before || !logits || !ys || !labels || !olens || !out4 || !ws || B <= 0 || L <= 0 || C <= 0 || r < 1) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  long long nb = ((long long)B * L + 3) / 4;
  if (nb > TL_BLOCKS) nb = TL_BLOCKS;
  hipLaunchKernelGGL(tacotron_loss_partial_kernel, dim3((unsigned)nb), dim3(256), 0, s, after, before, logits, ys, (long long)ys_bstride, labels,
                     (long long)labels_bstride, reinterpret_cast<const long long*>(olens), B, L, C, r, pos_weight, (float*)ws);
  hipLaunchKernelGGL(tacotron_loss_final_kernel, dim3(1), dim3(64), 0, s, (const float*)ws, (int)nb, reinterpret_cast<const long long*>(olens), B, L,
                     C, r, out4);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_tacotron_loss_bwd(const float* after, const float* before, const float* logits, const float* ys, int64_t ys_bstride,
                                     const float* labels, int64_t labels_bstride, const int64_t* olens, int32_t B, int32_t L, int32_t C,
                                     int32_t r, float pos_weight, const float* out4, const float* g_l1, const float* g_mse, const float* g_bce,
                                     float* d_after, float* d_before, float* d_logits, void* stream) {
  if (!after || !before || !logits || !ys || !labels || !olens || !out4 || B <= 0 || L <= 0 || C <= 0 || r < 1) return ST5_ERR_ARG;
  long long nb = ((long long)B * L + 3) / 4;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(tacotron_loss_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, after, before, logits, ys,
                     (long long)ys_bstride, labels, (long long)labels_bstride, reinterpret_cast<const long long*>(olens), B, L, C, r, pos_weight,
                     out4, g_l1, g_mse, g_bce, d_after, d_before, d_logits);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
