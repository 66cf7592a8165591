// Attention probabilities with Shaw-style relative-position bias, masks and dropout.
// Follows SpeechT5/speecht5/models/modules/multihead_attention.py:343-386:
//   attn_weights = q.k^T (+ q.pe_k[clip(i-j)]^T) (+ attn_mask) ; key padding -> -inf ;
//   softmax in fp32 ; dropout.
// The reference materialises pos_k [T,T,64] (encoder.py:240-244) and the bias tensor [BH,T,T];
// here the bias is gathered from QP = q.pe^T [BH,T,2*maxrel] (one small GEMM) inside the softmax.
// One wave per (bh, i) row; fp32 math; 16-byte vector IO.  HBM-bound.
#include "common.h"
#include "../../include/speecht5_hip.h"

namespace {

constexpr int MAXCH = 8;  // S <= 4096

__device__ __forceinline__ int bucket(int i, int j, int maxrel) {
  int d = i - j;
  d = d < -maxrel ? -maxrel : (d > maxrel - 1 ? maxrel - 1 : d);
  return d + maxrel;
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const T* __restrict__ scores, const T* __restrict__ qp,
                                                          const uint8_t* __restrict__ kpm, T* __restrict__ probs,
                                                          T* __restrict__ probs_drop, int BH, int H, int Tq, int S,
                                                          int lds, int nb, int maxrel, int causal, float dropout_p,
                                                          unsigned long long seed) {
  const int lane = threadIdx.x & 63;
  const long long rowid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rowid >= (long long)BH * Tq) return;
  const int bh = (int)(rowid / Tq), i = (int)(rowid % Tq);
  const int b = bh / H;
  const T* srow = scores + rowid * lds;
  const T* qrow = qp ? qp + rowid * nb : nullptr;
  const uint8_t* mrow = kpm ? kpm + (long long)b * S : nullptr;
  // causal offset: query i may attend keys j <= i + (S - Tq) (incremental decoding has S >= Tq)
  const int jmax = causal ? i + (S - Tq) : S - 1;
  float v[NCH][8];
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int j0 = c * 512 + lane * 8;
    if (j0 < lds) {
      load8f<T>(srow + j0, v[c]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = j0 + e;
        float x = v[c][e];
        if (j >= S || j > jmax || (mrow && mrow[j])) x = -INFINITY;
        else if (qrow) x += Elem<T>::to_f(qrow[bucket(i, j, maxrel)]);
        v[c][e] = x;
        mx = fmaxf(mx, x);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c][e] = -INFINITY;
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float p = (mx == -INFINITY) ? 0.f : __expf(v[c][e] - mx);
      v[c][e] = p;
      sum += p;
    }
  sum = wave_sum(sum);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;  // fully-masked row -> zeros (reference would give NaN)
  const unsigned int thresh = dropout_p > 0.f ? dropout_thresh(dropout_p) : 0u;
  const float inv_keep = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int j0 = c * 512 + lane * 8;
    if (j0 < lds) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c][e] *= inv;
      store8f<T>(probs + rowid * lds + j0, v[c]);
      if (probs_drop) {
        float d[8];
        dropout_scale8(seed, (unsigned long long)(rowid * drop_row_stride(lds) + j0), thresh, inv_keep, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] *= v[c][e];
        store8f<T>(probs_drop + rowid * lds + j0, d);
      }
    }
  }
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(T* __restrict__ dP, const T* __restrict__ probs,
                                                          const float* __restrict__ dP_extra, T* __restrict__ dqp,
                                                          int BH, int Tq, int S, int lds, int nb, int maxrel,
                                                          float dropout_p, unsigned long long seed) {
  extern __shared__ float sh[];  // 4 waves x nb floats
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long rowid = (long long)blockIdx.x * 4 + wave;
  const bool active = rowid < (long long)BH * Tq;
  float* acc = sh + wave * nb;
  if (dqp) {
    for (int k = lane; k < nb; k += 64) acc[k] = 0.f;
  }
  __syncthreads();
  if (active) {
    const int i = (int)(rowid % Tq);
    const unsigned int thresh = dropout_p > 0.f ? dropout_thresh(dropout_p) : 0u;
    const float inv_keep = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
    float g[NCH][8], p[NCH][8];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int j0 = c * 512 + lane * 8;
      if (j0 < lds) {
        load8f<T>(dP + rowid * lds + j0, g[c]);
        load8f<T>(probs + rowid * lds + j0, p[c]);
        float dsc[8];
        if (dropout_p > 0.f) dropout_scale8(seed, (unsigned long long)(rowid * drop_row_stride(lds) + j0), thresh, inv_keep, dsc);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float x = g[c][e];
          if (dropout_p > 0.f) x *= dsc[e];
          if (dP_extra && j0 + e < S) x += dP_extra[rowid * S + j0 + e];
          if (j0 + e >= S) x = 0.f;
          g[c][e] = x;
          dot += x * p[c][e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { g[c][e] = 0.f; p[c][e] = 0.f; }
      }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int j0 = c * 512 + lane * 8;
      if (j0 < lds) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float ds = p[c][e] * (g[c][e] - dot);
          g[c][e] = ds;
          if (dqp && j0 + e < S && ds != 0.f) atomicAdd(&acc[bucket(i, j0 + e, maxrel)], ds);
        }
        store8f<T>(dP + rowid * lds + j0, g[c]);
      }
    }
  }
  __syncthreads();
  if (active && dqp) {
    for (int k = lane; k < nb; k += 64) dqp[rowid * nb + k] = Elem<T>::from_f(acc[k]);
  }
}

}  // namespace

extern "C" int st5_softmax_fwd(const void* scores, const void* qp, const uint8_t* kpm, void* probs, void* probs_drop,
                               int32_t BH, int32_t H, int32_t T, int32_t S, int32_t lds, int32_t nb, int32_t maxrel,
                               int32_t causal, float dropout_p, uint64_t seed, int dtype, void* stream) {
  if (!scores || !probs || BH <= 0 || T <= 0 || S <= 0 || lds < S || lds % 8 || lds > MAXCH * 512 || H <= 0)
    return ST5_ERR_ARG;
  if (qp && (nb != 2 * maxrel)) return ST5_ERR_ARG;
  if (dropout_p > 0.f && !probs_drop) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long rows = (long long)BH * T;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype != ST5_BF16 && dtype != ST5_F32) return ST5_ERR_ARG;
#define SMF(TT, NCH)                                                                                                  \
  hipLaunchKernelGGL((softmax_fwd_kernel<TT, NCH>), grid, dim3(256), 0, s, (const TT*)scores, (const TT*)qp, kpm,     \
                     (TT*)probs, (TT*)probs_drop, BH, H, T, S, lds, nb, maxrel, causal, dropout_p, (unsigned long long)seed)
#define SMF_T(TT)                                                                                                     \
  do {                                                                                                                \
    if (lds <= 512) SMF(TT, 1); else if (lds <= 1024) SMF(TT, 2); else if (lds <= 2048) SMF(TT, 4); else SMF(TT, 8);  \
  } while (0)
  if (dtype == ST5_BF16) SMF_T(bf16_t); else SMF_T(float);
#undef SMF_T
#undef SMF
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_softmax_bwd(void* dP_inout, const void* probs, const float* dP_extra, void* dqp, int32_t BH,
                               int32_t T, int32_t S, int32_t lds, int32_t nb, int32_t maxrel, float dropout_p,
                               uint64_t seed, int dtype, void* stream) {
  if (!dP_inout || !probs || BH <= 0 || T <= 0 || S <= 0 || lds < S || lds % 8 || lds > MAXCH * 512) return ST5_ERR_ARG;
  if (dqp && (nb != 2 * maxrel || nb <= 0)) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long rows = (long long)BH * T;
  dim3 grid((unsigned)((rows + 3) / 4));
  const size_t shm = dqp ? (size_t)4 * nb * sizeof(float) : 16;
  if (dtype != ST5_BF16 && dtype != ST5_F32) return ST5_ERR_ARG;
#define SMB(TT, NCH)                                                                                               \
  hipLaunchKernelGGL((softmax_bwd_kernel<TT, NCH>), grid, dim3(256), shm, s, (TT*)dP_inout, (const TT*)probs, dP_extra, \
                     (TT*)dqp, BH, T, S, lds, nb, maxrel, dropout_p, (unsigned long long)seed)
#define SMB_T(TT)                                                                                                  \
  do {                                                                                                             \
    if (lds <= 512) SMB(TT, 1); else if (lds <= 1024) SMB(TT, 2); else if (lds <= 2048) SMB(TT, 4); else SMB(TT, 8); \
  } while (0)
  if (dtype == ST5_BF16) SMB_T(bf16_t); else SMB_T(float);
#undef SMB_T
#undef SMB
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
