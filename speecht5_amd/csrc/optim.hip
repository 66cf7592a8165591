// Fused optimizer step over the flat fp32 parameter / gradient buffers:
// Adam(beta1, beta2, eps) with decoupled weight decay as fairseq's `adam` (p -= lr*wd*p), global-norm
// gradient clipping and the 1/(W*U) gradient scale folded into one pass.  The squared gradient norm is
// read from device memory so that the step needs no host synchronisation.
// HBM-bound: reads p, g, m, v and writes p, m, v once (28 B per parameter).
#include "common.h"
#include "../../include/speecht5_hip.h"

namespace {
// hyper: optional device array {lr, step} that overrides the by-value lr / step count (a captured HIP graph replays
// constant kernel arguments; the host refreshes these two floats before every replay)
// g2 (optional): second gradient buffer, the step uses g + g2 (two micro-batches accumulated side by side, ddp.py);
// zero_grads: both buffers are left zeroed (saves the two fill launches of the next zero_grad()).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ g2, int zero_grads,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   bf16_t* __restrict__ mirror, long long n,
                                                   float lr, float b1, float b2, float eps, float wd, float t,
                                                   const float* __restrict__ gnorm_sq, float max_norm,
                                                   float gscale, const float* __restrict__ hyper) {
  if (hyper) { lr = hyper[0]; t = hyper[1]; }
  // bias corrections from the step count ON THE DEVICE in both forms: the by-value (eager) and the device-hyper (graph replay)
  // updates are then the same arithmetic, bit for bit (host pow() and device powf() differ in the last place for some t)
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  float coef = gscale;
  if (gnorm_sq && max_norm > 0.f) {
    const float norm = sqrtf(gnorm_sq[0]) * gscale;
    const float c = max_norm / (norm + 1e-6f);
    coef *= c < 1.f ? c : 1.f;
  }
  // fairseq/optim/adam.py: denom = sqrt(v) + eps (eps is NOT bias-corrected), step_size = lr * sqrt(bc2) / bc1
  const float step = lr * sqrtf(bc2) / bc1;
  const long long nv = n / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
    if (g2) {
      const f32x4 g2v = reinterpret_cast<const f32x4*>(g2)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) gg[e] += g2v[e];
    }
    if (zero_grads) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      reinterpret_cast<f32x4*>(g)[i] = z;
      if (g2) reinterpret_cast<f32x4*>(g2)[i] = z;
    }
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = gg[e] * coef;
      mm[e] = b1 * mm[e] + (1.f - b1) * ge;
      vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
      const float denom = sqrtf(vv[e]) + eps;
      pp[e] = pp[e] * (1.f - lr * wd) - step * mm[e] / denom;
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (mirror) {   // compute-dtype copy of the updated parameters (the next forward's weights: no cast kernels)
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16_t)pp[e];
      reinterpret_cast<bf16x4*>(mirror)[i] = o;
    }
  }
  for (long long i = nv * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float ge = (g[i] + (g2 ? g2[i] : 0.f)) * coef;
    if (zero_grads) { g[i] = 0.f; if (g2) g2[i] = 0.f; }
    const float mi = b1 * m[i] + (1.f - b1) * ge;
    const float vi = b2 * v[i] + (1.f - b2) * ge * ge;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] * (1.f - lr * wd) - step * mi / (sqrtf(vi) + eps);
    if (mirror) mirror[i] = (bf16_t)p[i];
  }
}

// Batched bf16 transposes: job j turns src[j] ([rows, cols], row-major) into dst[j] ([cols, rows]).  One launch per
// optimizer step for every transposed weight copy the data-gradient GEMMs use (instead of one launch per weight).
struct TrJob { long long src, dst; int rows, cols, tile0, pad; };   // element offsets into the two flat buffers
__global__ __launch_bounds__(256) void multi_transpose_kernel(const bf16_t* __restrict__ sflat, bf16_t* __restrict__ dflat,
                                                              const TrJob* __restrict__ jobs, int njobs) {
  __shared__ bf16_t tile[64][66];
  // find the job of this 64x64 tile (jobs are sorted by tile0; njobs is a few hundred: binary search)
  int lo = 0, hi = njobs - 1;
  const int t = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].tile0 <= t) lo = mid; else hi = mid - 1; }
  const TrJob jb = jobs[lo];
  const int tc = (jb.cols + 63) / 64;
  const int lt = t - jb.tile0, r0 = (lt / tc) * 64, c0 = (lt % tc) * 64;
  const bf16_t* src = sflat + jb.src;
  bf16_t* dst = dflat + jb.dst;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    const int r = r0 + j, c = c0 + tx;
    if (r < jb.rows && c < jb.cols) tile[j][tx] = src[(long long)r * jb.cols + c];
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    const int c = c0 + j, r = r0 + tx;
    if (r < jb.rows && c < jb.cols) dst[(long long)c * jb.rows + r] = tile[tx][j];
  }
}
}  // namespace

extern "C" int st5_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, int32_t step, const float* gnorm_sq, float max_norm, float grad_scale,
                                 void* bf16_mirror, const float* hyper_dev, void* stream);
extern "C" int st5_adam_step_pair(float* p, float* g, float* g2, int32_t zero_grads, float* m, float* v, int64_t n, float lr, float beta1,
                                  float beta2, float eps, float weight_decay, int32_t step, const float* gnorm_sq, float max_norm,
                                  float grad_scale, void* bf16_mirror, const float* hyper_dev, void* stream);

extern "C" int st5_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int32_t step, const float* gnorm_sq,
                             float max_norm, float grad_scale, void* bf16_mirror, void* stream) {
  return st5_adam_step_dev(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, gnorm_sq, max_norm, grad_scale, bf16_mirror, nullptr,
                           stream);
}

extern "C" int st5_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, int32_t step, const float* gnorm_sq, float max_norm, float grad_scale,
                                 void* bf16_mirror, const float* hyper_dev, void* stream) {
  return st5_adam_step_pair(p, const_cast<float*>(g), nullptr, 0, m, v, n, lr, beta1, beta2, eps, weight_decay, step, gnorm_sq, max_norm,
                            grad_scale, bf16_mirror, hyper_dev, stream);
}

extern "C" int st5_adam_step_pair(float* p, float* g, float* g2, int32_t zero_grads, float* m, float* v, int64_t n, float lr, float beta1,
                                  float beta2, float eps, float weight_decay, int32_t step, const float* gnorm_sq, float max_norm,
                                  float grad_scale, void* bf16_mirror, const float* hyper_dev, void* stream) {
  if (!p || !g || !m || !v || n < 0 || (step < 1 && !hyper_dev)) return ST5_ERR_ARG;
  if (n == 0) return ST5_OK;
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, g2, zero_grads, m, v, (bf16_t*)bf16_mirror,
                     (long long)n,
                     lr, beta1, beta2, eps, weight_decay, (float)step, gnorm_sq, max_norm, grad_scale, hyper_dev);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_multi_transpose_bf16(const void* src_flat, void* dst_flat, const void* jobs_dev, int32_t njobs,
                                        int32_t ntiles, void* stream) {
  if (!src_flat || !dst_flat || !jobs_dev || njobs <= 0 || ntiles <= 0) return ST5_ERR_ARG;
  hipLaunchKernelGGL(multi_transpose_kernel, dim3((unsigned)ntiles), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src_flat,
                     (bf16_t*)dst_flat, (const TrJob*)jobs_dev, njobs);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
