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

// The block's samples -> LDS.  All of a thread's loads are issued before the first store (a plain `for` with the bounds test inside
// made every iteration wait for its own global round trip: 6-10 sequential latencies at the head of every block).
__device__ __forceinline__ void stage_wav(float* seg, const float* __restrict__ wav, int b, int S, int t0, int nt, int k,
                                          int stride) {
  const int nseg = (nt - 1) * stride + k, ntot = nseg + MAXK;
  const float* src = wav + (long long)b * S + (long long)t0 * stride;
  for (int base = 0; base < ntot; base += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256 + threadIdx.x;
      v[u] = i < nseg ? src[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256 + threadIdx.x;
      if (i < ntot) seg[i] = v[u];
    }
  }
}

// ---- waveform moments and GroupNorm statistics in TWO launches (round 5; three launches of 22 + 14 + 8 us before, for 5 MB of input) --
// conv0_moments2_kernel: a block owns TCM = 1024 time steps, a thread 4 of them -- the 65 moment products are accumulated in registers
// over the thread's steps and reduced ONCE per block (the first form reduced 65 values per 256 steps).  part2[b][chunk][nmom].
// conv0_stats2_kernel: every block of (64 channels, clip) first folds the clip's chunk partials in fp64 (fixed order), block 0 of a
// clip also publishes them (the backward's final kernel reads them), then 64 threads evaluate mean / rstd of their channels.
constexpr int TCM = 1024;
template <int KW>
__global__ __launch_bounds__(256) void conv0_moments2_kernel(const float* __restrict__ wav, float* __restrict__ part, int S, int L, int k,
                                                             int stride, int nchm) {
  __shared__ float red[4][MAXMOM];
  extern __shared__ __attribute__((aligned(16))) float segm[];      // the block's samples: (TCM - 1) stride + k floats (+ alignment slack)
  const int b = blockIdx.y, ch = blockIdx.x;
  const int t0 = ch * TCM, nt = min(TCM, L - t0);
  {
    // coalesced 16-byte loads from the 16-byte-aligned address at or below the block's first sample
    const long long g0 = (long long)b * S + (long long)t0 * stride;
    const long long ga = g0 & ~3ll;
    const int shift = (int)(g0 - ga), nneed = shift + (nt - 1) * stride + k;
    const long long gend = (long long)(b + 1) * S;          // (never read past the clip: the next clip / the buffer's end)
    const float4* src = reinterpret_cast<const float4*>(wav + ga);
    const bool al16 = (reinterpret_cast<uintptr_t>(wav) & 15) == 0;
    for (int base = 0; base * 4 < nneed; base += 256 * 4) {      // (four 16-byte loads in flight per thread)
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * 256 + threadIdx.x;
        const long long p = ga + 4ll * i;
        if (i * 4 < nneed && al16 && p + 3 < gend) v[u] = src[i];
        else {
          v[u].x = (i * 4 < nneed && p < gend) ? wav[p] : 0.f; v[u].y = (i * 4 < nneed && p + 1 < gend) ? wav[p + 1] : 0.f;
          v[u].z = (i * 4 < nneed && p + 2 < gend) ? wav[p + 2] : 0.f; v[u].w = (i * 4 < nneed && p + 3 < gend) ? wav[p + 3] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * 256 + threadIdx.x;
        if (i * 4 < nneed) reinterpret_cast<float4*>(segm)[i] = v[u];
      }
    }
    __syncthreads();
  }
  const float* wb = segm + (int)(((long long)b * S + (long long)t0 * stride) & 3ll);
  float m1[KW], m2[KW * (KW + 1) / 2];
#pragma unroll
  for (int j = 0; j < KW; ++j) m1[j] = 0.f;
#pragma unroll
  for (int j = 0; j < KW * (KW + 1) / 2; ++j) m2[j] = 0.f;
  for (int t = threadIdx.x; t < nt; t += 256) {
    float x[KW];
#pragma unroll
    for (int j = 0; j < KW; ++j) x[j] = j < k ? wb[t * stride + j] : 0.f;
    int q = 0;
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      m1[j] += x[j];
#pragma unroll
      for (int jp = j; jp < KW; ++jp) { m2[q] = fmaf(x[j], x[jp], m2[q]); ++q; }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  {
    int q = 0;
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      const float s = wave_sum(m1[j]);
      if (lane == 0 && j < k) red[wv][j] = s;
#pragma unroll
      for (int jp = j; jp < KW; ++jp) {
        const float r = wave_sum(m2[q]); ++q;
        if (lane == 0 && jp < k) red[wv][ridx(k, j, jp)] = r;
      }
    }
  }
  __syncthreads();
  const int nm = nmom(k);
  for (int i = threadIdx.x; i < nm; i += 256)
    part[((long long)b * nchm + ch) * nm + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}
// the two halves of conv0_stats2_kernel as device functions (conv0_stats_wfrag_kernel below runs them too)
// chunk partials -> fp64 sums: two 128-lane groups take every second chunk, sixteen chunks per round (sixteen loads in flight per thread:
// with one -- a runtime-trip-count loop -- the fold was nchm / 4 dependent L2 round trips), folded in a fixed order.  The first round's
// loads can be issued early (fold_prefetch) so that they fly together with whatever else the caller loads.
struct FoldPre { float v[16]; };
__device__ __forceinline__ void fold_prefetch(const float* __restrict__ part, int b, int nm, int nchm, FoldPre& f) {
  const int grp = threadIdx.x >> 7, m = threadIdx.x & 127;
#pragma unroll
  for (int u = 0; u < 16; ++u) f.v[u] = (m < nm && grp + 2 * u < nchm) ? part[((long long)b * nchm + grp + 2 * u) * nm + m] : 0.f;
}
__device__ __forceinline__ void fold_chunk_moments(const float* __restrict__ part, double* __restrict__ mom, double* ms, double (*ps)[MAXMOM],
                                                   int b, int nm, int nchm, bool publish, const FoldPre* pre) {
  const int grp = threadIdx.x >> 7, ml = threadIdx.x & 127;
  for (int base = 0; base < nm; base += 128) {
    const int m = base + ml;
    double sacc = 0.0;
    if (m < nm)
      for (int ch0 = grp; ch0 < nchm; ch0 += 32) {
        float v[16];
        if (pre && base == 0 && ch0 == grp) {
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = pre->v[u];
        } else {
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = ch0 + 2 * u < nchm ? part[((long long)b * nchm + ch0 + 2 * u) * nm + m] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (ch0 + 2 * u < nchm) sacc += (double)v[u];
      }
    if (m < nm) ps[grp][m] = sacc;
  }
  __syncthreads();
  for (int m = threadIdx.x; m < nm; m += 256) {
    const double sacc = ps[0][m] + ps[1][m];
    ms[m] = sacc;
    if (publish) mom[(long long)b * nm + m] = sacc;
  }
  __syncthreads();
}
// mean and 1 / sqrt(var + eps) of one channel's convolution output from the waveform moments M_j = sum_t x_{t+j}, M_jj' = sum_t x_{t+j}
// x_{t+j'}: s1 = sum_j w_j M_j, s2 = sum_j w_j (M_jj w_j + 2 sum_{j' > j} M_jj' w_j'), in fp64 with EXPLICIT fused multiply-adds in a
// fixed order -- the two callers (taps from memory / from registers) round alike, so the forward's saved statistics are the same bits
// whichever launch form produced them.  tap(j): tap j as a float, j < k.
template <int KW, typename Tap>
__device__ __forceinline__ void channel_stats(Tap tap, const double* ms, int k, int L, float eps, float& mean, float& rstd) {
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int j = 0; j < KW; ++j) {
    if (j < k) {
      const double wj = (double)tap(j);
      s1 = fma(wj, ms[j], s1);
      double off = 0.0;
#pragma unroll
      for (int jp = j + 1; jp < KW; ++jp)
        if (jp < k) off = fma(ms[ridx(k, j, jp)], (double)tap(jp), off);
      s2 = fma(wj, fma(ms[ridx(k, j, j)], wj, 2.0 * off), s2);
    }
  }
  const double mu = s1 / L;
  double var = fma(-mu, mu, s2 / L);
  if (var < 0.0) var = 0.0;
  mean = (float)mu;
  rstd = (float)(1.0 / sqrt(var + (double)eps));
}
__global__ __launch_bounds__(256) void conv0_stats2_kernel(const float* __restrict__ part, double* __restrict__ mom, const float* __restrict__ w,
                                                           float* __restrict__ stats, int C, int k, int L, int nchm, float eps) {
  __shared__ double ms[MAXMOM];
  __shared__ double ps[2][MAXMOM];
  const int b = blockIdx.y, nm = nmom(k);
  fold_chunk_moments(part, mom, ms, ps, b, nm, nchm, blockIdx.x == 0, nullptr);
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (!stats || threadIdx.x >= 64 || c >= C) return;
  float mu, rs;
  const float* wc = w + c * k;
  if (k <= 10) channel_stats<10>([&](int j) { return wc[j]; }, ms, k, L, eps, mu, rs);
  else channel_stats<MAXK>([&](int j) { return wc[j]; }, ms, k, L, eps, mu, rs);
  stats[((long long)b * C + c) * 2 + 0] = mu;
  stats[((long long)b * C + c) * 2 + 1] = rs;
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
  constexpr bool FAST = sizeof(T) == 2;   // bf16 output: transcendental-free GELU (common.h gelu_poly), fp32 parity mode: libm erff
  for (int t = tli; t < nt; t += g.tl) {
    float y[8];
    conv8<KW>(seg + t * stride, wr, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float z = fmaf(y[e], sc[e], sh[e]);
      y[e] = FAST ? gelu_poly(z) : gelu_f(z);
    }
    store8f<T>(out + ((long long)b * L + t0 + t) * C + c0, y);
  }
}

// ---- forward apply on the matrix cores (bf16 output, k <= 10, C % 32 == 0) --------------------------------------------------------
// The VALU form above spends ~31 issue slots per output element (10 FMAs of the convolution, the affine, ~21 of the erf-GELU whose rcp
// and exp issue at quarter rate, the pack): 131 M outputs per 8-clip batch = 100 us of pure VALU issue on 256 CUs, against 33 us to
// WRITE the 262 MB at 8 TB/s -- it was VALU-bound (measured 110-117 us = 2.3 TB/s).  Here the convolution AND the GroupNorm affine are
// one GEMM on v_mfma_f32_32x32x16_bf16 with SPLIT operands:
//   x = xh + xl, sc w = ah + al (bf16 each, xl / al the rounding residues, sc = rstd gamma)     z ~= ah.xh + ah.xl + al.xh + sh
// laid out along K as [ah | ah | al | shh shl] . [xh ; xl ; xh ; 1 1] -- all 32 k-slots of TWO MFMAs per 32 channels x 32 time steps
// (the shift sh = beta - mean sc rides in the two slots the 3 x 10 taps leave free), the fp32 accumulator carrying ~16 mantissa bits
// of every product (the recipe's own fp16 autocast keeps 11; the dropped al.xl is ~2^-16 |w||x|).  The channel <-> MFMA row mapping is
// PERMUTED (c0_perm) so that a lane's 16 accumulator registers are 16 CONSECUTIVE channels of one time step: the channels-last store
// is two 16-byte stores per lane, no LDS transpose.  What is left per output element is the transcendental-free GELU (common.h
// gelu_poly) and the pack: ~15 issue slots.  The fragments depend on the clip (its statistics): conv0_wfrag_kernel builds them once per
// call (B x 32 KB for 512 channels); a block keeps its clip's 32 KB in LDS.
__device__ __forceinline__ int c0_perm(int m) { return 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3); }

// one 16-byte fragment entry: afrag[.][tile mt][q][lane l] -- rows hold sc w (sc = rstd gamma, split in bf16 high / low parts) and the
// two spare k slots (30, 31) hold the shift sh = beta - mean sc (high / low part) against ones on the x side: the MFMA pair returns
// z = sc (w . x) + sh, the GELU argument, directly
template <typename W>
__device__ __forceinline__ bf16x8 wfrag_entry(W wj, int k, float sc, float sh, int q, int hi) {   // wj(j): tap j of the entry's channel
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int kk = 16 * q + 8 * hi + e, sec = kk / 10, j = kk - 10 * sec;
    float v = (kk < 30 && j < k) ? sc * wj(j) : 0.f;
    if (kk >= 30) v = sh;
    const bf16_t vh = (bf16_t)v;
    const bool low = sec == 2 && kk < 30 || kk == 31;
    o[e] = low ? (bf16_t)(v - (float)vh) : vh;
  }
  return o;
}
__global__ __launch_bounds__(128) void conv0_wfrag_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ stats, bf16x8* __restrict__ afrag, int C, int k) {
  // afrag[b][tile][2][64 lanes]: the GroupNorm affine is folded in
  const int mt = blockIdx.x, b = blockIdx.y, q = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int c = 32 * mt + c0_perm(l & 31), hi = l >> 5;
  const float mu = stats[((long long)b * C + c) * 2], rs = stats[((long long)b * C + c) * 2 + 1];
  const float sc = rs * gamma[c], sh = beta[c] - mu * sc;
  afrag[((long long)(b * gridDim.x + mt) * 2 + q) * 64 + l] = wfrag_entry([&](int j) { return w[c * k + j]; }, k, sc, sh, q, hi);
}

// The forward's statistics and fragments in ONE launch (conv0_stats2_kernel + conv0_wfrag_kernel were ~9 + ~5 us of launch and load
// latency for 65 x nchm floats per clip): a block of (32-channel tile, clip) folds the clip's moment partials, every thread derives the
// statistics of the channel its fragment entry belongs to (four threads per channel, same arithmetic -- no exchange), one of them
// publishes `stats`.  Same device functions as the two-launch form: same bits.  (Deriving them in the apply kernel's prologue instead
// was measured: 1000 blocks x 512 channels of fp64 pushed that kernel from 56 to 212 registers or, capped at 128, into spills --
// 94 us for the call against 87.)
// The backward calls it with the forward's statistics (stats_in): no statistics are derived, and only the block of tile 0 folds the
// moments -- to publish `mom` for conv0_bwd_final_kernel (conv0_stats2_kernel did only that there: one launch less).
__global__ __launch_bounds__(256) void conv0_stats_wfrag_kernel(const float* __restrict__ part, const float* __restrict__ w,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ stats_in, float* __restrict__ stats,
                                                                double* __restrict__ mom, bf16x8* __restrict__ afrag, int C, int k, int L,
                                                                int nchm, float eps) {
  __shared__ double ms[MAXMOM];
  __shared__ double ps[2][MAXMOM];
  const int mt = blockIdx.x, b = blockIdx.y, q = (threadIdx.x >> 6) & 1, l = threadIdx.x & 63;
  const int c = 32 * mt + c0_perm(l & 31), hi = l >> 5;
  float wr[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) wr[j] = j < k ? w[c * k + j] : 0.f;
  const float ga = gamma[c], be = beta[c];
  float mu = 0.f, rs = 0.f;
  if (stats_in) { mu = stats_in[((long long)b * C + c) * 2]; rs = stats_in[((long long)b * C + c) * 2 + 1]; }
  if (!stats_in || (mom && mt == 0)) {      // (block-uniform)
    FoldPre pre;
    fold_prefetch(part, b, nmom(k), nchm, pre);
    fold_chunk_moments(part, mom, ms, ps, b, nmom(k), nchm, mom && mt == 0, &pre);
  }
  if (threadIdx.x >= 128) return;
  if (!stats_in) {
    channel_stats<10>([&](int j) { return wr[j]; }, ms, k, L, eps, mu, rs);
    if (q == 0 && hi == 0) { stats[((long long)b * C + c) * 2] = mu; stats[((long long)b * C + c) * 2 + 1] = rs; }
  }
  const float sc = rs * ga, sh = be - mu * sc;
  afrag[((long long)(b * gridDim.x + mt) * 2 + q) * 64 + l] = wfrag_entry([&](int j) { return wr[j]; }, k, sc, sh, q, hi);
}

// GELU through a 256-entry table in LDS (TAB): with the polynomial the kernel is VALU-issue-bound -- 14.5 slots per output element, 131 M
// elements per 8-clip batch = 48 us of issue on 256 CUs against 33 us to write them.  gelu(z) = max(z, 0) - |z| Q(|z|), Q = 1 - Phi the
// upper tail on [0, 4.25] as chords over 255 intervals, an entry (Q, rise) two floats read with one ds_read_b64; beyond 4.25 the last
// value (1.1e-5).  Q is the SMALL quantity on both sides (z > 0: z - z Q; z < 0: -|z| Q): |gelu error| <= 8.4e-6 |z| (the chord error
// h^2 / 8 max|Phi''|; the polynomial: 1.5e-5 |z|), tests/test_isa_audit.py::test_conv0_gelu_table_error_bound.  8.5 slots: scale,
// clamp, truncate, fract, address, the chord, max, the product.  (Entries packed as two fp16 in one dword + v_fma_mix_f32 were
// measured too: the same 60 us -- bank conflicts are not what bounds the kernel -- at 13x the error, so the fp32 pairs stay.)
// The table costs 2 KB of LDS (four blocks per CU still fit) and 256 erfcf calls per block.
constexpr float C0_GT_MAX = 4.25f, C0_GT_SCALE = 255.f / C0_GT_MAX;
// What is left after the table (60 us): the write stream itself.  With the SAME bytes stored lane-consecutively (wrong placement, timing
// experiment tools/r6b/call31.sh) the kernel takes 51.9 us = 5.0 TB/s of pure writes -- the chip's write ceiling as far as this kernel
// can see it; the real pattern (a lane's 16-byte chunks lie 1 KB apart: 32 rows per store instruction) costs 7 us over that.
// (Written per element; hipcc keeps ~4 table reads in flight.  Staging all sixteen reads of a tile first was measured slower -- 62.8 us
//  against 59.0 -- on the same kind of A/B, profiles/r6b_conv0_fold_ab.txt.)
__device__ __forceinline__ float gelu_tab(float z, const float2* gt) {
  const float a = fminf(fabsf(z) * C0_GT_SCALE, 255.f);
  const float2 e = gt[(int)a];
  const float q = fmaf(__builtin_amdgcn_fractf(a), e.y, e.x);
  float relu;
  asm("v_max_f32 %0, 0, %1" : "=v"(relu) : "v"(z));      // (one slot; fmaxf adds a canonicalising v_max z, z in front)
  return fmaf(-fabsf(z), q, relu);
}
template <bool TAB>
__global__ __launch_bounds__(256) void conv0_apply_mfma_kernel(const float* __restrict__ wav, const bf16x8* __restrict__ afrag_g,
                                                               bf16_t* __restrict__ out, int S, int L, int C, int k, int stride, int seg_bytes) {
  extern __shared__ __attribute__((aligned(16))) char sm0[];
  const int ntile = C / 32;
  bf16x8* afr = reinterpret_cast<bf16x8*>(sm0);                       // [ntile][2][64 lanes] x 16 B (this clip's)
  float* seg = reinterpret_cast<float*>(sm0 + (size_t)ntile * 2048);
  const float2* gt = reinterpret_cast<const float2*>(sm0 + (size_t)ntile * 2048 + seg_bytes);   // [256] (TAB)
  if constexpr (TAB) {
    const float h = C0_GT_MAX / 255.f, x0 = threadIdx.x * h;
    const float v0 = 0.5f * erfcf(x0 * 0.70710678118654752f);
    const float v1 = threadIdx.x < 255 ? 0.5f * erfcf((x0 + h) * 0.70710678118654752f) : v0;
    const_cast<float2*>(gt)[threadIdx.x] = make_float2(v0, v1 - v0);
  }
  const int b = blockIdx.y, t0 = blockIdx.x * TCH;
  const int nt = min(TCH, L - t0);
  stage_wav(seg, wav, b, S, t0, nt, k, stride);
  const bf16x8* afb = afrag_g + (long long)b * ntile * 128;
  for (int i = threadIdx.x; i < ntile * 128; i += 256) afr[i] = afb[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, hi = lane >> 5;
#pragma unroll 1
  for (int ntl = 0; ntl < 2; ++ntl) {
    const int tl = wave * 64 + ntl * 32 + n;
    const bool tv = tl < nt;
    const float* sp = seg + (tv ? tl : 0) * stride;
    bf16_t xh[10], xl[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const float xv = sp[j];            // (taps >= k read the zero tail of the staged segment or the next steps' samples: w is 0 there)
      xh[j] = (bf16_t)xv;
      xl[j] = (bf16_t)(xv - (float)xh[j]);
    }
    const bf16_t one = (bf16_t)1.f;
    bf16x8 b1, b2;
    if (hi == 0) {
      b1 = bf16x8{xh[0], xh[1], xh[2], xh[3], xh[4], xh[5], xh[6], xh[7]};
      b2 = bf16x8{xl[6], xl[7], xl[8], xl[9], xh[0], xh[1], xh[2], xh[3]};
    } else {
      b1 = bf16x8{xh[8], xh[9], xl[0], xl[1], xl[2], xl[3], xl[4], xl[5]};
      b2 = bf16x8{xh[4], xh[5], xh[6], xh[7], xh[8], xh[9], one, one};
    }
    bf16_t* orow = out + ((long long)b * L + t0 + (tv ? tl : 0)) * C + 16 * hi;
    for (int mt = 0; mt < ntile; ++mt) {
      const bf16x8 a1 = afr[(mt * 2 + 0) * 64 + lane], a2 = afr[(mt * 2 + 1) * 64 + lane];
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
      if (tv) {
        bf16x8 o0, o1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o0[e] = (bf16_t)(TAB ? gelu_tab(acc[e], gt) : gelu_poly(acc[e]));
          o1[e] = (bf16_t)(TAB ? gelu_tab(acc[8 + e], gt) : gelu_poly(acc[8 + e]));
        }
        *reinterpret_cast<bf16x8*>(orow + mt * 32) = o0;
        *reinterpret_cast<bf16x8*>(orow + mt * 32 + 8) = o1;
      }
    }
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
      const float dz0 = d0 * (FAST ? gelu_grad_poly(z0) : gelu_grad_f(z0));
      const float dz1 = d1 * (FAST ? gelu_grad_poly(z1) : gelu_grad_f(z1));
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
// ---- backward on the matrix cores (bf16 dY, k <= 10, C % 128 == 0) ---------------------------------------------------------------
// The VALU form above spends ~40 issue slots per element of dY (10 FMAs to recompute the convolution, 10 more for A_j += dz x[s t + j],
// ~20 for gelu', the rest) in dependent chains: 246 us for 262 MB.  Here both contractions run on v_mfma_f32_32x32x16_bf16:
//   1. y[c][t]   = [wh | wh | wl] . [xh ; xl ; xh]                 (the forward's fragments; lane = time step, 16 consecutive channels)
//   2. dz = dY gelu'(z), z = the MFMA result itself (affine folded into the fragments)   (the only per-element VALU work left)
//   3. G[c][j]  += dz[c][t] X[t][j],  X[t][j] = x[s t + j] (j < k), X[t][10] = 1     -> A_j (j < 10) and S1 (column 10) in ONE product,
//      dz as ONE bf16 operand (what every weight-gradient GEMM of the bf16 mode feeds the matrix cores: bf16 dY), X split (Xh + Xl: the
//      waveform keeps ~16 bits); two MFMAs per 16 time steps.  dz comes out of step 1's accumulator
//      layout with the TIME index across lanes, but step 3 contracts over time: the tile goes through a wave-private LDS tile
//      [t][c] and comes back as the A operand through the gfx950 transpose read (ds_read_b64_tr_b16), whose k order
//      (rows 4hi..4hi+3, 8+4hi..8+4hi+3 of a 16-step) the X fragments are built to match.
//   S2 = sum_t dz x_hat is NOT accumulated: x_hat = rs (w.x - mu) gives S2 = rs (sum_j w_j A_j - mu S1) exactly; the final kernel
//   derives it from the chunk-summed A and S1 in fp64 (s2_from_a).
// A wave owns a quarter of the channels (tiles w, w+4, ...: 64 accumulator registers) and walks the block's 8 time tiles.
typedef __attribute__((ext_vector_type(4))) short c0_s16x4;
typedef __attribute__((address_space(3))) c0_s16x4* c0_lds_s16x4_ptr;
constexpr int C0_TP = 80;                 // pitch (bytes) of a [32 t][32 c] bf16 transpose tile row: 64 + 16 of padding
constexpr int C0_TILE_B = 32 * C0_TP;     // 2560 bytes

__device__ __forceinline__ bf16x8 c0_tr_frag(const char* tile, int off_lo, int off_hi) {
  const c0_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c0_lds_s16x4_ptr)(tile + off_lo));
  const c0_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c0_lds_s16x4_ptr)(tile + off_hi));
  union { struct { c0_s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// gelu'(z) = Phi(z) + z phi(z) through a SIGNED 512-entry chord table in LDS (TAB; the forward's gelu_tab is its sibling): nodes
// -R + i h, R = 4.35, h = 2 R / 511, an entry (value, rise) two floats; outside [-R, R] the end values (1.0001 / -1.3e-4).  Six issue
// slots -- scale + offset, clamp (v_med3), truncate, fract, address, the chord -- against the polynomial's 13; |error| <= h^2 / 8 times
// the largest third derivative of gelu (0.77) = 2.8e-5 (the polynomial: 1.2e-4), tests/test_isa_audit.py::
// test_conv0_gelu_grad_table_error_bound.
constexpr float C0_GG_R = 4.35f, C0_GG_SCALE = 511.f / (2.f * C0_GG_R), C0_GG_OFF = 255.5f;
__device__ __forceinline__ float gelu_grad_tab(float z, const float2* gg) {
  const float a = __builtin_amdgcn_fmed3f(fmaf(z, C0_GG_SCALE, C0_GG_OFF), 0.f, 511.f);
  const float2 e = gg[(int)a];
  return fmaf(__builtin_amdgcn_fractf(a), e.y, e.x);
}
__device__ __forceinline__ float c0_gelu_grad_exact(float x) {
  return 0.5f * erfcf(-x * 0.70710678118654752f) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
#ifndef C0_BWD_OCC
#define C0_BWD_OCC 2      // waves per SIMD the backward kernel is compiled for (measured: 3 = 168 registers + 17-27 spilled dwords: 164-177 us against 120)
#endif
#ifndef C0_BWD_PREFETCH
#define C0_BWD_PREFETCH 4 // channel tiles whose dY loads are issued ahead (4 = the whole time step's)
#endif
template <bool TAB>
__global__ __launch_bounds__(256, C0_BWD_OCC) void conv0_bwd_mfma_kernel(const float* __restrict__ wav, const bf16x8* __restrict__ afrag_g,
                                                             const bf16_t* __restrict__ dY,
                                                             float* __restrict__ part, int S, int L, int C, int k, int stride, int nch) {
  extern __shared__ __attribute__((aligned(16))) char sm0[];
  const int ntile = C / 32, tpw = ntile / 4;                          // channel tiles in all / per wave
  bf16x8* afr = reinterpret_cast<bf16x8*>(sm0);                       // [ntile][2][64 lanes] x 16 B (this clip's, affine folded in)
  float* seg = reinterpret_cast<float*>(sm0 + (size_t)ntile * 2048);
  const int b = blockIdx.y, ch = blockIdx.x, t0 = ch * TCH;
  const int nt = min(TCH, L - t0);
  const int nseg_pad = (TCH - 1) * stride + k + MAXK;
  char* trbase = reinterpret_cast<char*>(seg + ((nseg_pad + 3) & ~3));  // one [32 t][32 c] tile per wave
  const float2* gg = reinterpret_cast<const float2*>(trbase + 4 * C0_TILE_B);   // [512] (TAB)
  stage_wav(seg, wav, b, S, t0, nt, k, stride);
  {
    const bf16x8* afb = afrag_g + (long long)b * ntile * 128;
    for (int i = threadIdx.x; i < ntile * 128; i += 256) afr[i] = afb[i];
  }
  if constexpr (TAB) {
    const float h = 2.f * C0_GG_R / 511.f;
    for (int i = threadIdx.x; i < 512; i += 256) {
      const float v0 = c0_gelu_grad_exact(-C0_GG_R + i * h);
      const float v1 = i < 511 ? c0_gelu_grad_exact(-C0_GG_R + (i + 1) * h) : v0;
      const_cast<float2*>(gg)[i] = make_float2(v0, v1 - v0);
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, hi = lane >> 5;
  char* trh = trbase + wave * C0_TILE_B;
  // transpose-read offsets (see c0_tr_frag): lane l of a 16-lane group reads row rbase + ((l & 15) >> 2), columns 16 ((l >> 4) & 1) +
  // 4 (l & 3) .. +3 and receives column l & 31, rows rbase .. rbase + 3;  rbase = 16 s + 8 half + 4 hi
  const int tr_lane = ((lane & 15) >> 2) * C0_TP + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  f32x16 G[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) G[q][r] = 0.f;
#pragma unroll 1
  for (int tt = 0; tt < TCH / 32; ++tt) {
    if (tt * 32 >= nt) break;
    // ---- fragments of this time tile -------------------------------------------------------------------------------------------
    const int tl = tt * 32 + n;                      // (step 1: lane = time step)
    const bool tv = tl < nt;
    bf16x8 b1, b2;
    {
      const float* sp = seg + (tv ? tl : 0) * stride;
      bf16_t xh[10], xl[10];
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const float xv = sp[j];
        xh[j] = (bf16_t)xv;
        xl[j] = (bf16_t)(xv - (float)xh[j]);
      }
      if (hi == 0) {
        b1 = bf16x8{xh[0], xh[1], xh[2], xh[3], xh[4], xh[5], xh[6], xh[7]};
        b2 = bf16x8{xl[6], xl[7], xl[8], xl[9], xh[0], xh[1], xh[2], xh[3]};
      } else {
        b1 = bf16x8{xh[8], xh[9], xl[0], xl[1], xl[2], xl[3], xl[4], xl[5]};
        b2 = bf16x8{xh[4], xh[5], xh[6], xh[7], xh[8], xh[9], (bf16_t)1.f, (bf16_t)1.f};
      }
    }
    const bf16_t* dyrow = dY + ((long long)b * L + t0 + (tv ? tl : 0)) * C + 16 * hi;
    // dY of this time step for the wave's four channel tiles (32 contiguous bytes each): all eight loads are in flight while the
    // first tile's MFMAs and the X fragments above are worked on (one load pair per tile, issued where it is used, left each tile
    // waiting for its own round trip -- two waves per SIMD do not cover that)
    uint4 dvq[4][2];
#pragma unroll
    for (int q = 0; q < C0_BWD_PREFETCH; ++q) {
      if (q < tpw) {
        dvq[q][0] = *reinterpret_cast<const uint4*>(dyrow + (wave + 4 * q) * 32);
        dvq[q][1] = *reinterpret_cast<const uint4*>(dyrow + (wave + 4 * q) * 32 + 8);
      }
    }
    bf16x8 Xh[2], Xl[2];                             // (step 3: lane = column j of X, k slots = time steps in the transpose read's order)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int tq = tt * 32 + 16 * s2 + 8 * (e >> 2) + 4 * hi + (e & 3);
        float xv = 0.f;
        if (n < k && tq < nt) xv = seg[tq * stride + n];
        if (n == 10) xv = 1.f;
        const bf16_t h = (bf16_t)xv;
        Xh[s2][e] = h;
        Xl[s2][e] = (bf16_t)(xv - (float)h);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q >= tpw) break;
      const int mt = wave + 4 * q;
      if (C0_BWD_PREFETCH < 4 && q + C0_BWD_PREFETCH < 4 && q + C0_BWD_PREFETCH < tpw && (q % C0_BWD_PREFETCH) == 0) {
#pragma unroll
        for (int u = 0; u < C0_BWD_PREFETCH; ++u) {      // (the next group of tiles, while this group is worked on)
          dvq[q + C0_BWD_PREFETCH + u][0] = *reinterpret_cast<const uint4*>(dyrow + (wave + 4 * (q + C0_BWD_PREFETCH + u)) * 32);
          dvq[q + C0_BWD_PREFETCH + u][1] = *reinterpret_cast<const uint4*>(dyrow + (wave + 4 * (q + C0_BWD_PREFETCH + u)) * 32 + 8);
        }
      }
      uint4 dv0 = dvq[q][0], dv1 = dvq[q][1];
      if (!tv) { dv0 = make_uint4(0u, 0u, 0u, 0u); dv1 = dv0; }      // (a time step past the clip's end contributes dz = 0)
      const bf16x8 a1 = afr[(mt * 2 + 0) * 64 + lane], a2 = afr[(mt * 2 + 1) * 64 + lane];
      f32x16 y;
#pragma unroll
      for (int r = 0; r < 16; ++r) y[r] = 0.f;
      y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, y, 0, 0, 0);
      y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, y, 0, 0, 0);
      const unsigned int dw_[8] = {dv0.x, dv0.y, dv0.z, dv0.w, dv1.x, dv1.y, dv1.z, dv1.w};
      bf16x8 dh0, dh1;
      typedef float c0_f2 __attribute__((ext_vector_type(2)));
      typedef bf16_t c0_b2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int r = 0; r < 16; r += 2) {      // (pairs: one v_cvt_pk_bf16_f32 per two elements)
        const unsigned int pk = dw_[r >> 1];
        c0_f2 dz;                              // (y IS the GELU argument: the affine rides in the fragments)
        dz[0] = __uint_as_float(pk << 16) * (TAB ? gelu_grad_tab(y[r], gg) : gelu_grad_poly(y[r]));
        dz[1] = __uint_as_float(pk & 0xffff0000u) * (TAB ? gelu_grad_tab(y[r + 1], gg) : gelu_grad_poly(y[r + 1]));
        const c0_b2 hb = __builtin_convertvector(dz, c0_b2);
        if (r < 8) { dh0[r] = hb[0]; dh0[r + 1] = hb[1]; } else { dh1[r - 8] = hb[0]; dh1[r - 7] = hb[1]; }
      }
      // wave-private transpose: rows = this lane's time step, 32 bytes of its 16 channels
      *reinterpret_cast<bf16x8*>(trh + n * C0_TP + 32 * hi) = dh0;
      *reinterpret_cast<bf16x8*>(trh + n * C0_TP + 32 * hi + 16) = dh1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (same wave wrote and reads: the writes have executed)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int o0 = (16 * s2 + 4 * hi) * C0_TP + tr_lane, o1 = (16 * s2 + 8 + 4 * hi) * C0_TP + tr_lane;
        const bf16x8 ah = c0_tr_frag(trh, o0, o1);
        G[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, Xh[s2], G[q], 0, 0, 0);
        G[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, Xl[s2], G[q], 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the tile is rewritten by the next channel tile)
    }
  }
  // G[q][r] of lane (n = column j, hi) = row (r & 3) + 8 (r >> 2) + 4 hi of channel tile wave + 4 q: A_j for j < k, S1 in column 10
  if (n <= 10 && (n < k || n == 10)) {
    const int KV = 12;      // (part rows are KW + 2 = 12 floats: A_0..A_9, S1, S2 -- S2 is derived by the final kernel)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q >= tpw) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = (wave + 4 * q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        part[(((long long)b * nch + ch) * C + c) * KV + n] = G[q][r];
      }
    }
  }
  if (k < 10 && n >= k && n < 10) {      // unused tap slots of a narrower kernel: defined zeros for the reduce kernel
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q >= tpw) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = (wave + 4 * q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        part[(((long long)b * nch + ch) * C + c) * 12 + n] = 0.f;
      }
    }
  }
  if (n == 11) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q >= tpw) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = (wave + 4 * q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        part[(((long long)b * nch + ch) * C + c) * 12 + 11] = 0.f;
      }
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
// One thread per (channel, tap, clip lane): 2 channels x 16 tap lanes x 8 clip lanes per block (tap lanes k..15 idle; tap lane 0 also
// owns dgamma / dbeta); the clip lanes' terms are added by an xor butterfly in a fixed order.  (First form: one thread per channel
// walked B x k x k fp64 terms alone, 124 us; second: one thread per (channel, tap) on C / 16 = 32 blocks, 42 us of dependent fp64
// chains on an eighth of the chip.)
__global__ __launch_bounds__(256) void conv0_bwd_final_kernel(const double* __restrict__ sums, const double* __restrict__ mom,
                                                             const float* __restrict__ w, const float* __restrict__ gamma,
                                                             const float* __restrict__ stats, float* __restrict__ dw,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C,
                                                             int k, int L, int nv, float gscale, int s2_from_a) {
  const int bl = threadIdx.x & 7, j = (threadIdx.x >> 3) & 15, c = blockIdx.x * 2 + (threadIdx.x >> 7);
  double g1 = 0.0, g2 = 0.0, acc = 0.0;
  if (c < C && j < k) {
    const double ga = (double)gamma[c], invL = 1.0 / (double)L;
    for (int b = bl; b < B; b += 8) {
      const double* sb = sums + ((long long)b * C + c) * nv;
      const double* mb = mom + (long long)b * nmom(k);
      const double mu = (double)stats[((long long)b * C + c) * 2], rs = (double)stats[((long long)b * C + c) * 2 + 1];
      const double S1 = sb[nv - 2];
      double S2 = sb[nv - 1];
      if (s2_from_a) {     // (matrix-core backward: S2 = sum_t dz x_hat = rs (sum_j w_j A_j - mu S1), x_hat = rs (w.x - mu))
        double wa = 0.0;
        for (int jp = 0; jp < k; ++jp) wa += (double)w[c * k + jp] * sb[jp];
        S2 = rs * (wa - mu * S1);
      }
      g1 += S1; g2 += S2;
      double rw = 0.0;   // (R w)_j
      for (int jp = 0; jp < k; ++jp) {
        const int a = j < jp ? j : jp, bq = j < jp ? jp : j;
        rw += (double)w[c * k + jp] * mb[ridx(k, a, bq)];
      }
      acc += rs * ga * (sb[j] - S1 * invL * mb[j] - S2 * invL * rs * (rw - mu * mb[j]));
    }
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    acc += __shfl_xor(acc, o, 64);
    g1 += __shfl_xor(g1, o, 64);
    g2 += __shfl_xor(g2, o, 64);
  }
  if (c >= C || j >= k || bl != 0) return;
  if (dw) dw[c * k + j] += gscale * (float)acc;
  if (j == 0) {
    if (dgamma) dgamma[c] += gscale * (float)g2;
    if (dbeta) dbeta[c] += gscale * (float)g1;
  }
}

// workspace layout (floats unless noted): [chunk partials: max(B nch nmom, B nch C (KWmax + 2))] [mom: B nmom doubles]
// [sums: B C (KWmax + 2) doubles]
struct Ws { float* part; double* mom; double* sums; bf16x8* afrag; };
__host__ inline int64_t ws_part_floats(int B, int nch, int C, int k) {
  const int64_t a = (int64_t)B * nch * nmom(k), b = (int64_t)B * nch * C * (MAXK + 2);
  return ((a > b ? a : b) + 1) / 2 * 2;   // keep the doubles 8-byte aligned
}
__host__ inline Ws carve(void* ws, int B, int nch, int C, int k) {
  Ws r;
  r.part = (float*)ws;
  r.mom = (double*)(r.part + ws_part_floats(B, nch, C, k));
  r.sums = r.mom + (int64_t)B * nmom(k);
  r.afrag = reinterpret_cast<bf16x8*>(r.sums + (int64_t)B * C * (MAXK + 2));      // (8-byte aligned doubles in front: a multiple of 16 bytes
  r.afrag = reinterpret_cast<bf16x8*>((reinterpret_cast<uintptr_t>(r.afrag) + 15) & ~(uintptr_t)15);   //  is enforced here)
  return r;
}

// waveform moments (+ statistics when `stats` is given): two launches
void launch_moments_stats(const float* wav, const float* w, float* stats, const Ws& W, int B, int S, int L, int C, int k, int stride, float eps,
                          hipStream_t s, bool with_stats = true, double* mom = nullptr) {
  const int nchm = (L + TCM - 1) / TCM;
  const size_t shm2 = (size_t)(((TCM - 1) * stride + k + 8 + 3) & ~3) * sizeof(float);
  if (shm2 > 60 * 1024) return;      // (strides far beyond the recipe's 5: the caller rejects them, see st5_conv0_gn_gelu_fwd)
  if (k <= 10) hipLaunchKernelGGL((conv0_moments2_kernel<10>), dim3(nchm, B), dim3(256), shm2, s, wav, W.part, S, L, k, stride, nchm);
  else hipLaunchKernelGGL((conv0_moments2_kernel<MAXK>), dim3(nchm, B), dim3(256), shm2, s, wav, W.part, S, L, k, stride, nchm);
  if (with_stats) hipLaunchKernelGGL(conv0_stats2_kernel, dim3((C + 63) / 64, B), dim3(256), 0, s, W.part, mom ? mom : W.mom, w, stats, C, k, L, nchm, eps);
}

}  // namespace

namespace { int g_conv0_mfma = 1, g_conv0_fold = 1, g_conv0_gelu_tab = 1; }
/* 1 (default): the matrix-core forward evaluates GELU through a 256-entry chord table of the normal upper tail in LDS (8.5 VALU slots
   per element); 0: the degree-19 polynomial (14.5 slots; A/B). */
extern "C" int st5_conv0_set_gelu_table(int on) { g_conv0_gelu_tab = on ? 1 : 0; return ST5_OK; }
/* 1 (default): the matrix-core forward derives statistics and weight fragments in one launch (three in all); 0: conv0_stats2_kernel +
   conv0_wfrag_kernel (four launches; A/B). */
extern "C" int st5_conv0_set_fold(int on) { g_conv0_fold = on ? 1 : 0; return ST5_OK; }
/* 1 (default): the bf16 forward apply pass on the matrix cores (split-bf16 operands); 0: the VALU form (A/B, and what fp32 always runs). */
extern "C" int st5_conv0_set_mfma(int on) { g_conv0_mfma = on ? 1 : 0; return ST5_OK; }

extern "C" int64_t st5_conv0_ws_bytes(int32_t B, int32_t S, int32_t C, int32_t k, int32_t stride) {
  const int L = out_len(S, k, stride);
  const int64_t nch = (L + TCH - 1) / TCH;
  return ws_part_floats(B, (int)nch, C, k) * (int64_t)sizeof(float) +
         ((int64_t)B * nmom(k) + (int64_t)B * C * (MAXK + 2)) * (int64_t)sizeof(double) + 16 + (int64_t)B * C * 64;   // + MFMA weight fragments (per clip: the GroupNorm affine is folded in)
}

extern "C" int32_t st5_conv0_mom_count(int32_t k) { return k >= 1 && k <= MAXK ? nmom(k) : 0; }

extern "C" int st5_conv0_gn_gelu_fwd(const float* wav, const float* w, const float* gamma, const float* beta,
                                     void* out, float* stats, void* ws, int32_t B, int32_t S, int32_t C, int32_t k,
                                     int32_t stride, float eps, int dtype, void* stream) {
  return st5_conv0_gn_gelu_fwd_m(wav, w, gamma, beta, out, stats, nullptr, ws, B, S, C, k, stride, eps, dtype, stream);
}
extern "C" int st5_conv0_gn_gelu_fwd_m(const float* wav, const float* w, const float* gamma, const float* beta,
                                       void* out, float* stats, double* mom, void* ws, int32_t B, int32_t S, int32_t C, int32_t k,
                                       int32_t stride, float eps, int dtype, void* stream) {
  if (!wav || !w || !gamma || !beta || !out || !stats || !ws) return ST5_ERR_ARG;
  if (C % 8 || C > 2048 || 256 % (C / 8 > 256 ? 256 : C / 8) || k > MAXK || k < 1 || stride < 1 || stride > 7) return ST5_ERR_ARG;
  if (C / 8 > 256) return ST5_ERR_ARG;
  const int L = out_len(S, k, stride);
  if (L <= 0 || B <= 0) return ST5_ERR_ARG;
  if (dtype != ST5_BF16 && dtype != ST5_F32) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nch = (L + TCH - 1) / TCH;
  const size_t shm = (size_t)(TCH * stride + k + MAXK) * sizeof(float);
  const Ws W = carve(ws, B, nch, C, k);
  const bool mfma = dtype == ST5_BF16 && g_conv0_mfma && k <= 10 && C % 32 == 0 && C <= 1024;
  const bool fold = mfma && g_conv0_fold;
  launch_moments_stats(wav, w, stats, W, B, S, L, C, k, stride, eps, s, !fold, mom);
#define APPLY(TT, KW)                                                                                           \
  hipLaunchKernelGGL((conv0_apply_kernel<TT, KW>), dim3(nch, B), dim3(256), shm, s, wav, w, gamma, beta, stats, \
                     (TT*)out, S, L, C, k, stride)
  if (mfma) {
    const int seg_bytes = (int)((shm + 7) & ~(size_t)7);
    const size_t shm_m = (size_t)(C / 32) * 2048 + seg_bytes + (g_conv0_gelu_tab ? 256 * sizeof(float2) : 0);
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)conv0_apply_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
          hipFuncSetAttribute((const void*)conv0_apply_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
        return ST5_ERR_LAUNCH;
      attr = true;
    }
    if (fold) hipLaunchKernelGGL(conv0_stats_wfrag_kernel, dim3(C / 32, B), dim3(256), 0, s, (const float*)W.part, w, gamma, beta, (const float*)nullptr,
                                 stats, mom, W.afrag, C, k, L, (L + TCM - 1) / TCM, eps);
    else hipLaunchKernelGGL(conv0_wfrag_kernel, dim3(C / 32, B), dim3(128), 0, s, w, gamma, beta, stats, W.afrag, C, k);
    if (g_conv0_gelu_tab)
      hipLaunchKernelGGL(conv0_apply_mfma_kernel<true>, dim3(nch, B), dim3(256), shm_m, s, wav, W.afrag, (bf16_t*)out, S, L, C, k, stride, seg_bytes);
    else
      hipLaunchKernelGGL(conv0_apply_mfma_kernel<false>, dim3(nch, B), dim3(256), shm_m, s, wav, W.afrag, (bf16_t*)out, S, L, C, k, stride, seg_bytes);
  } else if (dtype == ST5_BF16) { if (k <= 10) APPLY(bf16_t, 10); else APPLY(bf16_t, MAXK); }
  else { if (k <= 10) APPLY(float, 10); else APPLY(float, MAXK); }
#undef APPLY
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_conv0_gn_gelu_bwd(const float* wav, const float* w, const float* gamma, const float* beta,
                                     const float* stats, const void* dY, float* dw, float* dgamma, float* dbeta,
                                     void* ws, int32_t B, int32_t S, int32_t C, int32_t k, int32_t stride,
                                     float gscale, int dtype, void* stream) {
  return st5_conv0_gn_gelu_bwd_m(wav, w, gamma, beta, stats, nullptr, dY, dw, dgamma, dbeta, ws, B, S, C, k, stride, gscale, dtype, stream);
}
extern "C" int st5_conv0_gn_gelu_bwd_m(const float* wav, const float* w, const float* gamma, const float* beta,
                                       const float* stats, const double* mom, const void* dY, float* dw, float* dgamma, float* dbeta,
                                       void* ws, int32_t B, int32_t S, int32_t C, int32_t k, int32_t stride,
                                       float gscale, int dtype, void* stream) {
  if (!wav || !w || !gamma || !beta || !stats || !dY || !ws) return ST5_ERR_ARG;
  if (C % 8 || C / 8 > 256 || 256 % (C / 8) || k > MAXK || k < 1 || stride < 1 || stride > 7) return ST5_ERR_ARG;
  const int L = out_len(S, k, stride);
  if (L <= 0 || B <= 0) return ST5_ERR_ARG;
  if (dtype != ST5_BF16 && dtype != ST5_F32) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nch = (L + TCH - 1) / TCH;
  const size_t shm = (size_t)(TCH * stride + k + MAXK) * sizeof(float);
  const Ws W = carve(ws, B, nch, C, k);
  // waveform moments again (0.64 MB/clip; cheaper than keeping them alive between forward and backward)
  const bool mfma = dtype == ST5_BF16 && g_conv0_mfma && k <= 10 && C % 128 == 0 && C <= 512 && TCH % 32 == 0;
  const bool fold = mfma && g_conv0_fold && !mom;      // (the fragment launch folds and publishes the moments itself)
  // waveform moments: the forward's, when the caller kept them (B x nmom doubles); else once more from the waveform (0.64 MB / clip)
  if (!mom) launch_moments_stats(wav, w, nullptr, W, B, S, L, C, k, stride, 0.f, s, !fold);
  const int KWv = k <= 10 ? 10 : MAXK, nv = KWv + 2;
  {   // ST5_POISON=1 (debug): the partials region is NaN before the backward kernel fills it -- a reduce that ran ahead of a
      // block of conv0_bwd_kernel, or a block that never stored, then shows as NaN instead of as last step's value
    static const bool poison = [] { const char* e = getenv("ST5_POISON"); return e && e[0] == '1'; }();
    if (poison && hipMemsetAsync(W.part, 0xFF, (size_t)B * nch * C * nv * sizeof(float), s) != hipSuccess) return ST5_ERR_LAUNCH;
  }
#define BWD(TT, KW)                                                                                               \
  hipLaunchKernelGGL((conv0_bwd_kernel<TT, KW>), dim3(nch, B), dim3(256), shm, s, wav, w, gamma, beta, stats,     \
                     (const TT*)dY, W.part, S, L, C, k, stride, nch)
  // (C <= 512: conv0_bwd_mfma_kernel holds at most four 128-channel tiles per wave -- G[4], dvq[4]; wider layers take the VALU kernel,
  //  whose channel loop has no such bound.  ADVICE r5: the guard said 1024 and tiles 4.. were silently never computed.)
  if (mfma) {
    if (fold) hipLaunchKernelGGL(conv0_stats_wfrag_kernel, dim3(C / 32, B), dim3(256), 0, s, (const float*)W.part, w, gamma, beta, stats, (float*)nullptr,
                                 W.mom, W.afrag, C, k, L, (L + TCM - 1) / TCM, 0.f);
    else hipLaunchKernelGGL(conv0_wfrag_kernel, dim3(C / 32, B), dim3(128), 0, s, w, gamma, beta, stats, W.afrag, C, k);
    const size_t seg_f = (size_t)(((TCH - 1) * stride + k + MAXK + 3) & ~3);
    const size_t shm_m = (size_t)(C / 32) * 2048 + seg_f * sizeof(float) + (size_t)4 * C0_TILE_B + (g_conv0_gelu_tab ? 512 * sizeof(float2) : 0);
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)conv0_bwd_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
          hipFuncSetAttribute((const void*)conv0_bwd_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
        return ST5_ERR_LAUNCH;
      attr = true;
    }
    if (g_conv0_gelu_tab)
      hipLaunchKernelGGL(conv0_bwd_mfma_kernel<true>, dim3(nch, B), dim3(256), shm_m, s, wav, W.afrag, (const bf16_t*)dY, W.part, S, L, C, k, stride, nch);
    else
      hipLaunchKernelGGL(conv0_bwd_mfma_kernel<false>, dim3(nch, B), dim3(256), shm_m, s, wav, W.afrag, (const bf16_t*)dY, W.part, S, L, C, k, stride, nch);
  } else if (dtype == ST5_BF16) { if (k <= 10) BWD(bf16_t, 10); else BWD(bf16_t, MAXK); }
  else { if (k <= 10) BWD(float, 10); else BWD(float, MAXK); }
#undef BWD
  hipLaunchKernelGGL(conv0_bwd_reduce_kernel, dim3((C * nv + 31) / 32, B), dim3(256), 0, s, W.part, W.sums, C, nch, nv);
  hipLaunchKernelGGL(conv0_bwd_final_kernel, dim3((C + 1) / 2), dim3(256), 0, s, W.sums, mom ? mom : (const double*)W.mom, w, gamma, stats, dw, dgamma,
                     dbeta, B, C, k, L, nv, gscale, mfma ? 1 : 0);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
