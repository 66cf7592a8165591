// Shared device helpers for the SpeechT5 gfx950 kernels (wave64, CDNA4).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ST5_OK 0
#define ST5_ERR_ARG 1
#define ST5_ERR_ALIGN 2
#define ST5_ERR_LAUNCH 3

#define ST5_F32 0
#define ST5_BF16 1

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VEC = 4;  // elements per 16 bytes
  __device__ static float to_f(float v) { return v; }
  __device__ static float from_f(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static float to_f(bf16_t v) { return (float)v; }
  __device__ static bf16_t from_f(float v) { return (bf16_t)v; }
};

__device__ __forceinline__ float bf16_bits_to_f(unsigned int lo16) {
  return __uint_as_float(lo16 << 16);
}

// Load `n` (<= 8) consecutive elements as floats; 16-byte vector path when full & aligned.
template <typename T>
__device__ __forceinline__ void load8f(const T* p, float (&v)[8]);
template <>
__device__ __forceinline__ void load8f<float>(const float* p, float (&v)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
template <>
__device__ __forceinline__ void load8f<bf16_t>(const bf16_t* p, float (&v)[8]) {
  u32x4 a = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(a[i] << 16);
    v[2 * i + 1] = __uint_as_float(a[i] & 0xffff0000u);
  }
}
template <typename T>
__device__ __forceinline__ void store8f(T* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void store8f<float>(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <>
__device__ __forceinline__ void store8f<bf16_t>(bf16_t* p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
  *reinterpret_cast<bf16x8*>(p) = o;
}

// Wave64 sum on the DPP cross-lane paths (row_shr within 16-lane rows, then row_bcast:15 / row_bcast:31): six VALU
// instructions and one readlane, no LDS-crossbar (ds_bpermute) round trips as the __shfl_xor butterfly needs.
__device__ __forceinline__ float wave_sum(float v) {
  int x = __float_as_int(v);
#define ST5_DPP_ADD(ctrl, row_mask, bank_mask) \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, ctrl, row_mask, bank_mask, false)); x = __float_as_int(v)
  ST5_DPP_ADD(0x111, 0xf, 0xf);   // row_shr:1
  ST5_DPP_ADD(0x112, 0xf, 0xf);   // row_shr:2
  ST5_DPP_ADD(0x114, 0xf, 0xe);   // row_shr:4
  ST5_DPP_ADD(0x118, 0xf, 0xc);   // row_shr:8  -> lane 15 of every row holds the row sum
  ST5_DPP_ADD(0x142, 0xa, 0xf);   // row_bcast:15 -> rows 1 and 3 add the previous row's total
  ST5_DPP_ADD(0x143, 0xc, 0xf);   // row_bcast:31 -> lane 63 holds the wave total
#undef ST5_DPP_ADD
  return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// OCP MX quantisation (e4m3 elements, one e8m0 scale byte per 32 consecutive elements) of the NE = 4 or 8 consecutive elements one
// lane holds; 32 / NE neighbouring lanes (aligned in the wave) hold one block.  scale = floor(log2(max|finite elements|)) - 8 + 127,
// elements = RNE(x * 2^(127 - scale)) saturating finite values at +-448; a NaN / Inf element becomes the e4m3 NaN code 0x7f and its
// block's scale the e8m0 NaN 0xff (a diverged tensor must reach the loss, not be clamped).
// Round 6: fast path.  The block maximum is taken on the integer image of |x| (for finite values the integer order IS the float order,
// and a value >= 0x7f800000 is exactly "NaN or Inf"): v_and + v_max3_u32 instead of a compare / select pair per element, the cross-lane
// step on DPP quad permutes instead of ds_bpermute; when NO lane of the wave saw a non-finite value -- one ballot -- the elements are
// scaled, clamped with v_med3_f32 and converted without any per-element non-finite bookkeeping.  ~5 VALU slots per element instead of
// ~17: what made the quantiser affordable inside GEMM / LayerNorm epilogues.  Same bytes as the careful path for every input.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int mx8_dpp_max(unsigned int m, const int ctrl_sel) {
  unsigned int o;
  if (ctrl_sel == 0) o = (unsigned int)__builtin_amdgcn_update_dpp((int)m, (int)m, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
  else if (ctrl_sel == 1) o = (unsigned int)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  else o = (unsigned int)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x141, 0xf, 0xf, false);                    // row_half_mirror (quad 0 <-> quad 1 of 8 lanes)
  return o > m ? o : m;
}
template <int NE>
__device__ __forceinline__ void mx8_quant_careful(const float (&v)[NE], unsigned int (&q)[NE / 4], unsigned int& scale_byte) {
  float amax = 0.f;
  unsigned int nf = 0u;          // bit e: element e is NaN or Inf
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const bool bad = (__float_as_uint(v[e]) & 0x7f800000u) == 0x7f800000u;
    nf |= bad ? (1u << e) : 0u;
    amax = fmaxf(amax, bad ? 0.f : fabsf(v[e]));
  }
  unsigned int am = __float_as_uint(amax), nf_blk = nf != 0u ? 1u : 0u;
#pragma unroll
  for (int st = 0; st < (NE == 8 ? 2 : 3); ++st) {
    am = mx8_dpp_max(am, st);
    nf_blk = mx8_dpp_max(nf_blk, st);
  }
  int E = (int)((am >> 23) & 0xffu) - 8;      // biased exponent of amax, minus emax(e4m3)
  E = E < 0 ? 0 : (E > 254 ? 254 : E);
  const float inv = __uint_as_float((unsigned int)(254 - E) << 23);    // 2^(127 - E)
#pragma unroll
  for (int w = 0; w < NE / 4; ++w) {
    float f[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = ((nf >> (4 * w + e)) & 1u) ? 0.f : v[4 * w + e] * inv;
      f[e] = fminf(fmaxf(t, -448.f), 448.f);
    }
    int pk = 0;
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], pk, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], pk, true);
    unsigned int u = (unsigned int)pk;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if ((nf >> (4 * w + e)) & 1u) u = (u & ~(0xffu << (8 * e))) | (0x7fu << (8 * e));
    q[w] = u;
  }
  scale_byte = nf_blk ? 0xffu : (unsigned int)E;
}
template <int NE>
__device__ __forceinline__ void mx8_quant(const float (&v)[NE], unsigned int (&q)[NE / 4], unsigned int& scale_byte) {
  static_assert(NE == 4 || NE == 8, "a lane holds 4 or 8 elements of a 32-element block");
  unsigned int m = 0u;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const unsigned int a = __float_as_uint(v[e]) & 0x7fffffffu;
    m = a > m ? a : m;
  }
#pragma unroll
  for (int st = 0; st < (NE == 8 ? 2 : 3); ++st) m = mx8_dpp_max(m, st);
  if (__ballot(m >= 0x7f800000u) != 0ull) {       // (wave-uniform; never taken in a healthy run)
    mx8_quant_careful<NE>(v, q, scale_byte);
    return;
  }
  int E = (int)(m >> 23) - 8;
  E = E < 0 ? 0 : E;                               // (m < 0x7f800000: E <= 246)
  const float inv = __uint_as_float((unsigned int)(254 - E) << 23);
#pragma unroll
  for (int w = 0; w < NE / 4; ++w) {
    float f[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = __builtin_amdgcn_fmed3f(v[4 * w + e] * inv, -448.f, 448.f);
    int pk = 0;
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], pk, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], pk, true);
    q[w] = (unsigned int)pk;
  }
  scale_byte = (unsigned int)E;
}

// exact (erf) GELU as torch.nn.GELU() / fairseq "gelu" in fp32
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// GELU for the bf16 compute mode: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16 resolution) on
// v_rcp / v_exp instead of the branchy libm erff -- the GEMM epilogues of fc1 are VALU-bound on this function.
// g = gelu(x); when GRAD, g = d gelu / dx (shares the exp(-x^2/2) term).
template <bool GRAD>
__device__ __forceinline__ float gelu_fast(float x) {
  const float a = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = __builtin_amdgcn_exp2f(-a * a * 1.4426950408889634f);   // exp(-x^2 / 2)
  const float erf_abs = fmaf(-poly, e, 1.0f);
  const float erf_x = copysignf(erf_abs, x);
  const float cdf = fmaf(0.5f, erf_x, 0.5f);
  if (GRAD) return fmaf(x * 0.3989422804014327f, e, cdf);
  return x * cdf;
}

// GELU VALUE for the bf16 compute mode without transcendentals (round 5).  v_rcp_f32 / v_exp_f32 issue at a quarter of the VALU rate, so
// the two of gelu_fast<false> cost as much as its ten other instructions together -- and the epilogues that call it (conv layer 0's
// apply pass, the GEMM epilogue of fc1) are VALU-issue-bound.  Phi(x) = 0.5 + xc G(xc^2), xc = x clamped to +-3 sqrt 2, G a degree-9
// polynomial (Chebyshev fit of erf(sqrt u) / sqrt u on [0, 9] with the 1/sqrt 2 and 1/2 scalings folded into the coefficients, fp32
// Horner): |Phi error| <= 1.5e-5 incl. the clamp (1 - erf 3 = 2.2e-5), |gelu error| <= 2.4e-5 for |x| <= 4 and <= 1.5e-5 |x| beyond --
// two orders below the bf16 rounding of the value it feeds (tests/test_isa_audit.py::test_gelu_poly_error_bound).  13 full-rate
// instructions against 13 + 2 quarter-rate ones (= 21 issue slots): fc1's epilogue 87.4 -> 73.1 us at 8192 x 3072 x 768.
// The DERIVATIVE has its own polynomial (gelu_grad_poly below).
__device__ __forceinline__ float gelu_poly(float x) {
  const float xc = fminf(fmaxf(x, -4.242640495300293f), 4.242640495300293f);
  const float v = xc * xc;
  float g = fmaf(v, -3.086644655e-12f, 3.179429497e-10f);
  g = fmaf(g, v, -1.470189481e-08f);
  g = fmaf(g, v, 4.085038654e-07f);
  g = fmaf(g, v, -7.744979484e-06f);
  g = fmaf(g, v, 1.082136150e-04f);
  g = fmaf(g, v, -1.169087715e-03f);
  g = fmaf(g, v, 9.949624538e-03f);
  g = fmaf(g, v, -6.647801399e-02f);
  g = fmaf(g, v, 3.989412189e-01f);
  return x * fmaf(xc, g, 0.5f);
}

// GELU DERIVATIVE for the bf16 compute mode without transcendentals: gelu'(x) = Phi(x) + x phi(x) = 0.5 + xc H(xc^2), xc = x clamped to
// +-4.35, H a degree-10 polynomial (Chebyshev fit of (Phi(sqrt u) - 0.5) / sqrt u + phi(sqrt u) on [0, 4.35^2], fp32 Horner):
// |error| <= 1.2e-4 of a value in [-0.13, 1.13] -- the factor multiplies a bf16 dY (relative rounding 2e-3), so it is 16x below the
// operand's own rounding.  13 full-rate instructions against 15 + 2 quarter-rate ones of gelu_fast<true> (= 23 issue slots).
__device__ __forceinline__ float gelu_grad_poly(float x) {
  const float xc = fminf(fmaxf(x, -4.349999904632568f), 4.349999904632568f);
  const float v = xc * xc;
  float h = fmaf(v, 1.668488481e-12f, -1.931875215e-10f);
  h = fmaf(h, v, 1.003747396e-08f);
  h = fmaf(h, v, -3.115285097e-07f);
  h = fmaf(h, v, 6.503215900e-06f);
  h = fmaf(h, v, -9.763300477e-05f);
  h = fmaf(h, v, 1.097474480e-03f);
  h = fmaf(h, v, -9.375064634e-03f);
  h = fmaf(h, v, 5.970102549e-02f);
  h = fmaf(h, v, -2.658985257e-01f);
  h = fmaf(h, v, 7.978798151e-01f);
  return fmaf(xc, h, 0.5f);
}

#define ACT_NONE 0
#define ACT_GELU 1
#define ACT_RELU 2
#define ACT_TANH 3
#define ACT_LRELU_01 4   // LeakyReLU(0.1)  (HiFi-GAN)
#define ACT_LRELU_001 5  // LeakyReLU(0.01) (F.leaky_relu default)

template <bool FAST = false>
__device__ __forceinline__ float act_f(int act, float x) {
  switch (act) {
    case ACT_GELU: return FAST ? gelu_poly(x) : gelu_f(x);
    case ACT_RELU: return x > 0.f ? x : 0.f;
    case ACT_TANH: return tanhf(x);
    case ACT_LRELU_01: return x > 0.f ? x : 0.1f * x;
    case ACT_LRELU_001: return x > 0.f ? x : 0.01f * x;
    default: return x;
  }
}
// derivative given the PRE-activation value
template <bool FAST = false>
__device__ __forceinline__ float act_grad_f(int act, float x) {
  switch (act) {
    case ACT_GELU: return FAST ? gelu_grad_poly(x) : gelu_grad_f(x);
    case ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case ACT_TANH: { float t = tanhf(x); return 1.f - t * t; }
    case ACT_LRELU_01: return x > 0.f ? 1.f : 0.1f;
    case ACT_LRELU_001: return x > 0.f ? 1.f : 0.01f;
    default: return 1.f;
  }
}

// Counter-based RNG for dropout, two levels so that the per-element cost is a few VALU instructions:
//   * every aligned block of 64 consecutive element indices gets a 32-bit key = mix(fold(splitmix64(seed)) ^ (idx >> 6));
//   * every aligned PAIR of elements inside the block gets 32 bits = mix(key ^ pair * golden), 16 bits per
//     element (two xorshift-multiply rounds with 24-bit multiplies); keep <=> u16 >= p * 65536.
// The same (seed, idx) is re-evaluated by every kernel that needs the mask (GEMM epilogue, softmax, fused attention,
// backward passes), so no mask is ever stored.  Attention probabilities use idx = row * round_up(row_len, 64) + key so a
// 64-key tile of the fused kernels is exactly one block.
__device__ __forceinline__ unsigned long long rng_hash64(unsigned long long seed, unsigned long long group) {
  unsigned long long z = group + seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned int dropout_thresh(float p) { return (unsigned int)(p * 65536.0f); }
// 32-bit key of a 64-element block.  The seed part (splitmix64 of the seed, folded) is wave-uniform and loop-invariant,
// so it is computed once on the scalar unit; the per-block part is one xorshift-multiply mixer on full-rate 24-bit
// multiplies (different constants from the per-pair mixer below).
// A `seed` argument with bit 63 set is not a seed but a device POINTER (low 48 bits) to the 64-bit seed.  Host code hands out
// such slots when a training step is captured into a HIP graph (speecht5_amd/graph.py): the kernel arguments stay constant
// across replays while the host refreshes the slots' contents before every replay, so each replay draws fresh dropout masks.
// The pointer derives from a kernel argument, i.e. it is wave-uniform: the load is one scalar load per wave.
__device__ __forceinline__ unsigned long long resolve_seed(unsigned long long seed) {
  if (seed >> 63) seed = *reinterpret_cast<const unsigned long long*>(seed & 0x0000FFFFFFFFFFFFull);
  return seed;
}
// The seed part of the block key (loop-invariant).  Kernels that derive keys inside a loop that keeps LDS-DMA in flight call this ONCE in
// front of the loop: resolving a seed slot is a global load, and the s_waitcnt vmcnt(0) behind it drains the DMA queue -- inside the
// attention tile loops that was one full landing latency per key tile (and ~30 VALU instructions of 64-bit multiplies per tile).
__device__ __forceinline__ unsigned int drop_seed_fold(unsigned long long seed) {
  const unsigned long long h = rng_hash64(resolve_seed(seed), 0ull);
  return (unsigned int)h ^ (unsigned int)(h >> 32);
}
__device__ __forceinline__ unsigned int drop_block_key_folded(unsigned int seed_fold, unsigned long long block) {
  unsigned int x = (unsigned int)block ^ seed_fold ^ __umul24((unsigned int)(block >> 32), 0x9E3779u);
  x ^= x >> 15; x = __umul24(x, 0x9E3779u);
  x ^= x >> 12; x = __umul24(x, 0x85EBCBu);
  x ^= x >> 15;
  return x;
}
__device__ __forceinline__ unsigned int drop_block_key(unsigned long long seed, unsigned long long block) {
  return drop_block_key_folded(drop_seed_fold(seed), block);
}
// 2 x 16 random bits for elements (2*pair, 2*pair+1) of a block; pair in [0, 32)
// pc = pair * 0x9E3779B1 (callers with a compile-time pair pass the product)
__device__ __forceinline__ unsigned int drop_pair_bits_pc(unsigned int key, unsigned int pc) {
  unsigned int x = key ^ pc;
  x ^= x >> 16; x = __umul24(x, 0xCA6B85u);  // v_mul_u32_u24 is full rate (v_mul_lo_u32 is quarter rate)
  x ^= x >> 13; x = __umul24(x, 0xB2AE35u);
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned int drop_pair_bits(unsigned int key, unsigned int pair) {
  return drop_pair_bits_pc(key, pair * 0x9E3779B1u);
}
__device__ __forceinline__ bool drop_keep(unsigned int bits, int odd, unsigned int thresh16) {
  return (odd ? bits >> 16 : bits & 0xffffu) >= thresh16;
}
__device__ __forceinline__ float drop_pick(unsigned int bits, int odd, unsigned int thresh16, float inv_keep) {
  return ((odd ? bits >> 16 : bits & 0xffffu) >= thresh16) ? inv_keep : 0.f;
}
// scale factor (0 or inv_keep) of a single element
__device__ __forceinline__ float dropout_scale(unsigned long long seed, unsigned long long idx, unsigned int thresh16,
                                               float inv_keep) {
  const unsigned int key = drop_block_key(seed, idx >> 6);
  return drop_pick(drop_pair_bits(key, ((unsigned int)idx & 63u) >> 1), (int)(idx & 1), thresh16, inv_keep);
}
// scale factors of the 8 elements idx8 .. idx8+7 (idx8 % 8 == 0): one block key, four pair hashes
__device__ __forceinline__ void dropout_scale8(unsigned long long seed, unsigned long long idx8, unsigned int thresh16,
                                               float inv_keep, float (&out)[8]) {
  const unsigned int key = drop_block_key(seed, idx8 >> 6);
  const unsigned int p0 = ((unsigned int)idx8 & 63u) >> 1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned int bits = drop_pair_bits(key, p0 + e);
    out[2 * e] = drop_pick(bits, 0, thresh16, inv_keep);
    out[2 * e + 1] = drop_pick(bits, 1, thresh16, inv_keep);
  }
}
// The same with the seed part of the block key computed by the caller ONCE (drop_seed_fold): the splitmix hash of the seed -- 64-bit
// multiplies at a quarter of the VALU rate, behind a conditional load when the seed is a device slot -- is loop-invariant, but sits under
// the callers' row / column predicates where hipcc does not hoist it (round 6: the LayerNorm backward's dropped second output cost
// +9 us on a 13 us kernel, the dropout epilogue of the GEMMs +4 .. 15 us).  Same bits.
__device__ __forceinline__ void dropout_scale8_folded(unsigned int seed_fold, unsigned long long idx8, unsigned int thresh16, float inv_keep,
                                                      float (&out)[8]) {
  const unsigned int key = drop_block_key_folded(seed_fold, idx8 >> 6);
  const unsigned int p0 = ((unsigned int)idx8 & 63u) >> 1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned int bits = drop_pair_bits(key, p0 + e);
    out[2 * e] = drop_pick(bits, 0, thresh16, inv_keep);
    out[2 * e + 1] = drop_pick(bits, 1, thresh16, inv_keep);
  }
}
__device__ __forceinline__ void dropout_scale4_folded(unsigned int seed_fold, unsigned long long idx4, unsigned int thresh16, float inv_keep,
                                                      float (&out)[4]) {
  const unsigned int key = drop_block_key_folded(seed_fold, idx4 >> 6);
  const unsigned int p0 = ((unsigned int)idx4 & 63u) >> 1;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const unsigned int bits = drop_pair_bits(key, p0 + e);
    out[2 * e] = drop_pick(bits, 0, thresh16, inv_keep);
    out[2 * e + 1] = drop_pick(bits, 1, thresh16, inv_keep);
  }
}
// scale factors of the 4 elements idx4 .. idx4+3 (idx4 % 4 == 0): one block key, two pair hashes
__device__ __forceinline__ void dropout_scale4(unsigned long long seed, unsigned long long idx4, unsigned int thresh16,
                                               float inv_keep, float (&out)[4]) {
  const unsigned int key = drop_block_key(seed, idx4 >> 6);
  const unsigned int p0 = ((unsigned int)idx4 & 63u) >> 1;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const unsigned int bits = drop_pair_bits(key, p0 + e);
    out[2 * e] = drop_pick(bits, 0, thresh16, inv_keep);
    out[2 * e + 1] = drop_pick(bits, 1, thresh16, inv_keep);
  }
}
__device__ __forceinline__ long long drop_row_stride(int row_len) { return ((long long)row_len + 63) & ~63ll; }

// Kernels that are built for exactly two waves per SIMD (the 128x128 GEMM family: ~180 VGPRs) declare the WHOLE half of the
// register file (256 VGPRs), so that no third wave -- of this or of ANY OTHER kernel on another stream -- is placed on a SIMD
// that already holds two of them.  Measured on MI355X / ROCm 7.2 (tools/diag/diag_order2.py): a wave of an unrelated kernel
// (conv layer 0's backward: 88 VGPRs, 250 us per block) that shared a SIMD with two ~184-VGPR GEMM waves (general, LDS-DMA
// NT or TN form alike; never with one such wave, never with a copy kernel) had lanes 48..63 of its registers corrupted in
// 10-16 of 16 runs -- the source of the run-to-run differences of the side-by-side micro-batches at full size.  With the
// padding the combination cannot be scheduled; the GEMM kernels' own occupancy (2 waves/SIMD) is unchanged.
// Round 6: OFF by default.  Section 4c of DESIGN.md traced the corruption this padding was built against to one missing LDS wait in
// fa2::bwd_dkv_kernel -- the padded library of round 4 showed it all the same (tools/r5/dkv_pair.py) -- so what the padding still did
// was keep small kernels of the OTHER micro-batch's stream off the SIMDs of the 128x128 GEMMs (122-189 registers without it).  Measured
// on the benched update, same box, alternating (profiles/r6b_knob_ab.txt): 28.67 / 28.80 / 28.96 ms without against 28.82 / 28.95 /
// 29.05 with; replayed side by side == eager in turn bit for bit either way.  -DST5_PAD256 (tools/r6b/build_pad_lib.sh) restores it.
#ifdef ST5_PAD256
#define ST5_PAD_TO_256_VGPRS() asm volatile("; vgpr allocation padded to 256" ::: "v255")
#else
#define ST5_PAD_TO_256_VGPRS()
#endif

// Device allocation of the library's own workspaces / arenas.  ST5_POISON=1 (debug, read once) fills every new allocation with
// 0xFF bytes -- NaN as bf16 and fp32, -1 as an index -- so that a kernel consuming workspace it never wrote shows up as NaN /
// a fault instead of as a run-to-run difference (tests/test_poison_gpu.py).
inline hipError_t st5_dev_malloc(void** p, size_t bytes) {
  static const bool poison = [] { const char* e = getenv("ST5_POISON"); return e && e[0] == '1'; }();
  hipError_t rc = hipMalloc(p, bytes);
  if (rc == hipSuccess && poison) rc = hipMemset(*p, 0xFF, bytes);
  return rc;
}
template <typename T> inline hipError_t st5_dev_malloc(T** p, size_t bytes) { return st5_dev_malloc(reinterpret_cast<void**>(p), bytes); }

#define HIP_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e_ = hipGetLastError();                       \
    if (e_ != hipSuccess) return ST5_ERR_LAUNCH;             \
  } while (0)
