// Generic MFMA GEMM for gfx950 (CDNA4): C = epilogue(alpha * A.B^T), A [M x K], B [N x K].
//
// One kernel family covers every matmul-shaped op of the SpeechT5 hot path through a generalised
// operand descriptor (include/speecht5_hip.h): nn.Linear fwd/dgrad/wgrad, strided/grouped Conv1d as
// implicit GEMM over channels-last activations, and the per-(batch,head) attention matmuls.
//
// Tiling (wave64): 256 threads = 4 waves (2x2), block tile 128x128, wave tile 64x64 = 2x2 MFMA
// 32x32 tiles, k-tile = 128 bytes per row (64 bf16 / 32 f32).  Operands are staged
// HBM -> registers -> LDS (double buffered, one barrier per k-tile).  The LDS image is
// [row][8 x 16B chunks] with chunk' = chunk ^ ((row>>1)&7): conflict-free for the ds_read_b128
// fragment reads of a 32-row MFMA operand (lane groups per MI355X_MICROARCH.md "LDS").
// K-strided ("transposed") operands are transposed in registers on the way into LDS, so the MFMA
// fragment reads are identical for all four layout combinations.
// bf16 uses v_mfma_f32_32x32x16_bf16; f32 uses v_mfma_f32_32x32x2_f32 (exact f32, parity mode).
// The epilogue goes through LDS so that bias/activation/residual/dropout are applied on 8
// consecutive columns per lane and stored as 16-byte vectors.
#include <type_traits>
#include "common.h"
#include "../../include/speecht5_hip.h"

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int TILE_BYTES = 128 * 128;  // one operand tile: 128 rows x 128 B
constexpr int EP_LD = 68;              // floats per staged epilogue row (64 + 4 pad)

// operand addressing, by value (no pointers into the kernarg struct => stays in SGPRs)
struct OpAddr {
  long long ld, bstride, seg_stride;
  int rpb, seg;
  __device__ __forceinline__ long long outer(int idx) const {
    return rpb ? (long long)(idx / rpb) * bstride + (long long)(idx % rpb) * ld : (long long)idx * ld;
  }
  __device__ __forceinline__ long long inner(int idx) const {
    return seg ? (long long)(idx / seg) * seg_stride + (idx % seg) : (long long)idx;
  }
};
__device__ __forceinline__ OpAddr make_addr(long long ld, long long bstride, long long seg_stride, int rpb, int seg) {
  OpAddr a; a.ld = ld; a.bstride = bstride; a.seg_stride = seg_stride; a.rpb = rpb; a.seg = seg;
  return a;
}
__device__ __forceinline__ long long z_off(long long zs0, long long zs1, int z, int zdiv) {
  return (long long)(z / zdiv) * zs0 + (long long)(z % zdiv) * zs1;
}
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// Linear block number (after the XCD remap: every XCD walks a contiguous range of them, ~32 x blocks-per-CU at a time) -> tile
// coordinates.  Row-major order makes the blocks that run together on an XCD cover few rows and ALL columns of tiles (64 blocks on
// 24 columns: 3 A panels + 24 B panels stream through that XCD's L2 at once); walking column groups GEMM_GROUP_N wide instead (the
// "grouped" order of every tiled GEMM) makes them cover an 8 x 8 patch: 16 panels for the same 64 tiles.
#ifndef GEMM_GROUP_N
#define GEMM_GROUP_N 8
#endif
__device__ __forceinline__ void tile_of(const int bid, const int tiles_n, const int ntiles, int& tm, int& tn) {
  if (GEMM_GROUP_N == 0 || tiles_n <= GEMM_GROUP_N) { tm = bid / tiles_n; tn = bid - tm * tiles_n; return; }
  const int tiles_m = ntiles / tiles_n;
  const int gsz = GEMM_GROUP_N * tiles_m;
  const int g = bid / gsz, r = bid - g * gsz;
  const int tn0 = g * GEMM_GROUP_N;
  const int gw = tiles_n - tn0 < GEMM_GROUP_N ? tiles_n - tn0 : GEMM_GROUP_N;
  tm = r / gw; tn = tn0 + (r - tm * gw);
}

__device__ __forceinline__ unsigned int elem_bits(float v) { return __float_as_uint(v); }
__device__ __forceinline__ unsigned int elem_bits(bf16_t v) {
  return (unsigned int)__builtin_bit_cast(unsigned short, v);
}
// gather nvalid (< NE) consecutive elements into one 16-byte register, zero-filling the rest
template <typename T, int NE>
__device__ __forceinline__ u32x4 pack_tail(const T* p, int nvalid) {
  u32x4 r = {0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    if (e < nvalid) {
      if constexpr (sizeof(T) == 4) r[e] = elem_bits(p[e]);
      else r[e >> 1] |= elem_bits(p[e]) << ((e & 1) * 16);
    }
  }
  return r;
}

// ---------------- K-major operand: rows contiguous along k ----------------
template <typename T>
struct LoaderKM {
  static constexpr int VEC = Elem<T>::VEC;
  const T* base;
  OpAddr ad;
  long long rowoff[4];
  int chunk, rbase, K;
  u32x4 r[4];
  __device__ __forceinline__ void init(const OpAddr& a, const T* b, int row0, int limit, int K_, int tid) {
    ad = a; base = b; K = K_;
    chunk = tid & 7; rbase = tid >> 3;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int row = row0 + rbase + 32 * p;
      row = row < limit ? row : limit - 1;
      rowoff[p] = ad.outer(row);
    }
  }
  __device__ __forceinline__ void load(int k0) {
    const int k = k0 + chunk * VEC;
    if (k + VEC <= K) {
      const long long ko = ad.inner(k);
#pragma unroll
      for (int p = 0; p < 4; ++p) r[p] = *reinterpret_cast<const u32x4*>(base + rowoff[p] + ko);
    } else if (k < K) {  // K tail (a VEC group never straddles a segment: seg % VEC == 0)
      const long long ko = ad.inner(k);
#pragma unroll
      for (int p = 0; p < 4; ++p) r[p] = pack_tail<T, VEC>(base + rowoff[p] + ko, K - k);
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p) r[p] = u32x4{0, 0, 0, 0};
    }
  }
  __device__ __forceinline__ void store(char* tile) const {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = rbase + 32 * p;
      *reinterpret_cast<u32x4*>(tile + lds_off(row, chunk)) = r[p];
    }
  }
};

// ---------------- K-strided operand: rows contiguous along the row index ----------------
// Thread block = VEC k-values x 4 rows; transposed in registers, written as 4 x 16 B.
template <typename T>
struct LoaderKS {
  static constexpr int VEC = Elem<T>::VEC;
  static constexpr int W = 4 * sizeof(T) / 4;  // dwords per 4-row load (2 for bf16, 4 for f32)
  const T* base;  // already offset by the inner (row) index
  OpAddr ad;
  int kb, rb, K, nvalid;  // nvalid = valid rows of this thread's 4-row group (0..4)
  unsigned int g[VEC][W];
  __device__ __forceinline__ void init(const OpAddr& a, const T* b, int row0, int limit, int K_, int tid) {
    ad = a; K = K_;
    kb = tid >> 5; rb = tid & 31;
    const int i0 = row0 + rb * 4;
    nvalid = limit - i0; nvalid = nvalid > 4 ? 4 : (nvalid < 0 ? 0 : nvalid);
    base = b + ad.inner(nvalid > 0 ? i0 : 0);
  }
  __device__ __forceinline__ void load(int k0) {
    const int kf = k0 + kb * VEC;
    int q = 0, rm = kf;
    if (ad.rpb) { q = kf / ad.rpb; rm = kf % ad.rpb; }
#pragma unroll
    for (int kk = 0; kk < VEC; ++kk) {
      const int k = kf + kk;
      const long long oo = ad.rpb ? (long long)q * ad.bstride + (long long)rm * ad.ld : (long long)k * ad.ld;
      if (k < K && nvalid == 4) {
        if constexpr (W == 2) {
          const u32x2 v = *reinterpret_cast<const u32x2*>(base + oo);
          g[kk][0] = v[0]; g[kk][1] = v[1];
        } else {
          const u32x4 v = *reinterpret_cast<const u32x4*>(base + oo);
          g[kk][0] = v[0]; g[kk][1] = v[1]; g[kk][2] = v[2]; g[kk][3] = v[3];
        }
      } else if (k < K && nvalid > 0) {
        const u32x4 tw = pack_tail<T, 4>(base + oo, nvalid);
#pragma unroll
        for (int w = 0; w < W; ++w) g[kk][w] = tw[w];
      } else {
#pragma unroll
        for (int w = 0; w < W; ++w) g[kk][w] = 0u;
      }
      if (ad.rpb) { if (++rm == ad.rpb) { rm = 0; ++q; } }
    }
  }
  __device__ __forceinline__ void store(char* tile) const {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      u32x4 o;
      if constexpr (W == 2) {  // bf16: pick half (rr&1) of dword (rr>>1) from 8 k-rows
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const unsigned int a = g[2 * w][rr >> 1], b = g[2 * w + 1][rr >> 1];
          o[w] = (rr & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
        }
      } else {
#pragma unroll
        for (int w = 0; w < 4; ++w) o[w] = g[w][rr];
      }
      const int row = rb * 4 + rr;
      *reinterpret_cast<u32x4*>(tile + lds_off(row, kb)) = o;
    }
  }
};

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };

template <typename T>
__device__ __forceinline__ void mma(const typename Frag<T>::type& a, const typename Frag<T>::type& b, f32x16& c);
template <>
__device__ __forceinline__ void mma<bf16_t>(const bf16x8& a, const bf16x8& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma<float>(const f32x4& a, const f32x4& b, f32x16& c) {
#pragma unroll
  for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], c, 0, 0, 0);
}


template <typename T> __device__ __forceinline__ typename Frag<T>::type ones_frag();
template <> __device__ __forceinline__ bf16x8 ones_frag<bf16_t>() {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16_t)1.0f;
  return o;
}
template <> __device__ __forceinline__ f32x4 ones_frag<float>() { f32x4 o = {1.f, 1.f, 1.f, 1.f}; return o; }

// Row sums held in column 0 of the two 32x32 "A x ones" accumulators of a wave (rows mbase .. mbase+63).  Exactly one wave
// of the grid owns a given row (first tile column, wave column 0), so no atomics: without split-K the sums are added to
// asum[] in place; with split-K every split STORES its partial column behind the slabs (part = slabs + nsplit*M*N + split*M)
// and the slab-reduce kernel adds the splits in order -- the bias gradient is then bit-reproducible like the rest of the step.
__device__ __forceinline__ void flush_asum(float* dst, bool accumulate, const f32x16& s0, const f32x16& s1, int mbase, int M, int lane) {
  if ((lane & 31) != 0) return;
  const int hi = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (mbase + row < M) dst[mbase + row] = accumulate ? dst[mbase + row] + s0[r] : s0[r];
    if (mbase + 32 + row < M) dst[mbase + 32 + row] = accumulate ? dst[mbase + 32 + row] + s1[r] : s1[r];
  }
}
__device__ __forceinline__ float* asum_target(const st5_gemm_params& p, bool& accumulate) {
  accumulate = gridDim.y == 1;
  if (accumulate) return p.asum;
  return reinterpret_cast<float*>(const_cast<void*>(p.C.ptr)) + (long long)gridDim.y * p.M * p.N + (long long)blockIdx.y * p.M;
}

// predicated (static-index) tail accessors: arrays stay in registers
template <typename U>
__device__ __forceinline__ void load_vec(const U* p, bool full, int ne, float (&v)[8]) {
  if (full) load8f<U>(p, v);
  else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = e < ne ? Elem<U>::to_f(p[e]) : 0.f;
  }
}
template <typename U>
__device__ __forceinline__ void store_vec(U* p, bool full, int ne, const float (&v)[8]) {
  if (full) store8f<U>(p, v);
  else {
#pragma unroll
    for (int e = 0; e < 8; ++e) if (e < ne) p[e] = Elem<U>::from_f(v[e]);
  }
}

// One lane's share of an MX block: 8 consecutive bf16 of a row (4 lanes = one 32-element block, aligned in the wave) -> 8 e4m3 bytes
// and, from the block's first lane, the e8m0 scale byte (common.h mx8_quant: the rule, the non-finite handling and the fast path).
__device__ __forceinline__ void mx8_quant8(const float (&v)[8], unsigned char* __restrict__ q8, unsigned char* __restrict__ sbyte, const bool leader) {
  unsigned int q[2], sc;
  mx8_quant<8>(v, q, sc);
  u32x2 o;
  o[0] = q[0]; o[1] = q[1];
  *reinterpret_cast<u32x2*>(q8) = o;
  if (leader) *sbyte = (unsigned char)sc;
}

struct EpiArgs {  // everything the epilogue needs, by value
  void* C; const void* R; const void* P; void* Cpre; const float* bias;
  long long c_ld, c_bs, r_ld, r_bs, p_ld, p_bs, q_ld, q_bs;
  int rpb, M, N, act, out_f32, dact, c_vec_ok, atomic, fast;
  float alpha, beta, dropout_p;
  unsigned long long seed, ctr_base;
  // F_Q8 instantiations only (fp8 mode): the MX-fp8 image of the bf16 output for the NEXT fp8 GEMM's A operand, written beside it
  unsigned char* q8; unsigned char* s8; long long q8_ld, s8_ld;
};

__device__ __forceinline__ void stage_write(float* stage, const f32x16& acc0, const f32x16& acc1, const int lane) {
  const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * fhalf;
    stage[row * EP_LD + frow] = acc0[r];
    stage[row * EP_LD + 32 + frow] = acc1[r];
  }
}

// Applies the fused epilogue to one staged 32x64 half of the wave tile.  Kept as a rolled loop
// (the activation code is large); called from a rolled loop over the two halves.
template <typename T>
__device__ __forceinline__ void epilogue_rows(const EpiArgs& ea, const float* stage, const int mbase, const int nbase,
                                              const int lane) {
  const unsigned int thresh = ea.dropout_p > 0.f ? dropout_thresh(ea.dropout_p) : 0u;
  const float inv_keep = ea.dropout_p > 0.f ? 1.f / (1.f - ea.dropout_p) : 1.f;
#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {
    const int row = pass * 8 + (lane >> 3), cc = (lane & 7) * 8;
    const int gm = mbase + row, gn = nbase + cc;
    if (gm < ea.M && gn < ea.N) {
      float v[8];
      {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(stage + row * EP_LD + cc);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(stage + row * EP_LD + cc + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
      }
      const int ne = (ea.N - gn) < 8 ? (ea.N - gn) : 8;
      const bool full = (ne == 8) && ea.c_vec_ok;
      // C-class operands share the row split `rpb` (each has its own ld / block stride)
      int q = 0, rm = gm;
      if (ea.rpb) { q = gm / ea.rpb; rm = gm % ea.rpb; }
      const long long co = (long long)q * ea.c_bs + (long long)rm * ea.c_ld + gn;
      float rv[8], pv[8], ov[8], pre[8];
      if (ea.R) {
        const long long ro = (long long)q * ea.r_bs + (long long)rm * ea.r_ld + gn;
        if (ea.out_f32) load_vec<float>(reinterpret_cast<const float*>(ea.R) + ro, full, ne, rv);
        else load_vec<T>(reinterpret_cast<const T*>(ea.R) + ro, full, ne, rv);
      }
      if (ea.dact) {
        const long long po = (long long)q * ea.p_bs + (long long)rm * ea.p_ld + gn;
        load_vec<T>(reinterpret_cast<const T*>(ea.P) + po, full, ne, pv);
      }
      if (ea.beta != 0.f) {
        if (ea.out_f32) load_vec<float>(reinterpret_cast<const float*>(ea.C) + co, full, ne, ov);
        else load_vec<T>(reinterpret_cast<const T*>(ea.C) + co, full, ne, ov);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = v[e] * ea.alpha;
        if (ea.bias && e < ne) x += ea.bias[gn + e];
        pre[e] = x;
        // DACT: result *= act'(P) (forward activation not applied); else forward activation.
        v[e] = ea.dact ? x * act_grad_f(ea.act, pv[e]) : act_f(ea.act, x);
      }
      if (ea.dropout_p > 0.f) {
        const unsigned long long ctr = ea.ctr_base + (unsigned long long)gm * (unsigned long long)ea.N + gn;
        float dsc[8];  // ctr is a multiple of 8 (N % 8 == 0, gn % 8 == 0)
        dropout_scale8(ea.seed, ctr, thresh, inv_keep, dsc);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= dsc[e];
      }
      if (ea.R) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if (ea.beta != 0.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += ea.beta * ov[e];
      }
      if (ea.Cpre) {
        const long long qo = (long long)q * ea.q_bs + (long long)rm * ea.q_ld + gn;
        store_vec<T>(reinterpret_cast<T*>(ea.Cpre) + qo, full, ne, pre);
      }
      if (ea.out_f32) store_vec<float>(reinterpret_cast<float*>(ea.C) + co, full, ne, v);
      else store_vec<T>(reinterpret_cast<T*>(ea.C) + co, full, ne, v);
    }
  }
}


// Fast epilogue of a wave's 64x64 tile for the common layout (every C-class operand 16-byte
// aligned, N % 8 == 0 so a lane's 8 columns are all valid or all out of range).  Both 32-row halves are staged through the
// wave-private LDS slab; the per-lane constants (bias, column offsets) are loaded once, the uniform feature tests are
// per 8-vector, and the four row passes of a half are unrolled so their LDS reads / global loads overlap.
// FEAT < 0: every epilogue feature is tested at run time (uniform branches; ALL of the code is in the kernel).  FEAT >= 0: a bit
// mask of the features this instantiation has, everything else is compiled out.  Why: the run-time form of the 128^2 NT kernel is
// 18.6 k instructions (~130 KB against a 64 KB instruction cache shared by two CUs); a tile's epilogue walks through all of it
// while the other resident block is in its k-loop.  Measured: removing only the tanh / LeakyReLU cases from the activation switch
// made the step's NT launches 4.5 % faster; the hot combinations of the training step therefore get kernels of their own
// (nt_feat_of() picks, the run-time form stays the fallback).
enum { F_GELU = 1, F_DACT = 2, F_DROP = 4, F_RES = 8, F_PRE = 16, F_BETA = 32, F_Q8 = 64 };
// F_Q8 (round 6, fp8 mode; compile-time only): the output tile is also MX-quantised in the epilogue -- a lane holds 8 consecutive
// columns of a row, 4 lanes one 32-element MX block (tile columns start at multiples of 64) -- exactly as st5_quant_mxfp8 would
// quantise the bf16 output (the ROUNDED values are quantised: same bytes), so the consumer GEMM needs no quantisation pass: the
// K = 4096 passes over the FFN's hidden activations and their gradients were the largest of them (38-76 us per GEMM at Large B = 32).
template <typename T, typename OUT, int FEAT = -1, int HALVES = 2>   // (HALVES = 1: a 32 x 64 wave tile, acc00 / acc01 only)
__device__ __forceinline__ void epilogue_fast(const EpiArgs& ea, float* stage, const f32x16& acc00, const f32x16& acc01,
                                              const f32x16& acc10, const f32x16& acc11, const int mbase, const int nbase,
                                              const int lane) {
  // no mul + add contraction ACROSS the feature steps below: in the run-time form they sit in different basic blocks and are never
  // fused; a specialised instantiation has them in straight-line code -- with contraction it would round (x * keep + residual)
  // once instead of twice and stop being bit-identical to the run-time form
#pragma clang fp contract(off)
  constexpr bool FASTACT = sizeof(T) == 2;   // bf16 compute mode
  constexpr bool RT = FEAT < 0;
  const int cc = (lane & 7) * 8, rsub = lane >> 3;
  const int gn = nbase + cc;
  const bool col_ok = gn < ea.N;
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
  if (ea.bias && col_ok) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(ea.bias + gn);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(ea.bias + gn + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bias8[e] = b0[e]; bias8[4 + e] = b1[e]; }
  }
  const bool drop = RT ? ea.dropout_p > 0.f : (FEAT & F_DROP) != 0;
  const unsigned int thresh = drop ? dropout_thresh(ea.dropout_p) : 0u;
  const float inv_keep = drop ? 1.f / (1.f - ea.dropout_p) : 1.f;
  const unsigned int seedf = drop ? drop_seed_fold(ea.seed) : 0u;      // (once per tile: see dropout_scale8_folded)
  OUT* const Cb = reinterpret_cast<OUT*>(ea.C) + gn;
  const bool has_res = RT ? ea.R != nullptr : (FEAT & F_RES) != 0;
  const bool has_pre = RT ? ea.Cpre != nullptr : (FEAT & F_PRE) != 0;
  const bool dact = RT ? ea.dact != 0 : (FEAT & F_DACT) != 0;
  const bool has_beta = RT ? ea.beta != 0.f : (FEAT & F_BETA) != 0;
  const int act = RT ? ea.act : ((FEAT & (F_GELU | F_DACT)) ? ACT_GELU : ACT_NONE);
  const OUT* const Rb = has_res ? reinterpret_cast<const OUT*>(ea.R) + gn : nullptr;
  const T* const Pb = dact ? reinterpret_cast<const T*>(ea.P) + gn : nullptr;
  T* const Qb = has_pre ? reinterpret_cast<T*>(ea.Cpre) + gn : nullptr;
  // Row offsets (round 6).  A lane's rows are mbase + rsub + rr with rr = 32 h + 8 (2 pp + k) the SAME for every lane: the per-lane part
  // (row0 * ld: one 64-bit multiply per operand) is formed once per call, the per-row part rr * ld on the scalar unit.  The first form
  // multiplied row * ld per operand and row (and row * N for the dropout counter) on the VALU: ~34 quarter-rate integer multiplies per
  // 16 output elements, a third of the epilogue's issue time (the ISA of the bias + dropout + residual epilogue: 352 instructions per
  // two row passes).  Row-split operands (convolution layouts, rpb != 0) keep the division per row.
  const bool split = ea.rpb != 0;
  const long long row0 = (long long)mbase + rsub;
  long long co0 = row0 * ea.c_ld, ro0 = has_res ? row0 * ea.r_ld : 0, po0 = dact ? row0 * ea.p_ld : 0, qo0 = has_pre ? row0 * ea.q_ld : 0;
  unsigned long long ctr0 = drop ? ea.ctr_base + (unsigned long long)row0 * (unsigned long long)ea.N + (unsigned long long)gn : 0ull;
  // (opaque to the optimiser: hipcc otherwise re-associates co0 + rr * ld back into (row0 + rr) * ld -- one register less, the multiplies back)
  asm volatile("" : "+v"(co0), "+v"(ro0), "+v"(po0), "+v"(qo0), "+v"(ctr0));
#pragma unroll
  for (int h = 0; h < HALVES; ++h) {
    if (h == 0) stage_write(stage, acc00, acc01, lane);
    else stage_write(stage, acc10, acc11, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int pp = 0; pp < 2; ++pp) {   // two row passes at a time (keeps the register footprint of the epilogue small)
      float v[2][8];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int row = (pp * 2 + k) * 8 + rsub;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(stage + row * EP_LD + cc);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(stage + row * EP_LD + cc + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[k][e] = x0[e]; v[k][4 + e] = x1[e]; }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int rr = h * 32 + (pp * 2 + k) * 8;          // (uniform)
        const int gm = mbase + rr + rsub;
        if (gm < ea.M && col_ok) {
          long long c_off, r_off, p_off, q_off;
          if (split) {
            // row -> (block q, row rm inside the block) when the C-class operands are split every `rpb` rows
            // (convolution layouts: per-utterance halo rows); every operand has its own row / block stride
            const long long q = gm / ea.rpb, rm = gm - q * ea.rpb;
            c_off = q * ea.c_bs + rm * ea.c_ld; r_off = q * ea.r_bs + rm * ea.r_ld;
            p_off = q * ea.p_bs + rm * ea.p_ld; q_off = q * ea.q_bs + rm * ea.q_ld;
          } else {
            c_off = co0 + (long long)rr * ea.c_ld; r_off = ro0 + (long long)rr * ea.r_ld;
            p_off = po0 + (long long)rr * ea.p_ld; q_off = qo0 + (long long)rr * ea.q_ld;
          }
          float rv[8], pv[8], ov[8];
          if (has_res) load8f<OUT>(Rb + r_off, rv);
          if (dact) load8f<T>(Pb + p_off, pv);
          if (has_beta) load8f<OUT>(Cb + c_off, ov);
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fmaf(v[k][e], ea.alpha, bias8[e]);
          if (has_pre) store8f<T>(Qb + q_off, x);
          if (dact) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] *= act_grad_f<FASTACT>(act, pv[e]);
          } else if (act != ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = act_f<FASTACT>(act, x[e]);
          }
          if (drop) {
            float dsc[8];
            dropout_scale8_folded(seedf, ctr0 + (unsigned long long)rr * (unsigned long long)ea.N, thresh, inv_keep, dsc);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] *= dsc[e];
          }
          if (has_res) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += rv[e];
          }
          if (has_beta) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = fmaf(ea.beta, ov[e], x[e]);
          }
          store8f<OUT>(Cb + c_off, x);
          if constexpr (FEAT >= 0 && (FEAT & F_Q8) != 0) {
            float xr[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) xr[e] = Elem<OUT>::to_f(Elem<OUT>::from_f(x[e]));
            mx8_quant8(xr, ea.q8 + (long long)gm * ea.q8_ld + gn, ea.s8 + (long long)gm * ea.s8_ld + (gn >> 5), (lane & 3) == 0);
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab is rewritten by the next half
  }
}

template <typename T, int FEAT = -1, typename OUT = T, int HALVES = 2>
__device__ __forceinline__ void run_epilogue(const EpiArgs& ea, float* stage, const f32x16& acc00, const f32x16& acc01,
                                             const f32x16& acc10, const f32x16& acc11, const int mbase, const int nbase,
                                             const int lane) {
  if constexpr (FEAT >= 0) {     // (the launcher only picks a feature-specialised kernel for the fast layout with OUT outputs)
    epilogue_fast<T, OUT, FEAT, HALVES>(ea, stage, acc00, acc01, acc10, acc11, mbase, nbase, lane);
    return;
  }
  if (ea.fast) {
    if (ea.out_f32) epilogue_fast<T, float, -1, HALVES>(ea, stage, acc00, acc01, acc10, acc11, mbase, nbase, lane);
    else epilogue_fast<T, T, -1, HALVES>(ea, stage, acc00, acc01, acc10, acc11, mbase, nbase, lane);
    return;
  }
#pragma unroll 1
  for (int h = 0; h < HALVES; ++h) {
    if (h == 0) stage_write(stage, acc00, acc01, lane);
    else stage_write(stage, acc10, acc11, lane);
    // the stage slab is private to this wave: order its LDS writes/reads inside the wave only.  (A block barrier here
    // would also drain vmcnt, i.e. wait for the previous half's global stores.)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    epilogue_rows<T>(ea, stage, mbase + h * 32, nbase, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <typename T, bool AKS, bool BKS>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const st5_gemm_params p, const int c_vec_ok) {
  ST5_PAD_TO_256_VGPRS();
  constexpr int BK = 128 / (int)sizeof(T);
  typedef typename Frag<T>::type frag_t;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int z = blockIdx.z;
  const int tiles_n = (p.N + BN - 1) / BN;
  // XCD-aware remap (block b runs on XCD b % 8): give each XCD a contiguous range of tiles so that
  // neighbouring tiles (sharing an A row panel) hit the same L2.  Bijective for any grid size.
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, rmd = nwg & 7;
    bid = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (bid >> 3);
  }
  int tm, tn;
  tile_of(bid, tiles_n, (int)gridDim.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  const T* Ap = reinterpret_cast<const T*>(p.A.ptr) + z_off(p.A.zs0, p.A.zs1, z, p.zdiv);
  const T* Bp = reinterpret_cast<const T*>(p.B.ptr) + z_off(p.B.zs0, p.B.zs1, z, p.zdiv);

  typename std::conditional<AKS, LoaderKS<T>, LoaderKM<T>>::type la;
  typename std::conditional<BKS, LoaderKS<T>, LoaderKM<T>>::type lb;
  la.init(make_addr(p.A.ld, p.A.bstride, p.A.seg_stride, p.A.rpb, p.A.seg), Ap, m0, p.M, p.K, tid);
  lb.init(make_addr(p.B.ld, p.B.bstride, p.B.seg_stride, p.B.rpb, p.B.seg), Bp, n0, p.N, p.K, tid);

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

  // split-K: blockIdx.y owns k-tiles [kt0, kt0 + nk)
  const int nk_all = (p.K + BK - 1) / BK;
  const int per = (nk_all + gridDim.y - 1) / gridDim.y;
  const int kt0 = blockIdx.y * per;
  const int nk = (nk_all - kt0) < per ? (nk_all - kt0) : per;
  if (nk <= 0) return;
  la.load(kt0 * BK); lb.load(kt0 * BK);
  la.store(smem); lb.store(smem + TILE_BYTES);
  __syncthreads();

  const int frow = lane & 31, fhalf = lane >> 5;
  const int arow0 = wr * 64 + frow, brow0 = wc * 64 + frow;
  const bool do_asum = p.asum != nullptr && tn == 0 && wc == 0;   // bias-gradient column: first column of tiles only
  f32x16 sum0, sum1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { sum0[r] = 0.f; sum1[r] = 0.f; }
  for (int kt = 0; kt < nk; ++kt) {
    const char* cur = smem + (kt & 1) * 2 * TILE_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
    if (kt + 1 < nk) { la.load((kt0 + kt + 1) * BK); lb.load((kt0 + kt + 1) * BK); }
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const frag_t a0 = *reinterpret_cast<const frag_t*>(cur + lds_off(arow0, kg * 2 + fhalf));
      const frag_t a1 = *reinterpret_cast<const frag_t*>(cur + lds_off(arow0 + 32, kg * 2 + fhalf));
      const frag_t b0 = *reinterpret_cast<const frag_t*>(cur + TILE_BYTES + lds_off(brow0, kg * 2 + fhalf));
      const frag_t b1 = *reinterpret_cast<const frag_t*>(cur + TILE_BYTES + lds_off(brow0 + 32, kg * 2 + fhalf));
      mma<T>(a0, b0, acc00); mma<T>(a0, b1, acc01); mma<T>(a1, b0, acc10); mma<T>(a1, b1, acc11);
      if (do_asum) { mma<T>(a0, ones_frag<T>(), sum0); mma<T>(a1, ones_frag<T>(), sum1); }
    }
    if (kt + 1 < nk) { la.store(nxt); lb.store(nxt + TILE_BYTES); }
    __syncthreads();
  }
  if (do_asum) { bool acc_; float* dst_ = asum_target(p, acc_); flush_asum(dst_, acc_, sum0, sum1, m0 + wr * 64, p.M, lane); }

  // ------------------------------ epilogue ------------------------------
  EpiArgs ea;
  ea.C = const_cast<void*>(p.C.ptr); ea.R = p.R.ptr; ea.P = p.P.ptr; ea.Cpre = const_cast<void*>(p.Cpre.ptr);
  ea.bias = p.bias ? p.bias + (long long)z * p.bias_zs : nullptr;
  ea.c_ld = p.C.ld; ea.c_bs = p.C.bstride; ea.r_ld = p.R.ld; ea.r_bs = p.R.bstride;
  ea.p_ld = p.P.ld; ea.p_bs = p.P.bstride; ea.q_ld = p.Cpre.ld; ea.q_bs = p.Cpre.bstride;
  ea.rpb = p.C.rpb; ea.M = p.M; ea.N = p.N; ea.act = p.act;
  ea.out_f32 = (p.flags & ST5_GEMM_OUT_F32) != 0; ea.dact = (p.flags & ST5_GEMM_DACT) != 0; ea.c_vec_ok = c_vec_ok;
  ea.atomic = 0;
  ea.fast = c_vec_ok && (p.N % 8 == 0);
  if (gridDim.y > 1) {  // split-K: each split writes its own fp32 slab (p.C describes slab 0, slabs are M*N apart)
    ea.C = reinterpret_cast<float*>(ea.C) + (long long)blockIdx.y * p.M * p.N;
  }
  ea.alpha = p.alpha; ea.beta = p.beta; ea.dropout_p = p.dropout_p; ea.seed = p.seed;
  ea.ctr_base = (unsigned long long)z * (unsigned long long)p.M * (unsigned long long)p.N;
  // fold the batch offsets into the base pointers
  {
    const long long zc = z_off(p.C.zs0, p.C.zs1, z, p.zdiv);
    if (ea.out_f32) ea.C = reinterpret_cast<float*>(ea.C) + zc; else ea.C = reinterpret_cast<T*>(ea.C) + zc;
    if (ea.R) {
      const long long zr = z_off(p.R.zs0, p.R.zs1, z, p.zdiv);
      if (ea.out_f32) ea.R = reinterpret_cast<const float*>(ea.R) + zr; else ea.R = reinterpret_cast<const T*>(ea.R) + zr;
    }
    if (ea.P) ea.P = reinterpret_cast<const T*>(ea.P) + z_off(p.P.zs0, p.P.zs1, z, p.zdiv);
    if (ea.Cpre) ea.Cpre = reinterpret_cast<T*>(ea.Cpre) + z_off(p.Cpre.zs0, p.Cpre.zs1, z, p.zdiv);
  }
  float* stage = reinterpret_cast<float*>(smem) + wave * (32 * EP_LD);
  run_epilogue<T>(ea, stage, acc00, acc01, acc10, acc11, m0 + wr * 64, n0 + wc * 64, lane);
}

// C[m, n] = beta * C[m, n] + sum_s slab[s][m][n]   (slabs dense [M, N] fp32; C fp32 with row mapping)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C,
                                                            int nsplit, int M, int N, long long ldc, int rpb,
                                                            long long bstride, float beta, float* __restrict__ asum) {
  const long long nv = (long long)M * N / 4;  // N % 4 == 0 guaranteed by the host
  const long long slab = (long long)M * N;
  if (asum) {   // bias-gradient column: the splits' partial columns sit behind the slabs
    const float* part = slabs + (long long)nsplit * slab;
    for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long long)gridDim.x * blockDim.x) {
      float a = part[m];
      for (int s2 = 1; s2 < nsplit; ++s2) a += part[(long long)s2 * M + m];
      asum[m] += a;
    }
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    const long long e0 = i * 4;
    const int m = (int)(e0 / N), n = (int)(e0 % N);
    f32x4 acc = *reinterpret_cast<const f32x4*>(slabs + e0);
    for (int s2 = 1; s2 < nsplit; ++s2) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(slabs + s2 * slab + e0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += v[e];
    }
    const long long co = (rpb ? (long long)(m / rpb) * bstride + (long long)(m % rpb) * ldc : (long long)m * ldc) + n;
    f32x4* dst = reinterpret_cast<f32x4*>(C + co);
    if (beta != 0.f) {
      const f32x4 o = *dst;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += beta * o[e];
    }
    *dst = acc;
  }
}

// Deferred split-K reduction: with st5_gemm_defer_splitk(1) the slab reductions of consecutive weight-gradient GEMMs are
// not launched one by one; their descriptors queue up (each GEMM keeps its slabs alive in a bump arena) and
// st5_gemm_flush_splitk() folds all of them into their outputs with ONE launch (up to MR_MAX per launch).
constexpr int MR_MAX = 48;
struct MrDesc { const float* slabs; float* C; float* asum; long long ldc; int nsplit, M, N; float beta; int blk0; int pad; };
struct MrArgs { MrDesc d[MR_MAX]; int n; };
__global__ __launch_bounds__(256) void splitk_multi_reduce_kernel(const MrArgs a) {
  int j = 0;
  while (j + 1 < a.n && a.d[j + 1].blk0 <= (int)blockIdx.x) ++j;
  const MrDesc d = a.d[j];
  const int nblk = (j + 1 < a.n ? a.d[j + 1].blk0 : (int)gridDim.x) - d.blk0;
  const long long nv = (long long)d.M * d.N / 4, slab = (long long)d.M * d.N;
  if (d.asum) {
    const float* part = d.slabs + (long long)d.nsplit * slab;
    for (long long m = (long long)(blockIdx.x - d.blk0) * 256 + threadIdx.x; m < d.M; m += (long long)nblk * 256) {
      float a = part[m];
      for (int s2 = 1; s2 < d.nsplit; ++s2) a += part[(long long)s2 * d.M + m];
      d.asum[m] += a;
    }
  }
  for (long long i = (long long)(blockIdx.x - d.blk0) * 256 + threadIdx.x; i < nv; i += (long long)nblk * 256) {
    const long long e0 = i * 4;
    const int m = (int)(e0 / d.N), n = (int)(e0 % d.N);
    f32x4 acc = *reinterpret_cast<const f32x4*>(d.slabs + e0);
    for (int s2 = 1; s2 < d.nsplit; ++s2) acc += *reinterpret_cast<const f32x4*>(d.slabs + s2 * slab + e0);
    f32x4* dst = reinterpret_cast<f32x4*>(d.C + (long long)m * d.ldc + n);
    if (d.beta != 0.f) acc += d.beta * *dst;
    *dst = acc;
  }
}

bool g_defer = false;
// Deferred-reduction state, ONE PER STREAM: the queued descriptors belong to GEMMs of that stream and are folded by a launch on
// that stream (the backward passes of two micro-batches may run on two streams at once; a flush on one of them must not read
// slabs the other stream is still writing).  Each state has its own bump arena.
struct DeferState {
  hipStream_t stream;
  MrArgs pending;            // pending.n descriptors queued
  int pending_blocks;
  float* arena;              // slab arena for deferred reductions (grown between flushes only)
  size_t arena_bytes, arena_used;
};
constexpr int DEFER_STREAMS = 32;
DeferState g_dstates[DEFER_STREAMS] = {};
int g_ndstates = 0;
DeferState* defer_state(hipStream_t s, bool create) {
  for (int i = 0; i < g_ndstates; ++i)
    if (g_dstates[i].stream == s) return &g_dstates[i];
  if (!create) return nullptr;
  if (g_ndstates == DEFER_STREAMS) {
    // Stream churn (a test session that keeps creating streams): start the table over -- only when nothing is queued on
    // any state and after the device has drained, the rule of the LayerNorm table (norm.hip).  A state is never handed from
    // one stream to another while either may still have work in flight: the second micro-batch of a side-by-side update
    // would otherwise write its slabs into the arena the first one's queued reduction still has to read.
    for (int i = 0; i < g_ndstates; ++i)
      if (g_dstates[i].pending.n != 0) return nullptr;
    if (hipDeviceSynchronize() != hipSuccess) return nullptr;
    g_ndstates = 0;   // (the slots keep their arenas)
  }
  DeferState* d = &g_dstates[g_ndstates++];
  d->stream = s; d->pending.n = 0; d->pending_blocks = 0; d->arena_used = 0;
  return d;
}
#define g_pending (ds->pending)
#define g_pending_blocks (ds->pending_blocks)
#define g_arena (ds->arena)
#define g_arena_bytes (ds->arena_bytes)
#define g_arena_used (ds->arena_used)

int flush_state(DeferState* ds) {
  if (!ds || g_pending.n == 0) return ST5_OK;
  hipLaunchKernelGGL(splitk_multi_reduce_kernel, dim3((unsigned)g_pending_blocks), dim3(256), 0, ds->stream, g_pending);
  g_pending.n = 0; g_pending_blocks = 0; g_arena_used = 0;
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
int flush_pending(hipStream_t s) { return flush_state(defer_state(s, false)); }
// slab space for one deferred GEMM, or nullptr when the arena must grow (caller flushes, grows, retries)
float* arena_take(DeferState* ds, size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (g_arena_used + bytes > g_arena_bytes) return nullptr;
  float* p = reinterpret_cast<float*>(reinterpret_cast<char*>(g_arena) + g_arena_used);
  g_arena_used += bytes;
  return p;
}

// split-K slab workspace, one per stream (the weight-gradient stream runs split-K GEMMs beside the main stream's)
struct SlabWs { hipStream_t stream; float* ptr; size_t bytes; };
constexpr int SLAB_STREAMS = 8;
SlabWs g_slabs[SLAB_STREAMS] = {};
int g_nslabs = 0, g_slab_victim = 0;
float* slab_workspace(size_t bytes, hipStream_t stream) {
  SlabWs* w = nullptr;
  for (int i = 0; i < g_nslabs; ++i)
    if (g_slabs[i].stream == stream) w = &g_slabs[i];
  if (!w) {
    if (g_nslabs < SLAB_STREAMS) {
      w = &g_slabs[g_nslabs++];
      w->ptr = nullptr; w->bytes = 0;
    } else {
      // table full (streams come and go over a process's life): hand the oldest entry's allocation to the new stream.  The
      // previous owner's work may still be in flight, so drain the device once; this happens on stream churn only.
      w = &g_slabs[g_slab_victim];
      g_slab_victim = (g_slab_victim + 1) % SLAB_STREAMS;
      if (hipDeviceSynchronize() != hipSuccess) return nullptr;
    }
    w->stream = stream;
  }
  if (bytes > w->bytes) {
    if (w->ptr) (void)hipFree(w->ptr);  // synchronises with in-flight users
    w->ptr = nullptr;
    const size_t want = bytes < (size_t(64) << 20) ? (size_t(64) << 20) : bytes;
    if (st5_dev_malloc(&w->ptr, want) != hipSuccess) { w->bytes = 0; return nullptr; }
    w->bytes = want;
  }
  return w->ptr;
}

template <typename T>
int launch(const st5_gemm_params& p, int c_vec_ok, int nsplit, hipStream_t s) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  dim3 grid(tiles, nsplit, p.batch), block(NTHREADS);
  const bool aks = p.flags & ST5_GEMM_A_KSTRIDED, bks = p.flags & ST5_GEMM_B_KSTRIDED;
  if (!aks && !bks) hipLaunchKernelGGL((gemm_kernel<T, false, false>), grid, block, 0, s, p, c_vec_ok);
  else if (!aks && bks) hipLaunchKernelGGL((gemm_kernel<T, false, true>), grid, block, 0, s, p, c_vec_ok);
  else if (aks && !bks) hipLaunchKernelGGL((gemm_kernel<T, true, false>), grid, block, 0, s, p, c_vec_ok);
  else hipLaunchKernelGGL((gemm_kernel<T, true, true>), grid, block, 0, s, p, c_vec_ok);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}


// ------------------------------------------------------------------------------------------------------
// NT fast path: both operands K-major, K a multiple of the k-tile, no K segmentation.  Same tiling, MFMA
// fragments, swizzled LDS image and epilogue as gemm_kernel, but the operand tiles go HBM -> LDS directly
// (global_load_lds_dwordx4, 16 B per lane, no staging registers) through a 3-deep ring of LDS buffers with
// counted vmcnt waits and raw s_barrier, so two k-tiles of loads stay in flight across the barrier while the
// MFMAs of the current tile run (cdna_hip_programming.md section 5, "Pipelining across barriers").
// The LDS destination of an LDS-DMA is lane-linear (wave base + lane*16), so the XOR swizzle is applied to
// the per-lane SOURCE address instead: lane l of a wave-instruction covering rows r0..r0+7 loads logical chunk
// (l&7) ^ swz(r0 + l>>3) of its row, which is exactly what lands in physical chunk l&7.
// ------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

#ifndef GEMM_ABL
#define GEMM_ABL 0
#endif
#ifdef GEMM_TIMING
__device__ unsigned long long g_gemm_timing[8];
#define GPROBE(i) do { __builtin_amdgcn_s_waitcnt(0xC07F); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += (unsigned int)(t_ - tprev); tprev = t_; } while (0)
#else
#define GPROBE(i)
#endif

template <typename F, int OFF>
__device__ __forceinline__ F lds_read128_asm(const unsigned lds_addr) {
  static_assert(sizeof(F) == 16, "128-bit fragment");
  F v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
  return v;
}

// (cache-policy bits of the operand LDS-DMA loads, an A/B knob: 0 = default, 1 = sc0, 2 = nt, 16 = sc1; tools/r6/build_aux_lib.sh)
#ifndef GLDS_AUX_A
#define GLDS_AUX_A 0
#endif
#ifndef GLDS_AUX_B
#define GLDS_AUX_B 0
#endif
template <typename T, int NBUF, int FEAT = -1>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_glds_kernel(const st5_gemm_params p, const int c_vec_ok) {
  ST5_PAD_TO_256_VGPRS();
#ifdef GEMM_TIMING
  unsigned int tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
  constexpr int VEC = Elem<T>::VEC;
  constexpr int BK = 128 / (int)sizeof(T);
  typedef typename Frag<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) char dsm[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int z = blockIdx.z;
  const int tiles_n = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, rmd = nwg & 7;
    bid = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (bid >> 3);
  }
  int tm, tn;
  tile_of(bid, tiles_n, (int)gridDim.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const T* Ap = reinterpret_cast<const T*>(p.A.ptr) + z_off(p.A.zs0, p.A.zs1, z, p.zdiv);
  const T* Bp = reinterpret_cast<const T*>(p.B.ptr) + z_off(p.B.zs0, p.B.zs1, z, p.zdiv);
  const OpAddr aa = make_addr(p.A.ld, p.A.bstride, 0, p.A.rpb, 0);
  const OpAddr ab = make_addr(p.B.ld, p.B.bstride, 0, p.B.rpb, 0);

  const T* asrc[4];
  const T* bsrc[4];
  const int rsub = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wave) * 8 + rsub;
    const int c = pc ^ ((row >> 1) & 7);
    int gr = m0 + row; gr = gr < p.M ? gr : p.M - 1;
    asrc[i] = Ap + aa.outer(gr) + c * VEC;
    int gc = n0 + row; gc = gc < p.N ? gc : p.N - 1;
    bsrc[i] = Bp + ab.outer(gc) + c * VEC;
  }
  const int dst0 = wave * 1024;  // byte offset of this wave's 8-row slab inside a 4-slab group

  auto issue = [&](int kt, int buf) {
    char* base = dsm + buf * 2 * TILE_BYTES + dst0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(asrc[i] + (long long)kt * BK), (lds_ptr_t)(base + i * 4096), 16, 0, GLDS_AUX_A);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bsrc[i] + (long long)kt * BK), (lds_ptr_t)(base + TILE_BYTES + i * 4096), 16, 0, GLDS_AUX_B);
    }
  };
  // (NBUF == 5: five 16 KB operand slots instead of whole stages, see the k-loop)
  auto issue_a = [&](int kt, int slot) {
    char* base = dsm + slot * TILE_BYTES + dst0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(asrc[i] + (long long)kt * BK), (lds_ptr_t)(base + i * 4096), 16, 0, GLDS_AUX_A);
  };
  auto issue_b = [&](int kt, int slot) {
    char* base = dsm + slot * TILE_BYTES + dst0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bsrc[i] + (long long)kt * BK), (lds_ptr_t)(base + i * 4096), 16, 0, GLDS_AUX_B);
  };

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

  const int nk = p.K / BK;
  if constexpr (NBUF == 5) {
    issue_a(0, 0); issue_b(0, 1);
    if (nk > 1) issue_a(1, 2);
  } else {
#pragma unroll
    for (int t = 0; t < NBUF - 1; ++t)
      if (t < nk) issue(t, t);
  }
  const int frow = lane & 31, fhalf = lane >> 5;
  const int arow0 = wr * 64 + frow, brow0 = wc * 64 + frow;
  const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)dsm;
  const unsigned a_addr0 = lds_off(arow0, fhalf), b_addr0 = lds_off(brow0, fhalf);
  GPROBE(0);
  // NBUF == 5 ("2.5 stages", 80 KB: two blocks still share a CU): operand tiles go through a ring of five 16 KB slots in the order
  // A(0) B(0) A(1) B(1) A(2) ..., tile number s in slot s % 5.  At the top of k-tile t the tiles 2t, 2t+1, 2t+2 are in flight; the wait
  // leaves only A(t+1) outstanding, the barrier retires everybody's reads of A(t-1) / B(t-1), whose slots B(t+1) and A(t+2) then
  // take: 48 KB per block in flight under the MFMAs (32 KB with two whole stages) and 16 KB instead of nothing across the wait --
  // the operand fill of a CU is latency-bound (bytes in flight / landing latency), section 4.
  int sa = 0;   // slot of A(kt); B(kt) sits in the next one
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (NBUF == 5) {
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      // tile kt has landed when at most the 8 loads of tile kt+1 are still outstanding
      int ahead = nk - 1 - kt;
      ahead = ahead < NBUF - 2 ? ahead : NBUF - 2;  // tiles allowed to stay in flight
      if (ahead >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (ahead == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    GPROBE(1);
#if GEMM_ABL != 2
    // (tried and dropped: the eight LDS-DMA instructions two at a time behind the four MFMA groups instead of back to back in front of
    // them -- 3-10 % slower on the wide-N shapes, profiles/r4_gemm_nt_experiments.txt)
    if constexpr (NBUF == 5) {
      if (kt + 1 < nk) issue_b(kt + 1, sa + 3 >= 5 ? sa - 2 : sa + 3);
      if (kt + 2 < nk) issue_a(kt + 2, sa + 4 >= 5 ? sa - 1 : sa + 4);
    } else {
      if (kt + NBUF - 1 < nk) issue(kt + NBUF - 1, (kt + NBUF - 1) % NBUF);
    }
#endif
    // Fragments double-buffered in registers: the four reads of k-group kg + 1 are issued before the MFMAs of group kg, and each group
    // waits only for its own reads (LDS returns in order: lgkmcnt(4) = "all but the four newest").  Written as inline asm because hipcc
    // keeps the source order "read, wait for everything, multiply" per group otherwise (and, given both sets, still waits with
    // lgkmcnt(0)) -- one exposed LDS round trip per k-group instead of one per k-tile.  addr(kg) = addr(0) ^ (kg << 5): the swizzle
    // XORs the chunk index, the k-group is bits 1-2 of it.
    unsigned pa, pb;
    if constexpr (NBUF == 5) {
      const int sb = sa + 1 >= 5 ? 0 : sa + 1;
      pa = lds_base + sa * TILE_BYTES + a_addr0; pb = lds_base + sb * TILE_BYTES + b_addr0;
      sa = sa + 2 >= 5 ? sa - 3 : sa + 2;
    } else {
      const unsigned cb = lds_base + (kt % NBUF) * 2 * TILE_BYTES;
      pa = cb + a_addr0; pb = cb + TILE_BYTES + b_addr0;
    }
    frag_t fa0[2], fa1[2], fb0[2], fb1[2];
#define NT_READ(S, KG)                                                                                      \
  fa0[S] = lds_read128_asm<frag_t, 0>(pa ^ ((KG) << 5)); fb0[S] = lds_read128_asm<frag_t, 0>(pb ^ ((KG) << 5));          \
  fa1[S] = lds_read128_asm<frag_t, 4096>(pa ^ ((KG) << 5)); fb1[S] = lds_read128_asm<frag_t, 4096>(pb ^ ((KG) << 5));
#define NT_WAIT(S, CNT)                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(fa0[S]), "+v"(fa1[S]), "+v"(fb0[S]), "+v"(fb1[S]) :: "memory");
#if GEMM_ABL == 3
#define NT_MMA(S) asm volatile("" :: "v"(fa0[S]), "v"(fa1[S]), "v"(fb0[S]), "v"(fb1[S]));
#else
#define NT_MMA(S)                                                                                           \
  mma<T>(fa0[S], fb0[S], acc00); mma<T>(fa0[S], fb1[S], acc01); mma<T>(fa1[S], fb0[S], acc10); mma<T>(fa1[S], fb1[S], acc11);
#endif
    NT_READ(0, 0)
    NT_READ(1, 1)
    NT_WAIT(0, 4)
    NT_MMA(0) __builtin_amdgcn_sched_barrier(0);
    NT_READ(0, 2)
    NT_WAIT(1, 4)
    NT_MMA(1) __builtin_amdgcn_sched_barrier(0);
    NT_READ(1, 3)
    NT_WAIT(0, 4)
    NT_MMA(0) __builtin_amdgcn_sched_barrier(0);
    NT_WAIT(1, 0)
    NT_MMA(1) __builtin_amdgcn_sched_barrier(0);
#undef NT_READ
#undef NT_WAIT
#undef NT_MMA
    GPROBE(2);
  }
  __syncthreads();
  GPROBE(3);

  EpiArgs ea;
  ea.C = const_cast<void*>(p.C.ptr); ea.R = p.R.ptr; ea.P = p.P.ptr; ea.Cpre = const_cast<void*>(p.Cpre.ptr);
  ea.bias = p.bias ? p.bias + (long long)z * p.bias_zs : nullptr;
  ea.c_ld = p.C.ld; ea.c_bs = p.C.bstride; ea.r_ld = p.R.ld; ea.r_bs = p.R.bstride;
  ea.p_ld = p.P.ld; ea.p_bs = p.P.bstride; ea.q_ld = p.Cpre.ld; ea.q_bs = p.Cpre.bstride;
  ea.rpb = p.C.rpb; ea.M = p.M; ea.N = p.N; ea.act = p.act;
  ea.out_f32 = (p.flags & ST5_GEMM_OUT_F32) != 0; ea.dact = (p.flags & ST5_GEMM_DACT) != 0; ea.c_vec_ok = c_vec_ok;
  ea.atomic = 0;
  ea.fast = c_vec_ok && (p.N % 8 == 0);
  ea.alpha = p.alpha; ea.beta = p.beta; ea.dropout_p = p.dropout_p; ea.seed = p.seed;
  ea.ctr_base = (unsigned long long)z * (unsigned long long)p.M * (unsigned long long)p.N;
  {
    const long long zc = z_off(p.C.zs0, p.C.zs1, z, p.zdiv);
    if (ea.out_f32) ea.C = reinterpret_cast<float*>(ea.C) + zc; else ea.C = reinterpret_cast<T*>(ea.C) + zc;
    if (ea.R) {
      const long long zr = z_off(p.R.zs0, p.R.zs1, z, p.zdiv);
      if (ea.out_f32) ea.R = reinterpret_cast<const float*>(ea.R) + zr; else ea.R = reinterpret_cast<const T*>(ea.R) + zr;
    }
    if (ea.P) ea.P = reinterpret_cast<const T*>(ea.P) + z_off(p.P.zs0, p.P.zs1, z, p.zdiv);
    if (ea.Cpre) ea.Cpre = reinterpret_cast<T*>(ea.Cpre) + z_off(p.Cpre.zs0, p.Cpre.zs1, z, p.zdiv);
  }
  float* stage = reinterpret_cast<float*>(dsm) + wave * (32 * EP_LD);
#if GEMM_ABL == 1
  asm volatile("" :: "v"(acc00), "v"(acc01), "v"(acc10), "v"(acc11));
  if (ea.alpha == 123.f)
#endif
  run_epilogue<T, FEAT>(ea, stage, acc00, acc01, acc10, acc11, m0 + wr * 64, n0 + wc * 64, lane);
#ifdef GEMM_TIMING
  GPROBE(4);
  if (lane == 0) { for (int i = 0; i < 5; ++i) atomicAdd(&g_gemm_timing[i], (unsigned long long)tacc[i]); atomicAdd(&g_gemm_timing[7], 1ull); }
#endif
}


// ------------------------------------------------------------------------------------------------------
// NT fast path, 64 x 128 block tile (round 6): the transformer's Linear GEMMs at B = 8 are 120-1536 tiles of 128^2 on a chip with 512
// slots for them (two 64 KB blocks per CU), i.e. 0.25-3 rounds, and a block's k-step is bound by the LDS-DMA landing latency, not by
// its MFMAs (~1.2 us per pair of resident blocks against 0.43 us of MFMA time): an under-filled or unevenly filled round costs a whole
// round (8192 x 768 x 3072: 384 tiles, the CUs holding two of them set the time while the others idle after one; 3992 x 768 x K: 192
// tiles, one per CU on 75 % of the chip, nothing covering their latency).  Half-height tiles double the number of blocks and a block
// needs 48 KB of LDS and ~100 registers, so THREE fit on a CU (768 slots, 3 waves per SIMD): 384 tiles become 768 = exactly one round
// on every CU, 192 become 378 = every CU busy with one or two.  Same skeleton as gemm_nt_glds_kernel<bf16, 2>: 4 waves as 2 (m) x 2
// (n), wave tile 32 x 64 = 1 x 2 MFMA tiles of v_mfma_f32_32x32x16_bf16, k-tile 64, two-stage LDS-DMA ring (A 8 KB + B 16 KB per
// stage), source-side XOR swizzle, register-double-buffered asm fragment reads, wave-private epilogue slab.  Every output element
// sees the same MFMA chain (k-tiles ascending, four 16-deep groups each) as in the 128^2 and 256^2 kernels: bit-identical results.
// Costs: 1.5x the LDS fragment bytes per flop (3 reads per 2 MFMAs instead of 4 per 4) and 1.33x the L2 -> LDS bytes per flop.
// ------------------------------------------------------------------------------------------------------
constexpr int M64_A_BYTES = 64 * 128, M64_STAGE = M64_A_BYTES + TILE_BYTES;

template <int FEAT>
__global__ __launch_bounds__(NTHREADS, 3) void gemm_nt_m64_kernel(const st5_gemm_params p, const int c_vec_ok) {
  typedef bf16_t T;
  constexpr int VEC = 8, BK = 64;
  typedef bf16x8 frag_t;
  extern __shared__ __attribute__((aligned(16))) char dsm[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int z = blockIdx.z;
  const int tiles_n = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, rmd = nwg & 7;
    bid = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (bid >> 3);
  }
  int tm, tn;
  tile_of(bid, tiles_n, (int)gridDim.x, tm, tn);
  const int m0 = tm * 64, n0 = tn * BN;
  const T* Ap = reinterpret_cast<const T*>(p.A.ptr) + z_off(p.A.zs0, p.A.zs1, z, p.zdiv);
  const T* Bp = reinterpret_cast<const T*>(p.B.ptr) + z_off(p.B.zs0, p.B.zs1, z, p.zdiv);
  const OpAddr aa = make_addr(p.A.ld, p.A.bstride, 0, p.A.rpb, 0);
  const OpAddr ab = make_addr(p.B.ld, p.B.bstride, 0, p.B.rpb, 0);

  const T* asrc[2];
  const T* bsrc[4];
  const int rsub = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wave) * 8 + rsub;
    const int c = pc ^ ((row >> 1) & 7);
    if (i < 2) {
      int gr = m0 + row; gr = gr < p.M ? gr : p.M - 1;
      asrc[i] = Ap + aa.outer(gr) + c * VEC;
    }
    int gc = n0 + row; gc = gc < p.N ? gc : p.N - 1;
    bsrc[i] = Bp + ab.outer(gc) + c * VEC;
  }
  const int dst0 = wave * 1024;
  auto issue = [&](int kt, int buf) {
    char* base = dsm + buf * M64_STAGE + dst0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(asrc[i] + (long long)kt * BK), (lds_ptr_t)(base + i * 4096), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bsrc[i] + (long long)kt * BK), (lds_ptr_t)(base + M64_A_BYTES + i * 4096), 16, 0, 0);
  };

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

  const int nk = p.K / BK;
  issue(0, 0);
  const int frow = lane & 31, fhalf = lane >> 5;
  const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)dsm;
  const unsigned a_addr0 = lds_off(wr * 32 + frow, fhalf), b_addr0 = M64_A_BYTES + lds_off(wc * 64 + frow, fhalf);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile kt has landed (this wave's pieces; the barrier publishes everybody's)
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
    const unsigned cb = lds_base + (kt & 1) * M64_STAGE;
    const unsigned pa = cb + a_addr0, pb = cb + b_addr0;
    frag_t fa[2], fb0[2], fb1[2];
#define M64_READ(S, KG)                                                                                     \
  fa[S] = lds_read128_asm<frag_t, 0>(pa ^ ((KG) << 5)); fb0[S] = lds_read128_asm<frag_t, 0>(pb ^ ((KG) << 5));           \
  fb1[S] = lds_read128_asm<frag_t, 4096>(pb ^ ((KG) << 5));
#define M64_WAIT(S, CNT)                                                                                    \
  asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(fa[S]), "+v"(fb0[S]), "+v"(fb1[S]) :: "memory");
#define M64_MMA(S) mma<T>(fa[S], fb0[S], acc0); mma<T>(fa[S], fb1[S], acc1);
    M64_READ(0, 0)
    M64_READ(1, 1)
    M64_WAIT(0, 3)
    M64_MMA(0) __builtin_amdgcn_sched_barrier(0);
    M64_READ(0, 2)
    M64_WAIT(1, 3)
    M64_MMA(1) __builtin_amdgcn_sched_barrier(0);
    M64_READ(1, 3)
    M64_WAIT(0, 3)
    M64_MMA(0) __builtin_amdgcn_sched_barrier(0);
    M64_WAIT(1, 0)
    M64_MMA(1) __builtin_amdgcn_sched_barrier(0);
#undef M64_READ
#undef M64_WAIT
#undef M64_MMA
  }
  __syncthreads();

  EpiArgs ea;
  ea.C = const_cast<void*>(p.C.ptr); ea.R = p.R.ptr; ea.P = p.P.ptr; ea.Cpre = const_cast<void*>(p.Cpre.ptr);
  ea.bias = p.bias ? p.bias + (long long)z * p.bias_zs : nullptr;
  ea.c_ld = p.C.ld; ea.c_bs = p.C.bstride; ea.r_ld = p.R.ld; ea.r_bs = p.R.bstride;
  ea.p_ld = p.P.ld; ea.p_bs = p.P.bstride; ea.q_ld = p.Cpre.ld; ea.q_bs = p.Cpre.bstride;
  ea.rpb = p.C.rpb; ea.M = p.M; ea.N = p.N; ea.act = p.act;
  ea.out_f32 = (p.flags & ST5_GEMM_OUT_F32) != 0; ea.dact = (p.flags & ST5_GEMM_DACT) != 0; ea.c_vec_ok = c_vec_ok;
  ea.atomic = 0;
  ea.fast = c_vec_ok && (p.N % 8 == 0);
  ea.alpha = p.alpha; ea.beta = p.beta; ea.dropout_p = p.dropout_p; ea.seed = p.seed;
  ea.ctr_base = (unsigned long long)z * (unsigned long long)p.M * (unsigned long long)p.N;
  {
    const long long zc = z_off(p.C.zs0, p.C.zs1, z, p.zdiv);
    if (ea.out_f32) ea.C = reinterpret_cast<float*>(ea.C) + zc; else ea.C = reinterpret_cast<T*>(ea.C) + zc;
    if (ea.R) {
      const long long zr = z_off(p.R.zs0, p.R.zs1, z, p.zdiv);
      if (ea.out_f32) ea.R = reinterpret_cast<const float*>(ea.R) + zr; else ea.R = reinterpret_cast<const T*>(ea.R) + zr;
    }
    if (ea.P) ea.P = reinterpret_cast<const T*>(ea.P) + z_off(p.P.zs0, p.P.zs1, z, p.zdiv);
    if (ea.Cpre) ea.Cpre = reinterpret_cast<T*>(ea.Cpre) + z_off(p.Cpre.zs0, p.Cpre.zs1, z, p.zdiv);
  }
  float* stage = reinterpret_cast<float*>(dsm) + wave * (32 * EP_LD);
  run_epilogue<T, FEAT, T, 1>(ea, stage, acc0, acc1, acc0, acc1, m0 + wr * 32, n0 + wc * 64, lane);
}


// ------------------------------------------------------------------------------------------------------
// NT fast path, 256 x 256 block tile: 512 threads = 8 waves as 2 (m) x 4 (n), wave tile 128 x 64 = 4 x 2 MFMA tiles.
// Why: the 128^2 kernel above pays one LDS-DMA landing latency (~1.1 us issue -> landed on a busy chip) plus one
// barrier + LDS round trip per k-step of a block, so a CU retires 2 blocks x (128 x 128 x 64) MACs per ~1.2 us (~38 %
// MFMA busy), and its 64 flop/B tile needs the whole L1 fill rate at full MFMA speed.  Here the k-step is 64 bytes per
// row (32 bf16 / 16 f32), a stage is 2 x 256 rows x 64 B = 32 KB and the ring is 4 stages (128 KB, one block per CU):
// loads run THREE stages ahead (two stages = 64 KB always in flight, every stage has two full k-steps to land), the
// barrier of step kt publishes stage kt+1, so the first fragments of the next stage are read before the last MFMAs of
// the current one and no LDS round trip sits behind a barrier.  128 flop/B: half the L2 -> LDS bytes per flop.
// LDS image: [row][4 x 16 B chunks], chunk' = chunk ^ ((row >> 2) & 3) -- conflict-free for the ds_read_b128 lane
// groups of a 32-row fragment read (MI355X_MICROARCH.md "LDS": groups {0-3,12-15,20-27}, {4-11,16-19,28-31}).
// ------------------------------------------------------------------------------------------------------
constexpr int BM2 = 256, BN2 = 256, NT2 = 512;
constexpr int ROW2 = 64;                       // bytes per tile row and stage
constexpr int TILE2_BYTES = 256 * ROW2;        // one operand, one stage: 16 KB
constexpr int STAGE2_BYTES = 2 * TILE2_BYTES;  // A + B
constexpr int NSTAGE2 = 4;

__device__ __forceinline__ int lds_off2(int row, int chunk) { return row * ROW2 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <typename T>
__global__ __launch_bounds__(NT2) void gemm_nt256_kernel(const st5_gemm_params p, const int c_vec_ok) {
#ifdef GEMM_TIMING
  unsigned int tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
  constexpr int VEC = Elem<T>::VEC;
  constexpr int BK = ROW2 / (int)sizeof(T);
  typedef typename Frag<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) char dsm[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int z = blockIdx.z;
  const int tiles_n = (p.N + BN2 - 1) / BN2;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, rmd = nwg & 7;
    bid = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (bid >> 3);
  }
  int tm, tn;
  tile_of(bid, tiles_n, (int)gridDim.x, tm, tn);
  const int m0 = tm * BM2, n0 = tn * BN2;
  const T* Ap = reinterpret_cast<const T*>(p.A.ptr) + z_off(p.A.zs0, p.A.zs1, z, p.zdiv);
  const T* Bp = reinterpret_cast<const T*>(p.B.ptr) + z_off(p.B.zs0, p.B.zs1, z, p.zdiv);
  const OpAddr aa = make_addr(p.A.ld, p.A.bstride, 0, p.A.rpb, 0);
  const OpAddr ab = make_addr(p.B.ld, p.B.bstride, 0, p.B.rpb, 0);

  // wave-instruction i of wave w fills tile rows (i*8 + w)*16 .. +15 (1 KB, lane-linear: lane l -> row + (l >> 2),
  // physical chunk l & 3); the swizzle is applied on the source side
  const T* asrc[2];
  const T* bsrc[2];
  const int rsub = lane >> 2, pc = lane & 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (i * 8 + wave) * 16 + rsub;
    const int c = pc ^ ((row >> 2) & 3);
    int gr = m0 + row; gr = gr < p.M ? gr : p.M - 1;
    asrc[i] = Ap + aa.outer(gr) + c * VEC;
    int gc = n0 + row; gc = gc < p.N ? gc : p.N - 1;
    bsrc[i] = Bp + ab.outer(gc) + c * VEC;
  }
  auto issue = [&](int kt) {
    char* base = dsm + (kt & (NSTAGE2 - 1)) * STAGE2_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(asrc[i] + (long long)kt * BK), (lds_ptr_t)(base + i * 8192), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bsrc[i] + (long long)kt * BK), (lds_ptr_t)(base + TILE2_BYTES + i * 8192), 16, 0, 0);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  const int frow = lane & 31, fhalf = lane >> 5;
  int aoff[2][4], boff[2][2];   // byte offsets of this lane's fragments inside a stage, per k-group
#pragma unroll
  for (int kg = 0; kg < 2; ++kg) {
#pragma unroll
    for (int i = 0; i < 4; ++i) aoff[kg][i] = lds_off2(wr * 128 + 32 * i + frow, kg * 2 + fhalf);
#pragma unroll
    for (int j = 0; j < 2; ++j) boff[kg][j] = TILE2_BYTES + lds_off2(wc * 64 + 32 * j + frow, kg * 2 + fhalf);
  }
  frag_t a0[4], b0[2], a1[4], b1[2];
  auto read0 = [&](const char* st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a0[i] = *reinterpret_cast<const frag_t*>(st + aoff[0][i]);
#pragma unroll
    for (int j = 0; j < 2; ++j) b0[j] = *reinterpret_cast<const frag_t*>(st + boff[0][j]);
  };
  auto read1 = [&](const char* st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a1[i] = *reinterpret_cast<const frag_t*>(st + aoff[1][i]);
#pragma unroll
    for (int j = 0; j < 2; ++j) b1[j] = *reinterpret_cast<const frag_t*>(st + boff[1][j]);
  };

#pragma unroll
  for (int t = 0; t < NSTAGE2 - 1; ++t)
    if (t < nk) issue(t);
  // stage 0 landed: at most the loads of stages 1 and 2 (4 per stage and lane) may be outstanding
  if (nk >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nk == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read0(dsm);
  GPROBE(0);
  for (int kt = 0; kt < nk; ++kt) {
    // publish stage kt+1 (own loads landed, then the barrier); the same barrier retires every wave's reads of stage kt-1,
    // whose buffer the loads of stage kt+3 overwrite
#if GEMM_ABL == 5      // throughput probe: loads + barriers only, 12 loads (3 stages) allowed in flight per lane (results invalid)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#elif GEMM_ABL == 6    // the same without any wait (the issue rate of the LDS-DMA path itself)
#else
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
    GPROBE(1);
#if GEMM_ABL != 2
    if (kt + NSTAGE2 - 1 < nk) issue(kt + NSTAGE2 - 1);
#endif
    const char* cur = dsm + (kt & (NSTAGE2 - 1)) * STAGE2_BYTES;
    const char* nxt = dsm + ((kt + 1) & (NSTAGE2 - 1)) * STAGE2_BYTES;
#if GEMM_ABL < 4
    read1(cur);
#endif
#if GEMM_ABL >= 4
    asm volatile("" :: "v"(cur), "v"(nxt));
#elif GEMM_ABL == 3
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(a0[i]));
    asm volatile("" :: "v"(b0[0]), "v"(b0[1]));
#else
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) mma<T>(a0[i], b0[j], acc[i][j]);
#endif
#if GEMM_ABL < 4
    read0(nxt);   // after the last stage: reads a stale (valid) buffer, unused
#endif
#if GEMM_ABL >= 4
#elif GEMM_ABL == 3
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(a1[i]));
    asm volatile("" :: "v"(b1[0]), "v"(b1[1]));
#else
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) mma<T>(a1[i], b1[j], acc[i][j]);
#endif
    GPROBE(2);
  }
  __syncthreads();
  GPROBE(3);

  EpiArgs ea;
  ea.C = const_cast<void*>(p.C.ptr); ea.R = p.R.ptr; ea.P = p.P.ptr; ea.Cpre = const_cast<void*>(p.Cpre.ptr);
  ea.bias = p.bias ? p.bias + (long long)z * p.bias_zs : nullptr;
  ea.c_ld = p.C.ld; ea.c_bs = p.C.bstride; ea.r_ld = p.R.ld; ea.r_bs = p.R.bstride;
  ea.p_ld = p.P.ld; ea.p_bs = p.P.bstride; ea.q_ld = p.Cpre.ld; ea.q_bs = p.Cpre.bstride;
  ea.rpb = p.C.rpb; ea.M = p.M; ea.N = p.N; ea.act = p.act;
  ea.out_f32 = (p.flags & ST5_GEMM_OUT_F32) != 0; ea.dact = (p.flags & ST5_GEMM_DACT) != 0; ea.c_vec_ok = c_vec_ok;
  ea.atomic = 0;
  ea.fast = c_vec_ok && (p.N % 8 == 0);
  ea.alpha = p.alpha; ea.beta = p.beta; ea.dropout_p = p.dropout_p; ea.seed = p.seed;
  ea.ctr_base = (unsigned long long)z * (unsigned long long)p.M * (unsigned long long)p.N;
  {
    const long long zc = z_off(p.C.zs0, p.C.zs1, z, p.zdiv);
    if (ea.out_f32) ea.C = reinterpret_cast<float*>(ea.C) + zc; else ea.C = reinterpret_cast<T*>(ea.C) + zc;
    if (ea.R) {
      const long long zr = z_off(p.R.zs0, p.R.zs1, z, p.zdiv);
      if (ea.out_f32) ea.R = reinterpret_cast<const float*>(ea.R) + zr; else ea.R = reinterpret_cast<const T*>(ea.R) + zr;
    }
    if (ea.P) ea.P = reinterpret_cast<const T*>(ea.P) + z_off(p.P.zs0, p.P.zs1, z, p.zdiv);
    if (ea.Cpre) ea.Cpre = reinterpret_cast<T*>(ea.Cpre) + z_off(p.Cpre.zs0, p.Cpre.zs1, z, p.zdiv);
  }
  float* stage = reinterpret_cast<float*>(dsm) + wave * (32 * EP_LD);
  run_epilogue<T>(ea, stage, acc[0][0], acc[0][1], acc[1][0], acc[1][1], m0 + wr * 128, n0 + wc * 64, lane);
  run_epilogue<T>(ea, stage, acc[2][0], acc[2][1], acc[3][0], acc[3][1], m0 + wr * 128 + 64, n0 + wc * 64, lane);
#ifdef GEMM_TIMING
  GPROBE(4);
  if (lane == 0) { for (int i = 0; i < 5; ++i) atomicAdd(&g_gemm_timing[i], (unsigned long long)tacc[i]); atomicAdd(&g_gemm_timing[7], 1ull); }
#endif
}


// ------------------------------------------------------------------------------------------------------
// NT fast path, 256 x 256 block tile, PHASED schedule (round 4; cdna_hip_programming.md section 5 "The 256^2 8-phase template",
// rebuilt on this file's 32x32x16 fragments, swizzle and epilogues).  512 threads = 8 waves as 2 (m) x 4 (n), wave tile 128 x 64,
// k-tile 64 (128-byte rows).  LDS: two buffers x four HALF-tiles (A rows 0-127, A rows 128-255, B rows 0-127, B rows 128-255:
// 128 rows x 128 B = 16 KB each, the 128^2 kernel's operand image) = 128 KB, one block per CU.
// A k-tile is FOUR phases, each = {fragment reads + ONE half-tile of LDS-DMA staging} | barrier | 8 MFMAs (one 64 x 32 quadrant
// of the wave tile over the whole k-tile) | barrier:
//   phase 1: reads B cols 0-31 (4) then ALL A fragments (16); stages B0(t+1);      MFMA acc[0..1][0]
//   phase 2:                                                   stages B1(t+1);      MFMA acc[2..3][0]
//   phase 3: reads B cols 32-63 (4);                           stages A0(t+2);      MFMA acc[2..3][1]
//   phase 4:                                                   stages A1(t+2); vmcnt(4);  MFMA acc[0..1][1]
// A slots are last read in phase 1 and restaged from phase 3 (two phases later), B slots last read in phase 3 and restaged from the
// next tile's phase 1; the ONE counted wait per k-tile (phase 4) leaves the two youngest half-tiles in flight and retires tile t+1
// completely, one barrier before its first read; every half-tile has >= 3 phases (~1000 cycles) to land.  Two waves share a SIMD
// (w and w + 4 = the two m-halves): with STAGGER the second half runs one barrier behind, so one wave of every SIMD is in its MFMA
// section while the other issues reads / DMA, and s_setprio(1) around the MFMA sections has something to arbitrate.
// ------------------------------------------------------------------------------------------------------
template <int FEAT, bool STAGGER>
__global__ __launch_bounds__(512) void gemm_nt8p_kernel(const st5_gemm_params p, const int c_vec_ok) {
  typedef bf16_t T;
  constexpr int BK = 64;
  constexpr int HALF = TILE_BYTES;              // 16 KB: 128 rows x 128 B
  constexpr int BUF = 4 * HALF;                 // A0 A1 B0 B1
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int z = blockIdx.z;
  const int tiles_n = (p.N + 255) / 256;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, rmd = nwg & 7;
    bid = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (bid >> 3);
  }
  int tm, tn;
  tile_of(bid, tiles_n, (int)gridDim.x, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
  const T* Ap = reinterpret_cast<const T*>(p.A.ptr) + z_off(p.A.zs0, p.A.zs1, z, p.zdiv);
  const T* Bp = reinterpret_cast<const T*>(p.B.ptr) + z_off(p.B.zs0, p.B.zs1, z, p.zdiv);
  const OpAddr aa = make_addr(p.A.ld, p.A.bstride, 0, p.A.rpb, 0);
  const OpAddr ab = make_addr(p.B.ld, p.B.bstride, 0, p.B.rpb, 0);

  // LDS-DMA sources: half-tile h (0: A rows 0-127, 1: A rows 128-255, 2: B rows 0-127, 3: B rows 128-255), instruction i of this
  // wave covers the half-tile's rows (i * 8 + wave) * 8 .. +7 (lane -> row + (lane >> 3), physical chunk lane & 7)
  const T* src[4][2];
  {
    const int rsub = lane >> 3, pc = lane & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (i * 8 + wave) * 8 + rsub;
      const int c = pc ^ ((row >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int gr = m0 + h * 128 + row; gr = gr < p.M ? gr : p.M - 1;
        src[h][i] = Ap + aa.outer(gr) + c * 8;
        int gc = n0 + h * 128 + row; gc = gc < p.N ? gc : p.N - 1;
        src[2 + h][i] = Bp + ab.outer(gc) + c * 8;
      }
    }
  }
  auto stage = [&](const int h, const int kt) {        // half-tile h of k-tile kt -> buffer kt & 1
    char* base = dsm + (kt & 1) * BUF + h * HALF + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[h][i] + (long long)kt * BK), (lds_ptr_t)(base + i * 8192), 16, 0, 0);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  const int frow = lane & 31, fhalf = lane >> 5;
  // fragment byte offsets inside a half-tile: row = 32 * blk + frow (the swizzle term only sees frow), one per k16 group
  int foff[4];
#pragma unroll
  for (int kg = 0; kg < 4; ++kg) foff[kg] = lds_off(frow, 2 * kg + fhalf);
  const int a_half = wr * HALF;                                    // this wave's A half-tile
  const int b_base = (2 + (wc >> 1)) * HALF + (wc & 1) * 64 * 128; // its 64 B rows inside its B half-tile

  // prologue: all of k-tile 0, the A halves of k-tile 1
  stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
  if (nk > 1) { stage(0, 1); stage(1, 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();

  bf16x8 af[4][4], bfr[4];
  for (int t = 0; t < nk; ++t) {
    const char* cur = dsm + (t & 1) * BUF;
    // ---- phase 1 ----
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) bfr[kg] = *reinterpret_cast<const bf16x8*>(cur + b_base + foff[kg]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) af[i][kg] = *reinterpret_cast<const bf16x8*>(cur + a_half + i * 4096 + foff[kg]);
    if (t + 1 < nk) stage(2, t + 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) { mma<T>(af[0][kg], bfr[kg], acc[0][0]); mma<T>(af[1][kg], bfr[kg], acc[1][0]); }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2 ----
    if (t + 1 < nk) stage(3, t + 1);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) { mma<T>(af[2][kg], bfr[kg], acc[2][0]); mma<T>(af[3][kg], bfr[kg], acc[3][0]); }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3 ----
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) bfr[kg] = *reinterpret_cast<const bf16x8*>(cur + b_base + 4096 + foff[kg]);
    if (t + 2 < nk) stage(0, t + 2);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) { mma<T>(af[2][kg], bfr[kg], acc[2][1]); mma<T>(af[3][kg], bfr[kg], acc[3][1]); }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 4 ----
    if (t + 2 < nk) { stage(1, t + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) { mma<T>(af[0][kg], bfr[kg], acc[0][1]); mma<T>(af[1][kg], bfr[kg], acc[1][1]); }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();     // (every wave executes the same number of barriers)
  __syncthreads();

  EpiArgs ea;
  ea.C = const_cast<void*>(p.C.ptr); ea.R = p.R.ptr; ea.P = p.P.ptr; ea.Cpre = const_cast<void*>(p.Cpre.ptr);
  ea.bias = p.bias ? p.bias + (long long)z * p.bias_zs : nullptr;
  ea.c_ld = p.C.ld; ea.c_bs = p.C.bstride; ea.r_ld = p.R.ld; ea.r_bs = p.R.bstride;
  ea.p_ld = p.P.ld; ea.p_bs = p.P.bstride; ea.q_ld = p.Cpre.ld; ea.q_bs = p.Cpre.bstride;
  ea.rpb = p.C.rpb; ea.M = p.M; ea.N = p.N; ea.act = p.act;
  ea.out_f32 = (p.flags & ST5_GEMM_OUT_F32) != 0; ea.dact = (p.flags & ST5_GEMM_DACT) != 0; ea.c_vec_ok = c_vec_ok;
  ea.atomic = 0;
  ea.fast = c_vec_ok && (p.N % 8 == 0);
  ea.alpha = p.alpha; ea.beta = p.beta; ea.dropout_p = p.dropout_p; ea.seed = p.seed;
  ea.ctr_base = (unsigned long long)z * (unsigned long long)p.M * (unsigned long long)p.N;
  {
    const long long zc = z_off(p.C.zs0, p.C.zs1, z, p.zdiv);
    if (ea.out_f32) ea.C = reinterpret_cast<float*>(ea.C) + zc; else ea.C = reinterpret_cast<T*>(ea.C) + zc;
    if (ea.R) {
      const long long zr = z_off(p.R.zs0, p.R.zs1, z, p.zdiv);
      if (ea.out_f32) ea.R = reinterpret_cast<const float*>(ea.R) + zr; else ea.R = reinterpret_cast<const T*>(ea.R) + zr;
    }
    if (ea.P) ea.P = reinterpret_cast<const T*>(ea.P) + z_off(p.P.zs0, p.P.zs1, z, p.zdiv);
    if (ea.Cpre) ea.Cpre = reinterpret_cast<T*>(ea.Cpre) + z_off(p.Cpre.zs0, p.Cpre.zs1, z, p.zdiv);
  }
  float* stg = reinterpret_cast<float*>(dsm) + wave * (32 * EP_LD);
  run_epilogue<T, FEAT>(ea, stg, acc[0][0], acc[0][1], acc[1][0], acc[1][1], m0 + wr * 128, n0 + wc * 64, lane);
  run_epilogue<T, FEAT>(ea, stg, acc[2][0], acc[2][1], acc[3][0], acc[3][1], m0 + wr * 128 + 64, n0 + wc * 64, lane);
}

int g_p8_stagger = 1;     // st5_gemm_set_nt_tile(3 / 4): phased 256^2 kernel with / without the half-phase stagger of the two m-halves
template <int FEAT>
int launch_nt8p_as(const st5_gemm_params& p, int c_vec_ok, dim3 grid, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_nt8p_kernel<FEAT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_nt8p_kernel<FEAT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return ST5_ERR_LAUNCH;
    attr = true;
  }
  if (g_p8_stagger) hipLaunchKernelGGL((gemm_nt8p_kernel<FEAT, true>), grid, dim3(512), (size_t)8 * TILE_BYTES, s, p, c_vec_ok);
  else hipLaunchKernelGGL((gemm_nt8p_kernel<FEAT, false>), grid, dim3(512), (size_t)8 * TILE_BYTES, s, p, c_vec_ok);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
int nt_feat_of(const st5_gemm_params& p, int c_vec_ok);
int launch_nt8p(const st5_gemm_params& p, int c_vec_ok, hipStream_t s) {
  const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  dim3 grid(tiles, 1, p.batch);
  switch (nt_feat_of(p, c_vec_ok)) {
    case 0: return launch_nt8p_as<0>(p, c_vec_ok, grid, s);
    case F_GELU | F_PRE: return launch_nt8p_as<F_GELU | F_PRE>(p, c_vec_ok, grid, s);
    case F_DROP | F_RES: return launch_nt8p_as<F_DROP | F_RES>(p, c_vec_ok, grid, s);
    case F_DACT: return launch_nt8p_as<F_DACT>(p, c_vec_ok, grid, s);
    case F_BETA: return launch_nt8p_as<F_BETA>(p, c_vec_ok, grid, s);
    case F_RES: return launch_nt8p_as<F_RES>(p, c_vec_ok, grid, s);
    default: return launch_nt8p_as<-1>(p, c_vec_ok, grid, s);
  }
}

template <typename T>
int launch_nt256(const st5_gemm_params& p, int c_vec_ok, hipStream_t s) {
  const int tiles = ((p.M + BM2 - 1) / BM2) * ((p.N + BN2 - 1) / BN2);
  dim3 grid(tiles, 1, p.batch), block(NT2);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_nt256_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_nt256_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return ST5_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((gemm_nt256_kernel<T>), grid, block, (size_t)NSTAGE2 * STAGE2_BYTES, s, p, c_vec_ok);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
int g_splitk_target = 384;   // blocks a split-K weight-gradient GEMM aims for (st5_gemm_set_splitk_target)
int g_nt_tile = 0;   // 0 = choose per problem, 1 = always 128^2, 2 = always 256^2 (A/B switch, st5_gemm_set_nt_tile)

#ifdef GEMM_TIMING
}  // namespace
extern "C" int st5_gemm_timing(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_timing), sizeof(g_gemm_timing)) != hipSuccess) return ST5_ERR_LAUNCH;
  if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_timing), z, sizeof(z)) != hipSuccess) return ST5_ERR_LAUNCH; }
  return ST5_OK;
}
namespace {
#endif


// Ring depth of the 128^2 NT kernel (template parameter NBUF).  2 stages (64 KB LDS) let two blocks share a CU and cover each
// other's LDS-DMA landing latency (~1 us per k-step against 0.25 us of MFMAs): the right shape for grids of several blocks
// per CU.  A grid of <= 256 blocks has the CU to itself; there a deeper ring (loads two or three k-steps ahead) helps a little:
// measured on the pre-training update with the micro-batches in turn 48.24 -> 47.33 ms (4 stages), 47.41 (3 stages); deep rings
// for EVERY grid were 10-25 % slower (round 1).  With the two micro-batches side by side (ddp.accumulate_overlapped) the
// other stream's blocks are the latency cover and 128 KB blocks keep them off the CU: 37.70 -> 38.27 ms, so that mode turns
// the deep ring off (st5_gemm_set_deep_ring(0, 2)).
int g_deep_blocks = 256, g_deep_nbuf = 4;
int g_nt_slots5 = 1;   // grids of more than g_deep_blocks blocks on five operand slots (80 KB, 2.5 k-steps of loads in flight) instead of two
                       // stages (64 KB); st5_gemm_set_nt_slots.  Per shape the difference is inside the noise (-4 % .. +3 %,
                       // profiles/r4_gemm_nt_slots.txt); on the whole update it was -0.4 .. -0.9 % in three same-box pairs on two boxes
                       // (profiles/r4_knob_ab.txt), so it is the default.  Results are bit-identical either way.

// The feature mask of a launch when one of the specialised instantiations covers it exactly, else -1 (run-time form).
// The hot combinations of the training step (transformer_layer.py / multihead_attention.py call sites through functional.py):
//   0                plain / bias          (QKV and cross-attention projections, plain data gradients)
//   F_GELU | F_PRE   fc1 forward           (bias, GELU, pre-activation kept for the backward)
//   F_DROP | F_RES   fc2 / output projection forward (bias, dropout, residual)
//   F_DACT           data gradient through the GELU (x act'(pre))
//   F_BETA           accumulating data gradient (dX += ...)
//   F_RES            data gradient of a post-LN block whose input is also its residual (dX = dH W + dY: 78 of the 87 launches of an
//                    update that still ran the run-time form in round 5, ST5_GEMM_FEAT_LOG=1)
int nt_feat_of(const st5_gemm_params& p, int c_vec_ok) {
  // ST5_GEMM_FEAT_LOG=1: one stderr line per launch that falls back to the run-time form (which combinations deserve a kernel?)
  static const bool log_rt = getenv("ST5_GEMM_FEAT_LOG") && getenv("ST5_GEMM_FEAT_LOG")[0] == '1';
  const bool dact = (p.flags & ST5_GEMM_DACT) != 0;
  if (!c_vec_ok || p.N % 8 != 0 || (p.flags & ST5_GEMM_OUT_F32) || (p.act != ACT_NONE && p.act != ACT_GELU) || (dact && p.act != ACT_GELU)) {
    if (log_rt) fprintf(stderr, "st5_gemm rt-epilogue layout vec=%d f32=%d act=%d dact=%d M=%d N=%d K=%d batch=%d\n", c_vec_ok,
                        (int)((p.flags & ST5_GEMM_OUT_F32) != 0), p.act, (int)dact, p.M, p.N, p.K, p.batch);
    return -1;
  }
  int f = 0;
  if (dact) f |= F_DACT;
  else if (p.act == ACT_GELU) f |= F_GELU;
  if (p.dropout_p > 0.f) f |= F_DROP;
  if (p.R.ptr) f |= F_RES;
  if (p.Cpre.ptr) f |= F_PRE;
  if (p.beta != 0.f) f |= F_BETA;
  switch (f) {
    case 0: case F_GELU | F_PRE: case F_DROP | F_RES: case F_DACT: case F_BETA: case F_RES: return f;
    default: break;
  }
  if (log_rt) fprintf(stderr, "st5_gemm rt-epilogue feat=%d act=%d M=%d N=%d K=%d batch=%d\n", f, p.act, p.M, p.N, p.K, p.batch);
  return -1;
}

template <typename T, int NBUF, int FEAT>
int launch_glds_as(const st5_gemm_params& p, int c_vec_ok, dim3 grid, hipStream_t s) {
  static bool attr = false;      // (one flag per instantiation)
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_nt_glds_kernel<T, NBUF, FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return ST5_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((gemm_nt_glds_kernel<T, NBUF, FEAT>), grid, dim3(NTHREADS), (size_t)(NBUF == 5 ? 5 : NBUF * 2) * TILE_BYTES, s, p, c_vec_ok);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

template <typename T, int NBUF>
int launch_glds_feat(const st5_gemm_params& p, int c_vec_ok, dim3 grid, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {      // (bf16: the training step; the fp32 parity mode keeps the one run-time form)
    static const bool rt_only = getenv("ST5_GEMM_RT_EPILOGUE") && getenv("ST5_GEMM_RT_EPILOGUE")[0] == '1';   // A/B switch
    switch (rt_only ? -1 : nt_feat_of(p, c_vec_ok)) {
      case 0: return launch_glds_as<T, NBUF, 0>(p, c_vec_ok, grid, s);
      case F_GELU | F_PRE: return launch_glds_as<T, NBUF, F_GELU | F_PRE>(p, c_vec_ok, grid, s);
      case F_DROP | F_RES: return launch_glds_as<T, NBUF, F_DROP | F_RES>(p, c_vec_ok, grid, s);
      case F_DACT: return launch_glds_as<T, NBUF, F_DACT>(p, c_vec_ok, grid, s);
      case F_BETA: return launch_glds_as<T, NBUF, F_BETA>(p, c_vec_ok, grid, s);
      case F_RES: return launch_glds_as<T, NBUF, F_RES>(p, c_vec_ok, grid, s);
      default: break;
    }
  }
  return launch_glds_as<T, NBUF, -1>(p, c_vec_ok, grid, s);
}

template <int FEAT>
int launch_m64_as(const st5_gemm_params& p, int c_vec_ok, dim3 grid, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_nt_m64_kernel<FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return ST5_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((gemm_nt_m64_kernel<FEAT>), grid, dim3(NTHREADS), (size_t)2 * M64_STAGE, s, p, c_vec_ok);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
int launch_m64(const st5_gemm_params& p, int c_vec_ok, hipStream_t s) {
  const int tiles = ((p.M + 63) / 64) * ((p.N + BN - 1) / BN);
  dim3 grid(tiles, 1, p.batch);
  switch (nt_feat_of(p, c_vec_ok)) {
    case 0: return launch_m64_as<0>(p, c_vec_ok, grid, s);
    case F_GELU | F_PRE: return launch_m64_as<F_GELU | F_PRE>(p, c_vec_ok, grid, s);
    case F_DROP | F_RES: return launch_m64_as<F_DROP | F_RES>(p, c_vec_ok, grid, s);
    case F_DACT: return launch_m64_as<F_DACT>(p, c_vec_ok, grid, s);
    case F_BETA: return launch_m64_as<F_BETA>(p, c_vec_ok, grid, s);
    case F_RES: return launch_m64_as<F_RES>(p, c_vec_ok, grid, s);
    default: return launch_m64_as<-1>(p, c_vec_ok, grid, s);
  }
}
// Which NT launches take the half-height tiles: bf16, and at most g_m64_max_tiles tiles of 128^2 (st5_gemm_set_m64_max_tiles; 0 = never)
int g_m64_max_tiles = 0;      // (default: never -- every isolated gain below was a LOSS inside the update)
bool nt_m64_pays(const st5_gemm_params& p) {
  static const int env = getenv("ST5_M64_MAX_TILES") ? atoi(getenv("ST5_M64_MAX_TILES")) : -1;   // (A/B switch for whole-step measurements)
  const int lim = env >= 0 ? env : g_m64_max_tiles;
  const long long t128 = (long long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.batch;
  return t128 <= lim && p.M > 64;
}

template <typename T>
int launch_glds(const st5_gemm_params& p, int c_vec_ok, hipStream_t s) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  dim3 grid(tiles, 1, p.batch);
  const int nk = p.K / (128 / (int)sizeof(T));
  const bool deep = sizeof(T) == 2 && (long long)tiles * p.batch <= g_deep_blocks && nk >= 4 && g_deep_nbuf > 2;
  if constexpr (sizeof(T) == 2) {
    if (deep && g_deep_nbuf == 4) return launch_glds_feat<T, 4>(p, c_vec_ok, grid, s);
    if (deep) return launch_glds_feat<T, 3>(p, c_vec_ok, grid, s);
    if (g_nt_slots5 && nk >= 2) return launch_glds_feat<T, 5>(p, c_vec_ok, grid, s);
  }
  return launch_glds_feat<T, 2>(p, c_vec_ok, grid, s);
}

bool g_use_glds = true;

bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// ------------------------------------------------------------------------------------------------------
// TN fast path (weight gradients: C[M,N] = A^T B with A = dY [K, M], B = X [K, N], the reduction index is the slow
// one in memory for BOTH operands).  bf16 only.  The operand tiles are copied HBM -> LDS untransposed with LDS-DMA
// ([64 k][128 m] per operand and k-tile, 16-byte chunks, chunk position XOR 4*(k&3)) and the MFMA fragments are
// produced by the gfx950 transpose read ds_read_b64_tr_b16: a 16-lane group reads a [4 k][16 m] block (lane i loads
// 4 m-consecutive bf16 of row k0 + (i>>2) at column 4*(i&3)) and every lane receives the 4 k-consecutive values of
// its column -- two reads make the 8-deep k fragment of v_mfma_f32_32x32x16_bf16.  No register staging, no
// register transposes (the register-staged gemm_kernel<.., true, true> needs 4x the load instructions and a
// 4x4 shuffle per fragment).  Rows past K and columns past M/N read from a zero page.  Split-K as gemm_kernel.
// ------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(256))) char g_zero_page[256];
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ bf16x8 tr_frag(const char* tile_row, int off0, int off1) {
  // tile_row + off: this lane's 8-byte source for the two 4-row halves of the fragment
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile_row + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile_row + off1));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// Transpose-read fragment as inline asm.  The builtin form makes hipcc treat the read as a possible alias of the in-flight LDS-DMA
// writes and put s_waitcnt vmcnt(0) in front of it -- in the 128^2 kernel that serialised a k-tile's fragment reads and MFMAs behind the NEXT
// tile's DMA (no overlap inside a block), in the phased kernel it drained the DMA queue twice per k-tile (4.5 us per k-tile against
// 1.8 us for the NT kernel's plain ds_read_b128).  The asm read is invisible to that analysis; the kernels'
// own vmcnt + barrier order the DMA against it, and their lgkmcnt waits (tied to the fragment registers) sit in front of the MFMAs.
template <int OFF>
__device__ __forceinline__ bf16x8 tr_frag_asm(const unsigned lds_addr) {
  s16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(lds_addr), "n"(OFF) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(lds_addr), "n"(OFF + 1024) : "memory");
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// The block program of the TN kernels: block bx of nwg (tile index before the XCD remap), split by of ny, batch z.
template <int FEAT>   // (epilogue features, see epilogue_fast; >= 0: fp32 output in the fast layout, compile-time feature set)
__device__ __forceinline__ void tn_glds_body(const st5_gemm_params& p, const int c_vec_ok, const int bx, const int nwg_, const int by,
                                             const int ny, const int z, const bool remap = true) {
  typedef bf16_t T;
  constexpr int BK = 64;
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  int bid = bx;
  if (remap) {     // (remap == false: the caller has placed this block with the XCDs in mind already, see gemm_tn_glds_kernel)
    const int nwg = nwg_, xcd = bid & 7, q = nwg >> 3, rmd = nwg & 7;
    bid = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (bid >> 3);
  }
  int tm, tn;
  tile_of(bid, tiles_n, nwg_, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const T* Ap = reinterpret_cast<const T*>(p.A.ptr) + z_off(p.A.zs0, p.A.zs1, z, p.zdiv);
  const T* Bp = reinterpret_cast<const T*>(p.B.ptr) + z_off(p.B.zs0, p.B.zs1, z, p.zdiv);

  // split-K range
  const int nk_all = (p.K + BK - 1) / BK;
  const int per = (nk_all + ny - 1) / ny;
  const int kt0 = by * per;
  const int nk = (kt0 + per <= nk_all ? per : nk_all - kt0) > 0 ? (kt0 + per <= nk_all ? per : nk_all - kt0) : 0;

  // LDS-DMA sources: wave-instruction i of this wave covers tile rows (i*4 + wave)*4 .. +3; lane l -> row + (l>>4),
  // physical chunk l&15, logical chunk (l&15) ^ 4*(l>>4)
  const int lrow = lane >> 4;
  const int lchunk = (lane & 15) ^ (4 * lrow);
  const bool a_col_ok = m0 + lchunk * 8 < p.M, b_col_ok = n0 + lchunk * 8 < p.N;
  const T* const zero = reinterpret_cast<const T*>(g_zero_page) + (lane & 15) * 8;
  // (segmented inner index -- grouped convolution windows: column i lives at (i / seg) * seg_stride + i % seg; a lane's 8-column
  // chunk never straddles a segment because the host requires seg % 8 == 0, and the column is fixed for the whole k-loop)
  auto inner_off = [](int i, int seg, long long ss) { return seg ? (long long)(i / seg) * ss + (i % seg) : (long long)i; };
  const T* abase = Ap + inner_off(m0 + lchunk * 8, p.A.seg, p.A.seg_stride);
  const T* bbase = Bp + inner_off(n0 + lchunk * 8, p.B.seg, p.B.seg_stride);
  // Source pointers advance by a per-lane constant stride (64 k-rows; 0 for the lanes that read the zero page because their
  // 16-byte column chunk lies past M / N), so a k-step issues its 8 LDS-DMA loads with two adds each.  (The first version
  // recomputed base + k * ld per load: 64-bit multiplies and two exec-masked branches per load, ~150 VALU instructions in
  // front of every k-step's MFMAs -- the kernel ran at 0.25 PFLOP/s where the same tiling does 0.5 in the NT form.)  Only a
  // final partial k-tile (K % 64 != 0) takes the per-row bounds-checked path.
  // Row split (rpb > 0: k-row r lives at (r / rpb) * bstride + (r % rpb) * ld -- the convolution weight gradients read the
  // haloed per-utterance activation layouts): each source keeps its row-in-block counter; stepping 64 rows wraps at most
  // once (the host requires rpb >= 64) and adds the block jump.  rpb == 0 behaves as one endless block.
  const T* ap[4];
  const T* bp[4];
  int ta[4], tb[4];
  const int rpa = p.A.rpb ? p.A.rpb : 0x7fffffff, rpbb = p.B.rpb ? p.B.rpb : 0x7fffffff;
  const long long astep = a_col_ok ? (long long)BK * p.A.ld : 0, bstep = b_col_ok ? (long long)BK * p.B.ld : 0;
  const long long awrap = (a_col_ok && p.A.rpb) ? p.A.bstride - (long long)p.A.rpb * p.A.ld : 0;
  const long long bwrap = (b_col_ok && p.B.rpb) ? p.B.bstride - (long long)p.B.rpb * p.B.ld : 0;
  auto row_off = [](long long k, int rpb, long long ld, long long bs) {
    if (!rpb) return k * ld;
    const long long q = k / rpb;
    return q * bs + (k - q * rpb) * ld;
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long k = (long long)kt0 * BK + (i * 4 + wave) * 4 + lrow;
    ap[i] = a_col_ok ? abase + row_off(k, p.A.rpb, p.A.ld, p.A.bstride) : zero;
    bp[i] = b_col_ok ? bbase + row_off(k, p.B.rpb, p.B.ld, p.B.bstride) : zero;
    ta[i] = p.A.rpb ? (int)(k % p.A.rpb) : 0;
    tb[i] = p.B.rpb ? (int)(k % p.B.rpb) : 0;
  }
  auto issue_fast = [&](int buf) {
    char* base = dsm + buf * 2 * TILE_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)ap[i], (lds_ptr_t)(base + i * 4096), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)bp[i], (lds_ptr_t)(base + TILE_BYTES + i * 4096), 16, 0, 0);
      ta[i] += BK; tb[i] += BK;
      const bool wa = ta[i] >= rpa, wb = tb[i] >= rpbb;
      ta[i] -= wa ? rpa : 0; tb[i] -= wb ? rpbb : 0;
      ap[i] += astep + (wa ? awrap : 0); bp[i] += bstep + (wb ? bwrap : 0);
    }
  };
  auto issue_tail = [&](int kt, int buf) {   // k-tile that crosses K: rows past K read the zero page
    char* base = dsm + buf * 2 * TILE_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kt * BK + (i * 4 + wave) * 4 + lrow;
      const bool kin = k < p.K;
      const T* sa = (kin && a_col_ok) ? abase + row_off(k, p.A.rpb, p.A.ld, p.A.bstride) : zero;
      const T* sb = (kin && b_col_ok) ? bbase + row_off(k, p.B.rpb, p.B.ld, p.B.bstride) : zero;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)sa, (lds_ptr_t)(base + i * 4096), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)sb, (lds_ptr_t)(base + TILE_BYTES + i * 4096), 16, 0, 0);
    }
  };
  auto issue = [&](int kt, int buf) {
    if ((kt + 1) * BK <= p.K) issue_fast(buf); else issue_tail(kt, buf);
  };

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

  if (nk > 0) issue(kt0, 0);
  // transpose-read addressing of this lane: group (g = m sub-block of 16, h = k half), i = l&15 -> j = i>>2 (k row), q = i&3
  const int g = (lane >> 4) & 1, h = lane >> 5, j = (lane & 15) >> 2, q = lane & 3;
  // byte offset inside a tile for (k = 8h + j [+4 for the second read] [+16 per k-step], m = mbase + 16g + 4q)
  auto frag_off = [&](int mbase) {
    const int m = mbase + 16 * g + 4 * q;
    const int pos = (m >> 3) ^ (4 * j);
    return (8 * h + j) * 256 + pos * 16 + ((m >> 2) & 1) * 8;
  };
  const int oa0 = frag_off(wr * 64), oa1 = frag_off(wr * 64 + 32);
  const int ob0 = frag_off(wc * 64), ob1 = frag_off(wc * 64 + 32);

  const bool do_asum = p.asum != nullptr && tn == 0 && wc == 0;   // bias-gradient column: first column of tiles only
  f32x16 sum0, sum1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { sum0[r] = 0.f; sum1[r] = 0.f; }
  // k-loop: the next tile's DMA is in flight while this tile is read and multiplied; fragments are double-buffered in registers (the reads
  // of k16-group kg + 1 are issued before the MFMAs of group kg, lgkmcnt(8) = "all but the 8 newest reads have landed").
  const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)dsm;
  bf16x8 fa0[2], fa1[2], fb0[2], fb1[2];
#define TN_READ(S, KG)                                                                                      \
  fa0[S] = tr_frag_asm<(KG) * 4096>(pa0); fb0[S] = tr_frag_asm<TILE_BYTES + (KG) * 4096>(pb0);             \
  fa1[S] = tr_frag_asm<(KG) * 4096>(pa1); fb1[S] = tr_frag_asm<TILE_BYTES + (KG) * 4096>(pb1);
#define TN_WAIT(S, CNT)                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(fa0[S]), "+v"(fa1[S]), "+v"(fb0[S]), "+v"(fb1[S]) :: "memory");
#define TN_MMA(S)                                                                                           \
  mma<T>(fa0[S], fb0[S], acc00); mma<T>(fa0[S], fb1[S], acc01); mma<T>(fa1[S], fb0[S], acc10); mma<T>(fa1[S], fb1[S], acc11); \
  if (do_asum) { mma<T>(fa0[S], ones_frag<T>(), sum0); mma<T>(fa1[S], ones_frag<T>(), sum1); }
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) issue(kt0 + kt + 1, (kt + 1) & 1);
    const unsigned cur = lds_base + (kt & 1) * 2 * TILE_BYTES;
    const unsigned pa0 = cur + oa0, pa1 = cur + oa1, pb0 = cur + ob0, pb1 = cur + ob1;
    TN_READ(0, 0)
    TN_READ(1, 1)
    TN_WAIT(0, 8)
    TN_MMA(0) __builtin_amdgcn_sched_barrier(0);
    TN_READ(0, 2)
    TN_WAIT(1, 8)
    TN_MMA(1) __builtin_amdgcn_sched_barrier(0);
    TN_READ(1, 3)
    TN_WAIT(0, 8)
    TN_MMA(0) __builtin_amdgcn_sched_barrier(0);
    TN_WAIT(1, 0)
    TN_MMA(1) __builtin_amdgcn_sched_barrier(0);
  }
#undef TN_READ
#undef TN_WAIT
#undef TN_MMA
  __syncthreads();
  if (do_asum) {   // (asum_target's rule with THIS block's split number: in place without split-K, else the split's column behind the slabs)
    const bool acc_ = ny == 1;
    float* dst_ = acc_ ? p.asum : reinterpret_cast<float*>(const_cast<void*>(p.C.ptr)) + (long long)ny * p.M * p.N + (long long)by * p.M;
    flush_asum(dst_, acc_, sum0, sum1, m0 + wr * 64, p.M, lane);
  }

  EpiArgs ea;
  ea.C = const_cast<void*>(p.C.ptr); ea.R = p.R.ptr; ea.P = p.P.ptr; ea.Cpre = const_cast<void*>(p.Cpre.ptr);
  ea.bias = p.bias ? p.bias + (long long)z * p.bias_zs : nullptr;
  ea.c_ld = p.C.ld; ea.c_bs = p.C.bstride; ea.r_ld = p.R.ld; ea.r_bs = p.R.bstride;
  ea.p_ld = p.P.ld; ea.p_bs = p.P.bstride; ea.q_ld = p.Cpre.ld; ea.q_bs = p.Cpre.bstride;
  ea.rpb = p.C.rpb; ea.M = p.M; ea.N = p.N; ea.act = p.act;
  ea.out_f32 = (p.flags & ST5_GEMM_OUT_F32) != 0; ea.dact = (p.flags & ST5_GEMM_DACT) != 0; ea.c_vec_ok = c_vec_ok;
  ea.atomic = 0;
  ea.fast = c_vec_ok && (p.N % 8 == 0);
  if (ny > 1) ea.C = reinterpret_cast<float*>(ea.C) + (long long)by * p.M * p.N;
  ea.alpha = p.alpha; ea.beta = p.beta; ea.dropout_p = p.dropout_p; ea.seed = p.seed;
  ea.ctr_base = (unsigned long long)z * (unsigned long long)p.M * (unsigned long long)p.N;
  {
    const long long zc = z_off(p.C.zs0, p.C.zs1, z, p.zdiv);
    if (ea.out_f32) ea.C = reinterpret_cast<float*>(ea.C) + zc; else ea.C = reinterpret_cast<T*>(ea.C) + zc;
    if (ea.R) {
      const long long zr = z_off(p.R.zs0, p.R.zs1, z, p.zdiv);
      if (ea.out_f32) ea.R = reinterpret_cast<const float*>(ea.R) + zr; else ea.R = reinterpret_cast<const T*>(ea.R) + zr;
    }
    if (ea.P) ea.P = reinterpret_cast<const T*>(ea.P) + z_off(p.P.zs0, p.P.zs1, z, p.zdiv);
    if (ea.Cpre) ea.Cpre = reinterpret_cast<T*>(ea.Cpre) + z_off(p.Cpre.zs0, p.Cpre.zs1, z, p.zdiv);
  }
  float* stage = reinterpret_cast<float*>(dsm) + wave * (32 * EP_LD);
  run_epilogue<T, FEAT, float>(ea, stage, acc00, acc01, acc10, acc11, m0 + wr * 64, n0 + wc * 64, lane);
}

template <int FEAT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_glds_kernel(const st5_gemm_params p, const int c_vec_ok) {
  ST5_PAD_TO_256_VGPRS();
  // Split-K grids (round 6): the XCD-aware placement runs over the WHOLE (tile, split) grid, split-major -- XCD x takes the x-th
  // contiguous eighth of the list, i.e. most of the tiles of one or two splits, which share their k-range of BOTH operands.  The
  // first form remapped the tile index alone: every XCD held an eighth of the tiles of EVERY split, and each private L2 fetched every
  // split's operand panels for itself (the convolution weight gradients: 48 tiles x 5-8 splits; 182 MB through the fabric per launch).
  // A (tile, split) pair's slab does not depend on where it ran: same bits.
  const int nx = (int)gridDim.x, ny = (int)gridDim.y;
  int bx = (int)blockIdx.x, by = (int)blockIdx.y;
  const bool placed = ny > 1 && gridDim.z == 1;
  if (placed) {
    const int total = nx * ny, lin = by * nx + bx;
    const int xcd = lin & 7, q = total >> 3, rmd = total & 7;
    const int pidx = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (lin >> 3);
    by = pidx / nx;
    bx = pidx - by * nx;
  }
  tn_glds_body<FEAT>(p, c_vec_ok, bx, nx, by, ny, (int)blockIdx.z, !placed);
}

// Several weight-gradient GEMMs in ONE launch, no split-K (st5_gemm_tn_group): the four (encoder) or six (decoder) weight gradients of a
// transformer layer are 36-144 tiles of 128^2 each with a reduction over every token -- alone none of them fills the chip, which is why
// st5_gemm splits their K range into fp32 slabs that a second kernel sums (a third of the bytes a weight-gradient launch moves).
// Together they are 430-500 tiles, one round of two blocks per CU: every block runs its tile's whole reduction and accumulates
// straight into the gradient buffer.  Measured on the layer's shapes (tools/r5/tn_group_probe.py): 233 -> 162 us at K = 8192 tokens,
// 135 -> 83 us at 3992, 107 -> 58 us at 2504.  Block ranges start at multiples of 8 so that the XCD-aware tile order of the body
// (block b runs on XCD b % 8) holds inside every problem; the padding blocks exit at once.
constexpr int TNG_MAX = 8;
struct TnGroupArgs { st5_gemm_params p[TNG_MAX]; int first[TNG_MAX + 1]; int tiles[TNG_MAX]; int n; };
template <int FEAT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_group_kernel(const TnGroupArgs g) {
  ST5_PAD_TO_256_VGPRS();
  int j = 0;
  while (j + 1 < g.n && (int)blockIdx.x >= g.first[j + 1]) ++j;
  const int lb = (int)blockIdx.x - g.first[j];
  if (lb >= g.tiles[j]) return;
  tn_glds_body<FEAT>(g.p[j], 1, lb, g.tiles[j], 0, 1, 0);
}

// ------------------------------------------------------------------------------------------------------
// TN fast path, 256 x 256 block tile, PHASED schedule (round 4): the weight-gradient form C[M, N] = A^T B (A = dY [K, M], B = X [K, N],
// both k-strided, fp32 output) on the schedule of gemm_nt8p_kernel -- four phases per 64-deep k-tile, one half-tile of LDS-DMA staging per
// phase, one counted vmcnt per k-tile, the two m-halves half a phase apart -- with the operand images and transpose-read fragments of
// gemm_tn_glds_kernel: a half-tile is [64 k][128 m] (16 KB, 16-byte chunks XOR 4 * (k & 3)), fragments by ds_read_b64_tr_b16.
// Weight gradients are long reductions (K = tokens: 2.5k-8k) over few output tiles (9-36 of 256 x 256): split-K over blockIdx.y brings
// the grid to about one block per CU, every split stores its fp32 slab (the batched slab reduction of st5_gemm sums them).
// The bias gradient dY^T . 1 (p.asum) is summed on the VALU from the A fragments the waves of the first tile column hold anyway
// (four floats per lane; the 128^2 kernel spends an extra MFMA column on it), in a fixed order: deterministic, no atomics.
// Not here: row-split / segmented operands (conv weight gradients) -- those stay on gemm_tn_glds_kernel.
// ------------------------------------------------------------------------------------------------------
// (bx of nwg: this block's tile among the problem's 256 x 256 tiles; by of ny: its split of the reduction -- the launch grid's own numbers
//  for the single-problem kernel, problem-local ones inside a grouped launch)
template <int FEAT, bool STAGGER, bool REMAP = true>
__device__ __forceinline__ void tn8p_body(const st5_gemm_params& p, const int c_vec_ok, const int bx, const int nwg_, const int by, const int ny) {
  typedef bf16_t T;
  constexpr int BK = 64;
  constexpr int HALF = TILE_BYTES;
  constexpr int BUF = 4 * HALF;
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (p.N + 255) / 256;
  int bid = bx;
  if (REMAP) {     // (the grouped launch remaps over ALL its tiles and passes the problem-local tile number)
    const int nwg = nwg_, xcd = bid & 7, q = nwg >> 3, rmd = nwg & 7;
    bid = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (bid >> 3);
  }
  int tm, tn;
  tile_of(bid, tiles_n, nwg_, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
  const T* Ap = reinterpret_cast<const T*>(p.A.ptr);
  const T* Bp = reinterpret_cast<const T*>(p.B.ptr);

  const int nk_all = (p.K + BK - 1) / BK;
  const int per = (nk_all + ny - 1) / ny;
  const int kt0 = by * per;
  int nk = nk_all - kt0; nk = nk < per ? nk : per; nk = nk > 0 ? nk : 0;

  // LDS-DMA sources.  Instruction i of this wave covers half-tile rows (k) (i * 8 + wave) * 4 .. +3: lane l -> row + (l >> 4), physical
  // chunk l & 15, logical chunk (l & 15) ^ 4 * (l >> 4).  M and N are multiples of 256 here (tn8p_ok), so every column chunk is inside
  // the matrix and all lanes advance by the same (scalar) stride; only rows past K (the problem's last k-tile) read the zero page.
  // (Per-lane strides and column predicates cost the two registers that made hipcc spill inside the k-loop -- and a spill reload is an
  // ordinary VMEM load whose vmcnt(0) drains the LDS-DMA queue every k-tile: 5.6 us per k-tile instead of 1.8.)
  const int lrow = lane >> 4;
  const int lchunk = (lane & 15) ^ (4 * lrow);
  const T* const zero = reinterpret_cast<const T*>(g_zero_page) + (lane & 15) * 8;
  const long long astep = (long long)BK * p.A.ld, bstep = (long long)BK * p.B.ld;
  const T* src[4][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long k = (long long)kt0 * BK + (i * 8 + wave) * 4 + lrow;
      src[h][i] = Ap + k * p.A.ld + (m0 + h * 128 + lchunk * 8);
      src[2 + h][i] = Bp + k * p.B.ld + (n0 + h * 128 + lchunk * 8);
    }
  auto stage = [&](const int h, const int kt) {        // half-tile h of this block's k-tile kt (absolute tile kt0 + kt) -> buffer kt & 1
    char* base = dsm + (kt & 1) * BUF + h * HALF + wave * 1024;
    if ((long long)(kt0 + kt + 1) * BK <= p.K) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src[h][i], (lds_ptr_t)(base + i * 8192), 16, 0, 0);
        src[h][i] += h < 2 ? astep : bstep;
      }
    } else {   // the problem's last, partial k-tile: rows past K come from the zero page
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const long long k = (long long)(kt0 + kt) * BK + (i * 8 + wave) * 4 + lrow;
        const T* sp = k < p.K ? src[h][i] : zero;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)sp, (lds_ptr_t)(base + i * 8192), 16, 0, 0);
      }
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transpose-read addressing (gemm_tn_glds_kernel): group g = m sub-block of 16, hh = k half, jj = k row inside a 4-row group, q
  const int g = (lane >> 4) & 1, hh = lane >> 5, jj = (lane & 15) >> 2, q = lane & 3;
  auto frag_off = [&](int mbase) {
    const int m = mbase + 16 * g + 4 * q;
    const int pos = (m >> 3) ^ (4 * jj);
    return (8 * hh + jj) * 256 + pos * 16 + ((m >> 2) & 1) * 8;
  };
  int aoff[4], boff[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = wr * HALF + frag_off(32 * i);
#pragma unroll
  for (int j = 0; j < 2; ++j) boff[j] = (2 + (wc >> 1)) * HALF + frag_off((wc & 1) * 64 + 32 * j);

  const bool do_asum = p.asum != nullptr && tn == 0 && wc == 0;
  float rsum[4] = {0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
    if (nk > 1) { stage(0, 1); stage(1, 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();

  bf16x8 af[4][4], bfr[4];
  const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)dsm;
  for (int t = 0; t < nk; ++t) {
    const unsigned cur = lds_base + (t & 1) * BUF;
    // ---- phase 1: B cols 0-31, all A; stage B0(t+1) ----
    bfr[0] = tr_frag_asm<0>(cur + boff[0]); bfr[1] = tr_frag_asm<4096>(cur + boff[0]);
    bfr[2] = tr_frag_asm<8192>(cur + boff[0]); bfr[3] = tr_frag_asm<12288>(cur + boff[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i][0] = tr_frag_asm<0>(cur + aoff[i]); af[i][1] = tr_frag_asm<4096>(cur + aoff[i]);
      af[i][2] = tr_frag_asm<8192>(cur + aoff[i]); af[i][3] = tr_frag_asm<12288>(cur + aoff[i]);
    }
    if (t + 1 < nk) stage(2, t + 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) { mma<T>(af[0][kg], bfr[kg], acc[0][0]); mma<T>(af[1][kg], bfr[kg], acc[1][0]); }
    __builtin_amdgcn_s_setprio(0);
    if (do_asum) {   // bias gradient: row sums of A^T from the fragments in registers (lane: row 32 i + (lane & 31), 8 k of every 16)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
          for (int e = 0; e < 8; ++e) rsum[i] += (float)af[i][kg][e];
    }
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: stage B1(t+1) ----
    if (t + 1 < nk) stage(3, t + 1);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) { mma<T>(af[2][kg], bfr[kg], acc[2][0]); mma<T>(af[3][kg], bfr[kg], acc[3][0]); }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: B cols 32-63; stage A0(t+2) ----
    bfr[0] = tr_frag_asm<0>(cur + boff[1]); bfr[1] = tr_frag_asm<4096>(cur + boff[1]);
    bfr[2] = tr_frag_asm<8192>(cur + boff[1]); bfr[3] = tr_frag_asm<12288>(cur + boff[1]);
    if (t + 2 < nk) stage(0, t + 2);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) { mma<T>(af[2][kg], bfr[kg], acc[2][1]); mma<T>(af[3][kg], bfr[kg], acc[3][1]); }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 4: stage A1(t+2); the one counted wait of the k-tile ----
    if (t + 2 < nk) { stage(1, t + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) { mma<T>(af[0][kg], bfr[kg], acc[0][1]); mma<T>(af[1][kg], bfr[kg], acc[1][1]); }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();
  __syncthreads();

  if (do_asum) {
    // (as asum_target: in place without split-K, else this split's column behind the slabs)
    const bool accumulate = ny == 1;
    float* dst = accumulate ? p.asum : reinterpret_cast<float*>(const_cast<void*>(p.C.ptr)) + (long long)ny * p.M * p.N + (long long)by * p.M;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = rsum[i] + __shfl_xor(rsum[i], 32, 64);
      const int row = m0 + wr * 128 + 32 * i + (lane & 31);
      if (hh == 0 && row < p.M) dst[row] = accumulate ? dst[row] + v : v;
    }
  }

  EpiArgs ea;
  ea.C = const_cast<void*>(p.C.ptr); ea.R = p.R.ptr; ea.P = p.P.ptr; ea.Cpre = const_cast<void*>(p.Cpre.ptr);
  ea.bias = p.bias;
  ea.c_ld = p.C.ld; ea.c_bs = p.C.bstride; ea.r_ld = p.R.ld; ea.r_bs = p.R.bstride;
  ea.p_ld = p.P.ld; ea.p_bs = p.P.bstride; ea.q_ld = p.Cpre.ld; ea.q_bs = p.Cpre.bstride;
  ea.rpb = p.C.rpb; ea.M = p.M; ea.N = p.N; ea.act = p.act;
  ea.out_f32 = (p.flags & ST5_GEMM_OUT_F32) != 0; ea.dact = (p.flags & ST5_GEMM_DACT) != 0; ea.c_vec_ok = c_vec_ok;
  ea.atomic = 0;
  ea.fast = c_vec_ok && (p.N % 8 == 0);
  if (ny > 1) ea.C = reinterpret_cast<float*>(ea.C) + (long long)by * p.M * p.N;
  ea.alpha = p.alpha; ea.beta = p.beta; ea.dropout_p = p.dropout_p; ea.seed = p.seed;
  ea.ctr_base = 0ull;
  float* stg = reinterpret_cast<float*>(dsm) + wave * (32 * EP_LD);
  run_epilogue<T, FEAT, float>(ea, stg, acc[0][0], acc[0][1], acc[1][0], acc[1][1], m0 + wr * 128, n0 + wc * 64, lane);
  run_epilogue<T, FEAT, float>(ea, stg, acc[2][0], acc[2][1], acc[3][0], acc[3][1], m0 + wr * 128 + 64, n0 + wc * 64, lane);
}

template <int FEAT, bool STAGGER>
__global__ __launch_bounds__(512) void gemm_tn8p_kernel(const st5_gemm_params p, const int c_vec_ok) {
  tn8p_body<FEAT, STAGGER>(p, c_vec_ok, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, (int)gridDim.y);
}

// The grouped weight-gradient launch (gemm_tn_group_kernel) on the PHASED 256 x 256 schedule (round 6).  A layer's weight gradients are
// 108 (Base) / 192 (Large) tiles of 256 x 256 with whole token reductions; two layers' worth (or a layer and a bit) is one block on nearly
// every CU, and a whole-token reduction is 60-250 k-tiles: the regime where the phased schedule is 1.3-1.5x the 128^2 kernel per flop (its
// exposed prologue / 128 KB store epilogue amortised) and the fabric sees every operand stripe N / 256 instead of N / 128 times.  As in the
// 128^2 group: every tile runs its own whole reduction and accumulates into the gradient buffer (no slabs, no reduction kernel), block
// ranges start at multiples of 8 (XCD-aware tile order inside every problem), a problem's result does not depend on its group.  Every
// output element sees the MFMA chain of the 128^2 kernels (k-tiles ascending, four 16-deep groups each): the WEIGHT gradients are
// bit-identical to gemm_tn_group_kernel's; the bias-gradient column is summed on the VALU here (tn8p_body), in another -- fixed -- order.
// (a problem as the phased body needs it -- 72 bytes instead of st5_gemm_params' 400+, so that SIXTEEN fit into the kernel arguments: a
//  Base decoder layer has six weight gradients of 9-36 tiles, and eight problems filled only 56-77 % of a round)
constexpr int TNG8P_MAX = 16;
struct Tn8pProb { const void* A; const void* B; void* C; float* asum; long long lda, ldb, ldc; int M, N, K; float beta; };
struct Tn8pGroupArgs { Tn8pProb p[TNG8P_MAX]; int first[TNG8P_MAX + 1]; int tiles[TNG8P_MAX]; int n; };
template <int FEAT>
__global__ __launch_bounds__(512) void gemm_tn8p_group_kernel(const Tn8pGroupArgs g) {
  // XCD-aware order over the WHOLE launch (round 6): block b runs on XCD b % 8; XCD x takes the x-th contiguous eighth of the launch's tile
  // list (problems in queue order, tiles of a problem in tile_of's patch-friendly order), i.e. ~27-32 consecutive tiles of ONE or two
  // problems: a 9 x 3 patch reads 12 operand panels per k-step for 27 tiles.  The first form remapped inside every problem: with 9-36
  // tiles per problem every XCD held 1-5 tiles of EACH of the launch's eight problems -- ~35 panels for the same 27 tiles -- and the
  // counters showed it: 921 MB through the fabric per launch (2.1x the algorithmic bytes) at 4 TB/s, the bound of the kernel.
  const int T = g.first[g.n];
  int gidx;
  {
    const int b = (int)blockIdx.x, xcd = b & 7, q = T >> 3, rmd = T & 7;
    gidx = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (b >> 3);
  }
  int j = 0;
  while (j + 1 < g.n && gidx >= g.first[j + 1]) ++j;
  const int lb = gidx - g.first[j];
  st5_gemm_params p = {};
  p.A.ptr = g.p[j].A; p.A.ld = g.p[j].lda; p.B.ptr = g.p[j].B; p.B.ld = g.p[j].ldb; p.C.ptr = g.p[j].C; p.C.ld = g.p[j].ldc;
  p.asum = g.p[j].asum; p.M = g.p[j].M; p.N = g.p[j].N; p.K = g.p[j].K; p.batch = 1; p.zdiv = 1;
  p.alpha = 1.f; p.beta = g.p[j].beta; p.act = ACT_NONE;
  p.flags = ST5_GEMM_A_KSTRIDED | ST5_GEMM_B_KSTRIDED | ST5_GEMM_OUT_F32;
  tn8p_body<FEAT, true, false>(p, 1, lb, g.tiles[j], 0, 1);
}

int g_tn8p = 0;   // st5_gemm_set_tn_phased: 0 (default) = always the 128^2 kernel; 1 / 2 = eligible weight-gradient GEMMs on the phased 256^2 kernel (staggered / not).
                  // Measured (profiles/r4_gemm_tn_phased.txt): the phased kernel wins only on very long reductions over few tiles (512 x 1536 x 128k:
                  // 335 vs 343 us); on the transformer weight gradients its 256^2 tiles need 2x the split-K slabs and lose 20-35 %.
bool tn8p_ok(const st5_gemm_params& p, int c_vec_ok) {
  if (!g_tn8p || p.batch != 1 || p.A.rpb || p.B.rpb || p.A.seg || p.B.seg) return false;
  if (!c_vec_ok || p.N % 8 || !(p.flags & ST5_GEMM_OUT_F32) || (p.flags & ST5_GEMM_DACT) || p.act != ACT_NONE || p.dropout_p != 0.f ||
      p.R.ptr || p.Cpre.ptr || p.bias)
    return false;
  return p.M % 256 == 0 && p.N % 256 == 0 && p.K >= 512;
}
// split count of the phased kernel: about one block per CU, at least four k-tiles per split
int tn8p_splits(const st5_gemm_params& p) {
  const long long tiles = (long long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  const int nk = (p.K + 63) / 64;
  long long want = (256 + tiles - 1) / tiles;
  long long maxs = nk / 4;
  int ns = (int)(want < maxs ? want : maxs);
  if (ns < 1) ns = 1;
  const int per = (nk + ns - 1) / ns;
  return (nk + per - 1) / per;
}
int launch_tn8p(const st5_gemm_params& p, int c_vec_ok, int nsplit, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    const void* fns[] = {(const void*)gemm_tn8p_kernel<0, true>, (const void*)gemm_tn8p_kernel<F_BETA, true>,
                         (const void*)gemm_tn8p_kernel<0, false>, (const void*)gemm_tn8p_kernel<F_BETA, false>};
    for (const void* f : fns)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return ST5_ERR_LAUNCH;
    attr = true;
  }
  const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  dim3 grid(tiles, nsplit, 1), block(512);
  const size_t shm = (size_t)8 * TILE_BYTES;
  if (p.beta == 0.f) {
    if (g_tn8p == 2) hipLaunchKernelGGL((gemm_tn8p_kernel<0, false>), grid, block, shm, s, p, c_vec_ok);
    else hipLaunchKernelGGL((gemm_tn8p_kernel<0, true>), grid, block, shm, s, p, c_vec_ok);
  } else {
    if (g_tn8p == 2) hipLaunchKernelGGL((gemm_tn8p_kernel<F_BETA, false>), grid, block, shm, s, p, c_vec_ok);
    else hipLaunchKernelGGL((gemm_tn8p_kernel<F_BETA, true>), grid, block, shm, s, p, c_vec_ok);
  }
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

int launch_tn_glds(const st5_gemm_params& p, int c_vec_ok, int nsplit, hipStream_t s) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  dim3 grid(tiles, nsplit, p.batch), block(NTHREADS);
  // weight gradients: fp32 output, plain store (split-K slabs / first contribution) or accumulate -- own instantiations
  static const bool rt_only = getenv("ST5_GEMM_RT_EPILOGUE") && getenv("ST5_GEMM_RT_EPILOGUE")[0] == '1';   // A/B switch
  const bool plain = !rt_only && c_vec_ok && p.N % 8 == 0 && (p.flags & ST5_GEMM_OUT_F32) && !(p.flags & ST5_GEMM_DACT) && p.act == ACT_NONE &&
                     p.dropout_p == 0.f && !p.R.ptr && !p.Cpre.ptr;
  if (plain && p.beta == 0.f) hipLaunchKernelGGL(gemm_tn_glds_kernel<0>, grid, block, (size_t)4 * TILE_BYTES, s, p, c_vec_ok);
  else if (plain) hipLaunchKernelGGL(gemm_tn_glds_kernel<F_BETA>, grid, block, (size_t)4 * TILE_BYTES, s, p, c_vec_ok);
  else hipLaunchKernelGGL(gemm_tn_glds_kernel<-1>, grid, block, (size_t)4 * TILE_BYTES, s, p, c_vec_ok);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

// operands the TN LDS-DMA kernel can take: bf16, both k-strided (row split allowed, no k segments), 16-byte
// chunks along the contiguous index fully inside or outside the matrix
bool tn_glds_ok(const st5_gemm_params& p, int dtype) {
  if (dtype != ST5_BF16) return false;
  if (!(p.flags & ST5_GEMM_A_KSTRIDED) || !(p.flags & ST5_GEMM_B_KSTRIDED)) return false;
  if ((p.A.seg && (p.A.seg % 8 || p.A.seg_stride % 8)) || (p.B.seg && (p.B.seg % 8 || p.B.seg_stride % 8))) return false;
  if ((p.A.rpb && (p.A.rpb < 64 || p.A.bstride % 8)) || (p.B.rpb && (p.B.rpb < 64 || p.B.bstride % 8))) return false;
  if (p.M % 8 || p.N % 8 || p.A.ld % 8 || p.B.ld % 8 || p.A.zs0 % 8 || p.A.zs1 % 8 || p.B.zs0 % 8 || p.B.zs1 % 8) return false;
  return aligned(p.A.ptr, 16) && aligned(p.B.ptr, 16);
}

// Block-tile choice for the NT path (tools/bench_kernels.py nt256 with NT_MODES=1,3; tools/gemm_cases.py with ST5_NT_TILE=3; MI355X,
// round 4).  The phased 256^2 kernel keeps one block per CU, so its prologue and its epilogue (128 KB of stores per block) are
// exposed (~6.5 us per tile against 1.8 us per 64-deep k-step): it pays when there are several rounds of full-chip work to amortise
// that over -- the conv feature-extractor GEMMs (M = 32k..128k: 0.92-1.0 PFLOP/s against 0.70-0.80 on 128^2 tiles), large square
// problems (8192^3: 1.20 against 0.89) -- or when ONE round of 256^2 tiles nearly fills the chip with a short reduction (3992 x 3072 x
// 768: 192 tiles, 27.8 against 30.6 us).  Everything else of the transformer (96-384 tiles in 1.1-2 rounds, or long K on few tiles)
// stays on 128^2, two blocks per CU.
// (A/B knob, round 6, st5_gemm_set_nt_longk: problems of at least `g_nt_longk_tiles` tiles of 256^2 with a reduction of at least
//  `g_nt_longk_nk` k-tiles also take the phased kernel -- the N = 768, K = 2304 / 3072 shapes at 8192 rows are 96 tiles: one block on 96
//  CUs for 36-48 k-tiles instead of 384 tiles of 128^2 on every CU.  0 = off, the default.)
int g_nt_longk_tiles = 0, g_nt_longk_nk = 36;
bool nt256_pays(int M, int N, int nk64, int batch) {
  const long long t256 = (long long)((M + 255) / 256) * ((N + 255) / 256) * batch;
  if (t256 >= 448) return true;
  if (g_nt_longk_tiles > 0 && t256 >= g_nt_longk_tiles && nk64 >= g_nt_longk_nk) return true;
  return t256 >= 176 && t256 <= 256 && nk64 <= 16;
}

// ------------------------------------------------------------------------------------------------------
// MX-fp8 NT path (BASELINE.json configs[4]: SpeechT5-Large with fp8 MFMA GEMMs; arch models/speecht5.py:1402-1425).
// C = epilogue(A . B^T) with A [M x K], B [N x K] stored as OCP fp8 e4m3 bytes plus one e8m0 scale byte per 32 consecutive
// k-elements of a row (the OCP microscaling format: value = fp8 * 2^(scale - 127)), C-class operands bf16.  The block-scaled
// v_mfma_scale_f32_32x32x64_f8f6f4 is the only fp8 MFMA on gfx950 that runs at twice the bf16 rate (the unscaled fp8 forms
// run AT the bf16 rate): 64 k-elements = two MX blocks per instruction; a lane (row = lane & 31, h = lane >> 5) supplies 16 bytes of each
// block (k = 16h .. 16h+15) and the scale byte of block h -- the hardware's scale block IS the 32-element MX block.
// Same skeleton as gemm_nt_glds_kernel: 128 x 128 tile, 4 waves x 64 x 64, tile rows of 128 BYTES (= 128 k-elements, twice the
// reduction depth per LDS-DMA byte of the bf16 kernel: the per-CU operand fill rate that caps the 128^2 bf16 kernel at
// ~1 PFLOP/s caps this one at ~2), 16-byte XOR swizzle on the source side, two-stage ring, the shared fused epilogues.
// Scale bytes: the 4 blocks of a row's k-tile are one aligned dword; it is loaded one k-step ahead straight into registers
// (issued BEFORE that step's LDS-DMA loads, so the in-order vmcnt wait at the top of the next step covers it and no wait of
// its own drains the DMA queue); op_sel is ignored by the hardware (byte 0 is always taken), so the two 64-deep k-groups of the
// tile get pre-shifted copies of the dword.
// ------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) int i32x8;
// (F_Q8 instantiations: where the epilogue writes the MX-fp8 image of the output; null otherwise)
struct MxOut { unsigned char* q; long long q_ld; unsigned char* s; long long s_ld; };

__device__ __forceinline__ i32x8 mx_frag2(const char* tile, int row, int c0, int c1) {
  const u32x4 lo = *reinterpret_cast<const u32x4*>(tile + lds_off(row, c0));
  const u32x4 hi = *reinterpret_cast<const u32x4*>(tile + lds_off(row, c1));
  i32x8 r;
  r[0] = (int)lo[0]; r[1] = (int)lo[1]; r[2] = (int)lo[2]; r[3] = (int)lo[3];
  r[4] = (int)hi[0]; r[5] = (int)hi[1]; r[6] = (int)hi[2]; r[7] = (int)hi[3];
  return r;
}
template <int FEAT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_mx8_kernel(const st5_gemm_params p, const int c_vec_ok,
                                                                  const unsigned char* __restrict__ sa, const long long sa_ld,
                                                                  const unsigned char* __restrict__ sb, const long long sb_ld, const MxOut mo) {
  ST5_PAD_TO_256_VGPRS();
  typedef bf16_t T;                 // type of the C-class operands
  constexpr int BK = 128;           // k-elements (= bytes) per tile row
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, rmd = nwg & 7;
    bid = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (bid >> 3);
  }
  int tm, tn;
  tile_of(bid, tiles_n, (int)gridDim.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const unsigned char* Ap = reinterpret_cast<const unsigned char*>(p.A.ptr);
  const unsigned char* Bp = reinterpret_cast<const unsigned char*>(p.B.ptr);

  const unsigned char* asrc[4];
  const unsigned char* bsrc[4];
  const int rsub = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wave) * 8 + rsub;
    const int c = pc ^ ((row >> 1) & 7);
    int gr = m0 + row; gr = gr < p.M ? gr : p.M - 1;
    asrc[i] = Ap + (long long)gr * p.A.ld + c * 16;
    int gc = n0 + row; gc = gc < p.N ? gc : p.N - 1;
    bsrc[i] = Bp + (long long)gc * p.B.ld + c * 16;
  }
  const int dst0 = wave * 1024;
  auto issue = [&](int kt, int buf) {
    char* base = dsm + buf * 2 * TILE_BYTES + dst0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(asrc[i] + (long long)kt * BK), (lds_ptr_t)(base + i * 4096), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bsrc[i] + (long long)kt * BK), (lds_ptr_t)(base + TILE_BYTES + i * 4096), 16, 0, 0);
    }
  };
  const int frow = lane & 31, fhalf = lane >> 5;
  const int arow0 = wr * 64 + frow, brow0 = wc * 64 + frow;
  // scale rows of this lane's four fragment rows (dword kt of a row = the 4 block scales of k-tile kt)
  const unsigned int* srow[4];
  {
    int r0 = m0 + arow0, r1 = r0 + 32, c0 = n0 + brow0, c1 = c0 + 32;
    r0 = r0 < p.M ? r0 : p.M - 1; r1 = r1 < p.M ? r1 : p.M - 1;
    c0 = c0 < p.N ? c0 : p.N - 1; c1 = c1 < p.N ? c1 : p.N - 1;
    srow[0] = reinterpret_cast<const unsigned int*>(sa + (long long)r0 * sa_ld);
    srow[1] = reinterpret_cast<const unsigned int*>(sa + (long long)r1 * sa_ld);
    srow[2] = reinterpret_cast<const unsigned int*>(sb + (long long)c0 * sb_ld);
    srow[3] = reinterpret_cast<const unsigned int*>(sb + (long long)c1 * sb_ld);
  }
  const int sh = 8 * fhalf;

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

  const int nk = p.K / BK;
  unsigned int scn[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) scn[j] = srow[j][0];
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile kt and its scales (both issued one step ago)
    __builtin_amdgcn_s_barrier();
    // the scale operand is ALWAYS read from byte 0 of its register on this part (op_sel is ignored -- also composable_kernel's
    // finding, ck/utility/amd_xdlops.hpp), so each k-group gets its own pre-shifted copy: block 2g + (lane >> 5) of the row's dword
    int sc0[4], sc1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc0[j] = (int)(scn[j] >> sh); sc1[j] = (int)(scn[j] >> (16 + sh)); }
    if (kt + 1 < nk) {
#pragma unroll
      for (int j = 0; j < 4; ++j) scn[j] = srow[j][kt + 1];
      issue(kt + 1, (kt + 1) & 1);
    }
    const char* cur = dsm + (kt & 1) * 2 * TILE_BYTES;
    // operand layout measured on the part (tools/probe/mx_layout_probe.hip, tests/test_fp8_gpu.py): lane (row, h = lane >> 5)
    // supplies bytes k = 16h .. 16h+15 of the instruction's FIRST 32-block in registers 0-3 and k = 16h .. 16h+15 of its SECOND
    // block in registers 4-7; the scale of block b comes from the lanes with h == b (byte 0 of their scale register).
    // Both k-groups' fragments (g: bytes 64g .. 64g+63 of the tile rows = MX blocks 2g, 2g+1) are read before the first MFMA, so the
    // second group's LDS round trip runs under the first group's MFMAs (the bf16 kernel's round-4 fragment pipelining).
    i32x8 a0[2], a1[2], b0[2], b1[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int ca = 4 * g + fhalf, cb = 4 * g + 2 + fhalf;
      a0[g] = mx_frag2(cur, arow0, ca, cb); b0[g] = mx_frag2(cur + TILE_BYTES, brow0, ca, cb);
      a1[g] = mx_frag2(cur, arow0 + 32, ca, cb); b1[g] = mx_frag2(cur + TILE_BYTES, brow0 + 32, ca, cb);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int* sc = g == 0 ? sc0 : sc1;
      acc00 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0[g], b0[g], acc00, 0, 0, 0, sc[0], 0, sc[2]);
      acc01 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0[g], b1[g], acc01, 0, 0, 0, sc[0], 0, sc[3]);
      acc10 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1[g], b0[g], acc10, 0, 0, 0, sc[1], 0, sc[2]);
      acc11 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1[g], b1[g], acc11, 0, 0, 0, sc[1], 0, sc[3]);
    }
  }
  __syncthreads();

  EpiArgs ea;
  ea.C = const_cast<void*>(p.C.ptr); ea.R = p.R.ptr; ea.P = p.P.ptr; ea.Cpre = const_cast<void*>(p.Cpre.ptr);
  ea.bias = p.bias;
  ea.c_ld = p.C.ld; ea.c_bs = p.C.bstride; ea.r_ld = p.R.ld; ea.r_bs = p.R.bstride;
  ea.p_ld = p.P.ld; ea.p_bs = p.P.bstride; ea.q_ld = p.Cpre.ld; ea.q_bs = p.Cpre.bstride;
  ea.rpb = p.C.rpb; ea.M = p.M; ea.N = p.N; ea.act = p.act;
  ea.out_f32 = (p.flags & ST5_GEMM_OUT_F32) != 0; ea.dact = (p.flags & ST5_GEMM_DACT) != 0; ea.c_vec_ok = c_vec_ok;
  ea.atomic = 0;
  ea.fast = c_vec_ok && (p.N % 8 == 0);
  ea.alpha = p.alpha; ea.beta = p.beta; ea.dropout_p = p.dropout_p; ea.seed = p.seed;
  ea.ctr_base = 0ull;
  ea.q8 = mo.q; ea.s8 = mo.s; ea.q8_ld = mo.q_ld; ea.s8_ld = mo.s_ld;
  float* stage = reinterpret_cast<float*>(dsm) + wave * (32 * EP_LD);
  run_epilogue<T, FEAT>(ea, stage, acc00, acc01, acc10, acc11, m0 + wr * 64, n0 + wc * 64, lane);
}

template <int FEAT>
int launch_mx8_as(const st5_gemm_params& p, int c_vec_ok, const unsigned char* sa, long long sa_ld, const unsigned char* sb, long long sb_ld,
                  dim3 grid, hipStream_t s, const MxOut mo = MxOut{nullptr, 0, nullptr, 0}) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_nt_mx8_kernel<FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return ST5_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((gemm_nt_mx8_kernel<FEAT>), grid, dim3(NTHREADS), (size_t)4 * TILE_BYTES, s, p, c_vec_ok, sa, sa_ld, sb, sb_ld, mo);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

// ------------------------------------------------------------------------------------------------------
// MX-fp8 NT path, 256 x 256 block tile, PHASED schedule (round 6): gemm_nt8p_kernel's program on fp8 bytes.  An LDS row is still 128
// bytes, now 128 k-elements, so staging, swizzle, the four fragment offsets and the phase structure are the bf16 kernel's byte for
// byte; a phase's eight v_mfma_f32_32x32x16_bf16 (32 cycles each) become four v_mfma_scale_f32_32x32x64_f8f6f4 (64 cycles each): the
// same matrix-pipe time and the same LDS / LDS-DMA traffic per k-tile for TWICE the reduction depth.  That is what makes the fp8 mode
// pay: the 128^2 fp8 kernel ran at 0.85-0.97 PFLOP/s against the phased bf16 kernel's 0.9-1.2 on the Large shapes (a tie).
// Scale bytes: as in gemm_nt_mx8_kernel one aligned dword per row and k-tile, loaded straight into registers one k-tile ahead -- in
// phase 1, IN FRONT of that phase's LDS-DMA pieces, so the one counted vmcnt(4) of phase 4 (which leaves only the two youngest
// half-tiles in flight) retires them too; six dwords per wave and k-tile (4 A row blocks, 2 B row blocks of the 128 x 64 wave tile).
// Every output element sees the MFMA chain of the 128^2 kernel (k-tiles ascending, two 64-deep groups each): bit-identical results.
// ------------------------------------------------------------------------------------------------------
template <int FEAT, bool STAGGER>
__global__ __launch_bounds__(512) void gemm_nt8p_mx8_kernel(const st5_gemm_params p, const int c_vec_ok,
                                                            const unsigned char* __restrict__ sa, const long long sa_ld,
                                                            const unsigned char* __restrict__ sb, const long long sb_ld, const MxOut mo) {
  typedef bf16_t T;                             // type of the C-class operands
  constexpr int BK = 128;                       // k-elements (= bytes) per tile row
  constexpr int HALF = TILE_BYTES;              // 16 KB: 128 rows x 128 B
  constexpr int BUF = 4 * HALF;                 // A0 A1 B0 B1
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (p.N + 255) / 256;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, rmd = nwg & 7;
    bid = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (bid >> 3);
  }
  int tm, tn;
  tile_of(bid, tiles_n, (int)gridDim.x, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
  // Operands and scales through buffer resources (32-bit per-lane offsets + a scalar k offset instead of 64-bit pointers: this kernel
  // has 208 registers of accumulators and fragments, two waves per SIMD, and six scale dwords more to hold than its bf16 twin).  The
  // launcher guarantees that every image is < 2 GB.
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A.ptr), 0, (int)((long long)p.M * p.A.ld), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B.ptr), 0, (int)((long long)p.N * p.B.ld), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(sa), 0, (int)((long long)p.M * sa_ld), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(sb), 0, (int)((long long)p.N * sb_ld), 0x00020000);
  int voff[4][2];     // half-tile h (0, 1: A rows 0-127 / 128-255; 2, 3: B), instruction i of this wave: rows (i * 8 + wave) * 8 .. +7
  {
    const int rsub = lane >> 3, pc = lane & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (i * 8 + wave) * 8 + rsub;
      const int c = pc ^ ((row >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int gr = m0 + h * 128 + row; gr = gr < p.M ? gr : p.M - 1;
        voff[h][i] = gr * (int)p.A.ld + c * 16;
        int gc = n0 + h * 128 + row; gc = gc < p.N ? gc : p.N - 1;
        voff[2 + h][i] = gc * (int)p.B.ld + c * 16;
      }
    }
  }
  auto stage = [&](const int h, const int kt) {        // half-tile h of k-tile kt -> buffer kt & 1
    char* base = dsm + (kt & 1) * BUF + h * HALF + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(h < 2 ? ra : rb, (lds_ptr_t)(base + i * 8192), 16, voff[h][i], kt * BK, 0, 0);
  };

  const int nk = p.K / BK;
  const int frow = lane & 31, fhalf = lane >> 5;
  int foff[4];      // chunk 2 kg + fhalf of the lane's fragment row: k-group g of the k-tile reads kg = 2g (first MX block) and 2g + 1
#pragma unroll
  for (int kg = 0; kg < 4; ++kg) foff[kg] = lds_off(frow, 2 * kg + fhalf);
  const int a_half = wr * HALF;
  const int b_base = (2 + (wc >> 1)) * HALF + (wc & 1) * 64 * 128;
  // scale dwords of this lane's fragment rows: 4 A row blocks (32 rows each), 2 B row blocks; dword kt = the 4 block scales of k-tile kt
  int svoff[6];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = m0 + wr * 128 + i * 32 + frow; r = r < p.M ? r : p.M - 1;
    svoff[i] = r * (int)sa_ld;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int c = n0 + wc * 64 + j * 32 + frow; c = c < p.N ? c : p.N - 1;
    svoff[4 + j] = c * (int)sb_ld;
  }
  auto load_scales = [&](unsigned int (&dst)[6], const int kt) {
#pragma unroll
    for (int j = 0; j < 6; ++j) dst[j] = __builtin_amdgcn_raw_buffer_load_b32(j < 4 ? rsa : rsb, svoff[j], 4 * kt, 0);
  };
  const int sh = 8 * fhalf;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // prologue: the scales and all of k-tile 0, the A halves of k-tile 1
  unsigned int scn[6];
  load_scales(scn, 0);
  stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
  if (nk > 1) { stage(0, 1); stage(1, 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();

  // fragments: a[i][g] / b[g] = the two MX blocks of k-group g for row block i (registers 0-3: first block, 4-7: second; layout
  // measured on the part, see gemm_nt_mx8_kernel)
  auto frag = [&](const char* tile, const int g) {
    const u32x4 lo = *reinterpret_cast<const u32x4*>(tile + foff[2 * g]);
    const u32x4 hi = *reinterpret_cast<const u32x4*>(tile + foff[2 * g + 1]);
    i32x8 r;
    r[0] = (int)lo[0]; r[1] = (int)lo[1]; r[2] = (int)lo[2]; r[3] = (int)lo[3];
    r[4] = (int)hi[0]; r[5] = (int)hi[1]; r[6] = (int)hi[2]; r[7] = (int)hi[3];
    return r;
  };
  // A phase's four MFMAs (two accumulators x two k-groups; the dependent pair of an accumulator is separated by the other one's).  The
  // empty asm statements pin them INSIDE the phase: hipcc sinks the (side-effect-free) mfma_scale builtins of all four phases to the
  // end of the loop body otherwise -- behind the barriers, the counted vmcnt wait and every LDS-DMA issue, i.e. no overlap at all.
#define MX8P_PIN(I0, I1, J) asm volatile("" : "+v"(acc[I0][J]), "+v"(acc[I1][J]));
#define MX8P_MMA2(I0, I1, J)                                                                                                                  \
  MX8P_PIN(I0, I1, J)                                                                                                                         \
  acc[I0][J] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[I0][0], bfr[0], acc[I0][J], 0, 0, 0, (int)sc[I0], 0, (int)sc[4 + (J)]);      \
  acc[I1][J] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[I1][0], bfr[0], acc[I1][J], 0, 0, 0, (int)sc[I1], 0, (int)sc[4 + (J)]);      \
  acc[I0][J] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[I0][1], bfr[1], acc[I0][J], 0, 0, 0, (int)sc2[I0], 0, (int)sc2[4 + (J)]);    \
  acc[I1][J] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[I1][1], bfr[1], acc[I1][J], 0, 0, 0, (int)sc2[I1], 0, (int)sc2[4 + (J)]);    \
  MX8P_PIN(I0, I1, J)
  i32x8 af[4][2], bfr[2];
  unsigned int sc[6], sc2[6];
  // One k-tile; HAS1 / HAS2: k-tiles t + 1 / t + 2 exist.  The steady-state loop runs the <true, true> copy and the last two k-tiles
  // their own, so that no staging sits behind a branch: hipcc's vmcnt bookkeeping for the scale dwords (ordinary loads it must wait
  // for itself) only counts the LDS-DMA pieces it can PROVE younger; with conditional staging it waited vmcnt(2) at the bottom of every
  // k-tile -- i.e. for the half-tile issued one phase earlier -- instead of nothing (the counted vmcnt(4) of phase 4 covers them).
  auto ktile = [&](const int t, auto has1, auto has2) {
    constexpr bool HAS1 = decltype(has1)::value, HAS2 = decltype(has2)::value;
    const char* cur = dsm + (t & 1) * BUF;
    // the scale operand is read from byte 0 of its register (op_sel is ignored on this part): block 2g + (lane >> 5) of the row's dword
#pragma unroll
    for (int j = 0; j < 6; ++j) { sc[j] = scn[j] >> sh; sc2[j] = sc[j] >> 16; }
    // ---- phase 1 ----
#pragma unroll
    for (int g = 0; g < 2; ++g) bfr[g] = frag(cur + b_base, g);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int g = 0; g < 2; ++g) af[i][g] = frag(cur + a_half + i * 4096, g);
    if constexpr (HAS1) {
      load_scales(scn, t + 1);
      stage(2, t + 1);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    MX8P_MMA2(0, 1, 0)
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2 ----
    if constexpr (HAS1) stage(3, t + 1);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
    MX8P_MMA2(2, 3, 0)
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3 ----
#pragma unroll
    for (int g = 0; g < 2; ++g) bfr[g] = frag(cur + b_base + 4096, g);
    if constexpr (HAS2) stage(0, t + 2);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    MX8P_MMA2(2, 3, 1)
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 4 ----
    if constexpr (HAS2) { stage(1, t + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
    MX8P_MMA2(0, 1, 1)
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  };
  {
    typedef std::integral_constant<bool, true> yes_t;
    typedef std::integral_constant<bool, false> no_t;
    int t = 0;
    for (; t + 2 < nk; ++t) ktile(t, yes_t(), yes_t());
    if (t + 1 < nk) { ktile(t, yes_t(), no_t()); ++t; }
    ktile(t, no_t(), no_t());
  }
#undef MX8P_MMA2
#undef MX8P_PIN
  if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();     // (every wave executes the same number of barriers)
  __syncthreads();

  EpiArgs ea;
  ea.C = const_cast<void*>(p.C.ptr); ea.R = p.R.ptr; ea.P = p.P.ptr; ea.Cpre = const_cast<void*>(p.Cpre.ptr);
  ea.bias = p.bias;
  ea.c_ld = p.C.ld; ea.c_bs = p.C.bstride; ea.r_ld = p.R.ld; ea.r_bs = p.R.bstride;
  ea.p_ld = p.P.ld; ea.p_bs = p.P.bstride; ea.q_ld = p.Cpre.ld; ea.q_bs = p.Cpre.bstride;
  ea.rpb = p.C.rpb; ea.M = p.M; ea.N = p.N; ea.act = p.act;
  ea.out_f32 = (p.flags & ST5_GEMM_OUT_F32) != 0; ea.dact = (p.flags & ST5_GEMM_DACT) != 0; ea.c_vec_ok = c_vec_ok;
  ea.atomic = 0;
  ea.fast = c_vec_ok && (p.N % 8 == 0);
  ea.alpha = p.alpha; ea.beta = p.beta; ea.dropout_p = p.dropout_p; ea.seed = p.seed;
  ea.ctr_base = 0ull;
  ea.q8 = mo.q; ea.s8 = mo.s; ea.q8_ld = mo.q_ld; ea.s8_ld = mo.s_ld;
  float* stg = reinterpret_cast<float*>(dsm) + wave * (32 * EP_LD);
  run_epilogue<T, FEAT>(ea, stg, acc[0][0], acc[0][1], acc[1][0], acc[1][1], m0 + wr * 128, n0 + wc * 64, lane);
  run_epilogue<T, FEAT>(ea, stg, acc[2][0], acc[2][1], acc[3][0], acc[3][1], m0 + wr * 128 + 64, n0 + wc * 64, lane);
}

// (one block per CU: its prologue and 128 KB store epilogue are exposed, so it wants whole rounds -- see nt256_pays)
// A heavy epilogue (GELU + pre-activation copy, the GELU derivative, dropout, the fp8 image: two or three output tensors or a dozen
// VALU slots per element) is NOT hidden in a one-block-per-CU kernel; behind a short reduction it costs more than the phased schedule
// gains (measured inside the Large update, B = 32: fc1 forward N = 4096, K = 1024 with GELU + copy + fp8 image 147 us phased against
// 113 us + quantiser on 128^2 tiles, while the plain N = 3072 / 4096 launches gain 20-25 %): such launches need >= g_mx8_heavy_nk k-tiles.
int g_mx8_heavy_nk = 16;
bool mx8_256_pays(int M, int N, int nk, int feat) {
  const long long t256 = (long long)((M + 255) / 256) * ((N + 255) / 256);
  if (!(t256 >= 448 || (t256 >= 176 && t256 <= 256))) return false;
  const bool heavy = feat < 0 || (feat & (F_GELU | F_DACT | F_PRE | F_DROP | F_Q8)) != 0;
  return !heavy || nk >= g_mx8_heavy_nk;
}
int g_mx8_tile = 0;   // 0 = choose per problem, 1 = always 128^2, 2 = always the phased 256^2 kernel (st5_gemm_set_mx8_tile)
template <int FEAT>
int launch_mx8p_as(const st5_gemm_params& p, int c_vec_ok, const unsigned char* sa, long long sa_ld, const unsigned char* sb, long long sb_ld,
                   hipStream_t s, const MxOut mo = MxOut{nullptr, 0, nullptr, 0}) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_nt8p_mx8_kernel<FEAT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return ST5_ERR_LAUNCH;
    attr = true;
  }
  const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  hipLaunchKernelGGL((gemm_nt8p_mx8_kernel<FEAT, true>), dim3(tiles, 1, 1), dim3(512), (size_t)8 * TILE_BYTES, s, p, c_vec_ok, sa, sa_ld, sb, sb_ld, mo);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

// MX quantisation of a row-major bf16 matrix along its rows: per 32 consecutive elements one e8m0 scale byte
// E = floor(log2(amax)) - 8 + 127 (the OCP MX rule for e4m3, whose largest binade is 2^8) and 32 e4m3 bytes of x * 2^(127 - E),
// round-to-nearest-even, saturating FINITE values at +-448 (NaN / Inf propagate: see the kernel).  Four lanes per block, 8 elements each: coalesced 16-byte loads and 8-byte stores
// (the first version gave a lane a whole block: 64-byte stride between the lanes of every load instruction), the block maximum by
// two quad shuffles.
__global__ __launch_bounds__(256) void quant_mx8_kernel(const bf16_t* __restrict__ x, long long ld, unsigned char* __restrict__ q, long long q_ld,
                                                        unsigned char* __restrict__ s, long long s_ld, long long rows, int cols) {
  const int ng = cols >> 3;                 // 8-element groups per row (a multiple of 4: cols % 32 == 0)
  const long long total = rows * ng;        // (a multiple of 4, like every stride below: the 4 lanes of a block stay together)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long row = i / ng;
    const int g = (int)(i - row * ng);
    float v[8];
    load8f<bf16_t>(x + row * ld + g * 8, v);
    mx8_quant8(v, q + row * q_ld + g * 8, s + row * s_ld + (g >> 2), (g & 3) == 0);
  }
}

// Many matrices in ONE launch (round 6): the fp8 images of every eligible weight and of its transposed copy, refreshed once per
// optimizer step behind the batched transpose (functional._Fp8Mirror) -- ~280 launches of 2-5 us per update before, each produced by
// whichever micro-batch stream reached the Linear first, with the other stream waiting on an event (a cross-stream edge inside the
// replayed graph per weight).  Contiguous matrices (ld = cols); a block quantises 2048 consecutive elements of one job.
struct QuantJob { const bf16_t* x; unsigned char* q; unsigned char* s; long long elems; int cols; int blk0; };
__global__ __launch_bounds__(256) void multi_quant_mx8_kernel(const QuantJob* __restrict__ jobs, const int njobs) {
  int lo = 0, hi = njobs - 1;      // last job with blk0 <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const QuantJob j = jobs[lo];
  const long long i = ((long long)((int)blockIdx.x - j.blk0) * 256 + threadIdx.x) * 8;     // first element of this lane's group
  if (i >= j.elems) return;                 // (elems % 32 == 0: the four lanes of a block leave together)
  float v[8];
  load8f<bf16_t>(j.x + i, v);
  mx8_quant8(v, j.q + i, j.s + (i >> 5), (i & 31) == 0);
}

}  // namespace

extern "C" int st5_gemm(const st5_gemm_params* pp, int dtype, void* stream) {
  if (!pp) return ST5_ERR_ARG;
  st5_gemm_params p = *pp;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return ST5_ERR_ARG;
  if (!p.A.ptr || !p.B.ptr || !p.C.ptr) return ST5_ERR_ARG;
  if (p.batch <= 0) p.batch = 1;
  if (p.zdiv <= 0) p.zdiv = 1;
  if ((p.flags & ST5_GEMM_DACT) && !p.P.ptr) return ST5_ERR_ARG;
  if (p.asum && p.batch > 1) return ST5_ERR_ARG;
  if (dtype != ST5_BF16 && dtype != ST5_F32) return ST5_ERR_ARG;
  const int es = dtype == ST5_BF16 ? 2 : 4;
  const int vec = 16 / es;
  // operand alignment: 16-byte vectors along the contiguous index
  auto op_ok = [&](const st5_operand& o, bool kstrided) {
    const int64_t q = kstrided ? 4 : vec;
    if (!aligned(o.ptr, (size_t)q * es)) return false;
    if (o.ld % q || o.bstride % q || o.zs0 % q || o.zs1 % q) return false;
    if (o.seg && (o.seg % q || o.seg_stride % q)) return false;
    return true;
  };
  if (!op_ok(p.A, p.flags & ST5_GEMM_A_KSTRIDED)) return ST5_ERR_ALIGN;
  if (!op_ok(p.B, p.flags & ST5_GEMM_B_KSTRIDED)) return ST5_ERR_ALIGN;
  auto c_ok = [&](const st5_operand& o) {
    if (!o.ptr) return true;
    return aligned(o.ptr, 16) && o.ld % 8 == 0 && o.bstride % 8 == 0 && o.zs0 % 8 == 0 && o.zs1 % 8 == 0;
  };
  const int c_vec_ok = c_ok(p.C) && c_ok(p.R) && c_ok(p.P) && c_ok(p.Cpre);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // split-K for fp32 outputs (weight gradients: small M x N, very long K): raise the block count towards ~1.5 per
  // CU; every split writes its own dense fp32 slab (plain 16-byte stores), a second kernel sums the slabs into C.
  // (fp32 atomics into C were measured 1.6x slower than no split at all: device-scope atomics bypass the XCD L2.)
  int nsplit = 1;
  const bool plain = !p.bias && !p.R.ptr && !p.P.ptr && !p.Cpre.ptr && p.act == ST5_ACT_NONE && p.dropout_p == 0.f;
  if ((p.flags & ST5_GEMM_OUT_F32) && plain && p.batch == 1 && p.N % 8 == 0 && p.C.ld % 4 == 0) {
    const long long tiles = (long long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int bk = 128 / es;
    const int nk = (p.K + bk - 1) / bk;
    if (tiles < 200 && nk >= 16) {
      long long want = (g_splitk_target + tiles - 1) / tiles;
      long long maxs = nk / 8;  // at least 8 k-tiles per split
      nsplit = (int)(want < maxs ? want : maxs);
      if (nsplit < 1) nsplit = 1;
      const int per = (nk + nsplit - 1) / nsplit;
      nsplit = (nk + per - 1) / per;
    }
  }
  // weight gradients without row split / segments: the phased 256^2 kernel with its own split count (about one block per CU)
  const bool use8p = dtype == ST5_BF16 && g_use_glds && tn_glds_ok(p, dtype) && tn8p_ok(p, c_vec_ok) && p.C.ld % 4 == 0;
  if (use8p) nsplit = tn8p_splits(p);
  if (nsplit > 1 && g_defer && (p.flags & ST5_GEMM_DEFERRABLE) && !p.C.rpb && p.C.ld % 4 == 0) {
    const size_t need = ((size_t)nsplit * p.M * p.N + (p.asum ? (size_t)nsplit * p.M : 0)) * sizeof(float);
    // the same output twice in one batch (tied weights) would race inside the batched reduction: fold what is pending first
    DeferState* ds = defer_state(s, true);
    if (!ds) return ST5_ERR_LAUNCH;
    for (int j = 0; j < g_pending.n; ++j)
      if (g_pending.d[j].C == p.C.ptr) { const int rc = flush_pending(s); if (rc) return rc; break; }
    if (g_pending.n == MR_MAX) { const int rc = flush_pending(s); if (rc) return rc; }
    float* slabs = arena_take(ds, need);
    if (!slabs) {
      const int rc = flush_pending(s);
      if (rc) return rc;
      if (need > g_arena_bytes || g_arena_bytes == 0) {
        if (g_arena) (void)hipFree(g_arena);   // synchronises with in-flight users
        size_t want = g_arena_bytes ? g_arena_bytes * 2 : (size_t(1) << 30);
        while (want < need * 4) want *= 2;
        if (st5_dev_malloc(&g_arena, want) != hipSuccess) { g_arena = nullptr; g_arena_bytes = 0; return ST5_ERR_LAUNCH; }
        g_arena_bytes = want;
      }
      slabs = arena_take(ds, need);
      if (!slabs) return ST5_ERR_LAUNCH;
    }
    st5_gemm_params q = p;
    q.C.ptr = slabs; q.C.ld = p.N; q.C.rpb = 0; q.C.bstride = 0; q.C.zs0 = q.C.zs1 = 0; q.beta = 0.f;
    const int rc = use8p ? launch_tn8p(q, 1, nsplit, s) : (g_use_glds && tn_glds_ok(q, dtype)) ? launch_tn_glds(q, 1, nsplit, s)
                   : dtype == ST5_BF16 ? launch<bf16_t>(q, 1, nsplit, s) : launch<float>(q, 1, nsplit, s);
    if (rc) return rc;
    MrDesc& d = g_pending.d[g_pending.n++];
    d.slabs = slabs; d.C = reinterpret_cast<float*>(const_cast<void*>(p.C.ptr)); d.asum = p.asum; d.ldc = p.C.ld; d.nsplit = nsplit;
    d.M = p.M; d.N = p.N; d.beta = p.beta; d.blk0 = g_pending_blocks; d.pad = 0;
    long long blocks = ((long long)p.M * p.N / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    g_pending_blocks += (int)blocks;
    return ST5_OK;
  }
  if (nsplit > 1) {
    float* slabs = slab_workspace(((size_t)nsplit * p.M * p.N + (p.asum ? (size_t)nsplit * p.M : 0)) * sizeof(float), s);
    if (!slabs) return ST5_ERR_LAUNCH;
    st5_gemm_params q = p;
    q.C.ptr = slabs; q.C.ld = p.N; q.C.rpb = 0; q.C.bstride = 0; q.C.zs0 = q.C.zs1 = 0; q.beta = 0.f;
    const int rc = use8p ? launch_tn8p(q, 1, nsplit, s) : (g_use_glds && tn_glds_ok(q, dtype)) ? launch_tn_glds(q, 1, nsplit, s)
                   : dtype == ST5_BF16 ? launch<bf16_t>(q, 1, nsplit, s) : launch<float>(q, 1, nsplit, s);
    if (rc) return rc;
    long long blocks = ((long long)p.M * p.N / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, slabs,
                       reinterpret_cast<float*>(const_cast<void*>(p.C.ptr)), nsplit, p.M, p.N, (long long)p.C.ld, p.C.rpb,
                       (long long)p.C.bstride, p.beta, p.asum);
    HIP_CHECK_LAUNCH();
    return ST5_OK;
  }
  // LDS-DMA pipelined fast path for the plain NT form (Linear / conv forward and, with cached transposed weights,
  // the data-gradient GEMMs)
  const int bk = 128 / es;
  if (g_use_glds && !p.asum && !(p.flags & (ST5_GEMM_A_KSTRIDED | ST5_GEMM_B_KSTRIDED)) && p.K % bk == 0 && p.K >= 2 * bk && !p.A.seg && !p.B.seg)
  {
    if ((g_nt_tile == 3 || g_nt_tile == 4) && dtype == ST5_BF16 && p.K % 64 == 0) {
      g_p8_stagger = g_nt_tile == 3;
      return launch_nt8p(p, c_vec_ok, s);
    }
    if (g_nt_tile == 2) return dtype == ST5_BF16 ? launch_nt256<bf16_t>(p, c_vec_ok, s) : launch_nt256<float>(p, c_vec_ok, s);
    if (g_nt_tile == 5 && dtype == ST5_BF16) return launch_m64(p, c_vec_ok, s);
    if (g_nt_tile == 0 && dtype == ST5_BF16 && nt_m64_pays(p) && !nt256_pays(p.M, p.N, p.K / 64, p.batch)) return launch_m64(p, c_vec_ok, s);
    if (g_nt_tile == 0 && nt256_pays(p.M, p.N, p.K / 64, p.batch)) {
      // bf16: the phased kernel (round 4); fp32 parity mode: the first 256^2 kernel (4-stage ring)
      if (dtype == ST5_BF16 && p.K % 64 == 0) { g_p8_stagger = 1; return launch_nt8p(p, c_vec_ok, s); }
      if (dtype != ST5_BF16) return launch_nt256<float>(p, c_vec_ok, s);
    }
    return dtype == ST5_BF16 ? launch_glds<bf16_t>(p, c_vec_ok, s) : launch_glds<float>(p, c_vec_ok, s);
  }
  if (use8p) return launch_tn8p(p, c_vec_ok, nsplit, s);
  if (g_use_glds && tn_glds_ok(p, dtype)) return launch_tn_glds(p, c_vec_ok, nsplit, s);
  if (dtype == ST5_BF16) return launch<bf16_t>(p, c_vec_ok, nsplit, s);
  return launch<float>(p, c_vec_ok, nsplit, s);
}

/* Several weight-gradient GEMMs (C_j [M_j x N_j] (+)= A_j^T B_j: both operands k-strided bf16, fp32 output, no epilogue beyond beta and the
 * bias-gradient column) as ONE launch without split-K (gemm_tn_group_kernel).  A problem's result does not depend on the rest of the
 * group (every tile runs its own whole reduction).  Falls back to one st5_gemm call per problem when a problem does not have that form
 * or two problems write the same output. */
int g_tn_group_tile = 0;   // st5_gemm_set_tn_group_tile: 0 = per problem (phased 256^2 when M, N are multiples of 256 and K >= 512), 1 = 128^2 always
// Does this (group-eligible) problem run on the phased 256 x 256 grouped kernel?  A function of the problem alone.
bool tn_group_phased(const st5_gemm_params& p) {
  return g_tn_group_tile == 0 && p.M % 256 == 0 && p.N % 256 == 0 && p.K >= 512 && p.C.ld % 4 == 0 && !p.A.rpb && !p.B.rpb && !p.A.seg && !p.B.seg;
}
extern "C" int st5_gemm_tn_group(const st5_gemm_params* list, int32_t n, int dtype, void* stream) {
  if (!list || n < 0) return ST5_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_tn8p_group_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_tn8p_group_kernel<F_BETA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return ST5_ERR_LAUNCH;
    attr = true;
  }
  TnGroupArgs g;          // 128 x 128 class: up to TNG_MAX problems
  Tn8pGroupArgs g8;       // phased 256 x 256 class: up to TNG8P_MAX
  int m = 0;             // problems in the open group
  long long padded = 0;
  bool accum = false, phased = false;
  auto launch = [&]() {
    if (m == 0) return (int)ST5_OK;
    if (phased) {      // (first[]: cumulative tile counts, no padding -- the kernel remaps over the whole launch)
      g8.first[m] = (int)padded; g8.n = m;
      for (int i = m; i < TNG8P_MAX; ++i) { g8.tiles[i] = 0; g8.first[i + 1] = (int)padded; }
      if (accum) hipLaunchKernelGGL(gemm_tn8p_group_kernel<F_BETA>, dim3((unsigned)padded), dim3(512), (size_t)8 * TILE_BYTES, s, g8);
      else hipLaunchKernelGGL(gemm_tn8p_group_kernel<0>, dim3((unsigned)padded), dim3(512), (size_t)8 * TILE_BYTES, s, g8);
    } else {
      g.first[m] = (int)padded; g.n = m;
      for (int i = m; i < TNG_MAX; ++i) { g.tiles[i] = 0; g.first[i + 1] = (int)padded; }
      if (accum) hipLaunchKernelGGL(gemm_tn_group_kernel<F_BETA>, dim3((unsigned)padded), dim3(NTHREADS), (size_t)4 * TILE_BYTES, s, g);
      else hipLaunchKernelGGL(gemm_tn_group_kernel<0>, dim3((unsigned)padded), dim3(NTHREADS), (size_t)4 * TILE_BYTES, s, g);
    }
    m = 0; padded = 0;
    HIP_CHECK_LAUNCH();
    return (int)ST5_OK;
  };
  // Whether a problem runs here (whole reduction per tile) or through st5_gemm (split-K), and on which block tile, is decided by the
  // problem ALONE, never by what it is queued with: a replayed update and an eager one, or two micro-batch schedules, form different
  // groups and still have to agree bit for bit.  (No fall-back on the size of a group either; the under-filled ones are the leftovers at
  // the end of a backward pass.)
  for (int i = 0; i < n; ++i) {
    st5_gemm_params p = list[i];
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.A.ptr || !p.B.ptr || !p.C.ptr) return ST5_ERR_ARG;
    if (p.batch <= 0) p.batch = 1;
    if (p.zdiv <= 0) p.zdiv = 1;
    const int want = ST5_GEMM_A_KSTRIDED | ST5_GEMM_B_KSTRIDED | ST5_GEMM_OUT_F32;
    const bool plain = !p.bias && !p.R.ptr && !p.P.ptr && !p.Cpre.ptr && p.act == ST5_ACT_NONE && p.dropout_p == 0.f && !(p.flags & ST5_GEMM_DACT);
    const bool c_ok = aligned(p.C.ptr, 16) && p.C.ld % 8 == 0 && !p.C.rpb && p.C.zs0 == 0 && p.C.zs1 == 0;
    const bool ok = dtype == ST5_BF16 && g_use_glds && (p.flags & want) == want && plain && p.batch == 1 && p.N % 8 == 0 && c_ok &&
                    tn_glds_ok(p, dtype) && (!p.asum || aligned(p.asum, 4));
    if (!ok) {
      const int rc = st5_gemm(&list[i], dtype, stream);
      if (rc) return rc;
      continue;
    }
    const bool ph = tn_group_phased(p) && p.alpha == 1.f && p.A.zs0 == 0 && p.A.zs1 == 0 && p.B.zs0 == 0 && p.B.zs1 == 0;
    bool clash = (m > 0 && m == (phased ? TNG8P_MAX : TNG_MAX)) || (m > 0 && ((p.beta != 0.f) != accum || ph != phased));
    for (int j = 0; j < m && !clash; ++j)
      clash = phased ? (g8.p[j].C == p.C.ptr || (p.asum && g8.p[j].asum == p.asum)) : (g.p[j].C.ptr == p.C.ptr || (p.asum && g.p[j].asum == p.asum));
    if (clash) { const int rc = launch(); if (rc) return rc; }      // (same output twice, mixed beta or block tile: in order, one launch each)
    if (m == 0) { accum = p.beta != 0.f; phased = ph; }
    const int tiles = ph ? (p.M / 256) * (p.N / 256) : ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    if (ph) {
      Tn8pProb& q = g8.p[m];
      q.A = p.A.ptr; q.B = p.B.ptr; q.C = const_cast<void*>(p.C.ptr); q.asum = p.asum; q.lda = p.A.ld; q.ldb = p.B.ld; q.ldc = p.C.ld;
      q.M = p.M; q.N = p.N; q.K = p.K; q.beta = p.beta;
      g8.tiles[m] = tiles; g8.first[m] = (int)padded;
    } else {
      g.p[m] = p; g.tiles[m] = tiles; g.first[m] = (int)padded;
    }
    padded += ph ? tiles : (tiles + 7) / 8 * 8;
    ++m;
  }
  return launch();
}
/* Block tile of the grouped weight-gradient launch: 0 (default) = per problem -- the phased 256x256 kernel when M and N are multiples of
 * 256 and K >= 512 (every Linear of the transformer), else 128x128; 1 = 128x128 always (A/B measurements, tests). */
extern "C" int st5_gemm_set_tn_group_tile(int mode) { if (mode < 0 || mode > 1) return ST5_ERR_ARG; g_tn_group_tile = mode; return ST5_OK; }
/* 1 when st5_gemm_tn_group would run a weight gradient of this shape on the phased 256x256 kernel (host-side tile accounting). */
extern "C" int st5_gemm_tn_group_is_phased(int32_t M, int32_t N, int32_t K) {
  return g_tn_group_tile == 0 && M % 256 == 0 && N % 256 == 0 && K >= 512;
}

/* MX-fp8 GEMM (see gemm_nt_mx8_kernel): p describes fp8 operands A [M x K], B [N x K] (ld in BYTES = elements, K-major, no row
 * split / segments / batch) and bf16 C-class operands; a_scale / b_scale hold one e8m0 byte per 32 k-elements of a row. */
namespace {
int gemm_mxfp8_impl(const st5_gemm_params* pp, const uint8_t* a_scale, int64_t a_scale_ld, const uint8_t* b_scale, int64_t b_scale_ld,
                    const MxOut mo, void* stream) {
  if (!pp || !a_scale || !b_scale) return ST5_ERR_ARG;
  st5_gemm_params p = *pp;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.A.ptr || !p.B.ptr || !p.C.ptr) return ST5_ERR_ARG;
  if (p.K % 128 || (p.batch > 1) || p.asum || (p.flags & (ST5_GEMM_A_KSTRIDED | ST5_GEMM_B_KSTRIDED | ST5_GEMM_OUT_F32))) return ST5_ERR_ARG;
  if (p.A.rpb || p.B.rpb || p.A.seg || p.B.seg) return ST5_ERR_ARG;
  if ((p.flags & ST5_GEMM_DACT) && !p.P.ptr) return ST5_ERR_ARG;
  if (!aligned(p.A.ptr, 16) || !aligned(p.B.ptr, 16) || p.A.ld % 16 || p.B.ld % 16) return ST5_ERR_ALIGN;
  if (!aligned(a_scale, 4) || !aligned(b_scale, 4) || a_scale_ld % 4 || b_scale_ld % 4) return ST5_ERR_ALIGN;
  p.batch = 1; p.zdiv = 1;
  auto c_ok = [&](const st5_operand& o) {
    if (!o.ptr) return true;
    return aligned(o.ptr, 16) && o.ld % 8 == 0 && o.bstride % 8 == 0;
  };
  const int c_vec_ok = c_ok(p.C) && c_ok(p.R) && c_ok(p.P) && c_ok(p.Cpre);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long long al = a_scale_ld, bl = b_scale_ld;
  int feat = nt_feat_of(p, c_vec_ok);
  if (mo.q) {
    // output also MX-quantised in the epilogue: the two producers the training step has (fc1 forward: bias + GELU + pre-activation copy;
    // the data gradient through the GELU), plain layout only
    if (!mo.s || p.N % 32 || p.C.rpb || mo.q_ld % 8 || !aligned(mo.q, 8)) return ST5_ERR_ARG;
    if (feat != (F_GELU | F_PRE) && feat != F_DACT) return ST5_ERR_ARG;
    feat |= F_Q8;
  }
  // Block tile, the bf16 rule (nt256_pays): the phased 256^2 kernel for several rounds of the chip or one nearly full round
  // (the phased kernel addresses operands and scales through buffer resources: 32-bit byte offsets)
  const long long lim = 1ll << 31;
  const bool small_images = (long long)p.M * p.A.ld < lim && (long long)p.N * p.B.ld < lim && (long long)p.M * al < lim && (long long)p.N * bl < lim;
  if (small_images && (g_mx8_tile == 2 || (g_mx8_tile == 0 && mx8_256_pays(p.M, p.N, p.K / 128, feat)))) {
    switch (feat) {
      case F_GELU | F_PRE | F_Q8: return launch_mx8p_as<F_GELU | F_PRE | F_Q8>(p, c_vec_ok, a_scale, al, b_scale, bl, s, mo);
      case F_DACT | F_Q8: return launch_mx8p_as<F_DACT | F_Q8>(p, c_vec_ok, a_scale, al, b_scale, bl, s, mo);
      case 0: return launch_mx8p_as<0>(p, c_vec_ok, a_scale, al, b_scale, bl, s);
      case F_GELU | F_PRE: return launch_mx8p_as<F_GELU | F_PRE>(p, c_vec_ok, a_scale, al, b_scale, bl, s);
      case F_DROP | F_RES: return launch_mx8p_as<F_DROP | F_RES>(p, c_vec_ok, a_scale, al, b_scale, bl, s);
      case F_DACT: return launch_mx8p_as<F_DACT>(p, c_vec_ok, a_scale, al, b_scale, bl, s);
      case F_BETA: return launch_mx8p_as<F_BETA>(p, c_vec_ok, a_scale, al, b_scale, bl, s);
      case F_RES: return launch_mx8p_as<F_RES>(p, c_vec_ok, a_scale, al, b_scale, bl, s);
      default: return launch_mx8p_as<-1>(p, c_vec_ok, a_scale, al, b_scale, bl, s);
    }
  }
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  dim3 grid(tiles, 1, 1);
  switch (feat) {
    case F_GELU | F_PRE | F_Q8: return launch_mx8_as<F_GELU | F_PRE | F_Q8>(p, c_vec_ok, a_scale, al, b_scale, bl, grid, s, mo);
    case F_DACT | F_Q8: return launch_mx8_as<F_DACT | F_Q8>(p, c_vec_ok, a_scale, al, b_scale, bl, grid, s, mo);
    case 0: return launch_mx8_as<0>(p, c_vec_ok, a_scale, al, b_scale, bl, grid, s);
    case F_GELU | F_PRE: return launch_mx8_as<F_GELU | F_PRE>(p, c_vec_ok, a_scale, al, b_scale, bl, grid, s);
    case F_DROP | F_RES: return launch_mx8_as<F_DROP | F_RES>(p, c_vec_ok, a_scale, al, b_scale, bl, grid, s);
    case F_DACT: return launch_mx8_as<F_DACT>(p, c_vec_ok, a_scale, al, b_scale, bl, grid, s);
    case F_BETA: return launch_mx8_as<F_BETA>(p, c_vec_ok, a_scale, al, b_scale, bl, grid, s);
    default: return launch_mx8_as<-1>(p, c_vec_ok, a_scale, al, b_scale, bl, grid, s);
  }
}
}  // namespace
extern "C" int st5_gemm_mxfp8(const st5_gemm_params* pp, const uint8_t* a_scale, int64_t a_scale_ld, const uint8_t* b_scale,
                              int64_t b_scale_ld, void* stream) {
  return gemm_mxfp8_impl(pp, a_scale, a_scale_ld, b_scale, b_scale_ld, MxOut{nullptr, 0, nullptr, 0}, stream);
}
/* st5_gemm_mxfp8 whose bf16 output C [M x N] (N % 32 == 0) is ALSO written as its MX-fp8 image: out_q [M x N] e4m3 bytes (pitch out_q_ld),
 * out_s [M x N/32] e8m0 scale bytes (pitch out_s_ld) -- the bytes st5_quant_mxfp8 would produce from C.  For the two epilogues whose output
 * feeds the next fp8 GEMM as its A operand: bias + GELU with the pre-activation copy (fc1 forward), and x act'(P) (data gradient of fc2). */
extern "C" int st5_gemm_mxfp8_q(const st5_gemm_params* pp, const uint8_t* a_scale, int64_t a_scale_ld, const uint8_t* b_scale,
                                int64_t b_scale_ld, void* out_q, int64_t out_q_ld, uint8_t* out_s, int64_t out_s_ld, void* stream) {
  if (!out_q || !out_s) return ST5_ERR_ARG;
  return gemm_mxfp8_impl(pp, a_scale, a_scale_ld, b_scale, b_scale_ld,
                         MxOut{reinterpret_cast<unsigned char*>(out_q), (long long)out_q_ld, out_s, (long long)out_s_ld}, stream);
}

/* q[r, c] (e4m3 bytes) and s[r, c / 32] (e8m0 bytes) of the bf16 matrix x [rows x cols] (cols % 32 == 0), MX blocks along a row. */
extern "C" int st5_quant_mxfp8(const void* x, int64_t ld, void* q, int64_t q_ld, uint8_t* s, int64_t s_ld, int64_t rows, int32_t cols,
                               void* stream) {
  if (!x || !q || !s || rows < 0 || cols <= 0 || cols % 32) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  if (!aligned(x, 16) || !aligned(q, 16) || ld % 8 || q_ld % 16) return ST5_ERR_ALIGN;
  const long long total = (long long)rows * (cols / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(quant_mx8_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const bf16_t*)x, (long long)ld,
                     (unsigned char*)q, (long long)q_ld, (unsigned char*)s, (long long)s_ld, (long long)rows, (int)cols);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

/* njobs contiguous bf16 matrices quantised as st5_quant_mxfp8 would, in one launch.  jobs: device array of njobs records
 * { const void* x; void* q; uint8_t* s; int64 elems (= rows * cols, cols % 32 == 0); int32 cols; int32 blk0 } with blk0 = the job's first
 * block (2048 elements per block, jobs in ascending blk0 order, blk0 of job 0 == 0); nblocks = total block count. */
extern "C" int st5_multi_quant_mxfp8(const void* jobs, int32_t njobs, int32_t nblocks, void* stream) {
  if (!jobs || njobs < 0 || nblocks < 0) return ST5_ERR_ARG;
  if (njobs == 0 || nblocks == 0) return ST5_OK;
  hipLaunchKernelGGL(multi_quant_mx8_kernel, dim3((unsigned)nblocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const QuantJob*>(jobs), (int)njobs);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

#undef g_pending
#undef g_pending_blocks
#undef g_arena
#undef g_arena_bytes
#undef g_arena_used

/* Deferred split-K reductions (see splitk_multi_reduce_kernel).  While enabled, outputs of split-K GEMMs are complete only
 * after st5_gemm_flush_splitk(); disabling flushes. */
extern "C" int st5_gemm_defer_splitk(int enabled, void* stream) {
  (void)stream;
  if (!enabled && g_defer) {   // fold whatever is queued, every state on its own stream
    for (int i = 0; i < g_ndstates; ++i) { const int rc = flush_state(&g_dstates[i]); if (rc) return rc; }
  }
  g_defer = enabled != 0;
  return ST5_OK;
}
extern "C" int st5_gemm_flush_splitk(void* stream) { return flush_pending(reinterpret_cast<hipStream_t>(stream)); }

/* A/B switch for the LDS-DMA NT kernel (tools/bench_kernels.py uses it for within-process comparisons). */
extern "C" int st5_gemm_set_glds(int enabled) { g_use_glds = enabled != 0; return ST5_OK; }
/* NT block tile: 0 = per-problem choice (default), 1 = 128x128 always, 2 = 256x256 always (A/B measurements). */
/* Block count the split-K choice of the fp32-output (weight-gradient) GEMMs aims for (default 384 = 1.5 per CU). */
extern "C" int st5_gemm_set_splitk_target(int blocks) { if (blocks < 1 || blocks > 4096) return ST5_ERR_ARG; g_splitk_target = blocks; return ST5_OK; }
/* 128^2 NT kernel, grids of more than one block per CU: 5 = ring of five 16 KB operand slots (default), 4 = two whole stages. */
extern "C" int st5_gemm_set_nt_slots(int slots) {
  if (slots != 4 && slots != 5) return ST5_ERR_ARG;
  g_nt_slots5 = slots == 5; return ST5_OK;
}
/* 128^2 NT kernel: grids of at most `max_blocks` blocks run with an `nbuf`-stage operand ring (2 = off, 3, 4). */
extern "C" int st5_gemm_set_deep_ring(int max_blocks, int nbuf) {
  if (max_blocks < 0 || nbuf < 2 || nbuf > 4) return ST5_ERR_ARG;
  g_deep_blocks = max_blocks; g_deep_nbuf = nbuf; return ST5_OK;
}
/* Weight-gradient (TN) GEMMs without row split / segments: 0 (default) = always the 128^2 LDS-DMA kernel; 1 = phased 256^2 kernel,
 * 2 = the same without the stagger of the two m-halves (A/B measurements; the header documents the same default). */
extern "C" int st5_gemm_set_tn_phased(int mode) { if (mode < 0 || mode > 2) return ST5_ERR_ARG; g_tn8p = mode; return ST5_OK; }
extern "C" int st5_gemm_set_nt_tile(int mode) { if (mode < 0 || mode > 5) return ST5_ERR_ARG; g_nt_tile = mode; return ST5_OK; }
/* MX-fp8 NT block tile: 0 = per-problem choice (default), 1 = 128x128 always, 2 = phased 256x256 always (A/B measurements, tests). */
extern "C" int st5_gemm_set_mx8_tile(int mode) { if (mode < 0 || mode > 2) return ST5_ERR_ARG; g_mx8_tile = mode; return ST5_OK; }
/* (A/B) k-tiles of 128 an epilogue-heavy fp8 GEMM needs before the per-problem choice takes the phased 256x256 kernel; default 16, 0 = always. */
extern "C" int st5_gemm_set_mx8_heavy_nk(int nk) { if (nk < 0) return ST5_ERR_ARG; g_mx8_heavy_nk = nk; return ST5_OK; }
extern "C" int st5_gemm_set_nt_longk(int tiles, int nk) { if (tiles < 0 || nk < 1) return ST5_ERR_ARG; g_nt_longk_tiles = tiles; g_nt_longk_nk = nk; return ST5_OK; }
extern "C" int st5_gemm_set_m64_max_tiles(int tiles) { if (tiles < 0) return ST5_ERR_ARG; g_m64_max_tiles = tiles; return ST5_OK; }
