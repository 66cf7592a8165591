// Fused (flash-style) attention for head_dim 64 in bf16 on gfx950: QK^T + Shaw relative-position bias + masks +
// online softmax (+dropout) + PV without materialising the [BH,T,S] score/probability tensors.
// Same math as multihead_attention.py:340-389 (reference) and as the unfused st5_gemm/st5_softmax path, which
// stays as the fp32 parity implementation and as the cross-check of this kernel (tests/test_flash_gpu.py).
//
// Work decomposition (wave64, v_mfma_f32_32x32x16_bf16):
//   * block = 4 waves = 128 queries of one (batch, head); wave w owns queries [q0+32w, q0+32w+32);
//   * scores are computed TRANSPOSED: S^T[key][q] = K.Q^T so that every lane owns ONE query column
//     (q = lane&31; the two half-waves hold interleaved key rows).  Row max / sum are then in-lane plus one
//     cross-half shuffle, the running (m, l) are per-lane scalars and the P^T accumulator registers are directly the
//     B operand of O^T[d][q] += V^T[d][key] . P^T[key][q] -- no LDS round trip for P;
//   * K tiles [64 keys][64 d] and V^T tiles [64 d][64 keys] are staged through LDS (register-prefetched, double
//     buffered, 16-byte XOR-swizzled chunks); V is transposed in registers on the way in;
//   * the relative-position bias q.pe[clip(i-j)] is gathered from QP = scale*q.pe^T, computed once per wave with
//     MFMAs into a per-wave LDS table [32 q][2*maxrel] (bf16) instead of the reference's [T,T,64] gather.
#include "common.h"
#include "../../include/speecht5_hip.h"

namespace {

#ifndef FLASH_ABL
#define FLASH_ABL 0
#endif
#ifdef FLASH_TIMING
#define TPROBE(i) do { __builtin_amdgcn_s_waitcnt(0xC07F); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += (unsigned int)(t_ - tprev); tprev = t_; } while (0)
#else
#define TPROBE(i)
#endif

constexpr int HD = 64;
constexpr int KT = 64;  // keys per tile
constexpr int TILE_B = KT * HD * 2;  // 8 KB

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

struct Args {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o; float* lse;
  const bf16_t* pe; const uint8_t* kpm;
  bf16_t* qp_out;                           // optional [BH,T,nb]: the bucket table scale*log2e*q.pe^T, saved for the backward
  long long q_ld, k_ld, v_ld, o_ld;
  int B, H, T, S, nb, maxrel, causal, lds;
  float scale, dropout_p;
  unsigned long long seed;
};

// K tile: 2 x 16 B per thread; V^T tile: 4 keys x 4 d per thread (4 x 8 B loads, transposed in registers).
// Per-thread element offsets are computed once (init); a tile adds j0 * ld.  Rows past S load as zeros.
struct KVStage {
  u32x4 kreg[2];
  u32x2 vreg[4];
  unsigned int koff[2], voff[4];
  int krow[2], vrow[4];
  __device__ __forceinline__ void init(const Args& a, int b, int h, int tid) {
    const int chunk = tid & 7;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      krow[p] = (tid >> 3) + 32 * p;
      koff[p] = (unsigned int)(((long long)b * a.S + krow[p]) * a.k_ld + h * HD + chunk * 8);
    }
    const int dg = tid & 15, kg = tid >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      vrow[kk] = kg * 4 + kk;
      voff[kk] = (unsigned int)(((long long)b * a.S + vrow[kk]) * a.v_ld + h * HD + dg * 4);
    }
  }
  __device__ __forceinline__ void load(const Args& a, int j0) {
    const unsigned int kadd = (unsigned int)j0 * (unsigned int)a.k_ld, vadd = (unsigned int)j0 * (unsigned int)a.v_ld;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      u32x4 z = {0u, 0u, 0u, 0u};
      if (j0 + krow[p] < a.S) z = *reinterpret_cast<const u32x4*>(a.k + (koff[p] + kadd));
      kreg[p] = z;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      u32x2 z = {0u, 0u};
      if (j0 + vrow[kk] < a.S) z = *reinterpret_cast<const u32x2*>(a.v + (voff[kk] + vadd));
      vreg[kk] = z;
    }
  }
  __device__ __forceinline__ void store(char* kt, char* vt, int tid) const {
    const int chunk = tid & 7;
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(kt + lds_off((tid >> 3) + 32 * p, chunk)) = kreg[p];
    const int dg = tid & 15, kg = tid >> 4;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {  // d = dg*4 + rr holds keys kg*4 .. kg*4+3
      u32x2 o;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const unsigned int x = vreg[2 * w][rr >> 1], y = vreg[2 * w + 1][rr >> 1];
        o[w] = (rr & 1) ? ((x >> 16) | (y & 0xffff0000u)) : ((x & 0xffffu) | (y << 16));
      }
      *reinterpret_cast<u32x2*>(vt + lds_off(dg * 4 + rr, kg >> 1) + (kg & 1) * 8) = o;
    }
  }
};

__device__ __forceinline__ bf16x8 pack8(const f32x16& s, int base) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16_t)s[base + j];
  return r;
}


constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr unsigned int PAIR_MUL = 0x9E3779B1u;  // drop_pair_bits' pair multiplier (common.h)

// Relative-position bias handling of one (32-query wave) x (64-key tile) rectangle, chosen wave-uniformly:
//   BM_LIN   every delta = q - key lies strictly inside (-maxrel, maxrel-1): bucket = delta + maxrel, no clipping, the
//            table index is lane_base + compile-time offset;
//   BM_CONST every delta clips to the same end bucket: one table value per lane for the whole tile;
//   BM_GEN   mixed: clip per element.
enum { BM_NONE = 0, BM_GEN = 1, BM_LIN = 2, BM_CONST = 3 };
__device__ __forceinline__ int bias_mode(int dmin, int dmax, int maxrel) {
  if (dmax <= -maxrel || dmin >= maxrel - 1) return BM_CONST;
  if (dmin > -maxrel && dmax < maxrel - 1) return BM_LIN;
  return BM_GEN;
}
__device__ __forceinline__ int clip_rel(int d, int maxrel) { return d < -maxrel ? -maxrel : (d > maxrel - 1 ? maxrel - 1 : d); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float bf16_lds(const bf16_t* p) { return (float)*p; }

// Raw bf16 bits of the lane's 32 relative-position bias values of one tile, read from the per-wave LDS table ahead of
// the QK^T MFMAs.  Element i = 16t + r is key offset c + 4*hi with c = 32t + (r&3) + 8(r>>2); rel0 = q - j0 - 4*hi so
// delta = rel0 - c.
template <int BM>
__device__ __forceinline__ void load_bias_lds(unsigned int (&braw)[32], const bf16_t* qrow, int rel0, int maxrel) {
  const unsigned short* row = reinterpret_cast<const unsigned short*>(qrow);
  if (BM == BM_CONST) {
    const unsigned int v = row[clip_rel(rel0, maxrel) + maxrel];
#pragma unroll
    for (int i = 0; i < 32; ++i) braw[i] = v;
    return;
  }
  const unsigned short* lin = row + (rel0 + maxrel - 63);
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int c = (i >> 4) * 32 + (i & 3) + 8 * ((i & 15) >> 2);
    if (BM == BM_LIN) braw[i] = lin[63 - c];
    else braw[i] = row[clip_rel(rel0 - c, maxrel) + maxrel];
  }
}
__device__ __forceinline__ void load_bias_lds_mode(int bm, unsigned int (&braw)[32], const bf16_t* qrow, int rel0, int maxrel) {
  if (bm == BM_LIN) load_bias_lds<BM_LIN>(braw, qrow, rel0, maxrel);
  else if (bm == BM_CONST) load_bias_lds<BM_CONST>(braw, qrow, rel0, maxrel);
  else if (bm == BM_GEN) load_bias_lds<BM_GEN>(braw, qrow, rel0, maxrel);
}

// Scores of one tile in the log2 domain: x = s*scale*log2e + bias (+ masks); returns the lane's tile max.
// km: this lane's bad-key bits (bit c <=> key offset c + 4*hi unusable); jrel: offsets c > jrel are causally masked.
template <bool BIAS, bool MASK>
__device__ __forceinline__ float tile_scores(f32x16& s0, f32x16& s1, float sc2, const unsigned int (&braw)[32],
                                             unsigned long long km, int jrel) {
  float tmax = -INFINITY;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = t * 32 + (r & 3) + 8 * (r >> 2);
      const float b = BIAS ? __uint_as_float(braw[16 * t + r] << 16) : 0.f;
      float x = fmaf(t == 0 ? s0[r] : s1[r], sc2, b);
      if (MASK) { if (((km >> c) & 1ull) || c > jrel) x = -INFINITY; }
      if (t == 0) s0[r] = x; else s1[r] = x;
      tmax = fmaxf(tmax, x);
    }
  }
  return tmax;
}

// softmax numerators of one tile (log2 domain) + dropout; returns the lane's (undropped) sum
template <bool DROP>
__device__ __forceinline__ float tile_probs(f32x16& s0, f32x16& s1, float m_use, unsigned int key32, unsigned int hoff, unsigned int thresh) {
  float psum = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      unsigned int bits0 = 0u, bits1 = 0u;
      if (DROP) {   // pairs (16t + 4g + 2hi) and +1 of this row's 64-key block
        bits0 = drop_pair_bits_pc(key32, (unsigned int)(16 * t + 4 * g) * PAIR_MUL + hoff);
        bits1 = drop_pair_bits_pc(key32, (unsigned int)(16 * t + 4 * g + 1) * PAIR_MUL + hoff);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        float p = fast_exp2((t == 0 ? s0[r] : s1[r]) - m_use);  // exp2(-inf) = 0 for masked keys
        psum += p;
        if (DROP) p = drop_keep(e < 2 ? bits0 : bits1, e & 1, thresh) ? p : 0.f;   // 1/keep is applied to O at the end
        if (t == 0) s0[r] = p; else s1[r] = p;
      }
    }
  }
  return psum;
}

// Raw key-padding byte of key `key` (0 without a mask).  The per-tile wave-uniform 64-bit mask of unusable keys is
// ballot(raw != 0 || key >= S); the byte is fetched one tile ahead and only tested a tile later, so its latency (and the
// in-order vmcnt wait behind the K/V prefetch) stays off the critical path.
__device__ __forceinline__ unsigned int kpm_raw(const uint8_t* mrow, int key, int S) {
  return mrow ? (unsigned int)mrow[key < S ? key : S - 1] : 0u;
}

// QP^T[bucket][q] = sc2 * pe . q^T -> per-wave LDS table [32 q][qp_ld] (bf16); pe rows are fetched one bucket tile ahead
__device__ __forceinline__ void build_qp_table(bf16_t* qp, int qp_ld, const bf16_t* pe, int nb, float sc2, const bf16x8 (&qf)[4],
                                               int ql, int hi, bf16_t* gout = nullptr) {
  const int nbt = (nb + 31) / 32;
  bf16x8 pf[4], pn[4];
  {
    const int brow = ql < nb ? ql : nb - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) pf[ks] = *reinterpret_cast<const bf16x8*>(pe + brow * HD + ks * 16 + hi * 8);
  }
  for (int bt = 0; bt < nbt; ++bt) {
    if (bt + 1 < nbt) {
      int brow = (bt + 1) * 32 + ql;
      brow = brow < nb ? brow : nb - 1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) pn[ks] = *reinterpret_cast<const bf16x8*>(pe + brow * HD + ks * 16 + hi * 8);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[ks], qf[ks], acc, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int b0 = bt * 32 + 8 * g + 4 * hi;
      if (b0 < nb) {
        bf16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (bf16_t)(acc[4 * g + e] * sc2);
        *reinterpret_cast<bf16x4*>(qp + ql * qp_ld + b0) = w;
        if (gout) *reinterpret_cast<bf16x4*>(gout + b0) = w;   // this lane's query row of the global copy
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) pf[ks] = pn[ks];
  }
}

template <bool BIAS>
__global__ __launch_bounds__(256, BIAS ? 1 : 2) void flash_fwd_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kbuf = smem;                 // 2 x 8 KB
  char* vbuf = smem + 2 * TILE_B;    // 2 x 8 KB
  const int qp_ld = a.nb + 4;        // bf16 elements per row (8-byte aligned rows)
  bf16_t* qp = reinterpret_cast<bf16_t*>(smem + 4 * TILE_B) + (threadIdx.x >> 6) * 32 * qp_ld;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
  const int qblk = blockIdx.x * 128;
  const int qi = qblk + wave * 32 + ql;          // this lane's query
  const int qc = qi < a.T ? qi : a.T - 1;        // clamped for loads
  const bool qvalid = qi < a.T;

#ifdef FLASH_TIMING
  unsigned int tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
  // Q fragments: 4 k-steps x 8 bf16
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(a.q + ((long long)b * a.T + qc) * a.q_ld + h * HD + ks * 16 + hi * 8);

  // number of key tiles this block needs
  int nkeys = a.S;
  if (a.causal) {
    const int qmax = (qblk + 127 < a.T ? qblk + 127 : a.T - 1) + (a.S - a.T);
    nkeys = qmax + 1 < a.S ? qmax + 1 : a.S;
  }
  const int ntiles = (nkeys + KT - 1) / KT;

  KVStage st;
  st.init(a, b, h, tid);
  st.load(a, 0);
  const uint8_t* mrow = a.kpm ? a.kpm + (long long)b * a.S : nullptr;
  unsigned int raw_next = kpm_raw(mrow, lane, a.S);

  if (BIAS) build_qp_table(qp, qp_ld, a.pe, a.nb, a.scale * LOG2E, qf, ql, hi,
                           (a.qp_out && qvalid) ? a.qp_out + ((long long)bh * a.T + qi) * a.nb : nullptr);
  st.store(kbuf, vbuf, tid);
  __syncthreads();

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;   // running max in the log2 domain
  const bool drop = a.dropout_p > 0.f;
  const unsigned int thresh = drop ? dropout_thresh(a.dropout_p) : 0u;
  const float inv_keep = drop ? 1.f / (1.f - a.dropout_p) : 1.f;
  const float sc2 = a.scale * LOG2E;
  const int jmax = a.causal ? qi + (a.S - a.T) : 0x3fffffff;
  const int qw0 = qblk + __builtin_amdgcn_readfirstlane(wave) * 32;
  const unsigned long long ctr_blk = ((unsigned long long)bh * a.T + (unsigned long long)qc) * (unsigned long long)(drop_row_stride(a.lds) >> 6);
  const unsigned int hoff = hi ? 2u * PAIR_MUL : 0u;
  const bf16_t* qrow = qp + ql * qp_ld;

  TPROBE(10);
  for (int jt = 0; jt < ntiles; ++jt) {
    const char* kt = kbuf + (jt & 1) * TILE_B;
    const char* vt = vbuf + (jt & 1) * TILE_B;
    const int j0 = jt * KT;
    const unsigned long long kmask = __ballot(raw_next != 0u || j0 + lane >= a.S);
    if (jt + 1 < ntiles) {
      st.load(a, j0 + KT);
      raw_next = kpm_raw(mrow, j0 + KT + lane, a.S);
    }
    TPROBE(7);
    // issue every LDS read of the QK^T phase first (K fragments, then the bias values), then run the MFMAs
    bf16x8 kfa[4], kfb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kfa[ks] = *reinterpret_cast<const bf16x8*>(kt + lds_off(ql, 2 * ks + hi));
      kfb[ks] = *reinterpret_cast<const bf16x8*>(kt + lds_off(32 + ql, 2 * ks + hi));
    }
    const bool need_mask = kmask != 0ull || (a.causal && j0 + 63 > qw0 + (a.S - a.T));
    const int rel0 = qi - j0 - 4 * hi;
    const int jrel = jmax - j0 - 4 * hi;
    const unsigned long long km = kmask >> (4 * hi);
    unsigned int braw[32];
    if (BIAS) load_bias_lds_mode(bias_mode(qw0 - j0 - 63, qw0 + 31 - j0, a.maxrel), braw, qrow, rel0, a.maxrel);
    TPROBE(0);
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[ks], qf[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfb[ks], qf[ks], s1, 0, 0, 0);
    }
    // V^T fragments for the PV phase: in flight during the softmax arithmetic
    bf16x4 vlo[4][2], vhi[4][2];
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        vlo[sidx][dt] = *reinterpret_cast<const bf16x4*>(vt + lds_off(32 * dt + ql, 2 * sidx) + 8 * hi);
        vhi[sidx][dt] = *reinterpret_cast<const bf16x4*>(vt + lds_off(32 * dt + ql, 2 * sidx + 1) + 8 * hi);
      }
    }
    TPROBE(1);
    // scale, bias, masks, tile max (all in the log2 domain)
    float tmax;
    if (need_mask) tmax = tile_scores<BIAS, true>(s0, s1, sc2, braw, km, jrel);
    else tmax = tile_scores<BIAS, false>(s0, s1, sc2, braw, km, jrel);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = m_run == -INFINITY ? 0.f : fast_exp2(m_run - m_use);
    TPROBE(2);
    float psum;
    if (drop) psum = tile_probs<true>(s0, s1, m_use, drop_block_key(a.seed, ctr_blk + (unsigned long long)jt), hoff, thresh);
    else psum = tile_probs<false>(s0, s1, m_use, 0u, hoff, thresh);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    TPROBE(3);
    if (__ballot(alpha != 1.f)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    }
    // O^T += V^T . P^T   (4 k-steps of 16 keys, two 32-row d tiles)
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      const bf16x8 pf = pack8(sidx < 2 ? s0 : s1, 8 * (sidx & 1));
      bf16x8 v0, v1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v0[e] = vlo[sidx][0][e]; v0[4 + e] = vhi[sidx][0][e];
        v1[e] = vlo[sidx][1][e]; v1[4 + e] = vhi[sidx][1][e];
      }
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf, o1, 0, 0, 0);
    }
    TPROBE(4);
    if (jt + 1 < ntiles) st.store(kbuf + ((jt + 1) & 1) * TILE_B, vbuf + ((jt + 1) & 1) * TILE_B, tid);
    TPROBE(5);
    __syncthreads();
    TPROBE(6);
  }

  // finalize: combine the two half-wave partial sums, normalise, store O and the log-sum-exp
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? inv_keep / l_tot : 0.f;
  if (qvalid) {
    bf16_t* orow = a.o + ((long long)b * a.T + qi) * a.o_ld + h * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (bf16_t)((dt == 0 ? o0[4 * g + e] : o1[4 * g + e]) * inv);
        *reinterpret_cast<bf16x4*>(orow + dt * 32 + 8 * g + 4 * hi) = w;
      }
    }
    if (hi == 0 && a.lse) a.lse[(long long)bh * a.T + qi] = l_tot > 0.f ? m_run * LN2 + __logf(l_tot) : INFINITY;
  }
#ifdef FLASH_TIMING
  __syncthreads();
  TPROBE(11);
  if (lane < 12 && a.lse) { a.lse[(long long)bh * a.T + qblk + wave * 32 + 8 + lane] = (float)tacc[lane]; }
#endif
}


// =====================================================================================================
// Backward.  Two kernels that recompute P from (Q, K, bias, LSE); D[bh,q] = dO[q].O[q] is computed by the first:
//   * flash_bwd_dq_kernel : per 128-query block, loops over key tiles (same orientation as the forward:
//     every lane owns one query) -> dQ, and the relative-position bucket gradients dQP;
//   * flash_bwd_dkv_kernel: per 128-key block, loops over query tiles (every lane owns one key) -> dK, dV.
// No atomics: each output element has exactly one writer.
// =====================================================================================================
struct BwdArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* o; const bf16_t* dout;
  bf16_t* dq; bf16_t* dk; bf16_t* dv;
  const float* lse; float* dvec;            // dvec: D[bh*T + q]
  const bf16_t* pe; const bf16_t* qp;       // qp: scale*q.pe^T [BH,T,nb] (global, used by the dkv kernel)
  bf16_t* dqp;                              // [BH,T,nb], zero-initialised by the host
  const uint8_t* kpm;
  long long q_ld, k_ld, v_ld, o_ld, do_ld, dq_ld, dk_ld, dv_ld;
  int B, H, T, S, nb, maxrel, causal, lds;
  int write_dvec;                           // dq kernel publishes D (single-stream order: dq, then dkv)
  float scale, dropout_p;
  unsigned long long seed;
};

// D[bh,q] = dO[q] . O[q] on its own (two-stream backward: dq and dkv run side by side, both need D up front).  Two
// lanes per row, same summation order as the dq kernel's in-register version (bit-identical D).
__global__ __launch_bounds__(256) void flash_dvec_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                         float* __restrict__ dvec, long long o_ld, long long do_ld, int H,
                                                         int T, long long rows) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long r = t >> 1;
  const int hi = (int)(t & 1);
  float dsum = 0.f;
  if (r < rows) {
    const long long bh = r / T, q = r % T;
    const long long b = bh / H, h = bh % H;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 df = *reinterpret_cast<const bf16x8*>(dout + (b * T + q) * do_ld + h * HD + ks * 16 + hi * 8);
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(o + (b * T + q) * o_ld + h * HD + ks * 16 + hi * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) dsum = fmaf((float)df[e], (float)of[e], dsum);
    }
  }
  dsum += __shfl_xor(dsum, 1, 64);
  if (r < rows && hi == 0) dvec[r] = dsum;
}

// stage of one key tile for the dq kernel: K [key][d], K^T [d][key], V [key][d]
struct KVStageBwd {
  u32x4 kreg[2], vreg[2];
  u32x2 ktreg[4];
  unsigned int koff[2], voff[2], ktoff[4];
  int krow[2], ktrow[4];
  __device__ __forceinline__ void init(const BwdArgs& a, int b, int h, int tid) {
    const int chunk = tid & 7;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      krow[p] = (tid >> 3) + 32 * p;
      koff[p] = (unsigned int)(((long long)b * a.S + krow[p]) * a.k_ld + h * HD + chunk * 8);
      voff[p] = (unsigned int)(((long long)b * a.S + krow[p]) * a.v_ld + h * HD + chunk * 8);
    }
    const int dg = tid & 15, kg = tid >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      ktrow[kk] = kg * 4 + kk;
      ktoff[kk] = (unsigned int)(((long long)b * a.S + ktrow[kk]) * a.k_ld + h * HD + dg * 4);
    }
  }
  __device__ __forceinline__ void load(const BwdArgs& a, int j0) {
    const unsigned int kadd = (unsigned int)j0 * (unsigned int)a.k_ld, vadd = (unsigned int)j0 * (unsigned int)a.v_ld;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      u32x4 zk = {0u, 0u, 0u, 0u}, zv = {0u, 0u, 0u, 0u};
      if (j0 + krow[p] < a.S) {
        zk = *reinterpret_cast<const u32x4*>(a.k + (koff[p] + kadd));
        zv = *reinterpret_cast<const u32x4*>(a.v + (voff[p] + vadd));
      }
      kreg[p] = zk; vreg[p] = zv;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      u32x2 z = {0u, 0u};
      if (j0 + ktrow[kk] < a.S) z = *reinterpret_cast<const u32x2*>(a.k + (ktoff[kk] + kadd));
      ktreg[kk] = z;
    }
  }
  __device__ __forceinline__ void store(char* kt, char* ktt, char* vt, int tid) const {
    const int chunk = tid & 7;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      *reinterpret_cast<u32x4*>(kt + lds_off((tid >> 3) + 32 * p, chunk)) = kreg[p];
      *reinterpret_cast<u32x4*>(vt + lds_off((tid >> 3) + 32 * p, chunk)) = vreg[p];
    }
    const int dg = tid & 15, kg = tid >> 4;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      u32x2 o;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const unsigned int x = ktreg[2 * w][rr >> 1], y = ktreg[2 * w + 1][rr >> 1];
        o[w] = (rr & 1) ? ((x >> 16) | (y & 0xffff0000u)) : ((x & 0xffffu) | (y << 16));
      }
      *reinterpret_cast<u32x2*>(ktt + lds_off(dg * 4 + rr, kg >> 1) + (kg & 1) * 8) = o;
    }
  }
};

__device__ __forceinline__ bf16x8 read_t8(const char* tile, int row, int s, int hi) {
  // 8 k-slot values of a transposed [64][64] tile for k-step s: two 8-byte reads
  const bf16x4 lo = *reinterpret_cast<const bf16x4*>(tile + lds_off(row, 2 * s) + 8 * hi);
  const bf16x4 hi4 = *reinterpret_cast<const bf16x4*>(tile + lds_off(row, 2 * s + 1) + 8 * hi);
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = lo[e]; r[4 + e] = hi4[e]; }
  return r;
}


// dS of one tile for the dq kernel.  In: s = raw q.k scores, dp = dO.V^T; out: s = dS = P * (dP_drop - D) (natural-log
// domain gradient of the logits).  The clipped end buckets' gradient (delta <= -maxrel / >= maxrel-1) is summed in-lane
// into acc_lo / acc_hi; the unclipped buckets are written by the dkv kernel (coalesced there).
template <int BM, bool MASK, bool DROP>
__device__ __forceinline__ void dq_tile(f32x16& s0, f32x16& s1, const f32x16& p0, const f32x16& p1, float sc2,
                                        const unsigned int (&braw)[32], int rel0, int maxrel, unsigned long long km, int jrel,
                                        float lse2, float dsum, unsigned int key32, unsigned int hoff, unsigned int thresh,
                                        float inv_keep, float& acc_lo, float& acc_hi) {
  // (no mul + add contraction here: `acc += p * (dp - D)` must round the product first in BOTH generations of the kernels, or the
  // clipped-bucket sums differ in the last place between them -- tests/test_flash_gpu.py holds them bit-identical)
#pragma clang fp contract(off)
  float csum = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      unsigned int bits0 = 0u, bits1 = 0u;
      if (DROP) {
        bits0 = drop_pair_bits_pc(key32, (unsigned int)(16 * t + 4 * g) * PAIR_MUL + hoff);
        bits1 = drop_pair_bits_pc(key32, (unsigned int)(16 * t + 4 * g + 1) * PAIR_MUL + hoff);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const int c = t * 32 + (r & 3) + 8 * (r >> 2);
        const float b = BM == BM_NONE ? 0.f : __uint_as_float(braw[16 * t + r] << 16);
        const float x = fmaf(t == 0 ? s0[r] : s1[r], sc2, b);
        float p = fast_exp2(x - lse2);
        if (MASK) { if (((km >> c) & 1ull) || c > jrel) p = 0.f; }
        float dp = t == 0 ? p0[r] : p1[r];
        if (DROP) dp = drop_keep(e < 2 ? bits0 : bits1, e & 1, thresh) ? dp * inv_keep : 0.f;
        const float ds = p * (dp - dsum);
        if (t == 0) s0[r] = ds; else s1[r] = ds;
        if (BM == BM_CONST) csum += ds;
        if (BM == BM_GEN) {
          const int d = rel0 - c;
          if (d <= -maxrel) acc_lo += ds;
          else if (d >= maxrel - 1) acc_hi += ds;
        }
      }
    }
  }
  if (BM == BM_CONST) { if (rel0 < 0) acc_lo += csum; else acc_hi += csum; }
}

template <bool BIAS>
__global__ __launch_bounds__(256) void flash_bwd_dq_kernel(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // per buffer: K (8K) | K^T (8K) | V (8K)
  const int qp_ld = a.nb + 4;
  bf16_t* qp = reinterpret_cast<bf16_t*>(smem + 6 * TILE_B) + (threadIdx.x >> 6) * 32 * qp_ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
  const int qblk = blockIdx.x * 128;
  const int qi = qblk + wave * 32 + ql;
  const int qc = qi < a.T ? qi : a.T - 1;
  const bool qvalid = qi < a.T;

  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(a.q + ((long long)b * a.T + qc) * a.q_ld + h * HD + ks * 16 + hi * 8);
    dof[ks] = *reinterpret_cast<const bf16x8*>(a.dout + ((long long)b * a.T + qc) * a.do_ld + h * HD + ks * 16 + hi * 8);
  }
  const float lse = a.lse[(long long)bh * a.T + qc];
  // D[q] = dO[q] . O[q] (the softmax-backward row term), computed here from the fragments this lane already holds and
  // published for the dkv kernel (which runs after this one on the same stream) -- no separate preparation launch
  float dsum = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bf16x8 of = *reinterpret_cast<const bf16x8*>(a.o + ((long long)b * a.T + qc) * a.o_ld + h * HD + ks * 16 + hi * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) dsum = fmaf((float)dof[ks][e], (float)of[e], dsum);
  }
  dsum += __shfl_xor(dsum, 32, 64);
  if (a.write_dvec && qvalid && hi == 0) a.dvec[(long long)bh * a.T + qi] = dsum;
  bf16_t* dqp_row = a.dqp ? a.dqp + ((long long)bh * a.T + qc) * a.nb : nullptr;

  int nkeys = a.S;
  if (a.causal) {
    const int qmax = (qblk + 127 < a.T ? qblk + 127 : a.T - 1) + (a.S - a.T);
    nkeys = qmax + 1 < a.S ? qmax + 1 : a.S;
  }
  const int ntiles = (nkeys + KT - 1) / KT;
  KVStageBwd st;
  st.init(a, b, h, tid);
  st.load(a, 0);
  const uint8_t* mrow = a.kpm ? a.kpm + (long long)b * a.S : nullptr;
  unsigned int raw_next = kpm_raw(mrow, lane, a.S);
  if (BIAS) build_qp_table(qp, qp_ld, a.pe, a.nb, a.scale * LOG2E, qf, ql, hi);
  st.store(smem, smem + TILE_B, smem + 2 * TILE_B, tid);
  __syncthreads();

  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  float acc_lo = 0.f, acc_hi = 0.f;  // clipped relative-position buckets 0 and nb-1
  const bool drop = a.dropout_p > 0.f;
  const unsigned int thresh = drop ? dropout_thresh(a.dropout_p) : 0u;
  const float inv_keep = drop ? 1.f / (1.f - a.dropout_p) : 1.f;
  const float sc2 = a.scale * LOG2E;
  const float lse2 = lse * LOG2E;     // +inf for fully masked rows -> P = 0
  const int jmax = a.causal ? qi + (a.S - a.T) : 0x3fffffff;
  const int qw0 = qblk + __builtin_amdgcn_readfirstlane(wave) * 32;
  const unsigned long long ctr_blk = ((unsigned long long)bh * a.T + (unsigned long long)qc) * (unsigned long long)(drop_row_stride(a.lds) >> 6);
  const unsigned int hoff = hi ? 2u * PAIR_MUL : 0u;
  const bf16_t* qrow = qp + ql * qp_ld;

  for (int jt = 0; jt < ntiles; ++jt) {
    const char* buf = smem + (jt & 1) * 3 * TILE_B;
    const char* kt = buf; const char* ktt = buf + TILE_B; const char* vt = buf + 2 * TILE_B;
    const int j0 = jt * KT;
    const unsigned long long kmask = __ballot(raw_next != 0u || j0 + lane >= a.S);
    if (jt + 1 < ntiles) {
      st.load(a, j0 + KT);
      raw_next = kpm_raw(mrow, j0 + KT + lane, a.S);
    }
    // issue the LDS reads of the S / dP phase (K, V fragments, bias), then the MFMAs
    bf16x8 kfa[4], kfb[4], vfa[4], vfb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kfa[ks] = *reinterpret_cast<const bf16x8*>(kt + lds_off(ql, 2 * ks + hi));
      kfb[ks] = *reinterpret_cast<const bf16x8*>(kt + lds_off(32 + ql, 2 * ks + hi));
      vfa[ks] = *reinterpret_cast<const bf16x8*>(vt + lds_off(ql, 2 * ks + hi));
      vfb[ks] = *reinterpret_cast<const bf16x8*>(vt + lds_off(32 + ql, 2 * ks + hi));
    }
    const bool need_mask = kmask != 0ull || (a.causal && j0 + 63 > qw0 + (a.S - a.T));
    const int rel0 = qi - j0 - 4 * hi;
    const int jrel = jmax - j0 - 4 * hi;
    const unsigned long long km = kmask >> (4 * hi);
    const int bm = BIAS ? bias_mode(qw0 - j0 - 63, qw0 + 31 - j0, a.maxrel) : BM_NONE;
    unsigned int braw[32];
    if (BIAS) load_bias_lds_mode(bm, braw, qrow, rel0, a.maxrel);
    f32x16 s0, s1, p0, p1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; p0[r] = 0.f; p1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[ks], qf[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfb[ks], qf[ks], s1, 0, 0, 0);
      p0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfa[ks], dof[ks], p0, 0, 0, 0);   // dP^T[key][q]
      p1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfb[ks], dof[ks], p1, 0, 0, 0);
    }
    // K^T fragments of the dQ phase: in flight during the element-wise work
    bf16x8 ktf[4][2];
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      ktf[sidx][0] = read_t8(ktt, ql, sidx, hi);
      ktf[sidx][1] = read_t8(ktt, 32 + ql, sidx, hi);
    }
    unsigned int key32 = 0u;
    if (drop) key32 = drop_block_key(a.seed, ctr_blk + (unsigned long long)jt);
#define DQ_TILE(BM_, MASK_, DROP_) \
  dq_tile<BM_, MASK_, DROP_>(s0, s1, p0, p1, sc2, braw, rel0, a.maxrel, km, jrel, lse2, dsum, key32, hoff, thresh, inv_keep, acc_lo, acc_hi)
    if (drop) {
      if (need_mask) { if (bm == BM_NONE) DQ_TILE(BM_NONE, true, true); else if (bm == BM_LIN) DQ_TILE(BM_LIN, true, true); else if (bm == BM_CONST) DQ_TILE(BM_CONST, true, true); else DQ_TILE(BM_GEN, true, true); }
      else if (bm == BM_NONE) DQ_TILE(BM_NONE, false, true);
      else if (bm == BM_LIN) DQ_TILE(BM_LIN, false, true);
      else if (bm == BM_CONST) DQ_TILE(BM_CONST, false, true);
      else DQ_TILE(BM_GEN, false, true);
    } else {
      if (need_mask) { if (bm == BM_NONE) DQ_TILE(BM_NONE, true, false); else if (bm == BM_LIN) DQ_TILE(BM_LIN, true, false); else if (bm == BM_CONST) DQ_TILE(BM_CONST, true, false); else DQ_TILE(BM_GEN, true, false); }
      else if (bm == BM_NONE) DQ_TILE(BM_NONE, false, false);
      else if (bm == BM_LIN) DQ_TILE(BM_LIN, false, false);
      else if (bm == BM_CONST) DQ_TILE(BM_CONST, false, false);
      else DQ_TILE(BM_GEN, false, false);
    }
#undef DQ_TILE
    // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      const bf16x8 df = pack8(sidx < 2 ? s0 : s1, 8 * (sidx & 1));
      dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[sidx][0], df, dq0, 0, 0, 0);
      dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[sidx][1], df, dq1, 0, 0, 0);
    }
    if (jt + 1 < ntiles) {
      char* nb_ = smem + ((jt + 1) & 1) * 3 * TILE_B;
      st.store(nb_, nb_ + TILE_B, nb_ + 2 * TILE_B, tid);
    }
    __syncthreads();
  }
  if (BIAS) {
    acc_lo += __shfl_xor(acc_lo, 32, 64);
    acc_hi += __shfl_xor(acc_hi, 32, 64);
    if (qvalid && hi == 0) { dqp_row[0] = (bf16_t)acc_lo; dqp_row[a.nb - 1] = (bf16_t)acc_hi; }
  }
  if (qvalid) {
    bf16_t* row = a.dq + ((long long)b * a.T + qi) * a.dq_ld + h * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (bf16_t)((dt == 0 ? dq0[4 * g + e] : dq1[4 * g + e]) * a.scale);
        *reinterpret_cast<bf16x4*>(row + dt * 32 + 8 * g + 4 * hi) = w;
      }
  }
}

// stage of one query tile for the dkv kernel: Q [q][d], Q^T [d][q], dO [q][d], dO^T [d][q], lse2[64], D[64], dropout keys
struct QStage {
  u32x4 qreg[2], oreg[2];
  u32x2 qtreg[4], otreg[4];
  float lse_v, d_v;
  unsigned int dkey;   // tid < 128: dropout block key of (query row q0 + (tid & 63), key block kblk/64 + (tid >> 6))
  unsigned int qoff[2], ooff[2], qtoff[4], otoff[4];
  int qrow[2], qtrow[4];
  __device__ __forceinline__ void init(const BwdArgs& a, int b, int h, int tid) {
    const int chunk = tid & 7;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      qrow[p] = (tid >> 3) + 32 * p;
      qoff[p] = (unsigned int)(((long long)b * a.T + qrow[p]) * a.q_ld + h * HD + chunk * 8);
      ooff[p] = (unsigned int)(((long long)b * a.T + qrow[p]) * a.do_ld + h * HD + chunk * 8);
    }
    const int dg = tid & 15, kg = tid >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      qtrow[kk] = kg * 4 + kk;
      qtoff[kk] = (unsigned int)(((long long)b * a.T + qtrow[kk]) * a.q_ld + h * HD + dg * 4);
      otoff[kk] = (unsigned int)(((long long)b * a.T + qtrow[kk]) * a.do_ld + h * HD + dg * 4);
    }
  }
  __device__ __forceinline__ void load(const BwdArgs& a, int bh, int q0, int kblk, int tid) {
    const unsigned int qadd = (unsigned int)q0 * (unsigned int)a.q_ld, oadd = (unsigned int)q0 * (unsigned int)a.do_ld;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      u32x4 zq = {0u, 0u, 0u, 0u}, zo = {0u, 0u, 0u, 0u};
      if (q0 + qrow[p] < a.T) {
        zq = *reinterpret_cast<const u32x4*>(a.q + (qoff[p] + qadd));
        zo = *reinterpret_cast<const u32x4*>(a.dout + (ooff[p] + oadd));
      }
      qreg[p] = zq; oreg[p] = zo;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      u32x2 zq = {0u, 0u}, zo = {0u, 0u};
      if (q0 + qtrow[kk] < a.T) {
        zq = *reinterpret_cast<const u32x2*>(a.q + (qtoff[kk] + qadd));
        zo = *reinterpret_cast<const u32x2*>(a.dout + (otoff[kk] + oadd));
      }
      qtreg[kk] = zq; otreg[kk] = zo;
    }
    if (tid < 64) {
      const int qq = q0 + tid;
      lse_v = qq < a.T ? a.lse[(long long)bh * a.T + qq] * LOG2E : INFINITY;   // log2 domain; +inf => P = 0 for rows past T
      d_v = qq < a.T ? a.dvec[(long long)bh * a.T + qq] : 0.f;
    }
    if (tid < 128 && a.dropout_p > 0.f) {
      const int qq = q0 + (tid & 63);
      const unsigned long long row = (unsigned long long)bh * a.T + (unsigned long long)(qq < a.T ? qq : a.T - 1);
      dkey = drop_block_key(a.seed, row * (unsigned long long)(drop_row_stride(a.lds) >> 6) + (unsigned long long)((kblk >> 6) + (tid >> 6)));
    }
  }
  __device__ __forceinline__ void store(char* buf, int tid) const {
    char* qt = buf; char* qtt = buf + TILE_B; char* ot = buf + 2 * TILE_B; char* ott = buf + 3 * TILE_B;
    float* st = reinterpret_cast<float*>(buf + 4 * TILE_B);
    const int chunk = tid & 7;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      *reinterpret_cast<u32x4*>(qt + lds_off((tid >> 3) + 32 * p, chunk)) = qreg[p];
      *reinterpret_cast<u32x4*>(ot + lds_off((tid >> 3) + 32 * p, chunk)) = oreg[p];
    }
    const int dg = tid & 15, kg = tid >> 4;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      u32x2 o1, o2;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        unsigned int x = qtreg[2 * w][rr >> 1], y = qtreg[2 * w + 1][rr >> 1];
        o1[w] = (rr & 1) ? ((x >> 16) | (y & 0xffff0000u)) : ((x & 0xffffu) | (y << 16));
        x = otreg[2 * w][rr >> 1]; y = otreg[2 * w + 1][rr >> 1];
        o2[w] = (rr & 1) ? ((x >> 16) | (y & 0xffff0000u)) : ((x & 0xffffu) | (y << 16));
      }
      *reinterpret_cast<u32x2*>(qtt + lds_off(dg * 4 + rr, kg >> 1) + (kg & 1) * 8) = o1;
      *reinterpret_cast<u32x2*>(ott + lds_off(dg * 4 + rr, kg >> 1) + (kg & 1) * 8) = o2;
    }
    if (tid < 64) { st[tid] = lse_v; st[64 + tid] = d_v; }
    if (tid < 128) reinterpret_cast<unsigned int*>(st)[128 + tid] = dkey;
  }
};

constexpr int QBUF = 4 * TILE_B + 1024;  // bytes per query-tile buffer: 4 tiles | lse2[64] | D[64] | dropout keys[128]

// Raw bf16 relative-position bias of one 32-query sub-tile for the dkv kernel (lane = key ki, element r = query
// q0 + (r&3) + 8(r>>2)), fetched one sub-tile ahead of its use.  qpb = qp + bh*T*nb.  LIN: no clipping, rows < T.
template <bool LIN>
__device__ __forceinline__ void load_bias_dkv(unsigned short (&braw)[16], const bf16_t* qpb, int q0, int ki, int T, int nb, int maxrel) {
  const unsigned short* src = reinterpret_cast<const unsigned short*>(qpb);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = (r & 3) + 8 * (r >> 2);
    const int qq = q0 + c;
    if (LIN) {
      braw[r] = src[(unsigned int)(qq * (nb + 1) + (maxrel - ki))];
    } else {
      const int row = qq < T ? qq : T - 1;
      braw[r] = src[(unsigned int)(row * nb + clip_rel(qq - ki, maxrel) + maxrel)];
    }
  }
}

// One 32-query sub-tile of the dkv kernel: in s = raw scores S[q][key], dp = dO.V; out pd = dropped P, s = dS.
// Element r is query q0 + (r&3) + 8(r>>2) (q0 includes 4*hi); the lane owns key ki.  Unclipped relative-position bucket
// gradients dQP[q][q - key + maxrel] are stored here (consecutive lanes = consecutive keys = consecutive buckets: coalesced).
template <int BM, bool SLOW, bool DROP>
__device__ __forceinline__ void dkv_sub(f32x16& s, const f32x16& dp, f32x16& pd, const BwdArgs& a, const float* stv, const unsigned int* keyv,
                                        const unsigned short (&braw)[16], bf16_t* dqpb, int qbase /* tile-local index of q0 */, int q0,
                                        int ki, bool kvalid, bool kmasked, float sc2, unsigned int pcl, unsigned int kshift,
                                        unsigned int thresh, float inv_keep) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = (r & 3) + 8 * (r >> 2);
    const int qq = q0 + c;
    const int d = qq - ki;
    const float bv = BM == BM_NONE ? 0.f : __uint_as_float((unsigned int)braw[r] << 16);
    const float x = fmaf(s[r], sc2, bv);
    float p = fast_exp2(x - stv[qbase + c]);
    if (SLOW) { if (kmasked || (a.causal && ki > qq + (a.S - a.T))) p = 0.f; }
    float dpv = dp[r], pv = p;
    if (DROP) {
      const unsigned int bits = drop_pair_bits_pc(keyv[qbase + c], pcl);
      const bool keep = ((bits >> kshift) & 0xffffu) >= thresh;
      pv = keep ? p * inv_keep : 0.f;
      dpv = keep ? dpv * inv_keep : 0.f;
    }
    pd[r] = pv;
    const float ds = p * (dpv - stv[64 + qbase + c]);
    s[r] = ds;
    if (BM == BM_LIN || BM == BM_GEN) {
      bool st_ok = kvalid;
      if (SLOW) st_ok = st_ok && qq < a.T;
      if (BM == BM_GEN) st_ok = st_ok && d > -a.maxrel && d < a.maxrel - 1;
#if FLASH_ABL != 1
      if (st_ok) dqpb[(unsigned int)(qq * (a.nb + 1) + (a.maxrel - ki))] = (bf16_t)ds;
#endif
    }
  }
}

template <bool BIAS>
__global__ __launch_bounds__(256, BIAS ? 1 : 2) void flash_bwd_dkv_kernel(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kl = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
  const int kblk = blockIdx.x * 128;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int kw0 = kblk + wave_u * 32;
  const int ki = kw0 + kl;                        // this lane's key
  const int kc = ki < a.S ? ki : a.S - 1;
  const bool kvalid = ki < a.S;
  const bool kmasked = !kvalid || (a.kpm && a.kpm[(long long)b * a.S + kc]);
  const bool any_kmasked = __ballot(kmasked) != 0ull;

  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8*>(a.k + ((long long)b * a.S + kc) * a.k_ld + h * HD + ks * 16 + hi * 8);
    vf[ks] = *reinterpret_cast<const bf16x8*>(a.v + ((long long)b * a.S + kc) * a.v_ld + h * HD + ks * 16 + hi * 8);
  }
  // first query tile that can attend to this key block (causal: q >= key - (S - T))
  int qt0 = 0;
  if (a.causal) { const int qmin = kblk - (a.S - a.T); qt0 = qmin > 0 ? qmin / 64 : 0; }
  const int nqt = (a.T + 63) / 64;

  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
  const bool drop = a.dropout_p > 0.f;
  const unsigned int thresh = drop ? dropout_thresh(a.dropout_p) : 0u;
  const float inv_keep = drop ? 1.f / (1.f - a.dropout_p) : 1.f;
  const float sc2 = a.scale * LOG2E;
  const unsigned int pcl = ((unsigned int)(ki & 63) >> 1) * PAIR_MUL;   // this key's pair inside its 64-key block
  const unsigned int kshift = (ki & 1) ? 16u : 0u;
  const long long bhT = (long long)bh * a.T;

  const bf16_t* qpb = BIAS ? a.qp + bhT * a.nb : nullptr;
  bf16_t* dqpb = BIAS ? a.dqp + bhT * a.nb : nullptr;
  // relative-position mode of sub-tile n (32 queries from 32n) against this wave's 32 keys
  auto sub_mode = [&](int n) { return BIAS ? bias_mode(32 * n - (kw0 + 31), 32 * n + 31 - kw0, a.maxrel) : BM_NONE; };
  auto sub_lin = [&](int n) { return sub_mode(n) == BM_LIN && 32 * n + 32 <= a.T; };
  unsigned short bcur[16], bnext[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { bcur[r] = 0; bnext[r] = 0; }

  QStage st;
  st.init(a, b, h, tid);
  if (qt0 < nqt) {
    st.load(a, bh, qt0 * 64, kblk, tid);
    if (BIAS) {
      if (sub_lin(2 * qt0)) load_bias_dkv<true>(bcur, qpb, qt0 * 64 + 4 * hi, ki, a.T, a.nb, a.maxrel);
      else load_bias_dkv<false>(bcur, qpb, qt0 * 64 + 4 * hi, ki, a.T, a.nb, a.maxrel);
    }
    st.store(smem + (qt0 & 1) * QBUF, tid);
  }
  __syncthreads();
#ifdef FLASH_TIMING
  unsigned int tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
  for (int qt = qt0; qt < nqt; ++qt) {
    const char* buf = smem + (qt & 1) * QBUF;
    const char* qtl = buf; const char* qtt = buf + TILE_B; const char* otl = buf + 2 * TILE_B; const char* ott = buf + 3 * TILE_B;
    const float* stv = reinterpret_cast<const float*>(buf + 4 * TILE_B);
    const unsigned int* keyv = reinterpret_cast<const unsigned int*>(buf + 4 * TILE_B) + 128 + (wave_u >> 1) * 64;
    if (qt + 1 < nqt) st.load(a, bh, (qt + 1) * 64, kblk, tid);
    // whole-tile conditions (wave-uniform): key padding, causal boundary, rows past T
    const bool slow = any_kmasked || (a.causal && kw0 + 31 > qt * 64 + (a.S - a.T)) || qt * 64 + 64 > a.T;
    TPROBE(0);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (BIAS) {   // fetch the next sub-tile's bias now; it is consumed one sub-tile later
        const int nn = 2 * qt + sub + 1;
        if (FLASH_ABL != 2 && nn < 2 * nqt) {
          if (sub_lin(nn)) load_bias_dkv<true>(bnext, qpb, nn * 32 + 4 * hi, ki, a.T, a.nb, a.maxrel);
          else load_bias_dkv<false>(bnext, qpb, nn * 32 + 4 * hi, ki, a.T, a.nb, a.maxrel);
        }
      }
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 qa = *reinterpret_cast<const bf16x8*>(qtl + lds_off(sub * 32 + kl, 2 * ks + hi));
        const bf16x8 oa = *reinterpret_cast<const bf16x8*>(otl + lds_off(sub * 32 + kl, 2 * ks + hi));
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);    // S[q][key]
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vf[ks], dp, 0, 0, 0);  // dP[q][key]
      }
      TPROBE(1);
      f32x16 pd;
      const int qs0 = qt * 64 + sub * 32;           // first query of the sub-tile
      const int qbase = sub * 32 + 4 * hi, q0 = qs0 + 4 * hi;
      const int bm = sub_mode(2 * qt + sub);
#define DKV_SUB(BM_, SLOW_, DROP_) \
  dkv_sub<BM_, SLOW_, DROP_>(s, dp, pd, a, stv, keyv, bcur, dqpb, qbase, q0, ki, kvalid, kmasked, sc2, pcl, kshift, thresh, inv_keep)
      if (drop) {
        if (slow) { if (bm == BM_NONE) DKV_SUB(BM_NONE, true, true); else DKV_SUB(BM_GEN, true, true); }
        else if (bm == BM_NONE) DKV_SUB(BM_NONE, false, true);
        else if (bm == BM_LIN) DKV_SUB(BM_LIN, false, true);
        else if (bm == BM_CONST) DKV_SUB(BM_CONST, false, true);
        else DKV_SUB(BM_GEN, false, true);
      } else {
        if (slow) { if (bm == BM_NONE) DKV_SUB(BM_NONE, true, false); else DKV_SUB(BM_GEN, true, false); }
        else if (bm == BM_NONE) DKV_SUB(BM_NONE, false, false);
        else if (bm == BM_LIN) DKV_SUB(BM_LIN, false, false);
        else if (bm == BM_CONST) DKV_SUB(BM_CONST, false, false);
        else DKV_SUB(BM_GEN, false, false);
      }
#undef DKV_SUB
      TPROBE(2);
#pragma unroll
      for (int r = 0; r < 16; ++r) bcur[r] = bnext[r];
      // dV^T[d][key] += dO^T[d][q] . Pd[q][key] ;  dK^T[d][key] += Q^T[d][q] . dS[q][key]
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int sstep = 2 * sub + u;
        const bf16x8 pf = pack8(pd, 8 * u);
        const bf16x8 df = pack8(s, 8 * u);
        dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_t8(ott, kl, sstep, hi), pf, dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_t8(ott, 32 + kl, sstep, hi), pf, dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_t8(qtt, kl, sstep, hi), df, dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_t8(qtt, 32 + kl, sstep, hi), df, dk1, 0, 0, 0);
      }
      TPROBE(3);
    }
    if (qt + 1 < nqt) st.store(smem + ((qt + 1) & 1) * QBUF, tid);
    TPROBE(4);
    __syncthreads();
    TPROBE(5);
  }
#ifdef FLASH_TIMING
  if (lane < 8) a.dvec[(long long)bh * a.T + kblk + wave_u * 32 + 8 + lane] = (float)tacc[lane];
#endif
  if (kvalid) {
    bf16_t* krow = a.dk + ((long long)b * a.S + ki) * a.dk_ld + h * HD;
    bf16_t* vrow = a.dv + ((long long)b * a.S + ki) * a.dv_ld + h * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 wk, wv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          wk[e] = (bf16_t)((dt == 0 ? dk0[4 * g + e] : dk1[4 * g + e]) * a.scale);
          wv[e] = (bf16_t)(dt == 0 ? dv0[4 * g + e] : dv1[4 * g + e]);
        }
        *reinterpret_cast<bf16x4*>(krow + dt * 32 + 8 * g + 4 * hi) = wk;
        *reinterpret_cast<bf16x4*>(vrow + dt * 32 + 8 * g + 4 * hi) = wv;
      }
  }
}

}  // namespace

extern "C" int st5_flash1_attn_fwd_qp(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld,
                                     void* o, int64_t o_ld, float* lse, const void* pe, const uint8_t* kpm, int32_t B, int32_t H,
                                     int32_t T, int32_t S, int32_t head_dim, int32_t nb, int32_t maxrel, int32_t causal,
                                     int32_t lds, float scale, float dropout_p, uint64_t seed, void* qp_out, int dtype,
                                     void* stream) {
  if (!q || !k || !v || !o || B <= 0 || H <= 0 || T <= 0 || S <= 0) return ST5_ERR_ARG;
  if (dtype != ST5_BF16 || head_dim != HD) return ST5_ERR_ARG;  // fp32 / other head sizes use the unfused path
  if (q_ld % 8 || k_ld % 8 || v_ld % 4 || o_ld % 4) return ST5_ERR_ALIGN;
  if ((long long)B * S * k_ld >= (1ll << 31) || (long long)B * S * v_ld >= (1ll << 31)) return ST5_ERR_ARG;  // 32-bit element offsets
  if (pe && (nb != 2 * maxrel || nb % 8 || nb > 1024)) return ST5_ERR_ARG;
  Args a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o; a.lse = lse;
  a.pe = (const bf16_t*)pe; a.kpm = kpm; a.qp_out = pe ? (bf16_t*)qp_out : nullptr;
  a.q_ld = q_ld; a.k_ld = k_ld; a.v_ld = v_ld; a.o_ld = o_ld;
  a.B = B; a.H = H; a.T = T; a.S = S; a.nb = pe ? nb : 0; a.maxrel = maxrel; a.causal = causal; a.lds = lds;
  a.scale = scale; a.dropout_p = dropout_p; a.seed = seed;
  dim3 grid((T + 127) / 128, B * H), block(256);
  size_t shm = 4 * TILE_B + (pe ? (size_t)4 * 32 * (nb + 4) * 2 : 0);
  hipStream_t s = (hipStream_t)stream;
  if (pe) {
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)flash_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return ST5_ERR_LAUNCH;
      attr_set = true;
    }
    hipLaunchKernelGGL(flash_fwd_kernel<true>, grid, block, shm, s, a);
  } else {
    hipLaunchKernelGGL(flash_fwd_kernel<false>, grid, block, shm, s, a);
  }
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_flash1_attn_fwd(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld,
                                  void* o, int64_t o_ld, float* lse, const void* pe, const uint8_t* kpm, int32_t B, int32_t H,
                                  int32_t T, int32_t S, int32_t head_dim, int32_t nb, int32_t maxrel, int32_t causal,
                                  int32_t lds, float scale, float dropout_p, uint64_t seed, int dtype, void* stream) {
  return st5_flash1_attn_fwd_qp(q, q_ld, k, k_ld, v, v_ld, o, o_ld, lse, pe, kpm, B, H, T, S, head_dim, nb, maxrel, causal, lds, scale,
                               dropout_p, seed, nullptr, dtype, stream);
}

extern "C" int st5_flash1_attn_bwd_2s(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld,
                                     const void* o, int64_t o_ld, const void* dout, int64_t do_ld, void* dq, int64_t dq_ld, void* dk,
                                     int64_t dk_ld, void* dv, int64_t dv_ld, const float* lse, float* dvec, const void* pe,
                                     const void* qp, void* dqp, const uint8_t* kpm, int32_t B, int32_t H, int32_t T, int32_t S,
                                     int32_t head_dim, int32_t nb, int32_t maxrel, int32_t causal, int32_t lds, float scale,
                                     float dropout_p, uint64_t seed, int dtype, void* stream, void* stream2) {
  if (!q || !k || !v || !o || !dout || !dq || !dk || !dv || !lse || !dvec || B <= 0 || H <= 0 || T <= 0 || S <= 0) return ST5_ERR_ARG;
  if (dtype != ST5_BF16 || head_dim != HD) return ST5_ERR_ARG;
  if (q_ld % 8 || k_ld % 8 || v_ld % 8 || o_ld % 8 || do_ld % 8 || dq_ld % 4 || dk_ld % 4 || dv_ld % 4) return ST5_ERR_ALIGN;
  if ((long long)B * S * k_ld >= (1ll << 31) || (long long)B * S * v_ld >= (1ll << 31) || (long long)B * T * q_ld >= (1ll << 31) ||
      (long long)B * T * do_ld >= (1ll << 31) || (long long)B * H * T * (pe ? nb : 1) >= (1ll << 31))
    return ST5_ERR_ARG;  // 32-bit element offsets
  if (pe && (!qp || !dqp || nb != 2 * maxrel || nb % 8 || nb > 1024)) return ST5_ERR_ARG;
  BwdArgs a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o; a.dout = (const bf16_t*)dout;
  a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.lse = lse; a.dvec = dvec;
  a.pe = (const bf16_t*)pe; a.qp = (const bf16_t*)qp; a.dqp = (bf16_t*)dqp; a.kpm = kpm;
  a.q_ld = q_ld; a.k_ld = k_ld; a.v_ld = v_ld; a.o_ld = o_ld; a.do_ld = do_ld; a.dq_ld = dq_ld; a.dk_ld = dk_ld; a.dv_ld = dv_ld;
  a.B = B; a.H = H; a.T = T; a.S = S; a.nb = pe ? nb : 0; a.maxrel = maxrel; a.causal = causal; a.lds = lds;
  a.scale = scale; a.dropout_p = dropout_p; a.seed = seed;
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)flash_bwd_dq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)flash_bwd_dkv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)flash_bwd_dkv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return ST5_ERR_LAUNCH;
    attr_set = true;
  }
  if (pe && hipMemsetAsync(dqp, 0, (size_t)B * H * T * nb * 2, s) != hipSuccess) return ST5_ERR_LAUNCH;
  const size_t shm_dq = 6 * TILE_B + (pe ? (size_t)4 * 32 * (nb + 4) * 2 : 0);
  const size_t shm_dkv = 2 * QBUF;
  // Two-stream form: the dq and dkv kernels are independent given D and each fills the chip only ~1.5x over (384 blocks
  // of 256 threads at B*H = 96, one block per CU with the bias tables); side by side they share the CUs.
  hipStream_t s2 = (hipStream_t)stream2;
  a.write_dvec = s2 ? 0 : 1;
  if (s2) {
    const long long rows = (long long)B * H * T;
    hipLaunchKernelGGL(flash_dvec_kernel, dim3((unsigned)((rows * 2 + 255) / 256)), dim3(256), 0, s, a.o, a.dout, dvec, a.o_ld, a.do_ld,
                       H, T, rows);
    if (st5_stream_fork(s, s2) != ST5_OK) return ST5_ERR_LAUNCH;
  } else {
    s2 = s;
  }
  if (pe) {
    hipLaunchKernelGGL(flash_bwd_dq_kernel<true>, dim3((T + 127) / 128, B * H), dim3(256), shm_dq, s, a);
    hipLaunchKernelGGL(flash_bwd_dkv_kernel<true>, dim3((S + 127) / 128, B * H), dim3(256), shm_dkv, s2, a);
  } else {
    hipLaunchKernelGGL(flash_bwd_dq_kernel<false>, dim3((T + 127) / 128, B * H), dim3(256), shm_dq, s, a);
    hipLaunchKernelGGL(flash_bwd_dkv_kernel<false>, dim3((S + 127) / 128, B * H), dim3(256), shm_dkv, s2, a);
  }
  HIP_CHECK_LAUNCH();
  if (s2 != s && st5_stream_fork(s2, s) != ST5_OK) return ST5_ERR_LAUNCH;
  return ST5_OK;
}

extern "C" int st5_flash1_attn_bwd(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld,
                                  const void* o, int64_t o_ld, const void* dout, int64_t do_ld, void* dq, int64_t dq_ld, void* dk,
                                  int64_t dk_ld, void* dv, int64_t dv_ld, const float* lse, float* dvec, const void* pe,
                                  const void* qp, void* dqp, const uint8_t* kpm, int32_t B, int32_t H, int32_t T, int32_t S,
                                  int32_t head_dim, int32_t nb, int32_t maxrel, int32_t causal, int32_t lds, float scale,
                                  float dropout_p, uint64_t seed, int dtype, void* stream) {
  return st5_flash1_attn_bwd_2s(q, q_ld, k, k_ld, v, v_ld, o, o_ld, dout, do_ld, dq, dq_ld, dk, dk_ld, dv, dv_ld, lse, dvec, pe, qp, dqp,
                               kpm, B, H, T, S, head_dim, nb, maxrel, causal, lds, scale, dropout_p, seed, dtype, stream, nullptr);
}
