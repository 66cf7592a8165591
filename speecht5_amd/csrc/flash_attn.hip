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

constexpr int HD = 64;
constexpr int KT = 64;  // keys per tile
constexpr int TILE_B = KT * HD * 2;  // 8 KB

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

struct Args {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o; float* lse;
  const bf16_t* pe; const uint8_t* kpm;
  long long q_ld, k_ld, v_ld, o_ld;
  int B, H, T, S, nb, maxrel, causal, lds;
  float scale, dropout_p;
  unsigned long long seed;
};

// K tile: 2 x 16 B per thread; V^T tile: 4 keys x 4 d per thread (4 x 8 B loads, transposed in registers)
struct KVStage {
  u32x4 kreg[2];
  u32x2 vreg[4];
  __device__ __forceinline__ void load(const Args& a, int b, int h, int j0, int tid) {
    const int chunk = tid & 7;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int key = j0 + (tid >> 3) + 32 * p;
      key = key < a.S ? key : a.S - 1;
      kreg[p] = *reinterpret_cast<const u32x4*>(a.k + ((long long)b * a.S + key) * a.k_ld + h * HD + chunk * 8);
    }
    const int dg = tid & 15, kg = tid >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      int key = j0 + kg * 4 + kk;
      key = key < a.S ? key : a.S - 1;
      vreg[kk] = *reinterpret_cast<const u32x2*>(a.v + ((long long)b * a.S + key) * a.v_ld + h * HD + dg * 4);
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

template <bool BIAS>
__global__ __launch_bounds__(256) void flash_fwd_kernel(const Args a) {
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
  st.load(a, b, h, 0, tid);

  if (BIAS) {  // QP^T[bucket][q] = scale * pe . q^T  -> per-wave LDS table (bf16)
    const int nbt = (a.nb + 31) / 32;
    for (int bt = 0; bt < nbt; ++bt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      int brow = bt * 32 + ql;
      brow = brow < a.nb ? brow : a.nb - 1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 pf = *reinterpret_cast<const bf16x8*>(a.pe + (long long)brow * HD + ks * 16 + hi * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, qf[ks], acc, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int b0 = bt * 32 + 8 * g + 4 * hi;
        if (b0 < a.nb) {
          bf16x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (bf16_t)(acc[4 * g + e] * a.scale);
          *reinterpret_cast<bf16x4*>(qp + ql * qp_ld + b0) = w;
        }
      }
    }
  }

  st.store(kbuf, vbuf, tid);
  __syncthreads();

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  const unsigned int thresh = a.dropout_p > 0.f ? (unsigned int)((double)a.dropout_p * 4294967296.0) : 0u;
  const float inv_keep = a.dropout_p > 0.f ? 1.f / (1.f - a.dropout_p) : 1.f;
  const int jmax = a.causal ? qi + (a.S - a.T) : a.S - 1;
  const uint8_t* mrow = a.kpm ? a.kpm + (long long)b * a.S : nullptr;
  const unsigned long long ctr_row = ((unsigned long long)bh * a.T + (unsigned long long)qc) * (unsigned long long)a.lds;

  for (int jt = 0; jt < ntiles; ++jt) {
    const char* kt = kbuf + (jt & 1) * TILE_B;
    const char* vt = vbuf + (jt & 1) * TILE_B;
    if (jt + 1 < ntiles) st.load(a, b, h, (jt + 1) * KT, tid);
    const int j0 = jt * KT;

    // S^T tiles (two 32-key sub-tiles)
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kt + lds_off(ql, 2 * ks + hi));
      const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(kt + lds_off(32 + ql, 2 * ks + hi));
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[ks], s1, 0, 0, 0);
    }
    // scale, bias, masks, tile max
    float tmax = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = j0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float x = (t == 0 ? s0[r] : s1[r]) * a.scale;
        if (BIAS) {
          int dlt = qc - key;
          dlt = dlt < -a.maxrel ? -a.maxrel : (dlt > a.maxrel - 1 ? a.maxrel - 1 : dlt);
          x += (float)qp[ql * qp_ld + dlt + a.maxrel];
        }
        if (key >= a.S || key > jmax || (mrow && mrow[key < a.S ? key : 0])) x = -INFINITY;
        if (t == 0) s0[r] = x; else s1[r] = x;
        tmax = fmaxf(tmax, x);
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = m_run == -INFINITY ? 0.f : __expf(m_run - m_use);
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = t == 0 ? s0[r] : s1[r];
        float p = __expf(x - m_use);  // exp(-inf) = 0 for masked keys
        psum += p;
        if (a.dropout_p > 0.f) {
          const int key = j0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          p *= dropout_scale(a.seed, ctr_row + (unsigned long long)key, thresh, inv_keep);
        }
        if (t == 0) s0[r] = p; else s1[r] = p;
      }
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    // O^T += V^T . P^T   (4 k-steps of 16 keys, two 32-row d tiles)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 pf = pack8(s < 2 ? s0 : s1, 8 * (s & 1));
      bf16x8 v0, v1;
      {
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vt + lds_off(ql, 2 * s) + 8 * hi);
        const bf16x4 hi4 = *reinterpret_cast<const bf16x4*>(vt + lds_off(ql, 2 * s + 1) + 8 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v0[e] = lo[e]; v0[4 + e] = hi4[e]; }
      }
      {
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vt + lds_off(32 + ql, 2 * s) + 8 * hi);
        const bf16x4 hi4 = *reinterpret_cast<const bf16x4*>(vt + lds_off(32 + ql, 2 * s + 1) + 8 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v1[e] = lo[e]; v1[4 + e] = hi4[e]; }
      }
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf, o1, 0, 0, 0);
    }
    if (jt + 1 < ntiles) st.store(kbuf + ((jt + 1) & 1) * TILE_B, vbuf + ((jt + 1) & 1) * TILE_B, tid);
    __syncthreads();
  }

  // finalize: combine the two half-wave partial sums, normalise, store O and the log-sum-exp
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
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
    if (hi == 0 && a.lse) a.lse[(long long)bh * a.T + qi] = l_tot > 0.f ? m_run + __logf(l_tot) : INFINITY;
  }
}


// =====================================================================================================
// Backward.  D[bh,q] = dO[q].O[q] (prep), then two kernels that recompute P from (Q, K, bias, LSE):
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
  float scale, dropout_p;
  unsigned long long seed;
};

// D[bh, t] = sum_d dO[b,t,h,d] * O[b,t,h,d]: one wave per (b,t) row, 8 lanes per head chunk
__global__ __launch_bounds__(256) void flash_bwd_prep_kernel(const BwdArgs a) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)a.B * a.T) return;
  const int b = (int)(row / a.T), t = (int)(row % a.T);
  const int d = a.H * HD;
  for (int c = lane * 8; c < d; c += 512) {
    float x[8], y[8];
    load8f<bf16_t>(a.dout + row * a.do_ld + c, x);
    load8f<bf16_t>(a.o + row * a.o_ld + c, y);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(x[e], y[e], s);
    // 8 lanes cover one head (64 dims)
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if ((lane & 7) == 0) a.dvec[((long long)b * a.H + c / HD) * a.T + t] = s;
  }
}

// stage of one key tile for the dq kernel: K [key][d], K^T [d][key], V [key][d]
struct KVStageBwd {
  u32x4 kreg[2], vreg[2];
  u32x2 ktreg[4];
  __device__ __forceinline__ void load(const BwdArgs& a, int b, int h, int j0, int tid) {
    const int chunk = tid & 7;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int key = j0 + (tid >> 3) + 32 * p;
      key = key < a.S ? key : a.S - 1;
      kreg[p] = *reinterpret_cast<const u32x4*>(a.k + ((long long)b * a.S + key) * a.k_ld + h * HD + chunk * 8);
      vreg[p] = *reinterpret_cast<const u32x4*>(a.v + ((long long)b * a.S + key) * a.v_ld + h * HD + chunk * 8);
    }
    const int dg = tid & 15, kg = tid >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      int key = j0 + kg * 4 + kk;
      key = key < a.S ? key : a.S - 1;
      ktreg[kk] = *reinterpret_cast<const u32x2*>(a.k + ((long long)b * a.S + key) * a.k_ld + h * HD + dg * 4);
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
  const float dsum = a.dvec[(long long)bh * a.T + qc];

  int nkeys = a.S;
  if (a.causal) {
    const int qmax = (qblk + 127 < a.T ? qblk + 127 : a.T - 1) + (a.S - a.T);
    nkeys = qmax + 1 < a.S ? qmax + 1 : a.S;
  }
  const int ntiles = (nkeys + KT - 1) / KT;
  KVStageBwd st;
  st.load(a, b, h, 0, tid);
  if (BIAS) {
    const int nbt = (a.nb + 31) / 32;
    for (int bt = 0; bt < nbt; ++bt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      int brow = bt * 32 + ql;
      brow = brow < a.nb ? brow : a.nb - 1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 pf = *reinterpret_cast<const bf16x8*>(a.pe + (long long)brow * HD + ks * 16 + hi * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, qf[ks], acc, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int b0 = bt * 32 + 8 * g + 4 * hi;
        if (b0 < a.nb) {
          bf16x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (bf16_t)(acc[4 * g + e] * a.scale);
          *reinterpret_cast<bf16x4*>(qp + ql * qp_ld + b0) = w;
        }
      }
    }
  }
  st.store(smem, smem + TILE_B, smem + 2 * TILE_B, tid);
  __syncthreads();

  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  float acc_lo = 0.f, acc_hi = 0.f;  // clipped relative-position buckets 0 and nb-1
  const unsigned int thresh = a.dropout_p > 0.f ? (unsigned int)((double)a.dropout_p * 4294967296.0) : 0u;
  const float inv_keep = a.dropout_p > 0.f ? 1.f / (1.f - a.dropout_p) : 1.f;
  const int jmax = a.causal ? qi + (a.S - a.T) : a.S - 1;
  const uint8_t* mrow = a.kpm ? a.kpm + (long long)b * a.S : nullptr;
  const unsigned long long ctr_row = ((unsigned long long)bh * a.T + (unsigned long long)qc) * (unsigned long long)a.lds;
  bf16_t* dqp_row = a.dqp ? a.dqp + ((long long)bh * a.T + qc) * a.nb : nullptr;

  for (int jt = 0; jt < ntiles; ++jt) {
    const char* buf = smem + (jt & 1) * 3 * TILE_B;
    const char* kt = buf; const char* ktt = buf + TILE_B; const char* vt = buf + 2 * TILE_B;
    if (jt + 1 < ntiles) st.load(a, b, h, (jt + 1) * KT, tid);
    const int j0 = jt * KT;
    f32x16 s0, s1, p0, p1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; p0[r] = 0.f; p1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kt + lds_off(ql, 2 * ks + hi));
      const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(kt + lds_off(32 + ql, 2 * ks + hi));
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[ks], s1, 0, 0, 0);
      const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(vt + lds_off(ql, 2 * ks + hi));
      const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(vt + lds_off(32 + ql, 2 * ks + hi));
      p0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, dof[ks], p0, 0, 0, 0);   // dP^T[key][q]
      p1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, dof[ks], p1, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = j0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float x = (t == 0 ? s0[r] : s1[r]) * a.scale;
        int dlt = qc - key;
        if (BIAS) {
          const int dc = dlt < -a.maxrel ? -a.maxrel : (dlt > a.maxrel - 1 ? a.maxrel - 1 : dlt);
          x += (float)qp[ql * qp_ld + dc + a.maxrel];
        }
        const bool masked = key >= a.S || key > jmax || (mrow && mrow[key < a.S ? key : 0]);
        const float p = masked ? 0.f : __expf(x - lse);
        float dp = t == 0 ? p0[r] : p1[r];
        if (a.dropout_p > 0.f) dp *= dropout_scale(a.seed, ctr_row + (unsigned long long)key, thresh, inv_keep);
        const float ds = p * (dp - dsum);
        if (t == 0) s0[r] = ds; else s1[r] = ds;
        if (BIAS && qvalid && key < a.S) {
          if (dlt <= -a.maxrel) acc_lo += ds;
          else if (dlt >= a.maxrel - 1) acc_hi += ds;
          else dqp_row[dlt + a.maxrel] = (bf16_t)ds;
        }
      }
    }
    // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 df = pack8(s < 2 ? s0 : s1, 8 * (s & 1));
      const bf16x8 kt0 = read_t8(ktt, ql, s, hi);
      const bf16x8 kt1 = read_t8(ktt, 32 + ql, s, hi);
      dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt0, df, dq0, 0, 0, 0);
      dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt1, df, dq1, 0, 0, 0);
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

// stage of one query tile for the dkv kernel: Q [q][d], Q^T [d][q], dO [q][d], dO^T [d][q], lse[64], D[64]
struct QStage {
  u32x4 qreg[2], oreg[2];
  u32x2 qtreg[4], otreg[4];
  float lse_v, d_v;
  __device__ __forceinline__ void load(const BwdArgs& a, int b, int h, int bh, int q0, int tid) {
    const int chunk = tid & 7;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int qq = q0 + (tid >> 3) + 32 * p;
      qq = qq < a.T ? qq : a.T - 1;
      qreg[p] = *reinterpret_cast<const u32x4*>(a.q + ((long long)b * a.T + qq) * a.q_ld + h * HD + chunk * 8);
      oreg[p] = *reinterpret_cast<const u32x4*>(a.dout + ((long long)b * a.T + qq) * a.do_ld + h * HD + chunk * 8);
    }
    const int dg = tid & 15, kg = tid >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      int qq = q0 + kg * 4 + kk;
      qq = qq < a.T ? qq : a.T - 1;
      qtreg[kk] = *reinterpret_cast<const u32x2*>(a.q + ((long long)b * a.T + qq) * a.q_ld + h * HD + dg * 4);
      otreg[kk] = *reinterpret_cast<const u32x2*>(a.dout + ((long long)b * a.T + qq) * a.do_ld + h * HD + dg * 4);
    }
    if (tid < 64) {
      const int qq = q0 + tid;
      lse_v = qq < a.T ? a.lse[(long long)bh * a.T + qq] : INFINITY;   // +inf => P = 0 for rows past T
      d_v = qq < a.T ? a.dvec[(long long)bh * a.T + qq] : 0.f;
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
  }
};

constexpr int QBUF = 4 * TILE_B + 512;  // bytes per query-tile buffer

template <bool BIAS>
__global__ __launch_bounds__(256) void flash_bwd_dkv_kernel(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kl = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
  const int kblk = blockIdx.x * 128;
  const int ki = kblk + wave * 32 + kl;          // this lane's key
  const int kc = ki < a.S ? ki : a.S - 1;
  const bool kvalid = ki < a.S;
  const bool kmasked = !kvalid || (a.kpm && a.kpm[(long long)b * a.S + kc]);

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
  const unsigned int thresh = a.dropout_p > 0.f ? (unsigned int)((double)a.dropout_p * 4294967296.0) : 0u;
  const float inv_keep = a.dropout_p > 0.f ? 1.f / (1.f - a.dropout_p) : 1.f;

  QStage st;
  if (qt0 < nqt) {
    st.load(a, b, h, bh, qt0 * 64, tid);
    st.store(smem + (qt0 & 1) * QBUF, tid);
  }
  __syncthreads();
  for (int qt = qt0; qt < nqt; ++qt) {
    const char* buf = smem + (qt & 1) * QBUF;
    const char* qtl = buf; const char* qtt = buf + TILE_B; const char* otl = buf + 2 * TILE_B; const char* ott = buf + 3 * TILE_B;
    const float* stv = reinterpret_cast<const float*>(buf + 4 * TILE_B);
    if (qt + 1 < nqt) st.load(a, b, h, bh, (qt + 1) * 64, tid);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
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
      f32x16 pd;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qloc = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int qq = qt * 64 + qloc;
        float x = s[r] * a.scale;
        if (BIAS) {
          int dlt = qq - kc;
          dlt = dlt < -a.maxrel ? -a.maxrel : (dlt > a.maxrel - 1 ? a.maxrel - 1 : dlt);
          const int qqc = qq < a.T ? qq : a.T - 1;
          x += (float)a.qp[((long long)bh * a.T + qqc) * a.nb + dlt + a.maxrel];
        }
        const bool masked = kmasked || (a.causal && kc > qq + (a.S - a.T));
        const float p = masked ? 0.f : __expf(x - stv[qloc]);
        float dscale = 1.f;
        if (a.dropout_p > 0.f)
          dscale = dropout_scale(a.seed, ((unsigned long long)bh * a.T + (unsigned long long)(qq < a.T ? qq : a.T - 1)) * (unsigned long long)a.lds + (unsigned long long)kc, thresh, inv_keep);
        pd[r] = p * dscale;
        s[r] = p * (dp[r] * dscale - stv[64 + qloc]);  // dS
      }
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
    }
    if (qt + 1 < nqt) st.store(smem + ((qt + 1) & 1) * QBUF, tid);
    __syncthreads();
  }
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

extern "C" int st5_flash_attn_fwd(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld,
                                  void* o, int64_t o_ld, float* lse, const void* pe, const uint8_t* kpm, int32_t B, int32_t H,
                                  int32_t T, int32_t S, int32_t head_dim, int32_t nb, int32_t maxrel, int32_t causal,
                                  int32_t lds, float scale, float dropout_p, uint64_t seed, int dtype, void* stream) {
  if (!q || !k || !v || !o || B <= 0 || H <= 0 || T <= 0 || S <= 0) return ST5_ERR_ARG;
  if (dtype != ST5_BF16 || head_dim != HD) return ST5_ERR_ARG;  // fp32 / other head sizes use the unfused path
  if (q_ld % 8 || k_ld % 8 || v_ld % 4 || o_ld % 4) return ST5_ERR_ALIGN;
  if (pe && (nb != 2 * maxrel || nb % 8 || nb > 1024)) return ST5_ERR_ARG;
  Args a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o; a.lse = lse;
  a.pe = (const bf16_t*)pe; a.kpm = kpm;
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

extern "C" int st5_flash_attn_bwd(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld,
                                  const void* o, int64_t o_ld, const void* dout, int64_t do_ld, void* dq, int64_t dq_ld, void* dk,
                                  int64_t dk_ld, void* dv, int64_t dv_ld, const float* lse, float* dvec, const void* pe,
                                  const void* qp, void* dqp, const uint8_t* kpm, int32_t B, int32_t H, int32_t T, int32_t S,
                                  int32_t head_dim, int32_t nb, int32_t maxrel, int32_t causal, int32_t lds, float scale,
                                  float dropout_p, uint64_t seed, int dtype, void* stream) {
  if (!q || !k || !v || !o || !dout || !dq || !dk || !dv || !lse || !dvec || B <= 0 || H <= 0 || T <= 0 || S <= 0) return ST5_ERR_ARG;
  if (dtype != ST5_BF16 || head_dim != HD) return ST5_ERR_ARG;
  if (q_ld % 8 || k_ld % 8 || v_ld % 8 || o_ld % 8 || do_ld % 8 || dq_ld % 4 || dk_ld % 4 || dv_ld % 4) return ST5_ERR_ALIGN;
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
  hipLaunchKernelGGL(flash_bwd_prep_kernel, dim3((unsigned)(((long long)B * T + 3) / 4)), dim3(256), 0, s, a);
  if (pe && hipMemsetAsync(dqp, 0, (size_t)B * H * T * nb * 2, s) != hipSuccess) return ST5_ERR_LAUNCH;
  const size_t shm_dq = 6 * TILE_B + (pe ? (size_t)4 * 32 * (nb + 4) * 2 : 0);
  const size_t shm_dkv = 2 * QBUF;
  if (pe) {
    hipLaunchKernelGGL(flash_bwd_dq_kernel<true>, dim3((T + 127) / 128, B * H), dim3(256), shm_dq, s, a);
    hipLaunchKernelGGL(flash_bwd_dkv_kernel<true>, dim3((S + 127) / 128, B * H), dim3(256), shm_dkv, s, a);
  } else {
    hipLaunchKernelGGL(flash_bwd_dq_kernel<false>, dim3((T + 127) / 128, B * H), dim3(256), shm_dq, s, a);
    hipLaunchKernelGGL(flash_bwd_dkv_kernel<false>, dim3((S + 127) / 128, B * H), dim3(256), shm_dkv, s, a);
  }
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
