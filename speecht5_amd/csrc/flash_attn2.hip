// Fused attention for head_dim 64 in bf16 on gfx950, second generation (the kernels st5_flash_attn_* run).
// Same math, same dropout counters and same external interface as flash_attn.hip (kept as the A/B reference,
// st5_flash_attn_set_impl(0)); what changed is everything that cost VALU issue slots, registers and LDS there
// (rocprofv3 --pmc on the first generation: 30-45 VALU instructions per score element, MFMA busy 7-8 %, one wave per SIMD
// because of 320-410 registers and an 83 KB per-block bias table, 32-50 % of the wave cycles waiting):
//   * tiles of the looped-over side ([64 rows][64 d] bf16) go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging
//     registers, no VALU), double buffered, ONE image per tensor; operands that the MFMA wants transposed (V^T, K^T, Q^T,
//     dO^T) are read with the gfx950 transpose read ds_read_b64_tr_b16 instead of being transposed in registers into a
//     second LDS image;
//   * the Shaw relative-position bias comes from a global table QP[bh][q][8 | nb | 8] = scale*log2e * q.pe^T (bf16, the two
//     8-element end chunks replicate the clipped end buckets) built once per layer by qp_table_kernel.  Per key tile a wave
//     DMAs the 104-bucket window its 32 queries need into a 7 KB LDS scratch (clipping = clamping the chunk index) and
//     reads it back skewed (ds_read_u16 with immediate offsets).  No 83 KB table: 60-66 KB of LDS per block;
//   * <= 256 registers (launch bounds 2 waves per SIMD): two blocks per CU, so one wave's softmax arithmetic overlaps the
//     other's MFMA / LDS / DMA waits.
#include "common.h"
#include "../../include/speecht5_hip.h"

// FA2_ABL (timing experiments only, results wrong by construction; tools/r4/fa_abl.sh): 1 = no bias-window DMA, 2 = also no LDS
// reads of the window, 3 = dkv writes no dQP, 4 = every bias tile treated as fully clipped (uniform bias)
#ifndef FA2_ABL
#define FA2_ABL 0
#endif
// Discriminator builds for the side-by-side replay difference (tools/r5/sbs_arms.sh; results unchanged, timing only):
//   FA2_DMA_NOP     every LDS-DMA keeps its address register alive and is followed by 16 idle cycles (does the engine read the
//                   address registers late when another kernel's DMA traffic backs the queue up?)
//   FA2_SUB1_SLEEP  the dkv kernel's one window read that follows its own vmcnt(0) WITHOUT a barrier in between sleeps ~256 cycles
//                   first (is an LDS-DMA counted as landed before its LDS write is visible to the issuing wave?)
#ifdef FA2_DMA_NOP
#define FA2_AFTER_DMA(p) asm volatile("s_nop 7\n\ts_nop 7" ::"v"(p) : "memory")
#else
#define FA2_AFTER_DMA(p)
#endif
namespace fa2 {

constexpr int HD = 64;
constexpr int KT = 64;               // rows of a looped-over tile
constexpr int TILE_B = KT * HD * 2;  // 8 KB
constexpr int QP_PAD = 8;            // replicated end chunk (elements) on each side of a QP row
constexpr int WIN = 104;             // window elements per query (13 chunks of 8)
constexpr int WCH = 13;
constexpr int SCR_B = 7168;          // per-wave scratch: 7 DMA instructions x 1 KB
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr unsigned int PAIR_MUL = 0x9E3779B1u;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Block -> (tile index along x, batch x head) with the XCDs in mind (round 6).  Workgroups go to XCDs round-robin in linear order (x
// fastest), so the 4 query (key) blocks of one head sat on 4 DIFFERENT XCDs and every private L2 fetched that head's K / V (Q / dO)
// tiles through the fabric on its own.  Here the blocks of a head share an XCD: of every 8 * gridDim.x consecutive blocks, XCD k gets
// the gridDim.x blocks of head 8 * group + k.  (BH not a multiple of 8: the plain mapping.)  Results do not depend on the mapping.
__device__ __forceinline__ void block_coords(int& xb, int& bh) {
  const int nx = (int)gridDim.x, ny = (int)gridDim.y;
  if ((ny & 7) == 0) {
    const int lin = (int)blockIdx.y * nx + (int)blockIdx.x;
    const int per = 8 * nx, grp = lin / per, r = lin - grp * per;
    bh = grp * 8 + (r & 7);
    xb = r >> 3;
  } else {
    xb = (int)blockIdx.x; bh = (int)blockIdx.y;
  }
}

// ---- LDS-DMA staging of one [64][64] bf16 tile: 8 wave-instructions of 8 rows, two per wave -------------------------
// Lane l of instruction i (rows (i*4+wave)*8 .. +7) lands at physical chunk l&7 of row r = (i*4+wave)*8 + (l>>3) and
// therefore fetches logical chunk (l&7) ^ ((r>>1)&7).  Rows past `nrows` are clamped to the last valid row (their scores
// are masked, so the values only have to be finite).
struct TileStager {
  int r[2], lc[2];
  __device__ __forceinline__ void init(int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      r[i] = (i * 4 + wave) * 8 + (lane >> 3);
      lc[i] = ((lane & 7) ^ ((r[i] >> 1) & 7)) * 8;
    }
  }
  // base: element pointer of (row 0 of the tensor for this batch, head column h*64); row0: first row of the tile
  __device__ __forceinline__ void issue(const bf16_t* base, long long ld, int row0, int nrows, char* tile, int wave) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int row = row0 + r[i];
      row = row < nrows ? row : nrows - 1;
      const bf16_t* src = base + (long long)row * ld + lc[i];
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(tile + (i * 4 + wave) * 1024), 16, 0, 0);
      FA2_AFTER_DMA(src);
    }
  }
};

// ---- relative-position window: 32 queries x 104 buckets of QP into the wave's scratch ---------------------------------
// chunk index ci = i*64 + lane -> query ci / 13, chunk ci % 13 of the window (scratch rows are 13 chunks = 208 B,
// lane-linear).  wbc = window base in chunks relative to the data part of the row (may be negative / past the end:
// clamping the chunk index to [0, nb/8 + 1] selects the replicated end chunks = the clipped buckets).
struct WinStager {
  unsigned int pk[7];   // (row offset in 8-element chunks) << 4 | window chunk of the lane in DMA instruction i
  __device__ __forceinline__ void init(int lane, int qw0, int T, int nbp) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      int ci = i * 64 + lane;
      ci = ci < 32 * WCH ? ci : 32 * WCH - 1;          // the 7th instruction's spare lanes re-fetch the last chunk
      int q = qw0 + ci / WCH;
      q = q < T ? q : T - 1;
      pk[i] = (((unsigned int)q * (unsigned int)(nbp >> 3)) << 4) | (unsigned int)(ci % WCH);
    }
  }
  __device__ __forceinline__ void issue(const bf16_t* qpb, int wbc, int nchunks_data, char* scratch) const {
#if FA2_ABL == 1 || FA2_ABL == 2
    return;
#endif
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      int c = wbc + (int)(pk[i] & 15u) + 1;             // +1: data chunk c sits at row chunk c+1 (chunk 0 = low end replica)
      c = c < 0 ? 0 : (c > nchunks_data + 1 ? nchunks_data + 1 : c);
      const bf16_t* src = qpb + (((pk[i] >> 4) + (unsigned int)c) << 3);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(scratch + i * 1024), 16, 0, 0);
      FA2_AFTER_DMA(src);
    }
  }
};

// transposed MFMA A-operand fragment out of a [row][d] tile: lane (m = 32*dt + (l&31), hi) gets rows r0..r0+3 and r1..r1+3 of
// column m.  `laneoff` = this lane's (row sub-offset, column) part, precomputed.
struct TrAddr {
  int sub;     // (l&15)>>2 : row inside a 4-row group supplied by this lane
  int colb;    // 16*((l>>4)&1) + 4*(l&3) : column (d) of the lane's 8-byte chunk inside a 32-column d tile
  __device__ __forceinline__ void init(int lane) { sub = (lane & 15) >> 2; colb = 16 * ((lane >> 4) & 1) + 4 * (lane & 3); }
  __device__ __forceinline__ int off(int rbase, int dt) const {
    const int row = rbase + sub, col = 32 * dt + colb;
    return lds_off(row, col >> 3) + (col & 7) * 2;
  }
};
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int off0, int off1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + off1));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// (pairs through v_cvt_pk_bf16_f32: written element by element hipcc converted every score on its own -- one convert + half a v_perm per
//  element instead of half a convert; same round-to-nearest-even bits)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16x8 pack8(const f32x16& s, int base) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2_t p = {s[base + 2 * j], s[base + 2 * j + 1]};
    const bf16x2_t q = __builtin_convertvector(p, bf16x2_t);
    r[2 * j] = q[0]; r[2 * j + 1] = q[1];
  }
  return r;
}

__device__ __forceinline__ unsigned int kpm_raw(const uint8_t* mrow, int key, int S) {
  return mrow ? (unsigned int)mrow[key < S ? key : S - 1] : 0u;
}

// bias of a tile for the lane's query: element i = 16t + r is key offset c + 4*hi, c = 32t + (r&3) + 8(r>>2).
// rd = scratch byte address of window position (q_l + 63 - 4hi + mis) - 59 ... see caller; value index decreases with c.
__device__ __forceinline__ void read_bias(unsigned int (&braw)[32], const char* rd) {
  const unsigned short* p = reinterpret_cast<const unsigned short*>(rd);
#if FA2_ABL == 2
  for (int i = 0; i < 32; ++i) { braw[i] = 0x3c00u + i; asm volatile("" : "+v"(braw[i])); }
  return;
#endif
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int c = (i >> 4) * 32 + (i & 3) + 8 * ((i & 15) >> 2);
    braw[i] = p[59 - c];
  }
}

// =====================================================================================================================
// QP table: qp[bh][q][QP_PAD + b] = bf16(sc2 * q . pe[b]), end chunks replicated.  grid (ceil(T/128), B*H), 256 threads.
// =====================================================================================================================
template <bool PE_LDS>
__global__ __launch_bounds__(256) void qp_table_kernel(const bf16_t* __restrict__ qg, long long q_ld, const bf16_t* __restrict__ pe,
                                                       bf16_t* __restrict__ qp, int H, int T, int nb, float sc2) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int qi = blockIdx.x * 128 + wave * 32 + ql;
  const int qc = qi < T ? qi : T - 1;
  const int nbp = nb + 2 * QP_PAD;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(qg + ((long long)b * T + qc) * q_ld + h * HD + ks * 16 + hi * 8);
  bf16_t* row = qp + ((long long)bh * T + qc) * nbp + QP_PAD;
  const int nbt = (nb + 31) / 32;
  // Round 6: the PE table (nb x 64 bf16: 40 KB at nb = 320) is staged in LDS once per block, swizzled like every [row][64] tile of this
  // file.  Before, every bucket tile began with four dependent global loads of its PE fragment (L2 hits, ~1 us with the MFMAs and stores
  // behind them) and a wave walked its ten tiles one after the other: the kernel was bound by that chain, not by its 66 MB of stores.
  // Same fragments, same MFMA order: the table is bit-identical.  (PE_LDS == false: tables too large for LDS keep the global loads.)
  extern __shared__ __attribute__((aligned(16))) char pe_lds[];
  if (PE_LDS) {
    for (int ch = tid; ch < nb * 8; ch += 256) {
      const int r = ch >> 3, c = ch & 7;
      *reinterpret_cast<u32x4*>(pe_lds + lds_off(r, c)) = *reinterpret_cast<const u32x4*>(pe + r * HD + c * 8);
    }
    __syncthreads();
  }
  // The PE rows go into the MFMA's A operand PERMUTED (row m of the tile = bucket 16 ((m >> 2) & 1) + (m & 3) + 4 (m >> 3) of it), so
  // that accumulator register r of lane (query, hi) is bucket 32 bt + 16 hi + r: 16 CONSECUTIVE buckets = two 16-byte stores per lane
  // and tile (round 5; the natural order gave every lane groups of 4 buckets = 8-byte pieces 672 bytes apart, 2.4 TB/s of writes).
  // Every table entry is the same dot product in the same k order as before: bit-identical.
  const int prow = 16 * ((ql >> 2) & 1) + (ql & 3) + 4 * (ql >> 3);
  for (int bt = 0; bt < nbt; ++bt) {
    int brow = bt * 32 + prow;
    brow = brow < nb ? brow : nb - 1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 pf = PE_LDS ? *reinterpret_cast<const bf16x8*>(pe_lds + lds_off(brow, 2 * ks + hi))
                               : *reinterpret_cast<const bf16x8*>(pe + brow * HD + ks * 16 + hi * 8);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, qf[ks], acc, 0, 0, 0);
    }
    if (qi < T) {
      const int b0 = bt * 32 + 16 * hi;          // first bucket of this lane's 16
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (b0 + 8 * u < nb) {                   // (nb % 8 == 0: an 8-bucket chunk is inside the table or outside it)
          bf16x8 w;
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = (bf16_t)(acc[8 * u + e] * sc2);
          *reinterpret_cast<bf16x8*>(row + b0 + 8 * u) = w;
          if (b0 + 8 * u == 0) {                 // bucket 0: low end replica
            const bf16x8 lo = {w[0], w[0], w[0], w[0], w[0], w[0], w[0], w[0]};
            *reinterpret_cast<bf16x8*>(row - 8) = lo;
          }
          if (b0 + 8 * u + 8 == nb) {            // bucket nb - 1: high end replica
            const bf16x8 hv = {w[7], w[7], w[7], w[7], w[7], w[7], w[7], w[7]};
            *reinterpret_cast<bf16x8*>(row + nb) = hv;
          }
        }
      }
    }
  }
}

struct Args {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o; float* lse;
  const bf16_t* qp;                          // [BH, T, nb + 16] or NULL
  const uint8_t* kpm;
  long long q_ld, k_ld, v_ld, o_ld;
  int B, H, T, S, nb, maxrel, causal, lds;
  float scale, dropout_p;
  unsigned long long seed;
};

// scores of one tile in the log2 domain (+ masks); returns the lane's tile max.  UNI: one bias value for the whole tile.
template <int BMODE /*0 none, 1 per element, 2 uniform*/, bool MASK>
__device__ __forceinline__ float tile_scores(f32x16& s0, f32x16& s1, float sc2, const unsigned int (&braw)[32], float buni,
                                             unsigned long long km, int jrel) {
  float tmax = -INFINITY;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = t * 32 + (r & 3) + 8 * (r >> 2);
      const float b = BMODE == 1 ? __uint_as_float(braw[16 * t + r] << 16) : (BMODE == 2 ? buni : 0.f);
      float x = fmaf(t == 0 ? s0[r] : s1[r], sc2, b);
      if (MASK) { if (((km >> c) & 1ull) || c > jrel) x = -INFINITY; }
      if (t == 0) s0[r] = x; else s1[r] = x;
      tmax = fmaxf(tmax, x);
    }
  }
  return tmax;
}

template <bool DROP>
__device__ __forceinline__ float tile_probs(f32x16& s0, f32x16& s1, float m_use, unsigned int key32, unsigned int hoff, unsigned int thresh) {
  float psum = 0.f;
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
        float p = fast_exp2((t == 0 ? s0[r] : s1[r]) - m_use);
        psum += p;
        if (DROP) p = drop_keep(e < 2 ? bits0 : bits1, e & 1, thresh) ? p : 0.f;
        if (t == 0) s0[r] = p; else s1[r] = p;
      }
    }
  }
  return psum;
}

// window geometry of (32-query wave at qw0) x (64-key tile at j0)
struct WinGeom { int mode; int wbc; int mis; };   // mode 0: window, 1: all low end, 2: all high end
__device__ __forceinline__ WinGeom win_geom(int qw0, int j0, int maxrel, int nb) {
  WinGeom g;
  const int bmin = qw0 - (j0 + 63) + maxrel, bmax = qw0 + 31 - j0 + maxrel;   // unclamped bucket range of the rectangle
  g.mode = bmax <= 0 ? 1 : (bmin >= nb - 1 ? 2 : 0);
#if FA2_ABL == 4
  g.mode = g.mode == 0 ? 1 : g.mode;
#endif
  const int wb8 = bmin & ~7;            // floor to a multiple of 8 (two's complement: also for negatives)
  g.wbc = wb8 >> 3;
  g.mis = bmin - wb8;
  return g;
}

// DROP is a launch-time constant (dropout_p > 0), so it is a template parameter: the element-wise blocks exist once per kernel
// instead of twice behind a uniform branch (round 3's GEMM lesson -- code size against a 64 KB instruction cache shared by two
// CUs; bwd_dq_kernel<true> was 7.9 k instructions with its 16-way run-time dispatch).
template <bool BIAS, bool DROP>
__global__ __launch_bounds__(256, 2) void fwd_kernel(const Args a) {
#ifdef FA2_PAD256   // discriminator build (tools/r4/sbs_arms.sh): no third wave beside this kernel's wave and another big one on a SIMD
  ST5_PAD_TO_256_VGPRS();
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kbuf = smem;                 // 2 x 8 KB
  char* vbuf = smem + 2 * TILE_B;    // 2 x 8 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* scratch = smem + 4 * TILE_B + wave * SCR_B;
  const int ql = lane & 31, hi = lane >> 5;
  int xb_, bh;
  block_coords(xb_, bh);
  const int b = bh / a.H, h = bh % a.H;
  const int qblk = xb_ * 128;
  const int qw0 = qblk + wave * 32;
  const int qi = qw0 + ql;
  const int qc = qi < a.T ? qi : a.T - 1;
  const bool qvalid = qi < a.T;

  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(a.q + ((long long)b * a.T + qc) * a.q_ld + h * HD + ks * 16 + hi * 8);

  int nkeys = a.S;
  if (a.causal) {
    const int qmax = (qblk + 127 < a.T ? qblk + 127 : a.T - 1) + (a.S - a.T);
    nkeys = qmax + 1 < a.S ? qmax + 1 : a.S;
  }
  const int ntiles = (nkeys + KT - 1) / KT;

  const bf16_t* kbase = a.k + (long long)b * a.S * a.k_ld + h * HD;
  const bf16_t* vbase = a.v + (long long)b * a.S * a.v_ld + h * HD;
  TileStager ts;
  ts.init(wave, lane);
  const int nbp = a.nb + 2 * QP_PAD;
  const bf16_t* qpb = BIAS ? a.qp + (long long)bh * a.T * nbp : nullptr;
  WinStager wsg;
  if (BIAS) wsg.init(lane, qw0, a.T, nbp);
  float blo = 0.f, bhi = 0.f;   // clipped end buckets of this lane's query
  if (BIAS) {
    blo = (float)qpb[(long long)qc * nbp];
    bhi = (float)qpb[(long long)qc * nbp + QP_PAD + a.nb];
  }
  TrAddr tra;
  tra.init(lane);
  // transpose-read offsets of the V^T fragments: [d tile][first / second 4-row group] for k16 step 0; step s adds 16 rows =
  // 2048 bytes (the swizzle term (row >> 1) & 7 does not see multiples of 16 rows)
  int voff[2][2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    voff[dt][0] = tra.off(4 * hi, dt);
    voff[dt][1] = tra.off(8 + 4 * hi, dt);
  }

  const uint8_t* mrow = a.kpm ? a.kpm + (long long)b * a.S : nullptr;
  unsigned int raw_next = kpm_raw(mrow, lane, a.S);

  // prologue: tile 0 (+ its window)
  ts.issue(kbase, a.k_ld, 0, a.S, kbuf, wave);
  ts.issue(vbase, a.v_ld, 0, a.S, vbuf, wave);
  WinGeom wg = BIAS ? win_geom(qw0, 0, a.maxrel, a.nb) : WinGeom{0, 0, 0};
  if (BIAS && wg.mode == 0) wsg.issue(qpb, wg.wbc, a.nb >> 3, scratch);

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  constexpr bool drop = DROP;
  const unsigned int thresh = drop ? dropout_thresh(a.dropout_p) : 0u;
  const unsigned int seedf = drop ? drop_seed_fold(a.seed) : 0u;   // (once, in front of the tile loop: see drop_seed_fold)
  const float inv_keep = drop ? 1.f / (1.f - a.dropout_p) : 1.f;
  const float sc2 = a.scale * LOG2E;
  const int jmax = a.causal ? qi + (a.S - a.T) : 0x3fffffff;
  const unsigned long long ctr_blk = ((unsigned long long)bh * a.T + (unsigned long long)qc) * (unsigned long long)(drop_row_stride(a.lds) >> 6);
  const unsigned int hoff = hi ? 2u * PAIR_MUL : 0u;
  // scratch read base of this lane: window position of key offset c is (ql + 63 - 4hi - c) + mis; read_bias indexes [59 - c]
  const int rd_lane = (ql * WIN + ql + 4 - 4 * hi) * 2;

  for (int jt = 0; jt < ntiles; ++jt) {
    const char* kt = kbuf + (jt & 1) * TILE_B;
    const char* vt = vbuf + (jt & 1) * TILE_B;
    const int j0 = jt * KT;
    // tile jt (K, V, window) has landed; the barrier also retires every wave's reads of tile jt-1's buffers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned long long kmask = __ballot(raw_next != 0u || j0 + lane >= a.S);
    bf16x8 kfa[4], kfb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kfa[ks] = *reinterpret_cast<const bf16x8*>(kt + lds_off(ql, 2 * ks + hi));
      kfb[ks] = *reinterpret_cast<const bf16x8*>(kt + lds_off(32 + ql, 2 * ks + hi));
    }
    unsigned int braw[32];
    const int bmode = BIAS ? wg.mode : 3;
    if (BIAS && bmode == 0) read_bias(braw, scratch + rd_lane + wg.mis * 2);
    // the bias values are in registers (the compiler waits for them before the next DMA may overwrite the scratch)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (jt + 1 < ntiles) {
      ts.issue(kbase, a.k_ld, j0 + KT, a.S, kbuf + ((jt + 1) & 1) * TILE_B, wave);
      ts.issue(vbase, a.v_ld, j0 + KT, a.S, vbuf + ((jt + 1) & 1) * TILE_B, wave);
      raw_next = kpm_raw(mrow, j0 + KT + lane, a.S);
      if (BIAS) {
        wg = win_geom(qw0, j0 + KT, a.maxrel, a.nb);
        if (wg.mode == 0) wsg.issue(qpb, wg.wbc, a.nb >> 3, scratch);
      }
    }
    const bool need_mask = kmask != 0ull || (a.causal && j0 + 63 > qw0 + (a.S - a.T));
    const int jrel = jmax - j0 - 4 * hi;
    const unsigned long long km = kmask >> (4 * hi);
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[ks], qf[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfb[ks], qf[ks], s1, 0, 0, 0);
    }
    float tmax;
    const float buni = bmode == 1 ? blo : bhi;
    if (bmode == 0) tmax = need_mask ? tile_scores<1, true>(s0, s1, sc2, braw, 0.f, km, jrel) : tile_scores<1, false>(s0, s1, sc2, braw, 0.f, km, jrel);
    else if (bmode == 3) tmax = need_mask ? tile_scores<0, true>(s0, s1, sc2, braw, 0.f, km, jrel) : tile_scores<0, false>(s0, s1, sc2, braw, 0.f, km, jrel);
    else tmax = need_mask ? tile_scores<2, true>(s0, s1, sc2, braw, buni, km, jrel) : tile_scores<2, false>(s0, s1, sc2, braw, buni, km, jrel);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = m_run == -INFINITY ? 0.f : fast_exp2(m_run - m_use);
    float psum;
    if constexpr (DROP) psum = tile_probs<true>(s0, s1, m_use, drop_block_key_folded(seedf, ctr_blk + (unsigned long long)jt), hoff, thresh);
    else psum = tile_probs<false>(s0, s1, m_use, 0u, hoff, thresh);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (__ballot(alpha != 1.f)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    }
    // O^T += V^T . P^T   (4 k-steps of 16 keys, two 32-row d tiles; V^T fragments by transpose reads of the [key][d] tile)
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      const bf16x8 pf = pack8(sidx < 2 ? s0 : s1, 8 * (sidx & 1));
      const bf16x8 v0 = tr_frag(vt + sidx * 2048, voff[0][0], voff[0][1]);
      const bf16x8 v1 = tr_frag(vt + sidx * 2048, voff[1][0], voff[1][1]);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf, o1, 0, 0, 0);
    }
  }

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
}


// =====================================================================================================================
// Backward.  dq kernel (lane = query, loops over key tiles) -> dQ and the clipped end buckets of dQP;
//            dkv kernel (lane = key, loops over query tiles) -> dK, dV and the unclipped buckets of dQP (coalesced).
// Both recompute P from (Q, K, QP, LSE); D[bh,q] = dO[q].O[q] comes from dvec_kernel.  No atomics.
// =====================================================================================================================
struct BwdArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* o; const bf16_t* dout;
  bf16_t* dq; bf16_t* dk; bf16_t* dv;
  const float* lse; float* dvec;
  const bf16_t* qp;                         // [BH, T, nb + 16] (fa2 layout) or NULL
  bf16_t* dqp;                              // [BH, T, nb] (plain), zero-initialised by the host
  const uint8_t* kpm;
  long long q_ld, k_ld, v_ld, o_ld, do_ld, dq_ld, dk_ld, dv_ld;
  int B, H, T, S, nb, maxrel, causal, lds;
  float scale, dropout_p;
  unsigned long long seed;
  int dvec_in_dq;   // 1 (dq and dkv on ONE stream): the dq kernel computes D = dO . O itself, writes dvec for the dkv kernel behind it, and clears dQP
};

__global__ __launch_bounds__(256) void dvec_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, float* __restrict__ dvec,
                                                   long long o_ld, long long do_ld, int H, int T, long long rows) {
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

// dS of one 32-key half tile (t) for the dq kernel.  In: s = raw q.k, dp = dO.V^T; out: s = dS = P * (dP_drop - D).
// CLIP: the rectangle touches a clipped end bucket (unclamped bucket <= 0 or >= nb-1): those elements' dS are summed into
// acc_lo / acc_hi.  b0 = unclamped bucket of key offset 0 of the 64-key tile for this lane (decreases with the key offset).
// The three running sums come back BY VALUE (DqSums): as reference parameters -- threaded through sixteen differently instantiated
// inlined call sites behind uniform branches -- hipcc kept acc_lo / acc_hi / csum in SCRATCH memory (private_segment_fixed_size 12,
// 68 scratch_store + 68 scratch_load per tile): every `acc_lo += ds` was a round trip through memory.  That, not the window DMA or the
// LDS reads, was the 2.2x of bwd_dq_kernel<true> over its no-bias form (round-4 ablations: no DMA -5 us, no LDS reads -3 us, all tiles
// uniform -52 us of 105.6 us; profiles/r4_attention.md).
struct DqSums { float lo, hi, c; };
template <int BMODE /*0 none, 1 per element, 2 uniform*/, bool CLIP, bool MASK, bool DROP>
__device__ __forceinline__ DqSums dq_half(const int t, f32x16& s, const f32x16& dpv, float sc2, const unsigned int (&braw)[16], float buni,
                                          int b0, int nbm1, unsigned long long km, int jrel, float lse2, float dsum, unsigned int key32,
                                          unsigned int hoff, unsigned int thresh, float inv_keep, const DqSums in) {
#pragma clang fp contract(off)   // (see dq_tile in flash_attn.hip: the product is rounded before it is added, in both generations)
  float acc_lo = in.lo, acc_hi = in.hi, csum = in.c;     // (continued, not restarted: the summation order of the first generation)
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
      const float b = BMODE == 1 ? __uint_as_float(braw[r] << 16) : (BMODE == 2 ? buni : 0.f);
      const float x = fmaf(s[r], sc2, b);
      float p = fast_exp2(x - lse2);
      if (MASK) { if (((km >> c) & 1ull) || c > jrel) p = 0.f; }
      float dp = dpv[r];
      if (DROP) dp = drop_keep(e < 2 ? bits0 : bits1, e & 1, thresh) ? dp * inv_keep : 0.f;
      const float ds = p * (dp - dsum);
      s[r] = ds;
      if (BMODE == 2) csum += ds;
      if (BMODE == 1 && CLIP) {
        const int bk = b0 - c;
        if (bk <= 0) acc_lo += ds;
        else if (bk >= nbm1) acc_hi += ds;
      }
    }
  }
  return DqSums{acc_lo, acc_hi, csum};
}

// bias values of one 32-key half tile: element r is key offset c = 32t + (r&3) + 8(r>>2) (+ 4 hi)
__device__ __forceinline__ void read_bias_half(const int t, unsigned int (&braw)[16], const char* rd) {
  const unsigned short* p = reinterpret_cast<const unsigned short*>(rd);
#if FA2_ABL == 2
  for (int r = 0; r < 16; ++r) { braw[r] = 0x3c00u + r; asm volatile("" : "+v"(braw[r])); }
  return;
#endif
#pragma unroll
  for (int r = 0; r < 16; ++r) braw[r] = p[59 - (32 * t + (r & 3) + 8 * (r >> 2))];
}

template <bool BIAS, bool DROP>
__global__ __launch_bounds__(256, 2) void bwd_dq_kernel(const BwdArgs a) {
#ifdef FA2_PAD256   // discriminator build (tools/r4/sbs_arms.sh): no third wave beside this kernel's wave and another big one on a SIMD
  ST5_PAD_TO_256_VGPRS();
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kbuf = smem;                 // 2 x 8 KB
  char* vbuf = smem + 2 * TILE_B;    // 2 x 8 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* scratch = smem + 4 * TILE_B + wave * SCR_B;
  const int ql = lane & 31, hi = lane >> 5;
  int xb_, bh;
  block_coords(xb_, bh);
  const int b = bh / a.H, h = bh % a.H;
  const int qblk = xb_ * 128;
  const int qw0 = qblk + wave * 32;
  const int qi = qw0 + ql;
  const int qc = qi < a.T ? qi : a.T - 1;
  const bool qvalid = qi < a.T;

  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(a.q + ((long long)b * a.T + qc) * a.q_ld + h * HD + ks * 16 + hi * 8);
    dof[ks] = *reinterpret_cast<const bf16x8*>(a.dout + ((long long)b * a.T + qc) * a.do_ld + h * HD + ks * 16 + hi * 8);
  }
  const float lse2 = a.lse[(long long)bh * a.T + qc] * LOG2E;   // +inf for fully masked rows -> P = 0
  float dsum;
  if (a.dvec_in_dq) {
    // D[q] = dO[q] . O[q] here instead of in dvec_kernel (48 launches of ~6 us per update): the lane's 32 of the 64 head dims in
    // the SAME order as dvec_kernel's thread (row, hi) -- ks-major, then element -- and the two halves added once, so the value is
    // bit-identical; written out for the dkv kernel, which runs behind this one on the same stream
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(a.o + ((long long)b * a.T + qc) * a.o_ld + h * HD + ks * 16 + hi * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) part = fmaf((float)dof[ks][e], (float)of[e], part);
    }
    dsum = part + __shfl_xor(part, 32, 64);
    if (qvalid && hi == 0) a.dvec[(long long)bh * a.T + qi] = dsum;
  } else {
    dsum = a.dvec[(long long)bh * a.T + qc];
  }

  if (BIAS && a.dvec_in_dq) {
    // dQP rows of this block's queries start as zeros (buckets whose key falls outside [0, S) are never written): cleared here with
    // 16-byte stores instead of by a hipMemsetAsync over the whole array in front of every backward (24 fills of ~8 us per update).
    // The dkv kernel (window buckets) runs behind this kernel on the same stream; this block's own end-bucket stores come after the
    // tile loop, whose vmcnt(0) + barrier pairs order them behind these stores.
    const int nrows = a.T - qblk < 128 ? a.T - qblk : 128;
    uint4* zp = reinterpret_cast<uint4*>(a.dqp + ((long long)bh * a.T + qblk) * a.nb);
    const int n16 = nrows * (a.nb >> 3);
    for (int i = tid; i < n16; i += 256) zp[i] = make_uint4(0u, 0u, 0u, 0u);
    // explicit order against this block's own end-bucket stores behind the tile loop: the loop's vmcnt(0) + barrier
    // pairs would do it too, but only while a block has at least one key tile (causal with S < T can leave a query block none)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  int nkeys = a.S;
  if (a.causal) {
    const int qmax = (qblk + 127 < a.T ? qblk + 127 : a.T - 1) + (a.S - a.T);
    nkeys = qmax + 1 < a.S ? qmax + 1 : a.S;
  }
  const int ntiles = (nkeys + KT - 1) / KT;
  const bf16_t* kbase = a.k + (long long)b * a.S * a.k_ld + h * HD;
  const bf16_t* vbase = a.v + (long long)b * a.S * a.v_ld + h * HD;
  TileStager ts;
  ts.init(wave, lane);
  const int nbp = a.nb + 2 * QP_PAD;
  const bf16_t* qpb = BIAS ? a.qp + (long long)bh * a.T * nbp : nullptr;
  WinStager wsg;
  if (BIAS) wsg.init(lane, qw0, a.T, nbp);
  float blo = 0.f, bhi = 0.f;
  if (BIAS) {
    blo = (float)qpb[(long long)qc * nbp];
    bhi = (float)qpb[(long long)qc * nbp + QP_PAD + a.nb];
  }
  TrAddr tra;
  tra.init(lane);
  int koff[2][2];   // transpose-read offsets of the K^T fragments for k16 step 0 (same geometry as V^T in the forward)
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    koff[dt][0] = tra.off(4 * hi, dt);
    koff[dt][1] = tra.off(8 + 4 * hi, dt);
  }
  const uint8_t* mrow = a.kpm ? a.kpm + (long long)b * a.S : nullptr;
  unsigned int raw_next = kpm_raw(mrow, lane, a.S);

  ts.issue(kbase, a.k_ld, 0, a.S, kbuf, wave);
  ts.issue(vbase, a.v_ld, 0, a.S, vbuf, wave);
  WinGeom wg = BIAS ? win_geom(qw0, 0, a.maxrel, a.nb) : WinGeom{0, 0, 0};
  if (BIAS && wg.mode == 0) wsg.issue(qpb, wg.wbc, a.nb >> 3, scratch);

  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  float acc_lo = 0.f, acc_hi = 0.f;
  constexpr bool drop = DROP;
  const unsigned int thresh = drop ? dropout_thresh(a.dropout_p) : 0u;
  const unsigned int seedf = drop ? drop_seed_fold(a.seed) : 0u;   // (once, in front of the tile loop: see drop_seed_fold)
  const float inv_keep = drop ? 1.f / (1.f - a.dropout_p) : 1.f;
  const float sc2 = a.scale * LOG2E;
  const int jmax = a.causal ? qi + (a.S - a.T) : 0x3fffffff;
  const unsigned long long ctr_blk = ((unsigned long long)bh * a.T + (unsigned long long)qc) * (unsigned long long)(drop_row_stride(a.lds) >> 6);
  const unsigned int hoff = hi ? 2u * PAIR_MUL : 0u;
  const int rd_lane = (ql * WIN + ql + 4 - 4 * hi) * 2;

  for (int jt = 0; jt < ntiles; ++jt) {
    const char* kt = kbuf + (jt & 1) * TILE_B;
    const char* vt = vbuf + (jt & 1) * TILE_B;
    const int j0 = jt * KT;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned long long kmask = __ballot(raw_next != 0u || j0 + lane >= a.S);
    unsigned int braw0[16], braw1[16];
    const int bmode = BIAS ? wg.mode : 3;
    // unclamped bucket range of this wave's rectangle: does it touch a clipped end?
    const int bmin = qw0 - (j0 + 63) + a.maxrel, bmax = qw0 + 31 - j0 + a.maxrel;
    const bool clip = BIAS && (bmin <= 0 || bmax >= a.nb - 1);
    if (BIAS && bmode == 0) {
      read_bias_half(0, braw0, scratch + rd_lane + wg.mis * 2);
      read_bias_half(1, braw1, scratch + rd_lane + wg.mis * 2);
    }
    const bool need_mask = kmask != 0ull || (a.causal && j0 + 63 > qw0 + (a.S - a.T));
    const int jrel = jmax - j0 - 4 * hi;
    const unsigned long long km = kmask >> (4 * hi);
    unsigned int key32 = 0u;
    if (drop) key32 = drop_block_key_folded(seedf, ctr_blk + (unsigned long long)jt);
    const float buni = bmode == 1 ? blo : bhi;
    const int b0 = qi - j0 - 4 * hi + a.maxrel;   // unclamped bucket of key offset c = 0
    float csum = 0.f;
    // the tile is processed as two 32-key halves (S, dP, element-wise, dQ each): half the live accumulators
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x16 sv, pv;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sv[r] = 0.f; pv[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kfr = *reinterpret_cast<const bf16x8*>(kt + lds_off(32 * t + ql, 2 * ks + hi));
        const bf16x8 vfr = *reinterpret_cast<const bf16x8*>(vt + lds_off(32 * t + ql, 2 * ks + hi));
        sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr, qf[ks], sv, 0, 0, 0);
        pv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, dof[ks], pv, 0, 0, 0);   // dP^T[key][q]
      }
      if (t == 0) {
        // both halves' window values are in registers before the next DMA may overwrite the scratch (the K / V buffers of the
        // next tile are the ones read two tiles ago: safe behind this tile's barrier)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (jt + 1 < ntiles) {
          ts.issue(kbase, a.k_ld, j0 + KT, a.S, kbuf + ((jt + 1) & 1) * TILE_B, wave);
          ts.issue(vbase, a.v_ld, j0 + KT, a.S, vbuf + ((jt + 1) & 1) * TILE_B, wave);
          raw_next = kpm_raw(mrow, j0 + KT + lane, a.S);
          if (BIAS) {
            wg = win_geom(qw0, j0 + KT, a.maxrel, a.nb);
            if (wg.mode == 0) wsg.issue(qpb, wg.wbc, a.nb >> 3, scratch);
          }
        }
      }
#define DQ_HALF(BM_, CL_, MASK_, DROP_) \
  hs = dq_half<BM_, CL_, MASK_, DROP_>(t, sv, pv, sc2, t == 0 ? braw0 : braw1, buni, b0, a.nb - 1, km, jrel, lse2, dsum, key32, hoff, thresh, inv_keep, hs)
#define DQ_DISPATCH(MASK_, DROP_)                                                            \
  do {                                                                                       \
    if (bmode == 3) DQ_HALF(0, false, MASK_, DROP_);                                         \
    else if (bmode == 0) { if (clip) DQ_HALF(1, true, MASK_, DROP_); else DQ_HALF(1, false, MASK_, DROP_); } \
    else DQ_HALF(2, false, MASK_, DROP_);                                                    \
  } while (0)
      DqSums hs{acc_lo, acc_hi, csum};
      if (need_mask) DQ_DISPATCH(true, DROP); else DQ_DISPATCH(false, DROP);
#undef DQ_DISPATCH
#undef DQ_HALF
      acc_lo = hs.lo; acc_hi = hs.hi; csum = hs.c;
      // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]   (K^T fragments by transpose reads of the [key][d] tile)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int sidx = 2 * t + u;
        const bf16x8 df = pack8(sv, 8 * u);
        const bf16x8 k0 = tr_frag(kt + sidx * 2048, koff[0][0], koff[0][1]);
        const bf16x8 k1 = tr_frag(kt + sidx * 2048, koff[1][0], koff[1][1]);
        dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, df, dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, df, dq1, 0, 0, 0);
      }
    }
    if (bmode == 1) acc_lo += csum;
    if (bmode == 2) acc_hi += csum;
  }
  if (BIAS) {
    acc_lo += __shfl_xor(acc_lo, 32, 64);
    acc_hi += __shfl_xor(acc_hi, 32, 64);
    if (qvalid && hi == 0) {
      bf16_t* dqp_row = a.dqp + ((long long)bh * a.T + qi) * a.nb;
      dqp_row[0] = (bf16_t)acc_lo; dqp_row[a.nb - 1] = (bf16_t)acc_hi;
    }
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

// ---------------------------------------------------------------------------------------------------------------------
// dkv kernel.  Block = 128 keys (wave = 32 keys, lane = key kw0 + (l&31)); loops over 64-query tiles staged by LDS-DMA
// (Q [q][d], dO [q][d]; lse2 / D / dropout keys of the tile in a small side array).  Per 32-query sub-tile:
//   S[q][key] = Q.K^T, dP[q][key] = dO.V^T (A = Q / dO rows, B = this lane's K / V fragments, resident in registers),
//   element-wise -> Pd, dS;  dV^T[d][key] += dO^T.Pd, dK^T[d][key] += Q^T.dS (A by transpose reads of the same tiles).
// Bias: the sub-tile's window (32 queries x 72 buckets) is DMAed into the wave's scratch one sub-tile ahead.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int KWCH = 9;                  // window chunks per query in the dkv orientation (63 buckets + misalignment <= 72)
constexpr int KWIN = 72;
constexpr int KSCR_B = 5120;             // 5 DMA instructions x 1 KB (32 x 9 = 288 chunks used)
constexpr int QSIDE_B = 1536;            // lse2[64] | D[64] | dropout keys[128] | low-end bias[64] | high-end bias[64]
constexpr int QBUF_B = 2 * TILE_B + QSIDE_B;

struct KWinStager {
  int qrow[5], ch[5];
  __device__ __forceinline__ void init(int lane) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      int ci = i * 64 + lane;
      ci = ci < 32 * KWCH ? ci : 32 * KWCH - 1;
      qrow[i] = ci / KWCH;
      ch[i] = ci % KWCH;
    }
  }
  // q0: first query of the sub-tile
  __device__ __forceinline__ void issue(const bf16_t* qpb, int q0, int T, int nbp, int wbc, int nchunks_data, char* scratch) const {
#if FA2_ABL == 1 || FA2_ABL == 2
    return;
#endif
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      int q = q0 + qrow[i];
      q = q < T ? q : T - 1;
      int c = wbc + ch[i] + 1;
      c = c < 0 ? 0 : (c > nchunks_data + 1 ? nchunks_data + 1 : c);
      const bf16_t* src = qpb + (unsigned int)q * (unsigned int)nbp + (unsigned int)c * 8u;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(scratch + i * 1024), 16, 0, 0);
      FA2_AFTER_DMA(src);
    }
  }
};

// One 32-query sub-tile: in s = raw S[q][key], dp = dO.V; out pd = dropped P, s = dS.  Element r <-> query q0 + (r&3) + 8(r>>2)
// (q0 includes 4*hi).  Unclipped bucket gradients dQP[q][q - key + maxrel] are stored here (consecutive lanes = consecutive
// keys = consecutive buckets, descending: coalesced).
template <int BMODE /*0 none, 1 window (stores the unclipped bucket gradients), 2 clipped: per-query end value, no stores*/, bool SLOW, bool DROP>
__device__ __forceinline__ void dkv_sub(f32x16& s, const f32x16& dp, f32x16& pd, const BwdArgs& a, const float* stv, const unsigned int* keyv,
                                        const unsigned int (&braw)[16], bf16_t* dqpb, int qbase, int q0, int ki, bool kvalid,
                                        bool kmasked, float sc2, unsigned int pcl, unsigned int kshift, unsigned int thresh, float inv_keep) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = (r & 3) + 8 * (r >> 2);
    const int qq = q0 + c;
    const int bk = qq - ki + a.maxrel;     // unclamped bucket
    const float bv = BMODE == 0 ? 0.f : __uint_as_float(braw[r]);
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
    if (BMODE == 1) {
      bool st_ok = kvalid && bk > 0 && bk < a.nb - 1;
      if (SLOW) st_ok = st_ok && qq < a.T;
#if FA2_ABL != 3
      if (st_ok) dqpb[(unsigned int)(qq * a.nb + bk)] = (bf16_t)ds;
#else
      asm volatile("" :: "v"(st_ok));
#endif
    }
  }
}

template <bool BIAS, bool DROP>
__global__ __launch_bounds__(256, 2) void bwd_dkv_kernel(const BwdArgs a) {
#ifdef FA2_PAD256   // discriminator build (tools/r4/sbs_arms.sh): no third wave beside this kernel's wave and another big one on a SIMD
  ST5_PAD_TO_256_VGPRS();
#endif
#ifdef FA2_DKV_HEAD_PAD   // discriminator build (tools/r4/sbs_arms.sh): the kernel's data starts 16 KB into its LDS allocation
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  char* const smem = smem_raw + 16 * 1024;
#else
  extern __shared__ __attribute__((aligned(16))) char smem[];
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* scratch = smem + 2 * QBUF_B + wave * KSCR_B;
  const int kl = lane & 31, hi = lane >> 5;
  int xb_, bh;
  block_coords(xb_, bh);
  const int b = bh / a.H, h = bh % a.H;
  const int kblk = xb_ * 128;
  const int kw0 = kblk + wave * 32;
  const int ki = kw0 + kl;
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
  int qt0 = 0;
  if (a.causal) { const int qmin = kblk - (a.S - a.T); qt0 = qmin > 0 ? qmin / 64 : 0; }
  const int nqt = (a.T + 63) / 64;

  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
  constexpr bool drop = DROP;
  const unsigned int thresh = drop ? dropout_thresh(a.dropout_p) : 0u;
  const unsigned int seedf = drop ? drop_seed_fold(a.seed) : 0u;   // (once, in front of the tile loop: see drop_seed_fold)
  const float inv_keep = drop ? 1.f / (1.f - a.dropout_p) : 1.f;
  const float sc2 = a.scale * LOG2E;
  const unsigned int pcl = ((unsigned int)(ki & 63) >> 1) * PAIR_MUL;
  const unsigned int kshift = (ki & 1) ? 16u : 0u;
  const long long bhT = (long long)bh * a.T;
  const int nbp = a.nb + 2 * QP_PAD;
  const bf16_t* qpb = BIAS ? a.qp + bhT * nbp : nullptr;
  bf16_t* dqpb = BIAS ? a.dqp + bhT * a.nb : nullptr;
  const bf16_t* qbase_g = a.q + (long long)b * a.T * a.q_ld + h * HD;
  const bf16_t* dobase_g = a.dout + (long long)b * a.T * a.do_ld + h * HD;

  TileStager ts;
  ts.init(wave, lane);
  KWinStager kws;
  if (BIAS) kws.init(lane);
  TrAddr tra;
  tra.init(lane);
  int toff[2][2];   // transpose-read offsets inside a [q][d] tile for k16 step 0: [d tile][row group]; step s4 adds 2048 bytes
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    toff[dt][0] = tra.off(4 * hi, dt);
    toff[dt][1] = tra.off(8 + 4 * hi, dt);
  }
  // window geometry of sub-tile n (queries 32n ..) against this wave's keys
  auto sub_geom = [&](int n, int& mode, int& wbc, int& mis) {
    const int bmin = 32 * n - (kw0 + 31) + a.maxrel, bmax = 32 * n + 31 - kw0 + a.maxrel;
    mode = bmax <= 0 ? 1 : (bmin >= a.nb - 1 ? 2 : 0);
#if FA2_ABL == 4
    mode = mode == 0 ? 1 : mode;
#endif
    const int wb8 = bmin & ~7;
    wbc = wb8 >> 3;
    mis = bmin - wb8;
  };
  // side values of a query tile: lse2, D, dropout block keys -> registers (threads < 128), later to LDS.  side_load() only ISSUES the
  // global loads (clamped addresses, raw bits): every use of a loaded value -- the log2e scaling, the "past T" selects, the bf16 -> fp32
  // widening -- sits in side_store() at the END of the iteration.  (Round 4's form scaled lse inside side_load, so hipcc waited
  // vmcnt(0) right behind it: with the next tile's four LDS-DMA instructions just issued in front, every iteration began by waiting
  // out the whole tile prefetch it had been issued to hide.)
  float lse_raw = 0.f, d_raw = 0.f;
  unsigned short end_raw = 0;
  int side_q = 0;
  unsigned int dkey = 0u;
  auto side_load = [&](int q0t) {
    side_q = q0t + (tid & 63);
    const int qcl = side_q < a.T ? side_q : a.T - 1;
    if (tid < 64) {
      lse_raw = a.lse[bhT + qcl];
      d_raw = a.dvec[bhT + qcl];
    }
    if (BIAS && tid >= 128)   // clipped end buckets of the tile's queries: threads 128-191 the low end, 192-255 the high end
      end_raw = reinterpret_cast<const unsigned short*>(qpb)[(unsigned int)qcl * (unsigned int)nbp + (tid < 192 ? 0u : (unsigned int)(QP_PAD + a.nb))];
    if (tid < 128 && drop) {
      const unsigned long long row = (unsigned long long)bhT + (unsigned long long)qcl;
      dkey = drop_block_key_folded(seedf, row * (unsigned long long)(drop_row_stride(a.lds) >> 6) + (unsigned long long)((kblk >> 6) + (tid >> 6)));
    }
  };
  auto side_store = [&](char* buf) {
    float* st = reinterpret_cast<float*>(buf + 2 * TILE_B);
    if (tid < 64) { st[tid] = side_q < a.T ? lse_raw * LOG2E : INFINITY; st[64 + tid] = side_q < a.T ? d_raw : 0.f; }
    if (tid < 128) reinterpret_cast<unsigned int*>(st)[128 + tid] = dkey;
    else if (BIAS) reinterpret_cast<unsigned int*>(st)[128 + tid] = (unsigned int)end_raw << 16;    // [256 .. 319] low end, [320 .. 383] high end
  };

  int gmode = 3, gwbc = 0, gmis = 0;   // geometry of the sub-tile whose window is in flight / in the scratch
  if (qt0 < nqt) {
    char* buf = smem + (qt0 & 1) * QBUF_B;
    ts.issue(qbase_g, a.q_ld, qt0 * 64, a.T, buf, wave);
    ts.issue(dobase_g, a.do_ld, qt0 * 64, a.T, buf + TILE_B, wave);
    side_load(qt0 * 64);
    side_store(buf);
    if (BIAS) {
      sub_geom(2 * qt0, gmode, gwbc, gmis);
      if (gmode == 0) kws.issue(qpb, 64 * qt0, a.T, nbp, gwbc, a.nb >> 3, scratch);
    }
  }
  // scratch read base: window position of (query offset c in the sub-tile, this key) = c + 4hi - kl + 31 + mis
  const int rd_lane = ((4 * hi) * (KWIN + 1) + 31 - kl) * 2;

  for (int qt = qt0; qt < nqt; ++qt) {
    char* buf = smem + (qt & 1) * QBUF_B;
    const char* qtl = buf; const char* otl = buf + TILE_B;
    const float* stv = reinterpret_cast<const float*>(buf + 2 * TILE_B);
    const unsigned int* keyv = reinterpret_cast<const unsigned int*>(buf + 2 * TILE_B) + 128 + (wave >> 1) * 64;
    // vmcnt(0): this wave's share of tile qt (LDS-DMA) and its window have landed.  lgkmcnt(0): its ds_writes of the tile's side
    // array (side_store at the bottom of the previous iteration) have EXECUTED before it arrives at the barrier.  The second wait is
    // the root cause of rounds 3-4's cross-stream irreproducibility: __syncthreads()'s release fence asks for
    // it, but hipcc (ROCm 7.2) drops it at this loop header -- the barrier was reached behind `s_waitcnt vmcnt(0)` alone (tools/
    // barrier_audit.py finds exactly these kernels in the whole library) -- and with a block of ANOTHER launch's bias kernel on the
    // CU (32 ds_read_u16 per lane and tile in the same SIMD's LDS queue) a wave on another SIMD read the side array before the
    // write had executed: 20 % of the launches of tools/r5/dkv_pair.py had wrong dK / dV rows; 0 of 800 with this wait.
#ifdef FA2_NO_LGKM_BARRIER   // (the round-4 form, for the reproducer's A/B arm only)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    __syncthreads();   // tile qt (DMA + side values) visible; every wave is done with tile qt-1's buffer
    if (qt + 1 < nqt) {
      char* nb_ = smem + ((qt + 1) & 1) * QBUF_B;
      ts.issue(qbase_g, a.q_ld, (qt + 1) * 64, a.T, nb_, wave);
      ts.issue(dobase_g, a.do_ld, (qt + 1) * 64, a.T, nb_ + TILE_B, wave);
      side_load((qt + 1) * 64);
    }
    const bool slow = any_kmasked || (a.causal && kw0 + 31 > qt * 64 + (a.S - a.T)) || qt * 64 + 64 > a.T;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int n = 2 * qt + sub;
      const int bmode = BIAS ? gmode : 3;
      const int mis = gmis;
      unsigned int braw[16];   // fp32 bit patterns of the bias values
      if (BIAS && bmode == 0) {
        if (sub == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the window issued during sub 0 (own DMA, own reads)
#ifdef FA2_SUB1_SLEEP
        if (sub == 1) asm volatile("s_sleep 4" ::: "memory");
#endif
        const unsigned short* pw = reinterpret_cast<const unsigned short*>(scratch + rd_lane + mis * 2);
#if FA2_ABL == 2
        for (int r = 0; r < 16; ++r) { braw[r] = 0x3c000000u + (r << 16); asm volatile("" : "+v"(braw[r])); }
        (void)pw;
#else
#pragma unroll
        for (int r = 0; r < 16; ++r) braw[r] = (unsigned int)pw[((r & 3) + 8 * (r >> 2)) * (KWIN + 1)] << 16;
#endif
      } else if (BIAS && bmode != 3) {   // fully clipped sub-tile: the queries' end values from the side array
        const unsigned int* pe_ = reinterpret_cast<const unsigned int*>(stv) + (bmode == 1 ? 256 : 320) + sub * 32 + 4 * hi;
#pragma unroll
        for (int r = 0; r < 16; ++r) braw[r] = pe_[(r & 3) + 8 * (r >> 2)];
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
      const int qs0 = qt * 64 + sub * 32;
      const int qbase = sub * 32 + 4 * hi, q0 = qs0 + 4 * hi;
      // window of the NEXT sub-tile: issued once this sub-tile's values are in registers
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (BIAS) {
        const int nn = n + 1;
        if (nn < 2 * nqt) {
          sub_geom(nn, gmode, gwbc, gmis);
          if (gmode == 0) kws.issue(qpb, 32 * nn, a.T, nbp, gwbc, a.nb >> 3, scratch);
        }
      }
      f32x16 pd;
#define DKV_SUB(BM_, SLOW_, DROP_) \
  dkv_sub<BM_, SLOW_, DROP_>(s, dp, pd, a, stv, keyv, braw, dqpb, qbase, q0, ki, kvalid, kmasked, sc2, pcl, kshift, thresh, inv_keep)
#define DKV_DISPATCH(SLOW_, DROP_)                            \
  do {                                                        \
    if (bmode == 3) DKV_SUB(0, SLOW_, DROP_);                 \
    else if (bmode == 0) DKV_SUB(1, SLOW_, DROP_);            \
    else DKV_SUB(2, SLOW_, DROP_);                            \
  } while (0)
      if (slow) DKV_DISPATCH(true, DROP); else DKV_DISPATCH(false, DROP);
#undef DKV_DISPATCH
#undef DKV_SUB
      // dV^T[d][key] += dO^T[d][q] . Pd[q][key] ;  dK^T[d][key] += Q^T[d][q] . dS[q][key]
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int s4 = 2 * sub + u;
        const bf16x8 pf = pack8(pd, 8 * u);
        const bf16x8 df = pack8(s, 8 * u);
        dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(otl + s4 * 2048, toff[0][0], toff[0][1]), pf, dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(otl + s4 * 2048, toff[1][0], toff[1][1]), pf, dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(qtl + s4 * 2048, toff[0][0], toff[0][1]), df, dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(qtl + s4 * 2048, toff[1][0], toff[1][1]), df, dk1, 0, 0, 0);
      }
    }
    if (qt + 1 < nqt) side_store(smem + ((qt + 1) & 1) * QBUF_B);
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

}  // namespace fa2

// ------------------------------------------------------------------------------------------------------------------
// C ABI.  The public st5_flash_attn_* entry points live here and dispatch between the two generations
// (st5_flash_attn_set_impl: 2 = this file (default), 1 = flash_attn.hip, kept for A/B measurements and as the reference the
// tests compare against bit for bit where the arithmetic is identical).
// ------------------------------------------------------------------------------------------------------------------
extern "C" {
int st5_flash1_attn_fwd_qp(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld, void* o, int64_t o_ld,
                           float* lse, const void* pe, const uint8_t* kpm, int32_t B, int32_t H, int32_t T, int32_t S, int32_t head_dim,
                           int32_t nb, int32_t maxrel, int32_t causal, int32_t lds, float scale, float dropout_p, uint64_t seed,
                           void* qp_out, int dtype, void* stream);
int st5_flash1_attn_bwd_2s(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld, const void* o,
                           int64_t o_ld, const void* dout, int64_t do_ld, void* dq, int64_t dq_ld, void* dk, int64_t dk_ld, void* dv,
                           int64_t dv_ld, const float* lse, float* dvec, const void* pe, const void* qp, void* dqp, const uint8_t* kpm,
                           int32_t B, int32_t H, int32_t T, int32_t S, int32_t head_dim, int32_t nb, int32_t maxrel, int32_t causal,
                           int32_t lds, float scale, float dropout_p, uint64_t seed, int dtype, void* stream, void* stream2);
int st5_stream_fork(void* from_stream, void* to_stream);
}

namespace {
int g_impl = 2;
bool g_attr = false;
int set_attrs() {
  if (g_attr) return ST5_OK;
  const void* fns[] = {(const void*)fa2::fwd_kernel<true, true>, (const void*)fa2::fwd_kernel<false, true>, (const void*)fa2::fwd_kernel<true, false>,
                       (const void*)fa2::fwd_kernel<false, false>, (const void*)fa2::bwd_dq_kernel<true, true>, (const void*)fa2::bwd_dq_kernel<false, true>,
                       (const void*)fa2::bwd_dq_kernel<true, false>, (const void*)fa2::bwd_dq_kernel<false, false>, (const void*)fa2::bwd_dkv_kernel<true, true>,
                       (const void*)fa2::bwd_dkv_kernel<false, true>, (const void*)fa2::bwd_dkv_kernel<true, false>, (const void*)fa2::bwd_dkv_kernel<false, false>};
  for (const void* f : fns)
#ifdef FA2_DKV_FAT_LDS
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) return ST5_ERR_LAUNCH;
#else
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return ST5_ERR_LAUNCH;
#endif
  g_attr = true;
  return ST5_OK;
}
}  // namespace

extern "C" int st5_flash_attn_set_impl(int impl) {
  if (impl != 1 && impl != 2) return ST5_ERR_ARG;
  g_impl = impl;
  return ST5_OK;
}
extern "C" int32_t st5_flash_attn_qp_row(int32_t nb) { return g_impl == 2 ? nb + 2 * fa2::QP_PAD : nb; }

extern "C" int st5_flash_attn_qp_table(const void* q, int64_t q_ld, const void* pe, void* qp_out, int32_t B, int32_t H, int32_t T,
                                       int32_t nb, float scale, int dtype, void* stream) {
  if (!q || !pe || !qp_out || B <= 0 || H <= 0 || T <= 0 || nb <= 0 || nb % 8 || dtype != ST5_BF16 || q_ld % 8) return ST5_ERR_ARG;
  const size_t pe_bytes = (size_t)nb * fa2::HD * 2;
  if (pe_bytes <= 64 * 1024) {
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)fa2::qp_table_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) return ST5_ERR_LAUNCH;
      attr = true;
    }
    hipLaunchKernelGGL(fa2::qp_table_kernel<true>, dim3((T + 127) / 128, B * H), dim3(256), pe_bytes, (hipStream_t)stream, (const bf16_t*)q,
                       (long long)q_ld, (const bf16_t*)pe, (bf16_t*)qp_out, H, T, nb, scale * fa2::LOG2E);
  } else {
    hipLaunchKernelGGL(fa2::qp_table_kernel<false>, dim3((T + 127) / 128, B * H), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q,
                       (long long)q_ld, (const bf16_t*)pe, (bf16_t*)qp_out, H, T, nb, scale * fa2::LOG2E);
  }
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_flash_attn_fwd_qp(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld, void* o,
                                     int64_t o_ld, float* lse, const void* pe, const uint8_t* kpm, int32_t B, int32_t H, int32_t T,
                                     int32_t S, int32_t head_dim, int32_t nb, int32_t maxrel, int32_t causal, int32_t lds, float scale,
                                     float dropout_p, uint64_t seed, void* qp_out, int dtype, void* stream) {
  if (g_impl == 1 || (pe && !qp_out))   // no table workspace: the first generation keeps its table in LDS
    return st5_flash1_attn_fwd_qp(q, q_ld, k, k_ld, v, v_ld, o, o_ld, lse, pe, kpm, B, H, T, S, head_dim, nb, maxrel, causal, lds, scale,
                                  dropout_p, seed, g_impl == 1 ? qp_out : nullptr, dtype, stream);
  if (!q || !k || !v || !o || B <= 0 || H <= 0 || T <= 0 || S <= 0) return ST5_ERR_ARG;
  if (dtype != ST5_BF16 || head_dim != fa2::HD) return ST5_ERR_ARG;
  if (q_ld % 8 || k_ld % 8 || v_ld % 8 || o_ld % 4) return ST5_ERR_ALIGN;
  if (pe && (nb != 2 * maxrel || nb % 8 || nb > 1024)) return ST5_ERR_ARG;
  if ((long long)B * H * T * (pe ? nb + 16 : 1) >= (1ll << 31)) return ST5_ERR_ARG;   // 32-bit element offsets into the table
  if (set_attrs() != ST5_OK) return ST5_ERR_LAUNCH;
  hipStream_t s = (hipStream_t)stream;
  if (pe) {
    const int rc = st5_flash_attn_qp_table(q, q_ld, pe, qp_out, B, H, T, nb, scale, dtype, stream);
    if (rc) return rc;
  }
  fa2::Args a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o; a.lse = lse;
  a.qp = pe ? (const bf16_t*)qp_out : nullptr; a.kpm = kpm;
  a.q_ld = q_ld; a.k_ld = k_ld; a.v_ld = v_ld; a.o_ld = o_ld;
  a.B = B; a.H = H; a.T = T; a.S = S; a.nb = pe ? nb : 0; a.maxrel = maxrel; a.causal = causal; a.lds = lds;
  a.scale = scale; a.dropout_p = dropout_p; a.seed = seed;
  dim3 grid((T + 127) / 128, B * H), block(256);
  const bool dr = dropout_p > 0.f;
  const size_t shm = (size_t)4 * fa2::TILE_B + (pe ? 4 * fa2::SCR_B : 0);
  if (pe && dr) hipLaunchKernelGGL((fa2::fwd_kernel<true, true>), grid, block, shm, s, a);
  else if (pe) hipLaunchKernelGGL((fa2::fwd_kernel<true, false>), grid, block, shm, s, a);
  else if (dr) hipLaunchKernelGGL((fa2::fwd_kernel<false, true>), grid, block, shm, s, a);
  else hipLaunchKernelGGL((fa2::fwd_kernel<false, false>), grid, block, shm, s, a);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_flash_attn_fwd(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld, void* o,
                                  int64_t o_ld, float* lse, const void* pe, const uint8_t* kpm, int32_t B, int32_t H, int32_t T, int32_t S,
                                  int32_t head_dim, int32_t nb, int32_t maxrel, int32_t causal, int32_t lds, float scale, float dropout_p,
                                  uint64_t seed, int dtype, void* stream) {
  return st5_flash_attn_fwd_qp(q, q_ld, k, k_ld, v, v_ld, o, o_ld, lse, pe, kpm, B, H, T, S, head_dim, nb, maxrel, causal, lds, scale,
                               dropout_p, seed, nullptr, dtype, stream);
}

extern "C" int st5_flash_attn_bwd_2s(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld, const void* o,
                                     int64_t o_ld, const void* dout, int64_t do_ld, void* dq, int64_t dq_ld, void* dk, int64_t dk_ld,
                                     void* dv, int64_t dv_ld, const float* lse, float* dvec, const void* pe, const void* qp, void* dqp,
                                     const uint8_t* kpm, int32_t B, int32_t H, int32_t T, int32_t S, int32_t head_dim, int32_t nb,
                                     int32_t maxrel, int32_t causal, int32_t lds, float scale, float dropout_p, uint64_t seed, int dtype,
                                     void* stream, void* stream2) {
  if (g_impl == 1)
    return st5_flash1_attn_bwd_2s(q, q_ld, k, k_ld, v, v_ld, o, o_ld, dout, do_ld, dq, dq_ld, dk, dk_ld, dv, dv_ld, lse, dvec, pe, qp, dqp,
                                  kpm, B, H, T, S, head_dim, nb, maxrel, causal, lds, scale, dropout_p, seed, dtype, stream, stream2);
  if (!q || !k || !v || !o || !dout || !dq || !dk || !dv || !lse || !dvec || B <= 0 || H <= 0 || T <= 0 || S <= 0) return ST5_ERR_ARG;
  if (dtype != ST5_BF16 || head_dim != fa2::HD) return ST5_ERR_ARG;
  if (q_ld % 8 || k_ld % 8 || v_ld % 8 || o_ld % 8 || do_ld % 8 || dq_ld % 4 || dk_ld % 4 || dv_ld % 4) return ST5_ERR_ALIGN;
  if (pe && (!qp || !dqp || nb != 2 * maxrel || nb % 8 || nb > 1024)) return ST5_ERR_ARG;
  if (pe && (reinterpret_cast<uintptr_t>(dqp) % 16) != 0) return ST5_ERR_ALIGN;   // the dq kernel clears dQP with 16-byte stores (rows are nb % 8 == 0 bf16 long)
  if ((long long)B * H * T * (pe ? nb + 16 : 1) >= (1ll << 31)) return ST5_ERR_ARG;
  if (set_attrs() != ST5_OK) return ST5_ERR_LAUNCH;
  fa2::BwdArgs a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o; a.dout = (const bf16_t*)dout;
  a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.lse = lse; a.dvec = dvec;
  a.qp = pe ? (const bf16_t*)qp : nullptr; a.dqp = (bf16_t*)dqp; a.kpm = kpm;
  a.q_ld = q_ld; a.k_ld = k_ld; a.v_ld = v_ld; a.o_ld = o_ld; a.do_ld = do_ld; a.dq_ld = dq_ld; a.dk_ld = dk_ld; a.dv_ld = dv_ld;
  a.B = B; a.H = H; a.T = T; a.S = S; a.nb = pe ? nb : 0; a.maxrel = maxrel; a.causal = causal; a.lds = lds;
  a.scale = scale; a.dropout_p = dropout_p; a.seed = seed;
  hipStream_t s = (hipStream_t)stream, s2 = (hipStream_t)stream2;
  const long long rows = (long long)B * H * T;
  // one stream (the default): D = dO . O is computed by the dq kernel, which writes dvec for the dkv kernel behind it, and the dq
  // kernel clears dQP; with the helper stream (dq and dkv side by side) both are needed up front
  a.dvec_in_dq = (s2 == nullptr || s2 == s) ? 1 : 0;
  if (pe && !a.dvec_in_dq && hipMemsetAsync(dqp, 0, (size_t)B * H * T * nb * 2, s) != hipSuccess) return ST5_ERR_LAUNCH;
  if (!a.dvec_in_dq)
    hipLaunchKernelGGL(fa2::dvec_kernel, dim3((unsigned)((rows * 2 + 255) / 256)), dim3(256), 0, s, a.o, a.dout, dvec, a.o_ld, a.do_ld, H, T, rows);
  if (s2) { if (st5_stream_fork(s, s2) != ST5_OK) return ST5_ERR_LAUNCH; } else s2 = s;
  const size_t shm_dq = (size_t)4 * fa2::TILE_B + (pe ? 4 * fa2::SCR_B : 0);
#ifdef FA2_DKV_FAT_LDS   // discriminator build (tools/r4/sbs_arms.sh): 104+ KB per dkv block -> one per CU, and no GEMM block beside it
  const size_t shm_dkv = (size_t)2 * fa2::QBUF_B + (pe ? 4 * fa2::KSCR_B : 0) + 48 * 1024;
#elif defined(FA2_DKV_HEAD_PAD)
  const size_t shm_dkv = (size_t)2 * fa2::QBUF_B + (pe ? 4 * fa2::KSCR_B : 0) + 16 * 1024;
#else
  const size_t shm_dkv = (size_t)2 * fa2::QBUF_B + (pe ? 4 * fa2::KSCR_B : 0);
#endif
  const dim3 gq((T + 127) / 128, B * H), gk((S + 127) / 128, B * H), blk(256);
  const bool dr = dropout_p > 0.f;
#define FA2_BWD(BIAS_, DROP_)                                                                        \
  do {                                                                                               \
    hipLaunchKernelGGL((fa2::bwd_dq_kernel<BIAS_, DROP_>), gq, blk, shm_dq, s, a);    \
    hipLaunchKernelGGL((fa2::bwd_dkv_kernel<BIAS_, DROP_>), gk, blk, shm_dkv, s2, a); \
  } while (0)
  if (pe && dr) FA2_BWD(true, true);
  else if (pe) FA2_BWD(true, false);
  else if (dr) FA2_BWD(false, true);
  else FA2_BWD(false, false);
#undef FA2_BWD
  HIP_CHECK_LAUNCH();
  if (s2 != s && st5_stream_fork(s2, s) != ST5_OK) return ST5_ERR_LAUNCH;
  return ST5_OK;
}

extern "C" int st5_flash_attn_bwd(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld, const void* o,
                                  int64_t o_ld, const void* dout, int64_t do_ld, void* dq, int64_t dq_ld, void* dk, int64_t dk_ld, void* dv,
                                  int64_t dv_ld, const float* lse, float* dvec, const void* pe, const void* qp, void* dqp,
                                  const uint8_t* kpm, int32_t B, int32_t H, int32_t T, int32_t S, int32_t head_dim, int32_t nb,
                                  int32_t maxrel, int32_t causal, int32_t lds, float scale, float dropout_p, uint64_t seed, int dtype,
                                  void* stream) {
  return st5_flash_attn_bwd_2s(q, q_ld, k, k_ld, v, v_ld, o, o_ld, dout, do_ld, dq, dq_ld, dk, dk_ld, dv, dv_ld, lse, dvec, pe, qp, dqp,
                               kpm, B, H, T, S, head_dim, nb, maxrel, causal, lds, scale, dropout_p, seed, dtype, stream, nullptr);
}
