// Element-wise, gather/scatter and small reduction kernels of the SpeechT5 hot path (HBM-bound;
// 16-byte vector IO, grid-stride).  Each replaces a torch call site cited in include/speecht5_hip.h.
#include <mutex>
#include "common.h"
#include "../../include/speecht5_hip.h"

namespace {

inline dim3 grid_for(long long nvec) {
  long long b = (nvec + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

// ---- casts ----
template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, long long n) {
  const long long nv = n / 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    float v[8];
    load8f<float>(src + i * 8, v);
    store8f<T>(dst + i * 8, v);
  }
  for (long long i = nv * 8 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    dst[i] = Elem<T>::from_f(src[i]);
}
// dst[c, r] = src[r, c] via a 32x32 LDS tile
template <typename T>
__global__ void cast_transpose_kernel(const float* __restrict__ src, T* __restrict__ dst, long long rows,
                                      long long cols) {
  __shared__ float tile[32][33];
  const long long r0 = (long long)blockIdx.y * 32, c0 = (long long)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const long long r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < rows && c < cols) ? src[r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const long long c = c0 + j, r = r0 + tx;
    if (c < cols && r < rows) dst[c * rows + r] = Elem<T>::from_f(tile[tx][j]);
  }
}
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, long long n) {
  const long long nv = n / 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    float v[8];
    load8f<T>(src + i * 8, v);
    store8f<float>(dst + i * 8, v);
  }
  for (long long i = nv * 8 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    dst[i] = Elem<T>::to_f(src[i]);
}

// ---- generic unary / binary element-wise with a functor on 8-vectors ----
template <typename T, typename F>
__global__ void map1_kernel(const T* __restrict__ x, T* __restrict__ y, long long n, F f) {
  const long long nv = n / 8;
  const unsigned int st = f.init();       // (loop-invariant state of the functor, computed once per thread: Drop's seed hash)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    float v[8];
    load8f<T>(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = f(v[e], i * 8 + e, st);
    store8f<T>(y + i * 8, v);
  }
  for (long long i = nv * 8 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    y[i] = Elem<T>::from_f(f(Elem<T>::to_f(x[i]), i, st));
}
template <typename T, typename F>
__global__ void map2_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long long n, F f) {
  const long long nv = n / 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    float u[8], v[8];
    load8f<T>(a + i * 8, u);
    load8f<T>(b + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] = f(u[e], v[e]);
    store8f<T>(y + i * 8, u);
  }
  for (long long i = nv * 8 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    y[i] = Elem<T>::from_f(f(Elem<T>::to_f(a[i]), Elem<T>::to_f(b[i])));
}

struct ActF {
  int act;
  __device__ unsigned int init() const { return 0u; }
  __device__ float operator()(float x, long long, unsigned int) const { return act_f(act, x); }
};
struct ActB { int act; __device__ float operator()(float dy, float x) const { return dy * act_grad_f(act, x); } };
struct Axpby { float a, b; __device__ float operator()(float x, float y) const { return a * x + b * y; } };
struct Drop {
  unsigned long long seed; unsigned int thresh; float inv_keep;
  __device__ unsigned int init() const { return drop_seed_fold(seed); }      // (dropout_scale's seed part: same bits, once instead of per element)
  __device__ float operator()(float x, long long i, unsigned int seed_fold) const {
    const unsigned int key = drop_block_key_folded(seed_fold, (unsigned long long)i >> 6);
    return x * drop_pick(drop_pair_bits(key, ((unsigned int)i & 63u) >> 1), (int)(i & 1), thresh, inv_keep);
  }
};

// ---- LayerDrop as a device-side select (a captured step cannot branch on the host draw; encoder.py:251-257, decoder.py:64-67) ----
// y = keep ? layer_out : layer_in, bytes moved untouched (16-byte chunks); backward: (g_in, g_out) = keep ? (0, g) : (g, 0).
// `keep` is ONE float in device memory, refreshed before every replay from the host draw of that layer.
__global__ __launch_bounds__(256) void select_fwd_kernel(const float* __restrict__ keep, const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                                         u32x4* __restrict__ y, long long nv) {
  const u32x4* __restrict__ src = keep[0] != 0.f ? b : a;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) y[i] = src[i];
}
__global__ __launch_bounds__(256) void select_bwd_kernel(const float* __restrict__ keep, const u32x4* __restrict__ g, u32x4* __restrict__ ga,
                                                         u32x4* __restrict__ gb, long long nv) {
  const bool k = keep[0] != 0.f;
  const u32x4 z = {0u, 0u, 0u, 0u};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    const u32x4 v = g[i];
    ga[i] = k ? z : v;
    gb[i] = k ? v : z;
  }
}
// The LayerDrop gate's backward at the layer input (functional.LayerDropEnterFunction): dx = keep ? dx : g, in place.  A kept layer
// (19 of 20) costs one scalar load per wave and no traffic; a dropped layer's input gradient (exact zeros everywhere, its last
// LayerNorm's backward was gated) is replaced by the gradient of the layer's OUTPUT.
__global__ __launch_bounds__(256) void skip_grad_kernel(const float* __restrict__ keep, const u32x4* __restrict__ g, u32x4* __restrict__ dx,
                                                        long long nv) {
  if (keep[0] != 0.f) return;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) dx[i] = g[i];
}

// ---- sum of squares ----
template <typename T>
__global__ void sumsq_kernel(const T* __restrict__ x, float* __restrict__ part, long long n) {
  __shared__ float red[4];
  float s = 0.f;
  const long long nv = n / 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    float v[8];
    load8f<T>(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(v[e], v[e], s);
  }
  for (long long i = nv * 8 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float v = Elem<T>::to_f(x[i]);
    s = fmaf(v, v, s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// sum of squares of x + y (fp32): the gradient norm over two gradient buffers.  Same partition and summation order as
// sumsq_kernel<float> (8 elements per thread and trip), so norm(x + y) comes out bit-identical whether the buffers were summed
// first (several ranks: all-reduce in between) or are summed here.
__global__ void sumsq_pair_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ part, long long n) {
  __shared__ float red[4];
  float s = 0.f;
  const long long nv = n / 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 a0 = reinterpret_cast<const f32x4*>(x)[2 * i], a1 = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
    const f32x4 b0 = reinterpret_cast<const f32x4*>(y)[2 * i], b1 = reinterpret_cast<const f32x4*>(y)[2 * i + 1];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float v = a0[e] + b0[e]; s = fmaf(v, v, s); }
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float v = a1[e] + b1[e]; s = fmaf(v, v, s); }
  }
  for (long long i = nv * 8 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i] + y[i];
    s = fmaf(v, v, s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int n, float scale,
                                 int accumulate) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)part[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float r = (float)((red[0] + red[1] + red[2] + red[3]) * (double)scale);
    out[0] = accumulate ? out[0] + r : r;
  }
}

// ---- row ops: one wave per row ----
template <typename T>
__global__ void masked_fill_rows_kernel(T* __restrict__ x, const uint8_t* __restrict__ mask,
                                        const float* __restrict__ v, long long rows, int cols) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows || !mask[row]) return;
  for (int c = threadIdx.x & 63; c < cols; c += 64) x[row * cols + c] = Elem<T>::from_f(v[c]);
}
// dv[c] += sum_{masked rows} dx[r,c]; dx[r,:] = 0 for masked rows.  grid (column blocks of 256, row splits):
// each thread owns one column and strides over rows; block partials go to part[blockIdx.y][c] and a second kernel adds them
// to dv in order (fp32 atomics here made the mask-embedding gradient differ from run to run in the last bits).
__global__ void masked_fill_final_kernel(const float* __restrict__ part, float* __restrict__ dv, int ny, int cols) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  int y = 0;
  for (; y + 8 <= ny; y += 8) {   // 8 independent loads in flight, summed in order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(long long)(y + u) * cols + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; y < ny; ++y) s += part[(long long)y * cols + c];
  dv[c] += s;
}
template <typename T>
__global__ void masked_fill_rows_bwd_kernel(T* __restrict__ dx, const uint8_t* __restrict__ mask,
                                            float* __restrict__ dv, long long rows, int cols) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (long long r = blockIdx.y; r < rows; r += gridDim.y) {
    if (mask[r]) {
      s += Elem<T>::to_f(dx[r * cols + c]);
      dx[r * cols + c] = Elem<T>::from_f(0.f);
    }
  }
  if (dv) dv[(long long)blockIdx.y * cols + c] = s;   // (dv = the partial buffer here)
}
template <typename T>
__global__ void add_table_rows_kernel(const T* __restrict__ x, const float* __restrict__ table,
                                      const int32_t* __restrict__ idx, T* __restrict__ y, long long rows, int cols,
                                      float scale, const float* __restrict__ scale_dev) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (scale_dev) scale *= scale_dev[0];
  const float* trow = table + (long long)idx[row] * cols;
  for (int c = (threadIdx.x & 63) * 8; c < cols; c += 512) {
    if (c + 8 <= cols) {
      float v[8], t[8];
      load8f<T>(x + row * cols + c, v);
      load8f<float>(trow + c, t);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(scale, t[e], v[e]);
      store8f<T>(y + row * cols + c, v);
    } else {
      for (int e = 0; c + e < cols; ++e)
        y[row * cols + c + e] = Elem<T>::from_f(Elem<T>::to_f(x[row * cols + c + e]) + scale * trow[c + e]);
    }
  }
}
template <typename T>
__global__ void embed_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ tok,
                                  const float* __restrict__ pos, const int32_t* __restrict__ pidx,
                                  T* __restrict__ y, long long rows, int cols, float emb_scale, float pos_scale) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* trow = table + (long long)tok[row] * cols;
  const float* prow = pos ? pos + (long long)pidx[row] * cols : nullptr;
  for (int c = threadIdx.x & 63; c < cols; c += 64) {
    float v = emb_scale * trow[c];
    if (prow) v = fmaf(pos_scale, prow[c], v);
    y[row * cols + c] = Elem<T>::from_f(v);
  }
}
template <typename T>
__global__ void embed_rows_bwd_kernel(const T* __restrict__ dy, const int32_t* __restrict__ tok,
                                      float* __restrict__ dtable, long long rows, int cols, float scale) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* trow = dtable + (long long)tok[row] * cols;
  for (int c = threadIdx.x & 63; c < cols; c += 64)
    atomicAdd(&trow[c], scale * Elem<T>::to_f(dy[row * cols + c]));
}
// Deterministic form (no atomics; a training step is then bit-reproducible, the atomic form above is not: fp32 atomics commit
// in arrival order).  Two kernels: (1) rank sort of the rows by (table row, position): every thread counts the rows that sort
// before its own -- N^2 / 64 wave-steps on ids staged through LDS, ~10 us for 8k rows; (2) segmented sums over the sorted
// order: the block in whose range a run of equal ids STARTS adds the whole run in position order and is the table row's only
// writer.
template <typename T> __device__ __forceinline__ void ld4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float (&v)[4]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = a[i];
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float (&v)[4]) {
  const uint2 a = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
}
constexpr int RS_CHUNK = 2048;
// block = RS_ROWS rows x RS_PARTS parts: part p counts inside its share of every staged chunk, the counts are added at the end.
// (Round 6: 32 rows x 8 parts instead of 64 x 4 -- 8192 tokens were 128 blocks on 256 CUs, each thread walking 2048 ids; now every CU has
//  a block and a thread walks 1024.  Same ranks.)
constexpr int RS_ROWS = 32, RS_PARTS = 256 / RS_ROWS;
__global__ __launch_bounds__(256) void rank_sort_kernel(const int32_t* __restrict__ tok, int32_t* __restrict__ order,
                                                        int32_t* __restrict__ sid, int n) {
  __shared__ __attribute__((aligned(16))) int32_t ids[RS_CHUNK];
  __shared__ int cnt[RS_PARTS][RS_ROWS];
  const int tl = threadIdx.x % RS_ROWS, part = threadIdx.x / RS_ROWS;
  const int t = blockIdx.x * RS_ROWS + tl;
  const int my = t < n ? tok[t] : 0x7fffffff;
  int rank = 0;
  for (int base = 0; base < n; base += RS_CHUNK) {
    const int m = n - base < RS_CHUNK ? n - base : RS_CHUNK;
    __syncthreads();
    for (int i = threadIdx.x; i < RS_CHUNK; i += 256) ids[i] = i < m ? tok[base + i] : 0x7fffffff;
    __syncthreads();
    // (slots past m hold INT_MAX and never count, so whole pieces are walked with 16-byte LDS reads: 4 ids per round trip)
    const int lo = part * (RS_CHUNK / RS_PARTS), hi = lo + RS_CHUNK / RS_PARTS;
    if (lo >= m) continue;
    const int4* v4 = reinterpret_cast<const int4*>(ids + lo);
    if (base + hi <= t) {                        // the whole piece lies before t: ties count
#pragma unroll 2
      for (int i = 0; i < RS_CHUNK / (4 * RS_PARTS); ++i) { const int4 v = v4[i]; rank += (v.x <= my) + (v.y <= my) + (v.z <= my) + (v.w <= my); }
    } else if (base + lo > t) {                  // the whole piece lies after t: ties do not count
#pragma unroll 2
      for (int i = 0; i < RS_CHUNK / (4 * RS_PARTS); ++i) { const int4 v = v4[i]; rank += (v.x < my) + (v.y < my) + (v.z < my) + (v.w < my); }
    } else {
      for (int i = lo; i < hi; ++i) rank += ids[i] < my || (ids[i] == my && base + i < t);
    }
  }
  cnt[part][tl] = rank;
  __syncthreads();
  if (part == 0 && t < n) {
    int r = 0;
#pragma unroll
    for (int p = 0; p < RS_PARTS; ++p) r += cnt[p][tl];
    order[r] = t; sid[r] = my;
  }
}
// Segmented sums over the sorted order, balanced for runs of any length.  The sorted entries are cut into chunks of SEG; block
// (chunk, 256-column slab) sums every PIECE (maximal range of equal ids inside the chunk) -- wave w takes pieces w, w + 4, ...,
// lane = four consecutive table columns, 8 row loads in flight, rows of a piece added in sorted (= position) order.
//   * a piece that is a whole run goes straight into the table (the run's only writer);
//   * a piece continuing the previous chunk's last run is stored in head[chunk], a piece continuing into the next chunk in
//     tail[chunk]; the combine kernel walks every multi-chunk run once, tail -> head -> head ..., in chunk order.
// No atomics, no timing dependence; a 10 000-row run (a frequent token, a dominant code-book entry) costs its blocks 32 rows each.
constexpr int SEG = 32;
template <typename T>
__global__ __launch_bounds__(256) void seg_pieces_kernel(const T* __restrict__ dy, const int32_t* __restrict__ order,
                                                         const int32_t* __restrict__ sid, float* __restrict__ dtable,
                                                         float* __restrict__ head, float* __restrict__ tail, long long rows, int cols,
                                                         int vocab, float scale, const float* __restrict__ rw, int rw_div, int rw_mod) {
  __shared__ int sh_id[SEG + 2];    // ids of sorted entries s0 - 1 .. s0 + SEG
  __shared__ int sh_r[SEG];
  __shared__ float sh_w[SEG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = (blockIdx.y * 64 + lane) * 4;
  const bool live = c < cols;
  const long long s0 = (long long)blockIdx.x * SEG;
  const int n_in = (int)(rows - s0 < SEG ? rows - s0 : SEG);
  if (tid < SEG + 2) {
    const long long e = s0 - 1 + tid;
    sh_id[tid] = (e >= 0 && e < rows) ? sid[e] : (int)0x80000000;
  }
  if (tid < SEG) {
    const int r = tid < n_in ? order[s0 + tid] : -1;
    sh_r[tid] = r;
    sh_w[tid] = (r >= 0 && rw) ? rw[(r / rw_div) % rw_mod] : 1.f;
  }
  __syncthreads();
  int piece = 0;
  for (int a = 0; a < n_in;) {
    const int v = sh_id[a + 1];
    int b = a + 1;
    while (b < n_in && sh_id[b + 1] == v) ++b;
    if ((piece++ & 3) == wave && v >= 0 && v < vocab) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int k0 = a; k0 < b; k0 += 8) {
        float x[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (k0 + u < b && live) ld4<T>(dy + (long long)sh_r[k0 + u] * cols + c, x[u]);
          else { x[u][0] = x[u][1] = x[u][2] = x[u][3] = 0.f; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float w = k0 + u < b ? sh_w[k0 + u] : 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = fmaf(w, x[u][e], acc[e]);
        }
      }
      if (live) {
        const bool from_prev = a == 0 && sh_id[0] == v, to_next = b == n_in && sh_id[n_in + 1] == v;
        float* dst;
        float k = 1.f;
        if (from_prev) dst = head + (long long)blockIdx.x * cols + c;
        else if (to_next) dst = tail + (long long)blockIdx.x * cols + c;
        else { dst = dtable + (long long)v * cols + c; k = scale; }
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = (from_prev || to_next) ? acc[e] : dst[e] + k * acc[e];
      }
    }
    a = b;
  }
}
// one thread per (run that starts in chunk ch and continues past it, column): tail[ch] + head[ch+1] + ... in chunk order
__global__ __launch_bounds__(256) void seg_combine_kernel(const int32_t* __restrict__ sid, float* __restrict__ dtable,
                                                          const float* __restrict__ head, const float* __restrict__ tail, long long rows,
                                                          int cols, int vocab, float scale) {
  const long long ch = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  const long long last = (ch + 1) * SEG - 1;
  if (last + 1 >= rows || c >= cols) return;            // no next chunk
  const int v = sid[last];
  if (sid[last + 1] != v || v < 0 || v >= vocab) return; // the chunk's last run ends here
  if (sid[ch * SEG] == v && ch > 0 && sid[ch * SEG - 1] == v) return;   // the whole chunk continues an earlier run: not the start
  float acc = tail[ch * cols + c];
  const long long nch = (rows + SEG - 1) / SEG;
  bool more = true;
  for (long long k = ch + 1; more; k += 8) {            // eight chunks' loads in flight; consumed strictly in chunk order
    float h[8];
    bool go[8];                                          // chunk k + u consists of this id only AND continues into the next one
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long kk = k + u, kl = (kk + 1) * SEG - 1;
      h[u] = kk < nch ? head[kk * cols + c] : 0.f;       // (chunks past the run's end are loaded but never added)
      go[u] = kk < nch && kl + 1 < rows && sid[kl] == v && sid[kl + 1] == v;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (more) { acc += h[u]; more = go[u]; }
    }
  }
  dtable[(long long)v * cols + c] += scale * acc;
}

// x[b, t, :] = 0 for t < head or t >= tail_start  (x [B, Tp, C]): the halo / uncovered rows of the convolution-gradient buffers
template <typename T>
__global__ __launch_bounds__(256) void zero_time_edges_kernel(T* __restrict__ x, int B, int Tp, int C, int head, int tail_start) {
  const int nedge = head + (Tp - tail_start);
  const long long nrow = (long long)B * nedge;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrow) return;
  const int b = (int)(row / nedge), e = (int)(row % nedge);
  const int t = e < head ? e : tail_start + (e - head);
  T* d = x + ((long long)b * Tp + t) * C;
  const T zero = Elem<T>::from_f(0.f);
  for (int c = threadIdx.x & 63; c < C; c += 64) d[c] = zero;
}

// dst[a, b, c] (+)= src[off + a*sa + b*sb + c*sc]: the weight re-layouts of the implicit-GEMM convolutions (permutes, tap
// flips -- negative strides -- and the cast to the compute dtype) and the re-laid-out accumulation of their weight gradients
// in one pass each (torch needs permute().contiguous() + a cast, or a strided add_).
template <typename T>
__global__ __launch_bounds__(256) void gather3_kernel(const float* __restrict__ src, T* __restrict__ dst, int A, int B, int C, long long sa,
                                                      long long sb, long long sc, long long off, int accumulate) {
  const long long n = (long long)A * B * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long ab = i / C;
    const int b = (int)(ab % B), a = (int)(ab / B);
    const float v = src[off + a * sa + b * sb + c * sc];
    dst[i] = Elem<T>::from_f(accumulate ? Elem<T>::to_f(dst[i]) + v : v);
  }
}
template <typename T>
__global__ void pad_time_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int Tn, int C, int pad_l,
                                int pad_r) {
  const int Tp = pad_l + Tn + pad_r;
  const long long nrow = (long long)B * Tp;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrow) return;
  const int b = (int)(row / Tp), t = (int)(row % Tp) - pad_l;
  const bool in = t >= 0 && t < Tn;
  const T* s = src + ((long long)b * Tn + (in ? t : 0)) * C;
  T* d = dst + row * C;
  const T zero = Elem<T>::from_f(0.f);
  for (int c = threadIdx.x & 63; c < C; c += 64) d[c] = in ? s[c] : zero;
}

// dst[b, t', :] = act(src[b, t' - pad_l, :]) inside, 0 in the halo (every activation used here maps 0 to 0, so this is the padded
// copy of act(src)); one 16-byte vector per thread -- with narrow rows (HiFi-GAN's late stages: 32 channels = 64 bytes per row) a
// wave-per-row copy moves 64 bytes per wave.  Requires C * sizeof(T) % 16 == 0.
template <typename T>
__global__ void pad_time_act_vec_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int Tn, int C, int pad_l, int pad_r,
                                        int act) {
  constexpr int V = Elem<T>::VEC;
  const int cv = C / V, Tp = pad_l + Tn + pad_r;
  const long long n = (long long)B * Tp * cv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / cv;
    const int c = (int)(i - row * cv) * V;
    const int b = (int)(row / Tp), t = (int)(row - (long long)b * Tp) - pad_l;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (t >= 0 && t < Tn) {
      v = *reinterpret_cast<const uint4*>(src + ((long long)b * Tn + t) * C + c);
      if (act != ACT_NONE) {
        T* e = reinterpret_cast<T*>(&v);
#pragma unroll
        for (int j = 0; j < V; ++j) e[j] = Elem<T>::from_f(act_f(act, Elem<T>::to_f(e[j])));
      }
    }
    *reinterpret_cast<uint4*>(dst + row * C + c) = v;
  }
}
// zero rows [0, pad_l) and [pad_l + Tn, Tp) of every batch element of a [B, Tp, C] buffer (the halo of a convolution output
// that its producer writes straight into the padded layout the next convolution reads)
template <typename T>
__global__ void zero_halo_kernel(T* __restrict__ dst, int B, int Tn, int C, int pad_l, int pad_r) {
  const int P = pad_l + pad_r, Tp = P + Tn;
  const long long n = (long long)B * P * C;
  const T zero = Elem<T>::from_f(0.f);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long r = i / C;
    const int b = (int)(r / P), h = (int)(r % P);
    const int t = h < pad_l ? h : Tn + h;
    dst[((long long)b * Tp + t) * C + c] = zero;
  }
}

template <typename T>
__global__ void channel_affine_kernel(const T* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                      T* __restrict__ y, long long rows, int cols, int act) {
  const long long n = rows * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols);
    y[i] = Elem<T>::from_f(act_f(act, fmaf(Elem<T>::to_f(x[i]), a[c], b[c])));
  }
}

// ---- cross entropy: one wave per row ----
template <typename T>
__global__ void cross_entropy_kernel(const T* __restrict__ logits, const int32_t* __restrict__ target,
                                     float* __restrict__ loss_sum, float* __restrict__ nll_sum,
                                     T* __restrict__ dlogits, long long rows, int V, long long ld, float eps,
                                     int ignore_index, float grad_scale, float* __restrict__ row_loss,
                                     float* __restrict__ row_nll) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* lrow = logits + row * ld;
  const int tgt = target[row];
  const bool skip = tgt < 0 || tgt == ignore_index;
  float mx = -INFINITY;
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, Elem<T>::to_f(lrow[c]));
  mx = wave_max(mx);
  float se = 0.f, sl = 0.f, nf = 0.f;   // nf: classes with a finite logit (-inf = excluded class: no smoothing mass either)
  for (int c = lane; c < V; c += 64) {
    const float l = Elem<T>::to_f(lrow[c]);
    if (l != -INFINITY) { se += __expf(l - mx); sl += l; nf += 1.f; }
  }
  se = wave_sum(se); sl = wave_sum(sl); nf = wave_sum(nf);
  const float lse = mx + __logf(se);
  // fairseq label_smoothed_nll_loss (speech_to_text_loss.py:93-110):
  //   nll = -lprob[tgt]; smooth = -sum_c lprob[c]; loss = (1-eps-eps_i)*nll + eps_i*smooth, eps_i = eps/(V-1)
  if (!skip) {
    const float lt = Elem<T>::to_f(lrow[tgt]);
    const float nll = lse - lt;
    const float eps_i = V > 1 ? eps / (float)(V - 1) : 0.f;
    const float smooth = nf * lse - sl;
    if (lane == 0) {
      const float l = (1.f - eps - eps_i) * nll + eps_i * smooth;
      if (row_loss) { row_loss[row] = l; if (row_nll) row_nll[row] = nll; }   // per-row outputs: deterministic caller-side sum
      else { atomicAdd(loss_sum, l); if (nll_sum) atomicAdd(nll_sum, nll); }
    }
  } else if (lane == 0 && row_loss) {
    row_loss[row] = 0.f;
    if (row_nll) row_nll[row] = 0.f;
  }
  if (dlogits) {
    T* drow = dlogits + row * ld;
    const float eps_i = V > 1 ? eps / (float)(V - 1) : 0.f;
    for (int c = lane; c < V; c += 64) {
      float g = 0.f;
      if (!skip) {
        const float l = Elem<T>::to_f(lrow[c]);
        const float p = l == -INFINITY ? 0.f : __expf(l - lse);
        // d/dl_c [(1-eps-eps_i)*(lse - l_t) + eps_i*(nf*lse - sum l)] = (1-eps-eps_i)(p - [c==t]) + eps_i (nf p - 1)
        g = (1.f - eps - eps_i) * (p - (c == tgt ? 1.f : 0.f)) + eps_i * (nf * p - 1.f);
        if (l == -INFINITY) g = 0.f;
      }
      drow[c] = Elem<T>::from_f(g * grad_scale);
    }
    for (int c = V + lane; c < ld; c += 64) drow[c] = Elem<T>::from_f(0.f);
  }
}

// ---- log-mel front end (speech_dataset.py:142-181): framing with librosa's centred reflect padding, |STFT|, log10 ----
// out[(b, l), j] = wav[b, reflect(l * hop + j - n_fft / 2)], reflect(i) = -i below 0, 2 (S - 1) - i from S on
__global__ void stft_frames_kernel(const float* __restrict__ wav, float* __restrict__ out, int S, int n_fft, int hop, int L,
                                   long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % n_fft);
    const long long row = i / n_fft;
    const int l = (int)(row % L);
    const long long b = row / L;
    int t = l * hop + j - n_fft / 2;
    t = t < 0 ? -t : t;
    t = t >= S ? 2 * (S - 1) - t : t;
    out[i] = wav[b * S + t];
  }
}
// reim [rows, 2 * ldh]: real parts in columns [0, nbins), imaginary parts in [ldh, ldh + nbins) -> mag [rows, ldh] (pad 0)
__global__ void stft_magnitude_kernel(const float* __restrict__ reim, float* __restrict__ mag, int nbins, int ldh, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % ldh);
    const long long row = i / ldh;
    float v = 0.f;
    if (n < nbins) {
      const float re = reim[row * 2 * ldh + n], im = reim[row * 2 * ldh + ldh + n];
      v = sqrtf(re * re + im * im);
    }
    mag[i] = v;
  }
}
__global__ void log10_floor_kernel(const float* __restrict__ x, float* __restrict__ y, float floor_, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = log10f(fmaxf(x[i], floor_));
}

float* g_scratch = nullptr;  // 2048-float device scratch for block partials (lazily allocated)
float* scratch() {
  if (!g_scratch) { if (st5_dev_malloc(&g_scratch, 2048 * sizeof(float)) != hipSuccess) return nullptr; }
  return g_scratch;
}


// out[(b, t), j] = wav[b, t*stride + j] for j < k, 0 for k <= j < kpad: the windows of a Cin = 1 convolution as GEMM rows
template <typename T>
__global__ void unfold_rows_kernel(const float* __restrict__ wav, T* __restrict__ out, int B, int S, int L, int k, int stride,
                                   int kpad) {
  const long long n = (long long)B * L * kpad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % kpad);
    const long long row = i / kpad;
    const int t = (int)(row % L), b = (int)(row / L);
    out[i] = Elem<T>::from_f(j < k ? wav[(long long)b * S + (long long)t * stride + j] : 0.f);
  }
}

}  // namespace

#define DISPATCH(dtype, CALL_BF, CALL_F)   \
  if (dtype == ST5_BF16) { CALL_BF; }      \
  else if (dtype == ST5_F32) { CALL_F; }   \
  else return ST5_ERR_ARG;

extern "C" int st5_cast_from_f32(const float* src, void* dst, int64_t rows, int64_t cols, int32_t transpose, int dtype,
                                 void* stream) {
  if (!src || !dst || rows < 0 || cols < 0) return ST5_ERR_ARG;
  if (rows * cols == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  if (!transpose) {
    const long long n = rows * cols;
    DISPATCH(dtype, hipLaunchKernelGGL(cast_from_f32_kernel<bf16_t>, grid_for(n / 8 + 1), dim3(256), 0, s, src, (bf16_t*)dst, n),
             hipLaunchKernelGGL(cast_from_f32_kernel<float>, grid_for(n / 8 + 1), dim3(256), 0, s, src, (float*)dst, n));
  } else {
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    DISPATCH(dtype, hipLaunchKernelGGL(cast_transpose_kernel<bf16_t>, grid, dim3(256), 0, s, src, (bf16_t*)dst, (long long)rows, (long long)cols),
             hipLaunchKernelGGL(cast_transpose_kernel<float>, grid, dim3(256), 0, s, src, (float*)dst, (long long)rows, (long long)cols));
  }
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_cast_to_f32(const void* src, float* dst, int64_t n, int dtype, void* stream) {
  if (!src || !dst || n < 0) return ST5_ERR_ARG;
  if (n == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH(dtype, hipLaunchKernelGGL(cast_to_f32_kernel<bf16_t>, grid_for(n / 8 + 1), dim3(256), 0, s, (const bf16_t*)src, dst, (long long)n),
           hipLaunchKernelGGL(cast_to_f32_kernel<float>, grid_for(n / 8 + 1), dim3(256), 0, s, (const float*)src, dst, (long long)n));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_sumsq(const void* x, float* out, int64_t n, float scale, int32_t accumulate, int dtype, void* stream) {
  if (!x || !out || n < 0) return ST5_ERR_ARG;
  float* part = scratch();
  if (!part) return ST5_ERR_LAUNCH;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid = grid_for(n / 8 + 1);
  DISPATCH(dtype, hipLaunchKernelGGL(sumsq_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, part, (long long)n),
           hipLaunchKernelGGL(sumsq_kernel<float>, grid, dim3(256), 0, s, (const float*)x, part, (long long)n));
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, s, part, out, (int)grid.x, scale, accumulate);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_sumsq_pair(const float* x, const float* y, float* out, int64_t n, float scale, int32_t accumulate, void* stream) {
  if (!x || !y || !out || n < 0) return ST5_ERR_ARG;
  float* part = scratch();
  if (!part) return ST5_ERR_LAUNCH;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid = grid_for(n / 8 + 1);
  hipLaunchKernelGGL(sumsq_pair_kernel, grid, dim3(256), 0, s, x, y, part, (long long)n);
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, s, part, out, (int)grid.x, scale, accumulate);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_axpby(const void* x, void* y, int64_t n, float a, float b, int dtype, void* stream) {
  if (!x || !y || n < 0) return ST5_ERR_ARG;
  if (n == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  Axpby f{a, b};
  DISPATCH(dtype, hipLaunchKernelGGL((map2_kernel<bf16_t, Axpby>), grid_for(n / 8 + 1), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)y, (bf16_t*)y, (long long)n, f),
           hipLaunchKernelGGL((map2_kernel<float, Axpby>), grid_for(n / 8 + 1), dim3(256), 0, s, (const float*)x, (const float*)y, (float*)y, (long long)n, f));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_select(const float* keep_dev, const void* a, const void* b, void* y, int64_t nbytes, void* stream) {
  if (!keep_dev || !a || !b || !y || nbytes < 0 || nbytes % 16) return ST5_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y)) % 16) return ST5_ERR_ALIGN;
  if (nbytes == 0) return ST5_OK;
  hipLaunchKernelGGL(select_fwd_kernel, grid_for(nbytes / 16), dim3(256), 0, (hipStream_t)stream, keep_dev, (const u32x4*)a, (const u32x4*)b,
                     (u32x4*)y, (long long)(nbytes / 16));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_select_bwd(const float* keep_dev, const void* g, void* ga, void* gb, int64_t nbytes, void* stream) {
  if (!keep_dev || !g || !ga || !gb || nbytes < 0 || nbytes % 16) return ST5_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(ga) | reinterpret_cast<uintptr_t>(gb)) % 16) return ST5_ERR_ALIGN;
  if (nbytes == 0) return ST5_OK;
  hipLaunchKernelGGL(select_bwd_kernel, grid_for(nbytes / 16), dim3(256), 0, (hipStream_t)stream, keep_dev, (const u32x4*)g, (u32x4*)ga,
                     (u32x4*)gb, (long long)(nbytes / 16));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_skip_grad(const float* keep_dev, const void* g, void* dx, int64_t nbytes, void* stream) {
  if (!keep_dev || !g || !dx || nbytes < 0 || nbytes % 16) return ST5_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dx)) % 16) return ST5_ERR_ALIGN;
  if (nbytes == 0) return ST5_OK;
  hipLaunchKernelGGL(skip_grad_kernel, grid_for(nbytes / 16), dim3(256), 0, (hipStream_t)stream, keep_dev, (const u32x4*)g, (u32x4*)dx,
                     (long long)(nbytes / 16));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_act_fwd(const void* x, void* y, int64_t n, int32_t act, int dtype, void* stream) {
  if (!x || !y || n < 0) return ST5_ERR_ARG;
  if (n == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  ActF f{act};
  DISPATCH(dtype, hipLaunchKernelGGL((map1_kernel<bf16_t, ActF>), grid_for(n / 8 + 1), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, (long long)n, f),
           hipLaunchKernelGGL((map1_kernel<float, ActF>), grid_for(n / 8 + 1), dim3(256), 0, s, (const float*)x, (float*)y, (long long)n, f));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_act_bwd(const void* dy, const void* x, void* dx, int64_t n, int32_t act, int dtype, void* stream) {
  if (!dy || !x || !dx || n < 0) return ST5_ERR_ARG;
  if (n == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  ActB f{act};
  DISPATCH(dtype, hipLaunchKernelGGL((map2_kernel<bf16_t, ActB>), grid_for(n / 8 + 1), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dx, (long long)n, f),
           hipLaunchKernelGGL((map2_kernel<float, ActB>), grid_for(n / 8 + 1), dim3(256), 0, s, (const float*)dy, (const float*)x, (float*)dx, (long long)n, f));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_channel_affine(const void* x, const float* a, const float* b, void* y, int64_t rows, int32_t cols,
                                  int32_t act, int dtype, void* stream) {
  if (!x || !a || !b || !y || rows < 0 || cols <= 0) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH(dtype, hipLaunchKernelGGL(channel_affine_kernel<bf16_t>, grid_for(rows * cols), dim3(256), 0, s, (const bf16_t*)x, a, b, (bf16_t*)y, (long long)rows, cols, act),
           hipLaunchKernelGGL(channel_affine_kernel<float>, grid_for(rows * cols), dim3(256), 0, s, (const float*)x, a, b, (float*)y, (long long)rows, cols, act));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream) {
  if (!x || !y || n < 0 || p < 0.f || p >= 1.f) return ST5_ERR_ARG;
  if (n == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  Drop f{(unsigned long long)seed, (unsigned int)(p * 65536.0f), 1.f / (1.f - p)};
  DISPATCH(dtype, hipLaunchKernelGGL((map1_kernel<bf16_t, Drop>), grid_for(n / 8 + 1), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, (long long)n, f),
           hipLaunchKernelGGL((map1_kernel<float, Drop>), grid_for(n / 8 + 1), dim3(256), 0, s, (const float*)x, (float*)y, (long long)n, f));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_masked_fill_rows(void* x, const uint8_t* mask, const float* v, int64_t rows, int32_t cols, int dtype,
                                    void* stream) {
  if (!x || !mask || !v || rows < 0 || cols <= 0) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  DISPATCH(dtype, hipLaunchKernelGGL(masked_fill_rows_kernel<bf16_t>, grid, dim3(256), 0, s, (bf16_t*)x, mask, v, (long long)rows, cols),
           hipLaunchKernelGGL(masked_fill_rows_kernel<float>, grid, dim3(256), 0, s, (float*)x, mask, v, (long long)rows, cols));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_masked_fill_rows_bwd(void* dx, const uint8_t* mask, float* dv, int64_t rows, int32_t cols, int dtype,
                                        void* stream) {
  if (!dx || !mask || rows < 0 || cols <= 0) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  const unsigned ny = (unsigned)(rows < 256 ? rows : 256);
  dim3 grid((unsigned)((cols + 255) / 256), ny);
  float* part = nullptr;
  if (dv) {   // grow-only partial buffer [256][cols] PER STREAM (first use / growth happens outside stream capture): one buffer for the
    // whole process would be shared by two micro-batches whose backward passes run side by side on two streams
    struct Part { hipStream_t s; float* p; size_t n; };
    static Part g_parts[16] = {};
    static int g_nparts = 0;
    static std::mutex g_mu;
    std::lock_guard<std::mutex> lk(g_mu);
    Part* e = nullptr;
    for (int i = 0; i < g_nparts; ++i)
      if (g_parts[i].s == s) e = &g_parts[i];
    if (!e) {
      if (g_nparts == 16) {      // (more streams than slots: recycle the first slot; hipFree waits for the device)
        if (g_parts[0].p) (void)hipFree(g_parts[0].p);
        e = &g_parts[0];
      } else e = &g_parts[g_nparts++];
      e->s = s; e->p = nullptr; e->n = 0;
    }
    const size_t want = (size_t)256 * cols;
    if (want > e->n) {
      if (e->p) (void)hipFree(e->p);
      e->p = nullptr; e->n = 0;
      if (st5_dev_malloc(&e->p, want * sizeof(float)) != hipSuccess) return ST5_ERR_LAUNCH;
      e->n = want;
    }
    part = e->p;
  }
  DISPATCH(dtype, hipLaunchKernelGGL(masked_fill_rows_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (bf16_t*)dx, mask, part, (long long)rows, cols),
           hipLaunchKernelGGL(masked_fill_rows_bwd_kernel<float>, grid, dim3(256), 0, s, (float*)dx, mask, part, (long long)rows, cols));
  if (dv) hipLaunchKernelGGL(masked_fill_final_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s, part, dv, (int)ny, cols);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_add_table_rows(const void* x, const float* table, const int32_t* idx, void* y, int64_t rows,
                                  int32_t cols, float scale, int dtype, void* stream) {
  if (!x || !table || !idx || !y || rows < 0 || cols <= 0 || cols % 8) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  DISPATCH(dtype, hipLaunchKernelGGL(add_table_rows_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, table, idx, (bf16_t*)y, (long long)rows, cols, scale, (const float*)nullptr),
           hipLaunchKernelGGL(add_table_rows_kernel<float>, grid, dim3(256), 0, s, (const float*)x, table, idx, (float*)y, (long long)rows, cols, scale, (const float*)nullptr));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_add_table_rows_dev(const void* x, const float* table, const int32_t* idx, void* y, int64_t rows,
                                      int32_t cols, const float* scale_dev, int dtype, void* stream) {
  if (!x || !table || !idx || !y || !scale_dev || rows < 0 || cols <= 0 || cols % 8) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  DISPATCH(dtype, hipLaunchKernelGGL(add_table_rows_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, table, idx, (bf16_t*)y, (long long)rows, cols, 1.f, scale_dev),
           hipLaunchKernelGGL(add_table_rows_kernel<float>, grid, dim3(256), 0, s, (const float*)x, table, idx, (float*)y, (long long)rows, cols, 1.f, scale_dev));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_embed_rows(const float* table, const int32_t* tok, const float* pos, const int32_t* pidx, void* y,
                              int64_t rows, int32_t cols, float emb_scale, float pos_scale, int dtype, void* stream) {
  if (!table || !tok || !y || rows < 0 || cols <= 0 || (pos && !pidx)) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  DISPATCH(dtype, hipLaunchKernelGGL(embed_rows_kernel<bf16_t>, grid, dim3(256), 0, s, table, tok, pos, pidx, (bf16_t*)y, (long long)rows, cols, emb_scale, pos_scale),
           hipLaunchKernelGGL(embed_rows_kernel<float>, grid, dim3(256), 0, s, table, tok, pos, pidx, (float*)y, (long long)rows, cols, emb_scale, pos_scale));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_embed_rows_bwd(const void* dy, const int32_t* tok, float* dtable, int64_t rows, int32_t cols,
                                  float scale, int dtype, void* stream) {
  if (!dy || !tok || !dtable || rows < 0 || cols <= 0) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  DISPATCH(dtype, hipLaunchKernelGGL(embed_rows_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dy, tok, dtable, (long long)rows, cols, scale),
           hipLaunchKernelGGL(embed_rows_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, tok, dtable, (long long)rows, cols, scale));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
// workspace of the deterministic row scatter: order | sorted ids | head | tail (grow-only), one per stream
struct ScatterWs { hipStream_t stream; char* ptr; size_t bytes; };
ScatterWs g_scatter[4] = {};
int g_nscatter = 0;
extern "C" int st5_embed_rows_bwd_det_w(const void* dy, const int32_t* tok, float* dtable, int64_t rows, int32_t cols, int32_t vocab,
                                        float scale, const float* row_w, int32_t rw_div, int32_t rw_mod, int dtype, void* stream) {
  if (!dy || !tok || !dtable || rows < 0 || cols <= 0 || cols % 4 || vocab <= 0 || (row_w && (rw_div <= 0 || rw_mod <= 0))) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  if (rows > (1ll << 22)) return ST5_ERR_ARG;   // the rank sort is quadratic: meant for token / frame counts of one micro-batch
  hipStream_t s = (hipStream_t)stream;
  const long long nch = (rows + SEG - 1) / SEG;
  const size_t ints = (((size_t)rows * 2 * sizeof(int32_t)) + 255) & ~(size_t)255;
  const size_t need = ints + (size_t)2 * nch * cols * sizeof(float);
  ScatterWs* w = nullptr;
  for (int i = 0; i < g_nscatter; ++i)
    if (g_scatter[i].stream == s) w = &g_scatter[i];
  if (!w) {
    static int victim = 0;
    w = g_nscatter < 4 ? &g_scatter[g_nscatter++] : &g_scatter[victim++ % 4];   // (stream churn: recycle, the old owner is gone)
    if (w->stream != s && w->ptr && hipDeviceSynchronize() != hipSuccess) return ST5_ERR_LAUNCH;
    w->stream = s;
  }
  if (need > w->bytes) {   // (first use / growth: outside stream capture, like every other workspace of this library)
    if (w->ptr) (void)hipFree(w->ptr);
    w->ptr = nullptr; w->bytes = 0;
    const size_t want = need < (size_t(8) << 20) ? (size_t(8) << 20) : need;
    if (st5_dev_malloc(&w->ptr, want) != hipSuccess) return ST5_ERR_LAUNCH;
    w->bytes = want;
  }
  char* g_scatter_ws = w->ptr;
  int32_t* order = reinterpret_cast<int32_t*>(g_scatter_ws);
  int32_t* sid = order + rows;
  float* head = reinterpret_cast<float*>(g_scatter_ws + ints);
  float* tail = head + (size_t)nch * cols;
  hipLaunchKernelGGL(rank_sort_kernel, dim3((unsigned)((rows + RS_ROWS - 1) / RS_ROWS)), dim3(256), 0, s, tok, order, sid, (int)rows);
  dim3 grid((unsigned)nch, (unsigned)((cols + 255) / 256));
  DISPATCH(dtype, hipLaunchKernelGGL(seg_pieces_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dy, order, sid, dtable, head, tail, (long long)rows, cols, vocab, scale, row_w, rw_div, rw_mod),
           hipLaunchKernelGGL(seg_pieces_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, order, sid, dtable, head, tail, (long long)rows, cols, vocab, scale, row_w, rw_div, rw_mod));
  hipLaunchKernelGGL(seg_combine_kernel, grid, dim3(256), 0, s, sid, dtable, head, tail, (long long)rows, cols, vocab, scale);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_embed_rows_bwd_det(const void* dy, const int32_t* tok, float* dtable, int64_t rows, int32_t cols, int32_t vocab,
                                      float scale, int dtype, void* stream) {
  return st5_embed_rows_bwd_det_w(dy, tok, dtable, rows, cols, vocab, scale, nullptr, 1, 1, dtype, stream);
}
extern "C" int st5_zero_time_edges(void* x, int32_t B, int32_t Tp, int32_t C, int32_t head, int32_t tail_start, int dtype, void* stream) {
  if (!x || B <= 0 || Tp <= 0 || C <= 0 || head < 0 || tail_start < head || tail_start > Tp) return ST5_ERR_ARG;
  const long long nrow = (long long)B * (head + Tp - tail_start);
  if (nrow == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((nrow + 3) / 4));
  DISPATCH(dtype, hipLaunchKernelGGL(zero_time_edges_kernel<bf16_t>, grid, dim3(256), 0, s, (bf16_t*)x, B, Tp, C, head, tail_start),
           hipLaunchKernelGGL(zero_time_edges_kernel<float>, grid, dim3(256), 0, s, (float*)x, B, Tp, C, head, tail_start));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_gather3(const float* src, void* dst, int32_t A, int32_t B, int32_t C, int64_t sa, int64_t sb, int64_t sc, int64_t off,
                           int32_t accumulate, int dtype, void* stream) {
  if (!src || !dst || A <= 0 || B <= 0 || C <= 0) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long n = (long long)A * B * C;
  DISPATCH(dtype, hipLaunchKernelGGL(gather3_kernel<bf16_t>, grid_for(n), dim3(256), 0, s, src, (bf16_t*)dst, A, B, C, (long long)sa, (long long)sb, (long long)sc, (long long)off, accumulate),
           hipLaunchKernelGGL(gather3_kernel<float>, grid_for(n), dim3(256), 0, s, src, (float*)dst, A, B, C, (long long)sa, (long long)sb, (long long)sc, (long long)off, accumulate));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_pad_time(const void* src, void* dst, int32_t B, int32_t T, int32_t C, int32_t pad_l, int32_t pad_r,
                            int dtype, void* stream) {
  if (!src || !dst || B <= 0 || T <= 0 || C <= 0 || pad_l < 0 || pad_r < 0) return ST5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long nrow = (long long)B * (pad_l + T + pad_r);
  dim3 grid((unsigned)((nrow + 3) / 4));
  DISPATCH(dtype, hipLaunchKernelGGL(pad_time_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, B, T, C, pad_l, pad_r),
           hipLaunchKernelGGL(pad_time_kernel<float>, grid, dim3(256), 0, s, (const float*)src, (float*)dst, B, T, C, pad_l, pad_r));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_pad_time_act(const void* src, void* dst, int32_t B, int32_t T, int32_t C, int32_t pad_l, int32_t pad_r,
                                int32_t act, int dtype, void* stream) {
  if (!src || !dst || B <= 0 || T <= 0 || C <= 0 || pad_l < 0 || pad_r < 0) return ST5_ERR_ARG;
  const int esz = dtype == ST5_BF16 ? 2 : 4;
  if ((C * esz) % 16 != 0 || ((uintptr_t)src | (uintptr_t)dst) % 16 != 0) {   // odd rows: the row-per-wave copy, then the activation in place
    const int rc = st5_pad_time(src, dst, B, T, C, pad_l, pad_r, dtype, stream);
    if (rc != ST5_OK || act == ACT_NONE) return rc;
    return st5_act_fwd(dst, dst, (int64_t)B * (pad_l + T + pad_r) * C, act, dtype, stream);
  }
  hipStream_t s = (hipStream_t)stream;
  const long long n = (long long)B * (pad_l + T + pad_r) * (C * esz / 16);
  DISPATCH(dtype, hipLaunchKernelGGL(pad_time_act_vec_kernel<bf16_t>, grid_for(n), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, B, T, C, pad_l, pad_r, act),
           hipLaunchKernelGGL(pad_time_act_vec_kernel<float>, grid_for(n), dim3(256), 0, s, (const float*)src, (float*)dst, B, T, C, pad_l, pad_r, act));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_zero_halo(void* dst, int32_t B, int32_t T, int32_t C, int32_t pad_l, int32_t pad_r, int dtype, void* stream) {
  if (!dst || B <= 0 || T <= 0 || C <= 0 || pad_l < 0 || pad_r < 0) return ST5_ERR_ARG;
  if (pad_l + pad_r == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  const long long n = (long long)B * (pad_l + pad_r) * C;
  DISPATCH(dtype, hipLaunchKernelGGL(zero_halo_kernel<bf16_t>, grid_for(n), dim3(256), 0, s, (bf16_t*)dst, B, T, C, pad_l, pad_r),
           hipLaunchKernelGGL(zero_halo_kernel<float>, grid_for(n), dim3(256), 0, s, (float*)dst, B, T, C, pad_l, pad_r));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_unfold_rows(const float* wav, void* out, int32_t B, int32_t S, int32_t k, int32_t stride, int32_t kpad,
                               int dtype, void* stream) {
  if (!wav || !out || B <= 0 || k < 1 || stride < 1 || kpad < k || S < k) return ST5_ERR_ARG;
  const int L = (S - k) / stride + 1;
  hipStream_t s = (hipStream_t)stream;
  const long long n = (long long)B * L * kpad;
  DISPATCH(dtype, hipLaunchKernelGGL(unfold_rows_kernel<bf16_t>, grid_for(n), dim3(256), 0, s, wav, (bf16_t*)out, B, S, L, k, stride, kpad),
           hipLaunchKernelGGL(unfold_rows_kernel<float>, grid_for(n), dim3(256), 0, s, wav, (float*)out, B, S, L, k, stride, kpad));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_cross_entropy(const void* logits, const int32_t* target, float* loss_sum, float* nll_sum,
                                 void* dlogits, int64_t rows, int32_t V, int64_t ld, float label_smoothing,
                                 int32_t ignore_index, float grad_scale, int dtype, void* stream) {
  if (!logits || !target || !loss_sum || rows < 0 || V <= 0 || ld < V) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  DISPATCH(dtype, hipLaunchKernelGGL(cross_entropy_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)logits, target, loss_sum, nll_sum, (bf16_t*)dlogits, (long long)rows, V, (long long)ld, label_smoothing, ignore_index, grad_scale, (float*)nullptr, (float*)nullptr),
           hipLaunchKernelGGL(cross_entropy_kernel<float>, grid, dim3(256), 0, s, (const float*)logits, target, loss_sum, nll_sum, (float*)dlogits, (long long)rows, V, (long long)ld, label_smoothing, ignore_index, grad_scale, (float*)nullptr, (float*)nullptr));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_cross_entropy_rows(const void* logits, const int32_t* target, float* row_loss, float* row_nll,
                                      void* dlogits, int64_t rows, int32_t V, int64_t ld, float label_smoothing,
                                      int32_t ignore_index, float grad_scale, int dtype, void* stream) {
  if (!logits || !target || !row_loss || rows < 0 || V <= 0 || ld < V) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  DISPATCH(dtype, hipLaunchKernelGGL(cross_entropy_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)logits, target, (float*)nullptr, (float*)nullptr, (bf16_t*)dlogits, (long long)rows, V, (long long)ld, label_smoothing, ignore_index, grad_scale, row_loss, row_nll),
           hipLaunchKernelGGL(cross_entropy_kernel<float>, grid, dim3(256), 0, s, (const float*)logits, target, (float*)nullptr, (float*)nullptr, (float*)dlogits, (long long)rows, V, (long long)ld, label_smoothing, ignore_index, grad_scale, row_loss, row_nll));
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_stft_frames(const float* wav, float* out, int32_t B, int32_t S, int32_t n_fft, int32_t hop, void* stream) {
  if (!wav || !out || B <= 0 || n_fft <= 0 || n_fft % 2 || hop <= 0 || S <= n_fft / 2) return ST5_ERR_ARG;
  const int L = 1 + S / hop;
  const long long total = (long long)B * L * n_fft;
  hipLaunchKernelGGL(stft_frames_kernel, grid_for(total), dim3(256), 0, (hipStream_t)stream, wav, out, S, n_fft, hop, L, total);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_stft_magnitude(const float* reim, float* mag, int64_t rows, int32_t nbins, int32_t ldh, void* stream) {
  if (!reim || !mag || rows < 0 || nbins <= 0 || ldh < nbins) return ST5_ERR_ARG;
  if (rows == 0) return ST5_OK;
  const long long total = (long long)rows * ldh;
  hipLaunchKernelGGL(stft_magnitude_kernel, grid_for(total), dim3(256), 0, (hipStream_t)stream, reim, mag, nbins, ldh, total);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
extern "C" int st5_log10_floor(const float* x, float* y, int64_t n, float floor_value, void* stream) {
  if (!x || !y || n < 0 || !(floor_value > 0.f)) return ST5_ERR_ARG;
  if (n == 0) return ST5_OK;
  hipLaunchKernelGGL(log10_floor_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, x, y, floor_value, (long long)n);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" const char* st5_version(void) { return "speecht5_hip 0.1 (gfx950)"; }

// ---- collation of the speech-pretraining batch (speech_dataset.py:302-446): ragged gather + tail masks -------------------------
namespace {
template <typename U>
__global__ void ragged_rows_kernel(const void* const* __restrict__ src, const int32_t* __restrict__ off, const int32_t* __restrict__ hi,
                                   U* __restrict__ out, int T, int w, int step, int tmin, U pad, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i % w);
    const long long row = i / w;
    const int t = (int)(row % T), b = (int)(row / T);
    const long long j = (long long)off[b] + (long long)t * step;
    U v = pad;
    if (t >= tmin && j >= 0 && j < (long long)hi[b]) v = reinterpret_cast<const U*>(src[b])[j * w + e];
    out[i] = v;
  }
}
__global__ void tail_mask_kernel(const int32_t* __restrict__ n, uint8_t* __restrict__ out8, float* __restrict__ outf, int T, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T), b = (int)(i / T);
    const bool m = t >= n[b];
    if (out8) out8[i] = m ? 1 : 0; else outf[i] = m ? 1.f : 0.f;
  }
}
}  // namespace

extern "C" int st5_ragged_rows(const void* const* src, const int32_t* off, const int32_t* hi, void* out, int32_t B, int32_t T, int32_t w,
                               int32_t step, int32_t tmin, int32_t es, uint64_t pad_bits, void* stream) {
  if (!src || !off || !hi || !out || B <= 0 || T < 0 || w <= 0 || step <= 0 || (es != 1 && es != 4 && es != 8)) return ST5_ERR_ARG;
  const long long total = (long long)B * T * w;
  if (total == 0) return ST5_OK;
  long long blocks = (total + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipStream_t s = (hipStream_t)stream;
  if (es == 4) hipLaunchKernelGGL(ragged_rows_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, s, src, off, hi, (uint32_t*)out, T, w, step, tmin, (uint32_t)pad_bits, total);
  else if (es == 8) hipLaunchKernelGGL(ragged_rows_kernel<uint64_t>, dim3((unsigned)blocks), dim3(256), 0, s, src, off, hi, (uint64_t*)out, T, w, step, tmin, (uint64_t)pad_bits, total);
  else hipLaunchKernelGGL(ragged_rows_kernel<uint8_t>, dim3((unsigned)blocks), dim3(256), 0, s, src, off, hi, (uint8_t*)out, T, w, step, tmin, (uint8_t)pad_bits, total);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

extern "C" int st5_tail_mask(const int32_t* n, void* out, int32_t B, int32_t T, int32_t dtype_f32, void* stream) {
  if (!n || !out || B <= 0 || T < 0) return ST5_ERR_ARG;
  const long long total = (long long)B * T;
  if (total == 0) return ST5_OK;
  long long blocks = (total + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(tail_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, dtype_f32 ? nullptr : (uint8_t*)out,
                     dtype_f32 ? (float*)out : nullptr, T, total);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
