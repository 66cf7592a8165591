// Stand-alone reproducer for the claim of DESIGN.md section 4a (VERDICT r3 "missing" item 6): on MI355X / ROCm 7.2 a wave that
// executes PACKED fp32 VALU ops (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) returns stale lanes when waves of an MFMA-heavy
// kernel launched on ANOTHER stream share its SIMD.  No code of libspeecht5_hip.so is involved: two ~50-line kernels.
//
//   victim<PACKED>   one wave per block, ~88 VGPRs, a long loop  t = x * y ; acc += t  on eight register pairs, either as
//                    v_pk_mul_f32 + v_pk_add_f32 (PACKED) or as the same arithmetic in v_mul_f32 + v_add_f32.  All values are
//                    small integers, so every result is exact in fp32 and the expected sums are computed on the host.
//   aggressor<PAD>   4 waves per block, 2 blocks per CU (launch bounds), ~184 VGPRs: a loop of v_mfma_f32_32x32x16_bf16 on eight
//                    accumulators with LDS-fed operands -- the register / pipe profile of the 128x128 GEMM kernels.  PAD: the
//                    kernel declares 256 VGPRs (the mitigation the library ships: no third wave fits beside two of them).
// Schedule: the aggressor runs back to back on stream B while the victim grid is launched R times on stream A; every victim
// lane's eight sums are checked.  Output: one JSON line with, per (victim form, aggressor form), launches, wrong lanes, a
// histogram of the wrong lanes' positions inside their wave (16-lane quarters) and the first few mismatches.
// Build: hipcc --offload-arch=gfx950 -O2 tools/hazard/pk_hazard.hip -o tools/hazard/pk_hazard.bin   (__graft_entry__.build())
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

constexpr int NPAIR = 8;

template <bool PACKED>
__global__ __launch_bounds__(64) void victim(float* __restrict__ out, int iters) {
  asm volatile("; victim: allocation of 88 VGPRs" ::: "v87");
  const int lane = threadIdx.x;
  f2 acc[NPAIR], x[NPAIR];
#pragma unroll
  for (int p = 0; p < NPAIR; ++p) {
    acc[p] = f2{0.f, 0.f};
    x[p] = f2{(float)((lane + p) & 7), (float)((lane * 3 + p) & 7)};
  }
  for (int i = 0; i < iters; ++i) {
    // y changes every iteration (a stale temporary of the previous iteration changes the sum)
    const float yv = (float)((i & 3) + 1);
    f2 y = f2{yv, yv + 1.f};
    asm volatile("" : "+v"(y));
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
      f2 t;
      if (PACKED) {
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(x[p]), "v"(y));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc[p]) : "v"(acc[p]), "v"(t));
      } else {
        float t0, t1, a0 = acc[p].x, a1 = acc[p].y;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(x[p].x), "v"(y.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(x[p].y), "v"(y.y));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(a0) : "v"(a0), "v"(t0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(a1) : "v"(a1), "v"(t1));
        acc[p] = f2{a0, a1};
      }
    }
  }
  float* o = out + ((size_t)blockIdx.x * 64 + lane) * (2 * NPAIR);
#pragma unroll
  for (int p = 0; p < NPAIR; ++p) { o[2 * p] = acc[p].x; o[2 * p + 1] = acc[p].y; }
}

template <bool PAD>
__global__ __launch_bounds__(256, 2) void aggressor(float* __restrict__ sink, int iters) {
  if (PAD) asm volatile("; padded to 256 VGPRs" ::: "v255");
  else asm volatile("; 184 VGPRs" ::: "v183");
  __shared__ __attribute__((aligned(16))) unsigned short lds[8 * 1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8 * 1024; i += 256) lds[i] = (unsigned short)(0x3c00 + ((i * 7) & 0xff));   // bf16 around 0.008-0.01
  __syncthreads();
  f32x16 acc[8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(&lds[((tid & 63) * 8 + (i & 7) * 512) & 8191]);
    const bf16x8 fb = *reinterpret_cast<const bf16x8*>(&lds[((tid & 63) * 8 + ((i + 3) & 7) * 512 + 4096) & 8191]);
#pragma unroll
    for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[a], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (s == 12345.678f) sink[blockIdx.x * 256 + tid] = s;   // keeps the accumulators live
}

struct Result { long launches = 0, wrong = 0; long quarter[4] = {0, 0, 0, 0}; std::vector<std::string> first; double victim_ms = 0; };

template <bool PACKED, int AGG /*0 none, 1 plain, 2 padded*/>
Result run(int rounds, int vblocks, int viters, int ablocks, int aiters) {
  Result res;
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  float *out, *sink;
  const size_t n = (size_t)vblocks * 64 * 2 * NPAIR;
  CK(hipMalloc(&out, n * sizeof(float)));
  CK(hipMalloc(&sink, (size_t)ablocks * 256 * sizeof(float)));
  std::vector<float> host(n);
  // expected sums: sum_i x * y_i with y = (i & 3) + 1 (+ 1 for the odd element)
  double sy0 = 0, sy1 = 0;
  for (int i = 0; i < viters; ++i) { sy0 += (i & 3) + 1; sy1 += (i & 3) + 2; }
  for (int r = 0; r < rounds; ++r) {
    CK(hipMemsetAsync(out, 0xFF, n * sizeof(float), sa));
    if (AGG) {
      for (int k = 0; k < 3; ++k) {
        if (AGG == 1) hipLaunchKernelGGL(aggressor<false>, dim3(ablocks), dim3(256), 0, sb, sink, aiters);
        else hipLaunchKernelGGL(aggressor<true>, dim3(ablocks), dim3(256), 0, sb, sink, aiters);
      }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, sa));
    hipLaunchKernelGGL(victim<PACKED>, dim3(vblocks), dim3(64), 0, sa, out, viters);
    CK(hipEventRecord(e1, sa));
    CK(hipGetLastError());
    CK(hipMemcpyAsync(host.data(), out, n * sizeof(float), hipMemcpyDeviceToHost, sa));
    CK(hipStreamSynchronize(sa));
    CK(hipStreamSynchronize(sb));
    { float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1)); res.victim_ms += ms; CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); }
    ++res.launches;
    for (int b = 0; b < vblocks; ++b)
      for (int lane = 0; lane < 64; ++lane) {
        bool bad = false;
        for (int p = 0; p < NPAIR; ++p) {
          const float e0 = (float)(((lane + p) & 7) * sy0), e1 = (float)(((lane * 3 + p) & 7) * sy1);
          const float g0 = host[((size_t)b * 64 + lane) * 2 * NPAIR + 2 * p], g1 = host[((size_t)b * 64 + lane) * 2 * NPAIR + 2 * p + 1];
          if (g0 != e0 || g1 != e1) {
            bad = true;
            if (res.first.size() < 6) {
              char buf[200];
              snprintf(buf, sizeof buf, "round %d block %d lane %d pair %d: got (%.1f, %.1f) expected (%.1f, %.1f)", r, b, lane, p, g0, g1, e0, e1);
              res.first.push_back(buf);
            }
          }
        }
        if (bad) { ++res.wrong; ++res.quarter[lane >> 4]; }
      }
  }
  CK(hipFree(out)); CK(hipFree(sink));
  CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
  return res;
}

void emit(const char* name, const Result& r, bool last) {
  // victim_ms_mean: the victim grid's duration (HIP events on its stream) -- longer beside the aggressor = they really share the CUs
  printf("\"%s\": {\"victim_launches\": %ld, \"victim_ms_mean\": %.3f, \"wrong_lanes\": %ld, \"by_quarter_of_wave\": [%ld, %ld, %ld, %ld], \"first\": [", name, r.launches,
         r.launches ? r.victim_ms / r.launches : 0.0, r.wrong, r.quarter[0], r.quarter[1], r.quarter[2], r.quarter[3]);
  for (size_t i = 0; i < r.first.size(); ++i) printf("%s\"%s\"", i ? ", " : "", r.first[i].c_str());
  printf("]}%s", last ? "" : ", ");
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 24;
  // victim: 2048 one-wave blocks (8 per CU: spread over all SIMDs), ~200 us each; aggressor: 512 blocks (2 per CU), ~1 ms
  const int vblocks = 2048, viters = 1 << 14, ablocks = 512, aiters = 1 << 13;
  // exactness bound: sums stay below 2^24 (7 * 5 * 2^14 * ... ): x <= 7, y <= 5 -> 35 * 16384 = 573440 < 2^24
  printf("{\"what\": \"packed-fp32 VALU victim beside an MFMA aggressor on another stream (tools/hazard/pk_hazard.hip)\", \"rounds\": %d, ", rounds);
  emit("packed_alone", run<true, 0>(rounds / 4 + 1, vblocks, viters, ablocks, aiters), false);
  emit("scalar_beside_mfma_184vgpr", run<false, 1>(rounds, vblocks, viters, ablocks, aiters), false);
  emit("packed_beside_mfma_184vgpr", run<true, 1>(rounds, vblocks, viters, ablocks, aiters), false);
  emit("packed_beside_mfma_padded_256vgpr", run<true, 2>(rounds, vblocks, viters, ablocks, aiters), true);
  printf("}\n");
  return 0;
}
