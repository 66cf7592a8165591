// Probe of the operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) inputs on gfx950 (no public table at hand):
//  part 1: which (lane half, byte) of the A operand meets which (lane half, byte) of the B operand in the reduction:
//          one wave per (ha, ja): A has a single 1.0 at row 0 = lane 32*ha, byte ja; B[k][0] = 2^-(code) with a distinct power of two
//          per (hb, jb) -> C[0][0] identifies the partner.  Done in two passes (bytes' values limited by e4m3 range): jb & 15 coded
//          as value, the rest by which quarter is populated.
//  part 2: which scale byte (lane 0 / lane 32 register, byte op_sel) applies to which 16-byte half of lane 0's / lane 32's A bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

// e4m3 codes of 2^-6 .. 2^8 step: exponent field e (1..15), mantissa 0 -> 2^(e-7); code = e << 3
__device__ int pow2_code(int e) { return (e & 15) << 3; }

// out[(pa * 4 + quarter)] = C[0][0] with A one-hot at pa = (ha, ja) and B's quarter (hb, 16-byte half) populated with 2^(j-6), j = 0..15 -> e = j+1
__global__ void probe_k(float* out) {
  const int w = blockIdx.x;            // pa * 4 + quarter
  const int pa = w >> 2, qu = w & 3;
  const int ha = pa >> 5, ja = pa & 31, hb = qu >> 1, hf = qu & 1;
  const int lane = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (lane == 32 * ha && (ja >> 2) == i) ? (0x38 << (8 * (ja & 3))) : 0;
    int v = 0;
    if (lane == 32 * hb && (i >> 2) == hf) {
      const int j0 = (i & 3) * 4;     // byte index inside the 16-byte half
      v = pow2_code(j0 + 1) | (pow2_code(j0 + 2) << 8) | (pow2_code(j0 + 3) << 16) | (pow2_code(j0 + 4) << 24);
    }
    b[i] = v;
  }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  const int one = 0x7f7f7f7f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, one, 0, one);
  if (lane == 0) out[w] = c[0];
}

// out[osel * 4 + region]: A populated (1.0) only in region = (lane half h, 16-byte half f) of row 0; B column 0 all ones;
// lane 0's scale register = bytes {2^1, 2^2, 2^3, 2^4}, lane 32's = {2^5, 2^6, 2^7, 2^8}; result / 16 = the scale that region got
template <int OSEL>
__device__ float scale_case(int lane, int h, int f) {
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (lane == 32 * h && (i >> 2) == f) ? 0x38383838 : 0;
    b[i] = (lane & 31) == 0 ? 0x38383838 : 0;
  }
  int sa = 0x7f7f7f7f;
  if (lane == 0) sa = 0x83828180;
  if (lane == 32) sa = 0x87868584;
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  const int one = 0x7f7f7f7f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OSEL, sa, 0, one);
  return c[0];
}
__global__ void probe_scale(float* out) {
  const int lane = threadIdx.x;
  for (int r = 0; r < 4; ++r) {
    const float v0 = scale_case<0>(lane, r >> 1, r & 1), v1 = scale_case<1>(lane, r >> 1, r & 1);
    const float v2 = scale_case<2>(lane, r >> 1, r & 1), v3 = scale_case<3>(lane, r >> 1, r & 1);
    if (lane == 0) { out[0 * 4 + r] = v0; out[1 * 4 + r] = v1; out[2 * 4 + r] = v2; out[3 * 4 + r] = v3; }
  }
}
int main() {
  float* d; CK(hipMalloc(&d, 4096 * sizeof(float)));
  CK(hipMemset(d, 0, 4096 * sizeof(float)));
  hipLaunchKernelGGL(probe_k, dim3(256), dim3(64), 0, 0, d);
  CK(hipGetLastError()); CK(hipDeviceSynchronize());
  std::vector<float> h(4096); CK(hipMemcpy(h.data(), d, 4096 * sizeof(float), hipMemcpyDeviceToHost));
  for (int pa = 0; pa < 64; ++pa) {
    printf("A(half %d, byte %2d) meets", pa >> 5, pa & 31);
    for (int qu = 0; qu < 4; ++qu) {
      const float v = h[pa * 4 + qu];
      if (v != 0.f) {
        int j = -1;
        for (int t = 0; t < 16; ++t) if (v == ldexpf(1.f, t + 1 - 7)) j = t;
        printf("  B(half %d, byte %2d) [value %g]", qu >> 1, (qu & 1) * 16 + j, v);
      }
    }
    printf("\n");
  }
  hipLaunchKernelGGL(probe_scale, dim3(1), dim3(64), 0, 0, d);
  CK(hipGetLastError()); CK(hipDeviceSynchronize());
  CK(hipMemcpy(h.data(), d, 16 * sizeof(float), hipMemcpyDeviceToHost));
  for (int o = 0; o < 4; ++o)
    for (int r = 0; r < 4; ++r)
      printf("op_sel %d: A bytes of (lane half %d, 16-byte half %d) got scale %g  (lane 0 reg = 2,4,8,16; lane 32 reg = 32,64,128,256)\n", o, r >> 1, r & 1, h[o * 4 + r] / 16.f);
  return 0;
}
