// Debug canaries of the round-3 concurrency hunt: LDS, global-load and VGPR integrity probes that run beside
// other kernels.  NOT part of libspeecht5_hip.so: built on demand into tools/diag/libst5_diag.so by tools/diag/build.sh
// and loaded by tools/diag/diag_{lds,loads,vgpr}.py.
#include "../../speecht5_amd/csrc/common.h"
// ---- debug: LDS canary -----------------------------------------------------------------------------------------------------
// Every block fills `bytes` of dynamic LDS with a block-specific pattern and re-reads it `spins` times; a word that changed
// under it (another workgroup of ANY kernel on the same CU wrote outside its own LDS allocation) is counted in errors[0].
// Used by tools/diag_lds.py next to the GEMM kernels (whose LDS-DMA loads write LDS asynchronously).
namespace {
__global__ __launch_bounds__(256) void lds_canary_kernel(int* __restrict__ errors, int words, int spins) {
  extern __shared__ unsigned int canary[];
  const unsigned int pat = 0xA5000000u ^ (blockIdx.x * 2654435761u);
  for (int i = threadIdx.x; i < words; i += 256) canary[i] = pat ^ (unsigned int)i;
  __syncthreads();
  int bad = 0;
  for (int s = 0; s < spins; ++s) {
    for (int i = threadIdx.x; i < words; i += 256) bad += canary[i] != (pat ^ (unsigned int)i);
    __builtin_amdgcn_s_sleep(8);
  }
  if (bad) atomicAdd(errors, bad);
}
}  // namespace
extern "C" int st5_debug_lds_canary(int32_t* errors_dev, int32_t blocks, int32_t bytes, int32_t spins, void* stream) {
  if (!errors_dev || blocks <= 0 || bytes <= 0 || bytes % 4 || bytes > 64 * 1024 || spins <= 0) return ST5_ERR_ARG;
  hipLaunchKernelGGL(lds_canary_kernel, dim3((unsigned)blocks), dim3(256), (size_t)bytes, (hipStream_t)stream, errors_dev, bytes / 4, spins);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

// ---- debug: global-load checker -----------------------------------------------------------------------------------------------
// buf[i] must equal i * 2654435761 (filled by the caller); every thread re-reads its share `passes` times with 4-byte loads and
// counts words that come back different (errors[0] +=).  Detects wrong data returned for CONSTANT memory while other kernels run.
namespace {
__global__ __launch_bounds__(256) void load_check_kernel(const unsigned int* __restrict__ buf, long long n, int* __restrict__ errors, int passes) {
  int bad = 0;
  for (int p = 0; p < passes; ++p)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
      const unsigned int v = __builtin_nontemporal_load(buf + i) ;
      bad += v != (unsigned int)i * 2654435761u;
    }
  if (bad) atomicAdd(errors, bad);
}
}  // namespace
extern "C" int st5_debug_load_check(const uint32_t* buf, int64_t n, int32_t* errors_dev, int32_t blocks, int32_t passes, void* stream) {
  if (!buf || !errors_dev || n <= 0 || blocks <= 0 || passes <= 0) return ST5_ERR_ARG;
  hipLaunchKernelGGL(load_check_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, buf, (long long)n, errors_dev, passes);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}

// ---- debug: VGPR canary ---------------------------------------------------------------------------------------------------------
// Every lane keeps 48 values in registers (forced live with empty asm), sleeps / spins `spins` rounds doing a little arithmetic on
// them that must cancel, and compares with what it started from; mismatches are recorded per LANE (hist[lane] += 1).
namespace {
__global__ __launch_bounds__(256) void vgpr_canary_kernel(int* __restrict__ hist, int spins, unsigned int salt) {
  unsigned int r[48];
  const unsigned int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 48; ++i) { r[i] = (lane * 2654435761u) ^ (salt + i * 40503u) ^ (blockIdx.x << 8); asm volatile("" : "+v"(r[i])); }
  for (int s = 0; s < spins; ++s) {
#pragma unroll
    for (int i = 0; i < 48; ++i) { r[i] += s; asm volatile("" : "+v"(r[i])); }
    __builtin_amdgcn_s_sleep(4);
#pragma unroll
    for (int i = 0; i < 48; ++i) { r[i] -= s; asm volatile("" : "+v"(r[i])); }
  }
  int bad = 0;
#pragma unroll
  for (int i = 0; i < 48; ++i) bad += r[i] != ((lane * 2654435761u) ^ (salt + i * 40503u) ^ (blockIdx.x << 8));
  if (bad) atomicAdd(hist + lane, bad);
}
}  // namespace
extern "C" int st5_debug_vgpr_canary(int32_t* hist64_dev, int32_t blocks, int32_t spins, void* stream) {
  if (!hist64_dev || blocks <= 0 || spins <= 0) return ST5_ERR_ARG;
  hipLaunchKernelGGL(vgpr_canary_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, hist64_dev, spins, 0x9E3779B9u);
  HIP_CHECK_LAUNCH();
  return ST5_OK;
}
