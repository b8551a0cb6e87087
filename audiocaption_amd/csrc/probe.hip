// Diagnostic, not on the hot path: the matrix rate this part SUSTAINS on the conv tier's instruction
// (v_mfma_f32_32x32x16_bf16, 8 independent accumulators per wave, two 4-wave workgroups per CU, no memory traffic).
// bench.py runs it next to the timed region so that `roofline` can say how much of the gap to the nominal 2.5 PFLOP/s is
// the clock the part holds while its matrix pipes are busy (~1.7 of 2.4 GHz) and how much is the kernel.
#include "ac_common.h"

namespace {

typedef __bf16 pb_bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 2) void mfma_bf16_probe_kernel(float* out, int iters) {
  f32x16 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  pb_bf16x8 a[8], b;
#pragma unroll
  for (int e = 0; e < 8; ++e) b[e] = (__bf16)(0.001f * (float)((threadIdx.x + e) & 31));
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int e = 0; e < 8; ++e) a[m][e] = (__bf16)(0.01f * (float)(m + e + (threadIdx.x & 7)));
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b, acc[m], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[m][r];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// where a workgroup ran: out[2 b] = XCC_ID register, out[2 b + 1] = HW_ID register (cu_id bits 11:8, sh_id 12, se_id 15:13)
__global__ void placement_probe_kernel(int* out, int spin) {
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
    out[2 * blockIdx.x + 1] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
  }
  const long long t0 = clock64();
  while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);   // keep the workgroup resident so that the grid spreads
}

}  // namespace

// C ABI: see include/audiocaption_hip.h
extern "C" int ac_mfma_bf16_probe(float* out, int blocks, int iters, void* stream) {
  if (!out || blocks <= 0 || iters <= 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(mfma_bf16_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  return ac_check_launch();
}

extern "C" int ac_placement_probe(int* out, int blocks, int threads, int spin_ticks, void* stream) {
  if (!out || blocks <= 0 || threads <= 0 || threads > 1024) return AC_ERR_ARG;
  hipLaunchKernelGGL(placement_probe_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, spin_ticks);
  return ac_check_launch();
}

// A HIP stream whose kernels only run on bits first_cu .. first_cu + n_cus - 1 of the device's CU mask
// (hipExtStreamCreateWithCUMask; where those bits land is measured by ac_placement_probe, tools/cu_mask_probe.py).
extern "C" int ac_stream_create_cu_mask(int first_cu, int n_cus, void** out) {
  if (!out || n_cus <= 0 || first_cu < 0) return AC_ERR_ARG;
  int dev = 0, total = 0;
  if (hipGetDevice(&dev) != hipSuccess) return AC_ERR_LAUNCH;
  if (hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return AC_ERR_LAUNCH;
  if (first_cu >= total) return AC_ERR_ARG;
  if (first_cu + n_cus > total) n_cus = total - first_cu;
  const int words = (total + 31) / 32;
  unsigned mask[32] = {0};
  if (words > 32) return AC_ERR_ARG;
  for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask) != hipSuccess) return AC_ERR_LAUNCH;
  *out = (void*)s;
  return AC_OK;
}

extern "C" int ac_stream_destroy(void* stream) {
  if (!stream) return AC_ERR_ARG;
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? AC_OK : AC_ERR_LAUNCH;
}
