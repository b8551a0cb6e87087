// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the audio-captioning path.
// Wave size is 64 everywhere; nothing here compiles for another target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AC_WAVE 64

// Error codes of the C ABI (include/audiocaption_hip.h).
#define AC_OK 0
#define AC_ERR_ARG (-1)
#define AC_ERR_LAUNCH (-2)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_32x32x2_f32: D(32x32) += A(32x2) * B(2x32), exact f32 (k-ordered fmaf chain).
// Lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; the result register r
// of lane l is D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// 64-lane reductions on the DPP cross-lane network (quad permutes, row mirrors) plus four scalar lane
// reads: ~10x cheaper than a ds_bpermute butterfly, which goes through the LDS crossbar at every step.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;   // quad_perm [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

// sum / max over each 16-lane row, result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<DPP_QUAD_XOR1>(v);
  v += dpp_mov<DPP_QUAD_XOR2>(v);
  v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_mov<DPP_ROW_MIRROR>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<DPP_QUAD_XOR1>(v));
  v = fmaxf(v, dpp_mov<DPP_QUAD_XOR2>(v));
  v = fmaxf(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_mov<DPP_ROW_MIRROR>(v));
  return v;
}
__device__ __forceinline__ float lane_read(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (lane_read(v, 0) + lane_read(v, 16)) + (lane_read(v, 32) + lane_read(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(lane_read(v, 0), lane_read(v, 16)), fmaxf(lane_read(v, 32), lane_read(v, 48)));
}

// swish / sigmoid on the transcendental unit: v_exp_f32 and v_rcp_f32 (1 ulp each), 6 instructions instead of the ~25 of
// expf + an IEEE division.  EfficientNet applies one swish per activation, which made the accurate form the largest VALU
// item of its streaming kernels.  exp2 overflows to +inf for v < -88 (rcp(inf) = 0: the result is -0, the limit).
__device__ __forceinline__ float ac_sigmoid_fast(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * v));
}
__device__ __forceinline__ float ac_swish_fast(float v) { return v * ac_sigmoid_fast(v); }
// The accurate forms (expf + IEEE division) for the kernels documented as exact f32 (gemm_general / gemm_nt / gemm_kk).
__device__ __forceinline__ float ac_sigmoid_exact(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ float ac_swish_exact(float v) { return v * ac_sigmoid_exact(v); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter
// (s_waitcnt vmcnt(0)), which would force every global prefetch in flight to land at each barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: remembered per (launch site, device), so
// that a process which touches a second GPU sets it there as well (a plain function-local flag would skip it and the
// launch would fail with more than 64 KiB of dynamic LDS).  One word per site: bit d = device d has the attribute.
struct AcLdsAttr { unsigned long long done = 0; };
static inline int ac_allow_lds(const void* kernel, int bytes, AcLdsAttr* st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return AC_ERR_LAUNCH;
  const bool tracked = dev >= 0 && dev < 64;
  if (tracked && ((__atomic_load_n(&st->done, __ATOMIC_ACQUIRE) >> dev) & 1ull)) return AC_OK;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return AC_ERR_LAUNCH;
  if (tracked) __atomic_fetch_or(&st->done, 1ull << dev, __ATOMIC_RELEASE);
  return AC_OK;
}

static inline int ac_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? AC_OK : AC_ERR_LAUNCH;
}
