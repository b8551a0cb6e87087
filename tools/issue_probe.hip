// Development probe: what one in-order wave per SIMD pays for the NON-MFMA instructions of the F(4,3) conv kernel's K step.
// A group = 6 v_mfma_f32_32x32x16_bf16 on 12 accumulators (192 registers, one wave per SIMD, one workgroup per CU) + NL
// buffer_load_dwordx4 of weight-like fragments from an L2-resident buffer (consumed 8 groups later, ring of 9, like the
// kernel) + ND ds_read_b128 + NV dependent-free VALU instructions.  Reports shader cycles per group (192 = the matrix
// pipe's own rate).  WAVES: how many of the workgroup's four waves run the loop (1 = no contention for the CU's TA / L1).
//   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o tools/bin/issue_probe && tools/bin/issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_cycles[4];

template <int NL, int ND, int NV, int KIND, int NS = 0, int SCHED = 0>   // SCHED: sched_group_barrier pipelines (see below); NS 1: + one ds_write_b64 in each of the first 12 groups and a barrier at group 15; KIND 0: loads to VGPRs; 1: the same bytes as dwordx2 pairs; 2: LDS-DMA (dword x 4 instr)
__global__ __launch_bounds__(256, 1) void k(const float* src, float* out, int iters, int waves, int src_bytes) {
  extern __shared__ __attribute__((aligned(128))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16384; i += 256) ((float*)lds)[i] = 0.001f * (float)(i & 63);
  __syncthreads();
  if (wave >= waves) return;
  f32x16 acc[12];
#pragma unroll
  for (int m = 0; m < 12; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
  bf16x8 wr[9][4], af[3][4];
#pragma unroll
  for (int g = 0; g < 9; ++g)
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int e = 0; e < 8; ++e) wr[g][h][e] = (__bf16)(0.001f * (float)((lane + e + g) & 31));
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int e = 0; e < 8; ++e) af[g][h][e] = (__bf16)(0.01f * (float)((lane + e) & 7));
  float vf[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) vf[i] = 1.0f + 0.001f * (float)(lane + i);
  const unsigned voff = (unsigned)((blockIdx.x & 63) * 65536 + wave * 2048 + lane * 16);
  const unsigned ldsoff = (unsigned)(lane * 16 + wave * 4096);
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const unsigned so = (unsigned)((it & 7) * 8192);
#pragma unroll
    for (int gi = 0; gi < 18; ++gi) {
      // loads for group gi + 8 (ring of 9)
      if (NL > 0) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          if (KIND == 0) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (unsigned)(l & 1) * 1024u, so + (unsigned)(gi * 128), 0);
            wr[(gi + 8) % 9][l] = __builtin_bit_cast(bf16x8, v);
          } else if (KIND == 1) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rs, voff / 2 + (unsigned)(l & 1) * 1024u, so + (unsigned)(gi * 128), 0);
            const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(rs, voff / 2 + (unsigned)(l & 1) * 1024u + 512u, so + (unsigned)(gi * 128), 0);
            u32x4 v = {a.x, a.y, b.x, b.y};
            wr[(gi + 8) % 9][l] = __builtin_bit_cast(bf16x8, v);
          }
        }
      }
      if (ND > 0) {
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          const bf16x8 v = *(const bf16x8*)(lds + ldsoff + (unsigned)(((gi * 4 + d) & 15) * 1024));
          af[(gi + 2) % 3][d] = v;
        }
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int m = (gi % 6) * 2 + (j & 1);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[gi % 9][j % 2], af[gi % 3][(j >> 1) & 1], acc[m], 0, 0, 0);
      }
      // every fragment requested for this group is consumed HERE (like the kernel: A two groups, weights eight groups after the request)
      if (NL > 2) asm volatile("" :: "v"(wr[gi % 9][2]));
      if (NL > 3) asm volatile("" :: "v"(wr[gi % 9][3]));
      if (ND > 2) asm volatile("" :: "v"(af[gi % 3][2]));
      if (ND > 3) asm volatile("" :: "v"(af[gi % 3][3]));
      if (NV > 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) vf[v & 7] = __builtin_fmaf(vf[v & 7], 1.0001f, 0.5f);
      }
      if (NS > 0) {
        if (gi < 12) {
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          *(f32x2*)(lds + 65536 + (unsigned)(tid * 8 + (gi & 3) * 2048)) = (f32x2){vf[gi & 7], vf[(gi + 1) & 7]};
        }
        if (gi == 15) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      // SCHED 1: one filler class after another in every MFMA gap (memory first, then VALU); 2: all memory instructions of
      // the group before its first MFMA, VALU spread; 3: the six MFMAs back to back, everything else behind them;
      // 4: like 1 with the VALU in front of the memory instructions
      if (SCHED == 1 || SCHED == 4) {
#define GAP(J)                                                                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                             \
        if (SCHED == 4 && (NV + 5 - J) / 6 > 0) __builtin_amdgcn_sched_group_barrier(0x002, (NV + 5 - J) / 6, 0); \
        if (J < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                  \
        if (J >= 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                 \
        if (J == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                 \
        if (SCHED == 1 && (NV + 5 - J) / 6 > 0) __builtin_amdgcn_sched_group_barrier(0x002, (NV + 5 - J) / 6, 0);
        GAP(0) GAP(1) GAP(2) GAP(3) GAP(4) GAP(5)
#undef GAP
      } else if (SCHED == 2) {
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#define GAP(J)                                                                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                             \
        if ((NV + 5 - J) / 6 > 0) __builtin_amdgcn_sched_group_barrier(0x002, (NV + 5 - J) / 6, 0);    \
        if (J == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        GAP(0) GAP(1) GAP(2) GAP(3) GAP(4) GAP(5)
#undef GAP
      } else if (SCHED == 3) {
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 12; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[m][r];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += vf[i];
  out[(size_t)blockIdx.x * 256 + tid] = s;
  if (lane == 0 && wave == 0) { atomicAdd(&g_cycles[0], t1 - t0); atomicAdd(&g_cycles[1], 1ull); }
}

template <int NL, int ND, int NV, int KIND, int NS = 0, int SCHED = 0>
void run(const char* tag, const float* src, float* o, int waves, int src_bytes) {
  const int blocks = 256, iters = 2000;
  hipFuncSetAttribute((const void*)k<NL, ND, NV, KIND, NS, SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  unsigned long long z[4] = {0, 0, 0, 0};
  hipLaunchKernelGGL((k<NL, ND, NV, KIND, NS, SCHED>), dim3(blocks), dim3(256), 100 * 1024, 0, src, o, iters / 10, waves, src_bytes);
  hipDeviceSynchronize();
  hipMemcpyToSymbol(HIP_SYMBOL(g_cycles), z, 32);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NL, ND, NV, KIND, NS, SCHED>), dim3(blocks), dim3(256), 100 * 1024, 0, src, o, iters, waves, src_bytes);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpyFromSymbol(z, HIP_SYMBOL(g_cycles), 32);
  const double cyc = (double)z[0] / (double)(z[1] ? z[1] : 1) / ((double)iters * 18.0);
  printf("%-44s waves %d  %7.1f cycles / group of 6 MFMAs (192 = pipe rate)  %6.2f ms  %.2f GHz\n", tag, waves, cyc, ms,
         (double)z[0] / (double)(z[1] ? z[1] : 1) / (ms * 1e6));
}

int main() {
  float *src, *o;
  const int src_bytes = 64 * 65536;   // 4 MB: L2 / MALL resident
  hipMalloc(&src, src_bytes); hipMemset(src, 0, src_bytes);
  hipMalloc(&o, 256 * 256 * 4);
  for (int waves = 4; waves >= 1; waves -= 3) {
    run<0, 0, 0, 0>("MFMAs only", src, o, waves, src_bytes);
    run<2, 4, 0, 0>("+ 2 loads + 4 ds_read", src, o, waves, src_bytes);
    run<2, 4, 8, 0, 1, 0>("2 loads + 4 ds_read + 8 VALU + store + barrier: compiler", src, o, waves, src_bytes);
    run<2, 4, 8, 0, 1, 1>("  ... interleaved, memory before VALU", src, o, waves, src_bytes);
    run<2, 4, 8, 0, 1, 4>("  ... interleaved, VALU before memory", src, o, waves, src_bytes);
    run<2, 4, 8, 0, 1, 2>("  ... memory first, VALU spread", src, o, waves, src_bytes);
    run<2, 4, 8, 0, 1, 3>("  ... MFMAs first", src, o, waves, src_bytes);
    run<2, 4, 16, 0, 1, 0>("2 loads + 4 ds_read + 16 VALU + store + barrier: compiler", src, o, waves, src_bytes);
    run<2, 4, 16, 0, 1, 1>("  ... interleaved, memory before VALU", src, o, waves, src_bytes);
    run<2, 4, 16, 0, 1, 4>("  ... interleaved, VALU before memory", src, o, waves, src_bytes);
    run<2, 4, 16, 0, 1, 2>("  ... memory first, VALU spread", src, o, waves, src_bytes);
    run<2, 4, 16, 0, 1, 3>("  ... MFMAs first", src, o, waves, src_bytes);
  }
  return 0;
}
