// Development probe: what matrix rate does the F(2,3) kernel's OPERAND MIX allow, for different wave tilings?
// Synthetic: no staging, no transforms, no epilogue - per "group" a wave reads MW pixel-tile fragment pairs (hi, lo: 2 KB per
// tile) from LDS, CG weight fragment pairs (2 KB per 32-channel group) from an L2-resident global buffer, and issues
// MW x CG x 3 v_mfma_f32_32x32x16_bf16 into 4 x MW x CG accumulators (the 4 Winograd positions take turns).
//   <MW, CG, WPS>: WPS = waves per SIMD (WPS = 2: 256-thread workgroups, two per CU; WPS = 1: one 256-thread workgroup per CU)
//   today's wide kernel: <2, 1, 2>;  candidates for one 512-register wave per SIMD: <3, 2, 1>, <6, 1, 1>, <2, 2, 1>
// Build: hipcc --offload-arch=gfx950 -O3 tools/w1_mix_probe.hip -o tools/bin/w1_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// APAT: 0 = every fragment read is a wave-contiguous 1 KB (conflict free); 1 = the wide kernel's layout: an item = 16
// channels = 32 bytes, lane i of a 32-row tile reads pair i & 15 of column i >> 4 (column pitch 528 bytes), the upper 32
// lanes the second 16 bytes of the item
// STAGE bits (the wide kernel's staging, per K step and thread of the first three waves): 1 = six 16-byte row loads from a
// 1 GB activation buffer (requested at the top of the step), 2 = the F(2,3) input transform + bf16 hi / lo split of the
// 8 pieces (~110 VALU), 4 = their 16 ds_write_b64 at the kernel's addresses, a piece per group from group 4 on
// RING: weight fragments in flight (RING - 1 groups ahead).  NSET = 2: the rows of step s + 2 are requested in step s and
// consumed in step s + 1 (two register sets)
// NWAVE: waves per workgroup (8: the staging of one pixel tile feeds twice the channels; one workgroup per CU)
template <int MW, int CG, int WPS, int APAT = 0, int STAGE = 0, int RING = 3, int NSET = 1, int NWAVE = 4>
__global__ __launch_bounds__(64 * NWAVE, 2) void mix(const i32x4* wts, unsigned wbytes, float* out, int steps, int barrier,
                                                const float* act = nullptr, unsigned abytes = 0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 48 * 1024 / 16; i += 64 * NWAVE) ((i32x4*)lds)[i] = (i32x4){i, 1, 2, 3};
  __syncthreads();
  f32x16 acc[4][MW][CG];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int c = 0; c < CG; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][m][c][r] = 0.f;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wts, 0, (int)wbytes, 0x00020000);
  // barrier & 4: every workgroup streams the SAME weight bytes in the same order (what the real kernel's workgroups of one
  // channel tile do); otherwise the workgroups are spread over the buffer
  const unsigned wvoff = (unsigned)((((barrier & 4) ? 0 : blockIdx.x * 4) + wave) * 4096 % (wbytes / 2)) + lane * 16;
  bf16x8 a[2][MW][2], w[RING][CG][2];
  auto a_load = [&](int g, bf16x8 (&x)[MW][2]) {
#pragma unroll
    for (int m = 0; m < MW; ++m) {
      unsigned off = (unsigned)(((g * 7 + m * 13) & 31) * 1024 + lane * 16);   // wave-contiguous 1 KB, conflict free
      if (APAT == 1) {
        const int i = lane & 31;
        off = (unsigned)(((g % 3) + 2 * m + (i >> 4)) * 528 + (i & 15) * 32 + (lane >> 5) * 16 + (g & 3) * 6336);
      }
      x[m][0] = *(const bf16x8*)(lds + off);
      x[m][1] = *(const bf16x8*)(lds + ((off + (APAT == 1 ? 3168 : 16384)) & (48 * 1024 - 1) & ~15u));
    }
  };
  auto w_load = [&](int g, bf16x8 (&x)[CG][2]) {
#pragma unroll
    for (int c = 0; c < CG; ++c) {
      const unsigned so = (unsigned)((g * 8192u + c * 2048u) % (wbytes / 2));
      x[c][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff, so, 0));
      x[c][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff + 1024u, so, 0));
    }
  };
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)act, 0, (int)abytes, 0x00020000);
  const bool has_item = threadIdx.x < 192;
  const int tq = threadIdx.x & 3, tpc = (threadIdx.x >> 2) % 6, tpr = 2 * ((threadIdx.x >> 2) / 6);
  const unsigned lofs = (unsigned)(tpr * 32 + tpc * 528 + tq * 8);          // bytes inside a plane of 3168
  // like the kernel: 4 lanes share a 64-byte segment (16 channels of one position), positions 512 bytes apart (Cin = 128)
  const unsigned vbase = has_item ? (unsigned)(((size_t)blockIdx.x * 131072u + (threadIdx.x >> 2) * 512u + (threadIdx.x & 3) * 16u) % (abytes - (1u << 20))) : 0x80000000u;
  f32x4 pre[NSET][6];
  for (int t_ = 0; t_ < NSET; ++t_)
    for (int r = 0; r < 6; ++r) pre[t_][r] = (f32x4){1.f + r, 2.f, 3.f, 4.f + lane};
  auto cvt = [](float lo, float hi) { unsigned r_; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r_) : "v"(lo), "v"(hi)); return r_; };
  auto commit = [&](int piece, int buf, auto SET_) {
    constexpr int st = decltype(SET_)::value;
    if (!has_item) return;
    const int e = piece >> 2, q = piece & 3;
    const f32x4 d0 = pre[st][2 * e], d1 = pre[st][2 * e + 1], d2 = pre[st][2 * e + 2], d3 = pre[st][2 * e + 3];
    const f32x4 v = q == 0 ? d0 - d2 : (q == 1 ? d1 + d2 : (q == 2 ? d2 - d1 : d1 - d3));
    u32x2 hi, lo;
    hi.x = cvt(v[0], v[1]); hi.y = cvt(v[2], v[3]);
    const float h0 = __builtin_bit_cast(float, hi.x << 16), h1 = __builtin_bit_cast(float, hi.x & 0xffff0000u);
    const float h2 = __builtin_bit_cast(float, hi.y << 16), h3 = __builtin_bit_cast(float, hi.y & 0xffff0000u);
    lo.x = cvt(v[0] - h0, v[1] - h1); lo.y = cvt(v[2] - h2, v[3] - h3);
    if (STAGE & 4) {
      unsigned char* dst = lds + (buf ? 25344 : 0) + lofs + e * 32 + (2 * q) * 3168;
      *(u32x2*)dst = hi;
      *(u32x2*)(dst + 3168) = lo;
    } else {
      asm volatile("" :: "v"(hi), "v"(lo));
    }
  };
#pragma unroll
  for (int g0 = 0; g0 < RING - 1; ++g0) w_load(g0, w[g0]);
  a_load(0, a[0]);
  int g = 0;
  auto step = [&](int s, auto REQ_, auto USE_) {   // REQ_: the set this step's request fills; USE_: the set it transforms
    constexpr int rq = decltype(REQ_)::value;
    if (STAGE & 1) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
        pre[rq][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, vbase + (unsigned)((barrier & 2) ? ((unsigned)s * 9437184u + (unsigned)(s >> 4) * 64u) % (1u << 29) : (s & 7) * 64u) + r * 16384u, 0, 0));
    }
#pragma unroll
    for (int gi = 0; gi < 12; ++gi, ++g) {
      a_load(g + 1, a[(gi + 1) & 1]);
      w_load(g + RING - 1, w[(gi + RING - 1) % RING]);
#pragma unroll
      for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int c = 0; c < CG; ++c) {
          f32x16& d = acc[gi & 3][m][c];
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[gi & 1][m][1], w[gi % RING][c][0], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[gi & 1][m][0], w[gi % RING][c][1], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[gi & 1][m][0], w[gi % RING][c][0], d, 0, 0, 0);
        }
      if ((STAGE & 2) && gi >= 4) commit(gi - 4, s & 1, USE_);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (barrier & 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, NSET - 1>;
#pragma unroll 1
  for (int s = 0; s < steps; s += 2) {
    if (NSET == 2) { step(s, I0{}, I1{}); step(s + 1, I1{}, I0{}); }
    else { step(s, I0{}, I0{}); step(s + 1, I0{}, I0{}); }
  }
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int c = 0; c < CG; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[q][m][c][r];
  out[(size_t)blockIdx.x * 64 * NWAVE + threadIdx.x] = sum;
}

static int g_steps = 400;
static const float* g_act;
static unsigned g_abytes;
template <int MW, int CG, int WPS, int APAT = 0, int STAGE = 0, int RING = 3, int NSET = 1, int NWAVE = 4>
void run(const char* tag, const i32x4* w, unsigned wbytes, float* o, int barrier) {
  const int steps = g_steps, blocks = 256 * WPS * 4 * (400 / g_steps) * 4 / NWAVE;   // 4 rounds of resident workgroups at 400 steps
  hipFuncSetAttribute((const void*)mix<MW, CG, WPS, APAT, STAGE, RING, NSET, NWAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix<MW, CG, WPS, APAT, STAGE, RING, NSET, NWAVE>), dim3(blocks), dim3(64 * NWAVE), 64 * 1024, 0, w, wbytes, o, steps, barrier, g_act, g_abytes);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)blocks * NWAVE * steps * 12 * MW * CG * 3;
  const double lds_kb = (double)blocks * NWAVE * steps * 12 * MW * 2.0, l1_kb = (double)blocks * NWAVE * steps * 12 * CG * 2.0;
  printf("%-28s barrier %d: %7.2f ms  %7.1f TFLOP/s issued  (per MFMA: %4.0f B LDS, %4.0f B L1)\n", tag, barrier, ms,
         mfma * 32768.0 / ms / 1e9, lds_kb * 1024 / mfma, l1_kb * 1024 / mfma);
}

int main() {
  const unsigned wbytes = 3u << 20;
  i32x4* w; float* o;
  hipMalloc(&w, wbytes); hipMemset(w, 0x3c, wbytes);
  hipMalloc(&o, (size_t)256 * 2 * 4 * 50 * 256 * 4);
  float* act; g_abytes = 1u << 30;
  hipMalloc(&act, g_abytes); hipMemset(act, 0, g_abytes);
  g_act = act;
  for (int st : {400, 32, 8}) {
    g_steps = st;
    printf("---- %d K steps per workgroup; barrier 1 = rows from L2, 3 = rows streamed from HBM\n", st);
    run<2, 1, 2, 1, 0>("no staging", w, wbytes, o, 1);
    run<2, 1, 2, 1, 0>("no staging, one weight stream", w, wbytes, o, 5);
    run<2, 1, 2, 1, 7>("all staging, one weight stream", w, wbytes, o, 5);
    run<2, 1, 2, 1, 7>("all staging", w, wbytes, o, 1);
    run<2, 1, 2, 1, 7>("all staging", w, wbytes, o, 3);
    run<2, 1, 2, 1, 7, 6, 1>("all staging, ring 6", w, wbytes, o, 3);
    run<2, 1, 2, 1, 7, 3, 2>("all staging, 2 row sets", w, wbytes, o, 3);
    run<2, 1, 2, 1, 7, 3, 1, 8>("all staging, 8 waves per workgroup", w, wbytes, o, 1);
    run<2, 1, 2, 1, 0, 3, 1, 8>("no staging, 8 waves per workgroup", w, wbytes, o, 1);
  }
  for (int b = 0; b < 0; ++b) {
    run<2, 1, 2>("MW 2 CG 1, 2 waves/SIMD", w, wbytes, o, b);
    run<2, 1, 2, 1>("  with the kernel's LDS layout", w, wbytes, o, b);
    run<2, 1, 2, 1, 1>("  + row loads", w, wbytes, o, b);
    run<2, 1, 2, 1, 2>("  + transform / split", w, wbytes, o, b);
    run<2, 1, 2, 1, 6>("  + transform + LDS stores", w, wbytes, o, b);
    run<2, 1, 2, 1, 7>("  + all staging", w, wbytes, o, b);
    run<2, 1, 2, 1, 7, 6, 1>("  all staging, ring 6", w, wbytes, o, b);
    run<2, 1, 2, 1, 7, 3, 2>("  all staging, 2 row sets", w, wbytes, o, b);
    run<2, 1, 2, 1, 7, 4, 2>("  all staging, ring 4 + 2 sets", w, wbytes, o, b);
    run<2, 1, 2, 1, 7, 6, 2>("  all staging, ring 6 + 2 sets", w, wbytes, o, b);
    run<2, 2, 1>("MW 2 CG 2, 1 wave/SIMD", w, wbytes, o, b);
    run<4, 1, 1>("MW 4 CG 1, 1 wave/SIMD", w, wbytes, o, b);
  }
  return 0;
}
