// Development probe: calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the ACCESS PATTERNS of the F(4,3) conv kernel
// (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a wide coalesced streaming read; other patterns are
// uncalibrated).  Every kernel moves exactly 1 GiB (4x the 256 MiB memory-side cache), so bytes / counter = the factor.
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --stats -d <dir> -- tools/bin/traffic_calib      (and again with WRITE_SIZE)
//   k_stream_read   : 16 B per lane, consecutive lanes consecutive addresses (the guide's calibration case)
//   k_rows_read     : the conv kernel's row loads - [pixel][128 channels] f32, K step s reads channels 16 s .. 16 s + 15 of
//                     every pixel of the workgroup's tile: 64-byte segments at a 512-byte stride, the other half of each
//                     128-byte line one step later
//   k_rows_read_once: the same segments, every line's two halves by the same instruction (what one would get from a
//                     32-channel K step)
//   k_frag_read     : the weight fragments - 1 KiB contiguous per wave instruction, every workgroup of an XCD the SAME bytes
//                     (32 MiB, by each of 256 workgroups: what reaches the fabric is what the L2s miss)
//   k_lines_write   : the epilogue's stores - 16 B per lane, 8 lanes = one 128-byte line, lines 512 bytes apart
//   k_stream_write  : 16 B per lane contiguous
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr size_t GIB = 1ull << 30;

__global__ __launch_bounds__(256) void k_stream_read(const f32x4* src, float* out, size_t n16) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc += src[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}

// tile = 1024 pixels of 128 channels (512 KiB); 8 K steps of 16 channels; thread t: pixel t / 4 (+ 64 per instruction), channel quad t % 4
template <int CH_PER_STEP>
__global__ __launch_bounds__(256) void k_rows_read(const float* src, float* out, int tiles) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int LPP = CH_PER_STEP / 4;          // lanes per pixel
  constexpr int PPI = 256 / LPP;                // pixels per workgroup instruction
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const float* base = src + (size_t)tile * 1024 * 128;
    for (int s = 0; s < 128 / CH_PER_STEP; ++s) {
#pragma unroll 4
      for (int p0 = 0; p0 < 1024; p0 += PPI) {
        const int p = p0 + threadIdx.x / LPP;
        acc += *(const f32x4*)(base + (size_t)p * 128 + s * CH_PER_STEP + (threadIdx.x % LPP) * 4);
      }
      __syncthreads();
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void k_frag_read(const f32x4* src, float* out, size_t n16) {
  // EVERY workgroup walks the same n16 * 16 bytes in the same order, a wave instruction reads 1 KiB: what reaches the fabric
  // is what the eight L2s miss (8 x the buffer if the workgroups of an XCD stay in step)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = threadIdx.x; i < n16; i += 256) acc += src[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void k_lines_write(float* dst, size_t pixels) {
  // wave w of 4 owns 32 channels (128 B) of every pixel; lane l: pixel 8 k + l / 8, 16-byte piece l % 8
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t p0 = (size_t)blockIdx.x * 8; p0 < pixels; p0 += (size_t)gridDim.x * 8) {
    const size_t p = p0 + (lane >> 3);
    *(f32x4*)(dst + p * 128 + wave * 32 + (lane & 7) * 4) = v;
  }
}

__global__ __launch_bounds__(256) void k_stream_write(f32x4* dst, size_t n16) {
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = v;
}

int main() {
  float *buf, *out;
  hipMalloc(&buf, GIB); hipMalloc(&out, 64);
  hipMemset(buf, 0, GIB);
  hipDeviceSynchronize();
  const int tiles = (int)(GIB / (1024 * 128 * 4));   // 2048
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_stream_read, dim3(2048), dim3(256), 0, 0, (const f32x4*)buf, out, GIB / 16);
    hipLaunchKernelGGL(k_rows_read<16>, dim3(512), dim3(256), 0, 0, buf, out, tiles);
    hipLaunchKernelGGL(k_rows_read<32>, dim3(512), dim3(256), 0, 0, buf, out, tiles);
    hipLaunchKernelGGL(k_frag_read, dim3(256), dim3(256), 0, 0, (const f32x4*)buf, out, (size_t)(32 << 20) / 16);
    hipLaunchKernelGGL(k_lines_write, dim3(2048), dim3(256), 0, 0, buf, GIB / 512);
    hipLaunchKernelGGL(k_stream_write, dim3(2048), dim3(256), 0, 0, (f32x4*)buf, GIB / 16);
    hipDeviceSynchronize();
  }
  printf("every kernel but k_frag_read moved 1 GiB (k_frag_read: each of 256 workgroups reads the same 32 MiB)\n");
  return 0;
}
