// Micro-benchmark of the vector-memory front end (TA / vector L1) of one CU on gfx950: cycles per wave-instruction of loads of
// different widths and lane -> address patterns on cache-resident data.  What the producer waves of conv3h_kernel can afford.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/experimental/micro_ta tools/experimental/micro_ta.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE: lane -> byte offset inside a 64-KB window (per wave), W = bytes per lane (4, 8, 16)
template <int W>
__device__ __forceinline__ unsigned lane_off(int mode, int lane) {
  switch (mode) {
    case 0: return lane * W;                                   // contiguous, aligned
    case 1: return lane * W + 4;                               // contiguous, off by one float (W = 4 only)
    case 2: return (lane >> 2) * 1024 + (lane & 3) * W;        // lane quads contiguous, quads 1 KB apart
    case 3: return (lane >> 1) * 1024 + (lane & 1) * W;        // lane pairs contiguous
    case 4: return lane * 1024;                                // every lane its own line
    case 5: return (lane / 34) * 4096 + (lane % 34) * W + 124; // runs of 34 lanes starting one float before a line (the 6 x 34 halo tile rows)
    case 6: return (lane >> 3) * 1024 + (lane & 7) * W;        // lane octets contiguous
    case 7: return (lane >> 4) * 1024 + (lane & 15) * W;       // 16 lanes contiguous
    default: return 0;
  }
}
template <int W>
__global__ __launch_bounds__(256) void k(const char *buf, float *out, int mode, int iters, unsigned mask) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned off = lane_off<W>(mode, lane) + wave * 65536u + blockIdx.x * 262144u;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned o = (off + j * 8192u) & mask;
      if constexpr (W == 4) acc += *reinterpret_cast<const float *>(buf + o);
      else if constexpr (W == 8) { f32x2 v = *reinterpret_cast<const f32x2 *>(buf + o); acc += v.x + v.y; }
      else { f32x4 v = *reinterpret_cast<const f32x4 *>(buf + o); acc += v.x + v.y + v.z + v.w; }
    }
    off += 64;                                                 // stays inside the window's lines mostly (cache-resident)
  }
  if (acc == 1.2345e-30f) out[0] = acc;
}
int main(int argc, char **argv) {
  const size_t bytes = (size_t)256 * 262144 * 2;
  char *buf; float *out;
  CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes)); CK(hipMalloc(&out, 4));
  const int iters = 512;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char *names[] = {"contiguous aligned", "contiguous +4 B", "lane quads contiguous", "lane pairs contiguous", "every lane own line", "34-lane runs, misaligned", "lane octets contiguous", "16 lanes contiguous"};
  for (int w : {4, 8, 16})
    for (int mode = 0; mode < 8; ++mode) {
      if (mode == 1 && w != 4) continue;
      const unsigned mask = 0x7ffffffu & ~(unsigned)(w - 1);
      auto go = [&]() {
        if (w == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, buf, out, mode, iters, mask);
        else if (w == 8) hipLaunchKernelGGL(k<8>, dim3(256), dim3(256), 0, 0, buf, out, mode, iters, mask);
        else hipLaunchKernelGGL(k<16>, dim3(256), dim3(256), 0, 0, buf, out, mode, iters, mask);
      };
      go(); go();
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < 5; ++r) go();
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
      const double instr_per_cu = 4.0 * iters * 8;             // wave-instructions per CU (one 4-wave workgroup per CU)
      const double cyc = ms * 1e-3 * 2.4e9;
      printf("W=%2d B/lane  %-28s %7.1f cycles per wave-instruction (at 2.4 GHz), %6.1f B/clk/CU\n", w, names[mode], cyc / instr_per_cu, 64.0 * w * instr_per_cu / cyc);
    }
  return 0;
}
