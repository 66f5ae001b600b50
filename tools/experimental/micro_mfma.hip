// What does the chip sustain on a pure stream of matrix instructions?  Eight independent accumulators per wave, in-place inline-assembly MFMAs (the
// compiler cannot rotate or copy the registers), 1 / 2 / 4 waves per SIMD.  Under rocprofv3 --pmc GRBM_GUI_ACTIVE the same run gives the clock.
//   hipcc --offload-arch=gfx950 -O3 -o tools/experimental/micro_mfma tools/experimental/micro_mfma.hip && tools/experimental/micro_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MF(op, acc, a, b) asm volatile(op " %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  const float t = seed * (float)(threadIdx.x % 7 + 1);
  const f32x4 av = {t, t * 0.5f, -t, t * 0.25f}, bv = {t * 0.125f, -t, t * 2.f, t};
  const f16x8 ha = __builtin_bit_cast(f16x8, av), hb = __builtin_bit_cast(f16x8, bv);
  const float fa = t, fb = -t * 0.5f;
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {
      MF("v_mfma_f32_16x16x32_f16", c0, ha, hb); MF("v_mfma_f32_16x16x32_f16", c1, ha, hb); MF("v_mfma_f32_16x16x32_f16", c2, ha, hb); MF("v_mfma_f32_16x16x32_f16", c3, ha, hb);
      MF("v_mfma_f32_16x16x32_f16", c4, ha, hb); MF("v_mfma_f32_16x16x32_f16", c5, ha, hb); MF("v_mfma_f32_16x16x32_f16", c6, ha, hb); MF("v_mfma_f32_16x16x32_f16", c7, ha, hb);
    } else if constexpr (KIND == 1) {
      MF("v_mfma_f32_16x16x32_bf16", c0, ha, hb); MF("v_mfma_f32_16x16x32_bf16", c1, ha, hb); MF("v_mfma_f32_16x16x32_bf16", c2, ha, hb); MF("v_mfma_f32_16x16x32_bf16", c3, ha, hb);
      MF("v_mfma_f32_16x16x32_bf16", c4, ha, hb); MF("v_mfma_f32_16x16x32_bf16", c5, ha, hb); MF("v_mfma_f32_16x16x32_bf16", c6, ha, hb); MF("v_mfma_f32_16x16x32_bf16", c7, ha, hb);
    } else {
      MF("v_mfma_f32_16x16x4_f32", c0, fa, fb); MF("v_mfma_f32_16x16x4_f32", c1, fa, fb); MF("v_mfma_f32_16x16x4_f32", c2, fa, fb); MF("v_mfma_f32_16x16x4_f32", c3, fa, fb);
      MF("v_mfma_f32_16x16x4_f32", c4, fa, fb); MF("v_mfma_f32_16x16x4_f32", c5, fa, fb); MF("v_mfma_f32_16x16x4_f32", c6, fa, fb); MF("v_mfma_f32_16x16x4_f32", c7, fa, fb);
    }
  }
  const f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

template <int KIND>
static void run(const char *name, double flops_per, int wgs, float seed) {
  float *d;
  (void)hipMalloc(&d, (size_t)wgs * 256 * 4);
  const int iters = 40000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(256), 0, 0, d, 1000, seed);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(256), 0, 0, d, iters, seed);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)wgs * 4 * iters * 8;
  printf("%-26s %d waves/SIMD, operands %s: %8.3f ms  %6.2f ns per MFMA per SIMD  %7.1f TFLOP/s chip\n", name, wgs / 256, seed == 0.f ? "zero  " : "random", ms,
         ms * 1e6 / (iters * 8.0 * (wgs / 256)), n * flops_per / (ms * 1e-3) / 1e12);
  (void)hipFree(d);
}
int main() {
  for (float seed : {0.f, 1.37f})
    for (int w : {256, 512, 1024}) {
      run<0>("v_mfma_f32_16x16x32_f16", 2.0 * 16 * 16 * 32, w, seed);
      run<1>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, w, seed);
      run<2>("v_mfma_f32_16x16x4_f32", 2.0 * 16 * 16 * 4, w, seed);
    }
  return 0;
}
