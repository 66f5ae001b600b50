// What does the tile fetch of conv3h_kernel cost by itself?  256 persistent workgroups of four waves walk the level-0 tensor like the
// kernel does and only LOAD (and optionally store an output tile per step), three steps of loads in flight, nothing else running.
//   pattern 0: vertical walk, 4 new rows x 10 aligned quads (40 columns: one full line + one quad of each neighbour line) per channel -- the kernel's
//   pattern 1: vertical walk, 4 new rows x  8 aligned quads (the strip's own line only: no halo columns)          -- what the edge lines cost
//   pattern 2: horizontal walk, 6 rows x 8 aligned quads (own line only; halo rows fetched, halo columns kept from the neighbour tiles)
//   pattern 3: as 0 with the 8-lane groups on ONE channel plane (lane quads contiguous)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/experimental/micro_fetch tools/experimental/micro_fetch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Args { const float *x; float *y; float *sink; int B, T, F, pattern, stores; };   // stores: 0 none, 1 the kernel's (16 channels x 64 B per instruction), 2 = 8 channels x 128 B per instruction, 3 = as 1 nontemporal, 4 = as 2 nontemporal, 5 = 1 KB contiguous per instruction (ceiling, wrong addresses)

__global__ __launch_bounds__(256) void k(Args a) {
  const int tid = threadIdx.x, wg = blockIdx.x;
  const int64_t TF = (int64_t)a.T * a.F;
  const unsigned plane_bytes = (unsigned)(48 * TF * 4);
  // item decode per pattern
  int row, cig, q; bool ok;
  if (a.pattern == 0) {
    if (tid < 160) { q = (tid & 1) + 2 * ((tid >> 3) % 5); cig = (tid >> 1) & 3; row = tid / 40; }
    else { const int u = tid - 160; q = (u & 1) + 2 * ((u >> 3) % 5); cig = 4 + ((u >> 1) & 1); row = ((u >> 2) & 1) + 2 * (u / 40); }
    ok = tid < 240;
  } else if (a.pattern == 3) {
    q = tid % 10; cig = (tid / 10) % 6; row = tid / 60; ok = tid < 240;
  } else if (a.pattern == 1) {
    q = 1 + (tid & 7); cig = (tid >> 3) % 6; row = tid / 48; ok = tid < 192;
  } else {
    q = 1 + (tid & 7); cig = (tid >> 3) % 6; row = tid / 48; ok = tid < 288;   // 6 rows: 288 items > 256 lanes: rows 0..4 here, row 5 by a second round below
  }
  const int goff = (int)(((int64_t)cig * 8 * a.T + row) * a.F + 4 * q);
  const int x = wg & 7, jw = wg >> 3;
  const int tilesT = a.T / 4, tilesF = a.F / 32, nbands = tilesF / 32;
  float acc = 0.f;
  f32x4 raw[3][8];
  f32x4 raw2[3][2];
  int item = x, step = 0;
  int nsteps = 0;
  for (int it = x; it < a.B * nbands; it += 8) nsteps += (a.pattern == 2 ? tilesF / nbands * 0 + 64 : tilesT);
  auto coords = [&](int s, int &b, int &t1, int &f0) {
    const int itn = x + 8 * (s / 64);
    b = itn / nbands;
    const int band = itn - b * nbands;
    if (a.pattern == 2) {   // horizontal: the XCD's 32 workgroups take 32 vertically adjacent 4-row bands of one half of T... (T / 4 = 64 row bands: 2 groups), walk along F: 64 steps per item = 2/3 of a row; good enough for a bandwidth test
      const int rb = (band & 1) * 32 + jw;        // row band
      t1 = rb * 4 - 1;
      f0 = ((band >> 1) * 64 + (s % 64)) * 32 % a.F;
    } else {
      t1 = (s % 64) * 4 + 1;
      f0 = (band * 32 + jw) * 32;
    }
  };
  auto fetch = [&](int set, int s) {
    int b, t1, f0;
    coords(s < nsteps ? s : 0, b, t1, f0);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x + (int64_t)b * 48 * TF), 0, plane_bytes, 0x00020000);
    const int org = t1 * a.F + (f0 - 4);
    const bool in = s < nsteps && ok && (unsigned)(t1 + row) < (unsigned)a.T && (unsigned)(f0 - 4 + 4 * q) < (unsigned)a.F;
    const unsigned vo = in ? (unsigned)(goff + org) * 4u : 0xfffffff0u;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      raw[set][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, (int)(j * TF * 4), 0));
    if (a.pattern == 2) {   // sixth row: 48 items (6 groups x 8 quads) on lanes 0..47, 8 loads each -> emulate with 2 loads on lanes 0..191
      const int r5q = 1 + (tid & 7), r5c = (tid >> 3) % 24;     // 24 (channel-pair) x 8 quads = 192 lanes, 2 planes each
      const bool in5 = s < nsteps && tid < 192 && (unsigned)(t1 + 5) < (unsigned)a.T;
      const unsigned vo5 = in5 ? (unsigned)((((int64_t)r5c * 2 * a.T + t1 + 5) * a.F) + f0 - 4 + 4 * r5q) * 4u : 0xfffffff0u;
      raw2[set][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo5, 0, 0));
      raw2[set][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo5, (int)(TF * 4), 0));
    }
  };
  auto consume = [&](int set) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += raw[set][j].x + raw[set][j].y + raw[set][j].z + raw[set][j].w;
    if (a.pattern == 2) acc += raw2[set][0].x + raw2[set][1].y;
  };
  auto store = [&](int s) {   // the consumer's stores: four waves = four rows, lane (li = channel of a tile, g): six float4 per lane
    if (s >= nsteps) return;
    int b, t1, f0;
    coords(s, b, t1, f0);
    const int wave = tid >> 6, li = tid & 15, g = (tid >> 4) & 3;
    const int tt = (a.pattern == 2 ? t1 + 1 : t1 - 1) + wave;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.y + (int64_t)b * 48 * TF, 0, ((unsigned)tt < (unsigned)a.T) ? plane_bytes : 0u, 0x00020000);
    if (a.stores == 1 || a.stores == 3) {
      const int vo = ((li * a.T + tt) * a.F + f0 + g * 4) * 4;
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          if (a.stores == 1) __builtin_amdgcn_raw_buffer_store_b128((u32x4){1u, 2u, 3u, (unsigned)s}, rs, vo, (int)((c * 16 * TF + qq * 16) * 4), 0);
          else __builtin_amdgcn_raw_buffer_store_b128((u32x4){1u, 2u, 3u, (unsigned)s}, rs, vo, (int)((c * 16 * TF + qq * 16) * 4), 2);
        }
    } else if (a.stores == 2 || a.stores == 4) {
      // lane l of a wave: channel (l >> 3) of eight, 16-byte chunk l & 7 of the row's 128 bytes; six instructions cover 48 channels
      const int l = tid & 63;
      const int vo = (((l >> 3) * a.T + tt) * a.F + f0 + (l & 7) * 4) * 4;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        if (a.stores == 2) __builtin_amdgcn_raw_buffer_store_b128((u32x4){1u, 2u, 3u, (unsigned)s}, rs, vo, (int)((c * 8 * TF) * 4), 0);
        else __builtin_amdgcn_raw_buffer_store_b128((u32x4){1u, 2u, 3u, (unsigned)s}, rs, vo, (int)((c * 8 * TF) * 4), 2);
      }
    } else {
      const int l = tid & 63;
      const int vo = ((wave * 6) * 256 + l * 4) * 4 + ((tt * a.F + f0) % (a.T * a.F - 8192)) * 4 * 0 + (int)((((int64_t)s * 4 + wave) * 6 * 1024) % (int64_t)(plane_bytes - 65536));
#pragma unroll
      for (int c = 0; c < 6; ++c) __builtin_amdgcn_raw_buffer_store_b128((u32x4){1u, 2u, 3u, (unsigned)s}, rs, vo, c * 1024, 0);
    }
  };
  (void)item; (void)step;
  fetch(0, 0);
  fetch(1, 1);
  for (int s = 0; s < nsteps; s += 3) {
    fetch(2, s + 2); consume(0); if (a.stores) store(s);
    fetch(0, s + 3); consume(1); if (a.stores) store(s + 1);
    fetch(1, s + 4); consume(2); if (a.stores) store(s + 2);
  }
  if (acc == 1.2345e-30f) a.sink[0] = acc;
}
int main(int argc, char **argv) {
  const int B = 55, T = 256, F = 3072;
  const size_t n = (size_t)B * 48 * T * F;
  float *x, *y, *sink;
  CK(hipMalloc(&x, n * 4 + 4096)); CK(hipMalloc(&y, n * 4 + 4096)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(x, 0, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int stores = 0; stores < 6; ++stores)
    for (int pattern = 0; pattern < 2; ++pattern) {
      Args a{x, y, sink, B, T, F, pattern, stores};
      hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, a);
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, a);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
      const double steps = 21.0 * 64;
      printf("pattern %d stores %d: %.3f ms per launch, %.2f us per step\n", pattern, stores, ms, ms * 1e3 / steps);
    }
  return 0;
}
