// Row GEMM  y[M,N] = act(scale * (x[M,K] . W[N,K]^T + bias) + shift) (+ res), second generation of tdf_dma_kernel
// (kernels_net.h) for the TDF blocks of ConvTDFNet (uvr_lib_v5/modules.py:57-74) and every other nn.Linear on the path.
//
// Same tile (128 x 64*NREP x 32), same unpadded XOR-swizzled LDS image, same fp32 MFMA fragment maps and the same
// accumulation order per output element as tdf_dma_kernel -- results are bit-identical -- but the stage loop is
// rebuilt around what the round-1 counters showed (72 % MFMA-busy on K = 384, 82 % on K = 3072):
//   * fragment reads are software-pipelined by hand: the ds_read_b128 of group g+1 are issued before the 48 MFMAs of
//     group g (hipcc had placed them 2-4 MFMAs before their s_waitcnt, exposing the LDS latency four times a stage);
//   * the global->LDS DMA of the next stage is addressed as SGPR base + one constant 32-bit lane offset
//     (`global_load_lds_dwordx4 v, s[..]`): no per-lane 64-bit pointer arithmetic, no predicates, no branches in the
//     stage loop -- row groups past M / N are clamped to the last valid group and masked in the epilogue;
//   * optionally PERSISTENT over the column tiles of one row tile (short-K layers, e.g. Linear(F/8 -> F)): the first
//     stage of column tile j+1 is in flight while tile j runs its epilogue, and the x tile stays hot in L2;
//   * optional start stagger of half a tile for every second workgroup, so that the two co-resident workgroups of a
//     CU do not reach their (MFMA-idle) epilogues together.
// Preconditions (else the launcher falls back to tdf_dma_kernel): K % 32 == 0, M % 8 == 0, N % 8 == 0, 16-byte aligned rows.
#pragma once
#include "kernels_net.h"

namespace asx {

// ABL (ASX_TDF2_ABL, measurement-only instantiations, results are garbage): 1 = no DMA after the first stage, 2 = no MFMA,
// 4 = no epilogue traffic
// ABL bit 4 (16): timeline probe -- workgroups < 4096 of a launch with K == 384 record s_memtime at start / first data /
// end of the K loop / end of the epilogue plus HW_ID and XCC_ID into asx_dbg_trace (read back by asx_debug_trace)
__device__ unsigned long long asx_dbg_trace[4096 * 8];

// BK_ = 32: 80 KB of LDS, two workgroups per CU.  BK_ = 16: 40 KB and half-length stages -- a THIRD co-resident workgroup when
// the registers allow it (launch bound 3: <= 168 VGPRs); rows are then 64 bytes (4 chunks) and the chunk swizzle is
// chunk ^ h((row >> 2) & 3), h = {0, 2, 3, 1}, which keeps every 16-lane group of ds_read_b128 on four distinct chunk slots.
template <int NREP, int MREP, int ABL = 0, int BK_ = 32>
__global__ __launch_bounds__(256, (BK_ == 16 ? 3 : 2)) void tdf2_kernel(TdfDmaArgs a, int tiles_per_wg, int stagger_bit) {
  constexpr int abl = ABL;
  const bool probe = ((abl & 16) != 0) && a.K == 384 && blockIdx.x < 4096 && threadIdx.x == 0;
  if (probe) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asx_dbg_trace[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime();
    asx_dbg_trace[blockIdx.x * 8 + 4] = hw;
    asx_dbg_trace[blockIdx.x * 8 + 5] = xcc;
    asx_dbg_trace[blockIdx.x * 8 + 6] = (unsigned long long)gridDim.x;
  }
  constexpr int BK = BK_, BM = 16 * MREP, BN = 64 * NREP, BUF = (BM + BN) * BK;
  constexpr int CPR = BK / 4, RP = 64 / CPR;         // 16-byte chunks per row; rows per 1-KiB DMA piece (8 or 16)
  constexpr int XPW = BM / RP / 4, WPW = BN / RP / 4;  // pieces per wave and stage
  constexpr int NKK = BK / 16, MG = MREP / 4, NG = NKK * MG;   // MFMA groups per stage: (kk) x (row quads)
  static_assert(BK == 32 || BK == 16, "BK");
  static_assert((abl & 8) == 0 || BK == 32, "the staged epilogue needs a 40 KB stage buffer");
  static_assert(BM % (4 * RP) == 0 && BN % (4 * RP) == 0 && MREP % 4 == 0, "tile shape");
  extern __shared__ float lds_f[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;

  const int nbn = (a.N + BN - 1) / BN;
  const int groups = nbn / tiles_per_wg;             // launcher: tiles_per_wg divides nbn
  // Tile -> workgroup map.  The dispatcher places block b on XCD b % 8, each XCD has its own 4 MiB L2, and the level-0 TDF
  // weights are 4.7 MB: with every XCD sweeping all column tiles the W tiles thrash L2 and each workgroup re-streams its
  // 295 KB from MALL / HBM (timeline probe: ~690 KB of L2-miss traffic per workgroup against 196 KB algorithmic, the launch
  // ran at the memory system's ~4.6 TB/s).  Column tiles are therefore PARTITIONED over the XCDs whenever the shapes allow:
  //   groups % 8 == 0 : XCD x owns column groups [x * groups / 8, (x + 1) * groups / 8) for every row tile;
  //   8 % groups == 0 : 8 / groups XCDs share one column group and split the row tiles between them;
  // so an XCD keeps only its slice of W hot, and the x tile of a row is read by the XCDs through MALL.
  int bg;
  int64_t bm_i;
  {
    const int nbm_i = (int)((a.M + BM - 1) / BM);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if ((gridDim.x & 7) == 0 && groups >= 8 && (groups & 7) == 0) {
      const int cx = groups >> 3;
      bg = xcd * cx + slot % cx;
      bm_i = slot / cx;
    } else if ((gridDim.x & 7) == 0 && groups < 8 && (8 % groups) == 0 && nbm_i % (8 / groups) == 0) {
      const int r = 8 / groups;
      bg = xcd / r;
      bm_i = (int64_t)slot * r + (xcd % r);
    } else {
      const int lid = xcd_remap(blockIdx.x, gridDim.x);
      bg = lid % groups;
      bm_i = lid / groups;
    }
  }
  const int64_t m0 = bm_i * BM;
  const int64_t lda = a.lda ? a.lda : a.K, ldy = a.ldy ? a.ldy : a.N, ldr = a.ldr ? a.ldr : a.N;
  const int nk = a.K / BK;
  const int total = nk * tiles_per_wg;

  if (stagger_bit >= 0 && ((blockIdx.x >> stagger_bit) & 1)) {
    // half a tile of MFMA time: nk stages x 192 MFMAs x 32 cycles / 2, in s_sleep units of 64 cycles
    const int naps = (nk * NREP * MREP * (BK / 4) * 32 / 2) / (64 * 100);
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(100);
  }

  // ---- DMA addressing: piece q = wave + 4 i covers tile rows RP q .. RP q + RP - 1; lane (lr, lp) fetches the 16-byte chunk
  // that lands in physical slot lp of row lr, i.e. logical chunk lp ^ g(row):
  //   BK 32: g(row) = (row >> 1) & 7 = (4 (wave & 1) + (lr >> 1)) & 7      BK 16: g(row) = h((row >> 2) & 3) = h((lr >> 2) & 3)
  auto hsw = [](int g) { return g == 0 ? 0 : (g == 1 ? 2 : (g == 2 ? 3 : 1)); };
  const int lr = lane / CPR, lp = lane % CPR;
  const int gsw = BK == 32 ? ((((wave & 1) << 2) + (lr >> 1)) & 7) : hsw((lr >> 2) & 3);
  const uint32_t voff_x = (uint32_t)((lr * lda + ((lp ^ gsw) << 2)) * 4);
  const uint32_t voff_w = (uint32_t)(((int64_t)lr * a.K + ((lp ^ gsw) << 2)) * 4);
  const char *xrow[XPW];
#pragma unroll
  for (int i = 0; i < XPW; ++i) {
    int64_t r = m0 + RP * (wave + 4 * i);
    r = r < a.M - RP ? r : a.M - RP;                  // clamp whole row groups (launcher: M % RP == 0); masked at the store
    xrow[i] = reinterpret_cast<const char *>(a.x) + r * lda * 4;
  }

  const char *wrow[WPW];
  auto set_wrows = [&](int n0) {                      // once per column tile
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      int r = n0 + RP * (wave + 4 * i);
      r = r < a.N - RP ? r : a.N - RP;
      wrow[i] = reinterpret_cast<const char *>(a.w) + ((int64_t)r * a.K) * 4;
    }
  };
  auto issue = [&](int ks, int buf) {
    float *xs = lds_f + buf * BUF;
    float *ws = xs + BM * BK;
    const int kb = ks * (BK * 4);
#pragma unroll
    for (int i = 0; i < XPW; ++i) ASX_GLDS16(xrow[i] + kb + voff_x, xs + (wave + 4 * i) * 256);
#pragma unroll
    for (int i = 0; i < WPW; ++i) ASX_GLDS16(wrow[i] + kb + voff_w, ws + (wave + 4 * i) * 256);
  };

  f32x4 acc[NREP][MREP];
#pragma unroll
  for (int n = 0; n < NREP; ++n)
#pragma unroll
    for (int m = 0; m < MREP; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int sw = BK == 32 ? ((li >> 1) & 7) : hsw((li >> 2) & 3);   // g(row) of fragment rows 16 t + li
  const int pc0 = ((lk ^ sw) << 2), pc1 = (((4 + lk) ^ sw) << 2);     // pc1 (second 16-float half of a stage) only when BK = 32
  const int wrow_f = (wave * 16 * NREP + li) * BK;
  const int xrw = li * BK;

  int it_ks = 0, it_tile = 0;                         // cursor of the stage being issued (runs one ahead)
  set_wrows((bg * tiles_per_wg) * BN);
  issue(0, 0);
  it_ks = 1;
  if (it_ks == nk) {
    it_ks = 0;
    it_tile = 1;
    if (tiles_per_wg > 1) set_wrows((bg * tiles_per_wg + 1) * BN);
  }

  int ks = 0, tile = 0;
  for (int it = 0; it < total; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (probe && it == 0) asx_dbg_trace[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memtime();
    const float *xs = lds_f + (it & 1) * BUF;
    const float *ws = xs + BM * BK;
    f32x4 wa[2][NREP], xb[2][4];
#pragma unroll
    for (int n = 0; n < NREP; ++n) wa[0][n] = *reinterpret_cast<const f32x4 *>(&ws[wrow_f + n * 16 * BK + pc0]);
#pragma unroll
    for (int m = 0; m < 4; ++m) xb[0][m] = *reinterpret_cast<const f32x4 *>(&xs[xrw + m * 16 * BK + pc0]);
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 < total) {
      if constexpr (!(abl & 1)) issue(it_ks, (it + 1) & 1);
      if (++it_ks == nk) {
        it_ks = 0;
        ++it_tile;
        if (it_tile < tiles_per_wg) set_wrows((bg * tiles_per_wg + it_tile) * BN);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int kk = g / MG, mg = (g % MG) * 4;
      // the group's first MFMA carries the (free) wait for operands that were requested a whole group ago; the requests
      // for group g + 1 follow it, so their latency hides under the group's remaining MFMAs
      // (it reads the LAST registers of the group's request batch: LDS returns in order, so one wait covers the batch)
      if constexpr ((abl & 2) != 0) {
        acc[NREP - 1][mg + 3] += wa[kk & 1][NREP - 1] + xb[g & 1][3];
      } else {
        acc[NREP - 1][mg + 3] = ASX_MFMA(wa[kk & 1][NREP - 1][0], xb[g & 1][3][0], acc[NREP - 1][mg + 3]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (g + 1 < NG) {
        const int kk1 = (g + 1) / MG, mg1 = ((g + 1) % MG) * 4;
        const int pc = kk1 ? pc1 : pc0;
        if (kk1 != kk) {
#pragma unroll
          for (int n = 0; n < NREP; ++n) wa[kk1 & 1][n] = *reinterpret_cast<const f32x4 *>(&ws[wrow_f + n * 16 * BK + pc]);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) xb[(g + 1) & 1][m] = *reinterpret_cast<const f32x4 *>(&xs[xrw + (mg1 + m) * 16 * BK + pc]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((abl & 2) != 0) {
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[n][mg + m] += wa[kk & 1][n] * xb[g & 1][m];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < NREP; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m)
              if (j || n != NREP - 1 || m != 3) acc[n][mg + m] = ASX_MFMA(wa[kk & 1][n][j], xb[g & 1][m][j], acc[n][mg + m]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++ks < nk) continue;
    ks = 0;
    if (probe && tile == 0) asx_dbg_trace[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memtime();

    // ---- epilogue of column tile `tile` (the next tile's first stage is already in flight) -------------------------------
    const int n0 = (bg * tiles_per_wg + tile) * BN;
    ++tile;
    const bool full = (m0 + BM <= a.M) && (n0 + BN <= a.N);
    if constexpr ((abl & 4) != 0) {
      float chk = 0.f;
#pragma unroll
      for (int n = 0; n < NREP; ++n)
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          chk += acc[n][m].x + acc[n][m].y + acc[n][m].z + acc[n][m].w;
          acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      if (chk == 1.2345e-30f) a.y[0] = chk;
    } else
    if (((abl & 8) != 0) && full) {
      // ABL bit 3: row-coalesced epilogue.  The accumulators of three 16-row groups at a time go through the stage buffer
      // that was just consumed ([48][BN + 4] floats, conflict-free for the 8-lane groups of ds_write_b128), and come back
      // one full output row per 48 consecutive lanes, so residual loads and stores are 768-byte row segments instead of
      // sixteen 64-byte pieces per instruction.  Same arithmetic per element, bit-identical results.
      float *st = lds_f + (it & 1) * BUF;
      constexpr int SW = BN + 4, PM = 3, C4 = BN / 4;
#pragma unroll
      for (int p0 = 0; p0 < MREP; p0 += PM) {
        constexpr int dummy = 0;
        (void)dummy;
        const int pm = (MREP - p0) < PM ? (MREP - p0) : PM;
        __syncthreads();
#pragma unroll
        for (int m = 0; m < PM; ++m) {
          if (m < pm) {
#pragma unroll
            for (int n = 0; n < NREP; ++n) {
              *reinterpret_cast<f32x4 *>(&st[(m * 16 + li) * SW + wave * 16 * NREP + n * 16 + lk * 4]) = acc[n][p0 + m];
              acc[n][p0 + m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
          }
        }
        __syncthreads();
        const int nvec = pm * 16 * C4;
        for (int idx = tid; idx < nvec; idx += 256) {
          const int row_l = idx / C4, c4 = idx - row_l * C4;
          const int64_t row = m0 + p0 * 16 + row_l;
          const int col = n0 + c4 * 4;
          const f32x4 v = *reinterpret_cast<const f32x4 *>(&st[row_l * SW + c4 * 4]);
          const f32x4 b4 = (a.bias != nullptr) ? *reinterpret_cast<const f32x4 *>(a.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
          const f32x4 r4 = (a.res != nullptr) ? *reinterpret_cast<const f32x4 *>(a.res + row * ldr + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
          const int c = (int)(((uint32_t)row / (uint32_t)a.T) % (uint32_t)a.C);
          const float sc = a.scale ? a.scale[c] : 1.f, sh = a.shift ? a.shift[c] : 0.f;
          f32x4 o;
          o.x = tdf_act(sc * (v.x + b4.x) + sh, a.relu) + r4.x;
          o.y = tdf_act(sc * (v.y + b4.y) + sh, a.relu) + r4.y;
          o.z = tdf_act(sc * (v.z + b4.z) + sh, a.relu) + r4.z;
          o.w = tdf_act(sc * (v.w + b4.w) + sh, a.relu) + r4.w;
          *reinterpret_cast<f32x4 *>(a.y + row * ldy + col) = o;
        }
      }
    } else
    if (full) {
      f32x4 bz[NREP];
#pragma unroll
      for (int n = 0; n < NREP; ++n) {
        const int col = n0 + wave * 16 * NREP + n * 16 + lk * 4;
        bz[n] = (a.bias != nullptr) ? *reinterpret_cast<const f32x4 *>(a.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      // The timeline probe (ASX_TDF2_ABL=16) showed this epilogue to be LATENCY-bound: with the residual fetched in dependent
      // load -> compute -> store rounds of 6 float4 per wave it took as long as the whole 12-stage K loop (~100 k cycles on
      // K = 384, i.e. ~2 B/clk per workgroup).  On the ReLU path (every TDF layer of ConvTDFNet) the residual of FOUR 16-row
      // groups is kept in flight (a register ring refilled as soon as a group has been stored; the fragment registers of the
      // K loop are dead here), addressed as scalar base + one 32-bit lane offset to keep address pairs out of the budget.  Other activations (erf /
      // tanh / exp bodies) keep rounds of two 16-row groups.
      const uint32_t voff_r = (uint32_t)((li * ldr + wave * 16 * NREP + lk * 4) * 4);
      const uint32_t voff_y = (uint32_t)((li * ldy + wave * 16 * NREP + lk * 4) * 4);
      if (a.relu == 1) {
        constexpr int RING = BK_ == 16 ? 1 : (MREP < 4 ? MREP : 4);   // 16-row groups whose residual is in flight at any time (16-float stages: 168-register budget)
        f32x4 rs[RING][NREP];
        auto fetch = [&](int m) {
          const char *rb = reinterpret_cast<const char *>(a.res) + ((m0 + m * 16) * ldr + n0) * 4;
#pragma unroll
          for (int n = 0; n < NREP; ++n)
            rs[m % RING][n] = (a.res == nullptr) ? (f32x4){0.f, 0.f, 0.f, 0.f}
                              : ((a.nt & 2) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(rb + voff_r + n * 64))
                                            : *reinterpret_cast<const f32x4 *>(rb + voff_r + n * 64));
        };
#pragma unroll
        for (int m = 0; m < RING; ++m) fetch(m);
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          const uint32_t row = (uint32_t)(m0 + m * 16 + li);                      // launcher: M < 2^31 on this kernel
          const int c = (int)((row / (uint32_t)a.T) % (uint32_t)a.C);
          const float sc = a.scale ? a.scale[c] : 1.f, sh = a.shift ? a.shift[c] : 0.f;
          char *yb = reinterpret_cast<char *>(a.y) + ((m0 + m * 16) * ldy + n0) * 4;
#pragma unroll
          for (int n = 0; n < NREP; ++n) {
            const f32x4 v = acc[n][m];
            const f32x4 r = rs[m % RING][n];
            f32x4 o;
            o.x = fmaxf(sc * (v.x + bz[n].x) + sh, 0.f) + r.x;
            o.y = fmaxf(sc * (v.y + bz[n].y) + sh, 0.f) + r.y;
            o.z = fmaxf(sc * (v.z + bz[n].z) + sh, 0.f) + r.z;
            o.w = fmaxf(sc * (v.w + bz[n].w) + sh, 0.f) + r.w;
            if (a.nt & 1) __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(yb + voff_y + n * 64));
            else *reinterpret_cast<f32x4 *>(yb + voff_y + n * 64) = o;
            acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
          if (m + RING < MREP) fetch(m + RING);        // refill the slot just consumed: RING groups stay in flight
        }
      } else if constexpr (BK_ != 16) {               // (the 16-float-stage build is launched for ReLU layers only: 168 registers)
#pragma unroll
        for (int mg = 0; mg < MREP; mg += 2) {
          float sc[2], sh[2], rw[2];
          f32x4 rs[2][NREP];
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const uint32_t row = (uint32_t)(m0 + (mg + m) * 16 + li);
            const int c = (int)((row / (uint32_t)a.T) % (uint32_t)a.C);
            sc[m] = a.scale ? a.scale[c] : 1.f;
            sh[m] = a.shift ? a.shift[c] : 0.f;
            rw[m] = a.rscale ? a.rscale[row] : 1.f;
            const char *rb = reinterpret_cast<const char *>(a.res) + ((m0 + (mg + m) * 16) * ldr + n0) * 4;
#pragma unroll
            for (int n = 0; n < NREP; ++n)
              rs[m][n] = (a.res != nullptr) ? *reinterpret_cast<const f32x4 *>(rb + voff_r + n * 64) : (f32x4){0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            char *yb = reinterpret_cast<char *>(a.y) + ((m0 + (mg + m) * 16) * ldy + n0) * 4;
#pragma unroll
            for (int n = 0; n < NREP; ++n) {
              const f32x4 v = acc[n][mg + m];
              f32x4 o;
              o.x = tdf_act(sc[m] * __fmaf_rn(v.x, rw[m], bz[n].x) + sh[m], a.relu) + rs[m][n].x;
              o.y = tdf_act(sc[m] * __fmaf_rn(v.y, rw[m], bz[n].y) + sh[m], a.relu) + rs[m][n].y;
              o.z = tdf_act(sc[m] * __fmaf_rn(v.z, rw[m], bz[n].z) + sh[m], a.relu) + rs[m][n].z;
              o.w = tdf_act(sc[m] * __fmaf_rn(v.w, rw[m], bz[n].w) + sh[m], a.relu) + rs[m][n].w;
              *reinterpret_cast<f32x4 *>(yb + voff_y + n * 64) =
                  tdf_rot4(a, o, m0 + (mg + m) * 16 + li, n0 + wave * 16 * NREP + n * 16 + lk * 4);
              acc[n][mg + m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < MREP; ++m) {
        const int64_t row = m0 + m * 16 + li;
        const bool rok = row < a.M;
        const int c = rok ? (int)(((uint32_t)row / (uint32_t)a.T) % (uint32_t)a.C) : 0;
        const float sc = a.scale ? a.scale[c] : 1.f, sh = a.shift ? a.shift[c] : 0.f;
        const float rw = (rok && a.rscale) ? a.rscale[row] : 1.f;
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
          const int col = n0 + wave * 16 * NREP + n * 16 + lk * 4;
          const f32x4 v = acc[n][m];
          acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (!rok || col >= a.N) continue;          // N % 8 == 0 and col % 4 == 0: a float4 is inside or outside as a whole
          const f32x4 b4 = (a.bias != nullptr) ? *reinterpret_cast<const f32x4 *>(a.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
          const f32x4 r4 = (a.res != nullptr) ? *reinterpret_cast<const f32x4 *>(a.res + row * ldr + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
          f32x4 o;
          o.x = tdf_act(sc * __fmaf_rn(v.x, rw, b4.x) + sh, a.relu) + r4.x;
          o.y = tdf_act(sc * __fmaf_rn(v.y, rw, b4.y) + sh, a.relu) + r4.y;
          o.z = tdf_act(sc * __fmaf_rn(v.z, rw, b4.z) + sh, a.relu) + r4.z;
          o.w = tdf_act(sc * __fmaf_rn(v.w, rw, b4.w) + sh, a.relu) + r4.w;
          *reinterpret_cast<f32x4 *>(a.y + row * ldy + col) = tdf_rot4(a, o, row, col);
        }
      }
    }
    if (probe && tile == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asx_dbg_trace[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memtime();
    }
  }
}

}  // namespace asx
