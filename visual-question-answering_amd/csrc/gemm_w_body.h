// Device body of the pre-split-weight GEMM (gemm_w.hip has the description, the weight-split kernel and the host
// side).  In a header so that the backward's combined GEMM launch (gemm_tn.hip) can run its tiles too.
#pragma once
#include "common.h"
#include "fused.h"

// Developer switch: non-temporal stores for the result.  By itself the GEMM gains (the fp32 output leaves in one burst when
// the workgroups of a round finish together: P_v-shaped 58.3 -> 54.7 us at two pieces, dQ-shaped 28.5 -> 26.4, width 3
// 86.7 -> 85.3), but the kernels that read the result next lose what the stores no longer leave in the caches: whole
// forward + backward 550 -> 560 us at N = 196, 271 -> 270 at N = 49 (tools/ab_libs.sh, same box, twice).  Off.
#ifndef GEMMW_NTSTORE
#define GEMMW_NTSTORE 0
#endif

namespace gw {

constexpr int BM = 128, BN = 128, BK = 32;    // (BN: the four-wave tile; gemm_w_body<.., 8> takes 256 columns)
constexpr int LDR = 40;                        // [row][k] bf16 image row stride (elements): conflict-free b128 reads
constexpr int kFragBytes = 1024;               // one fragment: 64 lanes x 8 bf16
constexpr int kChunkBytes = 3 * kFragBytes;    // the three pieces of one (column tile, 16-k step)

struct WArgs {
  const float* A; const float* a_ptrs[8]; long a_sz; int a_sm;
  int a_sk, a_mdiv; long a_sdiv;               // AM: element (m, k) at (m / a_mdiv) * a_sdiv + m % a_mdiv + k * a_sk
  int kband_n, kband_lo[3], kband_hi[3];       // kband_n > 0: columns [j kband_n, (j+1) kband_n) contract over k in [lo_j, hi_j)
  const void* Wf; unsigned wf_bytes;
  float* C; float* c_ptrs[8]; long c_sz; int c_sm;
  const float* bias_n; float oscale;
  float ascale;                                // on the accumulators (1, or 1 / kF16WScale for an FP16 weight image)
  int M, N, K, xcd_group;
  float* status;                               // (H) word [0] of the call's status words (fused.h kStatusHdr), or NULL
  const unsigned* rowbits;                     // (AM = false, four waves) fused.h WGemm.rowbits: the tiles run over the set rows alone
  const unsigned* rowcnt; int rowcnt_words;    // != NULL: the bitmap is written by workgroups of THIS launch; rowcnt[z] reaches
                                               // rowcnt_words when batch entry z's words are complete
};

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int LDT = BM + 32;                   // AM: [k][row] image row stride (conflict-free writes + transposed reads)

// AM = false: A rows contiguous along k ([row][k] images, one ds_read_b128 per fragment).
// AM = true : A contiguous along m -- channel-major image features [B, d, N] read in place (model.py:215-217),
//             row m = (sample, location) split by a_mdiv: [k][row] images, fragments through ds_read_b64_tr_b16.
// NP        : bf16 pieces per operand.  3: the exact split (six partial products, fp32-accurate).  2: hi + mid (16
//             significand bits per operand; the three products mid*hi, hi*mid, hi*hi; ~2^-16 relative per product) --
//             the backward's gradient GEMMs, whose error budget allows it (tests/test_split_emulation.py).
//             1: reduced-precision mode (COATTN_FLAG_BF16_PROJ, the apex-O1 analogue): operands rounded to bf16 (the hi
//             piece alone -- of A while it is staged, of the weight from its image), ONE MFMA per product.  The schedule
//             keeps its 24 slots per half step; the pieces that do not exist are neither computed, written, read nor
//             multiplied, and with two pieces the twelve MFMAs take every other slot.
// NW = 8    : 512-thread workgroups, tile 128 x 256 (the eight waves as 2 x 4): the A rows staged once serve twice the
//             columns -- for the single-product mode at d = 2048, where the L2 -> CU traffic of A re-read by every
//             column tile is the bound, not the MFMAs.
// H (NP = 2)  : the two pieces are FP16 (fused.h: split_pair_h, v_mfma_f32_32x32x16_f16) -- the forward's projections.
template <bool AM, int NP = 3, int NW = 4, bool H = false>
__device__ __forceinline__ void gemm_w_body(const WArgs& g, const int id, short* const smem, int* const rowmap = nullptr) {
  static_assert(NP >= 1 && NP <= 3, "pieces per operand");
  static_assert(!H || NP == 2, "FP16 pieces: two per operand");
  constexpr bool P1 = NP == 1;
  constexpr int WCN = NW / 2, BN = 64 * WCN, NT = 64 * NW, AP = 1024 / NT;   // waves per tile row, tile width, threads, A float4 per thread and step
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WCN, wc = wave % WCN, li = lane & 31, lh = lane >> 5;
  // XCD-aware tile order (as gemm.hip): the column tiles of one row tile share an XCD's L2
  // (with k bands the column tiles are taken from the right: the bands with the long contractions start first)
  const bool cmp = !AM && NW == 4 && rowmap != nullptr && g.rowbits != nullptr;     // compacted rows (below)
  int m0, n0, z, grp8 = 0;
  {
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    const int x = id & 7, slot = id >> 3;
    if (!g.xcd_group) {
      const int per = ntm * ntn;
      z = id / per;
      const int t = id % per;
      m0 = (t / ntn) * BM; n0 = (g.kband_n > 0 ? ntn - 1 - t % ntn : t % ntn) * BN;
    } else {
      const int per = ntn * ((ntm + 7) / 8);
      z = slot / per;
      const int t = slot % per;
      grp8 = (t / ntn) * 8;
      const int mt = grp8 + x;
      m0 = mt * BM; n0 = (g.kband_n > 0 ? ntn - 1 - t % ntn : t % ntn) * BN;
      if (mt >= ntm && !cmp) return;                 // (a compacted job picks its row tile below)
    }
  }
  // Compacted rows (g.rowbits): tile row r stands for the (m0 + r)-th row of A_z whose bit is set; rowmap[r] = that row, or a
  // value past every buffer when the list is shorter (loads read 0, stores are dropped by the row test).  Every workgroup
  // finds its own 128 rows: exclusive prefix sums of the <= 512 words' popcounts (one wave: 8 words per lane, a DPP-free
  // shuffle scan over the lanes) into LDS (the image area, not yet in use), then per row a binary search for its word and a
  // walk over that word's bits -- a few hundred cycles of a 35 us tile; a tile past the end of the list returns.
  // XCD balance: the set rows fill the row tiles 0 .. nact - 1, so of every 8 consecutive row tiles (one per XCD) only the first
  // nact % 8 of the last group are live -- on the SAME XCDs for every batch entry z (the levels share their pad structure): the
  // group's tiles are rotated by z * (nact % 8) XCDs, so the levels' extra tiles tile the ring instead of piling on XCDs 0, 1, ..
  // (unrotated, N = 49: XCD 0 had 36 live P_q tiles, XCD 7 24 -- three tiles on some CUs, the launch as long as the dense one).
  if (cmp) {
    const int words = (g.M + 31) / 32;
    const unsigned* bits = g.rowbits + (long)z * words;
    int* pre = reinterpret_cast<int*>(smem);                     // [512 + 1] exclusive prefix sums (words past the end: the total)
    if (wave == 0) {
      if (g.rowcnt) {
        // The bitmap comes from the first workgroups of this launch (launch_gemm_w): wait until batch entry z's words are all
        // written.  They are dispatched before every tile and wait for nothing, so the wait ends; the words are stored, counted
        // and read at agent scope (write-through stores acknowledged before the count, loads that bypass this XCD's L2): no
        // device fence on either side -- an acquire here would invalidate the L2 the running tiles stream their operands from.
        while (__hip_atomic_load(&g.rowcnt[z], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)g.rowcnt_words)
          __builtin_amdgcn_s_sleep(16);
      }
      int pcw[8], sum = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = 8 * lane + e;
        pcw[e] = i < words ? __builtin_popcount(__hip_atomic_load(&bits[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0;
        sum += pcw[e];
      }
      int incl = sum;                                            // inclusive scan of the lanes' sums
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
      }
      int run = incl - sum;
#pragma unroll
      for (int e = 0; e < 8; ++e) { pre[8 * lane + e] = run; run += pcw[e]; }
      if (lane == 63) pre[512] = run;
    }
    __syncthreads();
    if (g.xcd_group) {
      const int e = ((pre[512] + BM - 1) / BM) & 7;
      m0 = (grp8 + (((id & 7) - z * e) & 7)) * BM;
    }
    if (tid < BM) {
      const int c = m0 + tid;
      int row = 0x3fffffff;
      if (c < pre[512]) {
        int lo = 0, hi = 511;                                    // the last word whose prefix is <= c
#pragma unroll
        for (int it = 0; it < 9; ++it) {
          const int mid = (lo + hi + 1) >> 1;
          if (pre[mid] <= c) lo = mid; else hi = mid - 1;
        }
        unsigned b = __hip_atomic_load(&bits[lo], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int k = c - pre[lo]; k > 0; --k) b &= b - 1;        // drop the lower set bits
        row = 32 * lo + __builtin_ctz(b);
      }
      rowmap[tid] = row;
    }
    __syncthreads();
    if (rowmap[0] >= g.M) return;                                  // (uniform: the list ends before this tile)
  }
  auto tile_row = [&](const int r) { return cmp ? rowmap[r] : m0 + r; };
  const float* Ab = g.a_ptrs[0] ? g.a_ptrs[z & 7] : g.A + (long)z * g.a_sz;
  const long a_bytes = AM ? ((long)((g.M - 1) / g.a_mdiv) * g.a_sdiv + (g.a_mdiv - 1) + (long)(g.K - 1) * g.a_sk + 1) * 4
                          : ((long)(g.M - 1) * g.a_sm + g.K) * 4;
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(Ab, (unsigned)a_bytes);
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(g.Wf, g.wf_bytes);
  int s0 = 0, KS = g.K / BK;                     // K % 32 == 0 (host check)
  if (g.kband_n > 0) {                           // (bands are multiples of 128 columns and of 32 k: one band per tile)
    const int band = n0 / g.kband_n;
    s0 = g.kband_lo[band] / BK;
    KS = (g.kband_hi[band] - g.kband_lo[band]) / BK;
  }

  // A staging: 4 float4 per thread and step; a wave's load covers 8 rows x 128 B (whole lines)
  // (AM: float4 = 4 consecutive rows of one k; a wave's load covers 2 k x 512 B; a_mdiv % 4 == 0 keeps the 4 rows in one sample)
  int a_voff[AP], a_lds[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    if (AM) {
      const int k = (tid >> 5) + (NT / 32) * i, m = (tid & 31) * 4, row = m0 + m;
      a_voff[i] = row < g.M ? (int)(((long)(row / g.a_mdiv) * g.a_sdiv + row % g.a_mdiv + (long)k * g.a_sk) * 4) : 0x40000000;
      a_lds[i] = k * LDT + m;
    } else {
      const int m = (tid >> 3) + (NT / 8) * i, k = (tid & 7) * 4;
      const int arow = tile_row(m);
      a_voff[i] = arow < g.M ? (arow * g.a_sm + k) * 4 : 0x40000000;           // rows past M read 0
      a_lds[i] = m * LDR + k;
    }
  }
  const int a_kstep = AM ? BK * g.a_sk * 4 : BK * 4;                            // bytes per 32-k step
  // B fragments: tile j of this wave, chunk (nt, ks16) at ((nt * K/16 + ks16) * 3 + piece) * 1 KB
  int w_voff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int nt = (n0 + wc * 64) / 32 + j;
    w_voff[j] = nt * (g.K / 16) * kChunkBytes + lane * 16;                      // tiles past N lie outside the image: 0
  }
  // AM: transposed fragment read -- each 16-lane group fetches a 4 (k) x 16 (rows) block; lane 4q+p of the group
  // supplies the address of block row q, columns 4p..4p+3, and receives the 4 k of row (lane & 15)
  const int a_rd = AM ? (8 * lh + ((lane & 15) >> 2)) * LDT + 16 * ((lane >> 4) & 1) + 4 * (lane & 3) + wr * 64
                      : (wr * 64 + li) * LDR + 8 * lh;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- main loop, scheduled by hand -------------------------------------------------------------------------
  // A 32-k step is two half steps of 24 MFMAs; every MFMA is followed by a few fillers and a scheduling fence, so
  // that loads, LDS traffic and the split arithmetic of the NEXT data sit in the shadow of MFMAs:
  //   B fragments: ring of three half-step register sets; the loads of half h + 2 go out during half h
  //   A rows     : raw[i] holds step s + 1 during step s; each is re-requested (step s + 2) right after the split
  //                consumed it, the split pieces go to the other LDS image; one barrier per step (in half 1),
  //                after it the A fragments of the next step's first half are read
  //   A fragments: af[h] for half h; those of half 1 are read during half 0
  f32x4 raw[AP];
  // B ring: three half-step register sets (loads two half steps ahead).  The two-piece four-wave kernels keep two sets
  // (one half step ahead): with the 16 registers that frees they fit THREE workgroups on a CU (<= 170 VGPRs, 3 x 40 KB of
  // LDS), and a third wave per SIMD hides more than the deeper ring did -- the forward's projection launch 91.2 -> 84.3 us at
  // N = 196 (1,376 tiles: 1.8 rounds instead of 2.7), 46.3 -> 42.3 at N = 49 (644 tiles: one round instead of 1.26).
  // -DGEMMW_RB3 (developer switch, with -DGEMMW_OCC2=2): the three-set ring at two workgroups per CU.
#ifdef GEMMW_RB3
  constexpr int RB = 3;
#else
  constexpr int RB = (NP == 2 && NW == 4) ? 2 : 3;
#endif
  bf16x8 bq[RB][2][3];                           // [ring][tile j][piece]
  bf16x8 af[2][3][2];                            // [half][piece][tile i]
  unsigned ph[2], pm[2], pl[2];                  // packed pieces of the raw[i] being split (its two pairs)
  float ra[2], rb[2];
  float amax = 0.f;                              // (H) largest |value| this thread converted to FP16 pieces or stored (range report)
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};      // (gemm.hip's order)
  constexpr int RQ[3] = {2, 0, 1};               // fragment read order = order of first use
  constexpr int IMG = BM * LDR;                  // elements of one piece image
  auto load_a = [&](int i, int s) {
#ifdef GEMMW_NOA
    if (s >= 2) return;                          // developer switch: no A reloads past the prologue
#endif
    if (i < AP) raw[i] = buf_load4(rs_a, a_voff[i], (s + s0) * a_kstep);
  };
  auto load_b = [&](int ring, int k, int half) {
    const int j = k / 3, q = k % 3;
    if (q >= NP) return;
    bq[ring][j][q] = __builtin_bit_cast(bf16x8, buf_load4(rs_w, w_voff[j] + q * kFragBytes, (half + 2 * s0) * kChunkBytes));
  };
  auto read_a = [&](const short* img, int h, int k) {
    const int q = RQ[k >> 1], i = k & 1;
    if (q >= NP) return;
    if (AM) {
      const short* ptr = img + q * IMG + a_rd + i * 32 + 16 * h * LDT;
      const bf16x4 lo = lds_tr16(ptr), hi = lds_tr16(ptr + 4 * LDT);
      af[h][q][i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    } else {
      af[h][q][i] = *reinterpret_cast<const bf16x8*>(&img[q * IMG + a_rd + i * 32 * LDR + 16 * h]);
    }
  };
  // split of raw[i], pair e (0 | 1), in three stages of 5, 5 and 1 VALU instructions
  auto stage = [&](int i, int e, int st) {
    if (i >= AP) return;
    if (P1) {
      if (st == 0) ph[e] = cvt_pk_bf16(raw[i][2 * e], raw[i][2 * e + 1]);
      return;
    }
#ifdef GEMMW_NOSPLIT
    if (st == 0) ph[e] = pm[e] = pl[e] = cvt_pk_bf16(raw[i][2 * e], raw[i][2 * e + 1]);
#else
    if (H) {
      if (st == 0) {
        amax = fmaxf(amax, fmaxf(fabsf(raw[i][2 * e]), fabsf(raw[i][2 * e + 1])));   // one v_max3_f32 with |.| modifiers
        const hfv2 hh = __builtin_convertvector((f32x2{raw[i][2 * e], raw[i][2 * e + 1]}), hfv2);
        ph[e] = __builtin_bit_cast(unsigned, hh);
        ra[e] = sub1(raw[i][2 * e], (float)hh[0]);
        rb[e] = sub1(raw[i][2 * e + 1], (float)hh[1]);
      } else if (st == 1) {
        pm[e] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{ra[e], rb[e]}), hfv2));
      }
      return;
    }
    if (st == 0) {
      ph[e] = cvt_pk_bf16(raw[i][2 * e], raw[i][2 * e + 1]);
      ra[e] = sub1(raw[i][2 * e], __builtin_bit_cast(float, ph[e] << 16));
      rb[e] = sub1(raw[i][2 * e + 1], __builtin_bit_cast(float, ph[e] & 0xffff0000u));
    } else if (st == 1) {
      pm[e] = cvt_pk_bf16(ra[e], rb[e]);
      if (NP == 2) return;
      ra[e] = sub1(ra[e], __builtin_bit_cast(float, pm[e] << 16));
      rb[e] = sub1(rb[e], __builtin_bit_cast(float, pm[e] & 0xffff0000u));
    } else if (NP == 3) {
      pl[e] = cvt_pk_bf16(ra[e], rb[e]);
    }
#endif
  };
  auto write_a = [&](short* img, int i, int q) {
    if (i >= AP) return;
    if (q >= NP) return;
    const u32x2 v = q == 0 ? u32x2{ph[0], ph[1]} : (q == 1 ? u32x2{pm[0], pm[1]} : u32x2{pl[0], pl[1]});
    *reinterpret_cast<u32x2*>(&img[q * IMG + a_lds[i]]) = v;
  };
  // half step HH of step s: MFMAs on af[HH] x bq[BU]; the loads of half 2 s + HH + 2 go to bq[BL]
  auto half = [&](auto HHc, auto BUc, auto BLc, int s, const short* cur, short* nxt) {
    constexpr int HH = decltype(HHc)::value, BU = decltype(BUc)::value, BL = decltype(BLc)::value;
#pragma unroll
    for (int n = 0; n < 24; ++n) {
      // the MFMA of slot n: all 24 (NP = 3), every other slot (NP = 2: products 3 .. 5), the last four (NP = 1)
      const int mi = NP == 2 ? n >> 1 : n;
      const bool mf = NP == 3 || (NP == 2 && (n & 1)) || (NP == 1 && n >= 20);
      const int t = NP == 2 ? 3 + (mi >> 2) : n >> 2, i = (mi >> 1) & 1, j = mi & 1;
#ifndef GEMMW_NOMFMA
      if (mf && H) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hfv8, af[HH][PA[t]][i]), __builtin_bit_cast(hfv8, bq[BU][j][PB[t]]), acc[i][j], 0, 0, 0);
      if (mf && !H) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[HH][PA[t]][i], bq[BU][j][PB[t]], acc[i][j], 0, 0, 0);
#else
      if (n < 4) acc[i][j][0] += __builtin_bit_cast(float, (int)af[HH][PA[t]][i][0] ^ (int)bq[BU][j][PB[t]][0]);
#endif
#ifndef GEMMW_NOB
      if (n < 6) load_b(BL, n, 2 * s + HH + RB - 1);
#endif
      if (HH == 0) {
        if (n < 6) read_a(cur, 1, n);
        // raw[0]: stages in slots 6..11, pieces written in 12..14; raw[1]: 12..17 and 18..20
        if (n >= 6 && n < 12) stage(0, (n - 6) / 3, (n - 6) % 3);
        if (n >= 12 && n < 15) write_a(nxt, 0, n - 12);
        if (n == 12) load_a(0, s + 2);
        if (n >= 12 && n < 18) stage(1, (n - 12) / 3, (n - 12) % 3);
        if (n >= 18 && n < 21) write_a(nxt, 1, n - 18);
        if (n == 18) load_a(1, s + 2);
      } else {
        if (n < 6) stage(2, n / 3, n % 3);
        if (n >= 6 && n < 9) write_a(nxt, 2, n - 6);
        if (n == 6) load_a(2, s + 2);
        if (n >= 6 && n < 12) stage(3, (n - 6) / 3, (n - 6) % 3);
        if (n >= 12 && n < 15) write_a(nxt, 3, n - 12);
        if (n == 12) load_a(3, s + 2);
        if (n == 16) lds_barrier();
        if (n >= 17 && n < 23) read_a(nxt, 0, n - 17);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // prologue: step 0 split into image 0, raw = step 1, B halves 0 and 1 in flight, fragments of half 0 read
#pragma unroll
  for (int i = 0; i < AP; ++i) load_a(i, 0);
#pragma unroll
  for (int k = 0; k < 6; ++k) { load_b(0, k, 0); if (RB == 3) load_b(1, k, 1); }
  short* const img0 = smem;
  short* const img1 = smem + NP * IMG;               // (only the pieces of the width are staged)
#pragma unroll
  for (int i = 0; i < AP; ++i) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int st = 0; st < 3; ++st) stage(i, e, st);
#pragma unroll
    for (int q = 0; q < 3; ++q) write_a(img0, i, q);
    load_a(i, 1);
  }
  lds_barrier();
#pragma unroll
  for (int k = 0; k < 6; ++k) read_a(img0, 0, k);
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  auto step = [&](auto U0, auto U1, auto U2, int s) {      // ring sets: half 0 uses U0 (loads U2), half 1 uses U1 (loads U0)
    const short* cur = (s & 1) ? img1 : img0;
    short* nxt = (s & 1) ? img0 : img1;
    half(I0{}, U0, U2, s, cur, nxt);
    half(I1{}, U1, U0, s, cur, nxt);
  };
  int s = 0;
  if constexpr (RB == 3) {
    for (; s + 3 <= KS; s += 3) {                          // (one loop exit: the accumulators stay in place)
      step(I0{}, I1{}, I2{}, s);
      step(I2{}, I0{}, I1{}, s + 1);
      step(I1{}, I2{}, I0{}, s + 2);
    }
    if (s < KS) {
      step(I0{}, I1{}, I2{}, s);
      if (s + 1 < KS) step(I2{}, I0{}, I1{}, s + 1);
    }
  } else {                                                 // two sets: half 0 uses set 0 and loads set 1, half 1 the reverse
    for (; s < KS; ++s) step(I0{}, I1{}, I1{}, s);
  }

  float* Cb = g.c_ptrs[0] ? g.c_ptrs[z & 7] : g.C + (long)z * g.c_sz;
  float bn[2];
  int col[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    col[j] = n0 + wc * 64 + j * 32 + li;
    bn[j] = (g.bias_n && col[j] < g.N) ? g.bias_n[col[j]] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tile_row(wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
      if (row >= g.M) continue;
      float* crow = Cb + (long)row * g.c_sm;
#ifdef GEMMW_NOSTORE
      if (acc[i][0][r] != 12345.678f) continue;  // developer switch: no C stores
#endif
#pragma unroll
      for (int j = 0; j < 2; ++j)
#if GEMMW_NTSTORE
        if (col[j] < g.N) __builtin_nontemporal_store((acc[i][j][r] + bn[j]) * g.oscale, &crow[col[j]]);
#else
        if (col[j] < g.N) {
          const float y = H ? fmaf(acc[i][j][r], g.ascale, bn[j]) * g.oscale : (acc[i][j][r] + bn[j]) * g.oscale;
          if (H) amax = fmaxf(amax, fabsf(y));   // (the stored projection is an FP16-piece operand of the fused kernels)
          crow[col[j]] = y;
        }
#endif
    }
  // Range report of the tolerance mode: a wave that met a magnitude beyond the exact-piece range raises the call's status
  // word (positive floats order like their bit patterns) -- the rare case; every other wave pays one compare.
  if (H && g.status && __builtin_amdgcn_ballot_w64(!(amax <= kF16Exact)) != 0) {
    amax = wave_max(amax);
    if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(g.status), __builtin_bit_cast(unsigned, amax));
  }
}

}  // namespace gw

// host side (gemm_w.hip): checks `d`, fills the kernel arguments and the number of workgroups
int gemm_w_fill_job(const WGemm& d, gw::WArgs& g, long* nblk, int bn = gw::BN);
