// The forward's projections in the tolerance mode (COATTN_FLAG_FAST16): fp32 activations x nn.Linear weight on two FP16
// pieces per operand (fused.h: 22 significand bits, three partial products on v_mfma_f32_32x32x16_f16), gfx950:
//
//   C[z][m][n] = ((sum_k A[z][m][k] * Bw(k, n)) / kF16WScale + bias_n[n]) * out_scale        A rows contiguous along k
//
// P_v = V W_v^T + b_v from location-major image features and P_q = Q W_q^T + b_q of the three levels (model.py:380-384).
//
// Why its own kernel (round 5).  gemm_w.hip's 128 x 128 tile feeds every wave its own B fragments from L2 and re-reads the
// A rows once per column tile: 48 KB through the CU's vector-memory path per 32-k step for 96 MFMAs = 512 B per MFMA.  With
// three MFMAs per product instead of six that path (64 B/clk per CU) is what binds: at the full matrix rate the tile would
// need 64 B/clk -- the kernel sat at 0.31-0.33 of its bound with the matrix pipe idle two thirds of the time (DESIGN.md
// section 3.2: knock-outs additive).  Here:
//   * tile 128 x 256 per 512-thread workgroup (8 waves as 2 x 4, 64 x 64 = 2 x 2 MFMA tiles each), 32 k per step:
//     A 16 KB + B 32 KB per step for 192 MFMAs = 256 B per MFMA, half of gemm_w's;
//   * B: the pre-split FP16 weight image (wsplit, pieces = 16: 1 KB per (32-column tile, 16-k step, piece), lane-major) goes
//     global -> LDS by LDS-DMA (no registers, no VALU), ONCE per workgroup, into a ring of three 32 KB buffers, two steps ahead;
//     the two waves that share a column read the fragments back lane-linear (conflict-free ds_read_b128);
//   * A: fp32 rows in whole 128-byte lines -> registers -> two FP16 pieces (6 VALU per pair, range tracked for coattn_status)
//     -> LDS [128][32 + 8] per piece, double-buffered, requested one step ahead;
//   * one barrier per step, inside the MFMA stream; every MFMA is followed by its share of the other work and a scheduling
//     fence (left alone hipcc clusters the staging behind the MFMAs, where both waves of a SIMD do it at the same time).
// LDS: 2 x 20 KB (A) + 3 x 32 KB (B) = 136 KB: one workgroup per CU, two waves per SIMD.
// Shapes: N % 256 == 0, K % 32 == 0, A rows 16-byte aligned, no k bands; everything else stays on gemm_w.hip.
// Ordering of the LDS-DMA ring (DESIGN.md "LDS-DMA rings"): a buffer is refilled two barriers after its last read (WAR), and
// read one barrier after the counted s_waitcnt vmcnt(6) that retires its DMA in the issuing wave (RAW).
#include "common.h"
#include "fused.h"
#include "gemm_w_body.h"
#include <type_traits>

#ifndef GEMMH2_STAMPS
#define GEMMH2_STAMPS 0    // diagnostic build (tools/probe_h2_stamps.py): wave 0 writes the shader clock at every half step of the
#endif                     // persistent kernel into the unused third KB of weight-image chunk blockIdx.x (128 stamps per workgroup)
#ifndef GEMMH2_KO
#define GEMMH2_KO 0        // developer knock-outs (wrong results): 1 no A reloads, 2 no weight DMA, 4 no MFMAs, 8 no split / LDS writes, 16 no C stores
#endif

namespace {

constexpr int HM = 128, HN = 256, HK = 32;
constexpr int LDA = HK + 8;                       // halfs per staged A row (80 B: conflict-free ds_read_b128, gemm_w_body.h)
constexpr int A_IMG = HM * LDA;                   // halfs of one piece image (10,240 B)
constexpr int A_BUF = 2 * A_IMG;                  // both pieces
constexpr int B_BUF = (HN / 32) * 2 * 2 * 512;    // halfs: 8 column tiles x 2 k-steps of 16 x 2 pieces x 1 KB
constexpr int kLds = (2 * A_BUF + 3 * B_BUF) * 2; // bytes: 139,264
constexpr int kLdsP = 2 * A_BUF * 2;              // the persistent kernel keeps only the A images in LDS: 40,960 B
constexpr int kChunk3 = 3 * 1024;                 // the weight image keeps gemm_w's 3 KB chunk stride (the third KB is unused)

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;

struct H2Jobs { gw::WArgs job[2]; int first1; };

__device__ __forceinline__ void gemm_h2_body(const gw::WArgs& g, const int id, short* const smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, li = lane & 31, lh = lane >> 5;
  // XCD-aware order: the column tiles of a row tile run on one XCD, one after the other (they re-read the same A rows)
  const int ntm = (g.M + HM - 1) / HM, ntn = g.N / HN;
  const int x = id & 7, slot = id >> 3, per = ntn * ((ntm + 7) / 8);
  const int z = slot / per, t = slot % per, mt = (t / ntn) * 8 + x;
  if (mt >= ntm) return;
  const int m0 = mt * HM, n0 = (t % ntn) * HN;
  const float* Ab = g.a_ptrs[0] ? g.a_ptrs[z & 7] : g.A + (long)z * g.a_sz;
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(Ab, (unsigned)(((long)(g.M - 1) * g.a_sm + g.K) * 4));
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(g.Wf, g.wf_bytes);
  const int KS = g.K / HK;

  short* const abuf0 = smem;
  short* const abuf1 = smem + A_BUF;
  short* const bbase = smem + 2 * A_BUF;

  // A staging: 2 float4 per thread and step; a wave's load covers 8 rows x 128 B (whole lines)
  const int a_row = tid >> 3, a_k = (tid & 7) * 4;
  int a_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) a_voff[i] = (m0 + a_row + 64 * i) < g.M ? ((m0 + a_row + 64 * i) * g.a_sm + a_k) * 4 : 0x40000000;   // rows past M read 0
  const int a_wr = a_row * LDA + a_k;             // + 64 i rows
  const int a_rd = (wr * 64 + li) * LDA + 8 * lh; // + 32 i rows, + 16 h
  // B: wave w moves column tile w of a step: its (k-step of 16, piece) chunks, 1 KB each, lane-linear
  const int b_src = (n0 / 32 + wave) * (g.K / 16) * kChunk3;          // + (2 s + h) * 3 KB + q * 1 KB; lane part in the vector offset
  const int b_dst = wave * 4 * 512;                                    // halfs inside a B buffer: ((ct * 2 + h) * 2 + q) * 512
  const int b_rd = (wc * 2) * 4 * 512 + lane * 8;                      // + ((j * 2 + h) * 2 + q) * 512

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 raw[2];
  bf16x8 af[2][2][2], bq[2][2][2];                // [fragment set][piece][tile]
  unsigned ph[2], pm[2];
  float ra[2], rb[2];
  float amax = 0.f;                               // largest |value| converted to FP16 pieces or stored (coattn_status)

  auto load_a = [&](int i, int s) {
    if ((GEMMH2_KO & 1) && s >= 2) return;
    raw[i] = buf_load4(rs_a, a_voff[i], s * HK * 4);
  };
  auto dma_b = [&](int c, int s, short* bbuf) {   // chunk c = h * 2 + q of step s -> the ring buffer `bbuf` (= s % 3)
    if (GEMMH2_KO & 2) return;
    short* dst = bbuf + b_dst + c * 512;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)dst, 16, lane * 16, b_src + (2 * s + (c >> 1)) * kChunk3 + (c & 1) * 1024, 0, 0);
  };
  // split of raw[i], pair e, in two stages (5 + 1 VALU, + the range maximum)
  auto stage = [&](int i, int e, int st) {
    if (GEMMH2_KO & 8) { if (st == 0) ph[e] = pm[e] = __builtin_bit_cast(unsigned, raw[i][2 * e]); return; }
    if (st == 0) {
      amax = fmaxf(amax, fmaxf(fabsf(raw[i][2 * e]), fabsf(raw[i][2 * e + 1])));
      const hfv2 hh = __builtin_convertvector((f32x2{raw[i][2 * e], raw[i][2 * e + 1]}), hfv2);
      ph[e] = __builtin_bit_cast(unsigned, hh);
      ra[e] = sub1(raw[i][2 * e], (float)hh[0]);
      rb[e] = sub1(raw[i][2 * e + 1], (float)hh[1]);
    } else {
      pm[e] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{ra[e], rb[e]}), hfv2));
    }
  };
  auto write_a = [&](short* img, int i, int q) {
    if (GEMMH2_KO & 8) return;
    *reinterpret_cast<u32x2*>(&img[q * A_IMG + 64 * i * LDA + a_wr]) = q == 0 ? u32x2{ph[0], ph[1]} : u32x2{pm[0], pm[1]};
  };
  auto read_af = [&](auto SETc, const short* img, int h, int k) {      // k = q * 2 + i
    constexpr int SET = decltype(SETc)::value;
    const int q = k >> 1, i = k & 1;
    af[SET][q][i] = *reinterpret_cast<const bf16x8*>(&img[q * A_IMG + a_rd + i * 32 * LDA + 16 * h]);
  };
  auto read_bq = [&](auto SETc, const short* bb, int h, int k) {       // k = q * 2 + j
    constexpr int SET = decltype(SETc)::value;
    const int q = k >> 1, j = k & 1;
    bq[SET][q][j] = *reinterpret_cast<const bf16x8*>(&bb[b_rd + ((j * 2 + h) * 2 + q) * 512]);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // the twelve MFMAs of a half step (smallest partial product first: lo * hi, hi * lo, hi * hi), each followed by fillers
  auto mfma_slot = [&](auto SETc, int n) {
    constexpr int SET = decltype(SETc)::value;
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
    const int tp = n >> 2, i = (n >> 1) & 1, j = n & 1;
    if (!(GEMMH2_KO & 4))
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hfv8, af[SET][PA[tp]][i]), __builtin_bit_cast(hfv8, bq[SET][PB[tp]][j]), acc[i][j], 0, 0, 0);
    else if (n == 0) acc[i][j][0] += __builtin_bit_cast(float, (int)af[SET][0][i][0] ^ (int)bq[SET][0][j][0]);
  };

  // ---- prologue: steps 0 and 1 of B in flight, step 0 of A split into image 0, step 1 of A requested -------------------
  load_a(0, 0); load_a(1, 0);
#pragma unroll
  for (int c = 0; c < 4; ++c) dma_b(c, 0, bbase);
  if (KS > 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) dma_b(c, 1, bbase + B_BUF);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 2; ++e) { stage(i, e, 0); stage(i, e, 1); }
    write_a(abuf0, i, 0); write_a(abuf0, i, 1);
  }
  if (KS > 1) { load_a(0, 1); load_a(1, 1); }
  // (every DMA above is older than the two row loads just issued: all but the two youngest operations retired)
  if (KS > 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int k = 0; k < 4; ++k) { read_af(I0{}, abuf0, 0, k); read_bq(I0{}, bbase, 0, k); }

  // One step; W1 / W2: steps s + 1 / s + 2 exist (compile-time: the last two steps are peeled, the loop body has no branches).
  // acur / anxt: the A images of this and the next step; bcur / bnxt / bdma: the B ring buffers of steps s, s + 1, s + 2.
  auto step = [&](auto W1c, auto W2c, int s, const short* acur, short* anxt, const short* bcur, const short* bnxt, short* bdma) {
    constexpr bool w1 = decltype(W1c)::value, w2 = decltype(W2c)::value;
    // half 0: MFMAs on fragment set 0; the fragments of half 1 are read; the rows of step s + 1 (requested one step ago) are
    // split and written to the other A image
#pragma unroll
    for (int n = 0; n < 12; ++n) {
      mfma_slot(I0{}, n);
      if (n < 4) read_af(I1{}, acur, 1, n);
      if (n >= 4 && n < 8) read_bq(I1{}, bcur, 1, n - 4);
      if (w1) {
        if (n == 4) stage(0, 0, 0);
        if (n == 5) { stage(0, 0, 1); stage(0, 1, 0); }
        if (n == 6) { stage(0, 1, 1); write_a(anxt, 0, 0); }
        if (n == 7) { write_a(anxt, 0, 1); stage(1, 0, 0); }
        if (n == 8) { stage(1, 0, 1); stage(1, 1, 0); }
        if (n == 9) { stage(1, 1, 1); write_a(anxt, 1, 0); }
        if (n == 10) write_a(anxt, 1, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // half 1: MFMAs on fragment set 1; the weight chunks of step s + 2 go out (their ring buffer was last read in step s - 1,
    // two barriers ago), then the rows of step s + 2; the barrier; the fragments of step s + 1's first half
#pragma unroll
    for (int n = 0; n < 12; ++n) {
      mfma_slot(I1{}, n);
      if (n < 4 && w2) dma_b(n, s + 2, bdma);
      if (n == 4 && w2) load_a(0, s + 2);
      if (n == 5 && w2) load_a(1, s + 2);
      if (n == 7 && w1) {
        // Everything this wave wrote for step s + 1 has landed -- its A pieces (lgkmcnt) and its share of the weight chunks
        // of step s + 1, issued one step ago: all but the six youngest vector-memory operations (the four DMAs and two row
        // loads of THIS half step; none in the step before the last) have retired -- then all waves meet.  The reads come
        // after the barrier, never before it.
        if (w2) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (n >= 8 && n < 10 && w1) {
        read_af(I0{}, anxt, 0, 2 * (n - 8)); read_af(I0{}, anxt, 0, 2 * (n - 8) + 1);
      }
      if (n >= 10 && w1) {
        read_bq(I0{}, bnxt, 0, 2 * (n - 10)); read_bq(I0{}, bnxt, 0, 2 * (n - 10) + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using TT = std::true_type;
  using FF = std::false_type;
  short* a0 = abuf0; short* a1 = abuf1;
  short* b0 = bbase; short* b1 = bbase + B_BUF; short* b2 = bbase + 2 * B_BUF;
  int s = 0;
  for (; s + 2 < KS; ++s) {
    step(TT{}, TT{}, s, a0, a1, b0, b1, b2);
    short* ta = a0; a0 = a1; a1 = ta;
    short* tb = b0; b0 = b1; b1 = b2; b2 = tb;
  }
  if (s + 1 < KS) {
    step(TT{}, FF{}, s, a0, a1, b0, b1, b2);
    short* ta = a0; a0 = a1; a1 = ta;
    b0 = b1;
    ++s;
  }
  step(FF{}, FF{}, s, a0, a1, b0, b1, b2);

  float* Cb = g.c_ptrs[0] ? g.c_ptrs[z & 7] : g.C + (long)z * g.c_sz;
  float bn[2];
  int col[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    col[j] = n0 + wc * 64 + j * 32 + li;
    bn[j] = g.bias_n ? g.bias_n[col[j]] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row >= g.M) continue;
      if ((GEMMH2_KO & 16) && acc[i][0][r] != 12345.678f) continue;
      float* crow = Cb + (long)row * g.c_sm;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float y = fmaf(acc[i][j][r], g.ascale, bn[j]) * g.oscale;
        amax = fmaxf(amax, fabsf(y));             // (the stored projection is an FP16-piece operand of the fused kernels)
        crow[col[j]] = y;
      }
    }
  // range report of the tolerance mode (as gemm_w_body.h): the rare wave that met a magnitude beyond the exact-piece range
  if (g.status && __builtin_amdgcn_ballot_w64(!(amax <= kF16Exact)) != 0) {
    amax = wave_max(amax);
    if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(g.status), __builtin_bit_cast(unsigned, amax));
  }
}

__global__ __launch_bounds__(512, 2) void gemm_h2_kernel(const H2Jobs jobs) {
  extern __shared__ __attribute__((aligned(16))) short h2_smem[];
  f16_saturating_conversions();
  if ((int)blockIdx.x < jobs.first1) gemm_h2_body(jobs.job[0], (int)blockIdx.x, h2_smem);
  else gemm_h2_body(jobs.job[1], (int)blockIdx.x - jobs.first1, h2_smem);
}


// ---- the same tiles as ONE software pipeline per CU: persistent workgroups (K % 512 == 0) ----------------------------------
// Knock-outs of the kernel above at the P_v shape (tools/ab_gemmh2.sh; 56 us on that box) came out ADDITIVE: MFMAs 22.5 us,
// result stores 10.8, split + LDS writes 6.8, row loads 3.6, weight DMA 3.8, everything off 9.4 -- with one workgroup per CU
// and every CU in the same phase, the chip loads (prologue), computes, and writes its 32 MB of results (epilogue) one after the
// other.  Here a workgroup walks its tiles (id, id + grid, ...) as one uninterrupted stream of K-steps:
//   * the loads run AHEAD of the tile boundary -- steps 14 and 15 of a tile request the first rows and weight chunks of the
//     NEXT tile: no prologue after the first tile;
//   * the result of a tile is stored DURING the next tile's steps -- two accumulator sets alternate, four stores per step of
//     the set that was finished (64 per thread = 16 steps): the writes are spread over the whole kernel instead of arriving in
//     bursts, and nothing waits for them (vmcnt retires in order: a store is older than the rows requested in the same step,
//     which are consumed a step later).  Stores go through a buffer descriptor of the tile's rows: rows past M are dropped.
// The 16 steps of a 512-deep pass are unrolled (the deferred stores name their registers), the LDS rings rotate at run time.
struct H2Load {                                    // a tile's loader side: its A rows and its weight column tiles
  __amdgpu_buffer_rsrc_t rs_a, rs_w;
  int a_voff[2];                                   // per thread: byte offset of its two staged rows' first float4 (or out of range)
  int b_src;                                       // byte offset of this wave's column tile in the weight image
};
struct H2Store {                                   // a tile's store side
  __amdgpu_buffer_rsrc_t rs_c;
  int c_voff, c_sm4;                               // per thread: byte offset of its first result element; row stride in bytes
  float bn0, bn1, oscale;
};

// Tile `id` of the launch -> (job, batch index z, first row m0, first column n0); false for the padding ids of the XCD-aware
// order (no such row tile).  The job table is read through the kernel-argument segment's address (a dynamic index into the
// by-value argument would be copied to scratch memory).
__device__ __forceinline__ bool h2_decode(const H2Jobs* kj, int id, const gw::WArgs*& g, int& z, int& m0, int& n0) {
  const int j1 = id >= kj->first1 ? 1 : 0;
  g = &kj->job[j1];
  if (j1) id -= kj->first1;
  const int ntm = (g->M + HM - 1) / HM, ntn = g->N / HN;
  const int x = id & 7, slot = id >> 3, per = ntn * ((ntm + 7) / 8);
  z = slot / per;
  const int tt = slot % per, mt = (tt / ntn) * 8 + x;
  m0 = mt * HM; n0 = (tt % ntn) * HN;
  return mt < ntm;
}
__device__ __forceinline__ void h2_load_side(const gw::WArgs* g, int z, int m0, int n0, int wave, int tid, H2Load& l) {
  const float* Ab = g->a_ptrs[0] ? g->a_ptrs[z & 7] : g->A + (long)z * g->a_sz;
  l.rs_a = make_rsrc(Ab, (unsigned)(((long)(g->M - 1) * g->a_sm + g->K) * 4));
  l.rs_w = make_rsrc(g->Wf, g->wf_bytes);
  const int a_row = tid >> 3, a_k = (tid & 7) * 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) l.a_voff[i] = (m0 + a_row + 64 * i) < g->M ? ((m0 + a_row + 64 * i) * g->a_sm + a_k) * 4 : 0x40000000;
  l.b_src = (n0 / 32 + wave) * (g->K / 16) * kChunk3;
}
__device__ __forceinline__ void h2_store_side(const gw::WArgs* g, int z, int m0, int n0, int wave, int lane, H2Store& c) {
  const int wr = wave >> 2, wc = wave & 3, li = lane & 31, lh = lane >> 5;
  float* Cb = g->c_ptrs[0] ? g->c_ptrs[z & 7] : g->C + (long)z * g->c_sz;
  // (rows past M lie beyond the descriptor's records -- the row stride covers the columns, c_sm >= N: their stores are dropped)
  c.rs_c = make_rsrc(Cb, (unsigned)(((long)(g->M - 1) * g->c_sm + g->N) * 4));
  c.c_voff = ((m0 + wr * 64 + 4 * lh) * g->c_sm + n0 + wc * 64 + li) * 4;
  c.c_sm4 = g->c_sm * 4;
  c.bn0 = g->bias_n ? g->bias_n[n0 + wc * 64 + li] : 0.f;
  c.bn1 = g->bias_n ? g->bias_n[n0 + wc * 64 + 32 + li] : 0.f;
  c.oscale = g->oscale;
}

// H: the two pieces are FP16 (the forward's projections; range-tracked) -- else bf16 hi + mid (the backward's dQ = dP_q W_q:
// gradients keep fp32's range; the weight image is the three-piece bf16 one, whose first two pieces are read)
template <bool H>
__global__ __launch_bounds__(512, 2) void gemm_h2p_kernel(const H2Jobs jobs_by_value, const int total, float* const status) {
  extern __shared__ __attribute__((aligned(16))) short h2_smem[];
#if defined(__HIP_DEVICE_COMPILE__)
  const H2Jobs* const kj = (const H2Jobs*)__builtin_amdgcn_kernarg_segment_ptr();   // (the first kernel argument, by address)
#else
  const H2Jobs* const kj = &jobs_by_value;
#endif
  if (H) f16_saturating_conversions();
  short* const smem = h2_smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, li = lane & 31, lh = lane >> 5;
  const int G = gridDim.x;
  short* const abuf0 = smem;
  short* const abuf1 = smem + A_BUF;
  const int a_wr = (tid >> 3) * LDA + (tid & 7) * 4;
  const int a_rd = (wr * 64 + li) * LDA + 8 * lh;
  const __amdgpu_buffer_rsrc_t rs_null = make_rsrc(smem, 0u);          // no records: loads read 0, stores are dropped

  // L: what the loader requests (the tile being computed, or -- towards the end of its last pass -- the next tile): the A rows
  // two steps ahead (they cross into the next tile at step 14), the weight fragments two half steps ahead (at step 15);
  // ps: the store side of the finished tile whose result is going out during the current tile's steps
  H2Load L;
  H2Store ps;
  const gw::WArgs* cg;                                                 // the tile being computed
  int cz, cm0, cn0;
  int cid = blockIdx.x;
  while (cid < total && !h2_decode(kj, cid, cg, cz, cm0, cn0)) cid += G;
  if (cid >= total) return;
  h2_load_side(cg, cz, cm0, cn0, wave, tid, L);
  __amdgpu_buffer_rsrc_t b_rs = L.rs_w, nb_rs = L.rs_w;               // weight image the fragment loads read / of the next tile
  int b_src = L.b_src + (wc * 2 - wave) * (cg->K / 16) * kChunk3 + lane * 16, nb_src = b_src;   // this wave's first column tile, lane part included
  int b_tile = (cg->K / 16) * kChunk3, nb_tile = b_tile;               // bytes between column tiles of the image
  ps.rs_c = rs_null; ps.c_voff = 0; ps.c_sm4 = 0; ps.bn0 = ps.bn1 = 0.f; ps.oscale = 1.f;
  int lk = 0;                                                          // the A loader requests k-step lk + (static step) + 2
  int lkb = 0;                                                         // the B loader requests half step lkb + (static half) + 2

  f32x16 acc[2][2][2];                                                 // [set][i][j]
  // (set 1 is "stored" during the first tile -- into a descriptor without records -- and its values enter the range maximum)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[1][i][j][r] = 0.f;
  f32x4 raw[2];
  bf16x8 af[2][2][2];                                                  // [fragment set][piece][tile i]: from LDS, one half step ahead
  bf16x8 bq[3][2][2];                                                  // [ring][piece][tile j]: from the weight image (L2), two half steps ahead
  unsigned ph[2], pm[2];
  float ra[2], rb[2];
  float amax = 0.f;
  const float ascale = H ? 1.0f / kF16WScale : 1.0f;

  auto load_a = [&](int i, int ks) { raw[i] = buf_load4(L.rs_a, L.a_voff[i], ks * HK * 4); };
  // weight fragments of half step `hs` (16 k): (piece q, tile j) of this wave's two column tiles, 1 KB per wave each
  auto load_b = [&](auto Rc, int hs, int q, int j) {
    constexpr int R = decltype(Rc)::value;
    if (GEMMH2_KO & 2) return;
    bq[R][q][j] = __builtin_bit_cast(bf16x8, buf_load4(b_rs, b_src + j * b_tile + q * 1024, hs * kChunk3));
  };
  // split of pair e of raw[i] in three sub-stages of 2, 4 and 1 VALU operations (the range maximum rides in the first)
  auto stage = [&](int i, int e, int st) {
    if (GEMMH2_KO & 8) { if (st == 0) ph[e] = pm[e] = __builtin_bit_cast(unsigned, raw[i][2 * e]); return; }
    if (!H) {                                                          // bf16 hi + mid (fused.h split_pair<2>)
      if (st == 0) ph[e] = cvt_pk_bf16(raw[i][2 * e], raw[i][2 * e + 1]);
      else if (st == 1) {
        ra[e] = sub1(raw[i][2 * e], __builtin_bit_cast(float, ph[e] << 16));
        rb[e] = sub1(raw[i][2 * e + 1], __builtin_bit_cast(float, ph[e] & 0xffff0000u));
      } else pm[e] = cvt_pk_bf16(ra[e], rb[e]);
      return;
    }
    if (st == 0) {
      // (one v_max3_f32 with |.| modifiers: fmaxf() costs a canonicalising v_max per operand on top)
      asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(raw[i][2 * e]), "v"(raw[i][2 * e + 1]));
      ph[e] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{raw[i][2 * e], raw[i][2 * e + 1]}), hfv2));
    } else if (st == 1) {
      const hfv2 hh = __builtin_bit_cast(hfv2, ph[e]);
      ra[e] = sub1(raw[i][2 * e], (float)hh[0]);
      rb[e] = sub1(raw[i][2 * e + 1], (float)hh[1]);
    } else {
      pm[e] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{ra[e], rb[e]}), hfv2));
    }
  };
  auto write_a = [&](short* img, int i, int q) {
    if (GEMMH2_KO & 8) return;
    *reinterpret_cast<u32x2*>(&img[q * A_IMG + 64 * i * LDA + a_wr]) = q == 0 ? u32x2{ph[0], ph[1]} : u32x2{pm[0], pm[1]};
  };
  // fragment (piece q, tile i) of half step h into fragment set SET
  auto read_af = [&](auto SETc, const short* img, int h, int q, int i) {
    constexpr int SET = decltype(SETc)::value;
    if (GEMMH2_KO & 32) return;
    af[SET][q][i] = *reinterpret_cast<const bf16x8*>(&img[q * A_IMG + a_rd + i * 32 * LDA + 16 * h]);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  // ---- prologue (once): step 0 of the first tile split into image 0, step 1 requested, weight fragments of half steps 0, 1 --
  load_a(0, 0); load_a(1, 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) { load_b(I0{}, 0, k >> 1, k & 1); load_b(I1{}, 1, k >> 1, k & 1); }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 2; ++e) { stage(i, e, 0); stage(i, e, 1); stage(i, e, 2); }
    write_a(abuf0, i, 0); write_a(abuf0, i, 1);
  }
  load_a(0, 1); load_a(1, 1);
  lds_barrier();
#pragma unroll
  for (int k = 0; k < 4; ++k) read_af(I0{}, abuf0, 0, k >> 1, k & 1);

  short* a0 = abuf0; short* a1 = abuf1;                                // A images of this / the next step
  int nid = cid;                                                       // the tile after the one being computed (set at step 14)
  bool more = true;
  int stamp_n = 1;
  unsigned long long* const stamp_p = reinterpret_cast<unsigned long long*>(const_cast<char*>(static_cast<const char*>(cg->Wf)) + (size_t)blockIdx.x * kChunk3 + 2048);
  if (GEMMH2_STAMPS && tid == 0) { stamp_p[0] = __builtin_amdgcn_s_memtime(); stamp_p[127] = __builtin_amdgcn_s_memrealtime(); }

  // The twelve MFMAs of a half step, smallest partial product first: n = 0..3 lo(A) x hi(B), 4..7 hi x lo, 8..11 hi x hi, on
  // A fragment set FS (LDS, read one half step ahead in the order of first use) and weight ring set RS (registers, loaded two
  // half steps ahead: ring set (g + 2) % 3 = (g - 1) % 3 was last used in the previous half step).
  // 16 steps of the current tile into accumulator set S; PS = the set being stored (four of its 64 values per step).
  // last: this is the tile's last pass -- towards its end the loaders work on the next tile.
  auto pass16 = [&](auto Sc, const __amdgpu_buffer_rsrc_t prs, const bool last) __attribute__((always_inline)) {
    constexpr int S = decltype(Sc)::value, PS = S ^ 1;
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s == 14 && last) {                                           // A rows: steps 14, 15 request steps 0, 1 of the next tile
        const gw::WArgs* ng; int nz, nm0, nn0;
        nid = cid + G;
        while (nid < total && !h2_decode(kj, nid, ng, nz, nm0, nn0)) nid += G;
        more = nid < total;
        if (more) {
          h2_load_side(ng, nz, nm0, nn0, wave, tid, L);
          nb_rs = L.rs_w; nb_tile = (ng->K / 16) * kChunk3;
          nb_src = L.b_src + (wc * 2 - wave) * nb_tile + lane * 16;
        } else { L.rs_a = rs_null; nb_rs = rs_null; }
        lk = -16;
      }
      if (s == 15 && last) { b_rs = nb_rs; b_src = nb_src; b_tile = nb_tile; lkb = -32; }   // weight fragments: step 15 requests half steps 0, 1 of the next tile
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (GEMMH2_STAMPS == 1 && tid == 0 && stamp_n < 126) {
          stamp_p[stamp_n] = __builtin_amdgcn_s_memtime();
          stamp_p[126] = __builtin_amdgcn_s_memrealtime();
          ++stamp_n;
        }
#pragma unroll
        for (int n = 0; n < 12; ++n) {
          const int tp = n >> 2, i = (n >> 1) & 1, j = n & 1;
          const int g = 2 * s + h;                                     // half step of the pass: A set g & 1, weight ring set g % 3
          auto mf = [&](auto FSc, auto RSc) __attribute__((always_inline)) {
            constexpr int FS = decltype(FSc)::value, RS = decltype(RSc)::value;
            if (!(GEMMH2_KO & 4)) acc[S][i][j] = mfma32_16<H>(af[FS][PA[tp]][i], bq[RS][PB[tp]][j], acc[S][i][j]);
            else if (n == 0) acc[S][i][j][0] += __builtin_bit_cast(float, (int)af[FS][0][i][0] ^ (int)bq[RS][0][j][0]);
          };
          auto ldb = [&](int q, int jj) __attribute__((always_inline)) {   // into ring set (g + 2) % 3
            if ((g + 2) % 3 == 0) load_b(I0{}, lkb + g + 2, q, jj);
            else if ((g + 2) % 3 == 1) load_b(I1{}, lkb + g + 2, q, jj);
            else load_b(I2{}, lkb + g + 2, q, jj);
          };
          if (GEMMH2_STAMPS == 2 && (s == 4 || s == 5) && tid == 0 && stamp_n < 126) { stamp_p[stamp_n] = __builtin_amdgcn_s_memtime(); ++stamp_n; }
          if (h == 0) {
            if (g % 3 == 0) mf(I0{}, I0{}); else if (g % 3 == 1) mf(I0{}, I1{}); else mf(I0{}, I2{});
          } else {
            if (g % 3 == 0) mf(I1{}, I0{}); else if (g % 3 == 1) mf(I1{}, I1{}); else mf(I1{}, I2{});
          }
          // weight fragments of half step g + 2: hi pieces first (first use: n = 0)
          if (n == 0) ldb(0, 0);
          if (n == 1) ldb(0, 1);
          if (n == 2) ldb(1, 0);
          if (n == 3) ldb(1, 1);
          if (h == 0) {
            // A fragments of half 1 (this step's image), lo pieces first.  The split of the rows requested a step ago, spread
            // over the twelve slots (pair p = n / 3 of the four, sub-stage n % 3: 2, 4, 1 VALU operations); each row register
            // set is re-requested (step s + 2) as soon as its last pair has been read; pieces of raw[0] written at n = 6, 7.
            if (n == 0) read_af(I1{}, a0, 1, 1, 0);
            if (n == 1) read_af(I1{}, a0, 1, 1, 1);
            if (n == 2) read_af(I1{}, a0, 1, 0, 0);
            if (n == 3) read_af(I1{}, a0, 1, 0, 1);
            if (n == 6) write_a(a1, 0, 0);
            if (n == 7) write_a(a1, 0, 1);
            stage(n / 6, (n / 3) & 1, n % 3);
            if (n == 5 && !(GEMMH2_KO & 1)) load_a(0, lk + s + 2);
            if (n == 11 && !(GEMMH2_KO & 1)) load_a(1, lk + s + 2);
          } else {
            // pieces of raw[1]; then the barrier -- this wave's A pieces of the next step have landed (lgkmcnt), all waves meet,
            // the reads come after it --; the A fragments of the next step's first half (lo pieces first); the deferred stores
            if (n == 0) write_a(a1, 1, 0);
            if (n == 1) write_a(a1, 1, 1);
            if (n == 2) {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              if (!(GEMMH2_KO & 64)) __builtin_amdgcn_s_barrier();
              asm volatile("" ::: "memory");
              read_af(I0{}, a1, 0, 1, 0);
            }
            if (n == 3) read_af(I0{}, a1, 0, 1, 1);
            if (n == 4) read_af(I0{}, a1, 0, 0, 0);
            if (n == 5) read_af(I0{}, a1, 0, 0, 1);
            if (n >= 6 && n < 10 && !(GEMMH2_KO & 16)) {               // element e = 4 s + (n - 6): (i, r, j) = (e >> 5, (e >> 1) & 15, e & 1)
              const int e = 4 * s + (n - 6), pi = e >> 5, pr = (e >> 1) & 15, pj = e & 1;
              const float y = fmaf(acc[PS][pi][pj][pr], ascale, pj ? ps.bn1 : ps.bn0) * ps.oscale;
              if (H) asm volatile("v_max_f32 %0, %0, |%1|" : "+v"(amax) : "v"(y));   // (here and now: left alone hipcc defers the 64 maxima to the end and spills every y until then)
              const int row = pi * 32 + (pr & 3) + 8 * (pr >> 2);
              int sm4 = ps.c_sm4;
              asm volatile("" : "+s"(sm4));                            // (computed where it is used: 32 hoisted products would not fit the SGPRs)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), prs, ps.c_voff + pj * 128, row * sm4, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      { short* ta = a0; a0 = a1; a1 = ta; }
    }
    // 32 half steps = 2 (mod 3): rotate the weight ring's two live sets (half steps 32, 33 -> ring sets 2, 0) back to sets 0, 1
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 2; ++j) { const bf16x8 t0 = bq[0][q][j]; bq[0][q][j] = bq[2][q][j]; bq[1][q][j] = t0; }
  };
  // one tile into set S; returns whether another tile follows
  auto run_tile = [&](auto Sc) __attribute__((always_inline)) {
    constexpr int S = decltype(Sc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[S][i][j][r] = 0.f;
    lk = 0; lkb = 0;
    pass16(Sc, ps.rs_c, true);                                          // (K == 512: one pass of 16 steps per tile)
    // this tile is the finished one now: its store side; the next tile (found at step 14) becomes the one being computed
    h2_store_side(cg, cz, cm0, cn0, wave, lane, ps);
    cid = nid;
    if (more) (void)h2_decode(kj, cid, cg, cz, cm0, cn0);
    return more;
  };
  auto flush = [&](auto PSc) __attribute__((always_inline)) {          // the last tile's result
    constexpr int PS = decltype(PSc)::value;
    if (GEMMH2_KO & 16) { if (acc[PS][0][0][0] != 12345.678f) return; }
#pragma unroll
    for (int e = 0; e < 64; ++e) {
      const int pi = e >> 5, pr = (e >> 1) & 15, pj = e & 1;
      const float y = fmaf(acc[PS][pi][pj][pr], ascale, pj ? ps.bn1 : ps.bn0) * ps.oscale;
      if (H) asm volatile("v_max_f32 %0, %0, |%1|" : "+v"(amax) : "v"(y));
      const int row = pi * 32 + (pr & 3) + 8 * (pr >> 2);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), ps.rs_c, ps.c_voff + pj * 128, row * ps.c_sm4, 0);
    }
  };
  for (;;) {
    if (!run_tile(I0{})) { flush(I0{}); break; }
    if (!run_tile(I1{})) { flush(I1{}); break; }
  }
  if (H && status && __builtin_amdgcn_ballot_w64(!(amax <= kF16Exact)) != 0) {
    amax = wave_max(amax);
    if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(status), __builtin_bit_cast(unsigned, amax));
  }
}

}  // namespace

// COATTN_GEMM_H2=0 (developer switch): gemm_w's two-FP16-piece mode instead
static int gemm_h2_enabled() {
  static const int on = dev_env_int("COATTN_GEMM_H2", 1);
  return on;
}

int gemm_h2_supported(const WGemm& d) {
  auto pal = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  // two FP16 pieces (the forward): any K % 32 == 0; two bf16 pieces (the backward's dQ projection, behind the developer switch
  // COATTN_OWN_DQ of coattn_fused_bwd.hip -- by default those tiles ride in the weight-gradient launch on gemm_w_body): the
  // persistent kernel only
  static const int bf_pieces = dev_env_int("COATTN_OWN_DQ", 0);
  bool ok = gemm_h2_enabled() && d.np == 2 && (d.f16 || (bf_pieces && d.K == 512 && d.c_sm >= d.N && d.M >= 128)) && !d.bf16 && !d.a_bf16 && !d.a_sk && d.kband_n == 0 && d.M >= 1 && d.N >= HN &&
            (d.N % HN) == 0 && d.K >= HK && (d.K % HK) == 0 && (d.a_sm & 3) == 0 && (d.a_sz & 3) == 0 && d.batch >= 1 && d.batch <= 8 &&
            (d.a_ptrs[0] ? true : pal(d.A)) && ((long)d.M * d.a_sm + d.K) * 4 < 0x40000000L && wsplit_bytes(d.N, d.K) < 0x40000000UL;
  for (int t = 0; t < 8; ++t) ok = ok && pal(d.a_ptrs[t]);
  return ok ? 1 : 0;
}

int launch_gemm_h2(const WGemm* d, int n, hipStream_t s) {
  CA_CHECK_ARG(n == 1 || n == 2, "gemm_h2: 1 or 2 jobs per launch");
  H2Jobs jobs = {};
  long nb[2] = {0, 0};
  for (int i = 0; i < n; ++i) {
    CA_CHECK_ARG(gemm_h2_supported(d[i]), "gemm_h2: unsupported shape M=%d N=%d K=%d", d[i].M, d[i].N, d[i].K);
    CA_CHECK_ARG((d[i].A || d[i].a_ptrs[0]) && d[i].Wf && (d[i].C || d[i].c_ptrs[0]), "gemm_h2: null operand");
    gw::WArgs& g = jobs.job[i];
    g = gw::WArgs{};
    g.A = d[i].A; g.a_sz = d[i].a_sz; g.a_sm = d[i].a_sm;
    g.Wf = d[i].Wf; g.wf_bytes = (unsigned)wsplit_bytes(d[i].N, d[i].K);
    g.C = d[i].C; g.c_sz = d[i].c_sz; g.c_sm = d[i].c_sm;
    for (int t = 0; t < 8; ++t) { g.a_ptrs[t] = d[i].a_ptrs[t]; g.c_ptrs[t] = d[i].c_ptrs[t]; }
    g.bias_n = d[i].bias_n; g.oscale = d[i].out_scale != 0.f ? d[i].out_scale : 1.f;
    g.ascale = d[i].f16 ? 1.0f / kF16WScale : 1.0f;
    g.status = d[i].f16 ? d[i].status : nullptr;
    g.M = d[i].M; g.N = d[i].N; g.K = d[i].K;
    const long ntm = (d[i].M + HM - 1) / HM, ntn = d[i].N / HN;
    nb[i] = (long)d[i].batch * ntn * ((ntm + 7) / 8) * 8;
  }
  CA_CHECK_ARG(nb[0] + nb[1] < 2147483647L, "gemm_h2: grid too large");
  jobs.first1 = (int)nb[0];
  static DeviceOnce once;
  static int n_cu[DeviceOnce::kMaxDev];
  CA_TRY(once.run([&] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2p_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2p_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    int dev = 0, cus = 256;
    if (e == hipSuccess) e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess && dev >= 0 && dev < DeviceOnce::kMaxDev) n_cu[dev] = cus;
    return e;
  }, "gemm_h2"));
  // persistent form (one workgroup per CU walks its tiles as one pipeline): 512-deep passes, both jobs; COATTN_GEMM_H2P=0
  // (developer switch): one workgroup per tile
  static const int persistent = dev_env_int("COATTN_GEMM_H2P", 1);
  bool pk = persistent != 0 || !d[0].f16;
  for (int i = 0; i < n; ++i) pk = pk && d[i].K == 512 && d[i].c_sm >= d[i].N;
  CA_CHECK_ARG(n == 1 || (d[0].f16 != 0) == (d[1].f16 != 0), "gemm_h2: the jobs of a launch share the piece format");
  // (the status word of a launch is shared by its jobs)
  float* status = d[0].status ? d[0].status : (n == 2 ? d[1].status : nullptr);
  if (pk) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int cus = (dev >= 0 && dev < DeviceOnce::kMaxDev && n_cu[dev] > 0) ? n_cu[dev] : 256;
    const long total = nb[0] + nb[1];
    const unsigned grid = (unsigned)(total < cus ? total : cus);
    if (d[0].f16) hipLaunchKernelGGL(gemm_h2p_kernel<true>, dim3(grid), dim3(512), kLdsP, s, jobs, (int)total, status);
    else hipLaunchKernelGGL(gemm_h2p_kernel<false>, dim3(grid), dim3(512), kLdsP, s, jobs, (int)total, status);
    CA_CHECK_LAUNCH("gemm_h2p");
    return 0;
  }
  CA_CHECK_ARG(d[0].f16, "gemm_h2: the bf16-piece form exists as the persistent kernel only");
  hipLaunchKernelGGL(gemm_h2_kernel, dim3((unsigned)(nb[0] + nb[1])), dim3(512), kLds, s, jobs);
  CA_CHECK_LAUNCH("gemm_h2");
  return 0;
}
