// Weight-gradient GEMM  C = A^T B  on 128 x 256 tiles and 512-thread workgroups (gfx950) -- the backward's one GEMM
// launch at the two-piece width.
//
// gemm_tn.hip's 128 x 128 tile was laid out for six partial products per fp32 product: its fillers hide under 24 MFMAs
// per step.  At the two-piece width (three products) the knock-outs of that kernel (tools/ab_gemmtn.sh) show
// the MFMAs hidden completely and the launch carried by what every tile repeats: the split of its operand blocks
// (each block is split by all four tiles that share it; under SUM3 the A operand costs 10 VALU per pair) and their loads
// (32 KB per step and workgroup under SUM3: the CU's vector-memory path at the MFMA-bound step rate).  Here a workgroup
// owns a 128 x 256 tile -- eight waves as 2 x 4, each still 64 x 64 (2 x 2 MFMA tiles, the same 12 MFMAs per step): the A
// rows (the SUM3 operand) are loaded, summed, split and staged ONCE for twice the columns, a thread stages one A float4 and
// two B float4 per step instead of two and two.  Per MFMA: 5 loads instead of 8, 44 split VALU instead of 64, 6 LDS writes
// instead of 8.  One workgroup per CU (8 waves = the two waves per SIMD of the four-wave kernel's two workgroups), 57 KB
// of LDS at two pieces; 32 split-K parts x 8 tiles = the 256 CUs.  Same numerics as gemm_tn.hip (same pieces, same order
// of the partial products, same split-K partition when the part count is the same); the dQ projection's tiles ride along on
// gemm_w_body's 128 x 256 / 512-thread form, the small reductions as workgroups whose first 256 threads work.
//
// What bounds it (round 4, measured on the weight gradients alone, COATTN_NO_COMBINE=1; 88 us at N = 196, 39 at N = 49):
// not the schedule.  Knock-outs (GEMMTNW_KO below): without MFMAs 88 us, without the requests 61.5, without split / image
// writes / fragment reads 56.5, MFMAs alone 50.  Two forms of this body that take the staging chain off the MFMA waves'
// critical path measured THE SAME time: requests two steps ahead (a second staging register set: 121 vs 122 us for the
// whole launch), and specialised waves -- four waves (one per SIMD) that only request / sum / split / write, four that
// only read fragments and multiply 64 x 128 each, one barrier per step, same bits -- 120.4 vs 121.8 us at N = 196, 63.2 vs
// 62.0 at N = 49.  The launch moves its 338 MB at 3.8 TB/s whatever the waves do in between: the rate at which a CU's
// 40 KB per step (512-byte and 1-KB row segments of 2-KB rows) come back, not their latency and not the issue slots.
// Neither form was kept.
//
// Shapes: M % 128 == 0, N % 256 == 0; B row-major over the contraction rows, or (first job, BCM) contiguous along them in
// groups -- channel-major image features; the masked phrase-level products stay on gemm_tn.hip.
#include "common.h"
#include "fused.h"
#include "gemm_w_body.h"
#include <stdlib.h>
#include <type_traits>

// Developer switches (tools/ab_gemmtn.sh with FILE=gemm_tn_wide MACRO=GEMMTNW_KO, never in the shipped library; wrong
// results): GEMMTNW_KO bit 0 no MFMAs, 1 no global loads after the prologue's, 2 no split arithmetic, 3 no LDS writes,
// 4 no fragment reads, 5 no barrier, 6 no stores of the split-K partial results.
#ifndef GEMMTNW_KO
#define GEMMTNW_KO 0
#endif
#ifndef TNW_INTERLEAVE
#define TNW_INTERLEAVE 0
#endif

namespace {

constexpr int BM = 128, BN = 256, BK = 16, NTHR = 512;
constexpr int LDTA = BM + 32, LDTB = BN + 32;      // [k][col] image row strides (elements): conflict-free writes + transposed reads
constexpr int IMGA = BK * LDTA, IMGB = BK * LDTB;  // one piece image of each operand
constexpr int LDRB = 24, IMGBC = BN * LDRB;        // BCM: the B image is [column][k] (16 k + 8 pad), one ds_read_b128 per fragment

struct TwArgs {
  const float* A; long a_sl; int a_ld; long a_term;
  const float* B; const float* b_ptrs[8]; long b_sl; int b_ld;
  int b_kdiv; long b_sdiv; unsigned b_bytes;     // BCM (gemm_tn.hip): k = (group, row inside the group of b_kdiv rows)
  float* C;
  int M, N, K, ksplit, S;
};
struct TwJobs {
  TwArgs job[2]; int first1; ReduceJobs red; int nred, red_bx, red_nparts, red_acc; long red_n;
  gw::WArgs wj; int nw;
  int wfirst;                     // > 0: the dQ tiles come FIRST, in this many block slots (nw rounded up to the XCD count), the split-K parts behind them
  TnDyn dyn;                      // bits != NULL: the parts of both jobs are planned on the device (fused.h)
};

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// BCM: the B operand is contiguous along k inside groups of b_kdiv rows (channel-major image features [B, d, N]: k =
// (sample, location), n = channel), as in gemm_tn.hip: float4 = 4 consecutive k of one column, [column][k] images.
// GATHER (job 1 of a launch planned on the device, TnDyn): the contraction runs over the rows of level `lvl` whose bit is set in
// `bits`; [kbeg, kend) then counts positions of that compacted list, and the rows come from a map in LDS (behind the images)
// that the workgroup builds first: exclusive prefix sums of the words' popcounts (one wave, 8 words per lane), then per position
// a binary search for its word and a walk over the word's bits (as gemm_w_body.h's compacted tiles).  Positions past the end map
// to row K: outside the operands' buffers, so the loads return zeros.
template <bool SUM3, int NP, bool BCM, bool GATHER>
__device__ __forceinline__ void gemm_tn_wide_core(const TwArgs& g, const int m0, const int n0, const int z, const int lvl, const int p,
                                                  const int kbeg, const int kend, short* const lds, const unsigned* bits, const int words) {
  static_assert(NP == 2 || NP == 3, "pieces per operand");
  static_assert(!GATHER || (!SUM3 && !BCM), "the gathered job is the plain row-major one");
  constexpr int OPERA = NP * IMGA, OPERB = NP * (BCM ? IMGBC : IMGB), BUF = OPERA + OPERB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3, li = lane & 31, lh = lane >> 5;
  // TNW_INTERLEAVE (developer build, location-major B only): part p takes the 16-row blocks p, p + S, p + 2 S, ... instead of a
  // contiguous range -- all workgroups then read one neighbourhood of rows at a time
  const bool il = TNW_INTERLEAVE && !BCM && !GATHER;
  const int nblk16 = (g.K + BK - 1) / BK;
  const int steps = il ? (nblk16 > p ? (nblk16 - p + g.S - 1) / g.S : 0) : (kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0);
  int* const rowmap = reinterpret_cast<int*>(lds + 2 * BUF) + 520;       // [(steps + 3) * 16]; the 513 prefix sums in front of it
  if (GATHER) {
    int* pre = reinterpret_cast<int*>(lds + 2 * BUF);
    const unsigned* lb = bits + (long)lvl * words;
    if (wave == 0) {
      int pcw[8], sum = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = 8 * lane + e;
        pcw[e] = i < words ? __builtin_popcount(lb[i]) : 0;
        sum += pcw[e];
      }
      int incl = sum;
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
    const int total = pre[512], nmap = (steps + 3) * BK;
    for (int e = tid; e < nmap; e += NTHR) {
      const int c = kbeg + e;
      int row = g.K;
      if (c < kend && c < total) {
        int lo = 0, hi = 511;
#pragma unroll
        for (int it = 0; it < 9; ++it) {
          const int mid = (lo + hi + 1) >> 1;
          if (pre[mid] <= c) lo = mid; else hi = mid - 1;
        }
        unsigned b = lb[lo];
        for (int k = c - pre[lo]; k > 0; --k) b &= b - 1;
        row = 32 * lo + __builtin_ctz(b);
      }
      rowmap[e] = row;
    }
    __syncthreads();
  }
  const float* Ab = g.A + (long)lvl * g.a_sl;
  const float* Bb = g.b_ptrs[0] ? g.b_ptrs[lvl & 7] : g.B + (long)lvl * g.b_sl;
  // rows past K read 0 (resource bound); rows past kend belong to the next part: ksplit % 16 == 0, so a step never straddles
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(Ab, (unsigned)((long)g.K * g.a_ld * 4));
  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(Bb, BCM ? g.b_bytes : (unsigned)((long)g.K * g.b_ld * 4));
  const unsigned tbytes = (SUM3 && g.a_term) ? (unsigned)((long)g.K * g.a_ld * 4) : 0u;
  const __amdgpu_buffer_rsrc_t rs_a1 = make_rsrc(Ab + (SUM3 ? g.a_term : 0), tbytes);
  const __amdgpu_buffer_rsrc_t rs_a2 = make_rsrc(Ab + (SUM3 ? 2 * g.a_term : 0), tbytes);

  // staging per step: A one float4 per thread (16 k-rows x 128 columns), B two (8 k-rows x 256 columns, twice)
  const int ska = tid >> 5, sma = (tid & 31) * 4, skb = tid >> 6, smb = (tid & 63) * 4;
  const int a_voff = ((kbeg + ska) * g.a_ld + m0 + sma) * 4, b_voff = ((kbeg + skb) * g.b_ld + n0 + smb) * 4;
  const int a_step = (il ? g.S : 1) * BK * g.a_ld * 4, b_step = (il ? g.S : 1) * BK * g.b_ld * 4, b_half = 8 * g.b_ld * 4;
  const int sta = ska * LDTA + sma, stb = OPERA + skb * LDTB + smb;          // (+ 8 * LDTB for the second B float4)
  // transposed fragment read: each 16-lane group fetches a 4 (k) x 16 (rows) block; lane 4q+p of the group supplies
  // the address of block row q, columns 4p..4p+3, and receives the 4 k of row (lane & 15)
  const int trq = 8 * lh + ((lane & 15) >> 2), trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int a_rd = trq * LDTA + trc + wr * 64;
  const int b_rd = BCM ? OPERA + (wc * 64 + li) * LDRB + 8 * lh : OPERA + trq * LDTB + trc + wc * 64;
  // BCM staging: 4 lanes cover the 16 k (64 B) of one column, 128 columns per load of the workgroup; the second float4
  // is column + 128.  The (group, row) position of the thread's 4-k group is tracked step by step (no division in the loop).
  const int bc_col = tid >> 2, bc_kq = tid & 3;
  int bc_n = 0, bc_voff = 0;
  if (BCM) {
    const int kg = kbeg + 4 * bc_kq;
    bc_n = kg % g.b_kdiv;
    bc_voff = (int)(((long)(kg / g.b_kdiv) * g.b_sdiv + bc_n + (long)(n0 + bc_col) * g.b_ld) * 4);
  }
  const int bc_st = OPERA + bc_col * LDRB + 4 * bc_kq;                     // + 128 * LDRB for the second float4

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 raw[3];                                  // A, B k 0..7, B k 8..15 of the step being staged
  f32x4 rawt[2];                                 // SUM3: the other two terms of raw[0]
  bf16x4 fa[2][3][2][2], fb[2][3][2][2];         // [set][piece][tile][k half]: fragment = {lo, hi}
  unsigned ph[2], pm[2], pl[2];
  float ra[2], rb[2];
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first (gemm.hip's order)
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
  auto load_raw = [&](int x, int s) {            // (the step goes into the VECTOR offset: the range check does not see the scalar one)
    if ((GEMMTNW_KO & 2) && s >= 2) return;
    if (GATHER) {                                // rows from the map (x = 0: A row ska; x = 1, 2: B rows skb, skb + 8 of the step)
      if (x == 0) raw[0] = buf_load4(rs_a, (rowmap[s * BK + ska] * g.a_ld + m0 + sma) * 4, 0);
      else raw[x] = buf_load4(rs_b, (rowmap[s * BK + skb + 8 * (x - 1)] * g.b_ld + n0 + smb) * 4, 0);
      return;
    }
    if (x == 0) {
      raw[0] = buf_load4(rs_a, a_voff + s * a_step, 0);
      if (SUM3) {
        rawt[0] = buf_load4(rs_a1, a_voff + s * a_step, 0);
        rawt[1] = buf_load4(rs_a2, a_voff + s * a_step, 0);
      }
    } else if (BCM) {
      raw[x] = buf_load4(rs_b, bc_voff + (x - 1) * 128 * g.b_ld * 4, 0);
      if (x == 2) {                              // both column halves of this step requested: on to the next 16 k
        bc_n += BK; bc_voff += BK * 4;
        if (bc_n >= g.b_kdiv) { bc_n -= g.b_kdiv; bc_voff += (int)((g.b_sdiv - g.b_kdiv) * 4); }
      }
    } else raw[x] = buf_load4(rs_b, b_voff + (x - 1) * b_half + s * b_step, 0);
  };
  auto stage = [&](int x, int e, int st) {       // split of raw[x], pair e, in three stages of 5, 5 and 1 VALU
    if (GEMMTNW_KO & 4) {
      if (st == 0) ph[e] = pm[e] = pl[e] = __builtin_bit_cast(unsigned, raw[x][2 * e]);
      return;
    }
    if (st == 0) {
      if (SUM3 && x == 0) {                      // (level order 0 + 1 + 2, as the separate summing pass adds them)
        raw[0][2 * e] = (raw[0][2 * e] + rawt[0][2 * e]) + rawt[1][2 * e];
        raw[0][2 * e + 1] = (raw[0][2 * e + 1] + rawt[0][2 * e + 1]) + rawt[1][2 * e + 1];
      }
      ph[e] = cvt_pk_bf16(raw[x][2 * e], raw[x][2 * e + 1]);
      ra[e] = sub1(raw[x][2 * e], __builtin_bit_cast(float, ph[e] << 16));
      rb[e] = sub1(raw[x][2 * e + 1], __builtin_bit_cast(float, ph[e] & 0xffff0000u));
    } else if (st == 1) {
      pm[e] = cvt_pk_bf16(ra[e], rb[e]);
      if (NP == 2) return;
      ra[e] = sub1(ra[e], __builtin_bit_cast(float, pm[e] << 16));
      rb[e] = sub1(rb[e], __builtin_bit_cast(float, pm[e] & 0xffff0000u));
    } else if (NP == 3) {
      pl[e] = cvt_pk_bf16(ra[e], rb[e]);
    }
  };
  auto write_piece = [&](short* buf, int x, int q) {
    if (q >= NP) return;
    if ((GEMMTNW_KO & 8) && buf != lds) return;      // (the prologue's image is still written)
    const u32x2 v = q == 0 ? u32x2{ph[0], ph[1]} : (q == 1 ? u32x2{pm[0], pm[1]} : u32x2{pl[0], pl[1]});
    if (x == 0) *reinterpret_cast<u32x2*>(&buf[q * IMGA + sta]) = v;
    else if (BCM) *reinterpret_cast<u32x2*>(&buf[q * IMGBC + (x - 1) * 128 * LDRB + bc_st]) = v;
    else *reinterpret_cast<u32x2*>(&buf[q * IMGB + (x - 1) * 8 * LDTB + stb]) = v;
  };
  // fragment reads in the order of first use: a2, b0, a0, b2, a1, b1 (tile 0, tile 1; lo, hi): r = 0..23
  auto read_frag = [&](auto SETc, const short* buf, int r) {
    constexpr int SET = decltype(SETc)::value;
    constexpr int QA[3] = {2, 0, 1}, QB[3] = {0, 2, 1};
    const int grp = r >> 2, isb = grp & 1, q = isb ? QB[grp >> 1] : QA[grp >> 1], tile = (r >> 1) & 1, hi = r & 1;
    if (q >= NP) return;
    if ((GEMMTNW_KO & 16) && buf != lds) return;
    if (BCM && isb) {                            // one 16-byte read per fragment (issued with its first half)
      if (hi == 0) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(buf + q * IMGBC + b_rd + tile * 32 * LDRB);
        fb[SET][q][tile][0] = bf16x4{v[0], v[1], v[2], v[3]};
        fb[SET][q][tile][1] = bf16x4{v[4], v[5], v[6], v[7]};
      }
      return;
    }
    if (isb) fb[SET][q][tile][hi] = lds_tr16(buf + q * IMGB + b_rd + tile * 32 + hi * 4 * LDTB);
    else fa[SET][q][tile][hi] = lds_tr16(buf + q * IMGA + a_rd + tile * 32 + hi * 4 * LDTA);
  };
  auto frag = [&](const bf16x4 (&f)[2]) { return bf16x8{f[0][0], f[0][1], f[0][2], f[0][3], f[1][0], f[1][1], f[1][2], f[1][3]}; };
  // one 16-k step: MFMAs on fragment set SET; raw (step s + 1) is split into image `nxt`, re-requested for step
  // s + 2, and after the barrier the fragments of step s + 1 are read into the other set
  auto step = [&](auto SETc, int s, short* nxt) {
    constexpr int SET = decltype(SETc)::value;
    using OTHER = std::integral_constant<int, SET ^ 1>;
#pragma unroll
    for (int n = 0; n < 24; ++n) {
      // the MFMA of slot n: all 24 (NP = 3), every other slot (NP = 2: products 3 .. 5)
      const int mi = NP == 2 ? n >> 1 : n;
      const bool mf = NP == 3 || (n & 1);
      const int tt = NP == 2 ? 3 + (mi >> 2) : n >> 2, i = (mi >> 1) & 1, j = mi & 1;
      if (mf && (GEMMTNW_KO & 1)) acc[i][j][n & 15] += __builtin_bit_cast(float, (int)fa[SET][PA[tt]][i][0][0] ^ (int)fb[SET][PB[tt]][j][0][0]);
      if (mf && !(GEMMTNW_KO & 1))
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(fa[SET][PA[tt]][i]), frag(fb[SET][PB[tt]][j]), acc[i][j], 0, 0, 0);
      // raw[x]: pair 0 stages in slots 4x, 4x+1, 4x+2; pair 1 in 4x+1, 4x+2, 4x+3; pieces written in 4x+3 .. 4x+5
      if (n < 12) {
        const int x = n >> 2, u = n & 3;
        if (u <= 2) stage(x, 0, u);
        if (u >= 1) stage(x, 1, u - 1);
        if (u == 3) load_raw(x, s + 2);
      }
      if (n >= 3 && n < 14) {
        const int w = n - 3, x = w >> 2, q = w & 3;
        if (q < 3) write_piece(nxt, x, q);
      }
      if (n == 18 && !(GEMMTNW_KO & 32)) lds_barrier();
      if (n >= 18) {
#pragma unroll
        for (int r = 4 * (n - 18); r < 4 * (n - 17); ++r) read_frag(OTHER{}, nxt, r);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  short* const img0 = lds;
  short* const img1 = lds + BUF;
  // prologue: step 0 into image 0, raw = step 1, fragments of step 0 in set 0
#pragma unroll
  for (int x = 0; x < 3; ++x) load_raw(x, 0);
#pragma unroll
  for (int x = 0; x < 3; ++x) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int st = 0; st < 3; ++st) stage(x, e, st);
#pragma unroll
    for (int q = 0; q < 3; ++q) write_piece(img0, x, q);
    load_raw(x, 1);
  }
  lds_barrier();
#pragma unroll
  for (int r = 0; r < 24; ++r) read_frag(I0{}, img0, r);
  // (loads past the part's last step read rows of the next part or 0; they are split into the idle image and never used)
  int s = 0;
  for (; s + 2 <= steps; s += 2) {               // (one loop exit: the accumulators stay in place)
    step(I0{}, s, img1);
    step(I1{}, s + 1, img0);
  }
  if (s < steps) step(I0{}, s, img1);

  float* Cb = g.C + (long)z * g.M * g.N;
  if ((GEMMTNW_KO & 64) && acc[0][0][0] != 12345.678f) return;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float* crow = Cb + (long)row * g.N + n0 + wc * 64 + li;
#pragma unroll
      for (int j = 0; j < 2; ++j) crow[j * 32] = acc[i][j][r];
    }
}

// the host's static plan: workgroup bid of a job's nblk -> (part, tile)
template <bool SUM3, int NP, bool BCM = false>
__device__ __forceinline__ void gemm_tn_wide_body(const TwArgs& g, const int bid, const int nblk, short* const lds) {
  // XCD-run order (gemm_tn.hip): the tiles of a part share an XCD's L2
  const int ntn = g.N / BN, ntiles = (g.M / BM) * ntn;
  int lin = bid;
  if ((nblk & 7) == 0) lin = (bid & 7) * (nblk >> 3) + (bid >> 3);
  const int z = lin / ntiles, t = lin % ntiles;
  const int m0 = (t / ntn) * BM, n0 = (t % ntn) * BN;
  const int lvl = z / g.S, p = z % g.S;
  const bool il = TNW_INTERLEAVE && !BCM;
  const int kbeg = il ? p * BK : p * g.ksplit, kend = min(g.K, kbeg + g.ksplit);
  gemm_tn_wide_core<SUM3, NP, BCM, false>(g, m0, n0, z, lvl, p, kbeg, kend, lds, nullptr, 0);
}
// the plan made on the device (TnDyn): workgroup bid of the launch's P x tiles -> job 0 part z < S0, or job 1 (level, part)
template <bool SUM3, int NP, bool BCM>
__device__ __forceinline__ void gemm_tn_wide_dyn(const TwArgs& g0, const TwArgs& g1, const TnDyn& dyn, const int bid, const int nblk,
                                                 short* const lds) {
  const TnDynPlan pl = *dyn.plan;                    // (uniform: scalar loads)
  const int ntn = g0.N / BN, ntiles = (g0.M / BM) * ntn;
  int lin = bid;
  if ((nblk & 7) == 0) lin = (bid & 7) * (nblk >> 3) + (bid >> 3);
  const int z = lin / ntiles, t = lin % ntiles;
  const int m0 = (t / ntn) * BM, n0 = (t % ntn) * BN;
  if (z < pl.S0) {
    const int kbeg = z * pl.ks0;
    gemm_tn_wide_core<SUM3, NP, BCM, false>(g0, m0, n0, z, 0, z, kbeg, min(g0.K, kbeg + pl.ks0), lds, nullptr, 0);
    return;
  }
  const int zz = z - pl.S0, lvl = zz / pl.S1, p = zz - lvl * pl.S1;
  if (lvl >= dyn.levels) return;                                      // (P - S0 is not a multiple of the levels: parts left over)
  int live = 0, ks1 = BK;
#pragma unroll
  for (int l = 0; l < kDynLevels; ++l)
    if (l == lvl) { live = pl.live[l]; ks1 = pl.ks1[l]; }
  const int kbeg = p * ks1;
  gemm_tn_wide_core<false, NP, false, true>(g1, m0, n0, z, lvl, p, kbeg, min(live, kbeg + ks1), lds, dyn.bits, dyn.words);
}

// [small reductions][job 0 parts][job 1 parts][tiles of the dQ projection on gemm_w_body<.., 8>]
template <bool SUM3, int NP, bool BCM = false>
// (__launch_bounds__' second argument is the minimum number of waves per SIMD: 2 = ONE 512-thread workgroup per CU, 256
//  registers per lane -- the design; 169 used, no scratch: tools/regs.py gemm_tn_wide)
__global__ __launch_bounds__(NTHR, 2) void gemm_tn_wide_kernel(const TwJobs jobs_by_value) {
  extern __shared__ __attribute__((aligned(16))) short lds_dyn[];
  // (read through the kernel-argument segment: a run-time index into the by-value copy -- b_ptrs[level] of the device-planned
  //  launch -- made hipcc spill the whole 832-byte argument to scratch at kernel entry)
#if defined(__HIP_DEVICE_COMPILE__)
  const TwJobs& jobs = *(const TwJobs*)__builtin_amdgcn_kernarg_segment_ptr();
#else
  const TwJobs& jobs = jobs_by_value;
#endif
  int id = (int)blockIdx.x - jobs.nred;
  const int ngemm = (int)gridDim.x - jobs.nred - (jobs.wfirst ? jobs.wfirst : jobs.nw);
  if (jobs.wfirst && id >= 0) {
    if (id < jobs.wfirst) {
      if (id < jobs.nw) gw::gemm_w_body<false, NP, 8>(jobs.wj, id, lds_dyn);
      return;
    }
    id -= jobs.wfirst;
  } else if (id >= ngemm) {
    gw::gemm_w_body<false, NP, 8>(jobs.wj, id - ngemm, lds_dyn);
    return;
  }
  if (id < 0) {
    reduce_jobs_block(jobs.red, jobs.red_nparts, jobs.red_n, jobs.red_acc, (int)blockIdx.x % jobs.red_bx,
                      (int)blockIdx.x / jobs.red_bx, reinterpret_cast<float(*)[64]>(lds_dyn), threadIdx.x < 256);
    return;
  }
  if (jobs.dyn.bits) gemm_tn_wide_dyn<SUM3, NP, BCM>(jobs.job[0], jobs.job[1], jobs.dyn, id, ngemm, lds_dyn);
  else if (id < jobs.first1) gemm_tn_wide_body<SUM3, NP, BCM>(jobs.job[0], id, jobs.first1, lds_dyn);
  else gemm_tn_wide_body<false, NP>(jobs.job[1], id - jobs.first1, ngemm - jobs.first1, lds_dyn);
}

}  // namespace

// pieces per operand of a job (TnGemm.np: 0 / 3 = the exact split)
static inline int tnw_np(int np) { return np == 2 ? 2 : 3; }
int gemm_tn_wide_supported(const TnGemm& d) {
  static const int on = dev_env_int("COATTN_TN_WIDE", 1);    // developer switches: the kernel at all / at the exact width
  static const int on3 = dev_env_int("COATTN_TN_WIDE3", 1);
  return on && (d.np == 2 || on3) && gemm_tn_supported(d) && !d.bf16 && d.mask_blk == 0 && (d.M % BM) == 0 && (d.N % BN) == 0;
}

// split-K plan for `max_parts` parts (32 parts x 8 tiles of 128 x 256 = one workgroup per CU at d = 512)
int gemm_tn_wide_plan(const TnGemm& d, int max_parts, int* ksplit, int* S) {
  int want = max_parts / d.levels;                 // (the caller shares its budget of parts between the jobs of a launch)
  if (want < 1) want = 1;
  int ks = (d.K + want - 1) / want;
  ks = (ks + BK - 1) / BK * BK;
  *ksplit = ks;
  *S = (d.K + ks - 1) / ks;
  return d.levels * *S;
}

// rows of the gathered job a device-planned launch takes (its row map lives in LDS behind the images: 4 bytes per row at worst)
constexpr int kDynMaxRows = 8192;
int launch_gemm_tn_wide(const TnGemm* d, const int* ksplit, const int* S, int n, hipStream_t s, const TnReduce* red,
                        const WGemm* wextra, const TnDyn* dyn) {
  CA_CHECK_ARG(n == 1 || n == 2, "gemm_tn_wide: 1 or 2 jobs per launch");
  TwJobs jobs = {};
  if (dyn) {
    CA_CHECK_ARG(n == 2 && dyn->bits && dyn->plan && dyn->levels == d[1].levels && dyn->levels <= kDynLevels && d[0].levels == 1 && dyn->K0 == d[0].K && dyn->P > dyn->levels &&
                 dyn->words == (d[1].K + 31) / 32 && dyn->words <= 512 && d[1].K <= kDynMaxRows && d[0].M == d[1].M && d[0].N == d[1].N &&
                 d[1].b_kdiv == 0 && d[1].a_term == 0 && d[0].C == d[1].C,
                 "gemm_tn_wide: bad device-planned launch");
    jobs.dyn = *dyn;
  }
  if (wextra) {
    long nbw = 0;
    CA_CHECK_ARG(wextra->a_sk == 0 && tnw_np(wextra->np) == tnw_np(d[0].np) && !wextra->f16 && !wextra->bf16 && wextra->N % 256 == 0 && wextra->kband_n == 0,
                 "gemm_tn_wide: the extra GEMM must be a row-major product of the launch's width with N %% 256 == 0");
    CA_TRY(gemm_w_fill_job(*wextra, jobs.wj, &nbw, 256));
    jobs.nw = (int)nbw;
    static const int first = dev_env_int("COATTN_DQ_FIRST", 0);   // developer switch
    if (first) jobs.wfirst = (jobs.nw + 7) / 8 * 8;
  }
  if (red) {
    CA_CHECK_ARG(red->njobs >= 1 && red->njobs <= 4 && red->n > 0, "gemm_tn_wide: bad reduction jobs");
    for (int i = 0; i < red->njobs; ++i) { jobs.red.src[i] = red->src[i]; jobs.red.dst[i] = red->dst[i]; }
    jobs.red.njobs = red->njobs;
    const int nsum = red->sum_x[0] ? 2 : 0;
    for (int i = 0; i < nsum; ++i) { jobs.red.sum_x[i] = red->sum_x[i]; jobs.red.sum_out[i] = red->sum_out[i]; }
    jobs.red.sum_n = red->sum_n; jobs.red.ld = red->ld;
    jobs.red_bx = (int)((red->n + 63) / 64);
    jobs.nred = jobs.red_bx * (red->njobs + nsum);
    jobs.red_nparts = red->nparts; jobs.red_n = red->n; jobs.red_acc = red->accumulate;
  }
  long nb[2] = {0, 0};
  for (int i = 0; i < n; ++i) {
    CA_CHECK_ARG(gemm_tn_wide_supported(d[i]), "gemm_tn_wide: unsupported shape M=%d N=%d K=%d", d[i].M, d[i].N, d[i].K);
    CA_CHECK_ARG(d[i].A && (d[i].B || d[i].b_ptrs[0]) && d[i].C && ksplit[i] > 0 && (ksplit[i] % BK) == 0 && (long)S[i] * ksplit[i] >= d[i].K,
                 "gemm_tn_wide: bad arguments");
    CA_CHECK_ARG(i == 0 || (d[i].a_term == 0 && d[i].b_kdiv == 0), "gemm_tn_wide: only the first job may sum three A terms or have a k-contiguous B");
    TwArgs& g = jobs.job[i];
    g = TwArgs{};
    g.A = d[i].A; g.a_sl = d[i].a_sl; g.a_ld = d[i].a_ld; g.a_term = d[i].a_term;
    g.B = d[i].B; g.b_sl = d[i].b_sl; g.b_ld = d[i].b_ld;
    g.b_kdiv = d[i].b_kdiv; g.b_sdiv = d[i].b_sdiv; g.b_bytes = d[i].b_kdiv ? (unsigned)((long)(d[i].K / d[i].b_kdiv) * d[i].b_sdiv * 4) : 0u;
    for (int t = 0; t < 8; ++t) g.b_ptrs[t] = d[i].b_ptrs[t];
    g.C = d[i].C; g.M = d[i].M; g.N = d[i].N; g.K = d[i].K; g.ksplit = ksplit[i]; g.S = S[i];
    nb[i] = (long)(d[i].M / BM) * (d[i].N / BN) * d[i].levels * S[i];
  }
  CA_CHECK_ARG(nb[0] + nb[1] < 2147483647L, "gemm_tn_wide: grid too large");
  jobs.first1 = (int)nb[0];
  const bool sum3 = d[0].a_term != 0, bcm = d[0].b_kdiv != 0;
  if (dyn) {                                          // P parts x tiles in all; the kernel shares them between the jobs
    nb[0] = (long)(d[0].M / BM) * (d[0].N / BN) * dyn->P;
    nb[1] = 0;
  }
  const dim3 grid((unsigned)(jobs.nred + nb[0] + nb[1] + (jobs.wfirst ? jobs.wfirst : jobs.nw)));
  // two buffers of NP pieces: 57,344 B at two pieces (69,632 B with the [column][k] image of a BCM first job), 86,016 /
  // 104,448 B at the exact width -- above the 64 KB default: the attribute is set once per device, for every instantiation, to
  // the LARGEST request any call can make (not to the first call's value: ADVICE r4)
  const int npv = tnw_np(d[0].np);
  CA_CHECK_ARG(n == 1 || tnw_np(d[1].np) == npv, "gemm_tn_wide: the jobs of a launch share the width");
  size_t lds = (size_t)2 * npv * (IMGA + (bcm ? IMGBC : IMGB)) * sizeof(short);
  if (dyn) lds += (size_t)(520 + (d[1].K + 15) / 16 * 16 + 3 * BK) * sizeof(int);      // prefix sums + the row map of the longest possible part
  if (wextra && lds < (size_t)2 * npv * gw::BM * gw::LDR * sizeof(short)) lds = (size_t)2 * npv * gw::BM * gw::LDR * sizeof(short);
  {
    constexpr size_t kImg = (size_t)2 * 3 * (IMGA + (IMGBC > IMGB ? IMGBC : IMGB)) * sizeof(short);
    constexpr size_t kW = (size_t)2 * 3 * gw::BM * gw::LDR * sizeof(short);
    constexpr size_t kLdsMax = (kImg > kW ? kImg : kW) + (size_t)(520 + kDynMaxRows + 3 * BK) * sizeof(int);
    static DeviceOnce once;
    CA_TRY(once.run([&] {
      const void* ks[8] = {reinterpret_cast<const void*>(gemm_tn_wide_kernel<true, 2, true>), reinterpret_cast<const void*>(gemm_tn_wide_kernel<false, 2, true>),
                           reinterpret_cast<const void*>(gemm_tn_wide_kernel<true, 2>), reinterpret_cast<const void*>(gemm_tn_wide_kernel<false, 2>),
                           reinterpret_cast<const void*>(gemm_tn_wide_kernel<true, 3, true>), reinterpret_cast<const void*>(gemm_tn_wide_kernel<false, 3, true>),
                           reinterpret_cast<const void*>(gemm_tn_wide_kernel<true, 3>), reinterpret_cast<const void*>(gemm_tn_wide_kernel<false, 3>)};
      hipError_t e = hipSuccess;
      for (int i = 0; i < 8 && e == hipSuccess; ++i) e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
      return e;
    }, "gemm_tn_wide"));
  }
  if (npv == 2) {
    if (sum3 && bcm) hipLaunchKernelGGL((gemm_tn_wide_kernel<true, 2, true>), grid, dim3(NTHR), lds, s, jobs);
    else if (bcm) hipLaunchKernelGGL((gemm_tn_wide_kernel<false, 2, true>), grid, dim3(NTHR), lds, s, jobs);
    else if (sum3) hipLaunchKernelGGL((gemm_tn_wide_kernel<true, 2>), grid, dim3(NTHR), lds, s, jobs);
    else hipLaunchKernelGGL((gemm_tn_wide_kernel<false, 2>), grid, dim3(NTHR), lds, s, jobs);
  } else {
    if (sum3 && bcm) hipLaunchKernelGGL((gemm_tn_wide_kernel<true, 3, true>), grid, dim3(NTHR), lds, s, jobs);
    else if (bcm) hipLaunchKernelGGL((gemm_tn_wide_kernel<false, 3, true>), grid, dim3(NTHR), lds, s, jobs);
    else if (sum3) hipLaunchKernelGGL((gemm_tn_wide_kernel<true, 3>), grid, dim3(NTHR), lds, s, jobs);
    else hipLaunchKernelGGL((gemm_tn_wide_kernel<false, 3>), grid, dim3(NTHR), lds, s, jobs);
  }
  CA_CHECK_LAUNCH("gemm_tn_wide");
  return 0;
}
