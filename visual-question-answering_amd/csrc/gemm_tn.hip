// fp32-accurate weight-gradient GEMM  C = A^T B  on v_mfma_f32_32x32x16_bf16 (gfx950).
//
//   part[z][m][n] = sum_{k in chunk(z)} A_l[k][m] * B_l[k][n]        A_l, B_l row-major, rows = contraction index
//
// dW_v = dP_v^T V (k = (sample, location)) and dW_q = sum_l dP_q,l^T Q_l (k = (sample, word)) of the co-attention
// backward (the nn.Linear weight gradients of model.py:380-384): a d x d result contracted over tens of thousands
// of rows, so the parallelism is split-K -- part z = (level l, chunk p) covers rows [p ksplit, (p+1) ksplit) of
// level l, and a deterministic reduce adds the parts (launch_reduce_partials).
//
// Tile 128 x 128 per 256-thread workgroup (4 waves, 2 x 2 MFMA tiles each), BK = 16.  Both operands arrive
// contiguous along their TILE index, not along k: each float4 (4 consecutive m of one k) is split into its three
// exact bf16 pieces and staged as [k][m] images; the MFMA fragments (8 consecutive k of one row) come back through
// ds_read_b64_tr_b16, the transposing LDS read of gfx950.  LDS images and fragment registers are double-buffered:
// one barrier per step, and everything of step s + 1 (split arithmetic, LDS writes, the fragment reads after the
// barrier) and the global loads of step s + 2 sit behind the 24 MFMAs of step s, placed by hand (one scheduling
// fence per MFMA).  Numerics: the six partial products of gemm.hip's split mode, same order.
#include "common.h"
#include "fused.h"
#include "gemm_w_body.h"
#include <type_traits>

// Developer switches (tools/ab_gemmtn.sh, never in the shipped library; wrong results): GEMMTN_KO bit 0 no MFMAs, 1 no global
// loads past the prologue, 2 no split arithmetic, 3 no LDS writes, 4 no fragment reads past the prologue, 5 no barrier.
// Measured at cfg 2, two-piece width (N = 196: 135 us for the whole backward launch): no MFMAs -1 us (hidden), no loads
// -47 (with them gone the split is loop-invariant and leaves too), no split -31, no LDS writes -36 (the split feeding them
// is dead code then), no fragment reads -3, no barrier 0; everything but the dQ tiles and the epilogues off: 51 us.  The
// split arithmetic of operands that every one of the four tiles sharing them repeats is the largest single item.
// Requesting the rows two steps ahead instead of one (two register sets): 142 us, no gain -- it is not latency.
#ifndef GEMMTN_KO
#define GEMMTN_KO 0
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDT = BM + 32;                   // [k][row] image row stride (elements): conflict-free writes + transposed reads
constexpr int IMG = BK * LDT;                  // one piece image
// BCM (B contiguous along the contraction index: channel-major image features): [col][k] images, 48-byte rows
// (conflict-free ds_read_b128 fragments)
constexpr int LDRB = 24, IMGB = BN * LDRB;

struct TnArgs {
  const float* A; long a_sl; int a_ld;         // level l at A + l * a_sl
  long a_term;                                 // SUM3: A = A0 + A1 + A2, term t at A + t * a_term
  const float* B; const float* b_ptrs[8]; long b_sl; int b_ld;
  int b_kdiv; long b_sdiv; unsigned b_bytes;   // BCM: element (k, n) at (k / b_kdiv) * b_sdiv + k % b_kdiv + n * b_ld
  float* C;                                    // parts [L * S][M][N]
  int mask_blk; unsigned tile_mask;            // mask_blk > 0: only the tiles whose (row, column) block bit is set exist
  int M, N, K, ksplit, S;
};

// blocks [0, nred): the small reductions riding along (reduce_jobs_block, red_bx blocks per row); then
// [nred, nred + first1): job 0; the rest: job 1
// the last nw blocks: tiles of a pre-split-weight GEMM (gemm_w_body.h) sharing the launch (dQ = dP_q W_q of the backward)
struct TnJobs {
  TnArgs job[2]; int first1; ReduceJobs red; int nred, red_bx, red_nparts, red_acc; long red_n;
  gw::WArgs wj; int nw;
};

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// SUM3: the A operand is the sum of three equally shaped arrays, added while staging (dP_v of the three question
// levels: saves the separate pass that would sum them in memory).  A job with a_term == 0 in a SUM3 launch reads
// its two extra terms through an empty buffer resource: zeros, without memory traffic.
// Up to two jobs per launch (dW_v and dW_q): the second one's workgroups fill the slots the first leaves idle.
// BCM: the B operand is contiguous along k inside groups of b_kdiv rows (channel-major image features [B, d, N]:
// k = (sample, location), n = channel): float4 = 4 consecutive k of one column, [col][k] images, ds_read_b128.
// NP: bf16 pieces per operand (gemm_w_body.h) -- 3: the exact split, six products; 2: hi + mid, three products (every other
// slot of the step carries an MFMA); 1: reduced-precision mode, the hi pieces alone, one MFMA per product.
template <bool SUM3, bool BCM, int NP>
__device__ __forceinline__ void gemm_tn_body(const TnArgs& g, const int bid, const int nblk, short* const lds) {
  static_assert(NP >= 1 && NP <= 3, "pieces per operand");
  constexpr bool P1 = NP == 1;
  // (only the pieces of the width are staged: two pieces -> 41 KB of LDS per workgroup)
  constexpr int OPER = NP * IMG, OPERB = NP * IMGB;
  constexpr int BUF = OPER + (BCM ? OPERB : OPER);                         // elements of one LDS buffer
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
  // XCD-aware order (workgroup i runs on XCD i % 8): the (part, tile) pairs in part-major order are cut into eight runs,
  // one per XCD -- the tiles of one part (they read the same rows of A and B) then share an XCD's L2 and run side by
  // side, whatever the number of parts (before: only for the first parts & ~7 of them; 23 parts left 7 spread over all
  // eight XCDs, their operands fetched once per XCD)
  const int ntn = g.N / BN, ntiles = (g.M / BM) * ntn;
  int z, t;
  {
    int lin = bid;
    if ((nblk & 7) == 0) lin = (bid & 7) * (nblk >> 3) + (bid >> 3);
    z = lin / ntiles; t = lin % ntiles;
  }
  if (g.mask_blk > 0 && !((g.tile_mask >> (((t / ntn) / g.mask_blk) * 3 + (t % ntn) / g.mask_blk)) & 1u)) return;
  const int m0 = (t / ntn) * BM, n0 = (t % ntn) * BN;
  const int lvl = z / g.S, p = z % g.S;
  const int kbeg = p * g.ksplit, kend = min(g.K, kbeg + g.ksplit);
  const int steps = (kend - kbeg + BK - 1) / BK;
  const float* Ab = g.A + (long)lvl * g.a_sl;
  const float* Bb = g.b_ptrs[0] ? g.b_ptrs[lvl & 7] : g.B + (long)lvl * g.b_sl;
  // rows past K read 0 (resource bound); rows past kend belong to the next part: ksplit % 16 == 0, so a step never straddles
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(Ab, (unsigned)((long)g.K * g.a_ld * 4));
  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(Bb, BCM ? g.b_bytes : (unsigned)((long)g.K * g.b_ld * 4));
  const unsigned tbytes = (SUM3 && g.a_term) ? (unsigned)((long)g.K * g.a_ld * 4) : 0u;
  const __amdgpu_buffer_rsrc_t rs_a1 = make_rsrc(Ab + (SUM3 ? g.a_term : 0), tbytes);
  const __amdgpu_buffer_rsrc_t rs_a2 = make_rsrc(Ab + (SUM3 ? 2 * g.a_term : 0), tbytes);

  // staging: per operand and step 2 float4 per thread; a wave's load covers 2 k-rows x 512 B
  const int sk = tid >> 5, sm = (tid & 31) * 4;
  const int a_voff = ((kbeg + sk) * g.a_ld + m0 + sm) * 4, b_voff = ((kbeg + sk) * g.b_ld + n0 + sm) * 4;
  const int a_step = BK * g.a_ld * 4, b_step = BK * g.b_ld * 4, a_half = 8 * g.a_ld * 4, b_half = 8 * g.b_ld * 4;
  const int st_off = sk * LDT + sm;                                        // + 8 * LDT for the second float4
  // BCM staging: 4 lanes cover the 16 k (64 B) of one column, 16 columns per wave load; the second float4 is
  // column + 64.  The (sample, row) position of the k group is tracked step by step (no division in the loop).
  const int bc_col = tid >> 2, bc_kq = tid & 3;
  int bc_n = 0, bc_voff = 0;
  if (BCM) {
    const int kg = kbeg + 4 * bc_kq;
    bc_n = kg % g.b_kdiv;
    bc_voff = (int)(((long)(kg / g.b_kdiv) * g.b_sdiv + bc_n + (long)(n0 + bc_col) * g.b_ld) * 4);
  }
  const int bc_st = OPER + bc_col * LDRB + 4 * bc_kq;                      // + 64 * LDRB for the second float4
  // transposed fragment read: each 16-lane group fetches a 4 (k) x 16 (rows) block; lane 4q+p of the group supplies
  // the address of block row q, columns 4p..4p+3, and receives the 4 k of row (lane & 15)
  const int tr_off = (8 * lh + ((lane & 15) >> 2)) * LDT + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int a_rd = tr_off + wr * 64;
  const int b_rd = BCM ? OPER + (wc * 64 + li) * LDRB + 8 * lh : OPER + tr_off + wc * 64;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 raw[4];                                  // A k 0..7, A k 8..15, B k 0..7, B k 8..15 of the step being staged
  f32x4 rawt[2][2];                              // SUM3: the other two terms of raw[0], raw[1]
  bf16x4 fa[2][3][2][2], fb[2][3][2][2];         // [set][piece][tile][k half]: fragment = {lo, hi}
  unsigned ph[2], pm[2], pl[2];
  float ra[2], rb[2];
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first (gemm.hip's order)
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
  auto load_raw = [&](int x, int s) {            // x: 0,1 = A halves, 2,3 = B halves; step s of this part
    if ((GEMMTN_KO & 2) && s >= 2) return;
    // (the step goes into the VECTOR offset: the resource's range check does not see the scalar offset)
    if (x < 2) {
      raw[x] = buf_load4(rs_a, a_voff + (x & 1) * a_half + s * a_step, 0);
      if (SUM3) {
        rawt[x][0] = buf_load4(rs_a1, a_voff + (x & 1) * a_half + s * a_step, 0);
        rawt[x][1] = buf_load4(rs_a2, a_voff + (x & 1) * a_half + s * a_step, 0);
      }
    } else if (BCM) {
      raw[x] = buf_load4(rs_b, bc_voff + (x & 1) * 64 * g.b_ld * 4, 0);
      if (x == 3) {                              // both columns of this step requested: on to the next 16 k
        bc_n += BK; bc_voff += BK * 4;
        if (bc_n >= g.b_kdiv) { bc_n -= g.b_kdiv; bc_voff += (int)((g.b_sdiv - g.b_kdiv) * 4); }
      }
    } else raw[x] = buf_load4(rs_b, b_voff + (x & 1) * b_half + s * b_step, 0);
  };
  auto stage = [&](int x, int e, int st) {       // split of raw[x], pair e, in three stages of 5, 5 and 1 VALU
    if (GEMMTN_KO & 4) {
      if (st == 0) ph[e] = pm[e] = pl[e] = __builtin_bit_cast(unsigned, raw[x][2 * e]);
      return;
    }
    if (st == 0) {
      if (SUM3 && x < 2) {                       // (level order 0 + 1 + 2, as the separate summing pass adds them)
        raw[x][2 * e] = (raw[x][2 * e] + rawt[x][0][2 * e]) + rawt[x][1][2 * e];
        raw[x][2 * e + 1] = (raw[x][2 * e + 1] + rawt[x][0][2 * e + 1]) + rawt[x][1][2 * e + 1];
      }
      ph[e] = cvt_pk_bf16(raw[x][2 * e], raw[x][2 * e + 1]);
      if (P1) return;
      ra[e] = sub1(raw[x][2 * e], __builtin_bit_cast(float, ph[e] << 16));
      rb[e] = sub1(raw[x][2 * e + 1], __builtin_bit_cast(float, ph[e] & 0xffff0000u));
    } else if (P1) {
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
    if ((GEMMTN_KO & 8) && buf != lds) return;       // (the prologue's image is still written)
    const u32x2 v = q == 0 ? u32x2{ph[0], ph[1]} : (q == 1 ? u32x2{pm[0], pm[1]} : u32x2{pl[0], pl[1]});
    if (BCM && x >= 2) *reinterpret_cast<u32x2*>(&buf[q * IMGB + (x & 1) * 64 * LDRB + bc_st]) = v;
    else *reinterpret_cast<u32x2*>(&buf[(x >> 1) * OPER + q * IMG + (x & 1) * 8 * LDT + st_off]) = v;
  };
  // fragment reads in the order of first use: a2, b0, a0, b2, a1, b1 (tile 0, tile 1; lo, hi): r = 0..23
  auto read_frag = [&](auto SETc, const short* buf, int r) {
    constexpr int SET = decltype(SETc)::value;
    constexpr int QA[3] = {2, 0, 1}, QB[3] = {0, 2, 1};
    const int grp = r >> 2, isb = grp & 1, q = isb ? QB[grp >> 1] : QA[grp >> 1], tile = (r >> 1) & 1, hi = r & 1;
    if (q >= NP) return;
    if ((GEMMTN_KO & 16) && buf != lds) return;
    if (BCM && isb) {                            // one 16-byte read per fragment (issued with its first half)
      if (hi == 0) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(buf + q * IMGB + b_rd + tile * 32 * LDRB);
        fb[SET][q][tile][0] = bf16x4{v[0], v[1], v[2], v[3]};
        fb[SET][q][tile][1] = bf16x4{v[4], v[5], v[6], v[7]};
      }
      return;
    }
    const short* ptr = buf + q * IMG + (isb ? b_rd : a_rd) + tile * 32 + hi * 4 * LDT;
    if (isb) fb[SET][q][tile][hi] = lds_tr16(ptr);
    else fa[SET][q][tile][hi] = lds_tr16(ptr);
  };
  auto frag = [&](const bf16x4 (&f)[2]) { return bf16x8{f[0][0], f[0][1], f[0][2], f[0][3], f[1][0], f[1][1], f[1][2], f[1][3]}; };
  // one 16-k step: MFMAs on fragment set SET; raw (step s + 1) is split into image `nxt`, re-requested for step
  // s + 2, and after the barrier the fragments of step s + 1 are read into the other set
  auto step = [&](auto SETc, int s, short* nxt) {
    constexpr int SET = decltype(SETc)::value;
    using OTHER = std::integral_constant<int, SET ^ 1>;
#pragma unroll
    for (int n = 0; n < 24; ++n) {
      // the MFMA of slot n: all 24 (NP = 3), every other slot (NP = 2: products 3 .. 5), the last four (NP = 1)
      const int mi = NP == 2 ? n >> 1 : n;
      const bool mf = NP == 3 || (NP == 2 && (n & 1)) || (NP == 1 && n >= 20);
      const int tt = NP == 2 ? 3 + (mi >> 2) : n >> 2, i = (mi >> 1) & 1, j = mi & 1;
      if (mf && (GEMMTN_KO & 1)) acc[i][j][n & 15] += __builtin_bit_cast(float, (int)fa[SET][PA[tt]][i][0][0] ^ (int)fb[SET][PB[tt]][j][0][0]);
      if (mf && !(GEMMTN_KO & 1))
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(fa[SET][PA[tt]][i]), frag(fb[SET][PB[tt]][j]), acc[i][j], 0, 0, 0);
      // raw[x]: pair 0 stages in slots 4x, 4x+1, 4x+2; pair 1 in 4x+1, 4x+2, 4x+3; pieces written in 4x+3 .. 4x+5
      if (n < 16) {
        const int x = n >> 2, u = n & 3;
        if (u <= 2) stage(x, 0, u);
        if (u >= 1) stage(x, 1, u - 1);
        if (u == 3) load_raw(x, s + 2);
      }
      if (n >= 3 && n < 18) {
        const int w = n - 3, x = w >> 2, q = w & 3;
        if (q < 3) write_piece(nxt, x, q);
      }
      if (n == 18 && !(GEMMTN_KO & 32)) lds_barrier();
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
  for (int x = 0; x < 4; ++x) load_raw(x, 0);
#pragma unroll
  for (int x = 0; x < 4; ++x) {
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

// Up to two jobs per launch (dW_v and dW_q): the second one's workgroups fill the slots the first leaves idle.
// SUM3 / BCM describe job 0; job 1 is always plain.  The backward's whole GEMM work is ONE launch of this kernel:
// [small reductions][dW_v parts][dW_q parts][tiles of dQ = dP_q W_q on the gemm_w body].
template <bool SUM3, bool BCM, int NP>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TnJobs jobs) {   // (three workgroups per CU at two pieces -- 168 VGPRs -- spill: 135 -> 163 us)
  extern __shared__ __attribute__((aligned(16))) short lds_dyn[];          // 2 buffers: 61,440 B at three pieces (BCM: 67,584 B)
  const int id = (int)blockIdx.x - jobs.nred, ngemm = (int)gridDim.x - jobs.nred - jobs.nw;
  if (id >= ngemm) {                                 // (61,440 B of the dynamic LDS)
    gw::gemm_w_body<false, NP>(jobs.wj, id - ngemm, lds_dyn);
    return;
  }
  if (id < 0) {                                      // the backward's small parameter-gradient reductions: a few
    reduce_jobs_block(jobs.red, jobs.red_nparts, jobs.red_n, jobs.red_acc, (int)blockIdx.x % jobs.red_bx,   // short
                      (int)blockIdx.x / jobs.red_bx, reinterpret_cast<float(*)[64]>(lds_dyn));   // workgroups, first in the grid
    return;
  }
  if (id < jobs.first1) gemm_tn_body<SUM3, BCM, NP>(jobs.job[0], id, jobs.first1, lds_dyn);
  else gemm_tn_body<false, false, NP>(jobs.job[1], id - jobs.first1, ngemm - jobs.first1, lds_dyn);
}

}  // namespace

int gemm_tn_supported(const TnGemm& d) {
  auto pal = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  bool ok = !d.a_bf16 && d.M > 0 && d.N > 0 && (d.M % BM) == 0 && (d.N % BN) == 0 && d.K >= BK && d.levels >= 1 && d.levels <= 8 &&
            (d.a_ld & 3) == 0 && (d.b_ld & 3) == 0 && (d.a_term & 3) == 0 && (d.a_sl & 3) == 0 && (d.b_sl & 3) == 0 && pal(d.A) &&
            (d.b_ptrs[0] ? true : pal(d.B)) && (long)(d.K + 2 * BK) * d.a_ld * 4 < 0x40000000L;
  if (d.b_kdiv) {    // B contiguous along k inside groups of b_kdiv rows
    ok = ok && d.levels == 1 && !d.b_ptrs[0] && d.b_kdiv >= BK && (d.b_kdiv & 3) == 0 && (d.b_sdiv & 3) == 0 &&
         (d.K % d.b_kdiv) == 0 && ((long)(d.K / d.b_kdiv + 2) * d.b_sdiv + (long)d.N * d.b_ld) * 4 < 0x40000000L;
  } else {
    ok = ok && (long)(d.K + 2 * BK) * d.b_ld * 4 < 0x40000000L;
  }
  for (int t = 0; t < 8; ++t) ok = ok && pal(d.b_ptrs[t]);
  return ok ? 1 : 0;
}

// parts = levels * S with S = ceil(K / ksplit); ksplit is chosen so that the grid fills the 512 workgroup slots once
int gemm_tn_plan(const TnGemm& d, int max_parts, int* ksplit, int* S) {
  const int ntiles = (d.M / BM) * (d.N / BN);
  int want = (512 + ntiles - 1) / ntiles / d.levels;          // parts per level
  if (want * d.levels > max_parts) want = max_parts / d.levels;
  if (want < 1) want = 1;
  int ks = (d.K + want - 1) / want;
  ks = (ks + BK - 1) / BK * BK;
  *ksplit = ks;
  *S = (d.K + ks - 1) / ks;
  return d.levels * *S;
}

static int fill_job(const TnGemm& d, int ksplit, int S, TnArgs& g, long* nblk) {
  CA_CHECK_ARG(gemm_tn_supported(d), "gemm_tn: unsupported shape M=%d N=%d K=%d", d.M, d.N, d.K);
  CA_CHECK_ARG(d.A && (d.B || d.b_ptrs[0]) && d.C && ksplit > 0 && (ksplit % BK) == 0 && (long)S * ksplit >= d.K, "gemm_tn: bad arguments");
  g = TnArgs{};
  g.A = d.A; g.a_sl = d.a_sl; g.a_ld = d.a_ld; g.a_term = d.a_term;
  g.B = d.B; g.b_sl = d.b_sl; g.b_ld = d.b_ld;
  g.b_kdiv = d.b_kdiv; g.b_sdiv = d.b_sdiv; g.b_bytes = d.b_kdiv ? (unsigned)((long)(d.K / d.b_kdiv) * d.b_sdiv * 4) : 0u;
  for (int t = 0; t < 8; ++t) g.b_ptrs[t] = d.b_ptrs[t];
  g.C = d.C; g.M = d.M; g.N = d.N; g.K = d.K; g.ksplit = ksplit; g.S = S;
  g.mask_blk = d.mask_blk; g.tile_mask = d.tile_mask;
  *nblk = (long)(d.M / BM) * (d.N / BN) * d.levels * S;
  return 0;
}

// one launch for n = 1 or 2 GEMMs (ksplit[i], S[i] from gemm_tn_plan)
// red (may be NULL): small reductions done by extra workgroups of the same launch (launch_reduce_jobs's arguments)
// wextra (may be NULL): one pre-split-weight GEMM (row-major A) whose tiles run as the last workgroups of the launch
int launch_gemm_tn(const TnGemm* d, const int* ksplit, const int* S, int n, hipStream_t s, const TnReduce* red,
                   const WGemm* wextra) {
  CA_CHECK_ARG(n == 1 || n == 2, "gemm_tn: 1 or 2 jobs per launch");
  TnJobs jobs = {};
  if (wextra) {
    long nbw = 0;
    CA_CHECK_ARG(wextra->a_sk == 0, "gemm_tn: the extra GEMM's A operand must be row-major");
    CA_TRY(gemm_w_fill_job(*wextra, jobs.wj, &nbw));
    jobs.nw = (int)nbw;
  }
  if (red) {
    CA_CHECK_ARG(red->njobs >= 1 && red->njobs <= 4 && red->n > 0, "gemm_tn: bad reduction jobs");
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
  for (int i = 0; i < n; ++i) CA_TRY(fill_job(d[i], ksplit[i], S[i], jobs.job[i], &nb[i]));
  CA_CHECK_ARG(nb[0] + nb[1] < 2147483647L, "gemm_tn: grid too large");
  jobs.first1 = (int)nb[0];
  CA_CHECK_ARG(n == 1 || (d[1].a_term == 0 && d[1].b_kdiv == 0), "gemm_tn: only the first job may sum three A terms or have a k-contiguous B");
  const bool sum3 = d[0].a_term != 0, bcm = d[0].b_kdiv != 0;
  const dim3 grid((unsigned)(jobs.nred + nb[0] + nb[1] + jobs.nw));
  const int np = d[0].bf16 ? 1 : (d[0].np == 2 ? 2 : 3);
  // two buffers of the staged pieces; the dQ projection's tiles (gemm_w_body) need 2 * np * 10,240 B of it
  size_t lds = (size_t)2 * np * (IMG + (bcm ? IMGB : IMG)) * sizeof(short);
  if (wextra && lds < (size_t)2 * np * gw::BM * gw::LDR * sizeof(short)) lds = (size_t)2 * np * gw::BM * gw::LDR * sizeof(short);
  // Two workgroups per CU, not three: at two pieces the kernel needs 161 VGPRs and 40 - 45 KB of LDS, and a third
  // workgroup per CU -- which the hardware then grants -- costs the weight-gradient parts more in L2 than it hides
  // (channel-major cfg 2 at N = 196: 176 us against 165).  The LDS request is what holds it at two (3 x 54 KB > 160 KB);
  // developer switch COATTN_TN_LDS=<bytes> (0: as computed).
  static const int lds_env = dev_env_int("COATTN_TN_LDS", 55296);
  const size_t lds_max = (size_t)2 * 3 * (IMG + IMGB) * sizeof(short);     // (three pieces, BCM: 67,584 B)
  if (np == 2 && lds_env > 0 && (size_t)lds_env > lds && (size_t)lds_env <= (bcm ? lds_max : (size_t)65536)) lds = (size_t)lds_env;
  auto np_of = [](int bf16, int npf) { return bf16 ? 1 : (npf == 2 ? 2 : 3); };
  CA_CHECK_ARG((n == 1 || np_of(d[1].bf16, d[1].np) == np) && (!wextra || np_of(wextra->bf16, wextra->np) == np),
               "gemm_tn: the jobs of a launch share the precision mode");
  if (bcm) {                                             // 67,584 B of dynamic LDS: above the 64 KB default limit
    static DeviceOnce once;
    CA_TRY(once.run([&] {
      hipError_t e = hipSuccess;
      const void* fns[6] = {reinterpret_cast<const void*>(gemm_tn_kernel<true, true, 3>), reinterpret_cast<const void*>(gemm_tn_kernel<false, true, 3>),
                            reinterpret_cast<const void*>(gemm_tn_kernel<true, true, 2>), reinterpret_cast<const void*>(gemm_tn_kernel<false, true, 2>),
                            reinterpret_cast<const void*>(gemm_tn_kernel<true, true, 1>), reinterpret_cast<const void*>(gemm_tn_kernel<false, true, 1>)};
      for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
      return e;
    }, "gemm_tn"));
  }
  auto go = [&](auto NPc) {
    constexpr int NP = decltype(NPc)::value;
    if (sum3 && bcm) hipLaunchKernelGGL((gemm_tn_kernel<true, true, NP>), grid, dim3(256), lds, s, jobs);
    else if (bcm) hipLaunchKernelGGL((gemm_tn_kernel<false, true, NP>), grid, dim3(256), lds, s, jobs);
    else if (sum3) hipLaunchKernelGGL((gemm_tn_kernel<true, false, NP>), grid, dim3(256), lds, s, jobs);
    else hipLaunchKernelGGL((gemm_tn_kernel<false, false, NP>), grid, dim3(256), lds, s, jobs);
  };
  if (np == 1) go(std::integral_constant<int, 1>()); else if (np == 2) go(std::integral_constant<int, 2>()); else go(std::integral_constant<int, 3>());
  CA_CHECK_LAUNCH("gemm_tn");
  return 0;
}
