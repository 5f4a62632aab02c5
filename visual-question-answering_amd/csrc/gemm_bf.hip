// Single-product bf16 GEMM against a WEIGHT operand for the reduced-precision mode (COATTN_FLAG_BF16_PROJ), gfx950:
//
//   C[z][m][n] = (sum_k bf16(A[z][m][k]) * bf16(Bw(k, n)) + bias_n[n]) * out_scale        fp32 accumulation / output
//
// the projections P_v = V W_v^T, P_q = Q W_q^T, dQ = dP_q W_q (model.py:380-384 and their autograd) and the phrase level's
// Z = Xcat Wcat^T / dXcat = dZ Wcat at config 4's sizes (d = 2048: M = 4,160-12,480, N = K = 2,048-6,144).
//
// Why its own kernel: gemm_w.hip is scheduled around the SIX MFMAs of an fp32-accurate product.  With one MFMA per
// product the arithmetic is 6x cheaper and what binds is the path into the CU: a 128 x 128 tile with fp32 A rows and
// per-wave B fragments needs 1 byte per 33 flops -- at the ~64 B/clk a CU's L1 takes, a third of the MFMA rate (measured
// with gemm_w's single-piece mode: 678 TFLOP/s = 0.27 of the bf16 peak; 766 with 128 x 256 tiles).  Here:
//   * tile 256 x 256 per 512-thread workgroup (8 waves as 2 x 4, 128 x 64 = 4 x 2 MFMA tiles each), 64 k per step: one
//     workgroup per CU, 87 flops per byte fetched;
//   * A: fp32 rows in whole 256-byte lines -> registers -> v_cvt_pk_bf16_f32 -> LDS [256][64 + 8] (conflict-free
//     ds_read_b128 fragments), the rows of step s + 1 requested before the 32 MFMAs of step s;
//   * B: the weight is pre-rounded ONCE per call into a fragment-ordered bf16 image (wsplit, pieces = 1: 1 KB per
//     (32-column tile, 16-k step), lane-major) and goes global -> LDS by LDS-DMA (no registers, no VALU), 32 chunks per
//     step, read back lane-linear; shared by the two waves that need the same columns;
//   * LDS images double-buffered (2 x 68 KB), one barrier per step.
// Shapes: N % 256 == 0, K (and k bands) % 64 == 0, A rows 16-byte aligned; everything else stays on gemm_w.
#include "common.h"
#include "fused.h"
#include "gemm_w_body.h"
#include <type_traits>

#ifndef BF_EARLY
#define BF_EARLY 1
#endif
#ifndef GEMMBF_KO
#define GEMMBF_KO 0        // developer knock-outs (wrong results), tools/ab_gemmbf.sh: 1 no A reloads, 2 no weight DMA, 4 no MFMAs, 8 no A LDS writes, 16 no fragment re-reads
#endif

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int LDA = TK + 8;                       // bf16 elements per staged A row (144 B)
constexpr int A_IMG = TM * LDA;                   // shorts: 36,864 B
constexpr int B_IMG = (TN / 32) * (TK / 16) * 512;   // shorts: 32 chunks of 1 KB
constexpr int BUF = A_IMG + B_IMG;                // one buffer: 69,632 B
constexpr int kChunk = 1024;                      // bytes of a (32-column tile, 16-k step) chunk of the hi-only image

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;

struct BfJobs { gw::WArgs job[2]; int first1; };

// ABF: A is stored as bf16 (the fused backward's dP_q in the reduced-precision mode; bf16 activations through
// coattn_linear_forward): 16-byte loads of 8 elements, half as many per step, written to the same image as they are.
template <bool ABF>
__device__ __forceinline__ void gemm_bf_body(const gw::WArgs& g, const int id, short* const smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, li = lane & 31, lh = lane >> 5;
  // XCD-aware order: the column tiles of a row tile run on one XCD, one after the other (they re-read the same A rows)
  const int ntm = (g.M + TM - 1) / TM, ntn = g.N / TN;
  const int x = id & 7, slot = id >> 3, per = ntn * ((ntm + 7) / 8);
  const int z = slot / per, t = slot % per, mt = (t / ntn) * 8 + x;
  if (mt >= ntm) return;
  const int m0 = mt * TM, n0 = (g.kband_n > 0 ? ntn - 1 - t % ntn : t % ntn) * TN;
  constexpr int ES = ABF ? 2 : 4;                 // bytes of a stored A element
  constexpr int NA = ABF ? 4 : 8;                 // 16-byte loads per thread and step
  constexpr int RP = ABF ? 64 : 32;               // rows per load sweep of the workgroup
  // (one select between a table entry and a computed address: a select between two LOADED fields of the by-value
  //  argument struct makes hipcc copy the whole struct to scratch)
  const char* Ab = g.a_ptrs[0] ? reinterpret_cast<const char*>(g.a_ptrs[z & 7]) : reinterpret_cast<const char*>(g.A) + (long)z * g.a_sz * ES;
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(Ab, (unsigned)(((long)(g.M - 1) * g.a_sm + g.K) * ES));
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(g.Wf, g.wf_bytes);
  int k_lo = 0, k_hi = g.K;
  if (g.kband_n > 0) {
    const int band = n0 / g.kband_n;
    k_lo = g.kband_lo[band]; k_hi = g.kband_hi[band];
  }
  const int KS = (k_hi - k_lo) / TK;              // (host check: multiples of 64)

  // A staging: 8 float4 per thread and step, a wave's load covers 4 rows x 256 B (bf16 A: 4 loads, 8 rows x 128 B)
  const int a_row = ABF ? tid >> 3 : tid >> 4, a_k = ABF ? (tid & 7) * 8 : (tid & 15) * 4;
  int a_voff[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) a_voff[i] = (m0 + a_row + RP * i) < g.M ? ((m0 + a_row + RP * i) * g.a_sm + a_k) * ES : 0x40000000;
  const int a_wr = a_row * LDA + a_k;             // + RP i rows
  const int a_rd = (wr * 128 + li) * LDA + 8 * lh;   // + 32 mt rows, + 16 ks
  // B: wave w moves the four 16-k chunks of column tile w of this step (lane-linear 1 KB each)
  const int b_src = ((n0 / 32 + wave) * (g.K / 16)) * kChunk;        // + (k / 16) chunks; lane part in the vector offset
  const int b_dst = A_IMG + wave * 4 * 512;                           // shorts
  const int b_rd = A_IMG + (wc * 2) * 4 * 512 + lane * 8;             // + (jj * 4 + ks) * 512

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 raw[NA];                                  // (bf16 A: 8 elements per register quad)
  auto load_a = [&](int s) {
#pragma unroll
    for (int i = 0; i < NA; ++i) raw[i] = buf_load4(rs_a, a_voff[i], (k_lo + s * TK) * ES);
  };
  auto dma_b = [&](int s, short* buf) {
    const int c0 = b_src + ((k_lo + s * TK) / 16) * kChunk;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(buf + b_dst + ks * 512), 16, lane * 16, c0 + ks * kChunk, 0, 0);
  };
  auto write_one = [&](int i, short* nxt) {
    if constexpr (ABF) {
      *reinterpret_cast<f32x4*>(&nxt[RP * i * LDA + a_wr]) = raw[i];
    } else {
      const u32x2 v = {cvt_pk_bf16(raw[i][0], raw[i][1]), cvt_pk_bf16(raw[i][2], raw[i][3])};
      *reinterpret_cast<u32x2*>(&nxt[RP * i * LDA + a_wr]) = v;
    }
  };
  auto write_a = [&](short* buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) write_one(i, buf);
  };
  // One step: the 32 MFMAs of step s with everything else in their shadow, placed by hand (left alone the compiler puts
  // the staging behind the MFMAs, where both waves of a SIMD do it at the same time and the matrix pipe idles):
  //   * fragments of 16-k group ks + 1 are read during the MFMAs of group ks (two fragment sets);
  //   * after every fourth MFMA one of the eight staged A float4 (rows of step s + 1, requested one step ago) is rounded,
  //     written to the other LDS buffer and re-requested for step s + 2.
  bf16x8 af[2][4], bf[2][2];
  auto read_frags = [&](auto SETc, const short* buf, int ks) {
    constexpr int SET = decltype(SETc)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) af[SET][i] = *reinterpret_cast<const bf16x8*>(&buf[a_rd + 32 * i * LDA + 16 * ks]);
#pragma unroll
    for (int j = 0; j < 2; ++j) bf[SET][j] = *reinterpret_cast<const bf16x8*>(&buf[b_rd + (j * 4 + ks) * 512]);
  };
  // (hipcc cannot count its own loads past an LDS-DMA in flight and waits vmcnt(0) at the next use of one: so the staged
  //  rows are consumed in the FIRST half of a step, while no DMA is in flight, and the DMA of step s + 1 and the reloads
  //  for step s + 2 are issued in the second half, in this order -- the barrier's vmcnt(8) counts on it)
  auto reload_one = [&](int i, int s) { raw[i] = buf_load4(rs_a, a_voff[i], (k_lo + (s + 2) * TK) * ES); };
  short* const buf0 = smem;
  short* const buf1 = smem + BUF;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  // (BF_EARLY: the rows of step s + 2 are re-requested right behind the LDS write that frees their registers -- a whole step
  //  ahead of their use.  In the second half of the step, behind the weight DMA, the youngest request was two MFMA slots old
  //  when the compiler's vmcnt(0) in front of the next step's first write waited for it: a memory latency per step.)
  auto group = [&](auto SETc, const short* cur, short* nxt, int s, auto KSc, auto Wc, auto Rc) {
    constexpr int SET = decltype(SETc)::value, ks = decltype(KSc)::value;
    constexpr bool write = decltype(Wc)::value, reload = decltype(Rc)::value;
    using OTHER = std::integral_constant<int, SET ^ 1>;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int i = m >> 1, j = m & 1;
      if (!(GEMMBF_KO & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[SET][i], bf[SET][j], acc[i][j], 0, 0, 0);
      else if (m == 0) acc[i][j][0] += __builtin_bit_cast(float, (int)af[SET][i][0] ^ (int)bf[SET][j][0]);
      if (m == 0 && ks < 3 && !(GEMMBF_KO & 16)) read_frags(OTHER{}, cur, ks + 1);
      if (ks < 2 && (m & 1) && 4 * ks + (m >> 1) < NA) {                                                // groups 0, 1: the staged registers
        if (write && !(GEMMBF_KO & 8)) write_one(4 * ks + (m >> 1), nxt);
        if (BF_EARLY && reload && !(GEMMBF_KO & 1)) reload_one(4 * ks + (m >> 1), s);                   // ... and the rows of step s + 2 into them
      }
      if (ks == 2 && m == 0 && write && !(GEMMBF_KO & 2)) dma_b(s + 1, nxt);                           // group 2: the weight chunks of step s + 1
      if (!BF_EARLY && ks >= 2 && (m & 1) && 4 * (ks - 2) + (m >> 1) < NA && reload && !(GEMMBF_KO & 1)) reload_one(4 * (ks - 2) + (m >> 1), s);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // Every wave's LDS-DMA and LDS writes have landed, then all waves meet (the DMA is ordered only by the issuing wave's
  // vmcnt: the wait comes BEFORE the barrier, the reads after it).  vmcnt counts in issue order: with the A requests issued
  // BEHIND the DMA (!BF_EARLY), "all but the NA youngest" retires the DMA and leaves those loads flying; with them in front
  // of it the wait is for everything (the requests are half a step to a step old by then).
  auto step_barrier = [&](bool a_in_flight) {
    if (!BF_EARLY && a_in_flight && !(GEMMBF_KO & 1)) {
      if constexpr (ABF) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    }
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  using TT = std::true_type;
  using FF = std::false_type;
  auto step = [&](int s, auto Wc, auto Rc) __attribute__((always_inline)) {
    short* cur = (s & 1) ? buf1 : buf0;             // (nxt was last read in step s - 1: every wave is past that barrier)
    short* nxt = (s & 1) ? buf0 : buf1;
    read_frags(I0{}, cur, 0);
    __builtin_amdgcn_sched_barrier(0);
    group(I0{}, cur, nxt, s, I0{}, Wc, Rc);
    group(I1{}, cur, nxt, s, I1{}, Wc, Rc);
    group(I0{}, cur, nxt, s, I2{}, Wc, Rc);
    group(I1{}, cur, nxt, s, I3{}, Wc, Rc);
    step_barrier(decltype(Rc)::value);
  };
  load_a(0);
  dma_b(0, buf0);
  write_a(buf0);
  if (KS > 1) load_a(1);
  step_barrier(KS > 1);
  int s = 0;
#pragma unroll 1
  for (; s + 2 < KS; ++s) step(s, TT{}, TT{});      // write = s + 1 < KS, reload = s + 2 < KS: compile-time in the main loop
  if (s + 1 < KS) { step(s, TT{}, FF{}); ++s; }
  step(s, FF{}, FF{});

  float* Cb = g.c_ptrs[0] ? g.c_ptrs[z & 7] : g.C + (long)z * g.c_sz;
  float bn[2];
  int col[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    col[j] = n0 + wc * 64 + j * 32 + li;
    bn[j] = g.bias_n ? g.bias_n[col[j]] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row >= g.M) continue;
      float* crow = Cb + (long)row * g.c_sm;
#pragma unroll
      for (int j = 0; j < 2; ++j) crow[col[j]] = (acc[i][j][r] + bn[j]) * g.oscale;
    }
}

template <bool ABF>
__global__ __launch_bounds__(512, 2) void gemm_bf_kernel(const BfJobs jobs) {
  extern __shared__ __attribute__((aligned(16))) short bf_smem[];           // 2 x 69,632 B
  if ((int)blockIdx.x < jobs.first1) gemm_bf_body<ABF>(jobs.job[0], (int)blockIdx.x, bf_smem);
  else gemm_bf_body<ABF>(jobs.job[1], (int)blockIdx.x - jobs.first1, bf_smem);
}

// ---- weight gradients in the same mode:  part[z][m][n] = sum_{k in part z} bf16(A[k][m]) * bf16(B[k][n]) ----------------------
// dW_v = dP_v^T V, dW_q = sum_l dP_q,l^T Q_l and the phrase level's dWcat = dZ^T Xcat at config 4's sizes: a 2048 x 2048
// (6144 x 6144) result contracted over thousands of rows.  gemm_tn.hip's 128 x 128 tiles in single-piece mode fetch a byte
// per 32 flops (16 k per barrier, both operands fp32): 0.23 of the bf16 peak.  Here: 256 x 256 tiles on 512 threads, 32 rows
// per barrier, both operands contiguous along their tile index: float4 = 4 consecutive columns of one row k, rounded and
// staged as [k][256 + 32] bf16 images (conflict-free 8-byte writes), fragments through the transposing LDS read
// (ds_read_b64_tr_b16), LDS double-buffered.  Split-K parts run over the CONCATENATED levels (a part may cross from one
// level's rows into the next: K % 32 == 0 per level), so that two rounds of workgroups fill the chip evenly; the partial
// results are added by the same deterministic reduce as gemm_tn's.  SUM3: A is the sum of three arrays (dP_v of the levels).
constexpr int TLD = 256 + 32;                     // bf16 elements per staged k-row
constexpr int T_IMG = 32 * TLD;                   // one operand image: 18,432 B
constexpr int T_BUF = 2 * T_IMG;                  // A and B

struct BfTnArgs {
  const float* A; long a_sl; int a_ld; long a_term;    // (ABF: A holds bf16 elements; a_sl, a_ld, a_term stay in elements)
  const float* B; const float* b_ptrs[8]; long b_sl; int b_ld;
  float* C;
  int M, N, K, levels, spp, P;                    // K rows per level; spp: 32-row steps per part; P parts
  int mask_blk; unsigned tile_mask;               // in units of 128 columns, as TnGemm
};
struct BfTnJobs { BfTnArgs job[2]; int first1; };

// ABF: the A operand is stored as bf16 (the fused backward's dP arrays in the reduced-precision mode): 16-byte loads of 8
// elements, no rounding on the way -- the image and everything after it are the same.
template <bool SUM3, bool ABF>
__device__ __forceinline__ void gemm_bf_tn_body(const BfTnArgs& g, const int id, short* const lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, li = lane & 31, lh = lane >> 5;
  const int ntn = g.N / 256, ntiles = (g.M / 256) * ntn;
  // XCD-aware order (workgroup i runs on XCD i % 8): the (part, tile) pairs in part-major, row-tile-major order are cut
  // into eight runs, one per XCD -- an XCD then works on one part's rows (or few) and on whole row tiles: its L2 serves
  // the A rows (three reads per element under SUM3) to all the column tiles of a row tile, and the B rows of its part to
  // its row tiles, instead of both coming from HBM once per tile
  int z = id / ntiles, t = id % ntiles;
  if ((g.P * ntiles) % 8 == 0) {
    const int lin = (id & 7) * (g.P * ntiles / 8) + (id >> 3);
    z = lin / ntiles; t = lin % ntiles;
  }
  const int mt = t / ntn, nt = t % ntn;
  if (g.mask_blk > 0 && !((g.tile_mask >> (((mt * 2) / g.mask_blk) * 3 + (nt * 2) / g.mask_blk)) & 1u)) return;
  const int m0 = mt * 256, n0 = nt * 256;
  const int spl = g.K / 32, total = spl * g.levels;                  // steps per level, steps in all
  const int g0 = z * g.spp, g1 = min(total, g0 + g.spp), steps = g1 - g0;
  if (steps <= 0) return;
  // staging: per operand and step 4 float4 per thread; a wave's load covers one k-row x 1 KB
  const int sk = tid >> 6, sm = (tid & 63) * 4;
  const int st_off = sk * TLD + sm;                                   // + 8 i rows
  const int tr_off = (8 * lh + ((lane & 15) >> 2)) * TLD + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int a_rd = tr_off + wr * 128, b_rd = T_IMG + tr_off + wc * 64;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  f32x4 ra[ABF ? 1 : 4], rb[4], rt[SUM3 && !ABF ? 8 : 1];
  u32x4 ha[ABF ? 2 : 1], ht[SUM3 && ABF ? 4 : 1];                    // ABF: two 16-byte loads of 8 bf16 per thread and step
  const int hk = tid >> 5, hm = (tid & 31) * 8;                       // ABF staging: a half wave covers one k-row x 512 B
  auto load = [&](int gs) {                                           // global step gs -> (level, row)
    const int lvl = gs / spl, k0 = (gs - lvl * spl) * 32;
    const float* Bb = g.b_ptrs[0] ? g.b_ptrs[lvl & 7] : g.B + (long)lvl * g.b_sl;
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(Bb, (unsigned)((long)g.K * g.b_ld * 4));
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = buf_load4(rs_b, ((sk + 8 * i) * g.b_ld + n0 + sm) * 4, k0 * g.b_ld * 4);
    if constexpr (ABF) {
      const unsigned short* Ab = reinterpret_cast<const unsigned short*>(g.A) + (long)lvl * g.a_sl;
      const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(Ab, (unsigned)((long)g.K * g.a_ld * 2));
#pragma unroll
      for (int i = 0; i < 2; ++i)
        ha[i] = __builtin_bit_cast(u32x4, buf_load4(rs_a, ((hk + 16 * i) * g.a_ld + m0 + hm) * 2, k0 * g.a_ld * 2));
      if constexpr (SUM3) {
        const __amdgpu_buffer_rsrc_t rs_1 = make_rsrc(Ab + g.a_term, (unsigned)((long)g.K * g.a_ld * 2));
        const __amdgpu_buffer_rsrc_t rs_2 = make_rsrc(Ab + 2 * g.a_term, (unsigned)((long)g.K * g.a_ld * 2));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ht[i] = __builtin_bit_cast(u32x4, buf_load4(rs_1, ((hk + 16 * i) * g.a_ld + m0 + hm) * 2, k0 * g.a_ld * 2));
          ht[2 + i] = __builtin_bit_cast(u32x4, buf_load4(rs_2, ((hk + 16 * i) * g.a_ld + m0 + hm) * 2, k0 * g.a_ld * 2));
        }
      }
    } else {
      const float* Ab = g.A + (long)lvl * g.a_sl;
      const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(Ab, (unsigned)((long)g.K * g.a_ld * 4));
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i] = buf_load4(rs_a, ((sk + 8 * i) * g.a_ld + m0 + sm) * 4, k0 * g.a_ld * 4);
      if constexpr (SUM3) {
        const __amdgpu_buffer_rsrc_t rs_1 = make_rsrc(Ab + g.a_term, (unsigned)((long)g.K * g.a_ld * 4));
        const __amdgpu_buffer_rsrc_t rs_2 = make_rsrc(Ab + 2 * g.a_term, (unsigned)((long)g.K * g.a_ld * 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          rt[i] = buf_load4(rs_1, ((sk + 8 * i) * g.a_ld + m0 + sm) * 4, k0 * g.a_ld * 4);
          rt[4 + i] = buf_load4(rs_2, ((sk + 8 * i) * g.a_ld + m0 + sm) * 4, k0 * g.a_ld * 4);
        }
      }
    }
  };
  auto write = [&](short* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x2 vb = {cvt_pk_bf16(rb[i][0], rb[i][1]), cvt_pk_bf16(rb[i][2], rb[i][3])};
      *reinterpret_cast<u32x2*>(&buf[T_IMG + 8 * i * TLD + st_off]) = vb;
    }
    if constexpr (ABF) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        u32x4 v = ha[i];
        if constexpr (SUM3) {                                         // (level order 0 + 1 + 2 in fp32, rounded once)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            auto lo = [](unsigned x) { return __builtin_bit_cast(float, x << 16); };
            auto hi = [](unsigned x) { return __builtin_bit_cast(float, x & 0xffff0000u); };
            v[e] = cvt_pk_bf16((lo(ha[i][e]) + lo(ht[i][e])) + lo(ht[2 + i][e]), (hi(ha[i][e]) + hi(ht[i][e])) + hi(ht[2 + i][e]));
          }
        }
        *reinterpret_cast<u32x4*>(&buf[(hk + 16 * i) * TLD + hm]) = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 a = ra[i];
        if constexpr (SUM3) a = (a + rt[i]) + rt[4 + i];              // (level order 0 + 1 + 2, as the separate summing pass)
        const u32x2 va = {cvt_pk_bf16(a[0], a[1]), cvt_pk_bf16(a[2], a[3])};
        *reinterpret_cast<u32x2*>(&buf[8 * i * TLD + st_off]) = va;
      }
    }
  };
  auto frag = [&](const short* p) {
    const bf16x4 lo = lds_tr16(p), hi = lds_tr16(p + 4 * TLD);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  auto compute = [&](const short* buf) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bf[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = frag(buf + a_rd + 32 * i + 16 * ks * TLD);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = frag(buf + b_rd + 32 * j + 16 * ks * TLD);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };
  short* const buf0 = lds;
  short* const buf1 = lds + T_BUF;
  load(g0);
  write(buf0);
  if (steps > 1) load(g0 + 1);
  lds_barrier();
  for (int s = 0; s < steps; ++s) {
    short* cur = (s & 1) ? buf1 : buf0;
    short* nxt = (s & 1) ? buf0 : buf1;
    compute(cur);
    if (s + 1 < steps) write(nxt);                                    // rows of step s + 1, requested one step ago
    if (s + 2 < steps) load(g0 + s + 2);
    lds_barrier();
  }
  float* Cb = g.C + (long)z * g.M * g.N;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float* crow = Cb + (long)row * g.N + n0 + wc * 64 + li;
#pragma unroll
      for (int j = 0; j < 2; ++j) crow[j * 32] = acc[i][j][r];
    }
}

template <bool SUM3, bool ABF>
__global__ __launch_bounds__(512, 2) void gemm_bf_tn_kernel(const BfTnJobs jobs) {
  extern __shared__ __attribute__((aligned(16))) short bf_tn_smem[];        // 2 x 36,864 B
  if ((int)blockIdx.x < jobs.first1) gemm_bf_tn_body<SUM3, ABF>(jobs.job[0], (int)blockIdx.x, bf_tn_smem);
  else gemm_bf_tn_body<false, ABF>(jobs.job[1], (int)blockIdx.x - jobs.first1, bf_tn_smem);
}

}  // namespace

int gemm_bf_enabled();
int gemm_bf_tn_supported(const TnGemm& d) {
  auto pal = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  bool ok = gemm_bf_enabled() && d.bf16 && !d.b_kdiv && d.M >= 256 && d.N >= 256 && (d.M % 256) == 0 && (d.N % 256) == 0 && d.K >= 32 &&
            (d.K % 32) == 0 && d.levels >= 1 && d.levels <= 8 && (d.a_ld & 3) == 0 && (d.b_ld & 3) == 0 && (d.a_term & 3) == 0 &&
            (d.a_sl & 3) == 0 && (d.b_sl & 3) == 0 && pal(d.A) && (d.b_ptrs[0] ? true : pal(d.B)) &&
            (long)d.K * d.a_ld * 4 < 0x40000000L && (long)d.K * d.b_ld * 4 < 0x40000000L && (d.mask_blk == 0 || (d.mask_blk % 2) == 0);
  for (int t = 0; t < 8; ++t) ok = ok && pal(d.b_ptrs[t]);
  if (d.a_bf16) ok = ok && (d.a_ld & 7) == 0 && (d.a_term & 7) == 0 && (d.a_sl & 7) == 0;   // 16-byte loads of 8 elements
  return ok ? 1 : 0;
}

// parts for a job: about `want` of them, each a whole number of 32-row steps over the concatenated levels
int gemm_bf_tn_plan(const TnGemm& d, int want, int* spp) {
  const int total = (d.K / 32) * d.levels;
  if (want < 1) want = 1;
  if (want > total) want = total;
  *spp = (total + want - 1) / want;
  return (total + *spp - 1) / *spp;
}

// n = 1 or 2 jobs; spp[i] / parts[i] from gemm_bf_tn_plan; d[i].C = the job's partial buffer [parts][M][N]
int launch_gemm_bf_tn(const TnGemm* d, const int* spp, const int* parts, int n, hipStream_t s) {
  CA_CHECK_ARG(n == 1 || n == 2, "gemm_bf_tn: 1 or 2 jobs per launch");
  BfTnJobs jobs = {};
  long nb[2] = {0, 0};
  for (int i = 0; i < n; ++i) {
    CA_CHECK_ARG(gemm_bf_tn_supported(d[i]) && d[i].A && (d[i].B || d[i].b_ptrs[0]) && d[i].C && spp[i] > 0 && parts[i] > 0,
                 "gemm_bf_tn: unsupported job M=%d N=%d K=%d", d[i].M, d[i].N, d[i].K);
    BfTnArgs& g = jobs.job[i];
    g = BfTnArgs{};
    g.A = d[i].A; g.a_sl = d[i].a_sl; g.a_ld = d[i].a_ld; g.a_term = d[i].a_term;
    g.B = d[i].B; g.b_sl = d[i].b_sl; g.b_ld = d[i].b_ld;
    for (int t = 0; t < 8; ++t) g.b_ptrs[t] = d[i].b_ptrs[t];
    g.C = d[i].C; g.M = d[i].M; g.N = d[i].N; g.K = d[i].K; g.levels = d[i].levels; g.spp = spp[i]; g.P = parts[i];
    g.mask_blk = d[i].mask_blk; g.tile_mask = d[i].tile_mask;
    nb[i] = (long)(d[i].M / 256) * (d[i].N / 256) * parts[i];
  }
  CA_CHECK_ARG(n == 1 || d[1].a_term == 0, "gemm_bf_tn: only the first job may sum three A terms");
  CA_CHECK_ARG(nb[0] + nb[1] < 2147483647L, "gemm_bf_tn: grid too large");
  jobs.first1 = (int)nb[0];
  const size_t lds = (size_t)2 * T_BUF * sizeof(short);
  static DeviceOnce once;
  CA_TRY(once.run([&] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf_tn_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf_tn_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf_tn_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf_tn_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return e;
  }, "gemm_bf_tn"));
  const dim3 grid((unsigned)(nb[0] + nb[1]));
  const bool abf = d[0].a_bf16 != 0;
  CA_CHECK_ARG(n == 1 || (d[1].a_bf16 != 0) == abf, "gemm_bf_tn: the jobs of a launch store A alike");
  if (d[0].a_term && abf) hipLaunchKernelGGL((gemm_bf_tn_kernel<true, true>), grid, dim3(512), lds, s, jobs);
  else if (d[0].a_term) hipLaunchKernelGGL((gemm_bf_tn_kernel<true, false>), grid, dim3(512), lds, s, jobs);
  else if (abf) hipLaunchKernelGGL((gemm_bf_tn_kernel<false, true>), grid, dim3(512), lds, s, jobs);
  else hipLaunchKernelGGL((gemm_bf_tn_kernel<false, false>), grid, dim3(512), lds, s, jobs);
  CA_CHECK_LAUNCH("gemm_bf_tn");
  return 0;
}

namespace {
}  // namespace

// COATTN_GEMM_BF=0 (developer switch): gemm_w's single-piece mode instead
int gemm_bf_enabled() {
  static const int on = dev_env_int("COATTN_GEMM_BF", 1);
  return on;
}

int gemm_bf_supported(const WGemm& d) {
  auto pal = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  bool ok = gemm_bf_enabled() && d.bf16 && !d.a_sk && d.M >= 256 && d.N >= TN && (d.N % TN) == 0 && d.K >= TK && (d.K % TK) == 0 &&
            (d.a_sm & 3) == 0 && (d.a_sz & 3) == 0 && d.batch >= 1 && d.batch <= 8 && (d.a_ptrs[0] ? true : pal(d.A)) &&
            ((long)d.M * d.a_sm + d.K) * 4 < 0x40000000L && wsplit_bytes(d.N, d.K) / 3 < 0x40000000UL;
  for (int t = 0; t < 8; ++t) ok = ok && pal(d.a_ptrs[t]);
  if (d.a_bf16) ok = ok && (d.a_sm & 7) == 0 && (d.a_sz & 7) == 0;      // 16-byte loads of 8 elements
  if (d.kband_n > 0) {
    ok = ok && (d.kband_n % TN) == 0 && (d.N + d.kband_n - 1) / d.kband_n <= 3;
    for (int t = 0; t < 3 && ok; ++t)
      ok = d.kband_lo[t] >= 0 && d.kband_hi[t] <= d.K && d.kband_lo[t] < d.kband_hi[t] && (d.kband_lo[t] % TK) == 0 && (d.kband_hi[t] % TK) == 0;
  }
  return ok ? 1 : 0;
}

int launch_gemm_bf(const WGemm* d, int n, hipStream_t s) {
  CA_CHECK_ARG(n == 1 || n == 2, "gemm_bf: 1 or 2 jobs per launch");
  BfJobs jobs = {};
  long nb[2] = {0, 0};
  for (int i = 0; i < n; ++i) {
    CA_CHECK_ARG(gemm_bf_supported(d[i]), "gemm_bf: unsupported shape M=%d N=%d K=%d", d[i].M, d[i].N, d[i].K);
    CA_CHECK_ARG((d[i].A || d[i].a_ptrs[0]) && d[i].Wf && (d[i].C || d[i].c_ptrs[0]), "gemm_bf: null operand");
    gw::WArgs& g = jobs.job[i];
    g = gw::WArgs{};
    g.A = d[i].A; g.a_sz = d[i].a_sz; g.a_sm = d[i].a_sm;
    g.kband_n = d[i].kband_n;
    for (int t = 0; t < 3; ++t) { g.kband_lo[t] = d[i].kband_lo[t]; g.kband_hi[t] = d[i].kband_hi[t]; }
    g.Wf = d[i].Wf; g.wf_bytes = (unsigned)(wsplit_bytes(d[i].N, d[i].K) / 3);
    g.C = d[i].C; g.c_sz = d[i].c_sz; g.c_sm = d[i].c_sm;
    for (int t = 0; t < 8; ++t) { g.a_ptrs[t] = d[i].a_ptrs[t]; g.c_ptrs[t] = d[i].c_ptrs[t]; }
    g.bias_n = d[i].bias_n; g.oscale = d[i].out_scale != 0.f ? d[i].out_scale : 1.f;
    g.M = d[i].M; g.N = d[i].N; g.K = d[i].K;
    const long ntm = (d[i].M + TM - 1) / TM, ntn = d[i].N / TN;
    nb[i] = (long)d[i].batch * ntn * ((ntm + 7) / 8) * 8;
  }
  CA_CHECK_ARG(nb[0] + nb[1] < 2147483647L, "gemm_bf: grid too large");
  jobs.first1 = (int)nb[0];
  const size_t lds = (size_t)2 * BUF * sizeof(short);
  static DeviceOnce once;
  CA_TRY(once.run([&] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return e;
  }, "gemm_bf"));
  CA_CHECK_ARG(n == 1 || (d[0].a_bf16 != 0) == (d[1].a_bf16 != 0), "gemm_bf: the jobs of a launch store A alike");
  if (d[0].a_bf16) hipLaunchKernelGGL(gemm_bf_kernel<true>, dim3((unsigned)(nb[0] + nb[1])), dim3(512), lds, s, jobs);
  else hipLaunchKernelGGL(gemm_bf_kernel<false>, dim3((unsigned)(nb[0] + nb[1])), dim3(512), lds, s, jobs);
  CA_CHECK_LAUNCH("gemm_bf");
  return 0;
}
