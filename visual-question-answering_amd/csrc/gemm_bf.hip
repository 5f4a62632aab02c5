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

__device__ __forceinline__ void gemm_bf_body(const gw::WArgs& g, const int id, short* const smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, li = lane & 31, lh = lane >> 5;
  // XCD-aware order: the column tiles of a row tile run on one XCD, one after the other (they re-read the same A rows)
  const int ntm = (g.M + TM - 1) / TM, ntn = g.N / TN;
  const int x = id & 7, slot = id >> 3, per = ntn * ((ntm + 7) / 8);
  const int z = slot / per, t = slot % per, mt = (t / ntn) * 8 + x;
  if (mt >= ntm) return;
  const int m0 = mt * TM, n0 = (g.kband_n > 0 ? ntn - 1 - t % ntn : t % ntn) * TN;
  const float* Ab = g.a_ptrs[0] ? g.a_ptrs[z & 7] : g.A + (long)z * g.a_sz;
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(Ab, (unsigned)(((long)(g.M - 1) * g.a_sm + g.K) * 4));
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(g.Wf, g.wf_bytes);
  int k_lo = 0, k_hi = g.K;
  if (g.kband_n > 0) {
    const int band = n0 / g.kband_n;
    k_lo = g.kband_lo[band]; k_hi = g.kband_hi[band];
  }
  const int KS = (k_hi - k_lo) / TK;              // (host check: multiples of 64)

  // A staging: 8 float4 per thread and step; a wave's load covers 4 rows x 256 B
  const int a_row = tid >> 4, a_k = (tid & 15) * 4;
  int a_voff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a_voff[i] = (m0 + a_row + 32 * i) < g.M ? ((m0 + a_row + 32 * i) * g.a_sm + a_k) * 4 : 0x40000000;
  const int a_wr = a_row * LDA + a_k;             // + 32 i rows
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

  f32x4 raw[8];
  auto load_a = [&](int s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) raw[i] = buf_load4(rs_a, a_voff[i], (k_lo + s * TK) * 4);
  };
  auto dma_b = [&](int s, short* buf) {
    const int c0 = b_src + ((k_lo + s * TK) / 16) * kChunk;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(buf + b_dst + ks * 512), 16, lane * 16, c0 + ks * kChunk, 0, 0);
  };
  auto write_a = [&](short* buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const u32x2 v = {cvt_pk_bf16(raw[i][0], raw[i][1]), cvt_pk_bf16(raw[i][2], raw[i][3])};
      *reinterpret_cast<u32x2*>(&buf[32 * i * LDA + a_wr]) = v;
    }
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
  auto write_one = [&](int i, short* nxt) {
    const u32x2 v = {cvt_pk_bf16(raw[i][0], raw[i][1]), cvt_pk_bf16(raw[i][2], raw[i][3])};
    *reinterpret_cast<u32x2*>(&nxt[32 * i * LDA + a_wr]) = v;
  };
  auto reload_one = [&](int i, int s) { raw[i] = buf_load4(rs_a, a_voff[i], (k_lo + (s + 2) * TK) * 4); };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto group = [&](auto SETc, const short* cur, short* nxt, int s, int ks, bool write, bool reload) {
    constexpr int SET = decltype(SETc)::value;
    using OTHER = std::integral_constant<int, SET ^ 1>;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int i = m >> 1, j = m & 1;
      if (!(GEMMBF_KO & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[SET][i], bf[SET][j], acc[i][j], 0, 0, 0);
      else if (m == 0) acc[i][j][0] += __builtin_bit_cast(float, (int)af[SET][i][0] ^ (int)bf[SET][j][0]);
      if (m == 0 && ks < 3 && !(GEMMBF_KO & 16)) read_frags(OTHER{}, cur, ks + 1);
      if (ks < 2 && (m & 1) && write && !(GEMMBF_KO & 8)) write_one(4 * ks + (m >> 1), nxt);           // groups 0, 1: the eight staged float4
      if (ks == 2 && m == 0 && write && !(GEMMBF_KO & 2)) dma_b(s + 1, nxt);                           // group 2: the weight chunks of step s + 1
      if (ks >= 2 && (m & 1) && reload && !(GEMMBF_KO & 1)) reload_one(4 * (ks - 2) + (m >> 1), s);    // groups 2, 3: rows of step s + 2
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // Every wave's LDS-DMA and LDS writes have landed, then all waves meet (the DMA is ordered only by the issuing wave's
  // vmcnt: the wait comes BEFORE the barrier, the reads after it).  vmcnt counts in issue order, so with the next-but-one
  // step's eight A loads issued BEHIND the DMA, "all but the 8 youngest" retires the DMA and leaves those loads flying.
  auto step_barrier = [&](bool a_in_flight) {
    if (a_in_flight && !(GEMMBF_KO & 1)) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  short* const buf0 = smem;
  short* const buf1 = smem + BUF;
  load_a(0);
  dma_b(0, buf0);
  write_a(buf0);
  if (KS > 1) load_a(1);
  step_barrier(KS > 1);
  for (int s = 0; s < KS; ++s) {
    short* cur = (s & 1) ? buf1 : buf0;
    short* nxt = (s & 1) ? buf0 : buf1;
    const bool write = s + 1 < KS, reload = s + 2 < KS;   // (nxt was last read in step s - 1: every wave is past that barrier)
    read_frags(I0{}, cur, 0);
    __builtin_amdgcn_sched_barrier(0);
    group(I0{}, cur, nxt, s, 0, write, reload);
    group(I1{}, cur, nxt, s, 1, write, reload);
    group(I0{}, cur, nxt, s, 2, write, reload);
    group(I1{}, cur, nxt, s, 3, write, reload);
    step_barrier(reload);
  }

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

__global__ __launch_bounds__(512, 2) void gemm_bf_kernel(const BfJobs jobs) {
  extern __shared__ __attribute__((aligned(16))) short bf_smem[];           // 2 x 69,632 B
  if ((int)blockIdx.x < jobs.first1) gemm_bf_body(jobs.job[0], (int)blockIdx.x, bf_smem);
  else gemm_bf_body(jobs.job[1], (int)blockIdx.x - jobs.first1, bf_smem);
}

}  // namespace

// COATTN_GEMM_BF=0 (developer switch): gemm_w's single-piece mode instead
int gemm_bf_enabled() {
  static const int on = [] { const char* e = getenv("COATTN_GEMM_BF"); return e ? atoi(e) : 1; }();
  return on;
}

int gemm_bf_supported(const WGemm& d) {
  auto pal = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  bool ok = gemm_bf_enabled() && d.bf16 && !d.a_sk && d.M >= 256 && d.N >= TN && (d.N % TN) == 0 && d.K >= TK && (d.K % TK) == 0 &&
            (d.a_sm & 3) == 0 && (d.a_sz & 3) == 0 && d.batch >= 1 && d.batch <= 8 && (d.a_ptrs[0] ? true : pal(d.A)) &&
            ((long)d.M * d.a_sm + d.K) * 4 < 0x40000000L && wsplit_bytes(d.N, d.K) / 3 < 0x40000000UL;
  for (int t = 0; t < 8; ++t) ok = ok && pal(d.a_ptrs[t]);
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
  CA_TRY(once.run([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); },
                  "gemm_bf"));
  hipLaunchKernelGGL(gemm_bf_kernel, dim3((unsigned)(nb[0] + nb[1])), dim3(512), lds, s, jobs);
  CA_CHECK_LAUNCH("gemm_bf");
  return 0;
}
