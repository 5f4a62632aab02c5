// fp32-accurate GEMM against a WEIGHT operand, on v_mfma_f32_32x32x16_bf16 (gfx950).
//
//   C[z][m][n] = (sum_k A[z][m][k] * Bw(k, n) + bias_n[n]) * out_scale          A rows contiguous along k
//
// The projections of the co-attention path multiply activations by a d x d nn.Linear weight that is the same
// for every row tile: P_v = V W_v^T + b_v, P_q = Q W_q^T + b_q (model.py:380-384), dQ = dP_q W_q in the
// backward.  gemm.hip splits BOTH operands into their three bf16 pieces while staging them (11 VALU
// instructions per pair of elements) in every workgroup; the weight's share of that is the same arithmetic
// repeated by each of the M / 128 row tiles.  Here the weight is split ONCE per call by wsplit_kernel into a
// fragment-ordered image -- for every (32-column tile, 16-k step, piece) the 1 KB that the 64 lanes of a wave
// hold as the MFMA's B operand, lane-major -- and the GEMM loads those fragments straight from global memory
// (L2 resident: 1.5 MB for 512 x 512) into registers with one coalesced 16-byte load per lane: no LDS traffic,
// no VALU work and no address arithmetic for B.  A goes through LDS as in gemm.hip (split while staging,
// [row][k] bf16 images, one ds_read_b128 per fragment), double-buffered so that a K step has one barrier.
//
// Tile 128 x 128 per 256-thread workgroup, 4 waves as 2 x 2, each wave 2 x 2 MFMA tiles; BK = 32.
// Developer switches (tools/ab_gemmw.sh, never in the shipped library): -DGEMMW_NOB / _NOSPLIT / _NOMFMA compile the
// B-fragment loads / the split arithmetic / the MFMAs out (wrong results) to price them.
// Numerics: the six partial products of gemm.hip's split mode in the same order (exact operand pieces,
// fp32 accumulation), so the two kernels agree to accumulation order.
#include "common.h"
#include "fused.h"
#include "gemm_w_body.h"
#include <type_traits>
#include <stdlib.h>

namespace {

using namespace gw;

struct WJobs { WArgs job[2]; int first1; RowFlagJob rows; int nflag; };   // blocks [0, nflag): the row-flag job; then [0, first1) of the rest work on job 0, the others on job 1

struct SplitJob { const float* W; void* out; int N, K, trans, ld, pieces; float* amax; };
struct SplitArgs { SplitJob job[3]; int njobs; float* status_hdr; float f16; RowFlagJob rows; int row_blocks, cb; unsigned* zero8; };   // grid: [row_blocks][njobs x cb]

// The RowFlagJob's workgroups (the FIRST row_blocks of the launch: they have the most to wait for): workgroup x takes 32 rows of one batch entry, a wave 8 of them, all
// requested before the first test (one memory latency per wave, not per row); a row is read in whole 1 KB segments (a lane
// per 16 bytes, KC = K / 256 of them), any non-zero (or NaN) element sets its bit; an all-zero row gets its output written
// here (16-byte stores; a live row's store is sent outside the buffer instead of branched around).
template <int KC>
__device__ __forceinline__ void rowflag_block(const RowFlagJob& j, const int bx) {
  __shared__ unsigned wbits[4];
  const int words = (j.M + 31) / 32;
  if (bx >= j.batch * words) return;
  const int z = bx / words, wd = bx % words;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* A = j.a_ptrs[z & 7];
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(A, (unsigned)(((long)(j.M - 1) * j.a_sm + j.K) * 4));   // rows past M read 0
  // the output rows of this batch entry as a buffer (rows past M: stores dropped); N == K here (host check): the same lane
  // pattern writes a row that reads one
  const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(j.C + (long)z * j.c_sz, (unsigned)(((long)(j.M - 1) * j.c_sm + j.N) * 4));
  const int r0 = 32 * wd + 8 * wave;
  f32x4 x[8][KC], fill[KC];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int k = 0; k < KC; ++k) x[i][k] = buf_load4(rs, ((r0 + i) * j.a_sm + 4 * lane + 256 * k) * 4, 0);
#pragma unroll
  for (int k = 0; k < KC; ++k) {                     // the dense product's value of an all-zero row: (acc = +0) + bias, scaled
    const f32x4 b = *reinterpret_cast<const f32x4*>(j.bias_n + 4 * lane + 256 * k);
    fill[k] = f32x4{(0.f + b[0]) * j.out_scale, (0.f + b[1]) * j.out_scale, (0.f + b[2]) * j.out_scale, (0.f + b[3]) * j.out_scale};
  }
  unsigned bits = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    bool nz = false;
#pragma unroll
    for (int k = 0; k < KC; ++k) nz = nz || !(x[i][k][0] == 0.f && x[i][k][1] == 0.f && x[i][k][2] == 0.f && x[i][k][3] == 0.f);
    const bool live = __builtin_amdgcn_ballot_w64(nz) != 0;
    if (live) bits |= 1u << (8 * wave + i);
    // (branch-free: a live row's store goes to an offset outside the buffer)
#pragma unroll
    for (int k = 0; k < KC; ++k) buf_store4(fill[k], rs_c, live ? 0x40000000 : ((r0 + i) * j.c_sm + 4 * lane + 256 * k) * 4, 0);
  }
  if (lane == 0) wbits[wave] = bits;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(&j.rowbits[(long)z * words + wd], wbits[0] | wbits[1] | wbits[2] | wbits[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (j.rowcnt) {                                  // (the job rides in a GEMM launch: tell its tiles, behind the word)
      // The word went out as an agent-scope (write-through) store: once it is acknowledged it is visible device-wide, so the
      // order is a wait for this thread's stores, NOT a device fence -- __threadfence() writes the XCD's L2 back and the
      // readers' acquire invalidated theirs: 390 + 396 of those under the running GEMM tiles took the launch from 52 to 100 us
      __builtin_amdgcn_s_waitcnt(0);
      atomicAdd(&j.rowcnt[z & 7], 1u);
    }
  }
}
template <typename = void>
__device__ __forceinline__ void rowflag_dispatch(const RowFlagJob& j, const int bid) {
  switch (j.K >> 8) {                                // K / 256 in 1 .. 4 (host check)
    case 1: rowflag_block<1>(j, bid); break;
    case 2: rowflag_block<2>(j, bid); break;
    case 3: rowflag_block<3>(j, bid); break;
    default: rowflag_block<4>(j, bid); break;
  }
}

// One wave per (32-column tile nt, 16-k step ks): lane (li = lane & 31, lh = lane >> 5) holds
// Bw(k = 16 ks + 8 lh + e, n = 32 nt + li), e = 0..7 -- the B operand layout of v_mfma_f32_32x32x16_bf16.
// trans = 0: Bw(k, n) = W[n][k] (y = x W^T); trans = 1: Bw(k, n) = W[k][n] (dx = dy W).
__global__ __launch_bounds__(256) void wsplit_kernel(const SplitArgs a) {
  int bid = (int)blockIdx.x;
  if (bid < a.row_blocks) {
    rowflag_dispatch(a.rows, bid);
    return;
  }
  bid -= a.row_blocks;
  if (a.zero8 && bid == 0 && threadIdx.x < 8) a.zero8[threadIdx.x] = 0u;
  const int by = bid / a.cb, bx = bid - by * a.cb;
  const SplitJob j = a.job[by];
  f16_saturating_conversions();                      // (only the pieces = 16 jobs convert to fp16)
  // header of the call's status words (fused.h kStatusHdr): the projection launch behind this one raises [0]
  if (a.status_hdr && bx == 0 && by == 0 && threadIdx.x == 0) { a.status_hdr[0] = 0.f; a.status_hdr[1] = a.f16; }
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int ks16 = (j.K + 15) / 16, nt32 = (j.N + 31) / 32;
  const int chunk = bx * 4 + (threadIdx.x >> 6);
  if (chunk >= ks16 * nt32) return;
  const int nt = chunk / ks16, ks = chunk % ks16;
  const int n = 32 * nt + li, k0 = 16 * ks + 8 * lh;
  f32x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e;
    v[e] = (n < j.N && k < j.K) ? (j.trans ? j.W[(size_t)k * j.ld + n] : j.W[(size_t)n * j.ld + k]) : 0.f;
  }
  if (j.pieces == 16) {                              // two FP16 pieces of kF16WScale * W (fused.h), in the slots of pieces 0 and 1
    u32x4 h, m;
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned hh, mm;
      const float w0 = v[2 * e] * kF16WScale, w1 = v[2 * e + 1] * kF16WScale;
      amax = fmaxf(amax, fmaxf(fabsf(w0), fabsf(w1)));
      split_pair_h(w0, w1, hh, mm);
      h[e] = hh; m[e] = mm;
    }
    if (j.amax) {                                    // range report (coattn_status): one word per wave, no atomics, no zeroing
      amax = wave_max(amax);
      if (lane == 0) j.amax[chunk] = amax;
    }
    char* out = (char*)j.out + (size_t)chunk * kChunkBytes + lane * 16;
    *reinterpret_cast<u32x4*>(out) = h;
    *reinterpret_cast<u32x4*>(out + kFragBytes) = m;
    return;
  }
  bf16x8 p[3];
  split3(v, p);
  if (j.pieces == 1) {                               // hi piece only (the value rounded to bf16), 1 KB chunks: gemm_bf.hip
    *reinterpret_cast<bf16x8*>((char*)j.out + (size_t)chunk * kFragBytes + lane * 16) = p[0];
    return;
  }
  char* out = (char*)j.out + (size_t)chunk * kChunkBytes + lane * 16;
#pragma unroll
  for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8*>(out + q * kFragBytes) = p[q];
}

// Up to two independent GEMMs per launch (P_v and P_q of the forward): the second one's workgroups fill the slots
// the first one's last, partial round of workgroups would leave idle.  AM0: layout of job 0's A operand; NP / NP1: the
// widths of job 0 / job 1 (the forward runs P_v on three pieces and P_q on two in one launch).
#ifndef GEMMW_OCC2           // workgroups per CU of the two-piece four-wave kernels (gemm_w_body.h: two-set B ring)
#define GEMMW_OCC2 3
#endif
template <bool AM0, int NP, int NW = 4, int NP1 = NP, bool H = false>
__global__ __launch_bounds__(64 * NW, (NP == 2 && NP1 == 2 && NW == 4) ? GEMMW_OCC2 : 2) void gemm_w_kernel(const WJobs jobs) {
  __shared__ __attribute__((aligned(16))) short smem[2 * (NP > NP1 ? NP : NP1) * BM * LDR + 2 * BM];  // three pieces: 61,440 B (+ 512 B: the row map of a compacted job); two workgroups of 256 threads per CU
  static_assert(BM * LDR == BK * LDT, "both image layouts have the same size");
  if (H) f16_saturating_conversions();
  int* const rowmap = reinterpret_cast<int*>(smem + 2 * (NP > NP1 ? NP : NP1) * BM * LDR);
  int bid = (int)blockIdx.x;
  if constexpr (NW == 4 && NP == 3 && NP1 == 3 && !H) {  // (the exact four-wave launch: the only one a row-flag job rides in)
    if (bid < jobs.nflag) {
      rowflag_dispatch(jobs.rows, bid);
      return;
    }
    bid -= jobs.nflag;
  }
  if (bid < jobs.first1) gemm_w_body<AM0, NP, NW, H>(jobs.job[0], bid, smem, rowmap);
  else gemm_w_body<false, NP1, NW, H>(jobs.job[1], bid - jobs.first1, smem, rowmap);
}

}  // namespace

size_t wsplit_bytes(int N, int K) { return (size_t)((N + 31) / 32) * ((K + 15) / 16) * kChunkBytes; }

static int check_rowflag_job(const RowFlagJob* rows) {
  CA_CHECK_ARG(rows->rowbits && rows->C && rows->bias_n && rows->M > 0 && rows->batch >= 1 && rows->batch <= 8 && rows->K % 256 == 0 &&
               rows->K <= 1024 && rows->N == rows->K && (rows->c_sm & 3) == 0 && ((long)(rows->M - 1) * rows->c_sm + rows->N) * 4 < 0x40000000L &&
               ((((uintptr_t)rows->C) | ((uintptr_t)rows->bias_n)) & 15) == 0 && ((rows->c_sz & 3) == 0) &&
               (rows->a_sm & 3) == 0 && ((long)(rows->M - 1) * rows->a_sm + rows->K) * 4 < 0x40000000L,
               "row-flag job: bad arguments");
  for (int t = 0; t < rows->batch; ++t) CA_CHECK_ARG((((uintptr_t)rows->a_ptrs[t]) & 15) == 0, "row-flag job: unaligned rows");
  return 0;
}

int launch_wsplit(const WSplit* jobs, int njobs, hipStream_t s, float* status_hdr, int f16, const RowFlagJob* rows, unsigned* zero8) {
  CA_CHECK_ARG(njobs >= 1 && njobs <= 3, "wsplit: 1 to 3 jobs per launch");
  SplitArgs a = {};
  int row_blocks = 0;
  if (rows) {
    CA_TRY(check_rowflag_job(rows));
    CA_CHECK_ARG(rows->rowcnt == nullptr, "wsplit: a row-flag job of this launch needs no counter");
    a.rows = *rows;
    row_blocks = rows->batch * ((rows->M + 31) / 32);
  }
  a.njobs = njobs;
  a.status_hdr = status_hdr; a.f16 = f16 ? 1.f : 0.f;
  int chunks = 0;
  for (int i = 0; i < njobs; ++i) {
    CA_CHECK_ARG(jobs[i].W && jobs[i].out && jobs[i].N > 0 && jobs[i].K > 0, "wsplit: bad job");
    a.job[i] = SplitJob{jobs[i].W, jobs[i].out, jobs[i].N, jobs[i].K, jobs[i].trans, jobs[i].ld,
                        jobs[i].pieces == 1 ? 1 : (jobs[i].pieces == 16 ? 16 : 3), jobs[i].pieces == 16 ? jobs[i].amax : nullptr};
    const int c = ((jobs[i].N + 31) / 32) * ((jobs[i].K + 15) / 16);
    chunks = c > chunks ? c : chunks;
  }
  a.row_blocks = row_blocks; a.cb = (chunks + 3) / 4; a.zero8 = zero8;
  hipLaunchKernelGGL(wsplit_kernel, dim3(row_blocks + njobs * a.cb), dim3(256), 0, s, a);
  CA_CHECK_LAUNCH("wsplit");
  return 0;
}

int gemm_w_supported(const WGemm& d) {
  auto pal = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  // (M >= 128: smaller products are better off on gemm.hip -- except the two-FP16-piece mode, whose range report lives in this
  //  kernel: there any M runs here, a lone partial tile, so that the tolerance mode covers small batches too)
  bool ok = !d.a_bf16 && d.M >= ((d.f16 && d.np == 2 && !d.bf16) ? 1 : 128) && d.N > 0 && d.K >= BK && (d.K % BK) == 0 && (d.a_sz & 3) == 0 && d.batch >= 1 && d.batch <= 8 &&
            wsplit_bytes(d.N, d.K) < 0x40000000UL && (d.a_ptrs[0] ? true : pal(d.A));
  if (d.a_sk) {      // A contiguous along m, rows split by a_mdiv
    ok = ok && d.a_mdiv > 0 && (d.a_mdiv & 3) == 0 && (d.a_sk & 3) == 0 && (d.a_sdiv & 3) == 0 && (d.M & 3) == 0 &&
         ((long)((d.M - 1) / d.a_mdiv + 1) * d.a_sdiv + (long)d.K * d.a_sk) * 4 < 0x40000000L;
  } else {
    ok = ok && (d.a_sm & 3) == 0 && ((long)d.M * d.a_sm + d.K) * 4 < 0x40000000L;
  }
  for (int t = 0; t < 8; ++t) ok = ok && pal(d.a_ptrs[t]);
  return ok ? 1 : 0;
}

int gemm_w_fill_job(const WGemm& d, gw::WArgs& g, long* nblk, int bn) {
  CA_CHECK_ARG(gemm_w_supported(d), "gemm_w: unsupported shape M=%d N=%d K=%d", d.M, d.N, d.K);
  CA_CHECK_ARG((d.A || d.a_ptrs[0]) && d.Wf && (d.C || d.c_ptrs[0]), "gemm_w: null operand");
  g = WArgs{};
  g.A = d.A; g.a_sz = d.a_sz; g.a_sm = d.a_sm; g.a_sk = d.a_sk; g.a_mdiv = d.a_mdiv; g.a_sdiv = d.a_sdiv;
  g.kband_n = d.kband_n;
  if (d.kband_n > 0) {
    CA_CHECK_ARG(d.kband_n % bn == 0 && (d.N + d.kband_n - 1) / d.kband_n <= 3, "gemm_w: kband_n must be a multiple of the tile width with at most 3 bands");
    for (int t = 0; t < 3; ++t) {
      CA_CHECK_ARG(d.kband_lo[t] >= 0 && d.kband_hi[t] <= d.K && d.kband_lo[t] < d.kband_hi[t] && d.kband_lo[t] % BK == 0 && d.kband_hi[t] % BK == 0,
                   "gemm_w: k bands must lie in [0,K] and be multiples of 32");
      g.kband_lo[t] = d.kband_lo[t]; g.kband_hi[t] = d.kband_hi[t];
    }
  }
  g.Wf = d.Wf; g.wf_bytes = (unsigned)wsplit_bytes(d.N, d.K);
  g.C = d.C; g.c_sz = d.c_sz; g.c_sm = d.c_sm;
  for (int t = 0; t < 8; ++t) { g.a_ptrs[t] = d.a_ptrs[t]; g.c_ptrs[t] = d.c_ptrs[t]; }
  g.bias_n = d.bias_n; g.oscale = d.out_scale != 0.f ? d.out_scale : 1.f;
  g.ascale = (d.f16 && d.np == 2 && !d.bf16) ? 1.0f / kF16WScale : 1.0f;
  g.status = g.ascale != 1.0f ? d.status : nullptr;
  CA_CHECK_ARG(!d.rowbits || (d.a_sk == 0 && bn == BN && (d.M + 31) / 32 <= kRowBitsMaxWords),
               "gemm_w: a row bitmap needs a row-major A, the four-wave tile and at most %d rows", 32 * kRowBitsMaxWords);
  g.rowbits = d.rowbits; g.rowcnt = nullptr; g.rowcnt_words = (d.M + 31) / 32;
  g.M = d.M; g.N = d.N; g.K = d.K;
  const long ntn = (d.N + bn - 1) / bn, ntm = (d.M + BM - 1) / BM;
  g.xcd_group = ntm >= 32 ? 1 : 0;
  *nblk = g.xcd_group ? (long)d.batch * ntn * ((ntm + 7) / 8) * 8 : (long)d.batch * ntn * ntm;
  return 0;
}

// one launch for n = 1 or 2 GEMMs
int launch_gemm_w(const WGemm* d, int n, hipStream_t s, const RowFlagJob* rows) {
  CA_CHECK_ARG(n == 1 || n == 2, "gemm_w: 1 or 2 jobs per launch");
  WJobs jobs = {};
  long nb[2] = {0, 0};
  // single-product mode with wide outputs: 128 x 256 tiles on 512 threads (the A rows are staged once for twice the columns)
  static const int wide2 = dev_env_int("COATTN_GEMMW_WIDE2", 0);   // developer switch
  static const int wide32 = dev_env_int("COATTN_GEMMW_WIDE32", 0);  // developer switch: the forward's (3, 2) launch
  static const int wide3 = dev_env_int("COATTN_GEMMW_WIDE3", 0);    // developer switch: the exact width on 128 x 256 tiles
  const bool mixed32 = wide32 && !d[0].bf16 && n == 2 && d[0].np != 2 && d[1].np == 2;
  const bool exact33 = wide3 && !d[0].bf16 && d[0].np != 2 && (n == 1 || d[1].np != 2);
  bool wide = d[0].bf16 != 0 || (wide2 && d[0].np == 2 && (n == 1 || d[1].np == 2)) || mixed32 || exact33;
  for (int i = 0; i < n; ++i) wide = wide && d[i].N % 256 == 0 && (d[i].kband_n == 0 || d[i].kband_n % 256 == 0);
  for (int i = 0; i < n; ++i) CA_TRY(gemm_w_fill_job(d[i], jobs.job[i], &nb[i], wide ? 256 : BN));
  CA_CHECK_ARG(nb[0] + nb[1] < 2147483647L, "gemm_w: grid too large");
  jobs.first1 = (int)nb[0];
  CA_CHECK_ARG(n == 1 || d[1].a_sk == 0, "gemm_w: only the first job may have an m-contiguous A operand");
  CA_CHECK_ARG(n == 1 || d[1].bf16 == d[0].bf16, "gemm_w: the jobs of a launch share the precision mode");
  const bool two0 = d[0].np == 2, two1 = n == 2 ? d[1].np == 2 : two0;      // fp32 mode: the width of each job
  if (rows) {                                         // its workgroups first; the tiles of the job with its bitmap wait for them
    CA_TRY(check_rowflag_job(rows));
    CA_CHECK_ARG(rows->rowcnt != nullptr && !wide && !d[0].bf16 && d[0].np != 2 && (n == 1 || d[1].np != 2) && !d[0].f16,
                 "gemm_w: a row-flag job rides in the exact four-wave launch only");
    jobs.rows = *rows;
    jobs.nflag = (rows->batch * ((rows->M + 31) / 32) + 7) / 8 * 8;   // (a multiple of the XCD count: the tiles keep their XCDs)
    for (int i = 0; i < n; ++i)
      if (d[i].rowbits == rows->rowbits) { jobs.job[i].rowcnt = rows->rowcnt; jobs.job[i].rowcnt_words = (rows->M + 31) / 32; }
  }
  const dim3 grid((unsigned)(jobs.nflag + nb[0] + nb[1]));
  const bool h0 = d[0].f16 && two0 && !d[0].bf16, h1 = n == 2 ? (d[1].f16 && two1 && !d[1].bf16) : h0;
  CA_CHECK_ARG(h0 == h1, "gemm_w: the jobs of a launch share the piece format");
  if (h0 && wide) {                                   // (developer switch COATTN_GEMMW_WIDE2: 128 x 256 tiles, eight waves --
                                                      //  the forward's launch 96.8 -> 93.9 us at N = 196, 48.8 -> 52.8 at N = 49: off)
    if (d[0].a_sk) hipLaunchKernelGGL((gemm_w_kernel<true, 2, 8, 2, true>), grid, dim3(512), 0, s, jobs);
    else hipLaunchKernelGGL((gemm_w_kernel<false, 2, 8, 2, true>), grid, dim3(512), 0, s, jobs);
  } else if (h0) {                                    // two FP16 pieces (the forward's projections)
    if (d[0].a_sk) hipLaunchKernelGGL((gemm_w_kernel<true, 2, 4, 2, true>), grid, dim3(256), 0, s, jobs);
    else hipLaunchKernelGGL((gemm_w_kernel<false, 2, 4, 2, true>), grid, dim3(256), 0, s, jobs);
  } else if (wide && mixed32) {
    if (d[0].a_sk) hipLaunchKernelGGL((gemm_w_kernel<true, 3, 8, 2>), grid, dim3(512), 0, s, jobs);
    else hipLaunchKernelGGL((gemm_w_kernel<false, 3, 8, 2>), grid, dim3(512), 0, s, jobs);
  } else if (wide && exact33) {
    if (d[0].a_sk) hipLaunchKernelGGL((gemm_w_kernel<true, 3, 8>), grid, dim3(512), 0, s, jobs);
    else hipLaunchKernelGGL((gemm_w_kernel<false, 3, 8>), grid, dim3(512), 0, s, jobs);
  } else if (wide && !d[0].bf16) {
    if (d[0].a_sk) hipLaunchKernelGGL((gemm_w_kernel<true, 2, 8>), grid, dim3(512), 0, s, jobs);
    else hipLaunchKernelGGL((gemm_w_kernel<false, 2, 8>), grid, dim3(512), 0, s, jobs);
  } else if (wide) {
    if (d[0].a_sk) hipLaunchKernelGGL((gemm_w_kernel<true, 1, 8>), grid, dim3(512), 0, s, jobs);
    else hipLaunchKernelGGL((gemm_w_kernel<false, 1, 8>), grid, dim3(512), 0, s, jobs);
  } else if (d[0].bf16) {
    if (d[0].a_sk) hipLaunchKernelGGL((gemm_w_kernel<true, 1>), grid, dim3(256), 0, s, jobs);
    else hipLaunchKernelGGL((gemm_w_kernel<false, 1>), grid, dim3(256), 0, s, jobs);
  } else if (two0 && two1) {
    if (d[0].a_sk) hipLaunchKernelGGL((gemm_w_kernel<true, 2>), grid, dim3(256), 0, s, jobs);
    else hipLaunchKernelGGL((gemm_w_kernel<false, 2>), grid, dim3(256), 0, s, jobs);
  } else if (two1) {                                  // P_v exact, P_q on two pieces
    if (d[0].a_sk) hipLaunchKernelGGL((gemm_w_kernel<true, 3, 4, 2>), grid, dim3(256), 0, s, jobs);
    else hipLaunchKernelGGL((gemm_w_kernel<false, 3, 4, 2>), grid, dim3(256), 0, s, jobs);
  } else if (two0) {
    if (d[0].a_sk) hipLaunchKernelGGL((gemm_w_kernel<true, 2, 4, 3>), grid, dim3(256), 0, s, jobs);
    else hipLaunchKernelGGL((gemm_w_kernel<false, 2, 4, 3>), grid, dim3(256), 0, s, jobs);
  } else {
    if (d[0].a_sk) hipLaunchKernelGGL((gemm_w_kernel<true, 3>), grid, dim3(256), 0, s, jobs);
    else hipLaunchKernelGGL((gemm_w_kernel<false, 3>), grid, dim3(256), 0, s, jobs);
  }
  CA_CHECK_LAUNCH("gemm_w");
  return 0;
}
