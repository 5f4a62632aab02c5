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
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDR = 40;                        // [row][k] bf16 image row stride (elements): conflict-free b128 reads
constexpr int kFragBytes = 1024;               // one fragment: 64 lanes x 8 bf16
constexpr int kChunkBytes = 3 * kFragBytes;    // the three pieces of one (column tile, 16-k step)

struct WArgs {
  const float* A; const float* a_ptrs[8]; long a_sz; int a_sm;
  int a_sk, a_mdiv; long a_sdiv;               // AM: element (m, k) at (m / a_mdiv) * a_sdiv + m % a_mdiv + k * a_sk
  const void* Wf; unsigned wf_bytes;
  float* C; float* c_ptrs[8]; long c_sz; int c_sm;
  const float* bias_n; float oscale;
  int M, N, K, xcd_group;
};

struct WJobs { WArgs job[2]; int first1; };   // blocks [0, first1) work on job 0, the rest on job 1

struct SplitJob { const float* W; void* out; int N, K, trans, ld; };
struct SplitArgs { SplitJob job[3]; int njobs; };

// One wave per (32-column tile nt, 16-k step ks): lane (li = lane & 31, lh = lane >> 5) holds
// Bw(k = 16 ks + 8 lh + e, n = 32 nt + li), e = 0..7 -- the B operand layout of v_mfma_f32_32x32x16_bf16.
// trans = 0: Bw(k, n) = W[n][k] (y = x W^T); trans = 1: Bw(k, n) = W[k][n] (dx = dy W).
__global__ __launch_bounds__(256) void wsplit_kernel(const SplitArgs a) {
  const SplitJob j = a.job[blockIdx.y];
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int ks16 = (j.K + 15) / 16, nt32 = (j.N + 31) / 32;
  const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (chunk >= ks16 * nt32) return;
  const int nt = chunk / ks16, ks = chunk % ks16;
  const int n = 32 * nt + li, k0 = 16 * ks + 8 * lh;
  f32x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e;
    v[e] = (n < j.N && k < j.K) ? (j.trans ? j.W[(size_t)k * j.ld + n] : j.W[(size_t)n * j.ld + k]) : 0.f;
  }
  bf16x8 p[3];
  split3(v, p);
  char* out = (char*)j.out + (size_t)chunk * kChunkBytes + lane * 16;
#pragma unroll
  for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8*>(out + q * kFragBytes) = p[q];
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int LDT = BM + 32;                   // AM: [k][row] image row stride (conflict-free writes + transposed reads)

// AM = false: A rows contiguous along k ([row][k] images, one ds_read_b128 per fragment).
// AM = true : A contiguous along m -- channel-major image features [B, d, N] read in place (model.py:215-217),
//             row m = (sample, location) split by a_mdiv: [k][row] images, fragments through ds_read_b64_tr_b16.
template <bool AM>
__device__ __forceinline__ void gemm_w_body(const WArgs& g, const int id, short* const smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
  // XCD-aware tile order (as gemm.hip): the column tiles of one row tile share an XCD's L2
  int m0, n0, z;
  {
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    const int x = id & 7, slot = id >> 3;
    if (!g.xcd_group) {
      const int per = ntm * ntn;
      z = id / per;
      const int t = id % per;
      m0 = (t / ntn) * BM; n0 = (t % ntn) * BN;
    } else {
      const int per = ntn * ((ntm + 7) / 8);
      z = slot / per;
      const int t = slot % per;
      const int mt = (t / ntn) * 8 + x;
      m0 = mt * BM; n0 = (t % ntn) * BN;
      if (mt >= ntm) return;
    }
  }
  const float* Ab = g.a_ptrs[0] ? g.a_ptrs[z & 7] : g.A + (long)z * g.a_sz;
  const long a_bytes = AM ? ((long)((g.M - 1) / g.a_mdiv) * g.a_sdiv + (g.a_mdiv - 1) + (long)(g.K - 1) * g.a_sk + 1) * 4
                          : ((long)(g.M - 1) * g.a_sm + g.K) * 4;
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(Ab, (unsigned)a_bytes);
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(g.Wf, g.wf_bytes);
  const int KS = g.K / BK;                       // K % 32 == 0 (host check)

  // A staging: 4 float4 per thread and step; a wave's load covers 8 rows x 128 B (whole lines)
  // (AM: float4 = 4 consecutive rows of one k; a wave's load covers 2 k x 512 B; a_mdiv % 4 == 0 keeps the 4 rows in one sample)
  int a_voff[4], a_lds[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (AM) {
      const int k = (tid >> 5) + 8 * i, m = (tid & 31) * 4, row = m0 + m;
      a_voff[i] = row < g.M ? (int)(((long)(row / g.a_mdiv) * g.a_sdiv + row % g.a_mdiv + (long)k * g.a_sk) * 4) : 0x40000000;
      a_lds[i] = k * LDT + m;
    } else {
      const int m = (tid >> 3) + 32 * i, k = (tid & 7) * 4;
      a_voff[i] = (m0 + m) < g.M ? ((m0 + m) * g.a_sm + k) * 4 : 0x40000000;   // rows past M read 0
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
  f32x4 raw[4];
  bf16x8 bq[3][2][3];                            // [ring][tile j][piece]
  bf16x8 af[2][3][2];                            // [half][piece][tile i]
  unsigned ph[2], pm[2], pl[2];                  // packed pieces of the raw[i] being split (its two pairs)
  float ra[2], rb[2];
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};      // (gemm.hip's order)
  constexpr int RQ[3] = {2, 0, 1};               // fragment read order = order of first use
  constexpr int IMG = BM * LDR;                  // elements of one piece image
  auto load_a = [&](int i, int s) { raw[i] = buf_load4(rs_a, a_voff[i], s * a_kstep); };
  auto load_b = [&](int ring, int k, int half) {
    const int j = k / 3, q = k % 3;
    bq[ring][j][q] = __builtin_bit_cast(bf16x8, buf_load4(rs_w, w_voff[j] + q * kFragBytes, half * kChunkBytes));
  };
  auto read_a = [&](const short* img, int h, int k) {
    const int q = RQ[k >> 1], i = k & 1;
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
#ifdef GEMMW_NOSPLIT
    if (st == 0) ph[e] = pm[e] = pl[e] = cvt_pk_bf16(raw[i][2 * e], raw[i][2 * e + 1]);
#else
    if (st == 0) {
      ph[e] = cvt_pk_bf16(raw[i][2 * e], raw[i][2 * e + 1]);
      ra[e] = sub1(raw[i][2 * e], __builtin_bit_cast(float, ph[e] << 16));
      rb[e] = sub1(raw[i][2 * e + 1], __builtin_bit_cast(float, ph[e] & 0xffff0000u));
    } else if (st == 1) {
      pm[e] = cvt_pk_bf16(ra[e], rb[e]);
      ra[e] = sub1(ra[e], __builtin_bit_cast(float, pm[e] << 16));
      rb[e] = sub1(rb[e], __builtin_bit_cast(float, pm[e] & 0xffff0000u));
    } else {
      pl[e] = cvt_pk_bf16(ra[e], rb[e]);
    }
#endif
  };
  auto write_a = [&](short* img, int i, int q) {
    const u32x2 v = q == 0 ? u32x2{ph[0], ph[1]} : (q == 1 ? u32x2{pm[0], pm[1]} : u32x2{pl[0], pl[1]});
    *reinterpret_cast<u32x2*>(&img[q * IMG + a_lds[i]]) = v;
  };
  // half step HH of step s: MFMAs on af[HH] x bq[BU]; the loads of half 2 s + HH + 2 go to bq[BL]
  auto half = [&](auto HHc, auto BUc, auto BLc, int s, const short* cur, short* nxt) {
    constexpr int HH = decltype(HHc)::value, BU = decltype(BUc)::value, BL = decltype(BLc)::value;
#pragma unroll
    for (int n = 0; n < 24; ++n) {
      const int t = n >> 2, i = (n >> 1) & 1, j = n & 1;
#ifndef GEMMW_NOMFMA
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[HH][PA[t]][i], bq[BU][j][PB[t]], acc[i][j], 0, 0, 0);
#else
      if (t == 0) acc[i][j][0] += __builtin_bit_cast(float, (int)af[HH][PA[t]][i][0] ^ (int)bq[BU][j][PB[t]][0]);
#endif
#ifndef GEMMW_NOB
      if (n < 6) load_b(BL, n, 2 * s + HH + 2);
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
  for (int i = 0; i < 4; ++i) load_a(i, 0);
#pragma unroll
  for (int k = 0; k < 6; ++k) { load_b(0, k, 0); load_b(1, k, 1); }
  short* const img0 = smem;
  short* const img1 = smem + 3 * IMG;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
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
  for (; s + 3 <= KS; s += 3) {                            // (one loop exit: the accumulators stay in place)
    step(I0{}, I1{}, I2{}, s);
    step(I2{}, I0{}, I1{}, s + 1);
    step(I1{}, I2{}, I0{}, s + 2);
  }
  if (s < KS) {
    step(I0{}, I1{}, I2{}, s);
    if (s + 1 < KS) step(I2{}, I0{}, I1{}, s + 1);
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
      const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row >= g.M) continue;
      float* crow = Cb + (long)row * g.c_sm;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (col[j] < g.N) crow[col[j]] = (acc[i][j][r] + bn[j]) * g.oscale;
    }
}

// Up to two independent GEMMs per launch (P_v and P_q of the forward): the second one's workgroups fill the slots
// the first one's last, partial round of workgroups would leave idle.  AM0: layout of job 0's A operand.
template <bool AM0>
__global__ __launch_bounds__(256, 2) void gemm_w_kernel(const WJobs jobs) {
  __shared__ __attribute__((aligned(16))) short smem[2 * 3 * BM * LDR];  // 61,440 B: two workgroups per CU
  static_assert(BM * LDR == BK * LDT, "both image layouts have the same size");
  if ((int)blockIdx.x < jobs.first1) gemm_w_body<AM0>(jobs.job[0], (int)blockIdx.x, smem);
  else gemm_w_body<false>(jobs.job[1], (int)blockIdx.x - jobs.first1, smem);
}

}  // namespace

size_t wsplit_bytes(int N, int K) { return (size_t)((N + 31) / 32) * ((K + 15) / 16) * kChunkBytes; }

int launch_wsplit(const WSplit* jobs, int njobs, hipStream_t s) {
  CA_CHECK_ARG(njobs >= 1 && njobs <= 3, "wsplit: 1 to 3 jobs per launch");
  SplitArgs a = {};
  a.njobs = njobs;
  int chunks = 0;
  for (int i = 0; i < njobs; ++i) {
    CA_CHECK_ARG(jobs[i].W && jobs[i].out && jobs[i].N > 0 && jobs[i].K > 0, "wsplit: bad job");
    a.job[i] = SplitJob{jobs[i].W, jobs[i].out, jobs[i].N, jobs[i].K, jobs[i].trans, jobs[i].ld};
    const int c = ((jobs[i].N + 31) / 32) * ((jobs[i].K + 15) / 16);
    chunks = c > chunks ? c : chunks;
  }
  hipLaunchKernelGGL(wsplit_kernel, dim3((chunks + 3) / 4, njobs), dim3(256), 0, s, a);
  CA_CHECK_LAUNCH("wsplit");
  return 0;
}

int gemm_w_supported(const WGemm& d) {
  auto pal = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  bool ok = d.M >= 128 && d.N > 0 && d.K >= BK && (d.K % BK) == 0 && (d.a_sz & 3) == 0 && d.batch >= 1 && d.batch <= 8 &&
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

static int fill_job(const WGemm& d, WArgs& g, long* nblk) {
  CA_CHECK_ARG(gemm_w_supported(d), "gemm_w: unsupported shape M=%d N=%d K=%d", d.M, d.N, d.K);
  CA_CHECK_ARG((d.A || d.a_ptrs[0]) && d.Wf && (d.C || d.c_ptrs[0]), "gemm_w: null operand");
  g = WArgs{};
  g.A = d.A; g.a_sz = d.a_sz; g.a_sm = d.a_sm; g.a_sk = d.a_sk; g.a_mdiv = d.a_mdiv; g.a_sdiv = d.a_sdiv;
  g.Wf = d.Wf; g.wf_bytes = (unsigned)wsplit_bytes(d.N, d.K);
  g.C = d.C; g.c_sz = d.c_sz; g.c_sm = d.c_sm;
  for (int t = 0; t < 8; ++t) { g.a_ptrs[t] = d.a_ptrs[t]; g.c_ptrs[t] = d.c_ptrs[t]; }
  g.bias_n = d.bias_n; g.oscale = d.out_scale != 0.f ? d.out_scale : 1.f;
  g.M = d.M; g.N = d.N; g.K = d.K;
  const long ntn = (d.N + BN - 1) / BN, ntm = (d.M + BM - 1) / BM;
  g.xcd_group = ntm >= 32 ? 1 : 0;
  *nblk = g.xcd_group ? (long)d.batch * ntn * ((ntm + 7) / 8) * 8 : (long)d.batch * ntn * ntm;
  return 0;
}

// one launch for n = 1 or 2 GEMMs
int launch_gemm_w(const WGemm* d, int n, hipStream_t s) {
  CA_CHECK_ARG(n == 1 || n == 2, "gemm_w: 1 or 2 jobs per launch");
  WJobs jobs = {};
  long nb[2] = {0, 0};
  for (int i = 0; i < n; ++i) CA_TRY(fill_job(d[i], jobs.job[i], &nb[i]));
  CA_CHECK_ARG(nb[0] + nb[1] < 2147483647L, "gemm_w: grid too large");
  jobs.first1 = (int)nb[0];
  CA_CHECK_ARG(n == 1 || d[1].a_sk == 0, "gemm_w: only the first job may have an m-contiguous A operand");
  if (d[0].a_sk) hipLaunchKernelGGL(gemm_w_kernel<true>, dim3((unsigned)(nb[0] + nb[1])), dim3(256), 0, s, jobs);
  else hipLaunchKernelGGL(gemm_w_kernel<false>, dim3((unsigned)(nb[0] + nb[1])), dim3(256), 0, s, jobs);
  CA_CHECK_LAUNCH("gemm_w");
  return 0;
}
