// General strided, batched fp32 GEMM of the library: the exact 3-way bf16 split on v_mfma_f32_32x32x16_bf16 for
// aligned shapes (MODE 2), the gfx950 f32 MFMA (v_mfma_f32_32x32x2_f32) otherwise.
//
// Every contraction of the general-shape implementation, dV, and whatever the two hand-scheduled kernels do not
// take (gemm_w.hip: projections P_v = V W_v^T + b_v, P_q = Q W_q^T + b_q of model.py:380-384 and dQ = dP_q W_q;
// gemm_tn.hip: the weight gradients): M < 128, K % 32, channel counts that are not multiples of 128, unaligned rows.
// Operands are addressed through element strides so that V is consumed in its physical
// channel-major [B,d,N] layout (model.py:215-217) without a transpose pass.
//
// Tile BM x 128 per 256-thread workgroup (4 waves), BK = 16.  Software pipeline: the global
// loads of K-step s+1 are issued into registers before the MFMAs of step s and written to the
// other LDS buffer afterwards (one barrier per step).  Per-thread element offsets (including
// the optional row split m -> (m / mdiv, m % mdiv)) are hoisted out of the K loop as 32-bit
// offsets.  LDS images are As[k][m] / Bs[k][n]: an MFMA operand read is 32 consecutive floats
// per half wave (conflict-free ds_read_b32).  Numerics: exact fp32 fmaf chain in k order.
#include "common.h"
#include <stdlib.h>

namespace {

struct GemmK {
  const float* A; const float* B; const float* Cin; float* C;
  const float* bias_n; const float* bias_m;
  int M, N, K, inner, inner_total, ksplit, act;
  float beta, oscale;                    // oscale: the products + biases are multiplied by it before Cin / act (1 = plain)
  int a_mfast, b_nfast;
  long a_sm, a_sk, a_sz, a_si, a_mdiv, a_sdiv;
  long b_sk, b_sn, b_sz, b_si;
  long c_sm, c_sn, c_sz, c_mdiv, c_sdiv;
  long cin_sm, cin_sn, cin_sz, cin_mdiv, cin_sdiv;
  const float* a_ptrs[8]; const float* b_ptrs[8]; float* c_ptrs[8]; const float* cin_ptrs[8];
  int ptr_by_inner, b_imod, xcd_group;
  int kband_n, kband_lo[3], kband_hi[3];
};

// The split-row case and the precise tanh are kept OUT OF LINE: the epilogues below are fully unrolled over the
// accumulator registers (64 .. 128 elements per thread), and an inlined 64-bit division pair plus tanhf per
// element made every kernel ~23,000 instructions (~190 KB) that each workgroup streams through the instruction
// cache -- a fixed ~20 us per launch, whatever the problem size.
__device__ __attribute__((noinline)) long row_off_split(long m, long sm, long mdiv, long sdiv) {
  return (m / mdiv) * sdiv + (m % mdiv) * sm;
}
__device__ __forceinline__ long row_off(long m, long sm, long mdiv, long sdiv) {
  return mdiv > 0 ? row_off_split(m, sm, mdiv, sdiv) : m * sm;
}
__device__ __attribute__((noinline)) float tanh_outlined(float x) { return tanhf(x); }

template <int BM>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmK g) {
  constexpr int BN = 128, BK = 16;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int WM = (BM >= 128) ? 2 : 1;  // waves along m
  constexpr int WN = 4 / WM;               // waves along n
  constexpr int TM = BM / WM / 32;         // 32x32 tiles per wave along m
  constexpr int TN = BN / WN / 32;
  constexpr int EA = BM * BK / 256;        // A elements per thread per K-step
  constexpr int EB = BN * BK / 256;
  __shared__ float As[2][BK * LDA];
  __shared__ float Bs[2][BK * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, z = blockIdx.z;

  int kbeg = 0, kend = g.K;
  if (g.ksplit > 0) {
    kbeg = z * g.ksplit;
    kend = min(g.K, kbeg + g.ksplit);
  }
  if (g.kband_n > 0) {                              // columns [j*kband_n, (j+1)*kband_n) contract over a sub-range of k
    const int band = n0 / g.kband_n;
    kbeg = max(kbeg, g.kband_lo[band]);
    kend = min(kend, g.kband_hi[band]);
  }
  int ninner = g.inner;
  if (g.inner_total > 0) ninner = max(0, min(g.inner, g.inner_total - z * g.inner));
  const int ksteps = (kend - kbeg + BK - 1) / BK;
  const int nsteps = ninner * max(ksteps, 0);

  // hoisted per-thread element coordinates: tile-local (m,k) / (k,n), 32-bit global offsets
  int a_off[EA], a_lds[EA], a_k[EA];
  bool a_ok[EA];
#pragma unroll
  for (int i = 0; i < EA; ++i) {
    const int idx = tid + i * 256;
    int m, k;
    if (g.a_mfast) { m = idx % BM; k = idx / BM; } else { k = idx % BK; m = idx / BK; }
    a_k[i] = k;
    a_lds[i] = k * LDA + m;
    a_ok[i] = (m0 + m) < g.M;
    a_off[i] = a_ok[i] ? (int)(row_off(m0 + m, g.a_sm, g.a_mdiv, g.a_sdiv) + (long)k * g.a_sk) : 0;
  }
  int b_off[EB], b_lds[EB], b_k[EB];
  bool b_ok[EB];
#pragma unroll
  for (int i = 0; i < EB; ++i) {
    const int idx = tid + i * 256;
    int n, k;
    if (g.b_nfast) { n = idx % BN; k = idx / BN; } else { k = idx % BK; n = idx / BK; }
    b_k[i] = k;
    b_lds[i] = k * LDB + n;
    b_ok[i] = (n0 + n) < g.N;
    b_off[i] = b_ok[i] ? (int)((long)k * g.b_sk + (long)(n0 + n) * g.b_sn) : 0;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float ra[EA], rb[EB];
  // step -> (inner index, k0)
  auto load_step = [&](int step) {
    const int ii = step / ksteps, k0 = kbeg + (step - ii * ksteps) * BK;
    const int ig = g.inner_total > 0 ? z * g.inner + ii : ii;      // global inner index
    const int pt = g.ptr_by_inner ? ig : z;
    const float* Ab = (g.a_ptrs[0] ? g.a_ptrs[pt & 7] : g.A + (long)z * g.a_sz + (long)ii * g.a_si)
                      + (long)k0 * g.a_sk;
    const float* Bb = (g.b_ptrs[0] ? g.b_ptrs[pt & 7]
                                   : g.B + (g.b_imod > 0 ? (long)(ig % g.b_imod) * g.b_si
                                                         : (long)z * g.b_sz + (long)ii * g.b_si))
                      + (long)k0 * g.b_sk;
    if (k0 + BK <= kend) {
#pragma unroll
      for (int i = 0; i < EA; ++i) ra[i] = a_ok[i] ? Ab[a_off[i]] : 0.f;
#pragma unroll
      for (int i = 0; i < EB; ++i) rb[i] = b_ok[i] ? Bb[b_off[i]] : 0.f;
    } else {
      const int klim = kend - k0;
#pragma unroll
      for (int i = 0; i < EA; ++i) ra[i] = (a_ok[i] && a_k[i] < klim) ? Ab[a_off[i]] : 0.f;
#pragma unroll
      for (int i = 0; i < EB; ++i) rb[i] = (b_ok[i] && b_k[i] < klim) ? Bb[b_off[i]] : 0.f;
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int i = 0; i < EA; ++i) As[buf][a_lds[i]] = ra[i];
#pragma unroll
    for (int i = 0; i < EB; ++i) Bs[buf][b_lds[i]] = rb[i];
  };

  if (nsteps > 0) {
    load_step(0);
    store_step(0);
  }
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int buf = step & 1;
    if (step + 1 < nsteps) load_step(step + 1);
    const float* as = As[buf];
    const float* bs = Bs[buf];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
      const int krow = kk + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = as[krow * LDA + wr * (TM * 32) + i * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bs[krow * LDB + wc * (TN * 32) + j * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (step + 1 < nsteps) store_step(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* Cb = g.c_ptrs[0] ? g.c_ptrs[z & 7] : g.C + (long)z * g.c_sz;
  const float* Cinb = g.cin_ptrs[0] ? g.cin_ptrs[z & 7] : (g.Cin ? g.Cin + (long)z * g.cin_sz : nullptr);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wc * (TN * 32) + j * 32 + (lane & 31);
      const float bn = (g.bias_n && col < g.N) ? g.bias_n[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.M && col < g.N) {
          float v = acc[i][j][r] + bn;
          if (g.bias_m) v += g.bias_m[row];
          v *= g.oscale;
          if (Cinb) v += g.beta * Cinb[row_off(row, g.cin_sm, g.cin_mdiv, g.cin_sdiv) + (long)col * g.cin_sn];
          if (g.act == 1) v = tanh_outlined(v);
          Cb[row_off(row, g.c_sm, g.c_mdiv, g.c_sdiv) + (long)col * g.c_sn] = v;
        }
      }
    }
}


typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ short f2bf(float x) {            // round to nearest even (inputs are finite)
  unsigned u = __builtin_bit_cast(unsigned, x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (short)(u >> 16);
}

// ---- aligned fast path: BK = 32, float4 global loads, one LDS buffer + register prefetch --------
// Operand images: an operand whose fast (contiguous) axis is the tile row/column index (AM / BN)
// is staged as [k][m] with 16-byte LDS writes; an operand that is contiguous along k is staged
// as [m][k] with an odd row stride (33), so that both the transposing 4-byte writes and the
// 32-lane MFMA operand reads are bank-conflict free.
// MODE 1: same addressing, but the operands are rounded to bf16 while they are staged ([row][k]
// images, 80-byte rows, one ds_read_b128 per operand) and contracted with v_mfma_f32_32x32x16_bf16
// (fp32 accumulate / output): the reduced-precision mode of COATTN_FLAG_BF16_PROJ.
// MODE 2: fp32-accurate product on the bf16 MFMA: every fp32 operand element is split exactly into
// three bf16 pieces x = hi + mid + lo (24 significand bits) while it is staged, and the six partial
// products down to relative order 2^-16 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) are accumulated
// in fp32.  Each bf16*bf16 product is exact in fp32; what is dropped (mid*lo, lo*mid, lo*lo) is below
// 2^-23 of |x||y|, the size of one fp32 rounding.  The bf16 MFMA has 16x the fp32 MFMA rate, so the
// six products cost 6/16 of the fp32 MFMA time.
typedef __bf16 bfv4 __attribute__((ext_vector_type(4)));
typedef float f32x2g __attribute__((ext_vector_type(2)));
typedef __bf16 bfv2g __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {      // one v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2g{a, b}), bfv2g));
}
__device__ __forceinline__ float sub1(float a, float b) {                 // one v_sub_f32 (never paired into v_pk_add_f32)
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <bool AM, bool BN_, int BM, int MODE = 0>
__global__ __launch_bounds__(256) void gemm_f32_vec_kernel(const GemmK g) {
  constexpr bool BF = MODE != 0;
  constexpr int NIMG = MODE == 2 ? 3 : 1;
  constexpr int BN = 128, BK = 32;
  constexpr int TM = BM / 64;              // 32-row MFMA tiles per wave along m (waves are 2 x 2)
  constexpr int FA = BM * BK / 4 / 256;    // float4 per thread per K-step for A (4 or 6)
  constexpr int LDM = BM + 4;              // [k][m] image row stride (A)
  constexpr int LDN = BN + 4;              // [k][n] image row stride (B)
  constexpr int LDK = BK + 1;              // [m][k] / [n][k] image row stride
  constexpr int ASZ = AM ? BK * LDM : BM * LDK;
  constexpr int BSZ = BN_ ? BK * LDN : BN * LDK;
  constexpr int LDR = 40;                  // bf16 [row][k] image row stride (elements)
  // MODE 2, operand contiguous along its tile index (A along m / B along n): the pieces are staged as [k][m]
  // bf16 images straight from the coalesced float4 loads (one 8-byte LDS write per piece) and the MFMA
  // fragments (8 consecutive k of one row) are read with ds_read_b64_tr_b16, the transposing LDS read of
  // gfx950.  Row stride BM + 32: conflict-free for both the writes and the transposed reads.
  constexpr bool TRA = MODE == 2 && AM, TRB = MODE == 2 && BN_;
  constexpr int LDTA = BM + 32, LDTB = BN + 32;
  constexpr int AIMG = TRA ? BK * LDTA : BM * LDR;            // elements of one A image
  constexpr int BIMG = TRB ? BK * LDTB : BN * LDR;
  constexpr int LDS_BYTES = BF ? NIMG * (AIMG + BIMG) * 2 : (ASZ + BSZ) * 4;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  float* const As = reinterpret_cast<float*>(smem);
  float* const Bs = As + ASZ;
  short* const Ah = reinterpret_cast<short*>(smem);          // NIMG images of A
  short* const Bh = Ah + NIMG * AIMG;                        // NIMG images of B

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware tile order (speed only): workgroup ids i and i + 8 share an XCD and its L2, so the
  // n-tiles of one m-tile (they read the same A panel) get ids equal mod 8 when there are many m-tiles.
  int m0, n0, z;
  {
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    const int id = blockIdx.x, x = id & 7, slot = id >> 3;
    if (!g.xcd_group) {                              // plain order: few m-tiles, nothing to group
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
  int kbeg = 0, kend = g.K;
  if (g.ksplit > 0) {
    kbeg = z * g.ksplit;
    kend = min(g.K, kbeg + g.ksplit);
  }
  if (g.kband_n > 0) {                              // columns [j*kband_n, (j+1)*kband_n) contract over a sub-range of k
    const int band = n0 / g.kband_n;
    kbeg = max(kbeg, g.kband_lo[band]);
    kend = min(kend, g.kband_hi[band]);
  }
  int ninner = g.inner;
  if (g.inner_total > 0) ninner = max(0, min(g.inner, g.inner_total - z * g.inner));
  const int ksteps = (kend - kbeg + BK - 1) / BK;
  const int nsteps = ninner * max(ksteps, 0);

  // FA / 4 float4 per thread per operand; offsets hoisted (32-bit, element units)
  int a_off[FA], a_lds[FA], a_k[FA];
  bool a_ok[FA];
#pragma unroll
  for (int i = 0; i < FA; ++i) {
    const int idx = tid + 256 * i;                  // float4 index in the BM x 32 tile
    int m, k;
    if (AM && BF && !TRA) { m = idx % BM; k = (idx / BM) * 4; }   // bf16 [row][k] images: lanes run along m
    else if (AM) { k = idx / (BM / 4); m = (idx % (BM / 4)) * 4; }   // (coalesced dword loads), 4 consecutive k per thread
    else { m = idx >> 3; k = (idx & 7) * 4; }
    a_k[i] = k;
    a_lds[i] = TRA ? k * LDTA + m : (BF ? m * LDR + k : (AM ? k * LDM + m : m * LDK + k));
    a_ok[i] = (m0 + m) < g.M;                      // M % 4 == 0 on this path: whole float4 in or out
    a_off[i] = a_ok[i] ? (int)(row_off(m0 + m, g.a_sm, g.a_mdiv, g.a_sdiv) + (long)k * g.a_sk) : 0;
  }
  int b_off[4], b_lds[4], b_k[4];
  bool b_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    int n, k;
    if (BN_ && BF && !TRB) { n = idx % BN; k = (idx / BN) * 4; }
    else if (BN_) { k = idx >> 5; n = (idx & 31) * 4; }
    else { n = idx >> 3; k = (idx & 7) * 4; }
    b_k[i] = k;
    b_lds[i] = TRB ? k * LDTB + n : (BF ? n * LDR + k : (BN_ ? k * LDN + n : n * LDK + k));
    b_ok[i] = (n0 + n) < g.N;
    b_off[i] = b_ok[i] ? (int)((long)k * g.b_sk + (long)(n0 + n) * g.b_sn) : 0;
  }

  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[FA], rb[4];
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  // operand base pointers of inner index ii (table lookups happen only here, not per K-step)
  auto base_a = [&](int ii) -> const float* {
    const int ig = g.inner_total > 0 ? z * g.inner + ii : ii;
    const int pt = g.ptr_by_inner ? ig : z;
    return (g.a_ptrs[0] ? g.a_ptrs[pt & 7] : g.A + (long)z * g.a_sz + (long)ii * g.a_si) + (long)kbeg * g.a_sk;
  };
  auto base_b = [&](int ii) -> const float* {
    const int ig = g.inner_total > 0 ? z * g.inner + ii : ii;
    const int pt = g.ptr_by_inner ? ig : z;
    return (g.b_ptrs[0] ? g.b_ptrs[pt & 7]
                        : g.B + (g.b_imod > 0 ? (long)(ig % g.b_imod) * g.b_si
                                              : (long)z * g.b_sz + (long)ii * g.b_si)) + (long)kbeg * g.b_sk;
  };
  auto load_regs = [&](const float* Ab, const float* Bb, int klim) {     // K range % 4 == 0 on this path
#pragma unroll
    for (int i = 0; i < FA; ++i) {
      if constexpr (AM && BF && !TRA) {                          // 4 consecutive k of one row: strided dword loads
        const bool ok = a_ok[i] && a_k[i] < klim;
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[i][e] = ok ? Ab[a_off[i] + (long)e * g.a_sk] : 0.f;
      } else {
        ra[i] = (a_ok[i] && a_k[i] < klim) ? *reinterpret_cast<const f32x4*>(Ab + a_off[i]) : zero4;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (BN_ && BF && !TRB) {
        const bool ok = b_ok[i] && b_k[i] < klim;
#pragma unroll
        for (int e = 0; e < 4; ++e) rb[i][e] = ok ? Bb[b_off[i] + (long)e * g.b_sk] : 0.f;
      } else {
        rb[i] = (b_ok[i] && b_k[i] < klim) ? *reinterpret_cast<const f32x4*>(Bb + b_off[i]) : zero4;
      }
    }
  };
  auto store_step = [&]() {
    if constexpr (MODE == 2) {
      // exact 3-way split of each element, one image per piece
      // (a float4 holds 4 consecutive k of one row for [row][k] images, 4 consecutive rows of one k for [k][row])
      auto put = [&](short* img, int piece, int off, const f32x4& v) {
        // pairwise: one v_cvt_pk_bf16_f32 per piece and pair; the rounded halves come back as floats by a
        // shift / a mask, the residuals by single v_sub_f32 (11 VALU ops per pair of elements)
        unsigned h[2], m[2], l[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float a = v[2 * e], b = v[2 * e + 1];
          h[e] = cvt_pk_bf16(a, b);
          const float ra_ = sub1(a, __builtin_bit_cast(float, h[e] << 16));
          const float rb_ = sub1(b, __builtin_bit_cast(float, h[e] & 0xffff0000u));
          m[e] = cvt_pk_bf16(ra_, rb_);
          const float sa_ = sub1(ra_, __builtin_bit_cast(float, m[e] << 16));
          const float sb_ = sub1(rb_, __builtin_bit_cast(float, m[e] & 0xffff0000u));
          l[e] = cvt_pk_bf16(sa_, sb_);
        }
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(img + off) = u32x2{h[0], h[1]};
        *reinterpret_cast<u32x2*>(img + piece + off) = u32x2{m[0], m[1]};
        *reinterpret_cast<u32x2*>(img + 2 * piece + off) = u32x2{l[0], l[1]};
      };
#pragma unroll
      for (int i = 0; i < FA; ++i) put(Ah, AIMG, a_lds[i], ra[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i) put(Bh, BIMG, b_lds[i], rb[i]);
      return;
    }
    if constexpr (MODE == 1) {
#pragma unroll
      for (int i = 0; i < FA; ++i) {
        *reinterpret_cast<bf16x4*>(&Ah[a_lds[i]]) = bf16x4{f2bf(ra[i][0]), f2bf(ra[i][1]), f2bf(ra[i][2]), f2bf(ra[i][3])};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<bf16x4*>(&Bh[b_lds[i]]) = bf16x4{f2bf(rb[i][0]), f2bf(rb[i][1]), f2bf(rb[i][2]), f2bf(rb[i][3])};
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < FA; ++i) {
      if (AM) {
        *reinterpret_cast<f32x4*>(&As[a_lds[i]]) = ra[i];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) As[a_lds[i] + e] = ra[i][e];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (BN_) {
        *reinterpret_cast<f32x4*>(&Bs[b_lds[i]]) = rb[i];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[b_lds[i] + e] = rb[i][e];
      }
    }
  };

  int ii = 0, kidx = 0;
  const float* Ap = nullptr;
  const float* Bp = nullptr;
  if (nsteps > 0) {
    Ap = base_a(0);
    Bp = base_b(0);
    load_regs(Ap, Bp, kend - kbeg);
    store_step();
  }
  __syncthreads();
  const int li = lane & 31, lh = lane >> 5;
  // LDS operand reads of k-pair kk (MFMA 32x32x2: lane holds A[m = li][k = lh], B[k = lh][n = li])
  auto read_ops = [&](int kk, float(&a)[TM], float(&b)[2]) {
    const int krow = kk + lh;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = wr * (TM * 32) + i * 32 + li;
      a[i] = AM ? As[krow * LDM + m] : As[m * LDK + krow];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = wc * 64 + j * 32 + li;
      b[j] = BN_ ? Bs[krow * LDN + n] : Bs[n * LDK + krow];
    }
  };
  for (int step = 0; step < nsteps; ++step) {
    if (step + 1 < nsteps) {                         // global loads of the next K-step fly under the MFMAs
      if (++kidx == ksteps) {
        kidx = 0; ++ii;
        Ap = base_a(ii); Bp = base_b(ii);
      } else {
        Ap += (long)BK * g.a_sk; Bp += (long)BK * g.b_sk;
      }
      load_regs(Ap, Bp, kend - kbeg - kidx * BK);
    }
    if constexpr (MODE == 2) {
#pragma unroll
      for (int ks = 0; ks < BK; ks += 16) {
        bf16x8 ah[3][TM], bh[3][2];
        // transposed fragment read of a [k][row] image: each 16-lane group fetches a 4 (k) x 16 (rows) block,
        // lane 4q+p of the group supplies the address of block row q, columns 4p .. 4p+3, and receives the 4 k
        // of row (lane & 15); two reads give the 8 consecutive k (8 lh .. 8 lh + 7) of MFMA row li
        auto read_tr = [&](const short* img, int ld, int row0) -> bf16x8 {
          const short* ptr = img + (ks + 8 * lh + ((lane & 15) >> 2)) * ld + row0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(const_cast<short*>(ptr)));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(const_cast<short*>(ptr + 4 * ld)));
          return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            if constexpr (TRA) ah[p][i] = read_tr(Ah + p * AIMG, LDTA, wr * (TM * 32) + i * 32);
            else ah[p][i] = *reinterpret_cast<const bf16x8*>(&Ah[p * AIMG + (wr * (TM * 32) + i * 32 + li) * LDR + ks + 8 * lh]);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if constexpr (TRB) bh[p][j] = read_tr(Bh + p * BIMG, LDTB, wc * 64 + j * 32);
            else bh[p][j] = *reinterpret_cast<const bf16x8*>(&Bh[p * BIMG + (wc * 64 + j * 32 + li) * LDR + ks + 8 * lh]);
          }
        }
        // smallest terms first: lo*hi, hi*lo, mid*mid, then mid*hi, hi*mid, then hi*hi
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[PA[t]][i], bh[PB[t]][j], acc[i][j], 0, 0, 0);
      }
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int ks = 0; ks < BK; ks += 16) {
        bf16x8 ah[TM], bh[2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          ah[i] = *reinterpret_cast<const bf16x8*>(&Ah[(wr * (TM * 32) + i * 32 + li) * LDR + ks + 8 * lh]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          bh[j] = *reinterpret_cast<const bf16x8*>(&Bh[(wc * 64 + j * 32 + li) * LDR + ks + 8 * lh]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
    } else {
    float a0[TM], b0[2], a1[TM], b1[2];              // operand double buffer: reads one k-pair ahead
    read_ops(0, a0, b0);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      read_ops(kk + 2, a1, b1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
      if (kk + 4 < BK) read_ops(kk + 4, a0, b0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
    }
    // pin the schedule: LDS operand reads run one k-pair ahead of the MFMAs that consume them
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + 2), 0);
#pragma unroll
    for (int p = 0; p < BK / 2 - 2; ++p) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, TM + 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM, 0);
    }
    __syncthreads();
    if (step + 1 < nsteps) {
      store_step();
      __syncthreads();
    }
  }

  float* Cb = g.c_ptrs[0] ? g.c_ptrs[z & 7] : g.C + (long)z * g.c_sz;
  const float* Cinb = g.cin_ptrs[0] ? g.cin_ptrs[z & 7] : (g.Cin ? g.Cin + (long)z * g.cin_sz : nullptr);
  // per-thread column terms once (2 columns), row terms once per row (16 TM rows): an element costs an add, the
  // optional terms and its store
  long ccol[2], cincol[2];
  float bn[2];
  bool cok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wc * 64 + j * 32 + li;
    cok[j] = col < g.N;
    bn[j] = (g.bias_n && cok[j]) ? g.bias_n[col] : 0.f;
    ccol[j] = (long)col * g.c_sn;
    cincol[j] = (long)col * g.cin_sn;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wr * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row >= g.M) continue;
      float* crow = Cb + row_off(row, g.c_sm, g.c_mdiv, g.c_sdiv);
      const float* cinrow = Cinb ? Cinb + row_off(row, g.cin_sm, g.cin_mdiv, g.cin_sdiv) : nullptr;
      const float bm = g.bias_m ? g.bias_m[row] : 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (!cok[j]) continue;
        float v = (acc[i][j][r] + bn[j] + bm) * g.oscale;
        if (cinrow) v += g.beta * cinrow[cincol[j]];
        if (g.act == 1) v = tanh_outlined(v);
        crow[ccol[j]] = v;
      }
    }
}


// ---- bf16-MFMA projection GEMM (fp32 in HBM, rounded to bf16 while staging, fp32 accumulate/out) -----
// C(m,n) = sum_k bf16(A(m,k)) * bf16(B(k,n)) + bias_n[n]; B must be contiguous along k (nn.Linear weight
// [n][k]); A is contiguous along k (AM = false) or along m (AM = true: channel-major V with the row split).
// v_mfma_f32_32x32x16_bf16: lane (i = lane&31, h = lane>>5) holds 8 consecutive k of row i -> both
// operands are staged [row][k] (k contiguous, 80-byte rows) and read with one ds_read_b128 per MFMA.
template <bool AM, bool AVEC>
__global__ __launch_bounds__(256) void gemm_bf16in_kernel(const GemmK g) {
  constexpr int BM = 128, BN = 128, BK = 32, LDR = 40;       // row stride in bf16 elements (80 B)
  __shared__ __attribute__((aligned(16))) short As[BM * LDR];
  __shared__ __attribute__((aligned(16))) short Bs[BN * LDR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware tile order (speed only): workgroup ids i and i + 8 share an XCD and its L2, so the
  // n-tiles of one m-tile (they read the same A panel) get ids equal mod 8 when there are many m-tiles.
  int m0, n0, z;
  {
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    const int id = blockIdx.x, x = id & 7, slot = id >> 3;
    if (!g.xcd_group) {                              // plain order: few m-tiles, nothing to group
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
  const int nsteps = (g.K + BK - 1) / BK;
  const float* Ab = g.a_ptrs[0] ? g.a_ptrs[z & 7] : g.A + (long)z * g.a_sz;
  const float* Bb = g.B + (long)z * g.b_sz;
  int a_off[4], a_m[4], a_k[4];
  bool a_ok[4];
  int a_off1[4][4];                                          // per-element offsets when rows are not 16-byte aligned
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    if (AM) { a_k[i] = idx >> 5; a_m[i] = (idx & 31) * 4; } else { a_m[i] = idx >> 3; a_k[i] = (idx & 7) * 4; }
    a_ok[i] = (m0 + a_m[i]) < g.M;
    a_off[i] = a_ok[i] ? (int)(row_off(m0 + a_m[i], g.a_sm, g.a_mdiv, g.a_sdiv) + (long)a_k[i] * g.a_sk) : 0;
    if (AM && !AVEC) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        a_off1[i][e] = (m0 + a_m[i] + e) < g.M
                           ? (int)(row_off(m0 + a_m[i] + e, g.a_sm, g.a_mdiv, g.a_sdiv) + (long)a_k[i] * g.a_sk) : -1;
    }
  }
  int b_off[4], b_n[4], b_k[4];
  bool b_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    b_n[i] = idx >> 3; b_k[i] = (idx & 7) * 4;
    b_ok[i] = (n0 + b_n[i]) < g.N;
    b_off[i] = b_ok[i] ? (int)((long)b_k[i] * g.b_sk + (long)(n0 + b_n[i]) * g.b_sn) : 0;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x4 ra[4], rb[4];
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_regs = [&](int k0) {
    const int klim = g.K - k0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (AM && !AVEC) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          ra[i][e] = (a_off1[i][e] >= 0 && a_k[i] < klim) ? Ab[(long)k0 * g.a_sk + a_off1[i][e]] : 0.f;
      } else {
        ra[i] = (a_ok[i] && a_k[i] < klim) ? *reinterpret_cast<const f32x4*>(Ab + (long)k0 * g.a_sk + a_off[i]) : zero4;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      rb[i] = (b_ok[i] && b_k[i] < klim) ? *reinterpret_cast<const f32x4*>(Bb + (long)k0 * g.b_sk + b_off[i]) : zero4;
  };
  auto store_regs = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (AM) {                                              // 4 rows m..m+3 of one k: transposing 2-byte writes
#pragma unroll
        for (int e = 0; e < 4; ++e) As[(a_m[i] + e) * LDR + a_k[i]] = f2bf(ra[i][e]);
      } else {
        *reinterpret_cast<bf16x4*>(&As[a_m[i] * LDR + a_k[i]]) =
            bf16x4{f2bf(ra[i][0]), f2bf(ra[i][1]), f2bf(ra[i][2]), f2bf(ra[i][3])};
      }
      *reinterpret_cast<bf16x4*>(&Bs[b_n[i] * LDR + b_k[i]]) =
          bf16x4{f2bf(rb[i][0]), f2bf(rb[i][1]), f2bf(rb[i][2]), f2bf(rb[i][3])};
    }
  };
  if (nsteps > 0) { load_regs(0); store_regs(); }
  __syncthreads();
  const int li = lane & 31, lh = lane >> 5;
  for (int step = 0; step < nsteps; ++step) {
    if (step + 1 < nsteps) load_regs((step + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK; ks += 16) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const bf16x8*>(&As[(wr * 64 + i * 32 + li) * LDR + ks + 8 * lh]);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(&Bs[(wc * 64 + j * 32 + li) * LDR + ks + 8 * lh]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (step + 1 < nsteps) {
      store_regs();
      __syncthreads();
    }
  }
  float* Cb = g.C + (long)z * g.c_sz;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wc * 64 + j * 32 + li;
      const float bn = (g.bias_n && col < g.N) ? g.bias_n[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < g.M && col < g.N) Cb[row_off(row, g.c_sm, g.c_mdiv, g.c_sdiv) + (long)col * g.c_sn] = (acc[i][j][r] + bn) * g.oscale;
      }
    }
}

}  // namespace

static long span(long n, long s) { return n > 0 ? (n - 1) * (s < 0 ? -s : s) : 0; }

// bf16: 0 = fp32 MFMA; 1 = bf16-input MFMA, 2 = 3-way bf16 split (fp32-accurate) on the aligned fast path:
// these two return 1 (nothing launched) when the shape is not eligible
static int launch_gemm_impl(const coattn_gemm_desc& d, hipStream_t s, int bf16) {
  CA_CHECK_ARG((d.A || d.a_ptrs[0]) && (d.B || d.b_ptrs[0]) && (d.C || d.c_ptrs[0]), "gemm: null operand");
  CA_CHECK_ARG(!(d.ptr_by_inner && (d.c_ptrs[0] || d.cin_ptrs[0])), "gemm: C tables are indexed by batch only");
  CA_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0 && d.batch > 0, "gemm: bad shape M=%d N=%d K=%d batch=%d", d.M, d.N, d.K, d.batch);
  CA_CHECK_ARG(d.batch <= 65535, "gemm: batch %d exceeds grid.z", d.batch);
  GemmK g;
  g.A = (const float*)d.A; g.B = (const float*)d.B; g.Cin = (const float*)d.Cin; g.C = (float*)d.C;
  g.bias_n = (const float*)d.bias_n; g.bias_m = (const float*)d.bias_m;
  g.M = d.M; g.N = d.N; g.K = d.K; g.inner = d.inner > 0 ? d.inner : 1; g.inner_total = d.inner_total;
  g.ksplit = d.ksplit; g.act = d.act; g.beta = d.beta; g.oscale = d.out_scale != 0.f ? d.out_scale : 1.f;
  g.a_sm = d.a_sm; g.a_sk = d.a_sk; g.a_sz = d.a_sz; g.a_si = d.a_si; g.a_mdiv = d.a_mdiv; g.a_sdiv = d.a_sdiv;
  g.b_sk = d.b_sk; g.b_sn = d.b_sn; g.b_sz = d.b_sz; g.b_si = d.b_si;
  g.c_sm = d.c_sm; g.c_sn = d.c_sn; g.c_sz = d.c_sz; g.c_mdiv = d.c_mdiv; g.c_sdiv = d.c_sdiv;
  g.cin_sm = d.cin_sm; g.cin_sn = d.cin_sn; g.cin_sz = d.cin_sz; g.cin_mdiv = d.cin_mdiv; g.cin_sdiv = d.cin_sdiv;
  for (int t = 0; t < 8; ++t) {
    g.a_ptrs[t] = (const float*)d.a_ptrs[t]; g.b_ptrs[t] = (const float*)d.b_ptrs[t];
    g.c_ptrs[t] = (float*)d.c_ptrs[t]; g.cin_ptrs[t] = (const float*)d.cin_ptrs[t];
  }
  g.ptr_by_inner = d.ptr_by_inner; g.b_imod = d.b_imod;
  g.kband_n = d.kband_n;
  if (d.kband_n > 0) {
    CA_CHECK_ARG(d.kband_n % 128 == 0 && (d.N + d.kband_n - 1) / d.kband_n <= 3, "gemm: kband_n must be a multiple of 128 with at most 3 bands");
    for (int t = 0; t < 3; ++t) {
      CA_CHECK_ARG(d.kband_lo[t] >= 0 && d.kband_hi[t] <= d.K && (d.kband_lo[t] & 3) == 0 && (d.kband_hi[t] & 3) == 0,
                   "gemm: k bands must lie in [0,K] and be multiples of 4");
      g.kband_lo[t] = d.kband_lo[t]; g.kband_hi[t] = d.kband_hi[t];
    }
  }
  CA_CHECK_ARG(!(d.a_ptrs[0] || d.b_ptrs[0] || d.c_ptrs[0] || d.cin_ptrs[0]) ||
                   (d.ptr_by_inner ? (d.inner_total > 0 ? d.inner_total : d.inner) : d.batch) <= 8,
               "gemm: pointer tables hold at most 8 entries");
  // the kernel keeps tile-relative element offsets in 32 bits
  const long a_rows = d.a_mdiv > 0 ? span((d.M + d.a_mdiv - 1) / d.a_mdiv + 1, d.a_sdiv) + span(d.a_mdiv, d.a_sm)
                                   : span(d.M, d.a_sm);
  CA_CHECK_ARG(a_rows + span(16, d.a_sk) < 2147483647L && span(d.N, d.b_sn) + span(16, d.b_sk) < 2147483647L,
               "gemm: operand extent exceeds 32-bit element offsets");
  // global-load thread mapping: run consecutive threads along the contiguous operand axis
  g.a_mfast = (d.a_sm == 1 && d.a_sk != 1) ? 1 : 0;
  g.b_nfast = (d.b_sn == 1) ? 1 : 0;
  const bool small_m = d.M <= 64;
  dim3 block(256);
  // aligned fast path: every float4 the kernel forms must be 16-byte aligned and lie inside one row
  auto al4 = [](long v) { return (v & 3) == 0; };
  auto pal = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  const bool a_m = g.a_mfast != 0, b_n = g.b_nfast != 0;
  bool vec = !small_m && d.M >= 128 && (d.a_sk == 1 || d.a_sm == 1) && (d.b_sk == 1 || d.b_sn == 1);
  if (vec) {
    // A: contiguous axis m (a_m) or k; the other strides / offsets must keep 16-byte alignment
    vec = vec && (a_m ? (al4(d.M) && al4(d.a_sk) && al4(d.a_mdiv) && al4(d.a_sdiv)) : (al4(d.a_sm) && al4(d.a_sdiv)));
    vec = vec && (b_n ? (al4(d.N) && al4(d.b_sk)) : al4(d.b_sn));
    vec = vec && al4(d.K) && al4(d.ksplit) && al4(d.a_sz) && al4(d.a_si) && al4(d.b_sz) && al4(d.b_si);
    vec = vec && (d.a_ptrs[0] ? true : pal(d.A)) && (d.b_ptrs[0] ? true : pal(d.B));
    for (int t = 0; t < 8; ++t) vec = vec && pal(d.a_ptrs[t]) && pal(d.b_ptrs[t]);
  }
  if (bf16) {
    if (!vec) return 1;
    const long ntn = (d.N + 127) / 128, ntm = (d.M + 127) / 128;
    g.xcd_group = ntm >= 32 ? 1 : 0;
    const long nblk = g.xcd_group ? (long)d.batch * ntn * ((ntm + 7) / 8) * 8 : (long)d.batch * ntn * ntm;
    CA_CHECK_ARG(nblk < 2147483647L, "gemm: grid too large");
    dim3 grid((unsigned)nblk);
    if (bf16 == 2) {
      if (a_m && b_n) hipLaunchKernelGGL((gemm_f32_vec_kernel<true, true, 128, 2>), grid, block, 0, s, g);
      else if (a_m) hipLaunchKernelGGL((gemm_f32_vec_kernel<true, false, 128, 2>), grid, block, 0, s, g);
      else if (b_n) hipLaunchKernelGGL((gemm_f32_vec_kernel<false, true, 128, 2>), grid, block, 0, s, g);
      else hipLaunchKernelGGL((gemm_f32_vec_kernel<false, false, 128, 2>), grid, block, 0, s, g);
      CA_CHECK_LAUNCH("gemm_bf16x3_vec");
      return 0;
    }
    if (a_m && b_n) hipLaunchKernelGGL((gemm_f32_vec_kernel<true, true, 128, 1>), grid, block, 0, s, g);
    else if (a_m) hipLaunchKernelGGL((gemm_f32_vec_kernel<true, false, 128, 1>), grid, block, 0, s, g);
    else if (b_n) hipLaunchKernelGGL((gemm_f32_vec_kernel<false, true, 128, 1>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_vec_kernel<false, false, 128, 1>), grid, block, 0, s, g);
    CA_CHECK_LAUNCH("gemm_bf16_vec");
    return 0;
  }
  if (vec) {
    // Tile height: 128 rows, or 64 rows when 128-row tiles would leave a mostly empty last round of
    // workgroups (P_v at B=160, N=196: 980 tiles for 768 resident -> 1,960 half-size tiles at 5/CU).
    // (192-row tiles were also tried for that case and measured slower: 279 vs 235 us.)
    const long ntn = (d.N + 127) / 128;
    const long wg128 = ntn * ((d.M + 127) / 128) * d.batch;
    static const int force_bm = dev_env_int("COATTN_GEMM_BM", 0);   // developer switch
    const bool small_tiles = force_bm ? force_bm == 64 : (wg128 > 768 && wg128 < 3 * 768);
    const int bm = small_tiles ? 64 : 128;
    const long ntm = (d.M + bm - 1) / bm;
    g.xcd_group = ntm >= 32 ? 1 : 0;                 // (grouping the tiles of a split index instead measured slower)
    const long nblk = g.xcd_group ? (long)d.batch * ntn * ((ntm + 7) / 8) * 8 : (long)d.batch * ntn * ntm;
    CA_CHECK_ARG(nblk < 2147483647L, "gemm: grid too large");
    dim3 grid((unsigned)nblk);
    if (small_tiles) {
      if (a_m && b_n) hipLaunchKernelGGL((gemm_f32_vec_kernel<true, true, 64>), grid, block, 0, s, g);
      else if (a_m) hipLaunchKernelGGL((gemm_f32_vec_kernel<true, false, 64>), grid, block, 0, s, g);
      else if (b_n) hipLaunchKernelGGL((gemm_f32_vec_kernel<false, true, 64>), grid, block, 0, s, g);
      else hipLaunchKernelGGL((gemm_f32_vec_kernel<false, false, 64>), grid, block, 0, s, g);
    } else {
      if (a_m && b_n) hipLaunchKernelGGL((gemm_f32_vec_kernel<true, true, 128>), grid, block, 0, s, g);
      else if (a_m) hipLaunchKernelGGL((gemm_f32_vec_kernel<true, false, 128>), grid, block, 0, s, g);
      else if (b_n) hipLaunchKernelGGL((gemm_f32_vec_kernel<false, true, 128>), grid, block, 0, s, g);
      else hipLaunchKernelGGL((gemm_f32_vec_kernel<false, false, 128>), grid, block, 0, s, g);
    }
    CA_CHECK_LAUNCH("gemm_f32_vec");
    return 0;
  }
  if (small_m) {
    dim3 grid((d.N + 127) / 128, (d.M + 31) / 32, d.batch);
    hipLaunchKernelGGL(gemm_f32_kernel<32>, grid, block, 0, s, g);
  } else {
    dim3 grid((d.N + 127) / 128, (d.M + 127) / 128, d.batch);
    hipLaunchKernelGGL(gemm_f32_kernel<128>, grid, block, 0, s, g);
  }
  CA_CHECK_LAUNCH("gemm_f32");
  return 0;
}

// fp32 GEMM: the fp32-accurate 3-way bf16 split (MODE 2) on the bf16 MFMA for every shape the aligned fast path
// takes, the f32 MFMA otherwise.  COATTN_GEMM_X3: 0 = never, 1 = auto (default) = whenever eligible.
// With the [k][row] bf16 images + transposed LDS reads for operands contiguous along their tile index the
// split is the faster mode on all layouts of the path (P_v 188 -> 128 us, dW_v 172 -> 124, dQ projection
// 87 -> 60, dW_q 73 -> 57, P_q 87 -> 62); before that it lost on transposing operands (P_v 203 vs 182 us).
int launch_gemm_f32(const coattn_gemm_desc& d, hipStream_t s) {
  static const int x3 = dev_env_int("COATTN_GEMM_X3", 1);
  if (x3 >= 1) {
    const int rc = launch_gemm_impl(d, s, 2);            // 0 launched, < 0 error, 1 not eligible
    if (rc <= 0) return rc;
  }
  return launch_gemm_impl(d, s, 0);
}

// GEMM with bf16 MFMA inputs (fp32 storage, fp32 accumulate / output): every aligned shape through the
// bf16 mode of the fast-path kernel; the unaligned channel-major projection (P_v at N = 49) through
// gemm_bf16in_kernel; anything else in exact fp32.
int launch_gemm_bf16in(const coattn_gemm_desc& d, hipStream_t s) {
  {
    const int rc = launch_gemm_impl(d, s, 1);
    if (rc <= 0) return rc;                              // launched (0) or argument error (< 0)
  }
  auto al4 = [](long v) { return (v & 3) == 0; };
  const bool a_m = (d.a_sm == 1 && d.a_sk != 1);
  bool tab_ok = true;
  for (int t = 0; t < 8; ++t) tab_ok = tab_ok && ((((uintptr_t)d.a_ptrs[t]) & 15) == 0);
  const bool ok = (d.A || d.a_ptrs[0]) && tab_ok && (!d.a_ptrs[0] || d.batch <= 8) && d.B && d.C && !d.Cin && !d.bias_m && d.act == 0 && d.inner <= 1 && d.ksplit == 0 &&
                  !d.ptr_by_inner && !d.b_ptrs[0] && !d.c_ptrs[0] && d.b_sk == 1 && al4(d.b_sn) && al4(d.K) &&
                  (a_m ? true : (d.a_sk == 1 && al4(d.a_sm))) &&
                  al4(d.a_sz) && al4(d.b_sz) && ((((uintptr_t)d.A) | ((uintptr_t)d.B)) & 15) == 0;
  if (!ok) return launch_gemm_f32(d, s);               // shapes the bf16 kernel does not take: exact fp32 path
  GemmK g = {};
  g.A = (const float*)d.A; g.B = (const float*)d.B; g.C = (float*)d.C; g.bias_n = (const float*)d.bias_n;
  for (int t = 0; t < 8; ++t) g.a_ptrs[t] = (const float*)d.a_ptrs[t];
  g.M = d.M; g.N = d.N; g.K = d.K; g.oscale = d.out_scale != 0.f ? d.out_scale : 1.f;
  g.a_sm = d.a_sm; g.a_sk = d.a_sk; g.a_sz = d.a_sz; g.a_mdiv = d.a_mdiv; g.a_sdiv = d.a_sdiv;
  g.b_sk = d.b_sk; g.b_sn = d.b_sn; g.b_sz = d.b_sz;
  g.c_sm = d.c_sm; g.c_sn = d.c_sn; g.c_sz = d.c_sz; g.c_mdiv = d.c_mdiv; g.c_sdiv = d.c_sdiv;
  const long ntn = (d.N + 127) / 128, ntm = (d.M + 127) / 128;
  g.xcd_group = ntm >= 32 ? 1 : 0;
  dim3 grid((unsigned)(g.xcd_group ? (long)d.batch * ntn * ((ntm + 7) / 8) * 8 : (long)d.batch * ntn * ntm)), block(256);
  const bool a_vec = al4(d.M) && al4(d.a_sk) && al4(d.a_mdiv) && al4(d.a_sdiv);
  if (a_m && a_vec) hipLaunchKernelGGL((gemm_bf16in_kernel<true, true>), grid, block, 0, s, g);
  else if (a_m) hipLaunchKernelGGL((gemm_bf16in_kernel<true, false>), grid, block, 0, s, g);
  else hipLaunchKernelGGL((gemm_bf16in_kernel<false, true>), grid, block, 0, s, g);
  CA_CHECK_LAUNCH("gemm_bf16in");
  return 0;
}
