// Fused co-attention kernels (coattn_fused.hip): entry points used by api.hip.
#pragma once
#include "common.h"

// Element strides of the logical image-feature view x_img[B,N,d] (model.py:215-217).  The fused kernels take two
// physical layouts: channel-major [B][d][N] (sN = 1, sD = N: what the reference's NCHW encoder leaves behind its
// permuted view) and location-major [B][N][d] (sD = 1, sN = d: a channels_last encoder); the general-shape
// kernels take any strides.
struct VLayout {
  long sB, sN, sD;
};
inline bool v_is_lm(const VLayout& v, int N, int d) { (void)N; return v.sD == 1 && v.sN == d; }
inline bool v_is_cm(const VLayout& v, int N, int d) { (void)d; return v.sN == 1 && v.sD == N; }

int fused_supported(int B, int N, int T, int d, int L);
inline bool fused_layout_ok(const VLayout& v, int N, int d) { return v_is_lm(v, N, d) || v_is_cm(v, N, d); }
// everything after the projections (P_v, P_q already in `saved`)
int fused_attention_forward(int B, int N, int T, int d, int L, const float* V, const VLayout& vl,
                            const float* const* Q, const coattn_params* p, float* v_out, float* q_out, float* saved,
                            float* ws, hipStream_t s, int bf16 = 0, int np = 3);   // bf16: reduced precision, one MFMA per product; np: width of phase 2
int fused_backward_supported(int B, int N, int T, int d, int L);
int fused_backward(int B, int N, int T, int d, int L, const float* V, const VLayout& vl, const float* const* Q,
                   const coattn_params* p, const float* saved, const float* gv, const float* gq, float* dV,
                   const VLayout& dvl, float* const* dQ, const coattn_param_grads* pg, int accumulate, float* ws,
                   hipStream_t s, int bf16_proj, int wgemm, int np = 3, int live_rows = 0);   // wgemm: gemm_w / gemm_tn enabled; np: width of the contractions (3 | 2); live_rows: `saved` holds the forward's bitmap of the non-zero question rows

// Diagnostic build only (tools/probe_stamps.py, -DCOATTN_STAMPS=1): wave 0 of every workgroup writes the
// 100 MHz constant clock at its phase boundaries into the (otherwise unused) forward workspace tail.
#ifndef COATTN_STAMPS
#define COATTN_STAMPS 0
#endif
#if COATTN_STAMPS
#define CA_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 64 + (k)] = __builtin_amdgcn_s_memrealtime();  \
  } while (0)
#define CA_STAMP_CYC(k)                                                                               \
  do {                                                                                                \
    if (threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 64 + (k)] = __builtin_amdgcn_s_memtime();      \
  } while (0)
#else
#define CA_STAMP(k)
#define CA_STAMP_CYC(k)
#endif


// ---- shared by the fused forward / backward translation units -------------------------------
// The fused path keeps the projections P_v, P_q in `saved` MULTIPLIED by 2 log2(e): every consumer needs them only
// inside tanh(P_v + C^T P_q) / tanh(P_q + C P_v) = 1 - 2 / (1 + 2^(scaled argument)), so the scale -- applied once by
// the projection GEMMs' epilogue -- saves the multiplication in front of every exponential of the forward and the
// backward; the two places that need P itself (the dC terms of bwd_dc_kernel) divide it out of their 16 operands.
// The general-shape path keeps plain P_v, P_q.
constexpr float kPScale = 2.8853900817779268f;
constexpr int kTS = 7;       // k-steps of 4 over T: T <= 28
constexpr int kTRows = 28;   // rows of C kept in LDS
constexpr int kSlotRows = 32; // rows of a cross-wave reduction slot (both 16-row MFMA tiles)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));      // bf16 MFMA operand: 8 k-values per lane
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));

// ---- fp32-accurate products on the bf16 MFMA: exact 3-way split x = hi + mid + lo ----------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bfv2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {      // one v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{a, b}), bfv2));
}
// a - b.  The residual subtractions must stay single v_sub_f32 (paired into v_pk_add_f32 they cost more issue
// cycles beside MFMAs than two plain subtractions): the kernels are compiled with -fno-slp-vectorize; an inline-asm
// v_sub_f32 would do the same but costs a wait state after every statement.
__device__ __forceinline__ float sub1(float a, float b) { return a - b; }
// two FP16 pieces of a pair (6 VALU: v_cvt_pk_f16_f32, two v_cvt_f32_f16, two v_sub_f32, v_cvt_pk_f16_f32; round to nearest even)
typedef _Float16 hfv2 __attribute__((ext_vector_type(2)));
typedef _Float16 hfv8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split_pair_h(float a, float b, unsigned& h, unsigned& m) {
  const hfv2 hh = __builtin_convertvector((f32x2{a, b}), hfv2);
  h = __builtin_bit_cast(unsigned, hh);
  const float ra = sub1(a, (float)hh[0]);
  const float rb = sub1(b, (float)hh[1]);
  m = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{ra, rb}), hfv2));
}
// one pair of fp32 values -> the three packed bf16 pairs (11 VALU ops)
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk_bf16(a, b);
  const float ra = sub1(a, __builtin_bit_cast(float, h << 16));
  const float rb = sub1(b, __builtin_bit_cast(float, h & 0xffff0000u));
  m = cvt_pk_bf16(ra, rb);
  const float sa = sub1(ra, __builtin_bit_cast(float, m << 16));
  const float sb = sub1(rb, __builtin_bit_cast(float, m & 0xffff0000u));
  l = cvt_pk_bf16(sa, sb);
}
__device__ __forceinline__ void split3(const f32x8& v, bf16x8 (&p)[3]) {
  u32x4 h, m, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned hh, mm, ll;
    split3_pair(v[2 * i], v[2 * i + 1], hh, mm, ll);
    h[i] = hh; m[i] = mm; l[i] = ll;
  }
  p[0] = __builtin_bit_cast(bf16x8, h);
  p[1] = __builtin_bit_cast(bf16x8, m);
  p[2] = __builtin_bit_cast(bf16x8, l);
}
// The width of a contraction = bf16 pieces kept per operand (NP):
//   3  the exact split: six partial products down to relative order 2^-16, fp32-accurate (the affinity A = Q V^T);
//   2  hi + mid: 16 significand bits per operand, the three products hi*mid, mid*hi, hi*hi (~2^-16 relative per product,
//      random in sign over a contraction) -- every contraction whose error the path does not amplify: 6 VALU operations
//      per pair instead of 11, half the MFMAs (budget: tests/test_split_emulation.py);
//   1  the reduced-precision mode (COATTN_FLAG_BF16_PROJ): the hi x hi product alone.
// Pieces that a width does not use are neither computed, stored nor read.
template <int NP>
__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  static_assert(NP >= 1 && NP <= 3, "pieces per operand");
  if constexpr (NP == 3) { split3_pair(a, b, h, m, l); return; }
  h = cvt_pk_bf16(a, b);
  if constexpr (NP == 2) {
    const float ra = sub1(a, __builtin_bit_cast(float, h << 16));
    const float rb = sub1(b, __builtin_bit_cast(float, h & 0xffff0000u));
    m = cvt_pk_bf16(ra, rb);
  }
}
template <int NP>
__device__ __forceinline__ void splitn(const f32x8& v, bf16x8 (&p)[3]) {
  u32x4 h, m, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned hh = 0, mm = 0, ll = 0;
    split_pair<NP>(v[2 * i], v[2 * i + 1], hh, mm, ll);
    h[i] = hh; m[i] = mm; l[i] = ll;
  }
  p[0] = __builtin_bit_cast(bf16x8, h);
  if constexpr (NP >= 2) p[1] = __builtin_bit_cast(bf16x8, m);
  if constexpr (NP == 3) p[2] = __builtin_bit_cast(bf16x8, l);
}
// the same with the pieces' format as a parameter: H = two FP16 pieces (NP = 2), else NP bf16 pieces
template <int NP, bool H>
__device__ __forceinline__ void split_pair_x(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  if constexpr (H) {
    static_assert(NP == 2, "FP16 pieces: two per operand");
    split_pair_h(a, b, h, m);
  } else {
    split_pair<NP>(a, b, h, m, l);
  }
}
template <int NP, bool H>
__device__ __forceinline__ void splitn_x(const f32x8& v, bf16x8 (&p)[3]) {
  if constexpr (H) {
    u32x4 h, m;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned hh, mm;
      split_pair_h(v[2 * i], v[2 * i + 1], hh, mm);
      h[i] = hh; m[i] = mm;
    }
    p[0] = __builtin_bit_cast(bf16x8, h);
    p[1] = __builtin_bit_cast(bf16x8, m);
  } else {
    splitn<NP>(v, p);
  }
}
// one 32x32x16 MFMA on 16-bit operand registers of either format
template <bool H>
__device__ __forceinline__ f32x16 mfma32_16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
  if constexpr (H) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hfv8, a), __builtin_bit_cast(hfv8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// partial products of a width in the order they are issued (smallest first)
template <int NP> __device__ __forceinline__ constexpr int n_products() { return NP == 3 ? 6 : NP == 2 ? 3 : 1; }
template <int NP> __device__ __forceinline__ constexpr int piece_a(int k) {      // A-operand piece of product k
  constexpr int A3[6] = {0, 2, 1, 0, 1, 0}, A2[3] = {0, 1, 0};
  return NP == 3 ? A3[k] : NP == 2 ? A2[k] : 0;
}
template <int NP> __device__ __forceinline__ constexpr int piece_b(int k) {      // B-operand piece of product k
  constexpr int B3[6] = {2, 0, 1, 1, 0, 0}, B2[3] = {1, 0, 0};
  return NP == 3 ? B3[k] : NP == 2 ? B2[k] : 0;
}
// Slot i (0 .. 5) of a six-slot MFMA group under width NP: the index of the product issued there, or -1.
// NP = 3: every slot; NP = 2: slots 1, 3, 5 (the VALU chunks of the empty slots keep their places); NP = 1: slot 5.
template <int NP>
__device__ __forceinline__ constexpr int slot_product(int i) {
  return NP == 3 ? i : NP == 2 ? ((i & 1) ? i >> 1 : -1) : (i == 5 ? 0 : -1);
}
// c += a . b over 16 k (32x32x16) at width NP (each bf16 x bf16 product is exact in the fp32 accumulator)
template <int NP>
__device__ __forceinline__ f32x16 mfma32_xn(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 c) {
#pragma unroll
  for (int k = 0; k < n_products<NP>(); ++k)
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[piece_a<NP>(k)], b[piece_b<NP>(k)], c, 0, 0, 0);
  return c;
}

// buffer resource over [ptr, ptr + bytes): out-of-range loads return 0, stores are dropped
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// 16-byte buffer store.  The scalar offset is folded into the vector offset ON PURPOSE: a buffer store of more than
// 64 bits reads its data registers late, and hipcc (ROCm 7.2) pads the following overwrite of those registers only
// when the store has NO scalar-offset register -- with one, a VALU write scheduled right behind the store corrupted
// the stored data on gfx950 (dP_v of the fused backward: wrong by O(1) in the elements the next v_pk_add reused).
__device__ __forceinline__ void buf_store4(f32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff + soff, 0, 0);
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// ---- GEMM against a pre-split weight (gemm_w.hip) ---------------------------------------------------------------
struct WSplit { const float* W; void* out; int N, K, trans, ld; int pieces; float* amax; };   // Bw(k,n) = trans ? W[k*ld+n] : W[n*ld+k]; pieces: 3 (0 = 3), 1 = hi piece only (gemm_bf.hip),
                                                                                 // 16 = two FP16 pieces of kF16WScale * W (WGemm.f16);
                                                                                 // amax (pieces = 16; may be NULL): one word per wave
                                                                                 // = per (32-column tile, 16-k step) chunk, max |256 W|
// Status words of the tolerance mode (COATTN_FLAG_FAST16; coattn_status / coattn_phrase_status), floats:
//   [0]  written 0 by the weight-split launch, then raised (atomicMax on the bit pattern) by any wave of the projection launch
//        that converted an activation beyond kF16Exact to FP16 pieces: the largest such |x|;
//   [1]  1.0 if the call that wrote the words used FP16 pieces, else 0 (written by the weight-split launch);
//   [kStatusHdr + job * chunks + chunk]  max |kF16WScale * W| per chunk of weight-split job 0 / 1 (FP16 images only).
constexpr int kStatusHdr = 64;
constexpr float kF16Exact = 65504.f;
// synchronises `s`, reads the words back: 0, or -4 with the error message set (api.hip); amax (host, may be NULL): [0] activations, [1] weights
int read_status_words(const float* status, int n_words, hipStream_t s, float* amax, const char* what);
// asynchronous: folds the words into acc[0] (activations), acc[1] (weights) by bit-pattern maxima (api.hip)
int fold_status_words(const float* status, int n_words, float* acc, hipStream_t s, const char* what);
inline size_t status_floats(int N, int K, int nweights) { return kStatusHdr + (size_t)nweights * ((N + 31) / 32) * ((K + 15) / 16); }
// Forward-side contractions on two FP16 pieces (x = hi + lo, 22 significand bits; the three products lo*hi, hi*lo, hi*hi on
// v_mfma_f32_32x32x16_f16): the cost of the two-piece bf16 width with 64 x less error -- for operands of ordinary magnitude
// only: features, projections, tanh values (|x| < 65,504; gfx950 keeps fp16 subnormals in conversions and in the MFMA, so
// small values lose precision only below 2^-24 absolute).  Gradients stay on bf16 pieces (fp32's range).  The weight image
// holds kF16WScale * W so that the lo pieces of ~0.04-sized weights are normal numbers; the GEMM divides it out.
constexpr float kF16WScale = 256.f;
// Kernels that split into FP16 pieces set MODE.FP16_OVFL first: a conversion that overflows fp16 then saturates at
// +-65,504 instead of becoming inf, so hi + lo represents magnitudes up to 131,008 (with fewer bits above 65,504) and larger
// ones clamp there -- finite results for any finite input, never inf - inf = NaN (probed on gfx950, tools/probe_f16_ovfl.hip).
__device__ __forceinline__ void f16_saturating_conversions() {
  __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);   // hwreg(HW_REG_MODE, offset 23, size 1) = FP16_OVFL
}
struct WGemm {
  const float* A; const float* a_ptrs[8]; long a_sz; int a_sm;       // A[z][m][k], k contiguous; z from the table or a_sz
  int a_sk, a_mdiv; long a_sdiv;                                     // a_sk != 0: A contiguous along m instead, element
                                                                     // (m, k) at (m / a_mdiv) * a_sdiv + m % a_mdiv + k * a_sk
  int kband_n, kband_lo[3], kband_hi[3];                             // kband_n > 0: column band j contracts over k in [lo_j, hi_j)
  const void* Wf;                                                    // wsplit image of Bw [K x N]
  float* C; float* c_ptrs[8]; long c_sz; int c_sm;                   // C[z][m][n], n contiguous
  const float* bias_n; float out_scale;
  int M, N, K, batch;
  int bf16;                                                          // 1: operands rounded to bf16, ONE MFMA per product (COATTN_FLAG_BF16_PROJ)
  int np;                                                            // (bf16 = 0) bf16 pieces per operand: 0 / 3 = the exact split (six products), 2 = hi + mid (three products)
  int f16;                                                           // (np = 2) the two pieces are FP16 (Wf: a pieces = 16 image)
  int a_bf16;                                                        // (gemm_bf_kernel) A is STORED as bf16; a_sm, a_sz stay in elements
  float* status;                                                     // (f16) status word [0] of the call (see kStatusHdr), or NULL
  const unsigned* rowbits;                                           // (gemm_w, row-major A, NULL = all rows) bitmap per batch entry z,
                                                                     // (M + 31) / 32 words each: bit m set = row m of A_z has a non-zero
                                                                     // element.  The GEMM then runs over the set rows alone (tiles of the
                                                                     // compacted row list); the rows whose bit is clear must already hold
                                                                     // (0 + bias) * out_scale -- RowFlagJob writes both
};
// Rows of exact zeros (the pad tokens of the question hierarchy: model.py:263 padding_idx, :292-296 pad_packed_sequence) project to
// the bias alone.  One job of the weight-split launch (its extra workgroups): per batch entry z and 32 rows one word of the bitmap
// WGemm.rowbits, and (0 + bias) * out_scale into the output rows whose input row is all zeros -- bit for bit what the dense
// product stores there.
struct RowFlagJob {
  const float* a_ptrs[8]; int a_sm;              // A_z[m][k], k contiguous
  float* C; long c_sz; int c_sm;                 // C_z[m][n]
  const float* bias_n; float out_scale;
  int M, N, K, batch;
  unsigned* rowbits;                             // [batch][(M + 31) / 32]
  unsigned* rowcnt;                              // (job inside a GEMM launch) [8] words finished per batch entry: zeroed by the launch
                                                 // before, raised by every workgroup of the job behind its word; NULL otherwise
};
constexpr int kRowBitsMaxWords = 512;            // per batch entry (M <= 16,384 rows): the GEMM tiles scan the words serially
inline size_t rowbits_words(int M, int batch) { return (size_t)batch * ((M + 31) / 32); }
size_t wsplit_bytes(int N, int K);
// status_hdr (may be NULL): the launch's first thread writes [0] = 0 and [1] = f16 ? 1 : 0 (the status words' header)
// rows (may be NULL): a RowFlagJob done by extra workgroups of the same launch; zero8 (may be NULL): eight words the launch zeroes
// (RowFlagJob.rowcnt of a job that rides in the NEXT launch)
int launch_wsplit(const WSplit* jobs, int njobs, hipStream_t s, float* status_hdr = nullptr, int f16 = 0, const RowFlagJob* rows = nullptr,
                  unsigned* zero8 = nullptr);
int gemm_w_supported(const WGemm& d);
// n = 1 or 2 GEMMs in one launch.  rows (may be NULL; exact four-wave launches only): a RowFlagJob whose workgroups come FIRST in
// the launch; the tiles of a job with WGemm.rowbits then wait (per batch entry) until rows->rowcnt says its words are written --
// the other job's tiles run meanwhile
int launch_gemm_w(const WGemm* d, int n, hipStream_t s, const RowFlagJob* rows = nullptr);
// single-product bf16 GEMM for wide shapes (gemm_bf.hip): reads a hi-piece-only weight image (WSplit.pieces = 1)
int gemm_bf_supported(const WGemm& d);
int launch_gemm_bf(const WGemm* d, int n, hipStream_t s);
inline int wimg_pieces(const WGemm& d) { return gemm_bf_supported(d) ? 1 : (d.f16 && d.np == 2 && !d.bf16 ? 16 : 3); }
// two FP16 pieces per operand on 128 x 256 tiles with the weight through LDS (gemm_h2.hip): the tolerance mode's projections
int gemm_h2_supported(const WGemm& d);
int launch_gemm_h2(const WGemm* d, int n, hipStream_t s);
// the kernel a pre-split-weight GEMM runs on: gemm_bf / gemm_h2 when they take the shape and mode, else gemm_w
inline int gemm_wx_kernel(const WGemm& d) { return gemm_bf_supported(d) ? 1 : (gemm_h2_supported(d) ? 2 : 0); }
inline int launch_gemm_wx(const WGemm* d, int n, hipStream_t s, const RowFlagJob* rows = nullptr) {
  if (n == 2 && gemm_wx_kernel(d[0]) != gemm_wx_kernel(d[1])) {   // one job on each kernel: two launches
    const int rc = launch_gemm_wx(&d[0], 1, s);
    return rc ? rc : launch_gemm_wx(&d[1], 1, s, rows);
  }
  const int k = gemm_wx_kernel(d[0]);
  return k == 1 ? launch_gemm_bf(d, n, s) : (k == 2 ? launch_gemm_h2(d, n, s) : launch_gemm_w(d, n, s, rows));
}

// ---- weight-gradient GEMM C = A^T B with split-K parts (gemm_tn.hip) ---------------------------------------------
struct TnGemm {
  const float* A; long a_sl; int a_ld;                               // A_l[k][m] at A + l * a_sl, row stride a_ld
  long a_term;                                                       // != 0: A = sum of the 3 arrays A + t * a_term
  const float* B; const float* b_ptrs[8]; long b_sl; int b_ld;       // B_l[k][n]: table entry l, or B + l * b_sl
  int b_kdiv; long b_sdiv;                                           // b_kdiv != 0: B contiguous along k instead, element
                                                                     // (k, n) at (k / b_kdiv) * b_sdiv + k % b_kdiv + n * b_ld
  float* C;                                                          // parts [levels * S][M][N]
  int M, N, K, levels;
  int mask_blk; unsigned tile_mask;                                  // mask_blk > 0: tile (mt, nt) is computed only if bit
                                                                     // (mt / mask_blk) * 3 + nt / mask_blk of tile_mask is set
  int bf16;                                                          // 1: operands rounded to bf16, ONE MFMA per product
  int np;                                                            // (bf16 = 0) bf16 pieces per operand: 0 / 3 = the exact split, 2 = hi + mid (three products)
  int a_bf16;                                                        // (gemm_bf_tn_kernel) A is STORED as bf16; a_ld, a_sl, a_term in elements
};
// Weight gradients over the LIVE question rows only (the pad rows of Q are exact zeros: their terms of dW_q = sum dP_q^T Q vanish).
// The launch of the two weight gradients then plans its split-K parts ON THE DEVICE from the forward's row bitmap (fused.h
// RowFlagJob; `saved` keeps it): P parts in all, S0 of them for job 0 (contraction length K0), S1 per level for job 1 over that
// level's live rows, chosen so that all parts are about equally long -- the host cannot know the count without a
// synchronisation.  Every workgroup of the launch and the workgroups that later add the partial results evaluate tn_dyn_plan
// on the same words: the same integers everywhere.
constexpr int kDynLevels = 4;                                            // levels a device-planned launch takes (the path has 3)
struct TnDynPlan { int S0, S1, ks0, live[kDynLevels], ks1[kDynLevels]; };
struct TnDyn { const unsigned* bits; int words, levels, P, K0;           // bits == NULL: the host's static plan
               const TnDynPlan* plan; };                                 // the plan, evaluated once (an extra workgroup of bwd_pre_kernel)
// by ONE full wave (all 64 lanes active); the result is uniform.  (Loops over the levels are unrolled over kDynLevels with the
// live ones selected by a compare: a run-time index into the arrays would put them in scratch.)
__device__ __forceinline__ TnDynPlan tn_dyn_plan(const TnDyn& d) {
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  TnDynPlan pl;
  int tot = 0;
#pragma unroll
  for (int l = 0; l < kDynLevels; ++l) {
    int c = 0;
    if (l < d.levels) {
      for (int i = lane; i < d.words; i += 64) c += __builtin_popcount(d.bits[(long)l * d.words + i]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    }
    pl.live[l] = c;
    tot += c;
  }
  const long den = (long)d.K0 + tot;
  int S0 = (int)((2L * d.P * d.K0 + den) / (2 * den));
  S0 = S0 < 1 ? 1 : (S0 > d.P - d.levels ? d.P - d.levels : S0);
  pl.S0 = S0;
  pl.S1 = (d.P - S0) / d.levels;
  pl.ks0 = (((d.K0 + S0 - 1) / S0) + 15) / 16 * 16;
#pragma unroll
  for (int l = 0; l < kDynLevels; ++l) {
    const int n = pl.live[l] > 0 ? pl.live[l] : 1;
    pl.ks1[l] = (((n + pl.S1 - 1) / pl.S1) + 15) / 16 * 16;
  }
  return pl;
}
int gemm_tn_supported(const TnGemm& d);
// the same products in the reduced-precision mode at wide shapes (gemm_bf.hip): 256 x 256 tiles, parts over the concatenated levels
int gemm_bf_tn_supported(const TnGemm& d);
int bf_tn_rounds();
int gemm_bf_tn_plan(const TnGemm& d, int want, int* spp);                 // returns the number of parts
int launch_gemm_bf_tn(const TnGemm* d, const int* spp, const int* parts, int n, hipStream_t s);
int gemm_tn_plan(const TnGemm& d, int max_parts, int* ksplit, int* S);   // returns the number of parts
struct TnReduce {                                                    // launch_reduce_jobs's arguments (common.h)
  const float* src[4]; float* dst[4]; int njobs, nparts; long n; int accumulate;
  const float* sum_x[2]; float* sum_out[2]; long sum_n;
  long ld;                                                           // row stride of the partial buffers (0: n)
};
// n = 1 or 2 GEMMs per launch; red (may be NULL): small reductions done by extra workgroups of the same launch
// wextra (may be NULL): one pre-split-weight GEMM (row-major A) whose tiles run as the last workgroups of the launch
int launch_gemm_tn(const TnGemm* d, const int* ksplit, const int* S, int n, hipStream_t s, const TnReduce* red = nullptr,
                   const WGemm* wextra = nullptr);

// the same launch on 128 x 256 tiles / 512-thread workgroups, for the two-piece width (gemm_tn_wide.hip)
int gemm_tn_wide_supported(const TnGemm& d);
int gemm_tn_wide_plan(const TnGemm& d, int max_parts, int* ksplit, int* S);
// dyn (may be NULL; n == 2, both jobs on the same tile grid): TnDyn of the launch -- job 1 contracts over the rows whose bit is set
int launch_gemm_tn_wide(const TnGemm* d, const int* ksplit, const int* S, int n, hipStream_t s, const TnReduce* red = nullptr,
                        const WGemm* wextra = nullptr, const TnDyn* dyn = nullptr);

// XCD-aware block -> (b, l): blocks i and i+8 share an XCD (round-robin dispatch), so give the
// L levels of one sample consecutive slots on one XCD.  Speed only; any mapping is correct.
__device__ __forceinline__ bool block_to_pair(int bid, int B, int L, int& b, int& l) {
  const int x = bid & 7, slot = bid >> 3;
  b = (slot / L) * 8 + x;
  l = slot % L;
  return b < B;
}

// gfx950 transposing LDS read: each 16-lane group fetches a 4 x 16 block of 16-bit elements transposed
__device__ __forceinline__ bf16x4 lds_tr16(const short* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(const_cast<short*>(p)));
}

// Workgroup barrier on the LDS counter alone: the LDS traffic of every wave has landed, global loads and stores stay
// in flight (a __syncthreads() drains them too).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// C/D row of accumulator register g in lane half h (32x32 MFMA)
__device__ __forceinline__ constexpr int crow(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// arguments of the fused forward kernel (coattn_fused.hip)
struct FwdArgs {
  const float* V;        // [B][d][N] (lm = 0) or [B][N][d] (lm = 1), sample stride v_sB
  long v_sB;
  int lm;
  const float* Q[8];     // L x [B][T][d]
  const float* Pv;       // [B][N][d]
  const float* Pq;       // [L][B][T][d]
  const float* wv; const float* cv; const float* wq; const float* cq;
  float* C;              // [L][B][T][N]
  float* av;             // [L][B][N]
  float* aq;             // [L][B][T]
  float* Hq;             // [L][B][T][d]
  float* q_out;          // [L][B][d]
  float* v_out;          // [L][B][d]: written by the kernel itself when it also attends the image features (location-major
                         // features on N <= 64 locations; otherwise NULL and attend_v(_lm)_kernel follows)
  unsigned long long* stamps;   // diagnostic builds only
  int B, N, T, d, L;
  int bf16;              // reduced-precision mode: operands rounded to bf16, one MFMA per product (d % 512 == 0)
  int np;                // (bf16 = 0) width of the phase-2 contractions C^T P_q, C P_v: 3 or 2 pieces (the affinity: always 3)
};

// arguments of the two big fused backward kernels (coattn_fused_bwd.hip, coattn_bwd32.hip)
struct BwdArgs {
  const float* Pv;        // [B][N][d]
  const float* Pq;        // [L][B][T][d]
  const float* C;         // [L][B][T][N]
  const float* dav_part;  // [B][nkc][3][N] channel-chunk partials of da_v = V gv (bwd_pre_kernel's extra blocks)
  int nkc;
  const float* av;        // saved a_v [L][B][N]
  float* dcs_part;        // [2][L*B]: dc_v partials (written by bwd_nat32_kernel)
  const float* Hq;        // saved H_q [L][B][T][d]: dZ_q = ds_q (x) w_q (.) (1 - H_q^2) is formed where it is used
  const float* dsq;       // [L*B][32] ds_q (bwd_pre_kernel; zeros for t >= T)
  const float* wq;
  const float* wv;
  float* dPv;             // [L][B][N][d]
  float* dPq;             // [L][B][T][d]
  float* dA;              // [L][B][T][N]
  float* dwv_part;        // [L*B][d]
  float* dbv_part;        // [L*B][d]   sum_n dP_v[n][:]
  float* dbq_part;        // [L*B][d]   sum_t dP_q[t][:]
  float* dwq_part;        // [L*B][d]   sum_t ds_q[t] H_q[t][:]   (bwd_nat32_kernel)
  int B, N, T, d, L;
  int bf16;               // reduced-precision mode: one MFMA per product (d % 512 == 0)
  int np;                 // (bf16 = 0) width of the contractions: 3 or 2 pieces
  int dp_bf16;            // (with bf16, bwd_nat32_kernel) dPv / dPq are bf16 arrays of the same index order
  int ko_dpv;             // developer knock-out (DEV builds only, wrong results): bwd_nat32 stores dP_v of level 0 alone
};

// Image-side softmax backward of one (sample, level) by ONE wave: da_v = the sum of the channel-chunk partials,
// ds_v = a_v (da_v - <a_v, da_v>) into the LDS array dsvs[0 .. npad) (zeros beyond N); returns sum_n ds_v (the dc_v partial,
// valid in every lane).  Both big backward kernels run it in their prologue (same arithmetic, same values) -- it used to
// be a launch of its own (one wave per pair, 5 us of launch latency).  N <= 256.
__device__ __forceinline__ float softmax_bwd_v(const BwdArgs& a, int b, int l, int lane, float* dsvs, int npad) {
  const int N = a.N, nkc = a.nkc;
  const float* pp = a.dav_part + (size_t)b * nkc * 3 * N + (size_t)l * N;
  const float* avp = a.av + ((size_t)l * a.B + b) * N;
  // the lane's four locations are requested TOGETHER, from clamped addresses (a guarded load is a branch and a wait of its own:
  // four memory latencies in a row at the head of both big kernels); per location the chunks add up in the same order as before
  float da[4] = {0.f, 0.f, 0.f, 0.f}, avv[4];
  int idx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) idx[k] = min(lane + 64 * k, N - 1);
#pragma unroll
  for (int k = 0; k < 4; ++k) avv[k] = avp[idx[k]];
  for (int kc = 0; kc < nkc; ++kc) {
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = pp[(size_t)kc * 3 * N + idx[k]];
#pragma unroll
    for (int k = 0; k < 4; ++k) da[k] += t[k];
  }
  float dot = 0.f, tot = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool in = lane + 64 * k < N;
    da[k] = in ? da[k] : 0.f;
    avv[k] = in ? avv[k] : 0.f;
    dot = fmaf(avv[k], da[k], dot);
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int n = lane + 64 * k;
    const float v = avv[k] * (da[k] - dot);
    if (n < npad) dsvs[n] = v;                       // (exact zeros beyond N: a_v reads as 0 there)
    tot += v;
  }
  return wave_sum(tot);
}

// dP_q on the bf16 MFMA 32x32x16 with the exact 3-way split (coattn_bwd32.hip); same shapes as the fused forward
int launch_bwd_nat32(const BwdArgs& a, hipStream_t s);
// dC and dA in the [channels][locations] orientation on the bf16 MFMA (coattn_bwd32.hip)
int launch_bwd_dc32(const BwdArgs& a, hipStream_t s);

// arguments of the dQ kernels (bwd_dq_kernel in coattn_fused_bwd.hip, bwd_dq32_kernel in coattn_bwd32.hip)
struct DqArgs {
  const float* V; long v_sB; const float* dA; const float* aq; const float* gq;
  float* dQ[8];
  int B, N, T, d, L;
  int accumulate;        // bwd_dq32_kernel: add onto dQ (which then already holds dP_q W_q) instead of overwriting it
  int bf16;              // reduced-precision mode: one MFMA per product
  int np;                // (bf16 = 0) width of the contraction: 3 or 2 pieces
  // bwd_dq32(x)_kernel: the sum of the weight gradients' split-K partials rides along as the launch's last workgroups
  // (red_blocks of them per job, red_jobs jobs: out[j] (+)= sum_c part[c][j], four floats per thread) -- nothing of it
  // depends on dQ, and a launch of its own cost a launch gap and 7 us during which nothing else ran
  const float* red_part[2]; float* red_out[2]; int red_np[2]; long red_n; int red_acc, red_blocks, red_jobs;
  TnDyn red_dyn;         // bits != NULL: the parts were planned on the device (tn_dyn_plan): job 0 = parts [0, S0), job 1 = the
                         // levels * S1 parts behind them, all in red_part[0]
};
__device__ __forceinline__ void reduce_partials4_block(const DqArgs& a, int id) {   // (as small_kernels.hip reduce_partials4_kernel)
  const int job = id / a.red_blocks, bx = id - job * a.red_blocks;
  const float* part = a.red_part[job];
  float* out = a.red_out[job];
  int nparts = a.red_np[job];
  if (a.red_dyn.bits) {                              // (the plan bwd_pre_kernel's extra workgroup left in the workspace)
    const int S0 = a.red_dyn.plan->S0, S1 = a.red_dyn.plan->S1;
    nparts = job == 0 ? S0 : a.red_dyn.levels * S1;
    part = a.red_part[0] + (job == 0 ? 0L : (long)S0 * a.red_n);
  }
  const long n = a.red_n, j = ((long)bx * 256 + threadIdx.x) * 4;
  if (j >= n) return;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  int c = 0;
  for (; c + 8 <= nparts; c += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(part + (long)(c + u) * n + j);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; c < nparts; ++c) acc += *reinterpret_cast<const f32x4*>(part + (long)c * n + j);
  f32x4* o = reinterpret_cast<f32x4*>(out + j);
  *o = a.red_acc ? *o + acc : acc;
}
// dQ_l = a_q (x) gq + dA_l V on the bf16 MFMA with the exact 3-way split: location-major V (lm), or channel-major V
// whose rows are 16-byte multiples (N % 4 == 0)
int launch_bwd_dq32(const DqArgs& a, int lm, hipStream_t s);

// bf16-split forward kernel on the 32x32x16 MFMA (coattn_fwd32.hip)
int fused32_forward(const FwdArgs& a, hipStream_t s);
int launch_attend_v_lm(const float* V, long v_sB, const float* av, float* v_out, int B, int N, int d, int L, hipStream_t s);

inline size_t fal64(size_t n) { return (n + 63) & ~(size_t)63; }

struct SavedOff {
  size_t Pv, Pq, C, av, aq, Hq, wqT, status, rowcnt, rowbits, total;
};
inline SavedOff saved_off(int B, int N, int T, int d, int L) {   // the one layout of `saved`
  SavedOff p;
  size_t o = 0;
  p.Pv = o; o += fal64((size_t)B * N * d);
  p.Pq = o; o += fal64((size_t)L * B * T * d);
  p.C = o;  o += fal64((size_t)L * B * T * N);
  p.av = o; o += fal64((size_t)L * B * N);
  p.aq = o; o += fal64((size_t)L * B * T);
  p.Hq = o; o += fal64((size_t)L * B * T * d);
  // W_q split for the backward's dQ projection (gemm_w.hip, wsplit_bytes(d, d)): written by the forward's weight-split
  // launch, so that the backward has no split launch of its own
  p.wqT = o; o += fal64((size_t)((d + 31) / 32) * ((d + 15) / 16) * 768);
  p.status = o; o += fal64(status_floats(d, d, 2));   // range report of the tolerance mode (W_v, W_q images)
  // bitmap of the question rows that are not all zeros (RowFlagJob; one word per 32 rows and level) and, in front of it, eight
  // words of counters: the exact forward writes it, the backward's weight gradients contract over the live rows only
  p.rowcnt = o; o += 64;
  p.rowbits = o; o += fal64(rowbits_words(B * T, L));
  p.total = o;
  return p;
}

// workspace of the fused backward (floats)
struct FusedBwdOff {
  size_t dsv, dsq, dPq, dPv, dA, dwv_part, dbv_part, dbq_part, dwq_part, dcs_part, dynplan, part, total;
};
constexpr int kMaxParts = 40;   // split-K parts of the weight-gradient GEMMs (32 shared by dW_v and dW_q, rounded up per level)
inline FusedBwdOff fused_bwd_off(int B, int N, int T, int d, int L) {
  FusedBwdOff p;
  size_t o = 0;
  p.dsv = o; o += fal64((size_t)L * B * N);
  p.dsq = o; o += fal64((size_t)L * B * 32);
  p.dPq = o; o += fal64((size_t)L * B * T * d);
  p.dPv = o; o += fal64((size_t)L * B * N * d);
  p.dA = o;  o += fal64((size_t)L * B * T * N);
  p.dwv_part = o; o += fal64((size_t)L * B * d);
  p.dbv_part = o; o += fal64((size_t)L * B * d);
  p.dbq_part = o; o += fal64((size_t)L * B * d);
  p.dwq_part = o; o += fal64((size_t)L * B * d);
  p.dcs_part = o; o += fal64((size_t)L * B * 2);
  p.dynplan = o; o += 64;                          // TnDynPlan of a device-planned weight-gradient launch
  // shared scratch: split-K partials of the weight-gradient GEMMs (<= kMaxParts x d x d) and, before them, the da_v
  // partials of bwd_dav_kernel ([B][d/64][3][N]), which outgrow the former at large B
  const size_t part_gemm = (size_t)kMaxParts * d * d, part_dav = (size_t)B * (d / 64) * 3 * N;
  p.part = o; o += fal64(part_gemm > part_dav ? part_gemm : part_dav);
  p.total = o;
  return p;
}
size_t fused_bwd_ws_floats(int B, int N, int T, int d, int L);
