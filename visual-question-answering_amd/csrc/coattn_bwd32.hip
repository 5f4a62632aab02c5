// The three big kernels of the fused backward (autograd of model.py:377-392; SURVEY.md section 8 "Backward"), all on
// the bf16 MFMA 32x32x16 with the exact 3-way split (fused.h):
//   bwd_nat32_kernel  orientation [locations][channels]: dP_q = dZ_q + C dZ_v, dP_v = dZ_v + C^T dZ_q, dw_v, db_v, db_q
//   bwd_dc32_kernel   orientation [channels][locations]: dC = P_q dZ_v^T + dZ_q P_v^T, dA = dC (.) (1 - C^2)
//   bwd_dq32_kernel   dQ_l (+)= a_q (x) gq + dA_l V (location-major image features; channel-major when N % 4 == 0)
// (an accumulator tile feeds the next MFMA only along its row index, hence the two orientations: each recomputes H_v).
//
// bwd_nat32_kernel<NT,NW>: one workgroup per (sample, level), NW waves owning 128-channel slices, on the bf16 MFMA
// 32x32x16 with the exact 3-way split (fused.h) -- the backward twin of the forward kernel's phase 2
// (coattn_fwd32.hip), whose operand layouts it shares:
//   * C (saved) is split once into the LDS image [piece][n][32 t]; row reads give C^T fragments (A operand of
//     H_v = P_v + C^T P_q and of dP_v = dZ_v + C^T dZ_q), transposing reads give C fragments (A operand of
//     dP_q += C dZ_v);
//   * unit pipeline over 32-location x 32-channel fragments of P_v (accumulator-shaped, 128 contiguous bytes per half
//     wave and load, two units ahead): the fragment is the accumulator of the recomputed H_v tile; tanh and
//     dZ_v = ds_v w_v (1 - H_v^2) = 4 ds_v w_v r (1 - r), r = 1 / (1 + e^{2 H}), happen in place; split, the dZ_v
//     fragment is the B operand of dP_q += C dZ_v (contraction over the fragment's row index = locations), and then
//     the accumulator of dP_v, stored as it lies (whole 128-byte row segments);
//   * the dP_q accumulators start from dZ_q; dw_v, db_v, db_q partials are in-lane sums over the accumulator rows;
//   * dZ_q = ds_q (x) w_q (.) (1 - H_q^2) is formed from the saved H_q as it is loaded (both kernels; ds_q comes from
//     bwd_pre_kernel), and bwd_nat32_kernel sums the dw_q partial sum_t ds_q[t] H_q[t][:] on the way.
// H_v is never stored.  Rows t >= T / n >= N fall outside the per-sample buffer descriptors (loads 0, stores dropped).
#include "fused.h"
#include <stdlib.h>
#include <type_traits>

#ifndef COATTN_DC_GT          // bwd_dc32_kernel: location tiles per group (their dC accumulators: 16 registers each)
#define COATTN_DC_GT 4
#endif

namespace {

// NP: width of the contractions (fused.h: 3 = exact split, 2 = hi + mid, 1 = the reduced-precision mode SP).
// DPB (with SP): dP_v and dP_q are stored as bf16 (same index order, 2-byte elements) -- every consumer is a GEMM of the
// reduced-precision mode, which would round them on its way in anyway.  Lane pairs (r, r + 1) meet through a DPP quad
// permute, the even lane stores one dword for both channels.
template <int NT, int NW, int NP, bool DPB>
__global__ __launch_bounds__(NW * 64, 2) void bwd_nat32_kernel(const BwdArgs a) {
  static_assert(!DPB || NP == 1, "bf16 dP storage belongs to the reduced-precision mode");
  constexpr int NPAD = 32 * NT;
  constexpr int PIECE = NPAD * 32;                   // bf16 elements of one piece of the C image [n][t = 32]
  constexpr int NTHR = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  short* Cimg = reinterpret_cast<short*>(smem);
  float* dsvs = reinterpret_cast<float*>(smem + 3 * PIECE * 2);   // [NPAD] ds_v, zero padded
  float* dsqs = dsvs + NPAD;                                      // [32] ds_q, zeros for t >= T
  int b, l;
  if (!block_to_pair(blockIdx.x, a.B, a.L, b, l)) return;
  const int N = a.N, T = a.T, d = a.d;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int tid = w * 64 + lane, r = lane & 31, h = lane >> 5;
  const size_t pair = (size_t)l * a.B + b;
  const __amdgpu_buffer_rsrc_t rs_pv = make_rsrc(a.Pv + (size_t)b * N * d, (unsigned)N * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_pq = make_rsrc(a.Pq + pair * (size_t)T * d, (unsigned)T * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_hq = make_rsrc(a.Hq + pair * (size_t)T * d, (unsigned)T * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_wq = make_rsrc(a.wq, (unsigned)d * 4u);
  const __amdgpu_buffer_rsrc_t rs_dwq = make_rsrc(a.dwq_part + pair * (size_t)d, (unsigned)d * 4u);
  constexpr int ES = DPB ? 2 : 4;                    // bytes of a stored dP element
  const __amdgpu_buffer_rsrc_t rs_dpq = make_rsrc(reinterpret_cast<const char*>(a.dPq) + pair * (size_t)T * d * ES, (unsigned)T * d * ES);
  const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(a.C + pair * (size_t)T * N, (unsigned)T * N * 4u);
  const int nsl = d / (128 * NW);                    // 128-channel slices per wave (1 at d = 512)
  const __amdgpu_buffer_rsrc_t rs_dpv = make_rsrc(reinterpret_cast<const char*>(a.dPv) + pair * (size_t)N * d * ES,
                                                  (a.ko_dpv && l > 0) ? 0u : (unsigned)N * d * ES);   // (ko_dpv: developer knock-out, stores dropped)
  // the lane's channel inside a 32-channel unit as the stores see it: bf16 stores leave from the even lanes only (an odd
  // lane's offset lies outside every buffer)
  const int rst = DPB && (lane & 1) ? 0x20000000 : r;
  // element (row, this lane's channel) of a dP array: row_elems = row * d (+ rst inside), so = the unit's scalar offset
  auto store_dp = [&](const __amdgpu_buffer_rsrc_t rs, float v, const int row_elems, const int so_elems) {
    if constexpr (DPB) {
      const float nb = dpp_mov<0xB1, 0xF>(v, v);     // quad_perm [1,0,3,2]: the neighbouring channel
      __builtin_amdgcn_raw_buffer_store_b32(cvt_pk_bf16(v, nb), rs, (row_elems + rst) * 2, so_elems * 2, 0);
    } else {
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (row_elems + rst) * 4, so_elems * 4, 0);
    }
  };
  // B operands of a pass (32 channels c0 .. c0 + 31, lane r <-> channel c0 + r): P_q in the k order of the C^T rows
  // (t = 16 ks + 8 h + i), raw; H_q accumulator-shaped (rows crow(g, h)): at the top of its pass it becomes dZ_q, which
  // is also the start of the dP_q accumulators -- its B-operand order comes from one v_permlane32_swap per register
  // pair (coattn_fwd32.hip).
  auto load_pass = [&](int c0, f32x8 (&praw)[2], f32x16& zf) {
    int bn = (8 * h * d + r) * 4, bc = (4 * h * d + r) * 4;
    asm volatile("" : "+v"(bn));
    asm volatile("" : "+v"(bc));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) praw[ks][i] = buf_load1(rs_pq, bn, (c0 + (16 * ks + i) * d) * 4);
#pragma unroll
    for (int g = 0; g < 16; ++g) zf[g] = buf_load1(rs_hq, bc, (c0 + ((g & 3) + 8 * (g >> 2)) * d) * 4);
  };
  f32x8 pq_raw[2];
  f32x16 zq_frag;
  load_pass(w * 128, pq_raw, zq_frag);               // the first pass' operands fly under the image build

  // ---- the image of C (three bf16 pieces, [piece][n][t], 64-byte rows with XOR-swizzled 16-byte chunks) and ds_v
  {
    constexpr int PER = 8 * NPAD / NTHR;             // (location, token quad) items per thread
    static_assert(8 * NPAD % NTHR == 0, "the image build covers the image in whole sweeps");
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int e = tid + k * NTHR, tq = e / NPAD, n = e - tq * NPAD;
      const int cvoff = n < N ? (4 * tq * N + n) * 4 : 0x40000000;     // padded columns and rows >= T read 0
      float c[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = buf_load1(rs_c, cvoff, i * N * 4);
      unsigned hh[2] = {0, 0}, mm[2] = {0, 0}, ll[2] = {0, 0};
      split_pair<NP>(c[0], c[1], hh[0], mm[0], ll[0]);
      split_pair<NP>(c[2], c[3], hh[1], mm[1], ll[1]);
      const int off = n * 32 + 8 * ((tq >> 1) ^ ((n >> 2) & 3)) + 4 * (tq & 1);
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<u32x2*>(Cimg + off) = u32x2{hh[0], hh[1]};
      if (NP >= 2) *reinterpret_cast<u32x2*>(Cimg + PIECE + off) = u32x2{mm[0], mm[1]};
      if (NP == 3) *reinterpret_cast<u32x2*>(Cimg + 2 * PIECE + off) = u32x2{ll[0], ll[1]};
    }
    if (w == NW - 1) {                               // ds_v of this (sample, level); its sum is the dc_v partial
      const float tot = softmax_bwd_v(a, b, l, lane, dsvs, NPAD);
      if (lane == 0) a.dcs_part[pair] = tot;
    }
    if (w == 0 && lane < 32) dsqs[lane] = a.dsq[pair * 32 + lane];
  }
  lds_barrier();

  const int ntiles = (N + 31) >> 5;
  // lane constants of the image reads (see coattn_fwd32.hip)
  const int tq = (lane & 15) >> 2, tp_ = lane & 3, g1 = (lane >> 4) & 1;
  const int tr_off0 = (4 * h + tq) * 32 + 8 * ((2 * g1 + (tp_ >> 1)) ^ h) + 4 * (tp_ & 1);
  const int tr_off1 = (4 * h + 8 + tq) * 32 + 8 * ((2 * g1 + (tp_ >> 1)) ^ (h + 2)) + 4 * (tp_ & 1);
  const int rk = (r >> 2) & 3;
  auto read_cq = [&](const short* img, const int s2, bf16x8 (&cq)[3]) {     // A = C (tokens x locations)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const bf16x4 lo = lds_tr16(img + p * PIECE + 16 * s2 * 32 + tr_off0);
      const bf16x4 hi = lds_tr16(img + p * PIECE + 16 * s2 * 32 + tr_off1);
      cq[p] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };
  auto read_ca = [&](const short* img, const int ks, bf16x8 (&ca)[3]) {      // A = C^T (locations x tokens)
#pragma unroll
    for (int p = 0; p < NP; ++p)
      ca[p] = *reinterpret_cast<const bf16x8*>(img + p * PIECE + r * 32 + 8 * ((2 * ks + h) ^ rk));
  };

#pragma unroll 1
  for (int pi = 0; pi < 4 * nsl; ++pi) {             // passes: 32 channels each, four per 128-channel slice
    const int c0 = ((pi >> 2) * NW + w) * 128 + 32 * (pi & 3);
    const int c0n = (((pi + 1) >> 2) * NW + w) * 128 + 32 * ((pi + 1) & 3);
    const float wv4 = 4.0f * a.wv[c0 + r];
    bf16x8 pqB[2][3], zqB[2][3];
    {
      float dwq = 0.f;                               // this lane's sum_t ds_q[t] H_q[t][c0 + r] over its rows
      int ho = h;                                    // (opaque: the lane constants below are rebuilt per pass instead of
      asm volatile("" : "+v"(ho));                   //  living in registers through the unit loop, which has none to spare)
      const float wqc = buf_load1(rs_wq, (lane - 32 * ho) * 4, c0 * 4);
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        const f32x4 sq = *reinterpret_cast<const f32x4*>(&dsqs[8 * gg + 4 * ho]);   // rows crow(4 gg + i, h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float hq = zq_frag[4 * gg + i];
          dwq = fmaf(sq[i], hq, dwq);
          zq_frag[4 * gg + i] = sq[i] * wqc * (1.0f - hq * hq);
        }
      }
      dwq = half_sum(dwq);                // (stored here: nothing of it lives through the pass)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dwq), rs_dwq, ho == 0 ? lane * 4 : 0x40000000, c0 * 4, 0);
    }
    f32x16 accq = zq_frag;                           // dP_q = dZ_q + C dZ_v
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      splitn<NP>(pq_raw[ks], pqB[ks]);
      f32x8 x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float lo = zq_frag[8 * ks + k], hi = zq_frag[8 * ks + 4 + k];
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
        x[k] = lo;
        x[4 + k] = hi;
      }
      splitn<NP>(x, zqB[ks]);
    }
    float dwacc = 0.f, dbacc = 0.f;                  // this lane's column sums over its rows: dw_v, db_v
    f32x16 ring[4];
    u32x4 Ph[2], Pm[2], Pl[2];                       // split dZ_v of the previous unit (two k-steps of 16 locations)
    // pv[g] = P_v[32 nt + crow(g, h)][c0 + r]; past the last tile: zeros through the buffer rule, no traffic
    auto load_unit = [&](int u, f32x16& dst) {
#pragma unroll
      for (int g = 0; g < 16; ++g) dst[g] = buf_load1(rs_pv, (crow(g, h) * d + r) * 4, (32 * u * d + c0) * 4);
    };
    // One step = unit u (location tile u of this pass' channels).
    //  [a] the 12 MFMAs of H_v(u) = P_v + C^T P_q on the fragment `cur`; beside them the split of dZ_v(u-1) (`dzp`);
    //  [b] the 12 MFMAs of dP_q += C dZ_v(u-1);
    //  [c] the 12 MFMAs of dP_v(u-1) = dZ_v(u-1) + C^T dZ_q on `dzp` (already split), then its stores and db_v terms;
    //      beside [b] and [c]: tanh / dZ_v of unit u in place in `cur`, and its dw_v terms.
    auto step = [&](const int u, f32x16& cur, f32x16& dzp) {
      const short* img = Cimg + 32 * u * 32;
      const short* imgp = Cimg + 32 * (u > 0 ? u - 1 : 0) * 32;
      bf16x8 ca0[3], ca1[3], cq0[3], cq1[3];
      f32x4 dsn[4];
      read_ca(img, 0, ca0);
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        const int ks = m / 6, i = m % 6;
        const int k = slot_product<NP>(i);         // (width 2: an MFMA in every other slot)
        if (k >= 0) cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks ? ca1[piece_a<NP>(k)] : ca0[piece_a<NP>(k)], pqB[ks][piece_b<NP>(k)], cur, 0, 0, 0);
        if (m == 1) read_ca(img, 1, ca1);           // operands are read one MFMA group ahead of their use
        if (m == 8) read_cq(imgp, 0, cq0);
        if (m == 11) {
#pragma unroll
          for (int gg = 0; gg < 4; ++gg) dsn[gg] = *reinterpret_cast<const f32x4*>(&dsvs[32 * u + 8 * gg + 4 * h]);
        }
        if (m < 8) {                                 // split pair m of dZ_v(u-1)
          unsigned hh = 0, mm = 0, ll = 0;
          split_pair<NP>(dzp[2 * m], dzp[2 * m + 1], hh, mm, ll);
          Ph[m >> 2][m & 3] = hh; Pm[m >> 2][m & 3] = mm; Pl[m >> 2][m & 3] = ll;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      auto dz_reg = [&](const int g) {
        const float rr = sig2_scaled(cur[g]);
        const float ds = dsn[g >> 2][g & 3];
        dwacc = fmaf(ds, fmaf(-2.0f, rr, 1.0f), dwacc);               // ds_v[n] H_v[n][k]
        cur[g] = fmaf(-rr, rr, rr) * ds * wv4;                        // r (1 - r) = (1 - tanh^2) / 4
      };
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        const int ks = m / 6, i = m % 6;
        const int k = slot_product<NP>(i), kk = k < 0 ? 0 : k;
        const bf16x8 bp = piece_b<NP>(kk) == 0 ? __builtin_bit_cast(bf16x8, Ph[ks]) : piece_b<NP>(kk) == 1 ? __builtin_bit_cast(bf16x8, Pm[ks])
                                                                                                          : __builtin_bit_cast(bf16x8, Pl[ks]);
        if (k >= 0) accq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks ? cq1[piece_a<NP>(kk)] : cq0[piece_a<NP>(kk)], bp, accq, 0, 0, 0);
        if (m == 1) read_cq(imgp, 1, cq1);
        if (m == 9) read_ca(imgp, 0, ca0);
        if (m == 11) read_ca(imgp, 1, ca1);
        if (m >= 4) dz_reg(m - 4);                   // dZ_v of unit u in place, registers 0 .. 7 (8 .. 15 follow in [c])
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        const int ks = m / 6, i = m % 6;
        const int k = slot_product<NP>(i);
        if (k >= 0) dzp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks ? ca1[piece_a<NP>(k)] : ca0[piece_a<NP>(k)], zqB[ks][piece_b<NP>(k)], dzp, 0, 0, 0);
        if (m < 8) dz_reg(8 + m);
        __builtin_amdgcn_sched_barrier(0);
      }
      // dP_v(u-1) out (u = 0: zeros into rows that do not exist -- the offset lies outside the buffer)
      const int so = u > 0 ? 32 * (u - 1) * d + c0 : 0x10000000;   // (elements: outside the buffer at either element size)
      const float live = u > 0 ? 1.f : 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        float v = dzp[g];
        asm volatile("" : "+v"(v));                  // (opaque scalar: see the epilogue)
        store_dp(rs_dpv, v, crow(g, h) * d, so);
        dbacc = fmaf(live, v, dbacc);
      }
    };
    load_unit(0, ring[0]);
    load_unit(1, ring[1]);
#pragma unroll
    for (int g = 0; g < 16; ++g) ring[3][g] = 0.f;   // "dZ_v of unit -1"
#pragma unroll 1
    for (int u0 = 0; u0 < ntiles; u0 += 4) {
      load_unit(u0 + 2, ring[2]);
      __builtin_amdgcn_sched_barrier(0);
      step(u0, ring[0], ring[3]);
      __builtin_amdgcn_sched_barrier(0);
      if (u0 + 1 < ntiles) {
        load_unit(u0 + 3, ring[3]);
        __builtin_amdgcn_sched_barrier(0);
        step(u0 + 1, ring[1], ring[0]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (u0 + 2 < ntiles) {
        load_unit(u0 + 4, ring[0]);
        __builtin_amdgcn_sched_barrier(0);
        step(u0 + 2, ring[2], ring[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (u0 + 3 < ntiles) {
        load_unit(u0 + 5, ring[1]);
        __builtin_amdgcn_sched_barrier(0);
        step(u0 + 3, ring[3], ring[2]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the next pass' operands fly under the tail and the epilogue (beyond the last pass: channel offsets >= d,
    // the values are never used)
    // (three pieces: the registers are all taken across the tail -- the request waits until it is done)
    if constexpr (NP != 3) load_pass(c0n, pq_raw, zq_frag);
    // the last unit's dZ_v: split, dP_q and dP_v contributions, dP_v out
    auto finish = [&](f32x16& dz) {
      const int lu = ntiles - 1;
      const short* imgp = Cimg + 32 * lu * 32;
      bf16x8 cq0[3], cq1[3], ca0[3], ca1[3], b0[3], b1[3];
      read_cq(imgp, 0, cq0);
      read_cq(imgp, 1, cq1);
      read_ca(imgp, 0, ca0);
      read_ca(imgp, 1, ca1);
      splitn<NP>(f32x8{dz[0], dz[1], dz[2], dz[3], dz[4], dz[5], dz[6], dz[7]}, b0);
      splitn<NP>(f32x8{dz[8], dz[9], dz[10], dz[11], dz[12], dz[13], dz[14], dz[15]}, b1);
      accq = mfma32_xn<NP>(cq0, b0, accq);
      accq = mfma32_xn<NP>(cq1, b1, accq);
      dz = mfma32_xn<NP>(ca0, zqB[0], dz);
      dz = mfma32_xn<NP>(ca1, zqB[1], dz);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        float v = dz[g];
        asm volatile("" : "+v"(v));
        store_dp(rs_dpv, v, crow(g, h) * d, 32 * lu * d + c0);
        dbacc += v;
      }
    };
    switch ((ntiles - 1) & 3) {                      // uniform: the ring slot of the last unit
      case 0: finish(ring[0]); break;
      case 1: finish(ring[1]); break;
      case 2: finish(ring[2]); break;
      default: finish(ring[3]); break;
    }
    if constexpr (NP == 3) load_pass(c0n, pq_raw, zq_frag);
    // epilogue: dP_q out; db_q = sum_t dP_q[t][:] (rows t >= T are exact zeros), dw_v, db_v partials of this
    // (sample, level): in-lane sums over the accumulator rows, then the two lane halves
    {
      int hrow = 4 * h * d;
      asm volatile("" : "+v"(hrow));
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        // (through an opaque scalar: given bit_cast(accq[g]) directly, hipcc (ROCm 7.2) stored element 0 sixteen
        // times -- caught by the parity tests)
        float v = accq[g];
        asm volatile("" : "+v"(v));
        store_dp(rs_dpq, v, hrow + ((g & 3) + 8 * (g >> 2)) * d, c0);
        s += v;
      }
      s = half_sum(s);
      dwacc = half_sum(dwacc);
      dbacc = half_sum(dbacc);
      if (h == 0) {
        a.dbq_part[pair * (size_t)d + c0 + r] = s;
        a.dwv_part[pair * (size_t)d + c0 + r] = dwacc;
        a.dbv_part[pair * (size_t)d + c0 + r] = dbacc;
      }
    }
  }
}

// dC = P_q dZ_v^T + dZ_q P_v^T, dA = dC (.) (1 - C^2)   -- the part of the backward that contracts over the CHANNELS, in
// the orientation [channels][locations], on the bf16 MFMA with the exact 3-way split.
// bwd_dc32_kernel<NT,NW>: one workgroup per (sample, level); the NW waves split the channels (128-channel slices, units
// of 32), the location tiles are taken in groups of up to four (their dC accumulators: 64 registers).  A unit:
//   * the transposed fragment P_v^T[k0 + crow(g, h)][32 nt + r] (16-byte loads, a lane per location row) is the
//     accumulator of H_v^T = P_v^T + P_q^T C (A = P_q^T in its natural lane = channel order, B = C^T rows of the image)
//     and, split, the B operand of dC += dZ_q P_v^T (contraction over the fragment's row index = channels);
//   * dZ_v^T = 4 ds_v r (1 - r) in place (w_v is folded into the P_q operand), split, is the B operand of
//     dC += (P_q w_v) dZ_v^T.
// The saved P_v, P_q carry the factor kPScale (fused.h): right for the exponential, divided out of the two dC operands.
// Cross-wave sum per location tile through LDS in a fixed order; rows t >= T / n >= N: loads 0, stores dropped.
#ifndef DC32_KO
#define DC32_KO 0   // developer knock-outs of bwd_dc32_kernel (wrong results; tools/ab_dc32.sh): 1 the tanh' arithmetic (one VALU
#endif              // operation per element instead of five), 2 the operand splits (one conversion per pair), 4 the MFMAs, 8 the P_v fragment loads
template <int NT, int NW, int NP>
__global__ __launch_bounds__(NW * 64, 2) void bwd_dc32_kernel(const BwdArgs a) {
  constexpr int NPAD = 32 * NT, PIECE = NPAD * 32, NTHR = NW * 64, SLD = 36;
  constexpr int GT = NT > COATTN_DC_GT ? COATTN_DC_GT : NT;   // location tiles per group
  extern __shared__ __attribute__((aligned(16))) char smem[];
  short* Cimg = reinterpret_cast<short*>(smem);
  float* dsvs = reinterpret_cast<float*>(smem + 3 * PIECE * 2);      // [NPAD] ds_v, zero padded
  float* slots = dsvs + NPAD;                                        // [NW][32 n][SLD]: one location tile per wave
  int b, l;
  if (!block_to_pair(blockIdx.x, a.B, a.L, b, l)) return;
  const int N = a.N, T = a.T, d = a.d;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int tid = w * 64 + lane, r = lane & 31, h = lane >> 5;
  const size_t pair = (size_t)l * a.B + b;
  const __amdgpu_buffer_rsrc_t rs_pv = make_rsrc(a.Pv + (size_t)b * N * d, (unsigned)N * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_pq = make_rsrc(a.Pq + pair * (size_t)T * d, (unsigned)T * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_hq = make_rsrc(a.Hq + pair * (size_t)T * d, (unsigned)T * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(a.C + pair * (size_t)T * N, (unsigned)T * N * 4u);
  const float dsq_r = a.dsq[pair * 32 + r];          // ds_q of this lane's token (0 for t >= T)
  const __amdgpu_buffer_rsrc_t rs_da = make_rsrc(a.dA + pair * (size_t)T * N, (unsigned)T * N * 4u);
  const int nsl = d / (128 * NW);
  // ---- the image of C (as in bwd_nat32_kernel) and ds_v
  {
    constexpr int PER = 8 * NPAD / NTHR;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int e = tid + k * NTHR, tq = e / NPAD, n = e - tq * NPAD;
      const int cvoff = n < N ? (4 * tq * N + n) * 4 : 0x40000000;
      float c[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = buf_load1(rs_c, cvoff, i * N * 4);
      unsigned hh[2] = {0, 0}, mm[2] = {0, 0}, ll[2] = {0, 0};
      split_pair<NP>(c[0], c[1], hh[0], mm[0], ll[0]);
      split_pair<NP>(c[2], c[3], hh[1], mm[1], ll[1]);
      const int off = n * 32 + 8 * ((tq >> 1) ^ ((n >> 2) & 3)) + 4 * (tq & 1);
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<u32x2*>(Cimg + off) = u32x2{hh[0], hh[1]};
      if (NP >= 2) *reinterpret_cast<u32x2*>(Cimg + PIECE + off) = u32x2{mm[0], mm[1]};
      if (NP == 3) *reinterpret_cast<u32x2*>(Cimg + 2 * PIECE + off) = u32x2{ll[0], ll[1]};
    }
    if (w == NW - 1) softmax_bwd_v(a, b, l, lane, dsvs, NPAD);   // ds_v of this (sample, level), as bwd_nat32_kernel computes it
  }
  lds_barrier();
  const int ntiles = (N + 31) >> 5;
  const int rk = (r >> 2) & 3;
  auto read_ca = [&](const short* img, const int ks, bf16x8 (&ca)[3]) {      // C^T rows: lane = location, 8 tokens
#pragma unroll
    for (int p = 0; p < NP; ++p)
      ca[p] = *reinterpret_cast<const bf16x8*>(img + p * PIECE + r * 32 + 8 * ((2 * ks + h) ^ rk));
  };
  auto split16 = [&](const f32x16& x, bf16x8 (&p0)[3], bf16x8 (&p1)[3]) {
    if (DC32_KO & 2) {                                // knock-out: one conversion per pair, every piece the same
      bf16x8 q0[3], q1[3];
      splitn<1>(f32x8{x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]}, q0);
      splitn<1>(f32x8{x[8], x[9], x[10], x[11], x[12], x[13], x[14], x[15]}, q1);
      for (int p = 0; p < 3; ++p) { p0[p] = q0[0]; p1[p] = q1[0]; }
      return;
    }
    splitn<NP>(f32x8{x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]}, p0);
    splitn<NP>(f32x8{x[8], x[9], x[10], x[11], x[12], x[13], x[14], x[15]}, p1);
  };
  // the transposed fragment of tile nt, channels k0 ..: register 4 c + j <-> channel k0 + 8 c + 4 h + j = k0 + crow
  auto load_frag = [&](int nt, int k0, bool live, f32x16& x) {   // !live: out-of-range addresses (zeros, no traffic)
    if ((DC32_KO & 8) && nt > 0) return;              // knock-out: the first tile's fragment stands for all
    const int vo = live ? (r * d + 4 * h) * 4 : 0x40000000;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 v = buf_load4(rs_pv, vo, (32 * nt * d + k0 + 8 * c) * 4);
      x[4 * c] = v[0]; x[4 * c + 1] = v[1]; x[4 * c + 2] = v[2]; x[4 * c + 3] = v[3];
    }
  };
  constexpr float kInv = 1.0f / kPScale;

  // A group of gt location tiles from tile t0 on.  gt is a compile-time value: the unit's tiles are then ONE basic block,
  // and the P_v fragments really run two tiles ahead, across unit boundaries too (behind a runtime `break` per tile the
  // compiler sank every fragment load to its first use: a full memory latency per tile with two waves per SIMD).
  auto group = [&](const int t0, auto gtc) __attribute__((always_inline)) {
    constexpr int gt = decltype(gtc)::value;
    const int ncu = 4 * nsl;
    auto unit_k0 = [&](int cu) { return ((cu >> 2) * NW + w) * 128 + 32 * (cu & 3); };
    auto flat_frag = [&](int f, f32x16& x) {         // fragment f of the group's (unit, tile) sequence; past its end: no traffic
      const int cu2 = f / gt;
      load_frag(t0 + (f - cu2 * gt), unit_k0(cu2), cu2 < ncu, x);
    };
    f32x16 dC[gt];
#pragma unroll
    for (int ti = 0; ti < gt; ++ti)
#pragma unroll
      for (int g = 0; g < 16; ++g) dC[ti][g] = 0.f;
    f32x16 cur, nxt, nx2;
    flat_frag(0, nxt);
    flat_frag(1, nx2);
#pragma unroll 1
    for (int cu = 0; cu < ncu; ++cu) {               // this wave's channel units of 32
      const int k0 = unit_k0(cu);
      bf16x8 pqB[2][3], pqA[2][3], zqA[2][3];
      {
        f32x8 raw[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 8; ++i) raw[ks][i] = buf_load1(rs_pq, (8 * h * d + r) * 4, (k0 + (16 * ks + i) * d) * 4);
        f32x4 qa[2][2], za[2][2], wa[2][2], wz[2][2];   // lane = token r: channels k0 + 16 ks + 4 h + {0..3}, + 8
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            qa[ks][u] = buf_load4(rs_pq, (r * d + 4 * h) * 4, (k0 + 16 * ks + 8 * u) * 4);
            za[ks][u] = buf_load4(rs_hq, (r * d + 4 * h) * 4, (k0 + 16 * ks + 8 * u) * 4);
            wa[ks][u] = *reinterpret_cast<const f32x4*>(a.wv + k0 + 16 * ks + 8 * u + 4 * h);
            wz[ks][u] = *reinterpret_cast<const f32x4*>(a.wq + k0 + 16 * ks + 8 * u + 4 * h);
          }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          splitn<NP>(raw[ks], pqB[ks]);
          f32x8 x, z;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            x[i] = qa[ks][i >> 2][i & 3] * kInv * wa[ks][i >> 2][i & 3];
            const float hq = za[ks][i >> 2][i & 3];                   // dZ_q = ds_q w_q (1 - H_q^2), as bwd_nat32_kernel forms it
            z[i] = dsq_r * wz[ks][i >> 2][i & 3] * (1.0f - hq * hq) * kInv;
          }
          splitn<NP>(x, pqA[ks]);
          splitn<NP>(z, zqA[ks]);
        }
      }
#pragma unroll
      for (int ti = 0; ti < gt; ++ti) {
        const int nt = t0 + ti;
        cur = nxt;
        nxt = nx2;
        flat_frag(cu * gt + ti + 2, nx2);
        __builtin_amdgcn_sched_barrier(0);           // (the request stays HERE, two tiles ahead of its use)
        const short* img = Cimg + 32 * nt * 32;
        bf16x8 F0[3], F1[3], Z0[3], Z1[3], ca[3];
        split16(cur, F0, F1);
        read_ca(img, 0, ca);
        if (!(DC32_KO & 4)) cur = mfma32_xn<NP>(pqB[0], ca, cur);
        read_ca(img, 1, ca);
        if (!(DC32_KO & 4)) cur = mfma32_xn<NP>(pqB[1], ca, cur);
        const float ds4 = 4.0f * dsvs[32 * nt + r];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          if (DC32_KO & 1) { cur[g] = cur[g] * ds4; continue; }
          const float rr = sig2_scaled(cur[g]);
          cur[g] = fmaf(-rr, rr, rr) * ds4;
        }
        split16(cur, Z0, Z1);
        if (!(DC32_KO & 4)) {
          dC[ti] = mfma32_xn<NP>(pqA[0], Z0, dC[ti]);
          dC[ti] = mfma32_xn<NP>(pqA[1], Z1, dC[ti]);
          dC[ti] = mfma32_xn<NP>(zqA[0], F0, dC[ti]);
          dC[ti] = mfma32_xn<NP>(zqA[1], F1, dC[ti]);
        } else {                                     // knock-out: every operand still has to exist
#pragma unroll
          for (int p = 0; p < NP; ++p)
            asm volatile("" ::"v"(Z0[p]), "v"(Z1[p]), "v"(F0[p]), "v"(F1[p]), "v"(ca[p]));
#pragma unroll
          for (int p = 0; p < NP; ++p)
            asm volatile("" ::"v"(pqA[0][p]), "v"(pqA[1][p]), "v"(zqA[0][p]), "v"(zqA[1][p]), "v"(pqB[0][p]), "v"(pqB[1][p]));
          dC[ti][0] += cur[0];
        }
      }
    }
    // ---- the group's tiles: cross-wave sum in a fixed order, dA = dC (1 - C^2)
#pragma unroll
    for (int ti = 0; ti < gt; ++ti) {
      const int nt = t0 + ti;
      float* mine = slots + w * 32 * SLD;
#pragma unroll
      for (int gg = 0; gg < 4; ++gg)                 // registers 4 gg .. 4 gg + 3: four consecutive tokens of location r
        *reinterpret_cast<f32x4*>(&mine[r * SLD + 8 * gg + 4 * h]) =
            f32x4{dC[ti][4 * gg], dC[ti][4 * gg + 1], dC[ti][4 * gg + 2], dC[ti][4 * gg + 3]};
      lds_barrier();
      for (int tq = tid >> 5; tq < 8; tq += NTHR / 32) {
        const int n = tid & 31;
        f32x4 s = *reinterpret_cast<const f32x4*>(&slots[n * SLD + 4 * tq]);
        s += *reinterpret_cast<const f32x4*>(&slots[32 * SLD + n * SLD + 4 * tq]);
        if constexpr (NW == 4) {
          f32x4 s2 = *reinterpret_cast<const f32x4*>(&slots[2 * 32 * SLD + n * SLD + 4 * tq]);
          s2 += *reinterpret_cast<const f32x4*>(&slots[3 * 32 * SLD + n * SLD + 4 * tq]);
          s += s2;
        }
        const int col = 32 * nt + n;
        const int voff = col < N ? (4 * tq * N + col) * 4 : 0x40000000;
        float c[4];                                  // (all four requested before the first store: the compiler
#pragma unroll                                       //  orders a load after a store it cannot tell apart from it)
        for (int i = 0; i < 4; ++i) c[i] = buf_load1(rs_c, voff, i * N * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = s[i] * (1.0f - c[i] * c[i]);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_da, voff, i * N * 4, 0);
        }
      }
      lds_barrier();
    }
  };
  static_assert(GT <= 4, "remainder groups of up to three tiles");
  int t0 = 0;
#pragma unroll 1
  for (; t0 + GT <= ntiles; t0 += GT) group(t0, std::integral_constant<int, GT>());
  const int rest = ntiles - t0;
  if (GT > 1 && rest == 1) group(t0, std::integral_constant<int, 1>());
  if (GT > 2 && rest == 2) group(t0, std::integral_constant<int, 2>());
  if (GT > 3 && rest == 3) group(t0, std::integral_constant<int, 3>());
}

template <int NT, int NW, int NP>
int launch_dc32(const BwdArgs& a, hipStream_t s) {
  constexpr int NPAD = 32 * NT;
  const size_t lds = (size_t)3 * NPAD * 32 * 2 + (size_t)NPAD * 4 + (size_t)NW * 32 * 36 * 4;
  const int groups = (a.B + 7) / 8;
  hipLaunchKernelGGL((bwd_dc32_kernel<NT, NW, NP>), dim3(groups * a.L * 8), dim3(NW * 64), lds, s, a);
  CA_CHECK_LAUNCH("bwd_dc32");
  return 0;
}

// dQ_l[b][t][k] (+)= a_q,l[t] gq_l[k] + sum_n dA_l[t][n] V[b][n][k]   (the image-feature part of dQ, added onto the
// dP_q W_q the projection GEMM has already written there when a.accumulate is set), location-major V.  One workgroup per (sample, 128-channel slice, level), 4 waves: a wave owns 32 channels and walks the
// location tiles; the V fragment (accumulator-shaped: 128 contiguous bytes per half wave and load, three tiles ahead)
// is split and used as the B operand (contraction over its row index = locations), the A operand is dA_l, split
// once per workgroup into an LDS image [piece][t][n] whose n order inside every group of 16 is the accumulator row
// order of a lane half (one 16-byte read per piece and k-step).
template <int NT, bool LM, int NP>
__global__ __launch_bounds__(256) void bwd_dq32_kernel(const DqArgs a) {
  constexpr int NPAD = 32 * NT;
  constexpr int LDR = NPAD + 8;                      // image row stride (bf16): 8 consecutive rows cover the banks once
  constexpr int PIECE = kTRows * LDR;                // rows t < 28 only: four workgroups fit a CU's LDS (lanes of rows
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 28 .. 31 read the next piece / the tail: finite, dropped)
  short* img = reinterpret_cast<short*>(smem);
  float* aqs = reinterpret_cast<float*>(smem + 3 * PIECE * 2 + 4 * LDR * 2);
  // blocks i and i + 8 share an XCD (round-robin dispatch): the L levels of one (sample, channel slice) take
  // consecutive slots of one XCD, so that V comes from HBM once and from that XCD's L2 for the other levels
  {
    const int main_blocks = ((a.B * (a.d / 128) + 7) / 8) * a.L * 8;
    if ((int)blockIdx.x >= main_blocks) { reduce_partials4_block(a, (int)blockIdx.x - main_blocks); return; }
  }
  int item, l;
  if (!block_to_pair(blockIdx.x, a.B * (a.d / 128), a.L, item, l)) return;
  const int nslice = a.d / 128, b = item / nslice, slice = item - b * nslice;
  const int N = a.N, T = a.T, d = a.d;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const size_t pair = (size_t)l * a.B + b;
  const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(a.V + (size_t)b * a.v_sB, (unsigned)N * d * 4u);   // either layout: N d floats
  const __amdgpu_buffer_rsrc_t rs_da = make_rsrc(a.dA + pair * (size_t)T * N, (unsigned)T * N * 4u);
  const int c0 = slice * 128 + 32 * w;
  // fragment of tile nt: register g of lane (r, h) <-> V[location 32 nt + crow(g, h)][channel c0 + r].  Location-major
  // V [N][d]: dword loads, 128 contiguous bytes per half wave (rows >= N read 0).  Channel-major V [d][N] (N % 4 == 0):
  // the lane's channel row, four consecutive locations per 16-byte load (locations >= N read the next row's head:
  // finite junk against zero columns of the dA image).
  auto load_tile = [&](int nt, f32x16& dst) {
    if constexpr (LM) {
#pragma unroll
      for (int g = 0; g < 16; ++g) dst[g] = buf_load1(rs_v, (crow(g, h) * d + c0 + r) * 4, 32 * nt * d * 4);
    } else {
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        const f32x4 v = buf_load4(rs_v, ((c0 + r) * N + 4 * h) * 4, (32 * nt + 8 * gg) * 4);
        dst[4 * gg] = v[0]; dst[4 * gg + 1] = v[1]; dst[4 * gg + 2] = v[2]; dst[4 * gg + 3] = v[3];
      }
    }
  };
  const int ntiles = (N + 31) >> 5;
  f32x16 ring[4];                                    // three tiles in flight (a wave has <= 7 of them); <= 120 VGPRs keep
  load_tile(0, ring[0]);                             // four workgroups on a CU
  load_tile(1, ring[1]);
  load_tile(2, ring[2]);
  // ---- dA_l -> image: a thread takes (t, two neighbouring n); rows >= T and columns >= N read 0
  constexpr int PER = (kTRows * (NPAD / 2) + 255) / 256;   // items per thread
  float x0[PER], x1[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {                    // all the loads first: one memory latency, not PER of them
    const int e = tid + 256 * k, t = e / (NPAD / 2), n = 2 * (e - t * (NPAD / 2));   // t >= 28: outside the buffer
    x0[k] = buf_load1(rs_da, n < N ? (t * N + n) * 4 : 0x40000000, 0);
    x1[k] = buf_load1(rs_da, n + 1 < N ? (t * N + n + 1) * 4 : 0x40000000, 0);
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int e = tid + 256 * k, t = e / (NPAD / 2), n = 2 * (e - t * (NPAD / 2));
    unsigned hh = 0, mm = 0, ll = 0;
    split_pair<NP>(x0[k], x1[k], hh, mm, ll);
    const int m16 = n & 15;                          // n order inside a group of 16: bits 2 and 3 trade places
    const int off = t * LDR + (n & ~15) + ((m16 & 3) | ((m16 & 4) << 1) | ((m16 & 8) >> 1));
    if (t < kTRows) {                                // (the last sweep is partial)
      *reinterpret_cast<unsigned*>(img + off) = hh;
      if (NP >= 2) *reinterpret_cast<unsigned*>(img + PIECE + off) = mm;
      if (NP == 3) *reinterpret_cast<unsigned*>(img + 2 * PIECE + off) = ll;
    }
  }
  if (tid < 32) aqs[tid] = tid < T ? a.aq[pair * (size_t)T + tid] : 0.f;
  lds_barrier();
  f32x16 acc;
#pragma unroll
  for (int g = 0; g < 16; ++g) acc[g] = 0.f;
  auto tile = [&](int nt, const f32x16& v) {
    bf16x8 b0[3], b1[3], a0[3], a1[3];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      a0[p] = *reinterpret_cast<const bf16x8*>(img + p * PIECE + r * LDR + 32 * nt + 8 * h);
      a1[p] = *reinterpret_cast<const bf16x8*>(img + p * PIECE + r * LDR + 32 * nt + 16 + 8 * h);
    }
    splitn<NP>(f32x8{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]}, b0);
    splitn<NP>(f32x8{v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]}, b1);
    acc = mfma32_xn<NP>(a0, b0, acc);
    acc = mfma32_xn<NP>(a1, b1, acc);
  };
#pragma unroll 1
  for (int nt = 0; nt < ntiles; nt += 4) {
    load_tile(nt + 3, ring[3]);
    tile(nt, ring[0]);
    if (nt + 1 < ntiles) { load_tile(nt + 4, ring[0]); tile(nt + 1, ring[1]); }
    if (nt + 2 < ntiles) { load_tile(nt + 5, ring[1]); tile(nt + 2, ring[2]); }
    if (nt + 3 < ntiles) { load_tile(nt + 6, ring[2]); tile(nt + 3, ring[3]); }
  }
  // dQ_l[b][t][c0 + r] = acc + a_q[t] gq[c0 + r]; rows t >= T lie outside the buffer
  const __amdgpu_buffer_rsrc_t rs_dq = make_rsrc(a.dQ[l] + (size_t)b * T * d, (unsigned)T * d * 4u);
  const float gqv = a.gq[pair * (size_t)d + c0 + r];
  float prev[16];
#pragma unroll
  for (int g = 0; g < 16; ++g)
    prev[g] = a.accumulate ? buf_load1(rs_dq, (crow(g, h) * d + c0 + r) * 4, 0) : 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    float v = fmaf(aqs[crow(g, h)], gqv, acc[g]) + prev[g];
    asm volatile("" : "+v"(v));                      // (opaque scalar: see bwd_nat32_kernel's epilogue)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_dq, (crow(g, h) * d + c0 + r) * 4, 0, 0);
  }
}

// The same for ALL levels of a (sample, channel slice) in one workgroup -- small location counts only (N <= 64: the
// dA images of the three levels together are 36 KB): the V fragments are loaded and split ONCE and serve the three
// levels' MFMAs (the split is what bwd_dq32_kernel spends its issue slots on: 88 VALU per 12 MFMAs there, per 36 here),
// and V is read once per sample instead of once per level.
template <int NT, bool LM, int NP>
__global__ __launch_bounds__(256) void bwd_dq32x_kernel(const DqArgs a) {
  static_assert(NT <= 2, "the images of all levels must fit the LDS of several workgroups per CU");
  constexpr int NPAD = 32 * NT;
  constexpr int LDR = NPAD + 8;
  constexpr int PIECE = kTRows * LDR;
  constexpr int LEVEL = 3 * PIECE + 4 * LDR;         // + 4 rows: what lanes 28 .. 31 of the last piece read
  constexpr int ML = 3;                              // levels (L <= 3)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  short* img = reinterpret_cast<short*>(smem);
  float* aqs = reinterpret_cast<float*>(smem + ML * LEVEL * 2);
  if ((int)blockIdx.x >= a.B * (a.d / 128)) { reduce_partials4_block(a, (int)blockIdx.x - a.B * (a.d / 128)); return; }
  const int nslice = a.d / 128, item = blockIdx.x, b = item / nslice, slice = item - b * nslice;
  const int N = a.N, T = a.T, d = a.d, L = a.L;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(a.V + (size_t)b * a.v_sB, (unsigned)N * d * 4u);
  const int c0 = slice * 128 + 32 * w;
  f32x16 vt[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {                  // (the fragment layouts of bwd_dq32_kernel)
    if constexpr (LM) {
#pragma unroll
      for (int g = 0; g < 16; ++g) vt[nt][g] = buf_load1(rs_v, (crow(g, h) * d + c0 + r) * 4, 32 * nt * d * 4);
    } else {
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        const f32x4 v = buf_load4(rs_v, ((c0 + r) * N + 4 * h) * 4, (32 * nt + 8 * gg) * 4);
        vt[nt][4 * gg] = v[0]; vt[nt][4 * gg + 1] = v[1]; vt[nt][4 * gg + 2] = v[2]; vt[nt][4 * gg + 3] = v[3];
      }
    }
  }
  // ---- dA_l -> images, all levels: a thread takes (level, t, two neighbouring n); rows >= T and columns >= N read 0
  constexpr int PER = (kTRows * (NPAD / 2) + 255) / 256;
  float x0[ML][PER], x1[ML][PER];
#pragma unroll
  for (int l = 0; l < ML; ++l) {
    const __amdgpu_buffer_rsrc_t rs_da = make_rsrc(a.dA + ((size_t)(l < L ? l : 0) * a.B + b) * (size_t)T * N, l < L ? (unsigned)T * N * 4u : 0u);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int e = tid + 256 * k, t = e / (NPAD / 2), n = 2 * (e - t * (NPAD / 2));
      x0[l][k] = buf_load1(rs_da, n < N ? (t * N + n) * 4 : 0x40000000, 0);
      x1[l][k] = buf_load1(rs_da, n + 1 < N ? (t * N + n + 1) * 4 : 0x40000000, 0);
    }
  }
#pragma unroll
  for (int l = 0; l < ML; ++l)
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int e = tid + 256 * k, t = e / (NPAD / 2), n = 2 * (e - t * (NPAD / 2));
      unsigned hh = 0, mm = 0, ll = 0;
      split_pair<NP>(x0[l][k], x1[l][k], hh, mm, ll);
      const int m16 = n & 15;                        // n order inside a group of 16: bits 2 and 3 trade places
      const int off = l * LEVEL + t * LDR + (n & ~15) + ((m16 & 3) | ((m16 & 4) << 1) | ((m16 & 8) >> 1));
      if (t < kTRows) {
        *reinterpret_cast<unsigned*>(img + off) = hh;
        if (NP >= 2) *reinterpret_cast<unsigned*>(img + PIECE + off) = mm;
        if (NP == 3) *reinterpret_cast<unsigned*>(img + 2 * PIECE + off) = ll;
      }
    }
  if (tid < 32 * ML) {
    const int l = tid >> 5, t = tid & 31;
    aqs[tid] = (l < L && t < T) ? a.aq[((size_t)l * a.B + b) * (size_t)T + t] : 0.f;
  }
  lds_barrier();
  f32x16 acc[ML];
#pragma unroll
  for (int l = 0; l < ML; ++l)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[l][g] = 0.f;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    bf16x8 b0[3], b1[3];
    splitn<NP>(f32x8{vt[nt][0], vt[nt][1], vt[nt][2], vt[nt][3], vt[nt][4], vt[nt][5], vt[nt][6], vt[nt][7]}, b0);
    splitn<NP>(f32x8{vt[nt][8], vt[nt][9], vt[nt][10], vt[nt][11], vt[nt][12], vt[nt][13], vt[nt][14], vt[nt][15]}, b1);
#pragma unroll
    for (int l = 0; l < ML; ++l) {
      bf16x8 a0[3], a1[3];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        a0[p] = *reinterpret_cast<const bf16x8*>(img + l * LEVEL + p * PIECE + r * LDR + 32 * nt + 8 * h);
        a1[p] = *reinterpret_cast<const bf16x8*>(img + l * LEVEL + p * PIECE + r * LDR + 32 * nt + 16 + 8 * h);
      }
      acc[l] = mfma32_xn<NP>(a0, b0, acc[l]);
      acc[l] = mfma32_xn<NP>(a1, b1, acc[l]);
    }
  }
  // dQ_l[b][t][c0 + r] (+)= acc_l + a_q,l[t] gq_l[c0 + r]; rows t >= T lie outside the buffer
#pragma unroll
  for (int l = 0; l < ML; ++l) {
    if (l >= L) break;
    const __amdgpu_buffer_rsrc_t rs_dq = make_rsrc(a.dQ[l] + (size_t)b * T * d, (unsigned)T * d * 4u);
    const float gqv = a.gq[((size_t)l * a.B + b) * (size_t)d + c0 + r];
    float prev[16];
#pragma unroll
    for (int g = 0; g < 16; ++g)
      prev[g] = a.accumulate ? buf_load1(rs_dq, (crow(g, h) * d + c0 + r) * 4, 0) : 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      float v = fmaf(aqs[32 * l + crow(g, h)], gqv, acc[l][g]) + prev[g];
      asm volatile("" : "+v"(v));                    // (opaque scalar: see bwd_nat32_kernel's epilogue)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_dq, (crow(g, h) * d + c0 + r) * 4, 0, 0);
    }
  }
}

template <int NT, bool LM, int NP>
int launch_dq32x(const DqArgs& a, hipStream_t s) {
  constexpr int NPAD = 32 * NT;
  const size_t lds = (size_t)3 * ((3 * kTRows + 4) * (NPAD + 8) * 2) + 3 * 32 * 4;
  hipLaunchKernelGGL((bwd_dq32x_kernel<NT, LM, NP>), dim3(a.B * (a.d / 128) + a.red_jobs * a.red_blocks), dim3(256), lds, s, a);
  CA_CHECK_LAUNCH("bwd_dq32x");
  return 0;
}

template <int NT, bool LM, int NP>
int launch_dq32(const DqArgs& a, hipStream_t s) {
  constexpr int NPAD = 32 * NT;
  const size_t lds = (size_t)(3 * kTRows + 4) * (NPAD + 8) * 2 + 32 * 4;   // + 4 rows: what lanes 28 .. 31 of the last piece read
  const int items = a.B * (a.d / 128);
  hipLaunchKernelGGL((bwd_dq32_kernel<NT, LM, NP>), dim3(((items + 7) / 8) * a.L * 8 + a.red_jobs * a.red_blocks), dim3(256), lds, s, a);
  CA_CHECK_LAUNCH("bwd_dq32");
  return 0;
}

template <int NT, int NW, int NP, bool DPB>
int launch_nat32(const BwdArgs& a, hipStream_t s) {
  constexpr int NPAD = 32 * NT;
  const size_t lds = (size_t)3 * NPAD * 32 * 2 + (size_t)NPAD * 4 + 32 * 4;
  const int groups = (a.B + 7) / 8;
  hipLaunchKernelGGL((bwd_nat32_kernel<NT, NW, NP, DPB>), dim3(groups * a.L * 8), dim3(NW * 64), lds, s, a);
  CA_CHECK_LAUNCH("bwd_nat32");
  return 0;
}

}  // namespace

// The reduced-precision mode (a.bf16: one MFMA per product) exists for the four-wave kernels (d % 512 == 0); other widths
// run the fp32 mode.  a.np = 2 selects the two-piece width of the fp32 mode (fused.h), anything else the exact split.
int launch_bwd_nat32(const BwdArgs& a, hipStream_t s) {
  const bool small_n = a.N <= 64, w2 = a.np == 2;
  if (a.d % 512 == 0) {
    if (a.bf16 && a.dp_bf16) return small_n ? launch_nat32<2, 4, 1, true>(a, s) : launch_nat32<7, 4, 1, true>(a, s);
    CA_CHECK_ARG(!a.dp_bf16, "bwd_nat32: bf16 dP storage exists in the reduced-precision mode only");
    if (a.bf16) return small_n ? launch_nat32<2, 4, 1, false>(a, s) : launch_nat32<7, 4, 1, false>(a, s);
    if (w2) return small_n ? launch_nat32<2, 4, 2, false>(a, s) : launch_nat32<7, 4, 2, false>(a, s);
    return small_n ? launch_nat32<2, 4, 3, false>(a, s) : launch_nat32<7, 4, 3, false>(a, s);
  }
  CA_CHECK_ARG(!a.dp_bf16, "bwd_nat32: bf16 dP storage needs d % 512 == 0");
  if (w2) return small_n ? launch_nat32<2, 2, 2, false>(a, s) : launch_nat32<7, 2, 2, false>(a, s);
  return small_n ? launch_nat32<2, 2, 3, false>(a, s) : launch_nat32<7, 2, 3, false>(a, s);
}

int launch_bwd_dq32(const DqArgs& a, int lm, hipStream_t s) {
  static const int shared = dev_env_int("COATTN_DQ32X", 1);   // developer switch
  const int np = a.bf16 ? 1 : (a.np == 2 ? 2 : 3);
  auto go = [&](auto NPc) -> int {
    constexpr int NP = decltype(NPc)::value;
    if (a.N <= 64 && a.L <= 3 && shared) return lm ? launch_dq32x<2, true, NP>(a, s) : launch_dq32x<2, false, NP>(a, s);
    if (lm) return a.N <= 64 ? launch_dq32<2, true, NP>(a, s) : launch_dq32<7, true, NP>(a, s);
    return a.N <= 64 ? launch_dq32<2, false, NP>(a, s) : launch_dq32<7, false, NP>(a, s);
  };
  if (np == 1) return go(std::integral_constant<int, 1>());
  if (np == 2) return go(std::integral_constant<int, 2>());
  return go(std::integral_constant<int, 3>());
}

int launch_bwd_dc32(const BwdArgs& a, hipStream_t s) {
  const bool small_n = a.N <= 64, w2 = a.np == 2;
  if (a.d % 512 == 0) {
    if (a.bf16) return small_n ? launch_dc32<2, 4, 1>(a, s) : launch_dc32<7, 4, 1>(a, s);
    if (w2) return small_n ? launch_dc32<2, 4, 2>(a, s) : launch_dc32<7, 4, 2>(a, s);
    return small_n ? launch_dc32<2, 4, 3>(a, s) : launch_dc32<7, 4, 3>(a, s);
  }
  if (w2) return small_n ? launch_dc32<2, 2, 2>(a, s) : launch_dc32<7, 2, 2>(a, s);
  return small_n ? launch_dc32<2, 2, 3>(a, s) : launch_dc32<7, 2, 3>(a, s);
}
