// Fused backward, question side: dP_q = dZ_q + C dZ_v   (autograd of model.py:380-388; SURVEY.md section 8 "Backward").
//
// bwd_dpq32_kernel<NT,NW>: one workgroup per (sample, level), NW waves owning 128-channel slices, on the bf16 MFMA
// 32x32x16 with the exact 3-way split (fused.h) -- the backward twin of the forward kernel's phase 2
// (coattn_fwd32.hip), whose operand layouts it shares:
//   * C (saved) is split once into the LDS image [piece][n][32 t]; row reads give C^T fragments (A operand of
//     H_v = P_v + C^T P_q), transposing reads give C fragments (A operand of dP_q += C dZ_v);
//   * unit pipeline over 32-location x 32-channel fragments of P_v (accumulator-shaped, 128 contiguous bytes per half
//     wave and load, two units ahead): the fragment is the accumulator of the recomputed H_v tile; tanh and
//     dZ_v = ds_v w_v (1 - H_v^2) = 4 ds_v w_v r (1 - r), r = 1 / (1 + e^{2 H}), happen in place; split, the dZ_v
//     fragment is the B operand of dP_q += C dZ_v (contraction over the fragment's row index = locations);
//   * the dP_q accumulators start from dZ_q; db_q partials are in-lane sums over the accumulator rows.
// H_v is never stored.  Rows t >= T / n >= N fall outside the per-sample buffer descriptors (loads 0, stores dropped).
#include "fused.h"

namespace {

template <int NT, int NW>
__global__ __launch_bounds__(NW * 64, 2) void bwd_dpq32_kernel(const BwdArgs a) {
  constexpr int NPAD = 32 * NT;
  constexpr int PIECE = NPAD * 32;                   // bf16 elements of one piece of the C image [n][t = 32]
  constexpr int NTHR = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  short* Cimg = reinterpret_cast<short*>(smem);
  float* dsvs = reinterpret_cast<float*>(smem + 3 * PIECE * 2);   // [NPAD] ds_v, zero padded
  int b, l;
  if (!block_to_pair(blockIdx.x, a.B, a.L, b, l)) return;
  const int N = a.N, T = a.T, d = a.d;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int tid = w * 64 + lane, r = lane & 31, h = lane >> 5;
  const size_t pair = (size_t)l * a.B + b;
  const __amdgpu_buffer_rsrc_t rs_pv = make_rsrc(a.Pv + (size_t)b * N * d, (unsigned)N * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_pq = make_rsrc(a.Pq + pair * (size_t)T * d, (unsigned)T * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_dzq = make_rsrc(a.dZq + pair * (size_t)T * d, (unsigned)T * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_dpq = make_rsrc(a.dPq + pair * (size_t)T * d, (unsigned)T * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(a.C + pair * (size_t)T * N, (unsigned)T * N * 4u);
  const int nsl = d / (128 * NW);                    // 128-channel slices per wave (1 at d = 512)
  constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};   // smallest piece products first

  // B operands of H_v (P_q, k index t = 16 ks + 8 h + i in lane half h), raw; split when needed
  auto load_pq_nat = [&](int c0, f32x8 (&raw)[2][2]) {
    int base = (8 * h * d + r) * 4;
    asm volatile("" : "+v"(base));
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i) raw[ct][ks][i] = buf_load1(rs_pq, base + 128 * ct, (c0 + (16 * ks + i) * d) * 4);
  };
  f32x8 pq_raw[2][2];
  load_pq_nat(w * 128, pq_raw);                      // the first pass' operands fly under the image build

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
      unsigned hh[2], mm[2], ll[2];
      split3_pair(c[0], c[1], hh[0], mm[0], ll[0]);
      split3_pair(c[2], c[3], hh[1], mm[1], ll[1]);
      const int off = n * 32 + 8 * ((tq >> 1) ^ ((n >> 2) & 3)) + 4 * (tq & 1);
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<u32x2*>(Cimg + off) = u32x2{hh[0], hh[1]};
      *reinterpret_cast<u32x2*>(Cimg + PIECE + off) = u32x2{mm[0], mm[1]};
      *reinterpret_cast<u32x2*>(Cimg + 2 * PIECE + off) = u32x2{ll[0], ll[1]};
    }
    const float* dg = a.dsv + pair * (size_t)N;
    for (int e = tid; e < NPAD; e += NTHR) dsvs[e] = e < N ? dg[e] : 0.f;
  }
  lds_barrier();

  const int ntiles = (N + 31) >> 5;
  const int U = 2 * ntiles;
  // lane constants of the image reads (see coattn_fwd32.hip)
  const int tq = (lane & 15) >> 2, tp_ = lane & 3, g1 = (lane >> 4) & 1;
  const int tr_off0 = (4 * h + tq) * 32 + 8 * ((2 * g1 + (tp_ >> 1)) ^ h) + 4 * (tp_ & 1);
  const int tr_off1 = (4 * h + 8 + tq) * 32 + 8 * ((2 * g1 + (tp_ >> 1)) ^ (h + 2)) + 4 * (tp_ & 1);
  const int rk = (r >> 2) & 3;
  auto read_cq = [&](const short* img, const int s2, bf16x8 (&cq)[3]) {     // A = C (tokens x locations)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const bf16x4 lo = lds_tr16(img + p * PIECE + 16 * s2 * 32 + tr_off0);
      const bf16x4 hi = lds_tr16(img + p * PIECE + 16 * s2 * 32 + tr_off1);
      cq[p] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };
  auto read_ca = [&](const short* img, const int ks, bf16x8 (&ca)[3]) {      // A = C^T (locations x tokens)
#pragma unroll
    for (int p = 0; p < 3; ++p)
      ca[p] = *reinterpret_cast<const bf16x8*>(img + p * PIECE + r * 32 + 8 * ((2 * ks + h) ^ rk));
  };

#pragma unroll 1
  for (int pi = 0; pi < 2 * nsl; ++pi) {             // passes: 64 channels each, two per 128-channel slice
    const int c0 = ((pi >> 1) * NW + w) * 128 + 64 * (pi & 1);
    const int c0n = (((pi + 1) >> 1) * NW + w) * 128 + 64 * ((pi + 1) & 1);
    const float wv4[2] = {4.0f * a.wv[c0 + r], 4.0f * a.wv[c0 + 32 + r]};
    bf16x8 pqB[2][2][3];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) split3(pq_raw[ct][ks], pqB[ct][ks]);
    // the dP_q accumulators start from dZ_q: register g of lane (r, h) <-> dZ_q[t = crow(g, h)][c0 + 32 ct + r]
    f32x16 accq[2];
    {
      int base = (4 * h * d + r) * 4;
      asm volatile("" : "+v"(base));
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g = 0; g < 16; ++g)
          accq[ct][g] = buf_load1(rs_dzq, base + 128 * ct, (c0 + ((g & 3) + 8 * (g >> 2)) * d) * 4);
    }
    f32x16 ring[4];
    u32x4 Ph[2], Pm[2], Pl[2];                       // split dZ_v of the previous unit (two k-steps of 16 locations)
    // pv[g] = P_v[32 nt + crow(g, h)][c0 + 32 ct + r]; past the last tile: zeros through the buffer rule, no traffic
    auto load_unit = [&](int u, f32x16& dst) {
#pragma unroll
      for (int g = 0; g < 16; ++g)
        dst[g] = buf_load1(rs_pv, (crow(g, h) * d + r) * 4, (32 * (u >> 1) * d + c0 + 32 * (u & 1)) * 4);
    };
    // One step = unit u.  First half: the 12 MFMAs of H_v(u) = P_v + C^T P_q on the fragment `cur`, each followed by a
    // share of the split of dZ_v(u-1) (`dzp`).  Second half: the 12 MFMAs of dP_q += C dZ_v(u-1), each followed by a
    // share of tanh / dZ_v of unit u, in place in `cur`.
    auto step = [&](const int u, const int ct, f32x16& cur, const f32x16& dzp) {
      const short* img = Cimg + 32 * (u >> 1) * 32;
      const short* imgp = Cimg + 32 * ((u > 0 ? u - 1 : 0) >> 1) * 32;
      bf16x8 ca0[3], ca1[3], cq0[3], cq1[3];
      f32x4 dsn[4];
      read_ca(img, 0, ca0);
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        const int ks = m / 6, i = m % 6;
        cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks ? ca1[PA[i]] : ca0[PA[i]], pqB[ct][ks][PB[i]], cur, 0, 0, 0);
        if (m == 1) read_ca(img, 1, ca1);           // operands are read one MFMA group ahead of their use
        if (m == 8) read_cq(imgp, 0, cq0);
        if (m == 11) {
#pragma unroll
          for (int gg = 0; gg < 4; ++gg) dsn[gg] = *reinterpret_cast<const f32x4*>(&dsvs[32 * (u >> 1) + 8 * gg + 4 * h]);
        }
        if (m < 8) {                                 // split pair m of dZ_v(u-1)
          unsigned hh, mm, ll;
          split3_pair(dzp[2 * m], dzp[2 * m + 1], hh, mm, ll);
          Ph[m >> 2][m & 3] = hh; Pm[m >> 2][m & 3] = mm; Pl[m >> 2][m & 3] = ll;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const int ctp = ct ^ 1;                        // unit u-1's channel half
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        const int ks = m / 6, i = m % 6;
        const bf16x8 bp = PB[i] == 0 ? __builtin_bit_cast(bf16x8, Ph[ks]) : PB[i] == 1 ? __builtin_bit_cast(bf16x8, Pm[ks])
                                                                                        : __builtin_bit_cast(bf16x8, Pl[ks]);
        accq[ctp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks ? cq1[PA[i]] : cq0[PA[i]], bp, accq[ctp], 0, 0, 0);
        if (m == 1) read_cq(imgp, 1, cq1);
        // dZ_v of unit u in place: two registers beside each of the first four MFMAs, one beside each of the others
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int g = m < 4 ? 2 * m + q : m + 4;
          if (q == 1 && m >= 4) continue;
          const float rr = sig2_fast(cur[g]);
          const float t = fmaf(-rr, rr, rr);         // r (1 - r) = (1 - tanh^2) / 4
          cur[g] = t * dsn[g >> 2][g & 3] * wv4[ct];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    load_unit(0, ring[0]);
    load_unit(1, ring[1]);
#pragma unroll
    for (int g = 0; g < 16; ++g) ring[3][g] = 0.f;   // "dZ_v of unit -1"
#pragma unroll 1
    for (int u0 = 0; u0 < U; u0 += 4) {              // U is even: units come in (ct = 0, ct = 1) pairs
      load_unit(u0 + 2, ring[2]);
      __builtin_amdgcn_sched_barrier(0);
      step(u0, 0, ring[0], ring[3]);
      __builtin_amdgcn_sched_barrier(0);
      load_unit(u0 + 3, ring[3]);
      __builtin_amdgcn_sched_barrier(0);
      step(u0 + 1, 1, ring[1], ring[0]);
      __builtin_amdgcn_sched_barrier(0);
      if (u0 + 2 < U) {
        load_unit(u0 + 4, ring[0]);
        __builtin_amdgcn_sched_barrier(0);
        step(u0 + 2, 0, ring[2], ring[1]);
        __builtin_amdgcn_sched_barrier(0);
        load_unit(u0 + 5, ring[1]);
        __builtin_amdgcn_sched_barrier(0);
        step(u0 + 3, 1, ring[3], ring[2]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the last unit's dZ_v (a second channel half, tile ntiles - 1): split and accumulate
    {
      const f32x16& dz = (U & 2) ? ring[1] : ring[3];
      const short* imgp = Cimg + 32 * (ntiles - 1) * 32;
      bf16x8 cq0[3], cq1[3], b0[3], b1[3];
      read_cq(imgp, 0, cq0);
      read_cq(imgp, 1, cq1);
      split3(f32x8{dz[0], dz[1], dz[2], dz[3], dz[4], dz[5], dz[6], dz[7]}, b0);
      split3(f32x8{dz[8], dz[9], dz[10], dz[11], dz[12], dz[13], dz[14], dz[15]}, b1);
      accq[1] = mfma32_x3(cq0, b0, accq[1]);
      accq[1] = mfma32_x3(cq1, b1, accq[1]);
    }
    // the next pass' P_q operands fly under the epilogue (beyond the last pass: channel offsets >= d, never used)
    load_pq_nat(c0n, pq_raw);
    // epilogue: dP_q out; db_q partial = sum_t dP_q[t][:] (rows t >= T are exact zeros)
    {
      int hrow = (4 * h * d + r) * 4;
      asm volatile("" : "+v"(hrow));
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          // (through an opaque scalar: given bit_cast(accq[ct][g]) directly, hipcc (ROCm 7.2) stored element 0 sixteen
          // times -- caught by the parity tests)
          float v = accq[ct][g];
          asm volatile("" : "+v"(v));
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_dpq,
                                                hrow + (((g & 3) + 8 * (g >> 2)) * d + 32 * ct) * 4, c0 * 4, 0);
          s += v;
        }
        s += __shfl_xor(s, 32, 64);
        if (h == 0) a.dbq_part[pair * (size_t)d + c0 + 32 * ct + r] = s;
      }
    }
  }
}

template <int NT, int NW>
int launch_dpq32(const BwdArgs& a, hipStream_t s) {
  constexpr int NPAD = 32 * NT;
  const size_t lds = (size_t)3 * NPAD * 32 * 2 + (size_t)NPAD * 4;
  const int groups = (a.B + 7) / 8;
  hipLaunchKernelGGL((bwd_dpq32_kernel<NT, NW>), dim3(groups * a.L * 8), dim3(NW * 64), lds, s, a);
  CA_CHECK_LAUNCH("bwd_dpq32");
  return 0;
}

}  // namespace

int launch_bwd_dpq32(const BwdArgs& a, hipStream_t s) {
  const bool small_n = a.N <= 64;
  if (a.d % 512 == 0) return small_n ? launch_dpq32<2, 4>(a, s) : launch_dpq32<7, 4>(a, s);
  return small_n ? launch_dpq32<2, 2>(a, s) : launch_dpq32<7, 2>(a, s);
}
