// Fused co-attention forward for gfx950, every contraction on the 16-bit MFMAs 32x32x16 through a split of the fp32
// operands into 16-bit pieces (fused.h): the exact 3-way bf16 split (x = hi + mid + lo, six partial products: fp32-accurate,
// the default, flags = 0), two FP16 pieces in BOTH phases (22 significand bits, three products: COATTN_FLAG_FAST16, template
// value NP_ = 4), or one bf16 piece (COATTN_FLAG_BF16_PROJ); fp32 accumulation throughout.
//
// "affinity + softmax + reduce" (model.py:377-392 after the projections), one workgroup per (sample b, level l),
// NW = d/128 waves (4 at d = 512), two workgroups per CU:
//   phase 1  A = Q V^T, K (= d) split over the waves: a wave streams its 128-channel slices of V and Q from HBM into a
//            per-wave LDS ring by LDS-DMA (location-major V [N][d]: units of 32 rows x 128 bytes, every DMA instruction
//            covers 8 whole cache lines; channel-major V [d][N]: units of 16 channel rows x 128 bytes) and reads them
//            back as MFMA operands one unit ahead; the [32 x N] partial tiles are summed across waves through LDS in a
//            fixed order; C = tanh(A) goes to `saved` and, split
//            once into its three bf16 pieces, into ONE LDS image [piece][n][t] (64-byte rows, 16-byte chunks
//            XOR-swizzled by (n >> 2) & 3) that serves both orientations conflict-free: row reads (ds_read_b128)
//            give C^T as the A operand of H_v, transposing reads (ds_read_b64_tr_b16) give C as the A operand of H_q.
//   phase 2  two passes per 128-channel slice (64 channels each), loop over 32-location tiles of P_v [N][d] (read
//            once): the raw tile is split and is the B operand of H_q += C . P_v (contraction over n: the
//            accumulator layout of a 32x32 tile IS the B-operand layout of the next MFMA), then the same registers
//            accumulate H_v = P_v + C^T P_q (contraction over t, one 32-deep step); the score
//            s_v[n] += H_v[n][:] . w_v is taken from 1 / (1 + e^{2 H}) without ever writing H_v (or the tanh), the
//            16 score registers of a tile are summed over their channel lanes by a transposing butterfly;
//            the H_q accumulators start from P_q.
//   phase 3  score reduction over waves / passes in a fixed order, un-masked row softmax over N and over T
//            (model.py:387-388) by wave shuffles, H_q saved, q = a_q^T Q.
//   attend_v_lm_kernel : v_l = a_{v,l}^T V for all levels with one more pass over a location-major V.
#include "fused.h"

#include <stdlib.h>

// Developer switches of the phase-1 experiments (diagnostic builds only).
#ifndef COATTN_P1_NODMA
#define COATTN_P1_NODMA 0      // developer switches of the phase-1 experiments (wrong results when set)
#endif
#ifndef COATTN_P1_NOMATH
#define COATTN_P1_NOMATH 0
#endif
#ifndef COATTN_P1_NB          // channel-major stream: tiles per k-step on the bf16 MFMA (the others on the f32 MFMA); -1 = all
#define COATTN_P1_NB -1
#endif
#ifndef COATTN_P1_RING        // location-major stream at N <= 64: slots of 4 KB in a wave's phase-1 ring (4: three units in flight)
#define COATTN_P1_RING 4
#endif

namespace {


// wait until at most n of this wave's vector-memory operations (loads, stores, LDS-DMA: one in-order counter) are
// still outstanding; hand-placed for the LDS-DMA ring, whose data dependences the compiler does not see.  n is a
// constant after unrolling: one s_waitcnt remains.
// (an "n"/"i" asm operand must be a constant before unrolling and __builtin_amdgcn_s_waitcnt would tell hipcc's own
//  counter bookkeeping about these waits, so the count is spelled out per case; the switch folds to one instruction)
__device__ __forceinline__ void vmcnt_wait(int n) {
#define CA_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    CA_VMW(0) CA_VMW(1) CA_VMW(2) CA_VMW(3) CA_VMW(4) CA_VMW(5) CA_VMW(6) CA_VMW(7) CA_VMW(8) CA_VMW(9) CA_VMW(10) CA_VMW(11) CA_VMW(12)
    CA_VMW(13) CA_VMW(14) CA_VMW(15) CA_VMW(16) CA_VMW(17) CA_VMW(18) CA_VMW(19) CA_VMW(20) CA_VMW(21) CA_VMW(22) CA_VMW(23) CA_VMW(24)
    CA_VMW(25) CA_VMW(26) CA_VMW(27) CA_VMW(28) CA_VMW(29) CA_VMW(30) CA_VMW(31) CA_VMW(32) CA_VMW(33) CA_VMW(34) CA_VMW(35) CA_VMW(36)
    CA_VMW(37) CA_VMW(38) CA_VMW(39) CA_VMW(40) CA_VMW(41) CA_VMW(42) CA_VMW(43) CA_VMW(44) CA_VMW(45) CA_VMW(46) CA_VMW(47) CA_VMW(48)
    CA_VMW(49) CA_VMW(50) CA_VMW(51) CA_VMW(52) CA_VMW(53) CA_VMW(54) CA_VMW(55) CA_VMW(56) CA_VMW(57) CA_VMW(58) CA_VMW(59) CA_VMW(60)
    CA_VMW(61) CA_VMW(62) CA_VMW(63)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef CA_VMW
}


// NP: width of the phase-2 contractions C^T P_q, C P_v (fused.h: 3 = the exact split, 2 = hi + mid bf16 pieces -- the latter
// only behind a developer switch: with bf16 pieces the affinity of phase 1 keeps the exact split, it is the one contraction
// whose error the tanh amplifies; the tolerance mode the library ships is NP_ = 4 below).
// NP = 1 (SP, single product): the reduced-precision mode (COATTN_FLAG_BF16_PROJ) -- every operand of BOTH phases rounded once
// to bf16, ONE MFMA per product (the hi x hi term of the split; the mid / lo pieces are neither computed, stored in the
// LDS image nor read back).
// NP_ = 4 (HF, the default of the fp32 mode): BOTH phases on two FP16 pieces per operand (fused.h: 22 significand bits, the
// three products lo*hi, hi*lo, hi*hi on v_mfma_f32_32x32x16_f16) -- every operand here is of ordinary magnitude (features,
// projections, tanh values); less error than the exact bf16 split of phase 1 next to two bf16 pieces in phase 2
// (tests/test_split_emulation.py), at half the MFMAs and 6 instead of 11 split instructions per pair in phase 1.
// FV: the kernel also attends the image features, v_l = a_v^T V (model.py:391) -- location-major V, NT = 2, four waves.
template <int NT, int NW, bool LM, int NP_, bool FV = false>
__global__ __launch_bounds__(NW * 64, 2) void coattn_fwd32_kernel(const FwdArgs a) {
  static_assert(!FV || (LM && NT == 2 && NW == 4), "the fused v pass: location-major features, N <= 64, 256 threads");
  constexpr bool HF = NP_ == 4;
  constexpr int NP = HF ? 2 : NP_;                   // pieces per operand in phase 2 (and, with HF, in phase 1)
  constexpr bool SP = NP == 1;
  constexpr int NPAD = 32 * NT;
  constexpr int SLD = 36;                            // row stride of the f32 reduction slots [n][t = 32]: 16-byte accesses
  constexpr int SLOT_FLOATS = NPAD * SLD;            //   of 8 consecutive rows cover the 32 banks once
  constexpr int PIECE = NPAD * 32;                   // bf16 elements of one piece of the C image [n][t = 32]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* slot0 = reinterpret_cast<float*>(smem);
  short* Cimg = reinterpret_cast<short*>(smem + SLOT_FLOATS * 4);
  float* slot1 = reinterpret_cast<float*>(Cimg);     // aliases the image until the image is written
  float* svpart = slot0;                             // [NW * 4][NPAD]   (aliases slot0 from phase 2 on)
  float* sqpart = slot0 + NW * 4 * NPAD;             // [NW * 2][32]
  float* aqs = sqpart + NW * 2 * 32;                 // 32
  float* avs = aqs + 32;                             // [NPAD]  a_v for the fused v pass (FV)
  static_assert(NW * 4 * NPAD + NW * 2 * 32 + 32 + NPAD <= SLOT_FLOATS, "phase-2 scratch must fit the slot");
  static_assert(SLOT_FLOATS * 4 <= 3 * PIECE * 2, "slot 1 must fit inside the image region");
  // Small grids (NT = 2): every wave has a slot of its own (4 x 9 KB inside the dead rings), the image pass adds the four
  // -- one barrier and the add / put round of waves 0 and 2 less; same order of additions ((w0 + w1) + (w2 + w3)).
  constexpr bool SLOT4 = NT == 2 && NW == 4;

  int b, l;
  if (!block_to_pair(blockIdx.x, a.B, a.L, b, l)) return;
  if (HF) f16_saturating_conversions();
  CA_STAMP(0);
  const int N = a.N, T = a.T, d = a.d;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // The lane id is re-derived (v_mbcnt) at every phase: kept from kernel entry, the values computed from it would
  // live across the fully unrolled phase 1 and be spilled (a kernel with scratch pays for it at every launch).
  auto lane_id = [] { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); };
  int lane = lane_id();
  int tid = w * 64 + lane;
  int r = lane & 31, h = lane >> 5;                  // MFMA row / column index, lane half
  const float* Qp = a.Q[l] + (size_t)b * T * d;
  const float* Vp = a.V + (size_t)b * a.v_sB;
  const float* Pvp = a.Pv + (size_t)b * N * d;
  const size_t pair = (size_t)l * a.B + b;
  const float* Pqp = a.Pq + pair * (size_t)T * d;
  // buffer resources over exactly this sample's tensors: rows beyond T / N read as 0, stores there are dropped
  const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(Qp, (unsigned)T * d * 4u);
  const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(Vp, (unsigned)d * N * 4u);
  const __amdgpu_buffer_rsrc_t rs_pq = make_rsrc(Pqp, (unsigned)T * d * 4u);
  const int nsl = d / (128 * NW);                    // 128-channel slices per wave (1 at d = 512)

  // P_q of a pass (64 channels from c0), as two accumulator-shaped fragments: register g of lane (r, h) of fragment ct
  // holds P_q[t = crow(g, h)][c0 + 32 ct + r] -- the initial value of the H_q accumulators.  (One lane-dependent
  // offset, recomputed at every call, and scalar row offsets: 16 hoisted vector offsets would live -- spilled --
  // across the unit pipeline.)
  auto load_pq_frag = [&](int c0, f32x16 (&f)[2]) {
    int base = (4 * h * d + r) * 4;
    asm volatile("" : "+v"(base));
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 16; ++g)
        f[ct][g] = buf_load1(rs_pq, base + 128 * ct, (c0 + ((g & 3) + 8 * (g >> 2)) * d) * 4);
  };

  // ------------------------------------------------------------------ phase 1: A = Q V^T
  f32x16 pqf[2];
  {
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[nt][g] = 0.f;
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};   // smallest piece products first
    if constexpr (LM) {
      // Location-major operand stream, whole cache lines.  Element i of a lane's fragment <-> channel
      // k0 + 4 h + (i & 3) + 8 (i >> 2), the same order for A and B.  A unit is 32 rows (tokens of Q, or the locations
      // of one tile of V) x 32 channels = 32 x 128 B: four LDS-DMA instructions (buffer_load_dwordx4 ... lds) of 8 rows
      // x 128 B each -- every instruction touches 8 whole lines (a lane per 16 bytes of a line); a fragment-shaped
      // access (a lane per row) touches 32 lines for the same kilobyte and the stream ran at half the rate.
      // The DMA writes LDS lane-linear, so the bank swizzle is applied on the global side: 16-byte position pos of
      // row n holds channel chunk pos ^ (n & 7).  Read-back: 4 x ds_read_b128 per lane (chunks h, 2+h, 4+h, 6+h of
      // the lane's row: the fragments of the unit's two 16-channel k-steps), 8 consecutive lanes = 8 rows at 8
      // distinct positions.  Per-wave ring of 4 slots of 4 KB (the LDS is idle until the cross-wave reduction): one
      // unit in the registers, one being read back, three in flight (12 KB per wave, counted vmcnt).
      constexpr int UPK = NT + 1;                    // units per 32-channel step: the Q unit, then the NT tiles
      constexpr int BODY = (UPK % 4 == 0) ? UPK : (UPK % 2 == 0) ? 2 * UPK : 4 * UPK;   // static slot indices
      constexpr int KU = BODY / UPK;
      // RS slots per wave: RS - 1 units (of four DMA instructions each) in flight behind the one being read back.  A slot index is
      // (unit number) % RS: a compile-time value when the body's BODY units are a multiple of RS (RS = 4) or when the wave has
      // ONE body (d = 512: G == KU); otherwise the body's first slot is carried in a scalar.
      constexpr int RS = (NT == 2) ? COATTN_P1_RING : 4;
      static_assert(RS >= 4 && RS <= 6, "phase-1 ring: 4 to 6 slots");
      char* ringb = smem + w * (RS * 4096);
      const int dvoff = ((lane >> 3) * d) * 4 + (((lane & 7) ^ (lane >> 3)) << 4);
      const int G = 4 * nsl;                         // 32-channel steps of this wave (a multiple of KU)
      auto chan0 = [&](int g) { return ((g >> 2) * NW + w) * 128 + 32 * (g & 3); };
      // beyond the last step the addresses fall outside the sample or on its next rows: harmless, never used
      auto dma_unit = [&](int g, const int j, const int slot) {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        if (COATTN_P1_NODMA) return;                 // developer switch: the compute side alone (stale LDS data)
        char* dst = ringb + slot * 4096;
        const int so = ((j == 0 ? 0 : 32 * (j - 1) * d) + chan0(g)) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (j == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (lds_ptr)(dst + 1024 * i), 16, dvoff, so + 8 * i * d * 4, 0, 0);
          else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_ptr)(dst + 1024 * i), 16, dvoff, so + 8 * i * d * 4, 0, 0);
        }
      };
      const int rd_off = r * 128, rd_key = r & 7;
      auto read_unit = [&](const int slot) -> f32x16 {
        const char* src = ringb + slot * 4096 + rd_off;
        f32x16 x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src + (((2 * c + h) ^ rd_key) << 4));
          x[4 * c] = v[0]; x[4 * c + 1] = v[1]; x[4 * c + 2] = v[2]; x[4 * c + 3] = v[3];
        }
        return x;
      };
      f32x16 nxt;
      bf16x8 qa[2][3];
      u32x4 vh[2], vm[2], vl[2];                     // split pieces of the current V unit (two k-steps)
#pragma unroll
      for (int p = 0; p < RS; ++p) dma_unit(p / UPK, p % UPK, p);
      vmcnt_wait(4 * (RS - 1));                      // unit 0 (the first Q unit) has landed; units 1 .. RS - 1 may fly
      nxt = read_unit(0);
      int sbase = 0;                                 // slot of the body's unit 0 (stays 0 when BODY % RS == 0)
#pragma unroll 1
      for (int g0 = 0; g0 < G; g0 += KU) {
#pragma unroll
        for (int p = 0; p < BODY; ++p) {             // unit p of the body: step g0 + p / UPK, position p % UPK
          const int j = p % UPK, jn = (p + 1) % UPK;
          const f32x16 cur = nxt;
          int s_cur = p % RS, s_nxt = (p + 1) % RS;
          if constexpr (BODY % RS != 0) {            // (scalar arithmetic; folds away for the first body)
            s_cur = (sbase + p % RS) % RS; s_nxt = (sbase + (p + 1) % RS) % RS;
          }
          // the read-back of unit p is complete: refill its slot with unit p + RS, then read back unit p + 1
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          dma_unit(g0 + (p + RS) / UPK, (p + RS) % UPK, s_cur);
          vmcnt_wait(4 * (RS - 1));                  // units p + 2 .. p + RS may fly
          nxt = read_unit(s_nxt);
          __builtin_amdgcn_sched_barrier(0);
          if (COATTN_P1_NOMATH) {                    // developer switch: the operand stream alone
            acc[0][p % 16] += nxt[0] + nxt[15];
            continue;
          }
          if (j == 0) {                              // Q unit: split it, and the first tile's fragment
            splitn_x<HF ? 2 : 3, HF>(f32x8{cur[0], cur[1], cur[2], cur[3], cur[4], cur[5], cur[6], cur[7]}, qa[0]);
            splitn_x<HF ? 2 : 3, HF>(f32x8{cur[8], cur[9], cur[10], cur[11], cur[12], cur[13], cur[14], cur[15]}, qa[1]);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
              unsigned hh = 0, mm = 0, ll = 0;
              split_pair_x<HF ? 2 : 3, HF>(nxt[2 * m], nxt[2 * m + 1], hh, mm, ll);
              vh[m >> 2][m & 3] = hh; vm[m >> 2][m & 3] = mm; vl[m >> 2][m & 3] = ll;
            }
          } else {
            const bf16x8 b3[2][3] = {{__builtin_bit_cast(bf16x8, vh[0]), __builtin_bit_cast(bf16x8, vm[0]), __builtin_bit_cast(bf16x8, vl[0])},
                                     {__builtin_bit_cast(bf16x8, vh[1]), __builtin_bit_cast(bf16x8, vm[1]), __builtin_bit_cast(bf16x8, vl[1])}};
            u32x4 nh[2], nm[2], nl[2];
#pragma unroll
            for (int m = 0; m < 12; ++m) {
              const int ks = m / 6, i = m % 6;
              // (HF: the products of pieces 0 / 1 are the last three of the six, in the same order)
              if (SP ? i == 5 : (!HF || i >= 3)) acc[j - 1] = mfma32_16<HF>(qa[ks][PA[i]], b3[ks][PB[i]], acc[j - 1]);
              if (jn != 0 && m >= 2 && m < 10) {     // the next unit is a V tile: split it under these MFMAs
                const int pr = m - 2;
                unsigned hh = 0, mm = 0, ll = 0;
                split_pair_x<HF ? 2 : 3, HF>(nxt[2 * pr], nxt[2 * pr + 1], hh, mm, ll);
                nh[pr >> 2][pr & 3] = hh; nm[pr >> 2][pr & 3] = mm; nl[pr >> 2][pr & 3] = ll;
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            if (jn != 0) {
#pragma unroll
              for (int ks = 0; ks < 2; ++ks) { vh[ks] = nh[ks]; vm[ks] = nm[ks]; vl[ks] = nl[ks]; }
            }
          }
        }
        if constexpr (BODY % RS != 0) sbase = (sbase + BODY) % RS;
      }
    } else {
      // Channel-major operand stream.  Element i of a lane's fragment <-> channel k0 + 4 h + (i & 3) + 8 (i >> 2), the
      // same order for A and B.  The fragments go HBM -> LDS by LDS-DMA (buffer_load ... lds, no VGPR destination) into
      // a per-wave ring of R slots of 2 KB -- the LDS is idle until the cross-wave reduction -- and are read back one
      // unit before their MFMAs.  Units of a k-step (16 channels): the Q fragment, then the NT location tiles.
      // The first NB tiles of a k-step run on the bf16 MFMA with the exact 3-way split, the others (none by default) on
      // the f32 MFMA straight from the registers (lane (., h) of a 32x32x2 MFMA supplies k = h: element i pairs
      // channels (k0 + (i&3) + 8(i>>2), the same + 4)).
      constexpr int NB = COATTN_P1_NB >= 0 ? (COATTN_P1_NB < NT ? COATTN_P1_NB : NT) : NT;
      constexpr int UPK = NT + 1;                      // units per k-step
      constexpr int KU = (UPK >= 6) ? 1 : 2;           // k-steps per loop body
      constexpr int R = UPK * KU;                      // ring slots = units per loop body: static slot indices
      char* ringb = smem + w * (R * 2048);
      const int q_voff = (r * d + 4 * h) * 4;
      const int v_voff = ((lane >> 3) * N + (lane & 7) * 4) * 4;
      const int G = 8 * nsl;                           // 16-channel steps of this wave
      auto chan0 = [&](int g) { return ((g >> 3) * NW + w) * 128 + 16 * (g & 7); };
      // DMA of unit (k-step g, position j) into its slot; beyond the last k-step the addresses fall outside the sample or
      // on its next rows: harmless, the data is never used
      auto dma_unit = [&](int g, const int j, const int slot) {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        char* dst = ringb + slot * 2048;
        const int k0 = chan0(g);
        if (COATTN_P1_NODMA) return;                   // developer switch: the compute side alone (stale LDS data)
        if (j == 0) {                                  // Q fragment
          // (the +32 bytes of the second piece go into the scalar offset: an instruction offset would also move the
          // LDS address)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (lds_ptr)dst, 16, q_voff, k0 * 4, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (lds_ptr)(dst + 1024), 16, q_voff, k0 * 4 + 32, 0, 0);
        } else {
          // channel-major V [d][N]: a unit is 16 channel rows x 32 locations (128 B per row), fetched by two 16-byte
          // DMAs of 8 rows each (a lane per 16 bytes of a row segment) into a row-major image [16][32]
          const int so = (k0 * N + 32 * (j - 1)) * 4;  // columns >= N: finite junk, zeroed when C is finalised
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_ptr)dst, 16, v_voff, so, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_ptr)(dst + 1024), 16, v_voff, so + 8 * N * 4, 0, 0);
        }
      };
      constexpr int DPU = 2;                           // DMA instructions of a unit
      auto read_unit = [&](const int j, const int slot) -> f32x8 {
        const char* src = ringb + slot * 2048;
        if (j == 0) {
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(src + lane * 16);
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(src + 1024 + lane * 16);
          return f32x8{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        }
        f32x8 x;                                       // element i <-> channel row 4h + (i&3) + 8(i>>2), location r
  #pragma unroll
        for (int i = 0; i < 8; ++i)
          x[i] = *reinterpret_cast<const float*>(src + (4 * h + (i & 3) + 8 * (i >> 2)) * 128 + r * 4);
        return x;
      };
      // DMAs still allowed in flight when unit p of the body is read back: those of the R - 2 units issued after it
      // (unit p+1 .. p+R-2; position 0 of a k-step is a Q unit)
      auto pending_after = [&](const int p) {
        int n = 0;
        for (int q = p + 1; q <= p + R - 2; ++q) n += (q % UPK == 0) ? 2 : DPU;
        return n;
      };
      f32x8 qraw, cur, nxt;
      bf16x8 qa[3];
      u32x4 vh, vm, vl;                                // split pieces of the current bf16 unit's V fragment
      auto split_pair_v = [&](const f32x8& x, const int pr) {
        unsigned hh = 0, mm = 0, ll = 0;
        split_pair_x<HF ? 2 : 3, HF>(x[2 * pr], x[2 * pr + 1], hh, mm, ll);
        vh[pr] = hh; vm[pr] = mm; vl[pr] = ll;
      };
      // prologue: units 0 .. R-2 in flight, unit 0 (the first Q fragment) read back
  #pragma unroll
      for (int p = 0; p < R - 1; ++p) dma_unit(p / UPK, p % UPK, p);
      vmcnt_wait(pending_after(0));                    // unit 0 has landed; units 1 .. R-2 may fly
      nxt = read_unit(0, 0);
  #pragma unroll 1
      for (int g0 = 0; g0 < G; g0 += KU) {
  #pragma unroll
        for (int p = 0; p < R; ++p) {                  // unit p of the body: k-step g0 + p / UPK, position p % UPK
          const int j = p % UPK, pn = (p + 1) % R, jn = (p + 1) % UPK;
          cur = nxt;
          // Refill the slot of the previous unit, then read back the next one.  The wait is the ordering between that
          // slot's read-back and its refill: the values read from it were consumed one iteration ago IN THE SOURCE, but
          // nothing ties the DMA issue to the instructions that consume them, and hipcc hoists it above them -- the
          // two-fp16-piece build issued the refill of the Q slot before the wait for the Q fragment's second
          // ds_read_b128, and on a busy LDS the refill (an L2 hit) now and then overtook the read: eight token rows of one
          // pair's affinity wrong in one launch out of eight at B = 640 (tools/probe_repeat.py found it; the three-piece
          // builds never showed it in thousands of soak rounds, but carried the same missing order).
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          dma_unit(g0 + (p + R - 1) / UPK, (p + R - 1) % UPK, (p + R - 1) % R);
          vmcnt_wait(pending_after(pn));
          nxt = read_unit(jn, pn);
          __builtin_amdgcn_sched_barrier(0);
          if (COATTN_P1_NOMATH) {                      // developer switch: the operand stream alone
            acc[0][p % 16] += nxt[0] + nxt[7];
            continue;
          }
          const bool next_bf = jn >= 1 && (jn - 1) < NB;    // the next unit is a V tile on the bf16 path: split it here
          if (j == 0) {                                // Q fragment of this k-step: split it, and the first tile's fragment
            qraw = cur;
            if (NB > 0) splitn_x<HF ? 2 : 3, HF>(qraw, qa);
            if (next_bf) {
  #pragma unroll
              for (int pr = 0; pr < 4; ++pr) split_pair_v(nxt, pr);
            }
          } else if (j - 1 < NB) {
            const bf16x8 b3[3] = {__builtin_bit_cast(bf16x8, vh), __builtin_bit_cast(bf16x8, vm), __builtin_bit_cast(bf16x8, vl)};
            u32x4 nh, nm, nl;
  #pragma unroll
            for (int m = 0; m < 6; ++m) {
              if (SP ? m == 5 : (!HF || m >= 3)) acc[j - 1] = mfma32_16<HF>(qa[PA[m]], b3[PB[m]], acc[j - 1]);
              if (next_bf && m < 4) {
                unsigned hh = 0, mm = 0, ll = 0;
                split_pair_x<HF ? 2 : 3, HF>(nxt[2 * m], nxt[2 * m + 1], hh, mm, ll);
                nh[m] = hh; nm[m] = mm; nl[m] = ll;
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            if (next_bf) { vh = nh; vm = nm; vl = nl; }
          } else {
  #pragma unroll
            for (int m = 0; m < 8; ++m) {
              acc[j - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qraw[m], cur[m], acc[j - 1], 0, 0, 0);
              if (next_bf && m < 4) split_pair_v(nxt, m);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
    }
    vmcnt_wait(0);                                   // the rings' memory becomes the reduction slots:
    __syncthreads();                                 // every wave is done with its ring
    CA_STAMP(1);
    // cross-wave sum in a fixed order through LDS.  C/D layout: col = r (location), rows crow(g, h) (tokens): registers
    // 4 gg .. 4 gg + 3 are four consecutive tokens of one location -> one 16-byte access of the slot [n][t].
    auto put = [&](float* slot) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
          *reinterpret_cast<f32x4*>(&slot[(32 * nt + r) * SLD + 8 * gg + 4 * h]) =
              f32x4{acc[nt][4 * gg], acc[nt][4 * gg + 1], acc[nt][4 * gg + 2], acc[nt][4 * gg + 3]};
    };
    auto add = [&](const float* slot) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(&slot[(32 * nt + r) * SLD + 8 * gg + 4 * h]);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[nt][4 * gg + i] += x[i];
        }
    };
    // (w0 + w1) -> slot 0, (w2 + w3) -> slot 1; the image pass adds the two
    if constexpr (SLOT4) {
      put(slot0 + w * SLOT_FLOATS);
      __syncthreads();
    } else if constexpr (NW == 4) {
      if (w == 1) put(slot0);
      if (w == 3) put(slot1);
      __syncthreads();
      if (w == 0) { add(slot0); put(slot0); }
      if (w == 2) { add(slot1); put(slot1); }
      __syncthreads();
    } else {
      if (w == 1) put(slot0);
      __syncthreads();
      if (w == 0) { add(slot0); put(slot0); }
      __syncthreads();
    }
  }
  lane = lane_id(); tid = w * 64 + lane; r = lane & 31; h = lane >> 5;
  load_pq_frag(w * 128, pqf);                        // the first pass' P_q fragments fly under the tanh / split pass
  bf16x8 pqB[2][2][3];
  CA_STAMP(6);
  // C = tanh(sum) by all threads: a thread takes four consecutive tokens 4 tq .. 4 tq + 3 of one location (one 16-byte
  // slot read per partial sum); rows >= T are tanh(0) = 0 (their Q rows read as 0).  The three bf16 pieces go to the
  // image [piece][n][t] (8-byte writes).  All the slot reads come first: slot 1 lies inside the image region.
  {
    // C of this (sample, level) as a buffer: rows t >= T fall outside it (stores dropped), and a lane of a padded
    // column n >= N is sent outside it through its offset -- no branches around the stores
    const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(a.C + pair * (size_t)T * N, (unsigned)T * N * 4u);
    constexpr int PER = 8 * NPAD / (NW * 64);        // (location, token quad) items per thread (exact: NPAD = 32 NT)
    static_assert(8 * NPAD % (NW * 64) == 0, "the image pass covers the slot in whole sweeps");
    f32x4 sum[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int e = tid + k * NW * 64, tq = e / NPAD, n = e - tq * NPAD;
      sum[k] = *reinterpret_cast<const f32x4*>(&slot0[n * SLD + 4 * tq]);
      if constexpr (SLOT4) {
        sum[k] += *reinterpret_cast<const f32x4*>(&slot0[SLOT_FLOATS + n * SLD + 4 * tq]);
        f32x4 s23 = *reinterpret_cast<const f32x4*>(&slot0[2 * SLOT_FLOATS + n * SLD + 4 * tq]);
        s23 += *reinterpret_cast<const f32x4*>(&slot0[3 * SLOT_FLOATS + n * SLD + 4 * tq]);
        sum[k] += s23;
      } else if constexpr (NW == 4) sum[k] += *reinterpret_cast<const f32x4*>(&slot1[n * SLD + 4 * tq]);
    }
    if constexpr (NW == 4) lds_barrier();            // every read of slots 1 .. is done: the image may overwrite them
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int e = tid + k * NW * 64, tq = e / NPAD, n = e - tq * NPAD;
      const bool in = n < N;                         // padded columns of a channel-major V carry junk
      const int cvoff = in ? (4 * tq * N + n) * 4 : 0x40000000;
      float c[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float th = tanh_fast(sum[k][i]);
        c[i] = in ? th : 0.f;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, th), rs_c, cvoff, i * N * 4, 0);
      }
      unsigned hh[2] = {0, 0}, mm[2] = {0, 0}, ll[2] = {0, 0};
      split_pair_x<NP, HF>(c[0], c[1], hh[0], mm[0], ll[0]);
      split_pair_x<NP, HF>(c[2], c[3], hh[1], mm[1], ll[1]);
      const int off = n * 32 + 8 * ((tq >> 1) ^ ((n >> 2) & 3)) + 4 * (tq & 1);
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<u32x2*>(Cimg + off) = u32x2{hh[0], hh[1]};
      if (NP >= 2) *reinterpret_cast<u32x2*>(Cimg + PIECE + off) = u32x2{mm[0], mm[1]};
      if (NP == 3) *reinterpret_cast<u32x2*>(Cimg + 2 * PIECE + off) = u32x2{ll[0], ll[1]};
    }
  }
  lds_barrier();
  CA_STAMP(2);

  // ------------------------------------------------------------------ phase 2: H_v scores, H_q
  const int ntiles = (N + 31) >> 5;
  // lane constants of the transposing reads (A operand of H_q): lane 4 q + p of a 16-lane group supplies the address
  // of image row n1 + q, tokens 16 g1 + 4 p .. + 3; the rows' swizzle key is ((n1 + q) >> 2) & 3 = h (first read,
  // n1 = 32 nt + 16 s + 4 h) and h + 2 (second read, n1 + 8)
  const int tq = (lane & 15) >> 2, tp_ = lane & 3, g1 = (lane >> 4) & 1;
  const int tr_off0 = (4 * h + tq) * 32 + 8 * ((2 * g1 + (tp_ >> 1)) ^ h) + 4 * (tp_ & 1);
  const int tr_off1 = (4 * h + 8 + tq) * 32 + 8 * ((2 * g1 + (tp_ >> 1)) ^ (h + 2)) + 4 * (tp_ & 1);
  const int rk = (r >> 2) & 3;                       // swizzle key of this lane's own image row (row reads)
  // A pass' P_q fragments (accumulator-shaped) become (a) the initial value of the H_q accumulators: H_q = P_q + C P_v,
  // and (b) the split B operands of H_v = P_v + C^T P_q, whose k index runs t = 16 ks + 8 h + i in lane half h: the
  // accumulator rows of a half are t = 16 ks + 4 h + {0..3, 8..11}, so registers 8 ks + k of the upper lane half
  // trade places with registers 8 ks + 4 + k of the lower half (v_permlane32_swap).
  f32x16 accq[2];
  auto take_pq = [&](f32x16& f0, f32x16& f1) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      f32x16& f = ct ? f1 : f0;
      accq[ct] = f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f32x8 x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // (inline asm, not __builtin_amdgcn_permlane32_swap: with both inputs live after the swap hipcc (ROCm 7.2)
          // copied ONE of them into both operands -- wrong B operands, caught by the parity tests; the no-ops cover
          // the VALU-write -> permlane-swap wait state the compiler would have inserted)
          float lo = f[8 * ks + k], hi = f[8 * ks + 4 + k];
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
          x[k] = lo;
          x[4 + k] = hi;
        }
        splitn_x<NP, HF>(x, pqB[ct][ks]);
      }
    }
  };
  take_pq(pqf[0], pqf[1]);
  {
#pragma unroll 1
    for (int pi = 0; pi < 2 * nsl; ++pi) {           // passes: 64 channels each, two per 128-channel slice
      const int sl = pi >> 1, ps = pi & 1;
      const int c0 = (sl * NW + w) * 128 + 64 * ps;
      // Scores without the tanh's last step: tanh(x) = 1 - 2 / (1 + e^{2x}), so sum_k w_k tanh(x_k) = sum_k w_k +
      // sum_k (-2 w_k) / (1 + e^{2 x_k}); the first sum is the same for every location and the softmax over the
      // locations is invariant to it -- it is dropped, and one VALU operation per element with it.
      const float wvr[2] = {-2.0f * a.wv[c0 + r], -2.0f * a.wv[c0 + 32 + r]};
      // the next pass' channels (beyond the last pass: offsets >= d, the values are never used)
      const int c0n = ((((pi + 1) >> 1) * NW + w) * 128 + 64 * ((pi + 1) & 1));
      // Unit pipeline over u = 2 nt + ct (a 32-location x 32-channel fragment of P_v, 16 VGPRs): while the MFMAs of
      // unit u run (H_q += C . P_v with the raw fragment as B operand, then H_v = P_v + C^T P_q accumulated onto
      // it), the VALU finishes unit u-1 (tanh, scores) and splits the operands of the following MFMAs; loads run
      // two units ahead.  Ring of four fragments: u+2 in flight, u+1 landed, u in the MFMAs, u-1 in the VALU.
      f32x16 ring[4];
      float svp[16];                                 // score partials of the current tile's first channel half
      bf16x8 pb0[3];                                 // split B operand of the next unit's first k-step
      const int U = 2 * ntiles;
      // pv[g] = P_v[32 nt + crow(g, h)][c0 + 32 ct + r]: 128 contiguous bytes per half wave and load
      // The two units past the last tile are the next pass' P_q fragments (same lane pattern: rows crow(g, h) of
      // P_q, channels c0n + 32 (u - U) + r): they fly under the last units and the epilogue.
      auto load_unit = [&](int u, f32x16& dst) {
        const bool tail = u >= U;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(tail ? Pqp : Pvp, (unsigned)(tail ? T : N) * d * 4u);
        const int so = tail ? (c0n + 32 * (u - U)) * 4 : (32 * (u >> 1) * d + c0 + 32 * (u & 1)) * 4;
#pragma unroll
        for (int g = 0; g < 16; ++g) dst[g] = buf_load1(rs, (crow(g, h) * d + r) * 4, so);
      };
      auto split_half = [&](const f32x16& pv, const int s2, bf16x8 (&pb)[3]) {
        const f32x8 x = f32x8{pv[8 * s2], pv[8 * s2 + 1], pv[8 * s2 + 2], pv[8 * s2 + 3],
                              pv[8 * s2 + 4], pv[8 * s2 + 5], pv[8 * s2 + 6], pv[8 * s2 + 7]};
        splitn_x<NP, HF>(x, pb);
      };
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
      auto store_scores = [&](int u, float mine) {   // u: the finished unit (second channel half of tile u >> 1)
        if (u >= 0) {
          float* dst = &svpart[((w * 2 + ps) * 2 + g1) * NPAD + 32 * (u >> 1) + crow(lane & 15, h)];
          if (sl > 0) mine += *dst;                  // accumulate over this wave's channel slices
          *dst = mine;
        }
      };
      // One unit = 24 MFMAs of unit u (fragment `cur`), each followed by one chunk of the VALU work that does not
      // depend on it: the split of the unit's second k-step, tanh / scores of unit u-1 (`prev`), the split of the
      // first k-step of unit u+1 (`next`), the 16-lane row sums.  The compiler's scheduler clusters MFMAs and VALU
      // work when left alone (a lone wave then pays both in series), so every (MFMA, chunk) pair is fenced.
      // Piece products of the exact split, smallest first: a0 b2, a2 b0, a1 b1, a0 b1, a1 b0, a0 b0.
      auto unit = [&](int u, const int ct, f32x16& cur, const f32x16& prev, const f32x16& next) {
        const short* img = Cimg + 32 * (u >> 1) * 32;
        bf16x8 cq0[3], cq1[3], ca0[3], ca1[3];       // A operands, read one MFMA group ahead of their use
        u32x4 h1, m1, l1, h0, m0, l0;                // pieces of pb1 (this unit, k-step 1) / the next unit's pb0
        float mine = 0.f, y[8], z[4], w2[2];
        read_cq(img, 0, cq0);
#pragma unroll
        for (int m = 0; m < 24; ++m) {
          // ---- the MFMA
          // (slot m % 6 of a group of six: every slot at width 3, every other one at width 2, the last at width 1)
          const int kp = slot_product<NP>(m % 6), grp = kp < 0 ? -1 : m / 6, pa = piece_a<NP>(kp < 0 ? 0 : kp), pb = piece_b<NP>(kp < 0 ? 0 : kp);
          if (grp == 0) accq[ct] = mfma32_16<HF>(cq0[pa], pb0[pb], accq[ct]);
          if (grp == 1) {
            const bf16x8 b = pb == 0 ? __builtin_bit_cast(bf16x8, h1) : pb == 1 ? __builtin_bit_cast(bf16x8, m1)
                                                                                : __builtin_bit_cast(bf16x8, l1);
            accq[ct] = mfma32_16<HF>(cq1[pa], b, accq[ct]);
          }
          if (grp == 2) cur = mfma32_16<HF>(ca0[pa], pqB[ct][0][pb], cur);
          if (grp == 3) cur = mfma32_16<HF>(ca1[pa], pqB[ct][1][pb], cur);
          // ---- the next group's A operands
          if (m == 1) read_cq(img, 1, cq1);
          if (m == 7) read_ca(img, 0, ca0);
          if (m == 13) read_ca(img, 1, ca1);
          // ---- its VALU chunk
          if (m < 4) {                               // split pair m of this unit's second k-step (registers 8 .. 15)
            unsigned hh = 0, mm = 0, ll = 0;
            split_pair_x<NP, HF>(cur[8 + 2 * m], cur[8 + 2 * m + 1], hh, mm, ll);
            h1[m] = hh; m1[m] = mm; l1[m] = ll;
          }
          if (m >= 4 && m < 20) {                    // register g = m - 4 of unit u-1: its score term
            const int g = m - 4;
            if (ct == 1) svp[g] = sig2_scaled(prev[g]) * wvr[0];                    // u-1 was a first channel half
            else svp[g] = fmaf(sig2_scaled(prev[g]), wvr[1], svp[g]);              // second half: add the first
            asm volatile("" : "+v"(svp[g]));         // computed HERE (machine sinking would move it to its use)
          }
          if (ct == 0) {                             // the tile's 16 score registers: transposing sum over the row
            if (m >= 13 && m <= 20) y[m - 13] = bfly_a(svp[m - 13], svp[m - 5]);
            if (m == 20) z[0] = bfly_b(y[0], y[4]);
            if (m == 21) { z[1] = bfly_b(y[1], y[5]); z[2] = bfly_b(y[2], y[6]); }
            if (m == 22) { z[3] = bfly_b(y[3], y[7]); w2[0] = bfly_c(z[0], z[2], lane); }
            if (m == 23) { w2[1] = bfly_c(z[1], z[3], lane); mine = bfly_d(w2[0], w2[1], lane); }
          }
          if (m >= 20) {                             // split pair m - 20 of the next unit's first k-step
            unsigned hh = 0, mm = 0, ll = 0;
            split_pair_x<NP, HF>(next[2 * (m - 20)], next[2 * (m - 20) + 1], hh, mm, ll);
            h0[m - 20] = hh; m0[m - 20] = mm; l0[m - 20] = ll;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        pb0[0] = __builtin_bit_cast(bf16x8, h0);
        pb0[1] = __builtin_bit_cast(bf16x8, m0);
        pb0[2] = __builtin_bit_cast(bf16x8, l0);
        if (ct == 0) store_scores(u - 1, mine);
      };
      // scores of the last unit (a second channel half), outside the pipeline
      auto finish_last = [&](const f32x16& pv) -> float {
        float x[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) x[g] = fmaf(sig2_scaled(pv[g]), wvr[1], svp[g]);
        return row16_sum16(x, lane);
      };
      CA_STAMP(8 + 3 * pi);
      load_unit(0, ring[0]);
      load_unit(1, ring[1]);
      split_half(ring[0], 0, pb0);
#pragma unroll
      for (int g = 0; g < 16; ++g) { svp[g] = 0.f; ring[3][g] = 0.f; }
#pragma unroll 1
      for (int u0 = 0; u0 < U; u0 += 4) {            // U is even: units come in (ct = 0, ct = 1) pairs
        load_unit(u0 + 2, ring[2]);
        __builtin_amdgcn_sched_barrier(0);
        unit(u0, 0, ring[0], ring[3], ring[1]);
        __builtin_amdgcn_sched_barrier(0);
        load_unit(u0 + 3, ring[3]);
        __builtin_amdgcn_sched_barrier(0);
        unit(u0 + 1, 1, ring[1], ring[0], ring[2]);
        __builtin_amdgcn_sched_barrier(0);
        if (u0 + 2 < U) {
          load_unit(u0 + 4, ring[0]);
          __builtin_amdgcn_sched_barrier(0);
          unit(u0 + 2, 0, ring[2], ring[1], ring[3]);
          __builtin_amdgcn_sched_barrier(0);
          load_unit(u0 + 5, ring[1]);
          __builtin_amdgcn_sched_barrier(0);
          unit(u0 + 3, 1, ring[3], ring[2], ring[0]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // the last unit's scores
      store_scores(U - 1, (U & 2) ? finish_last(ring[1]) : finish_last(ring[3]));
      CA_STAMP(9 + 3 * pi);
      // H_q epilogue: hq = tanh(acc) (the accumulators started from P_q); saved for backward; s_q partials.  Branch-free: rows t >= T fall
      // outside the per-sample buffers (loads give 0, stores are dropped)
      {
        const __amdgpu_buffer_rsrc_t rs_hq = make_rsrc(a.Hq + pair * (size_t)T * d, (unsigned)T * d * 4u);
        // the row offsets are recomputed here in every pass: hoisted out of the pass loop they would be 16 registers
        // that live (spilled) across the unit pipeline
        int hrow = (4 * h * d + r) * 4;
        asm volatile("" : "+v"(hrow));
        auto eoff = [&](const int g, const int ct) { return hrow + (((g & 3) + 8 * (g >> 2)) * d + 32 * ct) * 4; };
        float part[16];                              // s_q terms of token crow(g, h) over this lane's two channels
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const float wq = a.wq[c0 + 32 * ct + r];
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            const float hq = tanh_scaled(accq[ct][g]);          // P_q, P_v carry the factor 2 log2(e) (fused.h: kPScale)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hq), rs_hq, eoff(g, ct), c0 * 4, 0);
            part[g] = ct ? fmaf(hq, wq, part[g]) : hq * wq;
          }
        }
        // lane (j, .) of a 16-lane row keeps the sum of register g = j over the row's 16 channels
        float mine = row16_sum16(part, lane);
        float* dst = &sqpart[(w * 2 + g1) * 32 + crow(lane & 15, h)];
        if (pi > 0) mine += *dst;                    // summed into the wave's slot in a fixed order
        *dst = mine;
      }
      if (U & 2) take_pq(ring[2], ring[3]); else take_pq(ring[0], ring[1]);
      CA_STAMP(10 + 3 * pi);
    }
  }

  // ------------------------------------------------------------------ phase 3
  lane = lane_id(); tid = w * 64 + lane;
  CA_STAMP(3);
  // q = sum_t a_q[t] Q[t][:]   (model.py:392): the Q rows of the thread's first channels are requested before the
  // softmaxes, whose latency they hide (all kTRows row loads in flight at once; rows >= T read 0)
  constexpr int QPRE = 2;                            // channel sweeps prefetched (d = 512 with four waves: all of them)
  float xq[QPRE][kTRows];
#pragma unroll
  for (int i = 0; i < QPRE; ++i)
#pragma unroll
    for (int t = 0; t < kTRows; ++t)
      xq[i][t] = buf_load1(rs_q, (t * d + tid + i * NW * 64) * 4, 0);   // sweeps past d: rows past T or junk, unused
  // FV: v = sum_n a_v[n] V[n][:] by this workgroup.  A thread takes four consecutive channels (128 threads cover d = 512;
  // wider d: sweeps of 512) and every other location row (tid >> 7 picks the parity): 16-byte loads, a wave reads two whole
  // 1 KB row segments per instruction.  The rows of the first sweep's first half (locations < 32) are requested here, before
  // the softmaxes; rows >= N lie outside the sample's buffer and read 0.  V was streamed by phase 1: it comes from L2.
  constexpr int VB = 16;                             // rows per thread and batch (two batches cover 64 locations)
  const int v_c4 = tid & 127, v_par = tid >> 7;
  const int v_voff = (v_par * d + 4 * v_c4) * 4;
  f32x4 xv0[FV ? VB : 1], xv1[FV ? VB : 1];
  if constexpr (FV) {
#pragma unroll
    for (int i = 0; i < VB; ++i) xv0[i] = buf_load4(rs_v, v_voff, 2 * i * d * 4);
  }
  lds_barrier();                                     // every wave's score partials are in LDS
  CA_STAMP(4);
  if (w == 0) {
    // a_v = softmax_n(s_v + c_v): N <= 32 NT <= 256 -> <= 4 values per lane
    constexpr int PER = (NPAD + 63) / 64;
    float sc[PER];
    float m = -INFINITY;
    const float cv = a.cv[0];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int n = lane + 64 * k;
      float s = -INFINITY;
      if (n < N) {
        s = cv;
#pragma unroll
        for (int p = 0; p < NW * 4; ++p) s += svpart[p * NPAD + n];
      }
      sc[k] = s;
      m = fmaxf(m, s);
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      sc[k] = (lane + 64 * k < N) ? expf(sc[k] - m) : 0.f;
      sum += sc[k];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    float* avg = a.av + pair * (size_t)N;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int n = lane + 64 * k;
      if (n < N) avg[n] = sc[k] * inv;
      if (FV) avs[n] = sc[k] * inv;                  // (zeros beyond N)
    }
  }
  if (w == 1) {
    // a_q = softmax_t(s_q + c_q), un-masked over all T positions (model.py:388), by the second wave
    float s = -INFINITY;
    if (lane < T) {
      s = a.cq[0];
#pragma unroll
      for (int p = 0; p < NW * 2; ++p) s += sqpart[p * 32 + lane];
    }
    const float mq = wave_max(s);
    const float e = (lane < T) ? expf(s - mq) : 0.f;
    const float se = wave_sum(e);
    const float aqv = e / se;
    if (lane < 32) aqs[lane] = aqv;                  // zeros beyond T
    if (lane < T) a.aq[pair * (size_t)T + lane] = aqv;
  }
  lds_barrier();
  float aqr[kTRows];
#pragma unroll
  for (int t = 0; t < kTRows; ++t) aqr[t] = aqs[t];
#pragma unroll
  for (int i = 0; i < QPRE; ++i) {
    const int dd = tid + i * NW * 64;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < kTRows; ++t) acc = fmaf(aqr[t], xq[i][t], acc);
    if (dd < d) a.q_out[pair * (size_t)d + dd] = acc;
  }
  for (int dd = tid + QPRE * NW * 64; dd < d; dd += NW * 64) {
    float x[kTRows];
#pragma unroll
    for (int t = 0; t < kTRows; ++t) x[t] = buf_load1(rs_q, (t * d + dd) * 4, 0);
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < kTRows; ++t) acc = fmaf(aqr[t], x[t], acc);
    a.q_out[pair * (size_t)d + dd] = acc;
  }
  if constexpr (FV) {
    f32x4* vpart = reinterpret_cast<f32x4*>(Cimg);   // [128] the odd rows' partial sums (the image is dead by now)
    for (int c0 = 0; c0 < d; c0 += 512) {            // sweeps of 512 channels (one at d = 512)
      if (c0 > 0) {
#pragma unroll
        for (int i = 0; i < VB; ++i) xv0[i] = buf_load4(rs_v, v_voff, (2 * i * d + c0) * 4);
      }
#pragma unroll
      for (int i = 0; i < VB; ++i) xv1[i] = buf_load4(rs_v, v_voff, (2 * (VB + i) * d + c0) * 4);
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < VB; ++i) acc += xv0[i] * avs[v_par + 2 * i];
#pragma unroll
      for (int i = 0; i < VB; ++i) acc += xv1[i] * avs[v_par + 2 * (VB + i)];
      if (c0 > 0) lds_barrier();                     // the previous sweep's partial sums have been read
      if (v_par) vpart[v_c4] = acc;
      lds_barrier();
      if (!v_par) *reinterpret_cast<f32x4*>(a.v_out + pair * (size_t)d + c0 + 4 * v_c4) = acc + vpart[v_c4];
    }
  }
  CA_STAMP(5);
}

// v_l[b][ch] = sum_n a_v[l][b][n] V[b][n][ch]   (model.py:391), location-major V, all L levels in one pass.
// grid (d / 128, B); 256 threads = 32 float4 lanes (128 channels: whole 512-byte row segments) x 8 location phases,
// fixed-order sum over the phases.  A thread's rows (<= 32 at N <= 256) come in four batches of 8 with two batches in
// flight; the first is requested before the attention weights are staged, so their latencies overlap.  Rows >= N lie
// outside the sample's buffer and read 0.
__global__ __launch_bounds__(256) void attend_v_lm_kernel(const float* V, long v_sB, const float* av, float* v_out,
                                                          int B, int N, int d, int L) {
  __shared__ float aw[3][256];
  __shared__ f32x4 part[3][8][32];
  const int b = blockIdx.y, c0 = blockIdx.x * 128;
  const int tid = threadIdx.x, cl = tid & 31, ph = tid >> 5;
  const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(V + (size_t)b * v_sB, (unsigned)N * d * 4u);
  const int voff = (ph * d + c0 + 4 * cl) * 4;
  const int rstep = 8 * d * 4;                         // a thread's consecutive rows are 8 apart
  f32x4 x0[8], x1[8];
  auto load8 = [&](const int batch, f32x4 (&x)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = buf_load4(rs_v, voff, (8 * batch + k) * rstep);
  };
  load8(0, x0);
  {                                                    // the three levels' weights of location tid: clamped addresses, requested together
    float w3[3];                                       // (a guarded load is a branch and a wait of its own: three latencies in a row)
#pragma unroll
    for (int l = 0; l < 3; ++l) w3[l] = av[((size_t)min(l, L - 1) * B + b) * N + min(tid, N - 1)];
#pragma unroll
    for (int l = 0; l < 3; ++l) aw[l][tid] = (l < L && tid < N) ? w3[l] : 0.f;
  }
  __syncthreads();
  f32x4 acc[3];
#pragma unroll
  for (int l = 0; l < 3; ++l) acc[l] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto fma8 = [&](const int batch, const f32x4 (&x)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int n = (ph + 8 * (8 * batch + k)) & 255;  // n < 256 always (ph + 8 * 31 = 255 at most)
#pragma unroll
      for (int l = 0; l < 3; ++l) acc[l] += x[k] * aw[l][n];
    }
  };
  load8(1, x1);
  fma8(0, x0);
  if (N > 128) {                                       // uniform: rows 128.. exist only then
    load8(2, x0);
    fma8(1, x1);
    load8(3, x1);
    fma8(2, x0);
    fma8(3, x1);
  } else {
    fma8(1, x1);
  }
#pragma unroll
  for (int l = 0; l < 3; ++l) part[l][ph][cl] = acc[l];
  __syncthreads();
  if (tid < 96) {
    const int l = tid >> 5, c = tid & 31;
    f32x4 s = part[l][0][c];
#pragma unroll
    for (int p = 1; p < 8; ++p) s += part[l][p][c];
    if (l < L) *reinterpret_cast<f32x4*>(v_out + ((size_t)l * B + b) * d + c0 + 4 * c) = s;
  }
}

template <int NT, int NW, bool LM, int NP, bool FV = false>
int launch_fwd32(const FwdArgs& a, hipStream_t s) {
  constexpr int NPAD = 32 * NT;
  constexpr int RING_SLOTS = (NT + 1) * ((NT + 1 >= 6) ? 1 : 2);
  const size_t lds_p2 = (size_t)NPAD * 36 * 4 + (size_t)3 * NPAD * 32 * 2,
               lds_p1 = LM ? (size_t)NW * (NT == 2 ? COATTN_P1_RING : 4) * 4096 : (size_t)NW * RING_SLOTS * 2048;
  const size_t lds = lds_p2 > lds_p1 ? lds_p2 : lds_p1;
  static DeviceOnce once;                            // the attribute is per device
  CA_TRY(once.run([&] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(coattn_fwd32_kernel<NT, NW, LM, NP, FV>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }, "coattn_fwd32"));
  const int groups = (a.B + 7) / 8;
  dim3 grid(groups * a.L * 8), block(NW * 64);
  hipLaunchKernelGGL((coattn_fwd32_kernel<NT, NW, LM, NP, FV>), grid, block, lds, s, a);
  CA_CHECK_LAUNCH("coattn_fwd32");
  return 0;
}

template <bool LM>
int dispatch_fwd32(const FwdArgs& a, hipStream_t s) {
  const bool small_n = a.N <= 64;
  const bool w2 = a.np == 2;                         // phase 2 on two bf16 pieces
  const bool hf = a.np == 4;                         // both phases on two FP16 pieces (the default)
  if (a.d % 512 == 0) {
    if constexpr (LM) {
      if (small_n && a.v_out) {                      // the kernel attends the image features too (FwdArgs::v_out)
        if (a.bf16) return launch_fwd32<2, 4, true, 1, true>(a, s);
        if (hf) return launch_fwd32<2, 4, true, 4, true>(a, s);
        return w2 ? launch_fwd32<2, 4, true, 2, true>(a, s) : launch_fwd32<2, 4, true, 3, true>(a, s);
      }
    }
    if (a.bf16) return small_n ? launch_fwd32<2, 4, LM, 1>(a, s) : launch_fwd32<7, 4, LM, 1>(a, s);
    if (hf) return small_n ? launch_fwd32<2, 4, LM, 4>(a, s) : launch_fwd32<7, 4, LM, 4>(a, s);
    if (w2) return small_n ? launch_fwd32<2, 4, LM, 2>(a, s) : launch_fwd32<7, 4, LM, 2>(a, s);
    return small_n ? launch_fwd32<2, 4, LM, 3>(a, s) : launch_fwd32<7, 4, LM, 3>(a, s);
  }
  if (hf) return small_n ? launch_fwd32<2, 2, LM, 4>(a, s) : launch_fwd32<7, 2, LM, 4>(a, s);
  if (w2) return small_n ? launch_fwd32<2, 2, LM, 2>(a, s) : launch_fwd32<7, 2, LM, 2>(a, s);
  return small_n ? launch_fwd32<2, 2, LM, 3>(a, s) : launch_fwd32<7, 2, LM, 3>(a, s);   // (the fp32 mode at these widths)
}

}  // namespace

int fused32_forward(const FwdArgs& a, hipStream_t s) {
  CA_CHECK_ARG(!a.v_out || (a.lm && a.N <= 64 && a.d % 512 == 0), "fused forward: the in-kernel v pass needs location-major features, N <= 64, d %% 512 == 0");
  return a.lm ? dispatch_fwd32<true>(a, s) : dispatch_fwd32<false>(a, s);
}

int launch_attend_v_lm(const float* V, long v_sB, const float* av, float* v_out, int B, int N, int d, int L, hipStream_t s) {
  CA_CHECK_ARG(N <= 256 && L <= 3 && d % 128 == 0, "attend_v (location-major): unsupported shape");
  hipLaunchKernelGGL(attend_v_lm_kernel, dim3(d / 128, B), dim3(256), 0, s, V, v_sB, av, v_out, B, N, d, L);
  CA_CHECK_LAUNCH("attend_v_lm");
  return 0;
}
