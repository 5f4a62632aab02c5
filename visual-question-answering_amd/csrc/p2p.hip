// One-shot gradient exchange for a fully connected xGMI node (the reference has only the TODO
// at main.py:102-106): every rank maps the gradient buckets of its peers (HIP IPC handles, exchanged once) and the data
// path is two bandwidth kernels that read peer memory directly -- no ring, no staging copies:
//
//   reduce-scatter  rank j sums shard j of ALL ranks' buckets, read straight from the peers (w - 1 links carry S / w each,
//                   at the same time), in rank order 0 .. w-1 (the same order on every rank: the averaged values are
//                   identical everywhere and repeatable), scales by 1 / w and writes the result over shard j of its
//                   OWN bucket (that region is read by nobody else in this phase);
//   all-gather      rank j copies the reduced shard i of every peer i into its own bucket.
//
// The phases are separated by cross-rank synchronisation that the HOST side provides (dist.py: a one-element all-reduce on
// the stream under RCCL, a barrier under gloo): the kernels themselves never wait on another GPU.
// Shapes: fp32 buckets of world x shard_elems floats, shard_elems % 4 == 0, 16-byte aligned, world <= 8 (one node).
#include "common.h"

namespace {

struct P2PPeers { const float* p[8]; };

template <int W>
__global__ __launch_bounds__(256) void p2p_reduce_kernel(const P2PPeers peers, float* out, const long n4, const float scale) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 x[W];
#pragma unroll
    for (int r = 0; r < W; ++r) x[r] = reinterpret_cast<const f32x4*>(peers.p[r])[i];   // all the links at once
    f32x4 s = x[0];
#pragma unroll
    for (int r = 1; r < W; ++r) s += x[r];                                               // rank order
    reinterpret_cast<f32x4*>(out)[i] = s * scale;
  }
}

// grid (x, world): row r copies shard r from its owner (row `rank`: nothing to do)
__global__ __launch_bounds__(256) void p2p_gather_kernel(const P2PPeers peers, float* self, const long shard, const int rank) {
  const int r = blockIdx.y;
  if (r == rank) return;
  const f32x4* src = reinterpret_cast<const f32x4*>(peers.p[r] + (long)r * shard);
  f32x4* dst = reinterpret_cast<f32x4*>(self + (long)r * shard);
  const long n4 = shard / 4, stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

int check_p2p(const void* const* bufs, int world, int rank, int64_t shard) {
  CA_CHECK_ARG(bufs && world >= 1 && world <= 8 && rank >= 0 && rank < world && shard > 0 && (shard % 4) == 0 &&
               shard * world < (1L << 40), "p2p: bad argument (world %d, rank %d, shard %ld)", world, rank, (long)shard);
  for (int r = 0; r < world; ++r)
    CA_CHECK_ARG(bufs[r] && (((uintptr_t)bufs[r]) & 15) == 0, "p2p: bucket of rank %d missing or not 16-byte aligned", r);
  return 0;
}

int grid_for(long n4) {
  const long want = (n4 + 255) / 256;
  return (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));          // <= 8 workgroups per CU, grid-stride beyond
}

}  // namespace

// Peer access from the current device to `peer_device` (the device a mapped bucket lives on): legal P2P loads need it.
// 0: enabled (or the same device); -1: the devices cannot reach each other.
extern "C" int coattn_p2p_enable_peer(int peer_device) {
  int cur = -1, can = 0;
  CA_CHECK_ARG(hipGetDevice(&cur) == hipSuccess, "p2p: no current device");
  if (cur == peer_device) return 0;
  CA_CHECK_ARG(hipDeviceCanAccessPeer(&can, cur, peer_device) == hipSuccess && can, "p2p: device %d cannot access device %d", cur,
               peer_device);
  const hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
  if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return 0; }
  CA_CHECK_ARG(e == hipSuccess, "p2p: hipDeviceEnablePeerAccess(%d): %s", peer_device, hipGetErrorString(e));
  return 0;
}

extern "C" int coattn_p2p_reduce_scatter(const void* const* peer_bufs, int world, int rank, int64_t shard_elems, float scale,
                                         void* stream) {
  CA_TRY(check_p2p(peer_bufs, world, rank, shard_elems));
  P2PPeers pp = {};
  for (int r = 0; r < world; ++r) pp.p[r] = (const float*)peer_bufs[r] + (long)rank * shard_elems;
  float* out = (float*)const_cast<void*>(peer_bufs[rank]) + (long)rank * shard_elems;
  const long n4 = shard_elems / 4;
  const dim3 grid(grid_for(n4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (world) {
    case 1: hipLaunchKernelGGL(p2p_reduce_kernel<1>, grid, block, 0, s, pp, out, n4, scale); break;
    case 2: hipLaunchKernelGGL(p2p_reduce_kernel<2>, grid, block, 0, s, pp, out, n4, scale); break;
    case 3: hipLaunchKernelGGL(p2p_reduce_kernel<3>, grid, block, 0, s, pp, out, n4, scale); break;
    case 4: hipLaunchKernelGGL(p2p_reduce_kernel<4>, grid, block, 0, s, pp, out, n4, scale); break;
    case 5: hipLaunchKernelGGL(p2p_reduce_kernel<5>, grid, block, 0, s, pp, out, n4, scale); break;
    case 6: hipLaunchKernelGGL(p2p_reduce_kernel<6>, grid, block, 0, s, pp, out, n4, scale); break;
    case 7: hipLaunchKernelGGL(p2p_reduce_kernel<7>, grid, block, 0, s, pp, out, n4, scale); break;
    default: hipLaunchKernelGGL(p2p_reduce_kernel<8>, grid, block, 0, s, pp, out, n4, scale); break;
  }
  CA_CHECK_LAUNCH("p2p_reduce");
  return 0;
}

extern "C" int coattn_p2p_all_gather(const void* const* peer_bufs, int world, int rank, int64_t shard_elems, void* stream) {
  CA_TRY(check_p2p(peer_bufs, world, rank, shard_elems));
  if (world == 1) return 0;
  P2PPeers pp = {};
  for (int r = 0; r < world; ++r) pp.p[r] = (const float*)peer_bufs[r];
  int gx = grid_for(shard_elems / 4);
  gx = gx > 512 ? 512 : gx;                                            // x world rows
  hipLaunchKernelGGL(p2p_gather_kernel, dim3(gx, world), dim3(256), 0, (hipStream_t)stream, pp,
                     (float*)const_cast<void*>(peer_bufs[rank]), (long)shard_elems, rank);
  CA_CHECK_LAUNCH("p2p_gather");
  return 0;
}
