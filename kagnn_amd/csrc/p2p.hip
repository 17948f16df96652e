// Direct peer-to-peer exchange steps of the feature-sharded layer ("hand-rolled direct P2P -- hipIpcMemHandle peer
// buffers ... local 8-way sum" as the alternative to RCCL's ring): every rank maps its peers' exchange buffers (IPC handles, opened by
// the host side: kagnn_amd/p2p.py) and READS them over xGMI.
//
//   reduce-scatter: y[n][c] = sum_p partial_p[n][rank*w + c]      -- one launch, P strided column blocks summed in rank order
//                   (deterministic); no rank-major staging copy of the partial sums, no ring: each of the P-1 links carries
//                   N*w*4 bytes once, all at the same time
//   all-gather:     g[n][p*w + c] = shard_p[n][c]                 -- the gathered gradient written in its final [N, out] layout
//
// xGMI is point-to-point, so a pull from all peers at once uses all links of the GPU concurrently; the reads are 16-byte, row
// contiguous (w*4 bytes per row and peer: 32 B at out = 64, P = 8 -- the transfer granularity, not the kernel, is the limit there).
// The reference has no multi-GPU code; the contract is BASELINE.json's north_star.
#include "common.h"

namespace kagnn {

constexpr int kMaxPeers = 16;
struct PeerPtrs { const float* p[kMaxPeers]; };

// thread = (row, 4 consecutive columns of the shard); w % 4 == 0, ld % 4 == 0, 16-byte aligned bases
__global__ __launch_bounds__(256) void p2p_reduce_scatter_kernel(PeerPtrs parts, int P, int rank, long N, int w, long ld,
                                                                 float* __restrict__ y, long ldy) {
    const int q = w >> 2;                                     // float4 per shard row
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= N * q) return;
    const long n = i / q;
    const int c = (int)(i - n * q) * 4;
    const long off = n * ld + (long)rank * w + c;
    float4 v[kMaxPeers];
#pragma unroll
    for (int p = 0; p < kMaxPeers; ++p)
        if (p < P) v[p] = *reinterpret_cast<const float4*>(parts.p[p] + off);     // all peers' loads in flight together
    float4 a = v[0];
#pragma unroll
    for (int p = 1; p < kMaxPeers; ++p)
        if (p < P) { a.x += v[p].x; a.y += v[p].y; a.z += v[p].z; a.w += v[p].w; }
    *reinterpret_cast<float4*>(y + n * ldy + c) = a;
}

__global__ __launch_bounds__(256) void p2p_all_gather_kernel(PeerPtrs shards, int P, long N, int w, long lds, float* __restrict__ g,
                                                             long ldg) {
    const int q = w >> 2;
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= N * q * P) return;
    const int p = (int)(i % P);                               // neighbouring threads pull from different peers (links)
    const long r = i / P;
    const long n = r / q;
    const int c = (int)(r - n * q) * 4;
    *reinterpret_cast<float4*>(g + n * ldg + (long)p * w + c) = *reinterpret_cast<const float4*>(shards.p[p] + n * lds + c);
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int p2p_reduce_scatter(const float* const* parts, int P, int rank, long N, int out, long ld, float* y, long ldy, hipStream_t st) {
    if (P < 1 || P > kMaxPeers || rank < 0 || rank >= P || out % P) return fail(KAGNN_ERR_ARG, "%s: bad world / rank / width", "p2p_reduce_scatter");
    const int w = out / P;
    if (w % 4 || ld % 4 || ldy % 4 || !al16(y)) return fail(KAGNN_ERR_UNSUPPORTED, "%s: shard width and leading dimensions must be multiples of 4 floats", "p2p_reduce_scatter");
    PeerPtrs pp{};
    for (int p = 0; p < P; ++p) { if (!parts[p] || !al16(parts[p])) return fail(KAGNN_ERR_ARG, "%s: null / unaligned peer buffer", "p2p_reduce_scatter"); pp.p[p] = parts[p]; }
    if (N == 0) return KAGNN_OK;
    const long items = N * (w / 4);
    p2p_reduce_scatter_kernel<<<(unsigned)cdiv(items, 256), 256, 0, st>>>(pp, P, rank, N, w, ld, y, ldy);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

int p2p_all_gather(const float* const* shards, int P, long N, int w, long lds, float* g, long ldg, hipStream_t st) {
    if (P < 1 || P > kMaxPeers) return fail(KAGNN_ERR_ARG, "%s: bad world size", "p2p_all_gather");
    if (w % 4 || lds % 4 || ldg % 4 || !al16(g)) return fail(KAGNN_ERR_UNSUPPORTED, "%s: shard width and leading dimensions must be multiples of 4 floats", "p2p_all_gather");
    PeerPtrs pp{};
    for (int p = 0; p < P; ++p) { if (!shards[p] || !al16(shards[p])) return fail(KAGNN_ERR_ARG, "%s: null / unaligned peer buffer", "p2p_all_gather"); pp.p[p] = shards[p]; }
    if (N == 0) return KAGNN_OK;
    const long items = N * (w / 4) * P;
    p2p_all_gather_kernel<<<(unsigned)cdiv(items, 256), 256, 0, st>>>(pp, P, N, w, lds, g, ldg);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

}  // namespace kagnn
