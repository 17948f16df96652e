// global_load_lds_dwordx4 on gfx950: where do the 64 lanes' 16-byte pieces land in LDS?  (build: hipcc --offload-arch=gfx950
// -O3 tools/probes/lds_dma_probe.hip -o /tmp/lds_dma_probe)  Expected: LDS[base + 16 * lane .. +16) = the 16 bytes at the
// lane's own global address, `base` wave-uniform (M0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ void copy_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ out, int perm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl = perm ? (lane ^ 5) : lane;                  // which 16-byte piece this lane fetches
    __builtin_amdgcn_global_load_lds((gptr_t)(src + wave * 256 + sl * 4), (lptr_t)(smem + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < (int)blockDim.x * 4; i += blockDim.x) out[i] = reinterpret_cast<unsigned*>(smem)[i];
}

int main() {
    const int n = 512 * 4;
    std::vector<unsigned> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = i;
    unsigned *d, *r;
    hipMalloc(&d, n * 4); hipMalloc(&r, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int perm = 0; perm < 2; ++perm) {
        copy_kernel<<<1, 512, 8192>>>(d, r, perm);
        hipMemcpy(o.data(), r, n * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n; ++i) {
            const int wave = i / 256, piece = (i % 256) / 4, w = i % 4;
            const int src_piece = perm ? (piece ^ 5) : piece;
            const unsigned want = wave * 256 + src_piece * 4 + w;
            if (o[i] != want) { if (bad < 5) printf("perm %d: LDS dword %d = %u, expected %u\n", perm, i, o[i], want); ++bad; }
        }
        printf("perm %d: %s (%d mismatches): LDS piece `lane` <- global address of `lane`\n", perm, bad ? "DIFFERENT" : "as expected", bad);
    }
    return 0;
}
