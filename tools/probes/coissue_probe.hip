// What do the two waves that share a SIMD on gfx950 overlap?  One 512-thread workgroup per CU (waves w and w+4 land
// on the same SIMD); waves 0..3 run role A, waves 4..7 role B; roles: 0 idle, 1 MFMA stream (v_mfma_f32_16x16x32_f16,
// 4 independent accumulators), 2 VALU stream (independent v_fma_f32), 3 mixed (per MFMA: `mix` VALU in the same wave).
// Prints wall time per configuration -> cycles per instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o coissue_probe coissue_probe.hip && ./coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// one VALU instruction of the kind the KAN kernels issue, opaque to the SLP vectoriser (no v_pk_* packing)
template <int VKIND>
__device__ __forceinline__ void valu1(float& a) {
    if constexpr (VKIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(1.0001f), "v"(0.5f));
    else if constexpr (VKIND == 1) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(0x01020304), "v"(0x07060100));
    else if constexpr (VKIND == 2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a) : "v"(0.5f));
    else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<double*>(&a)) : "v"(1.0), "v"(2.0));
}

template <int MIX, int VKIND = 0>
__device__ __forceinline__ void mfma_stream(int iters, float* sink, int lane) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(1.0f - i * 0.01f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k & 3], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MIX; ++m) valu1<VKIND>(v[(k + m) & 7]);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) *sink = s;
}

template <int VKIND>
__device__ __forceinline__ void valu_stream(int iters, float* sink, int lane) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = lane + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) valu1<VKIND>(v[k & 15]);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) *sink = s;
}

// roles: 0 idle, 1 mfma, 2 valu, 3 mfma+2 valu/mfma, 4 mfma + 4 valu/mfma
__device__ __forceinline__ void run_role(int role, int iters, float* sink, int lane) {
    switch (role) {
        case 1: mfma_stream<0>(iters, sink, lane); break;
        case 2: valu_stream<0>(iters, sink, lane); break;
        case 3: mfma_stream<2>(iters, sink, lane); break;
        case 4: mfma_stream<4>(iters, sink, lane); break;
        case 5: valu_stream<1>(iters, sink, lane); break;
        case 6: valu_stream<2>(iters, sink, lane); break;
        case 7: mfma_stream<4, 1>(iters, sink, lane); break;
        case 8: mfma_stream<1>(iters, sink, lane); break;
        case 9: mfma_stream<3>(iters, sink, lane); break;
        default: break;
    }
}

__global__ __launch_bounds__(512) void probe(int roleA, int roleB, int itA, int itB, int prioA, int prioB, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (prioA) __builtin_amdgcn_s_setprio(3);
        run_role(roleA, itA, sink, lane);
    } else {
        if (prioB) __builtin_amdgcn_s_setprio(3);
        run_role(roleB, itB, sink, lane);
    }
}

int main() {
    float* sink;
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const double mhz = prop.clockRate / 1000.0;
    printf("%s  %d CUs  clockRate %.0f MHz\n", prop.name, prop.multiProcessorCount, mhz);
    struct Cfg { const char* name; int rA, rB, itA, itB, pA, pB; };
    const int IT = 20000;
    std::vector<Cfg> cfgs = {
        {"A=perm(32/it)          B=idle", 5, 0, IT, 0, 0, 0},
        {"A=perm                 B=perm", 5, 5, IT, IT, 0, 0},
        {"A=cvt_pkrtz(32/it)     B=idle", 6, 0, IT, 0, 0, 0},
        {"A=mfma                 B=perm", 1, 5, IT, IT, 0, 0},
        {"A=perm                 B=mfma", 5, 1, IT, IT, 0, 0},
        {"A=mfma                 B=perm prio", 1, 5, IT, IT, 0, 1},
        {"A=mfma prio            B=perm", 1, 5, IT, IT, 1, 0},
        {"A=mfma+1fma/mfma       B=idle", 8, 0, IT, 0, 0, 0},
        {"A=mfma+3fma/mfma       B=idle", 9, 0, IT, 0, 0, 0},
        {"A=mfma+4perm/mfma      B=idle", 7, 0, IT, 0, 0, 0},
        {"A=mfma+4perm/mfma      B=same", 7, 7, IT, IT, 0, 0},
        {"A=mfma(8/it)           B=idle", 1, 0, IT, 0, 0, 0},
        {"A=valu(32/it)          B=idle", 2, 0, IT, 0, 0, 0},
        {"A=valu                 B=valu", 2, 2, IT, IT, 0, 0},
        {"A=mfma                 B=mfma", 1, 1, IT, IT, 0, 0},
        {"A=mfma                 B=valu(same count of iterations)", 1, 2, IT, IT, 0, 0},
        {"A=mfma prio            B=valu", 1, 2, IT, IT, 1, 0},
        {"A=mfma                 B=valu prio", 1, 2, IT, IT, 0, 1},
        {"A=valu                 B=mfma", 2, 1, IT, IT, 0, 0},
        {"A=valu                 B=mfma prio", 2, 1, IT, IT, 0, 1},
        {"A=mfma+2valu/mfma      B=idle", 3, 0, IT, 0, 0, 0},
        {"A=mfma+4valu/mfma      B=idle", 4, 0, IT, 0, 0, 0},
        {"A=mfma+2valu/mfma      B=same", 3, 3, IT, IT, 0, 0},
        {"A=mfma+4valu/mfma      B=same", 4, 4, IT, IT, 0, 0},
    };
    for (auto& c : cfgs) {
        probe<<<256, 512>>>(c.rA, c.rB, 100, 100, c.pA, c.pB, sink);   // warm
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<<<256, 512>>>(c.rA, c.rB, c.itA, c.itB, c.pA, c.pB, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-62s %8.3f ms   %7.1f us per 1000 iterations\n", c.name, ms, ms * 1e3 / (IT / 1000.0));
    }
    return 0;
}
