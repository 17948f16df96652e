// Layout probe for v_smfmac_f32_32x32x32_f16 on gfx950: which (lane, element) holds which K index of the
// dense B operand and of the 2:4-compressed A operand, and how the index VGPR is read.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/smfmac_probe.hip -o gpurun_out/smfmac_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const _Float16* a /*[64][8]*/, const _Float16* b /*[64][16]*/, const int* idx /*[64]*/, float* d /*[64][16]*/, int abid) {
    const int l = threadIdx.x;
    f16x8 av; f16x16 bv; f32x16 acc;
    for (int i = 0; i < 8; ++i) av[i] = a[l * 8 + i];
    for (int i = 0; i < 16; ++i) bv[i] = b[l * 16 + i];
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (abid == 0) acc = __builtin_amdgcn_smfmac_f32_32x32x32_f16(av, bv, acc, idx[l], 0, 0);
    else acc = __builtin_amdgcn_smfmac_f32_32x32x32_f16(av, bv, acc, idx[l], 0, 1);
    for (int i = 0; i < 16; ++i) d[l * 16 + i] = acc[i];
}

int main() {
    srand(1);
    // dense reference: A[32][32] with 2:4 sparsity along K, B[32][32]
    static float A[32][32], B[32][32], D[32][32];
    static int pos[32][8][2];                       // for row m, group g (8 groups of 4 along K): the two non-zero positions
    for (int m = 0; m < 32; ++m) for (int g = 0; g < 8; ++g) {
        int p0 = rand() % 4, p1 = rand() % 4; while (p1 == p0) p1 = rand() % 4;
        if (p0 > p1) { int t = p0; p0 = p1; p1 = t; }
        pos[m][g][0] = p0; pos[m][g][1] = p1;
        for (int j = 0; j < 4; ++j) A[m][4 * g + j] = 0.f;
        A[m][4 * g + p0] = (float)(rand() % 7 + 1); A[m][4 * g + p1] = -(float)(rand() % 5 + 1);
    }
    for (int kk = 0; kk < 32; ++kk) for (int n = 0; n < 32; ++n) B[kk][n] = (float)((rand() % 9) - 4);
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int kk = 0; kk < 32; ++kk) s += A[m][kk] * B[kk][n]; D[m][n] = s; }
    _Float16 *da, *db; int* di; float* dd;
    hipMalloc(&da, 64 * 8 * 2); hipMalloc(&db, 64 * 16 * 2); hipMalloc(&di, 64 * 4); hipMalloc(&dd, 64 * 16 * 4);
    // hypotheses: kmapB(h, kg, e) = K index of B element e in lane half kg;  A groups: lane half kg stores groups gmap(h, kg, i), i = 0..3
    for (int hb = 0; hb < 2; ++hb) for (int ha = 0; ha < 2; ++ha) for (int abid = 0; abid < 2; ++abid) for (int ish = 0; ish < 2; ++ish) {
        std::vector<_Float16> a(64 * 8), b(64 * 16); std::vector<int> ix(64);
        for (int l = 0; l < 64; ++l) {
            const int r = l & 31, kg = l >> 5;
            for (int e = 0; e < 16; ++e) {
                const int kk = hb == 0 ? 16 * kg + e : (8 * kg + (e & 7) + 16 * (e >> 3));
                b[l * 16 + e] = (_Float16)B[kk][r];
            }
            unsigned id = 0;
            for (int i = 0; i < 4; ++i) {
                const int g = ha == 0 ? 4 * kg + i : (2 * kg + (i & 1) + 4 * (i >> 1));
                a[l * 8 + 2 * i] = (_Float16)A[r][4 * g + pos[r][g][0]];
                a[l * 8 + 2 * i + 1] = (_Float16)A[r][4 * g + pos[r][g][1]];
                id |= (unsigned)(pos[r][g][0] | (pos[r][g][1] << 2)) << (4 * i);
            }
            ix[l] = (int)(ish ? (id << 16) : id);
        }
        hipMemcpy(da, a.data(), 64 * 8 * 2, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 64 * 16 * 2, hipMemcpyHostToDevice);
        hipMemcpy(di, ix.data(), 64 * 4, hipMemcpyHostToDevice);
        k<<<1, 64>>>(da, db, di, dd, abid);
        std::vector<float> out(64 * 16);
        hipMemcpy(out.data(), dd, 64 * 16 * 4, hipMemcpyDeviceToHost);
        // D layout of 32x32 MFMA: lane l, reg i -> row (i&3) + 8*(i>>2) + 4*(l>>5), col l&31
        double err = 0;
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;
            err = fmax(err, fabs(out[l * 16 + i] - D[row][col]));
        }
        printf("hypB=%d hypA=%d abid=%d idx_in_high16=%d : max err %g %s\n", hb, ha, abid, ish, err, err == 0 ? "<== MATCH" : "");
    }
    return 0;
}
