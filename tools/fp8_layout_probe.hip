#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const unsigned char* a, const unsigned char* b, float* c) {   // a,b: per lane 32 bytes
    i32x8 av, bv;
    const int* ai = (const int*)(a + threadIdx.x * 32);
    const int* bi = (const int*)(b + threadIdx.x * 32);
    for (int i = 0; i < 8; ++i) { av[i] = ai[i]; bv[i] = bi[i]; }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int i = 0; i < 4; ++i) c[threadIdx.x * 4 + i] = acc[i];
}
// e4m3 OCP encode of small integers / halves exactly
static unsigned char enc(float v) {
    if (v == 0) return 0;
    unsigned char s = v < 0 ? 0x80 : 0; v = fabsf(v);
    int e; float m = frexpf(v, &e);          // v = m * 2^e, m in [0.5,1)
    int E = e - 1 + 7; float frac = m * 2 - 1; // 1.frac
    int M = (int)roundf(frac * 8);
    return s | (E << 3) | M;
}
static int kmap(int h, int g, int j) {
    if (h == 0) return 32 * g + j;
    if (h == 1) return 16 * g + (j < 16 ? j : 64 + j - 16);
    return 8 * g + 32 * (j / 8) + j % 8;
}
int main() {
    float A[16][128], B[128][16];
    srand(1);
    float vals[] = {0, 1, -1, 2, -2, 0.5f, -0.5f, 3, 1.5f, -3, 4};
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 128; ++kk) A[i][kk] = vals[rand() % 11];
    for (int kk = 0; kk < 128; ++kk) for (int j = 0; j < 16; ++j) B[kk][j] = vals[rand() % 11];
    float C[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int kk = 0; kk < 128; ++kk) s += A[i][kk] * B[kk][j]; C[i][j] = s; }
    unsigned char *da, *db; float* dc;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dc, 64 * 4 * 4);
    for (int h = 0; h < 3; ++h) {
        unsigned char ha[64 * 32], hb[64 * 32];
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
            int kk = kmap(h, l / 16, j);
            ha[l * 32 + j] = enc(A[l % 16][kk]);
            hb[l * 32 + j] = enc(B[kk][l % 16]);
        }
        hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc);
        float hc[64 * 4]; hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
        int bad = 0, badT = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            int col = l & 15, row = (l >> 4) * 4 + r;
            if (hc[l * 4 + r] != C[row][col]) ++bad;
            if (hc[l * 4 + r] != C[col][row]) ++badT;
        }
        printf("hypothesis %d: mismatches %d (transposed-output reading: %d)\n", h, bad, badT);
    }
    return 0;
}
