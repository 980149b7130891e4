// micro: (1) does global_load_dwordx4 work at 4-byte-aligned (not 16-byte-aligned) addresses on gfx950?  (2) C/D layout of
// v_mfma_f32_16x16x4_f32 (asymmetric operands).  Build: hipcc --offload-arch=gfx950 -O3 unaligned_x4.hip -o unaligned_x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void k_un(const float* p, float* out, int off)
{
    const f4v v = *reinterpret_cast<const f4v*>(p + off + 5 * threadIdx.x);   // 20-byte stride: every alignment class mod 16
    out[4 * threadIdx.x + 0] = v.x; out[4 * threadIdx.x + 1] = v.y; out[4 * threadIdx.x + 2] = v.z; out[4 * threadIdx.x + 3] = v.w;
}
__global__ void k_mfma(const float* A, const float* B, float* D)   // A [16][4] row-major (i,k), B [4][16] (k,j)
{
    const int l = threadIdx.x;
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];   // hypothesis: row = 4 (l >> 4) + r, col = l & 15
}
int main()
{
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 4096 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int off = 0; off < 4; ++off) {
        k_un<<<1, 64>>>(d, o, off);
        std::vector<float> r(256);
        if (hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost) != hipSuccess) { printf("unaligned load: FAULT\n"); return 1; }
        for (int t = 0; t < 64; ++t) for (int c = 0; c < 4; ++c) if (r[4 * t + c] != (float)(off + 5 * t + c)) ++bad;
    }
    printf("unaligned dwordx4 loads: %s (%d wrong)\n", bad ? "WRONG" : "ok", bad);
    std::vector<float> A(64), B(64), D(256), W(256, 0.f);
    for (int i = 0; i < 64; ++i) { A[i] = (float)((i * 7) % 11) - 3.f; B[i] = (float)((i * 5) % 13) - 6.f; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) W[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    float *dA, *dB, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int badm = 0;
    for (int i = 0; i < 256; ++i) if (D[i] != W[i]) ++badm;
    printf("mfma_f32_16x16x4 C/D layout row = 4 (l >> 4) + r, col = l & 15: %s (%d wrong)\n", badm ? "WRONG" : "ok", badm);
    return bad || badm;
}
