// Probe: range check of a STRUCTURED buffer descriptor (stride != 0) on gfx950:
// is a load with offset >= stride out of range (returns 0), and what does num_records count?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float *p, int stride_bytes, int records, int flags, float *out, int imm_test) {
    const unsigned long long a = (unsigned long long)p;
    u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) | ((unsigned)stride_bytes << 16));
    r.z = __builtin_amdgcn_readfirstlane((unsigned)records);
    r.w = __builtin_amdgcn_readfirstlane((unsigned)flags);
    const int idx = threadIdx.x / 16 - 1;       // -1 .. 2
    const int off = (threadIdx.x % 16 - 2) * 4;  // -8 .. 52 bytes
    float v;
    if (imm_test) {
        u32x2 va = {(unsigned)idx, (unsigned)(off - 16)};
        asm volatile("buffer_load_dword %0, %1, %2, 0 idxen offen offset:16\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(va), "s"(r) : "memory");
    } else {
        u32x2 va = {(unsigned)idx, (unsigned)off};
        asm volatile("buffer_load_dword %0, %1, %2, 0 idxen offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(va), "s"(r) : "memory");
    }
    out[threadIdx.x] = v;
}
int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1000.f + i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 64 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int flags : {0x00020000}) for (int imm = 0; imm < 2; ++imm) {
        // stride 40 bytes (10 floats per record), 2 records; data starts at element 100
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d + 100, 40, 2, flags, o, imm);
        float r[64];
        hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        printf("flags %x imm %d: stride 40 B, 2 records, base element 100\n", flags, imm);
        for (int i = 0; i < 4; ++i) {
            printf(" idx %2d:", i - 1);
            for (int j = 0; j < 16; ++j) printf(" %6.0f", r[16 * i + j]);
            printf("\n");
        }
    }
    return 0;
}
