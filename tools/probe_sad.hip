// Probe of the gfx950 packed-SAD instructions: semantics check and issue-rate microbenchmark.
// Build: hipcc --offload-arch=gfx950 -O3 -o probe_sad probe_sad.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void sem_kernel(const uint32_t* a, const uint32_t* b, uint32_t* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s0 = a[2 * i], s0h = a[2 * i + 1], s1 = b[i];
    uint64_t s064 = ((uint64_t)s0h << 32) | s0;
    out[8 * i + 0] = __builtin_amdgcn_sad_u8(s0, s1, 7u);
    out[8 * i + 1] = __builtin_amdgcn_msad_u8(s0, s1, 7u);
    uint64_t q = __builtin_amdgcn_qsad_pk_u16_u8(s064, s1, 0x0001000200030004ull);
    out[8 * i + 2] = (uint32_t)q; out[8 * i + 3] = (uint32_t)(q >> 32);
    uint64_t m = __builtin_amdgcn_mqsad_pk_u16_u8(s064, s1, 0x0001000200030004ull);
    out[8 * i + 4] = (uint32_t)m; out[8 * i + 5] = (uint32_t)(m >> 32);
    out[8 * i + 6] = __builtin_amdgcn_sad_hi_u8(s0, s1, 7u);
    out[8 * i + 7] = __builtin_amdgcn_sad_u16(s0, s1, 7u);
}

template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(uint32_t* out, int iters, uint32_t seed) {
    uint32_t x = seed + threadIdx.x * 2654435761u + blockIdx.x;
    uint32_t r0 = x, r1 = x * 3 + 1, r2 = x * 5 + 2, r3 = x * 7 + 3;
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    uint64_t q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0, q7 = 0;
    uint64_t w0 = ((uint64_t)r1 << 32) | r0, w1 = ((uint64_t)r3 << 32) | r2;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // v_sad_u8, 8 independent chains
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_sad_u8(r0, r1, a0); a1 = __builtin_amdgcn_sad_u8(r1, r2, a1);
                a2 = __builtin_amdgcn_sad_u8(r2, r3, a2); a3 = __builtin_amdgcn_sad_u8(r3, r0, a3);
                a4 = __builtin_amdgcn_sad_u8(r0, r2, a4); a5 = __builtin_amdgcn_sad_u8(r1, r3, a5);
                a6 = __builtin_amdgcn_sad_u8(r2, r0, a6); a7 = __builtin_amdgcn_sad_u8(r3, r1, a7);
            }
        } else if (MODE == 1) {   // v_qsad_pk_u16_u8, 8 independent chains
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                q0 = __builtin_amdgcn_qsad_pk_u16_u8(w0, r0, q0); q1 = __builtin_amdgcn_qsad_pk_u16_u8(w1, r1, q1);
                q2 = __builtin_amdgcn_qsad_pk_u16_u8(w0, r2, q2); q3 = __builtin_amdgcn_qsad_pk_u16_u8(w1, r3, q3);
                q4 = __builtin_amdgcn_qsad_pk_u16_u8(w0, r1, q4); q5 = __builtin_amdgcn_qsad_pk_u16_u8(w1, r2, q5);
                q6 = __builtin_amdgcn_qsad_pk_u16_u8(w0, r3, q6); q7 = __builtin_amdgcn_qsad_pk_u16_u8(w1, r0, q7);
            }
        } else if (MODE == 2) {   // alignbyte + sad_u8 pairs
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(r1, r0, 1), r2, a0);
                a1 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(r2, r1, 2), r3, a1);
                a2 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(r3, r2, 3), r0, a2);
                a3 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(r0, r3, 1), r1, a3);
                a4 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(r1, r0, 2), r3, a4);
                a5 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(r2, r1, 3), r0, a5);
                a6 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(r3, r2, 1), r1, a6);
                a7 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(r0, r3, 2), r2, a7);
            }
        } else if (MODE == 3) {   // v_pk_add_u16-ish: plain v_add_u32 as the VALU baseline
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 += r0; a1 += r1; a2 += r2; a3 += r3; a4 += r0 ^ 1; a5 += r1 ^ 2; a6 += r2 ^ 3; a7 += r3 ^ 4;
            }
        }
    }
    uint32_t s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 +
        (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) + (uint32_t)((q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static uint32_t sad4(uint32_t a, uint32_t b) {
    uint32_t s = 0;
    for (int k = 0; k < 4; ++k) { int x = (a >> (8 * k)) & 255, y = (b >> (8 * k)) & 255; s += abs(x - y); }
    return s;
}

template <int MODE> double run_rate(uint32_t* d_out, int iters, int blocks) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<MODE><<<blocks, 256>>>(d_out, iters, 123u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    rate_kernel<MODE><<<blocks, 256>>>(d_out, iters, 456u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main() {
    const int n = 4096;
    std::vector<uint32_t> a(2 * n), b(n), out(8 * n);
    uint32_t s = 12345;
    for (auto& v : a) { s = s * 1664525u + 1013904223u; v = s; }
    for (auto& v : b) { s = s * 1664525u + 1013904223u; v = s; }
    // force zero bytes into some operands to expose the masked forms
    for (int i = 0; i < n; i += 3) b[i] &= 0xFF00FFFFu;
    for (int i = 1; i < n; i += 5) a[2 * i] &= 0xFFFF00FFu;
    uint32_t *da, *db, *dout;
    CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dout, out.size() * 4));
    CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    sem_kernel<<<n / 256, 256>>>(da, db, dout, n);
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    int bad_sad = 0, bad_q = 0, msad_eq_sad = 0, mq_eq_q = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t s0 = a[2 * i], s0h = a[2 * i + 1], s1 = b[i];
        uint64_t w = ((uint64_t)s0h << 32) | s0;
        if (out[8 * i] != sad4(s0, s1) + 7) bad_sad++;
        uint64_t q = ((uint64_t)out[8 * i + 3] << 32) | out[8 * i + 2];
        uint64_t acc = 0x0001000200030004ull, exp = 0;
        for (int k = 0; k < 4; ++k) {
            uint32_t win = (uint32_t)(w >> (8 * k));
            uint64_t v = (sad4(win, s1) + ((acc >> (16 * k)) & 0xFFFF)) & 0xFFFF;
            exp |= v << (16 * k);
        }
        if (q != exp) { if (bad_q < 4) printf("qsad mismatch i=%d got %016llx exp %016llx\n", i, (unsigned long long)q, (unsigned long long)exp); bad_q++; }
        if (out[8 * i + 1] == out[8 * i]) msad_eq_sad++;
        uint64_t m = ((uint64_t)out[8 * i + 5] << 32) | out[8 * i + 4];
        if (m == q) mq_eq_q++;
    }
    printf("semantics: sad_u8 mismatches=%d qsad_pk_u16_u8 mismatches=%d (of %d); msad==sad in %d, mqsad==qsad in %d\n",
           bad_sad, bad_q, n, msad_eq_sad, mq_eq_q);
    // rate
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount; double ghz = prop.clockRate / 1e6;
    printf("device %s CUs=%d clock=%.2f GHz\n", prop.name, cus, ghz);
    int blocks = cus * 8, iters = 2000;
    uint32_t* d_out; CK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    const char* names[4] = {"v_sad_u8", "v_qsad_pk_u16_u8", "alignbyte+sad_u8", "v_add_u32"};
    double ms[4] = {run_rate<0>(d_out, iters, blocks), run_rate<1>(d_out, iters, blocks), run_rate<2>(d_out, iters, blocks), run_rate<3>(d_out, iters, blocks)};
    for (int m = 0; m < 4; ++m) {
        double instr = (double)blocks * 4 /*waves*/ * iters * 64.0 * (m == 2 ? 2 : 1);  // wave-instructions
        double per_cu_cycle = instr / (ms[m] * 1e-3) / cus / (ghz * 1e9);
        double absd = (m == 0 || m == 2) ? 4.0 : (m == 1 ? 16.0 : 0.0);
        double sad_instr = (double)blocks * 4 * iters * 64.0;
        printf("%-18s %.3f ms  wave-instr/cycle/CU=%.3f  abs-diff rate=%.2f Tops/s\n", names[m], ms[m], per_cu_cycle,
               sad_instr * 64.0 * absd / (ms[m] * 1e-3) / 1e12);
    }
    return 0;
}
