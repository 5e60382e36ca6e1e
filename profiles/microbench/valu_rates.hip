// valu_rates.hip -- issue cost of the vector instructions k_e16 is made of: cycles per wave-instruction of a long run of one
// kind, eight independent chains each (so that latency does not show), at 1 and 3 wavefronts per SIMD.
//   hipcc -O3 --offload-arch=gfx950 profiles/microbench/valu_rates.hip -o profiles/microbench/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY_FMA(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define BODY_FMAC_DPP(i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
#define BODY_MOVDPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(ia[i]) : "v"(ib));
#define BODY_MOV64DPP(i) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
#define BODY_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia[i]) : "v"(ib) : );
#define BODY_ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define BODY_MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define BODY_RSQ(i) asm volatile("v_rsq_f64 %0, %1" : "=v"(a[i]) : "v"(b));
#define BODY_RCP(i) asm volatile("v_rcp_f64 %0, %1" : "=v"(a[i]) : "v"(b));
#define BODY_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(b));
#define BODY_ADD32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
#define BODY_XOR32(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
#define BODY_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(ia[i]) : "v"(ib));
#define BODY_PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define BODY_FMA32(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fa[i]) : "v"(fb), "v"(fc));
#define BODY_SNOP(i) asm volatile("s_nop 0");
#define BODY_MFMA(i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0);

#define KERNEL(NAME, BODY)                                                                                 \
    __global__ void k_##NAME(double* out, const int reps, long long* cyc) {                                \
        double a[8], b = 1.0 + 1e-9 * threadIdx.x, c = 0.999999;                                           \
        int ia[8], ib = threadIdx.x;                                                                       \
        float fa[8], fb = 1.0f, fc = 0.5f;                                                                 \
        typedef double v4d __attribute__((ext_vector_type(4)));                                            \
        v4d acc = {0, 0, 0, 0};                                                                            \
        for (int i = 0; i < 8; ++i) { a[i] = i + threadIdx.x; ia[i] = i; fa[i] = i; }                      \
        asm volatile("v_cmp_gt_i32 vcc, 32, %0" ::"v"(ib) : "vcc");                                        \
        const long long t0 = __builtin_readcyclecounter();                                                 \
        for (int r = 0; r < reps; ++r) { REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY) }                     \
        const long long t1 = __builtin_readcyclecounter();                                                 \
        double s = acc[0] + acc[1] + acc[2] + acc[3];                                                      \
        for (int i = 0; i < 8; ++i) s += a[i] + ia[i] + fa[i];                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                    \
        if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                           \
    }
KERNEL(fma_f64, BODY_FMA)
KERNEL(fmac_f64_dpp, BODY_FMAC_DPP)
KERNEL(mov_b32_dpp, BODY_MOVDPP)
KERNEL(mov_b64_dpp, BODY_MOV64DPP)
KERNEL(cndmask_b32, BODY_CNDMASK)
KERNEL(add_f64, BODY_ADD)
KERNEL(mul_f64, BODY_MUL)
KERNEL(rsq_f64, BODY_RSQ)
KERNEL(rcp_f64, BODY_RCP)
KERNEL(mov_b64, BODY_MOV64)
KERNEL(add_u32, BODY_ADD32)
KERNEL(xor_b32, BODY_XOR32)
KERNEL(alignbit_b32, BODY_ALIGNBIT)
KERNEL(pk_fma_f32, BODY_PKFMA32)
KERNEL(fma_f32, BODY_FMA32)
KERNEL(s_nop, BODY_SNOP)
KERNEL(mfma_f64_16x16x4, BODY_MFMA)

int main() {
    double* d;
    long long* c;
    hipMalloc(&d, 1 << 24);
    hipMalloc(&c, 8);
    const int reps = 2000;
#define RUN(NAME)                                                                                          \
    for (int waves = 1; waves <= 3; waves += 2) {                                                          \
        hipEvent_t e0, e1;                                                                                 \
        hipEventCreate(&e0);                                                                               \
        hipEventCreate(&e1);                                                                               \
        k_##NAME<<<256 * waves, 256>>>(d, reps, c);                                                        \
        hipEventRecord(e0);                                                                                \
        k_##NAME<<<256 * waves, 256>>>(d, reps, c);                                                        \
        hipEventRecord(e1);                                                                                \
        hipEventSynchronize(e1);                                                                           \
        float ms;                                                                                          \
        hipEventElapsedTime(&ms, e0, e1);                                                                  \
        long long cy;                                                                                      \
        hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);                                                       \
        printf("%-18s %d wavefront(s)/SIMD: %6.2f ns per wave-instruction per SIMD, %5.2f counter ticks per instruction of one wave\n", #NAME, waves, \
               ms * 1e6 / (reps * 32.0 * waves), (double)cy / (reps * 32.0));                              \
    }
    RUN(fma_f64) RUN(fmac_f64_dpp) RUN(mov_b32_dpp) RUN(mov_b64_dpp) RUN(cndmask_b32) RUN(add_f64) RUN(mul_f64) RUN(rsq_f64) RUN(rcp_f64)
    RUN(mov_b64) RUN(add_u32) RUN(xor_b32) RUN(alignbit_b32) RUN(pk_fma_f32) RUN(fma_f32) RUN(s_nop) RUN(mfma_f64_16x16x4)
    return 0;
}
