// VALU issue cost per instruction class on gfx950, measured the way the tracer's kernels run:
// every CU busy, up to 8 waves per SIMD, each wave a stream of ONE instruction type (8, 2 or 1
// independent dependency chains).  Output: SIMD-cycles per wave64 instruction at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue && ./valu_issue
// Round-1 result: profiles/microbench/r01_valu_issue.txt.  Reading: the SIMD-32 executes an
// fma/mul/add-class wave64 instruction in 2 cycles (spec), a busy chip sustains 2.3-2.9; compares,
// min/max and DPP forms cost ~4.2, integer ops ~3.1, transcendentals ~8.1, f64 fma ~5.2; a single
// wave alone sustains one instruction per ~3.3 cycles even on a fully dependent chain.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define BODY8(INS) \
    asm volatile(INS : "+v"(x0) : "v"(a), "v"(b)); asm volatile(INS : "+v"(x1) : "v"(a), "v"(b)); \
    asm volatile(INS : "+v"(x2) : "v"(a), "v"(b)); asm volatile(INS : "+v"(x3) : "v"(a), "v"(b)); \
    asm volatile(INS : "+v"(x4) : "v"(a), "v"(b)); asm volatile(INS : "+v"(x5) : "v"(a), "v"(b)); \
    asm volatile(INS : "+v"(x6) : "v"(a), "v"(b)); asm volatile(INS : "+v"(x7) : "v"(a), "v"(b));
#define BODY2(INS) \
    asm volatile(INS : "+v"(x0) : "v"(a), "v"(b)); asm volatile(INS : "+v"(x1) : "v"(a), "v"(b)); \
    asm volatile(INS : "+v"(x0) : "v"(a), "v"(b)); asm volatile(INS : "+v"(x1) : "v"(a), "v"(b)); \
    asm volatile(INS : "+v"(x0) : "v"(a), "v"(b)); asm volatile(INS : "+v"(x1) : "v"(a), "v"(b)); \
    asm volatile(INS : "+v"(x0) : "v"(a), "v"(b)); asm volatile(INS : "+v"(x1) : "v"(a), "v"(b));
#define BODY1(INS) \
    asm volatile(INS : "+v"(x0) : "v"(a), "v"(b)); asm volatile(INS : "+v"(x0) : "v"(a), "v"(b)); \
    asm volatile(INS : "+v"(x0) : "v"(a), "v"(b)); asm volatile(INS : "+v"(x0) : "v"(a), "v"(b)); \
    asm volatile(INS : "+v"(x0) : "v"(a), "v"(b)); asm volatile(INS : "+v"(x0) : "v"(a), "v"(b)); \
    asm volatile(INS : "+v"(x0) : "v"(a), "v"(b)); asm volatile(INS : "+v"(x0) : "v"(a), "v"(b));
#define KERNEL(NAME, BODY, INS) \
__global__ void NAME(float *out, float a, float b, int iters) { \
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    for (int i = 0; i < iters; ++i) { BODY(INS) } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; }
KERNEL(fma8, BODY8, "v_fma_f32 %0, %0, %1, %2")
KERNEL(fma2, BODY2, "v_fma_f32 %0, %0, %1, %2")
KERNEL(fma1, BODY1, "v_fma_f32 %0, %0, %1, %2")
KERNEL(mul8, BODY8, "v_mul_f32 %0, %0, %1")
KERNEL(mul1, BODY1, "v_mul_f32 %0, %0, %1")
KERNEL(add8, BODY8, "v_add_f32 %0, %0, %2")
KERNEL(sub8, BODY8, "v_sub_f32 %0, %0, %2")
KERNEL(max8, BODY8, "v_max_f32 %0, %0, %2")
KERNEL(mov8, BODY8, "v_mov_b32 %0, %1")
KERNEL(cmp8, BODY8, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL(xor8, BODY8, "v_xor_b32 %0, %0, %1")
KERNEL(addu8, BODY8, "v_add_u32 %0, %0, %1")
KERNEL(fmac8, BODY8, "v_fmac_f32 %0, %1, %2")
KERNEL(sqrt8, BODY8, "v_sqrt_f32 %0, %0")
KERNEL(rcp8, BODY8, "v_rcp_f32 %0, %0")
KERNEL(dpp8, BODY8, "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
__global__ void fma64(double *out, double a, double b, int iters) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x4) : "v"(a), "v"(b)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x5) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x6) : "v"(a), "v"(b)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x7) : "v"(a), "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; }
// packed fp32 (two IEEE operations per lane and instruction, operands in VGPR pairs)
#define KERNEL_PK(NAME, INS) \
__global__ void NAME(double *out, double a, double b, int iters) { \
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    for (int i = 0; i < iters; ++i) { BODY8(INS) } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; }
KERNEL_PK(pkmul8, "v_pk_mul_f32 %0, %0, %1")
KERNEL_PK(pkadd8, "v_pk_add_f32 %0, %0, %2")
KERNEL_PK(pkfma8, "v_pk_fma_f32 %0, %0, %1, %2")
template <typename K, typename T> void run(const char *name, K k, T *d, int iters, int blocks_per_cu, T a, T b)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu * 4;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, a, b, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, a, b, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)grid * 4 * iters * 8;
    const double cyc = ms * 1e-3 * 2.4e9 * 1024 / insts;   // SIMD-cycles per wave instruction at 2.4 GHz
    printf("%-8s waves/SIMD<=%d  %.3f ms  %.3g wave-inst/s  %.2f cycles/inst/SIMD\n", name, blocks_per_cu, ms, insts / (ms * 1e-3), cyc);
}
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 8 * 4 * 256 * 8);
    const int it = 4000;
    for (int w : {8, 2, 1}) {
        run("fma x8", fma8, d, it, w, 1.0001f, 0.5f);
        run("fma x2", fma2, d, it, w, 1.0001f, 0.5f);
        run("fma x1", fma1, d, it, w, 1.0001f, 0.5f);
    }
    run("mul x8", mul8, d, it, 8, 1.0001f, 0.5f);
    run("mul x1", mul1, d, it, 8, 1.0001f, 0.5f);
    run("add x8", add8, d, it, 8, 1.0001f, 0.5f);
    run("sub x8", sub8, d, it, 8, 1.0001f, 0.5f);
    run("max x8", max8, d, it, 8, 1.0001f, 0.5f);
    run("fmac x8", fmac8, d, it, 8, 1.0001f, 0.5f);
    run("mov x8", mov8, d, it, 8, 1.0001f, 0.5f);
    run("cmp", cmp8, d, it, 8, 1.0001f, 0.5f);
    run("xor", xor8, d, it, 8, 1.0001f, 0.5f);
    run("add_u32", addu8, d, it, 8, 1.0001f, 0.5f);
    run("sqrt", sqrt8, d, it, 8, 1.0001f, 0.5f);
    run("rcp", rcp8, d, it, 8, 1.0001f, 0.5f);
    run("add dpp", dpp8, d, it, 8, 1.0001f, 0.5f);
    run("fma f64", fma64, (double *)d, it, 8, 1.0001, 0.5);
    run("pk_mul", pkmul8, (double *)d, it, 8, 1.0001, 0.5);
    run("pk_add", pkadd8, (double *)d, it, 8, 1.0001, 0.5);
    run("pk_fma", pkfma8, (double *)d, it, 8, 1.0001, 0.5);
    return 0;
}
