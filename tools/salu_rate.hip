// Dev tool (GPU box): how many scalar ALU instructions does a CU of gfx950 issue per clock?  Waves that do nothing but independent
// s_add_u32 -- one wave per SIMD, then two, four -- timed with HIP events; the clock from clock64 against the 100 MHz wall_clock64.
//   hipcc --offload-arch=gfx950 -O2 tools/salu_rate.hip -o course-assignment-danielhalachev_amd/build/salu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void salu_chain(uint32_t *out, int iters) {
    uint32_t a = 1, b = 2, c = 3, d = 4, e = 5, f = 6, g = 7, h = 8;
    for (int i = 0; i < iters; i++)   // 16 x 8 = 128 independent-enough scalar adds per trip (eight chains)
        asm volatile(REP16("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                           "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n")
                     : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h));
    if (threadIdx.x == 0) out[blockIdx.x] = a + b + c + d + e + f + g + h;
}
__global__ void clock_probe(unsigned long long *out, uint32_t *sink) {
    // shader clocks (clock64) against the 100 MHz wall clock (wall_clock64) over a BOUNDED loop
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    uint32_t a = 1;
    for (int i = 0; i < 200000; i++) asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(a));
    out[0] = clock64() - t0; out[1] = wall_clock64() - r0; sink[0] = a;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000;
    uint32_t *out; hipMalloc(&out, 1 << 20);
    unsigned long long *clk; hipMalloc(&clk, 16);
    hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, 0, clk, out);
    unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)hc[0] / ((double)hc[1] * 10.0);   // cycles per ns
    printf("CUs %d, shader clock while spinning %.3f GHz\n", cus, ghz);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_cu : {1, 2, 4, 8, 16}) {
        const int threads = waves_per_cu >= 4 ? 256 : 64 * waves_per_cu, blocks = cus * (waves_per_cu >= 4 ? waves_per_cu / 4 : 1);
        hipLaunchKernelGGL(salu_chain, dim3(blocks), dim3(threads), 0, 0, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(salu_chain, dim3(blocks), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)blocks * (threads / 64) * iters * 128.0;
        printf("%2d waves per CU (%d x %d threads): %.3f ms, %.2f scalar instructions per ns per CU = %.2f per CU clock\n", waves_per_cu, blocks, threads, ms,
               insts / (ms * 1e6) / cus, insts / (ms * 1e6) / cus / ghz);
    }
    return 0;
}
