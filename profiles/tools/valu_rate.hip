// valu_rate.hip -- issue rates of the instructions the tracking loop is made of, on one SIMD, as a function of the waves per SIMD.
// Each kernel runs N_ITER x 64 independent instances of one instruction (16 accumulator chains per lane, so no dependency stalls)
// in `waves` waves per SIMD on every CU; cycles per wave-instruction per SIMD = elapsed * clock / (instructions per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHAINS 16
#define REP 64
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int n_iter, float seed)
{
    float a[CHAINS];
    int ai[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++)
        {
            a[c] = seed + threadIdx.x * 0.001f + c;
            ai[c] = threadIdx.x + c;
        }
    const float m = 1.0001f, b = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[CHAINS / 2];
#pragma unroll
    for (int c = 0; c < CHAINS / 2; c++) p[c] = f2{a[2 * c], a[2 * c + 1]};
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    for (int it = 0; it < n_iter; it++)
        {
#pragma unroll
            for (int r = 0; r < REP / CHAINS; r++)
                {
#pragma unroll
                    for (int c = 0; c < CHAINS; c++)
                        {
                            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(b));
                            if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                            if (OP == 2) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(ai[c]) : "v"(a[c]));
                            if (OP == 3) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(ai[c]) : "v"(ai[(c + 1) % CHAINS]));
                            if (OP == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[c % (CHAINS / 2)]) : "v"(p[(c + 1) % (CHAINS / 2)]));
                            if (OP == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c % (CHAINS / 2)]) : "v"(p[(c + 1) % (CHAINS / 2)]));
                            if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[c % (CHAINS / 2)]) : "v"(p[(c + 1) % (CHAINS / 2)]));
                            if (OP == 7) asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(a[c]) : "v"(b));
                            if (OP == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(m));
                            if (OP == 9) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[c]) : "v"(ai[c]));
                            if (OP == 10)
                                {
                                    // the tracking loop's tap step: add, subrev, cvt_flr, lshl_add, ds_read_b32, 2 fmac
                                    float t, v;
                                    int idx, addr;
                                    asm volatile("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(a[c]), "v"(b));
                                    asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(t) : "v"(m));
                                    asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(idx) : "v"(t));
                                    asm volatile("v_and_b32 %0, 1023, %0" : "+v"(idx));
                                    asm volatile("v_lshlrev_b32 %0, 2, %1" : "=v"(addr) : "v"(idx));
                                    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
                                    asm volatile("s_waitcnt lgkmcnt(8)");
                                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[c]) : "v"(v), "v"(m));
                                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[(c + 1) % CHAINS]) : "v"(v), "v"(b));
                                }
                        }
                }
        }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += a[c] + (float)ai[c];
#pragma unroll
    for (int c = 0; c < CHAINS / 2; c++) s += p[c].x + p[c].y;
    if (s == 123.456f) out[0] = s;
}

template <int OP>
static void run(const char* name, int per_inst, float* d_out, int n_cus, double clock_ghz)
{
    for (int waves : {1, 2, 4, 8})
        {
            const int n_iter = 2000;
            const int blocks = n_cus * waves;  // 256-thread blocks: one wave per SIMD each
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 16384, 0, d_out, 10, 1.0f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 16384, 0, d_out, n_iter, 1.0f);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double inst_per_simd = (double)waves * n_iter * REP * per_inst;
            std::printf("%-28s waves/SIMD %d: %.2f cycles per wave-instruction (at %.2f GHz)\n", name, waves, ms * 1e-3 * clock_ghz * 1e9 / inst_per_simd, clock_ghz);
        }
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate * 1e-6;
    float* d_out;
    hipMalloc(&d_out, 64);
    const int n = prop.multiProcessorCount;
    std::printf("%s, %d CUs, nominal clock %.2f GHz (cycle figures assume it; the chip may run lower under load)\n", prop.name, n, ghz);
    run<0>("v_fma_f32", 1, d_out, n, ghz);
    run<1>("v_add_f32", 1, d_out, n, ghz);
    run<8>("v_mul_f32", 1, d_out, n, ghz);
    run<7>("v_subrev_f32", 1, d_out, n, ghz);
    run<2>("v_cvt_flr_i32_f32", 1, d_out, n, ghz);
    run<9>("v_cvt_f32_i32", 1, d_out, n, ghz);
    run<3>("v_lshl_add_u32", 1, d_out, n, ghz);
    run<4>("v_pk_fma_f32", 1, d_out, n, ghz);
    run<5>("v_pk_add_f32", 1, d_out, n, ghz);
    run<6>("v_pk_mul_f32", 1, d_out, n, ghz);
    run<10>("tap step (8 instr + ds_read)", 9, d_out, n, ghz);
    return 0;
}
