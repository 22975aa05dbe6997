// read_ceiling.hip -- what a read-only stream shaped like the tracking kernel's reaches on the chip when the arithmetic is (almost)
// removed: 8192 workgroups of 256 threads, each sweeping its own 200 KB window with 16-byte loads per lane, two (or PF) loads in
// flight per lane, 8 waves per SIMD; FMAS extra v_fma_f32 per loaded dword model the multicorrelator's instruction load.
// Build: hipcc --offload-arch=gfx950 -O3 read_ceiling.hip -o bin/read_ceiling ; run: bin/read_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
#define GL __attribute__((address_space(1)))

// LOOKUPS: per loaded sample and "tap" an index from float arithmetic on the sample number, a dependent ds_read_b32 and two FMAs with it
// (the multicorrelator's inner step); PROLOGUE: fill a 1087-float LDS table from global memory and evaluate a sincosf before streaming
template <int PF, int FMAS, bool NT, int LOOKUPS = 0, bool PROLOGUE = false>
__global__ __launch_bounds__(256, 8) void sweep(const float* __restrict__ buf, float* __restrict__ out, int n_pairs /* f4 per window */,
    const float* __restrict__ table = nullptr, float step = 0.0409f)
{
    __shared__ float lds[LOOKUPS || PROLOGUE ? 1152 : 1];
    float pro = 0.f;
    if (PROLOGUE)
        {
            for (int k = threadIdx.x; k < 1087; k += 256) lds[k] = table[(k + blockIdx.x) % 1023];
            float sn, cs;
            sincosf((float)blockIdx.x * 1e-3f + threadIdx.x, &sn, &cs);
            pro = sn + cs;
            __syncthreads();
        }
    else if (LOOKUPS)
        {
            for (int k = threadIdx.x; k < 1152; k += 256) lds[k] = 1.0f;
            __syncthreads();
        }
    // same job mapping as trk_multicorrelator_kernel: blocks with equal (blockIdx % 8) walk neighbouring windows
    const size_t win = (size_t)blockIdx.x;
    const GL f4* p = (const GL f4*)(buf) + win * (size_t)n_pairs + threadIdx.x;
    const int n_it = n_pairs / 256;
    f4 q[PF];
    float acc0 = pro, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
    for (int u = 0; u < PF; u++) q[u] = NT ? __builtin_nontemporal_load(p + (size_t)min(u, n_it - 1) * 256) : p[(size_t)min(u, n_it - 1) * 256];
    for (int it = 0; it < n_it; it += PF)
        {
#pragma unroll
            for (int u = 0; u < PF; u++)
                {
                    const f4 v = q[u];
                    const int nx = min(it + PF + u, n_it - 1);
                    q[u] = NT ? __builtin_nontemporal_load(p + (size_t)nx * 256) : p[(size_t)nx * 256];
                    acc0 += v.x, acc1 += v.y, acc2 += v.z, acc3 += v.w;
                    if (LOOKUPS)
                        {
                            const float s0 = step * (float)((it + u) * 512 + 2 * threadIdx.x), s1 = s0 + step;
#pragma unroll
                            for (int t = 0; t < LOOKUPS; t++)
                                {
                                    const int i0 = (int)floorf((s0 + (0.5f * t)) - 0.25f), i1 = (int)floorf((s1 + (0.5f * t)) - 0.25f);
                                    const float c0 = lds[i0], c1 = lds[i1];
                                    acc0 = fmaf(v.x, c0, acc0);
                                    acc1 = fmaf(v.y, c0, acc1);
                                    acc2 = fmaf(v.z, c1, acc2);
                                    acc3 = fmaf(v.w, c1, acc3);
                                }
                        }
#pragma unroll
                    for (int f = 0; f < FMAS; f++)
                        {
                            acc0 = fmaf(acc0, 1.0001f, v.x);
                            acc1 = fmaf(acc1, 1.0001f, v.y);
                            acc2 = fmaf(acc2, 1.0001f, v.z);
                            acc3 = fmaf(acc3, 1.0001f, v.w);
                        }
                }
        }
    const float s = acc0 + acc1 + acc2 + acc3;
    if (s == 1.2345e-33f) out[blockIdx.x] = s;
}

template <int PF, int FMAS, bool NT, int LOOKUPS = 0, bool PROLOGUE = false>
static void run(const char* name, const float* d, float* o, int n_win, int n_pairs)
{
    static float* table = nullptr;
    if (!table)
        {
            hipMalloc(&table, 4096);
            hipMemset(table, 0, 4096);
        }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((sweep<PF, FMAS, NT, LOOKUPS, PROLOGUE>), dim3(n_win), dim3(256), 0, 0, d, o, n_pairs, table, 0.0409f);
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((sweep<PF, FMAS, NT, LOOKUPS, PROLOGUE>), dim3(n_win), dim3(256), 0, 0, d, o, n_pairs, table, 0.0409f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = (double)n_win * n_pairs * 16.0;
    printf("%-34s %8.4f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
}

__global__ void fill_random(float* p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        {
            unsigned h = (unsigned)i * 2654435761u + (unsigned)(i >> 32) * 40503u;
            h ^= h >> 15;
            h *= 2246822519u;
            h ^= h >> 13;
            p[i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;  // noise-like samples in [-1, 1)
        }
}

int main(int argc, char** argv)
{
    const int n_win = 8192, n_pairs = 12544;  // 12544 f4 = 25088 samples of 8 bytes (49 x 256): 1.644 GB in total
    float *d, *o;
    const size_t bytes = (size_t)n_win * n_pairs * 16;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&o, n_win * 4) != hipSuccess) return 1;
    hipMemset(d, 0, bytes);
    if (argc > 1)
        {
            // random data instead of zeros: what the IQ stream looks like to the memory system
            hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, d, bytes / 4);
            hipDeviceSynchronize();
            printf("buffer filled with noise\n");
        }
    run<2, 0, false>("pf2, sums only", d, o, n_win, n_pairs);
    run<2, 0, true>("pf2, sums only, nontemporal", d, o, n_win, n_pairs);
    run<4, 0, false>("pf4, sums only", d, o, n_win, n_pairs);
    run<2, 4, false>("pf2 + 16 fma per 16 bytes", d, o, n_win, n_pairs);
    run<2, 8, false>("pf2 + 32 fma per 16 bytes", d, o, n_win, n_pairs);
    run<2, 12, false>("pf2 + 48 fma per 16 bytes", d, o, n_win, n_pairs);
    run<2, 14, false>("pf2 + 56 fma per 16 bytes", d, o, n_win, n_pairs);
    run<2, 14, true>("pf2 + 56 fma per 16 bytes, nt", d, o, n_win, n_pairs);
    run<4, 14, false>("pf4 + 56 fma per 16 bytes", d, o, n_win, n_pairs);
    run<2, 6, false, 3>("pf2 + 24 fma + 3 lookups", d, o, n_win, n_pairs);
    run<2, 6, true, 3>("pf2 + 24 fma + 3 lookups, nt", d, o, n_win, n_pairs);
    run<2, 6, false, 3, true>("pf2 + 24 fma + 3 lookups + prologue", d, o, n_win, n_pairs);
    run<2, 6, true, 3, true>("the same, nt", d, o, n_win, n_pairs);
    run<2, 6, true, 5, true>("pf2 + 24 fma + 5 lookups + prologue, nt", d, o, n_win, n_pairs);
    {
        // a long back-to-back run of the model closest to the tracking kernel, launch by launch: does it slow down as the real one does
        // (242 us for the first ten launches, 262-289 us some 3 ms into a run)?
        static float* table = nullptr;
        hipMalloc(&table, 4096);
        hipMemset(table, 0, 4096);
        const int reps = 60;
        hipEvent_t ev[reps + 1];
        for (int r = 0; r <= reps; r++) hipEventCreate(&ev[r]);
        hipEventRecord(ev[0]);
        for (int r = 0; r < reps; r++)
            {
                hipLaunchKernelGGL((sweep<2, 10, true, 3, true>), dim3(n_win), dim3(256), 0, 0, d, o, n_pairs, table, 0.0409f);
                hipEventRecord(ev[r + 1]);
            }
        hipEventSynchronize(ev[reps]);
        printf("pf2 + 40 fma + 3 lookups + prologue, nt, 60 launches back to back (us):");
        for (int r = 0; r < reps; r++)
            {
                float ms = 0;
                hipEventElapsedTime(&ms, ev[r], ev[r + 1]);
                printf(" %.0f", ms * 1e3);
            }
        printf("\n");
    }
    return 0;
}
