// v_mfma_f64_16x16x4_f64 on gfx950: (1) the order of its internal summation, checked bit for bit against host fma
// chains; (2) its issue interval, dependent latency and whether f64 VALU work of ANOTHER wave on the same SIMD runs
// beside it.  Stand-alone: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/_bin/mfma_f64_probe
// (decides how the DEG_ELEV > 0 elevations may use the matrix pipe and stay bit-identical to the lane-per-item forms:
// DESIGN.md section 4.1b)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// D[16][16] = A[16][K] B[K][16] + C, K = 4 * ksteps, one wave
__global__ void k_mfma(const double* A, const double* B, const double* C, double* D, int ksteps)
{
    const int l = threadIdx.x, K = 4 * ksteps;
    v4d acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
    for (int s = 0; s < ksteps; ++s) {
        const double a = A[(l & 15) * K + 4 * s + (l >> 4)];
        const double b = B[(4 * s + (l >> 4)) * 16 + (l & 15)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

// mode 0: NACC independent accumulators, iters rounds of MFMAs (issue interval); mode 1: one accumulator (latency);
// mode 2: v_fma_f64 only (8 chains); mode 3: even waves MFMA (mode 0), odd waves v_fma_f64 (co-issue on one SIMD)
template <int NACC>
__global__ __launch_bounds__(512) void k_rate(double* out, int iters, int mode, long long* cyc)
{
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    double a = 1.0 + l * 1e-9, b = 1.0 - l * 1e-9;
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = i + r;
    double f[8];
    for (int i = 0; i < 8; ++i) f[i] = i + l;
    int m = mode;
    if (mode == 3) m = (w & 4) ? 2 : 0;       // waves 0-3 (first on each SIMD) MFMA, waves 4-7 (second on each SIMD) VALU
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (m == 0) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    } else if (m == 1) {
        for (int it = 0; it < iters * NACC; ++it) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
    } else {
        for (int it = 0; it < iters * NACC * 2; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = fma(f[i], a, b);
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (l == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}

static bool same(double x, double y) { return memcmp(&x, &y, 8) == 0; }

int main()
{
    int ndev = 0;
    CHECK(hipGetDeviceCount(&ndev));
    hipDeviceProp_t pr;
    CHECK(hipGetDeviceProperties(&pr, 0));
    printf("device: %s, %d CUs, clock %d kHz\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate);

    // ---------------- (1) order of summation
    for (int ksteps : {1, 6}) {
        const int K = 4 * ksteps;
        std::vector<double> A(16 * K), B(K * 16), C(256), D(256);
        srand(1234 + ksteps);
        auto rnd = []() { return (rand() / (double)RAND_MAX - 0.5) * pow(2.0, rand() % 9 - 4); };
        int n_asc = 0, n_desc = 0, n_unfused = 0, n_pair = 0, n_total = 0;
        double worst = 0.0;
        for (int trial = 0; trial < 50; ++trial) {
            for (auto& v : A) v = rnd();
            for (auto& v : B) v = rnd();
            for (auto& v : C) v = ksteps == 1 ? rnd() : 0.0;
            double *dA, *dB, *dC, *dD;
            CHECK(hipMalloc(&dA, A.size() * 8)); CHECK(hipMalloc(&dB, B.size() * 8)); CHECK(hipMalloc(&dC, 2048)); CHECK(hipMalloc(&dD, 2048));
            CHECK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, ksteps);
            CHECK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
            CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dC)); CHECK(hipFree(dD));
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double asc = C[i * 16 + j], unf = C[i * 16 + j];
                    for (int k = 0; k < K; ++k) { asc = fma(A[i * K + k], B[k * 16 + j], asc); unf = unf + A[i * K + k] * B[k * 16 + j]; }
                    double desc = C[i * 16 + j];
                    for (int s = 0; s < ksteps; ++s)       // k-steps in order, descending inside a step
                        for (int k = 4 * s + 3; k >= 4 * s; --k) desc = fma(A[i * K + k], B[k * 16 + j], desc);
                    double pw = C[i * 16 + j];
                    for (int s = 0; s < ksteps; ++s) {      // pairwise inside a step
                        const int k = 4 * s;
                        const double p01 = fma(A[i * K + k], B[k * 16 + j], A[i * K + k + 1] * B[(k + 1) * 16 + j]);
                        const double p23 = fma(A[i * K + k + 2], B[(k + 2) * 16 + j], A[i * K + k + 3] * B[(k + 3) * 16 + j]);
                        pw = pw + (p01 + p23);
                    }
                    const double d = D[i * 16 + j];
                    n_total++;
                    n_asc += same(d, asc); n_desc += same(d, desc); n_unfused += same(d, unf); n_pair += same(d, pw);
                    worst = fmax(worst, fabs(d - asc) / fmax(fabs(asc), 1e-300));
                }
        }
        printf("order test, %d k-step(s): %d results; equal to k-ascending fma chain %d, descending-in-step %d, unfused %d, pairwise %d; "
               "worst rel. distance from the ascending chain %.3g\n", ksteps, n_total, n_asc, n_desc, n_unfused, n_pair, worst);
    }

    // ---------------- (2) rates
    double* dout; long long* dcyc;
    CHECK(hipMalloc(&dout, 8ull * 512 * 256 * 8)); CHECK(hipMalloc(&dcyc, 64));
    const int iters = 2000;
    struct { const char* what; int mode; int threads; int blocks; } runs[] = {
        { "MFMA, 8 independent accumulators, 1 wave/SIMD, one CU", 0, 256, 1 },
        { "MFMA, dependent chain,            1 wave/SIMD, one CU", 1, 256, 1 },
        { "v_fma_f64 (8 chains),             1 wave/SIMD, one CU", 2, 256, 1 },
        { "MFMA, 8 accumulators,             2 waves/SIMD, one CU", 0, 512, 1 },
        { "v_fma_f64,                        2 waves/SIMD, one CU", 2, 512, 1 },
        { "waves 0-3 MFMA + waves 4-7 v_fma_f64 (one of each per SIMD), one CU", 3, 512, 1 },
        { "MFMA, 8 accumulators, 1 wave/SIMD, every CU", 0, 256, 256 },
        { "waves 0-3 MFMA + waves 4-7 v_fma_f64, every CU", 3, 512, 256 },
    };
    for (auto& r : runs) {
        long long cyc[8] = {};
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_rate<8>, dim3(r.blocks), dim3(r.threads), 0, 0, dout, 10, r.mode, dcyc);   // warm
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rate<8>, dim3(r.blocks), dim3(r.threads), 0, 0, dout, iters, r.mode, dcyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(cyc, dcyc, 64, hipMemcpyDeviceToHost));
        const double n_ops = (double)iters * 8;     // MFMAs per MFMA wave; VALU waves run 2 * 8 * n_ops v_fma_f64
        printf("%-75s %.3f ms; wave 0: %lld clocks = %.1f per MFMA (or %.2f per v_fma_f64)", r.what, ms, cyc[0], cyc[0] / n_ops, cyc[0] / (n_ops * 16));
        if (r.threads == 512) printf("; wave 4: %lld clocks = %.1f per MFMA-equivalent (%.2f per v_fma_f64)", cyc[4], cyc[4] / n_ops, cyc[4] / (n_ops * 16));
        printf("\n");
    }
    printf("(clocks are s_memtime / readcyclecounter ticks of a constant 100 MHz counter when the figure is far below 1: compare the rows)\n");
    return 0;
}
