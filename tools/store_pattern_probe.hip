// What write rate does a given STORE PATTERN reach on MI355X?  The elevated separation rows (DEG_ELEV = 100: 2.26 GB per
// launch) leave as 15.5 KB runs, one per workgroup and 16-row tile; with the arithmetic switched off the kernel still takes
// 0.426 ms = 5.3 TB/s against a fill rate of 6.8 TB/s on the same box.  This probe writes the same volume with the pattern
// as a parameter: who owns which burst, how long the bursts are, non-temporal or write-back stores, how many workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/store_pattern_probe.hip -o tools/_bin/store_pattern_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2_t __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// total = n_bursts * burst_bytes.  mode 0: workgroup g owns bursts [g * per, (g + 1) * per) (a contiguous chunk, walked
// in order); mode 1: workgroup g owns bursts g, g + G, g + 2 G, ... (the workgroups running together write neighbours)
template <bool NT>
__global__ __launch_bounds__(256) void k_store(d2_t* out, long n_bursts, int burst_pieces /*16-byte pieces per burst*/, int mode, int spin)
{
    const long G = gridDim.x, g = blockIdx.x;
    const long per = (n_bursts + G - 1) / G;
    d2_t v; v.x = (double)threadIdx.x; v.y = (double)g;
    for (long k = 0; k < per; ++k) {
        const long bi = mode == 0 ? g * per + k : k * G + g;
        if (bi >= n_bursts) break;
        d2_t* o = out + bi * burst_pieces;
        for (int m = threadIdx.x; m < burst_pieces; m += 256) {
            if (NT) __builtin_nontemporal_store(v, o + m); else o[m] = v;
        }
        for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(8);     // ~64 clocks each: the arithmetic between two bursts
        __syncthreads();
    }
}

int main()
{
    const size_t total = 2263053056ull;          // the C5 separation block
    d2_t* d;
    CHECK(hipMalloc(&d, total + (1 << 20)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-10s %-6s %-12s %-8s %-6s %10s %10s\n", "mode", "nt", "burst", "wgs", "spin", "ms", "TB/s");
    const bool quick = getenv("PROBE_ONE_BURST_PER_WG") != nullptr;      // only: every workgroup writes ONE burst and leaves
    for (int burst_bytes : { 15488, 61952, 4096, 247808, 1024 })
        for (int mode : { 0, 1 })
            for (int nt : { 0, 1 })
                for (int wgs : { 768, 2048, 2306, 9224, -1 })
                    for (int spin : { 0, 8 }) {
                        if (burst_bytes != 15488 && (spin || wgs == 2048)) continue;
                        if (quick != (wgs < 0)) continue;
                        if (wgs < 0 && (mode == 0 || spin)) continue;
                        if (burst_bytes == 1024 && wgs > 0) continue;
                        const long n_bursts = (long)(total / burst_bytes);
                        if (wgs < 0) wgs = (int)n_bursts;
                        const int pieces = burst_bytes / 16;
                        float best = 1e9f;
                        for (int rep = 0; rep < 4; ++rep) {
                            CHECK(hipEventRecord(e0));
                            if (nt) hipLaunchKernelGGL(k_store<true>, dim3(wgs), dim3(256), 0, 0, d, n_bursts, pieces, mode, spin);
                            else hipLaunchKernelGGL(k_store<false>, dim3(wgs), dim3(256), 0, 0, d, n_bursts, pieces, mode, spin);
                            CHECK(hipEventRecord(e1));
                            CHECK(hipDeviceSynchronize());
                            float ms = 0;
                            CHECK(hipEventElapsedTime(&ms, e0, e1));
                            if (rep && ms < best) best = ms;
                        }
                        printf("%-10s %-6d %-12d %-8d %-6d %10.4f %10.3f\n", mode ? "interleave" : "chunk", nt, burst_bytes, wgs, spin, best,
                               (double)n_bursts * burst_bytes / (best * 1e-3) / 1e12);
                    }
    // the reference point: hipMemsetAsync of the same bytes
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        CHECK(hipMemsetAsync(d, 0, total, 0));
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("hipMemsetAsync of the same bytes: %.4f ms = %.3f TB/s\n", ms, total / (ms * 1e-3) / 1e12);
    }
    return 0;
}
