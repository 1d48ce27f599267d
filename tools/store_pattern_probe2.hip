// Second look at the write rate of store patterns (tools/store_pattern_probe.hip found 7.0 TB/s for "one aligned 4 KB block per
// short-lived workgroup" against 5.4-5.8 TB/s for everything shaped like the elevated separation rows).  Which property is it:
// the size of what one workgroup writes, its alignment, the number of stores a wave has in flight, the barrier, or the
// workgroup living on?  Same volume as the C5 separation block.
//   hipcc --offload-arch=gfx950 -O3 tools/store_pattern_probe2.hip -o tools/_bin/store_pattern_probe2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef double d2_t __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// One burst per workgroup: workgroup g writes pieces [g * pieces, (g + 1) * pieces) (16-byte pieces) from byte `skew` on.
// order 0: piece m by thread m % T in pass m / T (a wave instruction = 1 KB contiguous);  order 1: each thread owns
// pieces / T ADJACENT pieces (a wave instruction = 64 runs of 16 bytes, pieces / T * 16 bytes apart)
__global__ __launch_bounds__(256) void k_one(d2_t* out, int pieces, int order, long scatter_mul = 0)
{
    d2_t v; v.x = (double)threadIdx.x; v.y = (double)blockIdx.x;
    // scatter_mul: burst = (id * mul) mod grid (mul coprime to the grid: a permutation that tears the address order apart)
    const size_t bi = scatter_mul ? (size_t)(((unsigned long long)blockIdx.x * (unsigned long long)scatter_mul) % gridDim.x) : (size_t)blockIdx.x;
    d2_t* o = out + bi * pieces;
    const int T = blockDim.x;
    if (order == 0) {
        for (int m = threadIdx.x; m < pieces; m += T) o[m] = v;
    } else {
        const int per = pieces / T;
        for (int i = 0; i < per; ++i) o[threadIdx.x * per + i] = v;
    }
}

// Persistent: G workgroups, workgroup g writes bursts g, g + G, ... (what runs together writes neighbours); barrier 0/1
__global__ __launch_bounds__(256) void k_loop(d2_t* out, long n_bursts, int pieces, int barrier, int spin = 0)
{
    d2_t v; v.x = (double)threadIdx.x; v.y = (double)blockIdx.x;
    for (long bi = blockIdx.x; bi < n_bursts; bi += gridDim.x) {
        d2_t* o = out + bi * pieces;
        for (int m = threadIdx.x; m < pieces; m += blockDim.x) o[m] = v;
        for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(8);      // ~64 clocks each: the arithmetic between two bursts
        if (barrier) __syncthreads();
    }
}

// Is it WHICH XCD writes a 4 KB page?  Workgroup ids go round the 8 XCDs; page = (id with its low three bits advanced by
// `shift`): shift 0 = page p written by XCD p % 8.  One page per workgroup (grid = pages) or persistent (workgroup g
// takes pages of its own residue class, 8 * (g / 8 + k * G / 8) + (g + shift) % 8).
__global__ __launch_bounds__(256) void k_xcd(d2_t* out, long n_pages, int shift, int persistent)
{
    d2_t v; v.x = (double)threadIdx.x; v.y = (double)blockIdx.x;
    const long g = blockIdx.x, res = (g + shift) & 7;
    if (!persistent) {
        const long pg = (g & ~7l) | res;
        if (pg < n_pages) out[pg * 256 + threadIdx.x] = v;
        return;
    }
    const long G8 = gridDim.x >> 3;
    for (long q = g >> 3; q * 8 + res < n_pages; q += G8) out[(q * 8 + res) * 256 + threadIdx.x] = v;
}

// Persistent, bursts handed out in address order by a counter (one atomic per burst and workgroup)
__global__ __launch_bounds__(256) void k_queue(d2_t* out, long n_bursts, int pieces, unsigned long long* counter)
{
    __shared__ long s_bi;
    d2_t v; v.x = (double)threadIdx.x; v.y = (double)blockIdx.x;
    for (;;) {
        if (threadIdx.x == 0) s_bi = (long)atomicAdd(counter, 1ull);
        __syncthreads();
        const long bi = s_bi;
        __syncthreads();
        if (bi >= n_bursts) break;
        d2_t* o = out + bi * pieces;
        for (int m = threadIdx.x; m < pieces; m += blockDim.x) o[m] = v;
    }
}

int main()
{
    const size_t total = 2263053056ull;          // the C5 separation block
    char* base;
    CHECK(hipMalloc(&base, total + (4 << 20)));
    unsigned long long* counter;
    CHECK(hipMalloc(&counter, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));

    printf("one burst per workgroup\n%-8s %-8s %-6s %-6s %10s %10s %10s\n", "burst", "threads", "skew", "order", "wgs", "ms", "TB/s");
    struct One { int burst, threads, skew, order; };
    const One ones[] = {
        { 4096, 256, 0, 0 }, { 4096, 256, 2048, 0 }, { 4096, 256, 128, 0 }, { 4096, 256, 16, 0 }, { 4096, 128, 0, 0 }, { 4096, 64, 0, 0 },
        { 2048, 128, 0, 0 }, { 8192, 256, 0, 0 }, { 8192, 256, 0, 1 }, { 12288, 256, 0, 0 }, { 16384, 256, 0, 0 }, { 16384, 256, 0, 1 },
        { 16384, 256, 2048, 0 }, { 15488, 256, 0, 0 }, { 15360, 256, 0, 0 }, { 32768, 256, 0, 0 }, { 65536, 256, 0, 0 }, { 65536, 256, 0, 1 },
        { 3872, 256, 0, 0 }, { 7744, 256, 0, 0 },
    };
    for (const One& c : ones) {
        const long n = (long)(total / c.burst);
        const int pieces = c.burst / 16;
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_one, dim3((unsigned)n), dim3(c.threads), 0, 0, (d2_t*)(base + c.skew), pieces, c.order);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("%-8d %-8d %-6d %-6d %10ld %10.4f %10.3f\n", c.burst, c.threads, c.skew, c.order, n, best, (double)n * c.burst / (best * 1e-3) / 1e12);
    }

    printf("persistent workgroups\n%-8s %-8s %-10s %10s %10s\n", "burst", "wgs", "kind", "ms", "TB/s");
    for (int burst : { 4096, 16384, 15488 })
        for (int wgs : { 768, 1024, 2048, 4096 })
            for (int kind : { 0, 1 }) {        // (kind 2, the counter: 0.34-1.3 TB/s -- the one atomic address is the bottleneck)
                const long n = (long)(total / burst);
                const int pieces = burst / 16;
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    if (kind == 2) CHECK(hipMemsetAsync(counter, 0, 8, 0));
                    CHECK(hipEventRecord(e0));
                    if (kind == 2) hipLaunchKernelGGL(k_queue, dim3(wgs), dim3(256), 0, 0, (d2_t*)base, n, pieces, counter);
                    else hipLaunchKernelGGL(k_loop, dim3(wgs), dim3(256), 0, 0, (d2_t*)base, n, pieces, kind);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipDeviceSynchronize());
                    float ms = 0;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep && ms < best) best = ms;
                }
                printf("%-8d %-8d %-10s %10.4f %10.3f\n", burst, wgs, kind == 0 ? "stride" : kind == 1 ? "stride+bar" : "queue", best,
                       (double)n * burst / (best * 1e-3) / 1e12);
            }
    printf("one burst per workgroup, bursts in address order or scattered (id * 40503 mod grid; 8 MB apart: id * 2053 at 4 KB)\n%-8s %-10s %10s %10s\n", "burst", "mul", "ms", "TB/s");
    for (int burst : { 4096, 16384, 15488 })
        for (long mul : { 0l, 40503l, 2053l, 9l }) {
            long n = (long)(total / burst);
            while (mul && (n % 3 == 0 || n % 23 == 0 || n % 587 == 0 || n % 2053 == 0 || n % 2 == 0)) --n;      // coprime to all the multipliers
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_one, dim3((unsigned)n), dim3(256), 0, 0, (d2_t*)base, burst / 16, 0, mul);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("%-8d %-10ld %10.4f %10.3f\n", burst, mul, best, (double)n * burst / (best * 1e-3) / 1e12);
        }
    printf("how many persistent writers, how much time between two bursts (15488-byte bursts, stride order, barrier)\n%-8s %-6s %10s %10s\n", "wgs", "spin", "ms", "TB/s");
    for (int wgs : { 256, 384, 512, 640, 768, 1024, 1280 })
        for (int spin : { 0, 4, 8, 16, 32 }) {
            const int burst = 15488;
            const long n = (long)(total / burst);
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_loop, dim3(wgs), dim3(256), 0, 0, (d2_t*)base, n, burst / 16, 1, spin);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("%-8d %-6d %10.4f %10.3f\n", wgs, spin, best, (double)n * burst / (best * 1e-3) / 1e12);
        }
    printf("which XCD writes which 4 KB page\n%-12s %-8s %-6s %10s %10s\n", "kind", "wgs", "shift", "ms", "TB/s");
    {
        const long n = (long)(total / 4096) & ~7l;
        for (int wgs : { 0, 768, 1024, 2048 })
            for (int shift = 0; shift < 8; ++shift) {
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    CHECK(hipEventRecord(e0));
                    hipLaunchKernelGGL(k_xcd, dim3(wgs ? wgs : (unsigned)n), dim3(256), 0, 0, (d2_t*)base, n, shift, wgs ? 1 : 0);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipDeviceSynchronize());
                    float ms = 0;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep && ms < best) best = ms;
                }
                printf("%-12s %-8ld %-6d %10.4f %10.3f\n", wgs ? "persistent" : "one per wg", wgs ? (long)wgs : n, shift, best, (double)n * 4096 / (best * 1e-3) / 1e12);
            }
    }
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        CHECK(hipMemsetAsync(base, 0, total, 0));
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("hipMemsetAsync of the same bytes: %.4f ms = %.3f TB/s\n", ms, total / (ms * 1e-3) / 1e12);
    }
    return 0;
}
