// The collective behind the C ABI (include/obtg.h, obtg_comm_*): one process per GPU, RCCL over xGMI.
//
// north_star asks for "an RCCL all-gather of inter-vehicle separation minima over xGMI only when the swarm is
// partitioned" (the reference itself is single-process: optimization.py:311-346 loops over every pair).
// Until round 5 the only collective lived in the Python layer (torch.distributed, distributed.py); a caller that binds the
// library from C or through ctypes without torch had no multi-GPU path.  RCCL is bound at RUN time (dlopen of
// librccl.so.1 -- OBTG_RCCL_LIB names another file): the library keeps depending on the HIP runtime alone, and a process
// that already holds an RCCL (PyTorch bundles one) shares it when the soname matches.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <new>

#include "obtg_internal.h"

namespace obtg {
namespace {

// the few RCCL declarations used, as rccl.h has them (ncclUniqueId is 128 opaque bytes, ncclUint8 = 1)
struct NcclUniqueId { char internal[128]; };
using NcclComm = void*;
using FnGetUniqueId = int (*)(NcclUniqueId*);
using FnCommInitRank = int (*)(NcclComm*, int, NcclUniqueId, int);
using FnCommDestroy = int (*)(NcclComm);
using FnAllGather = int (*)(const void*, void*, size_t, int, NcclComm, hipStream_t);
using FnGetErrorString = const char* (*)(int);

struct Rccl {
    void* handle = nullptr;
    FnGetUniqueId get_unique_id = nullptr;
    FnCommInitRank comm_init_rank = nullptr;
    FnCommDestroy comm_destroy = nullptr;
    FnAllGather all_gather = nullptr;
    FnGetErrorString error_string = nullptr;
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl r = [] {
        Rccl q;
        const char* names[] = { getenv("OBTG_RCCL_LIB"), "librccl.so.1", "librccl.so" };
        for (const char* n : names) {
            if (!n || !n[0]) continue;
            q.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (q.handle) break;
        }
        if (!q.handle) return q;
        q.get_unique_id = reinterpret_cast<FnGetUniqueId>(dlsym(q.handle, "ncclGetUniqueId"));
        q.comm_init_rank = reinterpret_cast<FnCommInitRank>(dlsym(q.handle, "ncclCommInitRank"));
        q.comm_destroy = reinterpret_cast<FnCommDestroy>(dlsym(q.handle, "ncclCommDestroy"));
        q.all_gather = reinterpret_cast<FnAllGather>(dlsym(q.handle, "ncclAllGather"));
        q.error_string = reinterpret_cast<FnGetErrorString>(dlsym(q.handle, "ncclGetErrorString"));
        q.ok = q.get_unique_id && q.comm_init_rank && q.comm_destroy && q.all_gather;
        return q;
    }();
    return r;
}

// rank blocks [G][B * cmax] -> rows [B][P]: rank r's block holds its `cnt_r` pairs of every row back to back
__global__ __launch_bounds__(256) void k_unpack_pair_blocks(const double* __restrict__ recv, double* __restrict__ out, int B, int P,
                                                            int G, int base, int extra, int cmax)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * P) return;
    const int b = (int)(t / P), q = (int)(t - (long)b * P);
    // contiguous balanced blocks: the first `extra` ranks hold base + 1 pairs (distributed.partition)
    const int split = extra * (base + 1);
    int r, j, cnt;
    if (q < split) { r = q / (base + 1); j = q - r * (base + 1); cnt = base + 1; }
    else { r = extra + (base ? (q - split) / base : 0); j = q - split - (r - extra) * base; cnt = base; }
    out[t] = recv[(size_t)r * B * cmax + (size_t)b * cnt + j];
}

}  // namespace
}  // namespace obtg

using namespace obtg;

struct obtg_comm {
    NcclComm comm = nullptr;
    int n_ranks = 1, rank = 0, device = 0;
    DevBuf send, recv;
    std::string last_error;
};

// (failures before a communicator exists -- ncclGetUniqueId, ncclCommInitRank -- keep their text per thread:
// obtg_comm_last_error(NULL) returns it)
static thread_local std::string t_last_error;

static int comm_fail(obtg_comm* m, int nccl_rc, const char* where)
{
    const char* txt = rccl().error_string ? rccl().error_string(nccl_rc) : "RCCL error";
    std::string& dst = m ? m->last_error : t_last_error;
    dst = std::string(where) + ": " + (txt ? txt : "RCCL error");
    return OBTG_ERR_DEVICE;
}

extern "C" {

int obtg_comm_unique_id(unsigned char* id /*[128]*/)
{
    if (!id) return OBTG_ERR_ARG;
    if (!rccl().ok) return OBTG_ERR_UNSUPPORTED;
    NcclUniqueId u;
    if (int rc = rccl().get_unique_id(&u)) return comm_fail(nullptr, rc, "ncclGetUniqueId");
    std::memcpy(id, u.internal, sizeof(u.internal));
    return OBTG_OK;
}

int obtg_comm_create(obtg_comm** out, int n_ranks, int rank, const unsigned char* id /*[128]*/, int device)
{
    if (!out) return OBTG_ERR_ARG;
    *out = nullptr;
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return OBTG_ERR_ARG;
    if (!rccl().ok) return OBTG_ERR_UNSUPPORTED;
    if (device < 0 || device >= obtg_device_count()) return OBTG_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return OBTG_ERR_NO_DEVICE; }
    obtg_comm* m = new (std::nothrow) obtg_comm();
    if (!m) return OBTG_ERR_OOM;
    m->n_ranks = n_ranks; m->rank = rank; m->device = device;
    NcclUniqueId u;
    std::memcpy(u.internal, id, sizeof(u.internal));
    if (int rc = rccl().comm_init_rank(&m->comm, n_ranks, u, rank)) { delete m; return comm_fail(nullptr, rc, "ncclCommInitRank"); }
    *out = m;
    return OBTG_OK;
}

void obtg_comm_destroy(obtg_comm* m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->comm && rccl().ok) (void)rccl().comm_destroy(m->comm);
    m->send.release();
    m->recv.release();
    delete m;
}

int obtg_comm_size(const obtg_comm* m) { return m ? m->n_ranks : 0; }
int obtg_comm_rank(const obtg_comm* m) { return m ? m->rank : -1; }
const char* obtg_comm_last_error(const obtg_comm* m) { return m ? m->last_error.c_str() : t_last_error.c_str(); }

int obtg_comm_all_gather_dev(obtg_comm* m, obtg_ctx* c, const void* d_send, void* d_recv, size_t bytes_per_rank)
{
    if (!m || !c || !d_send || !d_recv) return OBTG_ERR_ARG;
    if (bytes_per_rank == 0) return OBTG_OK;
    if (c->device != m->device) return OBTG_ERR_ARG;
    (void)hipSetDevice(m->device);
    if (int rc = rccl().all_gather(d_send, d_recv, bytes_per_rank, 1 /* ncclUint8 */, m->comm, c->stream))
        return comm_fail(m, rc, "ncclAllGather");
    return OBTG_OK;
}

}  // extern "C"

// the pair partition of the ranks (distributed.partition): block of rank r
static void pair_block(int P, int G, int r, int* begin, int* count)
{
    const int base = P / G, extra = P % G;
    *count = base + (r < extra ? 1 : 0);
    *begin = r * base + (r < extra ? r : extra);
}

extern "C" {

int obtg_pair_block(const obtg_ctx* c, int n_ranks, int rank, int* begin, int* count)
{
    if (!c || !begin || !count || n_ranks < 1 || rank < 0 || rank >= n_ranks) return OBTG_ERR_ARG;
    pair_block(c->n_pairs, n_ranks, rank, begin, count);
    return OBTG_OK;
}

int obtg_unpack_pair_blocks_dev(obtg_ctx* c, const double* d_blocks, int B, int n_ranks, double* d_rows)
{
    if (!c || !d_blocks || !d_rows || B < 0 || n_ranks < 1) return OBTG_ERR_ARG;
    const int P = c->n_pairs;
    const long n = (long)B * P;
    if (n == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    hipLaunchKernelGGL(k_unpack_pair_blocks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_blocks, d_rows, B, P,
                       n_ranks, P / n_ranks, P % n_ranks, P / n_ranks + (P % n_ranks ? 1 : 0));
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

}  // extern "C"

namespace obtg {

int comm_gather_pair_minima(::obtg_comm* m, obtg_ctx* c, const double* dY, int B, double max_sep, double* d_min_all)
{
    if (c->device != m->device) return OBTG_ERR_ARG;      // (the buffers live on the context's device, the communicator is bound to its own)
    const int P = c->n_pairs, G = m->n_ranks;
    int b0, cnt;
    pair_block(P, G, m->rank, &b0, &cnt);
    const int cmax = P / G + (P % G ? 1 : 0);
    int rc;
    const size_t per_rank = sizeof(double) * (size_t)B * cmax;
    if ((rc = m->send.reserve(per_rank))) return rc;
    if ((rc = m->recv.reserve(per_rank * G))) return rc;
    if (cnt && (rc = launch_temporal_sep(c, dY, B, max_sep, b0, cnt, true, m->send.as<double>()))) return rc;
    if (int nrc = rccl().all_gather(m->send.p, m->recv.p, per_rank, 1 /* ncclUint8 */, m->comm, c->stream))
        return comm_fail(m, nrc, "ncclAllGather");
    return obtg_unpack_pair_blocks_dev(c, m->recv.as<double>(), B, G, d_min_all);
}

}  // namespace obtg
