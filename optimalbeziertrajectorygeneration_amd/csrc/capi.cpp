// C ABI of libobtg_hip.so (include/obtg.h): context, tables, host-buffer entry points.
// Compiled with hipcc (host code + HIP runtime API); the kernels live in
// bern_kernels.hip and gjk_kernels.hip.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#include "obtg_internal.h"

namespace obtg {

int DevBuf::reserve(size_t bytes, bool zero_copy)
{
    if (io && zero_copy && bytes <= kZeroCopyBytes && !host_failed) {
        if (!host) {
            if (hipHostMalloc(&host, kZeroCopyBytes, hipHostMallocMapped) != hipSuccess ||
                hipHostGetDevicePointer(&host_dev, host, 0) != hipSuccess) {
                (void)hipGetLastError();
                if (host) (void)hipHostFree(host);
                host = nullptr; host_dev = nullptr; host_failed = true;
            }
        }
        if (host) { p = host_dev; cap = kZeroCopyBytes; on_host = true; return OBTG_OK; }
    }
    on_host = false;
    if (bytes <= dev_cap && dev) { p = dev; cap = dev_cap; return OBTG_OK; }
    if (bytes == 0) bytes = 8;
    if (dev) { (void)hipFree(dev); dev = nullptr; dev_cap = 0; }
    p = nullptr; cap = 0;
    // grow geometrically to keep repeated host-entry calls from reallocating
    size_t want = bytes + bytes / 4;
    if (hipMalloc(&dev, want) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc(&dev, bytes) != hipSuccess) { dev = nullptr; return OBTG_ERR_OOM; }
        want = bytes;
    }
    dev_cap = want;
    p = dev; cap = dev_cap;
    return OBTG_OK;
}

void DevBuf::release()
{
    if (dev) (void)hipFree(dev);
    if (host) (void)hipHostFree(host);
    dev = nullptr; dev_cap = 0; host = nullptr; host_dev = nullptr;
    p = nullptr; cap = 0; on_host = false;
}

int set_error(obtg_ctx* c, hipError_t e, const char* where)
{
    if (c) {
        c->last_error = std::string(hipGetErrorString(e)) + " at " + where;
    }
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? OBTG_ERR_OOM : OBTG_ERR_DEVICE;
}

ScopedKernelTimer::ScopedKernelTimer(obtg_ctx* c_, int id_, bool ext_) : c(c_), id(id_), ext(ext_)
{
    if (!c->profiling || !((c->profile_mask >> id) & 1u)) return;
    if (c->profile_period > 1 && (c->profile_seen[id]++ % c->profile_period) != 0) return;
    auto take = [&](hipEvent_t& e) {
        if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); return true; }
        return hipEventCreate(&e) == hipSuccess;
    };
    if (!take(a) || !take(b)) { a = b = nullptr; return; }
    if (!ext) (void)hipEventRecord(a, c->stream);
}

ScopedKernelTimer::~ScopedKernelTimer()
{
    if (!a || !b) return;
    if (!ext) (void)hipEventRecord(b, c->stream);
    c->pending_events.push_back({ id, { a, b } });
}

void flush_pending_events(obtg_ctx* c)
{
    for (auto& pe : c->pending_events) {
        float ms = 0.f;
        if (hipEventSynchronize(pe.second.second) == hipSuccess &&
            hipEventElapsedTime(&ms, pe.second.first, pe.second.second) == hipSuccess) {
            c->stats[pe.first].ms += ms;
            c->stats[pe.first].launches += 1;
        }
        c->event_pool.push_back(pe.second.first);
        c->event_pool.push_back(pe.second.second);
    }
    c->pending_events.clear();
}

static int upload(obtg_ctx* c, DevBuf& buf, const void* src, size_t bytes)
{
    int rc = buf.reserve(bytes);
    if (rc) return rc;
    if (bytes) OBTG_HIP(c, hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    // the source is usually a temporary: complete the copy before returning
    OBTG_HIP(c, hipStreamSynchronize(c->stream));
    return OBTG_OK;
}

// plain (unfolded) equal-degree product weights w(k,j), layout [2n+1][n+1]
static std::vector<double> plain_product_weights(int n)
{
    int L = 2 * n + 1, nc = n + 1;
    std::vector<double> W((size_t)L * nc, 0.0);
    for (int k = 0; k < L; ++k) {
        double den = binom(2 * n, k);
        for (int j = (k - n > 0 ? k - n : 0); j <= (n < k ? n : k); ++j)
            W[(size_t)k * nc + j] = binom(n, j) * binom(n, k - j) / den;
    }
    // behind them, for the separable form of the speed / angular-rate arithmetic (bern_device.h ang_raw_*): C(n, .), 1 / C(2n, .)
    for (int j = 0; j < nc; ++j) W.push_back(binom(n, j));
    for (int k = 0; k < L; ++k) W.push_back(1.0 / binom(2 * n, k));
    return W;
}

int ensure_tables(obtg_ctx* c)
{
    if (c->tables_R == c->R) return OBTG_OK;
    (void)hipSetDevice(c->device);                 // (obtg_ctx_set_deg_elev comes here from whatever device the caller was on)
    const int n = c->deg, L = 2 * n + 1;
    if (c->tables_R < 0) {
        auto W2 = folded_product_weights(n, c->dim);
        int rc = upload(c, c->d_w2, W2.data(), W2.size() * sizeof(double));
        if (rc) return rc;
        if (c->dim == 2 && n <= 15) {
            auto a = folded_product_weights(n, 2);        // factor dim/2 = 1
            auto b = folded_product_weights(2 * n, 2);
            auto w = plain_product_weights(n);
            if ((rc = upload(c, c->d_ang_w2n, a.data(), a.size() * sizeof(double)))) return rc;
            if ((rc = upload(c, c->d_ang_w22n, b.data(), b.size() * sizeof(double)))) return rc;
            if ((rc = upload(c, c->d_ang_wn, w.data(), w.size() * sizeof(double)))) return rc;
        }
    }
    // the elevation tables belong to ONE R: a context that moves to R = 0 or beyond 512 must not keep the previous R's
    // (the launchers' `d_Tf.p == nullptr` guards mean "no table for this R")
    c->d_Tt.release();
    c->d_Td.release();
    c->d_Tf.release();
    if (c->R > 0 && c->R <= 512) {
        auto Tt = elev_conv_tables(L, c->R);
        int rc = upload(c, c->d_Tt, Tt.data(), Tt.size() * sizeof(double));
        if (rc) return rc;
        auto Td = elev_table_T_ld(L, c->R);
        if ((rc = upload(c, c->d_Td, Td.data(), Td.size() * sizeof(double)))) return rc;
        auto Tf = elev_table_frag(L, c->R);
        if ((rc = upload(c, c->d_Tf, Tf.data(), Tf.size() * sizeof(double)))) return rc;
    }
    c->d_ang_T4.release();
    c->d_ang_cv2.release();
    if (c->R > 0 && c->dim == 2 && n <= 15 && 4 * c->R <= 1000) {   // C(4R, .) and C(2n+R, .) finite in binary64
        auto cv4 = elev_conv_padded(4 * n + 1, 4 * c->R, 8, true, false);
        auto cv2 = elev_conv_padded(2 * n + 1, c->R, 8, false, true);
        int rc = upload(c, c->d_ang_T4, cv4.data(), cv4.size() * sizeof(double));
        if (rc) return rc;
        if ((rc = upload(c, c->d_ang_cv2, cv2.data(), cv2.size() * sizeof(double)))) return rc;
    }
    c->tables_R = c->R;
    return OBTG_OK;
}

int binrow_offset(obtg_ctx* c, int n)
{
    if (n < 0) return OBTG_ERR_ARG;
    if (n > 1029) return OBTG_ERR_UNSUPPORTED;   // C(n, n/2) must be finite in binary64
    if ((int)c->binrow_off.size() <= n) c->binrow_off.resize(n + 1, -1);
    if (c->binrow_off[n] >= 0) return c->binrow_off[n];
    auto row = binom_row(n);
    const int off = (int)c->h_binrows.size();
    c->h_binrows.insert(c->h_binrows.end(), row.begin(), row.end());
    // re-upload the whole (small) table; in-flight kernels keep reading the old allocation's
    // contents only if it is not freed, so drain the stream first
    if (hipStreamSynchronize(c->stream) != hipSuccess) return OBTG_ERR_DEVICE;
    int rc = upload(c, c->d_binrows, c->h_binrows.data(), c->h_binrows.size() * sizeof(double));
    if (rc) return rc;
    c->binrow_off[n] = off;
    return off;
}

static bool check_ctx(const obtg_ctx* c) { return c != nullptr; }

}  // namespace obtg

using namespace obtg;

extern "C" {

const char* obtg_strerror(int code)
{
    switch (code) {
        case OBTG_OK: return "ok";
        case OBTG_ERR_ARG: return "invalid argument";
        case OBTG_ERR_DEVICE: return "HIP runtime error";
        case OBTG_ERR_NO_DEVICE: return "no usable gfx950 device";
        case OBTG_ERR_OOM: return "out of memory";
        case OBTG_ERR_UNSUPPORTED: return "degree / size not supported by the kernels";
        default: return "unknown error";
    }
}

const char* obtg_last_error(const obtg_ctx* c) { return c ? c->last_error.c_str() : ""; }

int obtg_abi_version(void) { return OBTG_ABI_VERSION; }

int obtg_fast_kernels(int dim, int deg)
{
    const int nc = deg + 1;
    int mask = 0;
    if ((dim == 2 || dim == 3) && nc_in_sep(nc)) mask |= 1;
    if (nc_in_dyn(nc)) mask |= 2;
    if (dim == 2 && nc_in_elev(nc)) mask |= 4;
    return mask;
}

int obtg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

const char* obtg_abi_symbols(void)
{
    static const char syms[] =
        "obtg_strerror\0obtg_last_error\0obtg_abi_version\0obtg_source_hash\0obtg_libm_pow_matches\0obtg_fast_kernels\0obtg_device_count\0obtg_abi_symbols\0obtg_host_alloc\0obtg_host_free\0"
        "obtg_ctx_create\0obtg_ctx_destroy\0obtg_ctx_set_stream\0obtg_ctx_use_own_stream\0obtg_ctx_set_deg_elev\0obtg_ctx_set_ang_rate_order\0obtg_ctx_ang_rate_order_in_effect\0obtg_ctx_set_second_speed_bound\0obtg_sync\0"
        "obtg_len_temporal_sep\0obtg_len_speed\0obtg_len_ang_rate\0obtg_num_pairs\0"
        "obtg_temporal_sep\0obtg_speed\0obtg_ang_rate\0obtg_temporal_sep_min\0obtg_temporal_sep_min_range\0obtg_temporal_sep_active\0obtg_temporal_sep_active_dev\0obtg_temporal_sep_min_gather_dev\0obtg_temporal_sep_fd_min_rows_dev\0"
        "obtg_comm_unique_id\0obtg_comm_create\0obtg_comm_destroy\0obtg_comm_size\0obtg_comm_rank\0obtg_comm_last_error\0obtg_comm_all_gather_dev\0obtg_pair_block\0obtg_unpack_pair_blocks_dev\0"
        "obtg_temporal_sep_fd\0obtg_temporal_sep_fd_dev\0obtg_one_vs_many_min\0obtg_one_vs_many_min_dev\0"
        "obtg_temporal_sep_dev\0obtg_temporal_sep_min_dev\0obtg_speed_dev\0obtg_ang_rate_dev\0obtg_dynamics_dev\0"
        "obtg_fd_batch_dev\0obtg_fd_view_begin\0obtg_fd_view_begin_rows\0obtg_fd_view_end\0obtg_fd_forms_on_the_fly\0obtg_pair_sweep_fd_dev\0obtg_dynamics_fd_dev\0obtg_gjk_pairs\0obtg_ctx_set_polygons\0obtg_ctx_set_hull_pairs\0"
        "obtg_ctx_set_fd_dedup\0obtg_ctx_set_gjk_history\0obtg_pair_sweep_dev\0obtg_constraint_sweep_dev\0obtg_constraint_sweep_fd_structured_dev\0obtg_constraint_sweep_fd_structured_rows_dev\0obtg_gjk_swarm_dev\0obtg_gjk_swarm\0obtg_min_dist\0obtg_min_dist_robust\0obtg_min_dist2poly\0obtg_min_dist2poly_robust\0obtg_gjk_true_pairs\0"
        "obtg_bern_elev\0obtg_bern_diff\0obtg_bern_mul\0obtg_bern_normsq\0obtg_bern_split\0obtg_bern_eval\0"
        "obtg_euclidean_obj\0obtg_accel_obj\0obtg_jerk_obj\0"
        "obtg_set_profiling\0obtg_set_profile_period\0obtg_kernel_stats\0obtg_reset_kernel_stats\0obtg_kernel_name\0";
    return syms;
}

int obtg_ctx_create(obtg_ctx** out, int n_veh, int dim, int deg, int deg_elev, int n_point_obs,
                    const double* point_obs, int device)
{
    if (!out) return OBTG_ERR_ARG;
    *out = nullptr;
    if (n_veh < 1 || dim < 1 || dim > 3 || deg < 1 || deg_elev < 0 || n_point_obs < 0) return OBTG_ERR_ARG;
    if (n_point_obs > 0 && !point_obs) return OBTG_ERR_ARG;
    int ndev = obtg_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) return OBTG_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); return OBTG_ERR_NO_DEVICE; }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return OBTG_ERR_NO_DEVICE;  // code objects are gfx950 only
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return OBTG_ERR_NO_DEVICE; }
    obtg_ctx* c = new (std::nothrow) obtg_ctx();
    if (!c) return OBTG_ERR_OOM;
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->device = device; c->n_veh = n_veh; c->dim = dim; c->deg = deg; c->R = deg_elev;
    c->n_obs = n_point_obs; c->n_obj = n_veh + n_point_obs;
    c->n_pairs = c->n_obj * (c->n_obj - 1) / 2;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError(); delete c; return OBTG_ERR_DEVICE;
    }
    c->stream = c->own_stream;
    // OBTG_ZERO_COPY=0 keeps every staging buffer in device memory (the path for large batches)
    { const char* e = getenv("OBTG_ZERO_COPY"); const bool zc = !(e && e[0] == '0'); c->ws_in.io = c->ws_in2.io = c->ws_out.io = zc; }
    int rc = OBTG_OK;
    c->h_pairs.resize((size_t)2 * c->n_pairs);
    {
        size_t p = 0;
        for (int i = 0; i < c->n_obj - 1; ++i)
            for (int j = i + 1; j < c->n_obj; ++j) { c->h_pairs[p++] = i; c->h_pairs[p++] = j; }
    }
    rc = upload(c, c->d_pairs, c->h_pairs.data(), c->h_pairs.size() * sizeof(int));
    if (!rc) rc = upload(c, c->d_obs, point_obs, sizeof(double) * (size_t)n_point_obs * dim);
    if (!rc) rc = ensure_tables(c);
    if (rc) { obtg_ctx_destroy(c); return rc; }
    *out = c;
    return OBTG_OK;
}

void obtg_ctx_destroy(obtg_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    flush_pending_events(c);
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    DevBuf* bufs[] = { &c->d_pairs, &c->d_obs, &c->d_w2, &c->d_Tt, &c->d_Td, &c->d_Tf, &c->d_ang_dd, &c->d_ang_flags, &c->d_vp_off, &c->d_vp_idx, &c->d_ang_w2n, &c->d_ang_w22n, &c->d_ang_wn, &c->d_ang_T4, &c->d_ang_cv2,
                       &c->d_binrows, &c->d_tiles, &c->d_poly_pts, &c->d_poly_off, &c->d_hp_a, &c->d_hp_b, &c->d_tile_chunk_off, &c->d_tile_order, &c->d_tile_pslots,
                       &c->d_tile_cobj_off, &c->d_tile_cobjs, &c->d_tile_ij, &c->ws_in,
                       &c->ws_in2, &c->ws_out, &c->ws_fd };
    for (DevBuf* b : bufs) b->release();
    for (auto& b : c->ws_misc) b.release();
    for (auto& b : c->d_gjk_len) b.release();
    if (c->ring) {
        (void)hipHostFree(c->ring);
        for (hipEvent_t e : c->ring_ev) if (e) (void)hipEventDestroy(e);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int obtg_ctx_set_stream(obtg_ctx* c, void* hip_stream)
{
    if (!check_ctx(c)) return OBTG_ERR_ARG;
    OBTG_HIP(c, hipStreamSynchronize(c->stream));
    c->stream = static_cast<hipStream_t>(hip_stream);        // as given: NULL is the null stream
    return OBTG_OK;
}

int obtg_ctx_use_own_stream(obtg_ctx* c)
{
    if (!check_ctx(c)) return OBTG_ERR_ARG;
    OBTG_HIP(c, hipStreamSynchronize(c->stream));
    c->stream = c->own_stream;
    return OBTG_OK;
}

int obtg_ctx_set_deg_elev(obtg_ctx* c, int deg_elev)
{
    if (!check_ctx(c) || deg_elev < 0) return OBTG_ERR_ARG;
    if (deg_elev == c->R) return OBTG_OK;
    OBTG_HIP(c, hipStreamSynchronize(c->stream));
    c->R = deg_elev;
    return ensure_tables(c);
}

int obtg_host_alloc(size_t bytes, void** out)
{
    if (!out) return OBTG_ERR_ARG;
    *out = nullptr;
    if (bytes == 0) bytes = 8;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return OBTG_ERR_OOM; }
    return OBTG_OK;
}

int obtg_host_free(void* p)
{
    if (!p) return OBTG_OK;
    if (hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); return OBTG_ERR_ARG; }
    return OBTG_OK;
}

int obtg_ctx_set_ang_rate_order(obtg_ctx* c, int elevate_first)
{
    if (!check_ctx(c)) return OBTG_ERR_ARG;
    if (elevate_first < 0 || elevate_first > 2) return OBTG_ERR_ARG;
    c->ang_elevate_first = elevate_first == 1;
    c->ang_exact = elevate_first == 2;
    return OBTG_OK;
}

int obtg_ctx_ang_rate_order_in_effect(obtg_ctx* c)
{
    if (!check_ctx(c)) return OBTG_ERR_ARG;
    return ang_rate_order_in_effect(c);
}

int obtg_ctx_set_second_speed_bound(obtg_ctx* c, double bound, int is_max, double* d_out)
{
    if (!check_ctx(c)) return OBTG_ERR_ARG;
    c->speed2.bound = bound; c->speed2.is_max = is_max != 0; c->speed2.d_out = d_out;
    return OBTG_OK;
}

int obtg_sync(obtg_ctx* c)
{
    if (!check_ctx(c)) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    OBTG_HIP(c, hipStreamSynchronize(c->stream));
    flush_pending_events(c);
    return OBTG_OK;
}

int obtg_len_temporal_sep(const obtg_ctx* c) { return c ? c->n_pairs * (2 * c->deg + c->R + 1) : 0; }
int obtg_len_speed(const obtg_ctx* c) { return c ? c->n_veh * (2 * c->deg + c->R + 1) : 0; }
int obtg_len_ang_rate(const obtg_ctx* c) { return c ? c->n_veh * (4 * (c->deg + c->R) + 1) : 0; }
int obtg_num_pairs(const obtg_ctx* c) { return c ? c->n_pairs : 0; }

// ------------------------------------------------------------------ virtual finite-difference batch
// dY == NULL in a `_dev` sweep means "the batch of the open view" (obtg_fd_view_begin): the launcher gets the view's
// single row with c->fd set; kernels that form the rows while staging them use it, the others answer kNeedBatch and
// the batch is written to a context buffer -- once per view -- and handed over instead.
static int fd_materialise(obtg_ctx* c, const double* dY0, int n_fixed_cols, double h, int B, int row0);

extern "C++" {
// can_fd: can EVERY kernel of this call form the view's rows itself?  Decided before anything is launched, so that a
// call of several launches never runs its first ones twice (a launcher answering kNeedBatch after an earlier launch of
// the same call had gone out used to make the whole call run again on the materialised batch: right results, twice
// the work, two flips of the sweep's trip-count history).
// launch_dynamics is one launch on the specialised shapes; otherwise a speed launch and / or a generic angular-rate one
static bool dynamics_can_fd(const obtg_ctx* c, bool want_speed, bool want_ang)
{
    if (dynamics_fd_on_the_fly(c, want_ang)) return true;
    return !want_ang && want_speed && bernstein_fd_on_the_fly(c);
}
// launch_pair_sweep: one launch, or a gjkNew sweep (forms the rows itself unless it de-duplicates) + the separate
// temporal-separation kernel
static bool pair_sweep_can_fd(const obtg_ctx* c)
{
    return pair_sweep_is_one_launch(c) || (!c->fd_dedup && bernstein_fd_on_the_fly(c));
}

template <class Launch>
static int with_batch(obtg_ctx* c, const double* dY, int B, bool can_fd, Launch launch)
{
    if (dY) return launch(dY);
    if (!c->view.Y0 || B != c->view.B) return OBTG_ERR_ARG;
    int rc = kNeedBatch;
    if (!c->view.materialised && can_fd) {
        c->fd.Y0 = c->view.Y0; c->fd.h = c->view.h; c->fd.fixed = c->view.fixed; c->fd.row0 = c->view.row0;
        rc = launch(c->view.Y0);
        c->fd.Y0 = nullptr; c->fd.row0 = 0;
    }
    if (rc != kNeedBatch) return rc;
    if (!c->view.materialised) {
        if ((rc = fd_materialise(c, c->view.Y0, c->view.fixed, c->view.h, c->view.B, c->view.row0))) return rc;
        c->view.materialised = true;
    }
    return launch(c->ws_fd.as<double>());
}
}  // extern "C++"

static int fd_args_ok(const obtg_ctx* c, int n_fixed_cols, int row_begin, int B)
{
    const int rows = c->n_veh * c->dim, nc = c->deg + 1;
    if (n_fixed_cols < 0 || nc - 2 * n_fixed_cols <= 0 || row_begin < 0) return OBTG_ERR_ARG;
    if (B < 1 || row_begin + B > rows * (nc - 2 * n_fixed_cols) + 1) return OBTG_ERR_ARG;
    return OBTG_OK;
}

int obtg_fd_view_begin_rows(obtg_ctx* c, const double* dY0, int n_fixed_cols, double h, int row_begin, int B)
{
    if (!check_ctx(c) || !dY0) return OBTG_ERR_ARG;
    if (int rc = fd_args_ok(c, n_fixed_cols, row_begin, B)) return rc;
    c->view.Y0 = dY0; c->view.h = h; c->view.fixed = n_fixed_cols; c->view.B = B; c->view.row0 = row_begin; c->view.materialised = false;
    return OBTG_OK;
}

int obtg_fd_view_begin(obtg_ctx* c, const double* dY0, int n_fixed_cols, double h, int B)
{
    return obtg_fd_view_begin_rows(c, dY0, n_fixed_cols, h, 0, B);
}

int obtg_fd_view_end(obtg_ctx* c)
{
    if (!check_ctx(c)) return OBTG_ERR_ARG;
    c->view.Y0 = nullptr; c->view.B = 0; c->view.row0 = 0; c->view.materialised = false;
    return OBTG_OK;
}

// ------------------------------------------------------------------ device-pointer sweeps
int obtg_temporal_sep_dev(obtg_ctx* c, const double* dY, int B, double max_sep, int pair_begin,
                          int pair_count, double* d_out)
{
    if (!check_ctx(c) || !d_out || B < 0) return OBTG_ERR_ARG;
    if (pair_begin < 0 || pair_count < 0 || pair_begin + pair_count > c->n_pairs) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    return with_batch(c, dY, B, true, [&](const double* src) {
        return launch_temporal_sep(c, src, B, max_sep, pair_begin, pair_count, false, d_out); });
}

int obtg_temporal_sep_min_dev(obtg_ctx* c, const double* dY, int B, double max_sep, int pair_begin,
                              int pair_count, double* d_out)
{
    if (!check_ctx(c) || !d_out || B < 0) return OBTG_ERR_ARG;
    if (pair_begin < 0 || pair_count < 0 || pair_begin + pair_count > c->n_pairs) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    return with_batch(c, dY, B, true, [&](const double* src) {
        return launch_temporal_sep(c, src, B, max_sep, pair_begin, pair_count, true, d_out); });
}

int obtg_temporal_sep_active_dev(obtg_ctx* c, const double* dY, int B, double max_sep, int k, int pair_begin,
                                 int pair_count, double* d_out_val, int* d_out_idx)
{
    if (!check_ctx(c) || !d_out_val || B < 0 || k < 1 || k > 4 || k > 2 * c->deg + c->R + 1) return OBTG_ERR_ARG;
    if (pair_begin < 0 || pair_count < 0 || pair_begin + pair_count > c->n_pairs) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    return with_batch(c, dY, B, true, [&](const double* src) {
        return launch_temporal_sep(c, src, B, max_sep, pair_begin, pair_count, true, d_out_val, k, d_out_idx); });
}

int obtg_temporal_sep_fd_min_rows_dev(obtg_ctx* c, const double* dY0, int n_fixed_cols, double h, int row_begin, int n_rows,
                                      double max_sep, double* d_out)
{
    if (!check_ctx(c) || !dY0 || !d_out || n_rows < 0 || row_begin < 1) return OBTG_ERR_ARG;
    if (int rc = fd_args_ok(c, n_fixed_cols, row_begin, n_rows > 0 ? n_rows : 1)) return rc;
    if (n_rows == 0 || c->n_obj < 2) return OBTG_OK;
    (void)hipSetDevice(c->device);
    return launch_temporal_sep_fd(c, dY0, n_rows, nullptr, nullptr, nullptr, max_sep, d_out, 1, row_begin, n_fixed_cols, h);
}

int obtg_temporal_sep_min_gather_dev(obtg_ctx* c, obtg_comm* m, const double* dY, int B, double max_sep, double* d_min_all)
{
    if (!check_ctx(c) || !m || !d_min_all || B < 0) return OBTG_ERR_ARG;
    if (B == 0 || c->n_pairs == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    return with_batch(c, dY, B, true, [&](const double* src) { return comm_gather_pair_minima(m, c, src, B, max_sep, d_min_all); });
}

int obtg_speed_dev(obtg_ctx* c, const double* dY, const double* d_tf, int B, double bound, int is_max,
                   double* d_out)
{
    if (!check_ctx(c) || !d_tf || !d_out || B < 0) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    return with_batch(c, dY, B, true, [&](const double* src) { return launch_speed(c, src, d_tf, B, bound, is_max, d_out); });
}

int obtg_ang_rate_dev(obtg_ctx* c, const double* dY, const double* d_tf, int B, double max_rate, double* d_out)
{
    if (!check_ctx(c) || !d_tf || !d_out || B < 0) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    return with_batch(c, dY, B, true, [&](const double* src) { return launch_ang_rate(c, src, d_tf, B, max_rate, d_out); });
}

int obtg_dynamics_dev(obtg_ctx* c, const double* dY, const double* d_tf, int B, double speed_bound,
                      int speed_is_max, double max_rate, double* d_out_speed, double* d_out_ang)
{
    if (!check_ctx(c) || !d_tf || B < 0 || (!d_out_speed && !d_out_ang)) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    return with_batch(c, dY, B, dynamics_can_fd(c, d_out_speed != nullptr, d_out_ang != nullptr), [&](const double* src) {
        return launch_dynamics(c, src, d_tf, B, speed_bound, speed_is_max, max_rate, d_out_speed, d_out_ang); });
}

static int fd_materialise(obtg_ctx* c, const double* dY0, int n_fixed_cols, double h, int B, int row0)
{
    int rc = c->ws_fd.reserve(sizeof(double) * (size_t)B * c->n_veh * c->dim * (c->deg + 1));
    if (rc) return rc;
    return launch_fd_batch(c, dY0, n_fixed_cols, h, B, c->ws_fd.as<double>(), row0);
}

int obtg_fd_forms_on_the_fly(const obtg_ctx* c)
{
    if (!c) return 0;
    return (pair_sweep_is_one_launch(c) ? 1 : 0) | ((c->dim == 2 && dynamics_fd_on_the_fly(c, true)) ? 2 : 0);
}

// one-call forms: a view around a single sweep
int obtg_pair_sweep_fd_dev(obtg_ctx* c, const double* dY0, int n_fixed_cols, double h, int B, double max_sep,
                           double* d_out_sep, int max_iter, int md_cap, int* d_flag, double* d_p1, double* d_p2,
                           double* d_dist, int* d_nsup, int* d_status)
{
    int rc = obtg_fd_view_begin(c, dY0, n_fixed_cols, h, B);
    if (rc) return rc;
    rc = obtg_pair_sweep_dev(c, nullptr, B, max_sep, d_out_sep, max_iter, md_cap, d_flag, d_p1, d_p2, d_dist, d_nsup, d_status);
    (void)obtg_fd_view_end(c);
    return rc;
}

int obtg_dynamics_fd_dev(obtg_ctx* c, const double* dY0, int n_fixed_cols, double h, const double* d_tf, int B,
                         double speed_bound, int speed_is_max, double max_rate, double* d_out_speed, double* d_out_ang)
{
    int rc = obtg_fd_view_begin(c, dY0, n_fixed_cols, h, B);
    if (rc) return rc;
    rc = obtg_dynamics_dev(c, nullptr, d_tf, B, speed_bound, speed_is_max, max_rate, d_out_speed, d_out_ang);
    (void)obtg_fd_view_end(c);
    return rc;
}

int obtg_fd_batch_dev(obtg_ctx* c, const double* dY0, int n_fixed_cols, double h, int B, double* dY)
{
    if (!check_ctx(c) || !dY0 || !dY || B < 1 || n_fixed_cols < 0) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    return launch_fd_batch(c, dY0, n_fixed_cols, h, B, dY);
}

// ------------------------------------------------------------------ host-buffer sweeps
// Host <-> device copies of the host-buffer entry points.  A caller's NumPy array is pageable memory; handing it
// to hipMemcpyAsync makes the runtime bounce it through its own staging at ~10 GB/s (measured: 427 MB D2H in
// 41 ms).  Instead: (a) buffers the caller allocated with obtg_host_alloc (pinned) are DMA targets as they are;
// (b) pageable buffers go through the context's pinned ring in chunks, the DMA of chunk i+1 running while the host
// copies chunk i out of (into) the ring.
constexpr size_t kRingChunk = 4u << 20;     // bytes per ring slot
constexpr int kRingSlots = 4;
constexpr size_t kRingMin = 256u << 10;     // smaller copies: one pageable hipMemcpyAsync is as fast

static bool is_pinned(const void* p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

static int ensure_ring(obtg_ctx* c)
{
    if (c->ring) return OBTG_OK;
    if (hipHostMalloc(&c->ring, kRingChunk * kRingSlots, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError(); c->ring = nullptr; return OBTG_ERR_OOM;
    }
    for (int i = 0; i < kRingSlots; ++i)
        if (hipEventCreateWithFlags(&c->ring_ev[i], hipEventDisableTiming) != hipSuccess) {
            // partial failure: leave no half-built ring behind (the next call would take it for a complete one)
            (void)hipGetLastError();
            for (int j = 0; j < i; ++j) { (void)hipEventDestroy(c->ring_ev[j]); c->ring_ev[j] = nullptr; }
            c->ring_ev[i] = nullptr;
            (void)hipHostFree(c->ring);
            c->ring = nullptr;
            return OBTG_ERR_DEVICE;
        }
    return OBTG_OK;
}

// wait until the DMA that last used ring slot `sl` is done (the CPU is about to touch the slot)
static int ring_slot_ready(obtg_ctx* c, int sl)
{
    if (c->ring_pending[sl]) {
        OBTG_HIP(c, hipEventSynchronize(c->ring_ev[sl]));
        c->ring_pending[sl] = false;
    }
    return OBTG_OK;
}

// zero_copy: inputs of up to kZeroCopyIn bytes may stay in mapped host memory (the kernel fetches them across PCIe:
// a one-row call of 64 vehicles, 11 KB read by several workgroups, is already slower that way than one DMA transfer)
constexpr size_t kZeroCopyIn = 8u << 10;
static int h2d(obtg_ctx* c, DevBuf& b, const void* src, size_t bytes, bool zero_copy = false)
{
    int rc = b.reserve(bytes, zero_copy && bytes <= kZeroCopyIn);
    if (rc) return rc;
    if (b.on_host) {        // mapped host block: the device is idle between host entry points, nothing can be reading it
        std::memcpy(b.host, src, bytes);
        return OBTG_OK;
    }
    if (bytes < kRingMin || is_pinned(src) || ensure_ring(c) != OBTG_OK) {
        OBTG_HIP(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
        return OBTG_OK;
    }
    char* ring = static_cast<char*>(c->ring);
    for (size_t off = 0; off < bytes; off += kRingChunk) {
        const int sl = c->ring_next;
        c->ring_next = (c->ring_next + 1) % kRingSlots;
        const size_t nb = std::min(kRingChunk, bytes - off);
        if ((rc = ring_slot_ready(c, sl))) return rc;
        std::memcpy(ring + sl * kRingChunk, static_cast<const char*>(src) + off, nb);
        OBTG_HIP(c, hipMemcpyAsync(static_cast<char*>(b.p) + off, ring + sl * kRingChunk, nb, hipMemcpyHostToDevice, c->stream));
        OBTG_HIP(c, hipEventRecord(c->ring_ev[sl], c->stream));
        c->ring_pending[sl] = true;
    }
    return OBTG_OK;
}

// device -> host, complete on return for THIS array (the stream may still hold other work)
// the CPU address of `src` when it lies in the mapped host block of a staging buffer, else nullptr
static const void* mapped_alias(const obtg_ctx* c, const void* src)
{
    for (const DevBuf* b : { &c->ws_in, &c->ws_in2, &c->ws_out }) {
        if (!b->host) continue;
        const char* lo = static_cast<const char*>(b->host_dev);
        const char* q = static_cast<const char*>(src);
        if (q >= lo && q < lo + kZeroCopyBytes) return static_cast<const char*>(b->host) + (q - lo);
    }
    return nullptr;
}

static int d2h_copy(obtg_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return OBTG_OK;
    if (const void* h = mapped_alias(c, src)) {     // the kernel wrote host memory itself: wait for it, then a plain copy
        OBTG_HIP(c, hipStreamSynchronize(c->stream));
        std::memcpy(dst, h, bytes);
        return OBTG_OK;
    }
    if (bytes < kRingMin || is_pinned(dst) || ensure_ring(c) != OBTG_OK) {
        OBTG_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
        return OBTG_OK;
    }
    char* ring = static_cast<char*>(c->ring);
    const size_t n_chunks = (bytes + kRingChunk - 1) / kRingChunk;
    int slot_of[kRingSlots];
    int rc;
    auto issue = [&](size_t i) -> int {
        const size_t off = i * kRingChunk, nb = std::min(kRingChunk, bytes - off);
        const int sl = c->ring_next;
        c->ring_next = (c->ring_next + 1) % kRingSlots;
        slot_of[i % kRingSlots] = sl;
        // a pending H2D out of this slot precedes us on the same stream: DMA order is safe, only CPU access waits
        OBTG_HIP(c, hipMemcpyAsync(ring + sl * kRingChunk, static_cast<const char*>(src) + off, nb, hipMemcpyDeviceToHost, c->stream));
        OBTG_HIP(c, hipEventRecord(c->ring_ev[sl], c->stream));
        c->ring_pending[sl] = true;
        return OBTG_OK;
    };
    size_t issued = 0;
    for (; issued < n_chunks && issued < (size_t)kRingSlots; ++issued) if ((rc = issue(issued))) return rc;
    for (size_t i = 0; i < n_chunks; ++i) {
        const int sl = slot_of[i % kRingSlots];
        if ((rc = ring_slot_ready(c, sl))) return rc;
        const size_t off = i * kRingChunk, nb = std::min(kRingChunk, bytes - off);
        std::memcpy(static_cast<char*>(dst) + off, ring + sl * kRingChunk, nb);
        if (issued < n_chunks) { if ((rc = issue(issued))) return rc; ++issued; }   // the slot just emptied is next in the rotation
    }
    return OBTG_OK;
}

// the last output of a host entry point: copy, then leave the device idle
static int d2h(obtg_ctx* c, void* dst, const void* src, size_t bytes)
{
    int rc = d2h_copy(c, dst, src, bytes);
    if (rc) return rc;
    OBTG_HIP(c, hipStreamSynchronize(c->stream));
    for (bool& f : c->ring_pending) f = false;
    flush_pending_events(c);
    return OBTG_OK;
}

static size_t ysize(const obtg_ctx* c) { return (size_t)c->n_veh * c->dim * (c->deg + 1); }

static int host_sep(obtg_ctx* c, const double* Y, int B, double max_sep, bool min_only, double* out,
                    int pair_begin = 0, int pair_count = -1)
{
    if (!check_ctx(c) || !Y || !out || B < 0) return OBTG_ERR_ARG;
    if (pair_count < 0) pair_count = c->n_pairs - pair_begin;
    if (pair_begin < 0 || pair_begin + pair_count > c->n_pairs) return OBTG_ERR_ARG;
    if (B == 0 || pair_count == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    const size_t per = min_only ? (size_t)pair_count : (size_t)pair_count * (2 * c->deg + c->R + 1);
    int rc = h2d(c, c->ws_in, Y, sizeof(double) * ysize(c) * B, true);
    if (rc) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * per * B, true))) return rc;
    rc = launch_temporal_sep(c, c->ws_in.as<double>(), B, max_sep, pair_begin, pair_count, min_only,
                             c->ws_out.as<double>());
    if (rc) return rc;
    return d2h(c, out, c->ws_out.p, sizeof(double) * per * B);
}

int obtg_temporal_sep_active(obtg_ctx* c, const double* Y, int B, double max_sep, int k, double* out_val, int* out_idx)
{
    if (!check_ctx(c) || !Y || !out_val || B < 0 || k < 1 || k > 4 || k > 2 * c->deg + c->R + 1) return OBTG_ERR_ARG;
    if (B == 0 || c->n_pairs == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    const size_t n = (size_t)c->n_pairs * k * B;
    int rc = h2d(c, c->ws_in, Y, sizeof(double) * ysize(c) * B, true);
    if (rc) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * n))) return rc;
    DevBuf& di = c->ws_misc[4];
    if ((rc = di.reserve(sizeof(int) * n))) return rc;
    rc = launch_temporal_sep(c, c->ws_in.as<double>(), B, max_sep, 0, c->n_pairs, true, c->ws_out.as<double>(), k, di.as<int>());
    if (rc) return rc;
    if (out_idx && (rc = d2h_copy(c, out_idx, di.p, sizeof(int) * n))) return rc;
    return d2h(c, out_val, c->ws_out.p, sizeof(double) * n);
}

int obtg_temporal_sep_min_range(obtg_ctx* c, const double* Y, int B, double max_sep, int pair_begin,
                                int pair_count, double* out)
{
    if (pair_count < 0) return OBTG_ERR_ARG;
    return host_sep(c, Y, B, max_sep, true, out, pair_begin, pair_count);
}

int obtg_temporal_sep(obtg_ctx* c, const double* Y, int B, double max_sep, double* out)
{
    return host_sep(c, Y, B, max_sep, false, out);
}

int obtg_temporal_sep_min(obtg_ctx* c, const double* Y, int B, double max_sep, double* out)
{
    return host_sep(c, Y, B, max_sep, true, out);
}

// Examples/SequentialSwarm.py:43-70: one curve against K others, per-pair minimum of the elevated control points
int obtg_one_vs_many_min_dev(obtg_ctx* c, const double* d_one, int B, const double* d_many, int K, double max_sep, double* d_out)
{
    if (!check_ctx(c) || B < 0 || K < 0) return OBTG_ERR_ARG;
    if (B == 0 || K == 0) return OBTG_OK;
    if (!d_one || !d_many || !d_out) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    return launch_one_vs_many_min(c, d_one, B, d_many, K, max_sep, d_out);
}

int obtg_one_vs_many_min(obtg_ctx* c, const double* one, int B, const double* many, int K, double max_sep, double* out)
{
    if (!check_ctx(c) || B < 0 || K < 0) return OBTG_ERR_ARG;
    if (B == 0 || K == 0) return OBTG_OK;
    if (!one || !many || !out) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    const size_t curve = sizeof(double) * (size_t)c->dim * (c->deg + 1);
    int rc = h2d(c, c->ws_in, one, curve * B, true);
    if (rc) return rc;
    // the planned trajectories grow by one curve per vehicle: the staging buffer grows with them, nothing else does
    if ((rc = h2d(c, c->ws_in2, many, curve * K))) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * (size_t)B * K, true))) return rc;
    if ((rc = launch_one_vs_many_min(c, c->ws_in.as<double>(), B, c->ws_in2.as<double>(), K, max_sep, c->ws_out.as<double>()))) return rc;
    return d2h(c, out, c->ws_out.p, sizeof(double) * (size_t)B * K);
}

static int check_perts(const obtg_ctx* c, int n_pert, const int* prow, const int* pcol)
{
    for (int t = 0; t < n_pert; ++t)
        if (prow[t] < 0 || prow[t] >= c->n_veh * c->dim || pcol[t] < 0 || pcol[t] > c->deg) return OBTG_ERR_ARG;
    return OBTG_OK;
}

int obtg_temporal_sep_fd(obtg_ctx* c, const double* Y0, int n_pert, const int* pert_row, const int* pert_col,
                         const double* pert_val, double max_sep, double* out_blk)
{
    if (!check_ctx(c) || n_pert < 0) return OBTG_ERR_ARG;
    if (n_pert == 0 || c->n_obj < 2) return OBTG_OK;
    if (!Y0 || !pert_row || !pert_col || !pert_val || !out_blk) return OBTG_ERR_ARG;
    if (int rc = check_perts(c, n_pert, pert_row, pert_col)) return rc;
    (void)hipSetDevice(c->device);
    const size_t per = (size_t)(c->n_obj - 1) * (2 * c->deg + c->R + 1);
    int rc = h2d(c, c->ws_in, Y0, sizeof(double) * ysize(c));
    if (rc) return rc;
    if ((rc = h2d(c, c->ws_misc[0], pert_row, sizeof(int) * (size_t)n_pert))) return rc;
    if ((rc = h2d(c, c->ws_misc[1], pert_col, sizeof(int) * (size_t)n_pert))) return rc;
    if ((rc = h2d(c, c->ws_in2, pert_val, sizeof(double) * (size_t)n_pert))) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * per * n_pert))) return rc;
    rc = launch_temporal_sep_fd(c, c->ws_in.as<double>(), n_pert, c->ws_misc[0].as<int>(), c->ws_misc[1].as<int>(),
                                c->ws_in2.as<double>(), max_sep, c->ws_out.as<double>());
    if (rc) return rc;
    return d2h(c, out_blk, c->ws_out.p, sizeof(double) * per * n_pert);
}

int obtg_temporal_sep_fd_dev(obtg_ctx* c, const double* dY0, int n_pert, const int* d_pert_row,
                             const int* d_pert_col, const double* d_pert_val, double max_sep, double* d_out_blk)
{
    if (!check_ctx(c) || n_pert < 0) return OBTG_ERR_ARG;
    if (n_pert == 0 || c->n_obj < 2) return OBTG_OK;
    if (!dY0 || !d_pert_row || !d_pert_col || !d_pert_val || !d_out_blk) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    return launch_temporal_sep_fd(c, dY0, n_pert, d_pert_row, d_pert_col, d_pert_val, max_sep, d_out_blk);
}

int obtg_speed(obtg_ctx* c, const double* Y, const double* tf, int B, double bound, int is_max, double* out)
{
    if (!check_ctx(c) || !Y || !tf || !out || B < 0) return OBTG_ERR_ARG;
    if (B == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    const size_t per = (size_t)obtg_len_speed(c);
    int rc = h2d(c, c->ws_in, Y, sizeof(double) * ysize(c) * B, true);
    if (rc) return rc;
    if ((rc = h2d(c, c->ws_in2, tf, sizeof(double) * B, true))) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * per * B, true))) return rc;
    rc = launch_speed(c, c->ws_in.as<double>(), c->ws_in2.as<double>(), B, bound, is_max, c->ws_out.as<double>());
    if (rc) return rc;
    return d2h(c, out, c->ws_out.p, sizeof(double) * per * B);
}

int obtg_ang_rate(obtg_ctx* c, const double* Y, const double* tf, int B, double max_rate, double* out)
{
    if (!check_ctx(c) || !Y || !tf || !out || B < 0) return OBTG_ERR_ARG;
    if (c->dim != 2) return OBTG_ERR_ARG;
    if (B == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    const size_t per = (size_t)obtg_len_ang_rate(c);
    int rc = h2d(c, c->ws_in, Y, sizeof(double) * ysize(c) * B, true);
    if (rc) return rc;
    if ((rc = h2d(c, c->ws_in2, tf, sizeof(double) * B, true))) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * per * B, true))) return rc;
    rc = launch_ang_rate(c, c->ws_in.as<double>(), c->ws_in2.as<double>(), B, max_rate, c->ws_out.as<double>());
    if (rc) return rc;
    return d2h(c, out, c->ws_out.p, sizeof(double) * per * B);
}

// ------------------------------------------------------------------ GJK
// AoS pts[n][3] + offsets -> SoA per polygon (x[K] y[K] z[K])
static std::vector<double> to_soa(const double* pts, const int* off, int n_poly)
{
    std::vector<double> s((size_t)3 * off[n_poly]);
    for (int a = 0; a < n_poly; ++a) {
        const int o = off[a], K = off[a + 1] - o;
        for (int k = 0; k < K; ++k)
            for (int cdim = 0; cdim < 3; ++cdim) s[(size_t)3 * o + (size_t)cdim * K + k] = pts[(size_t)3 * (o + k) + cdim];
    }
    return s;
}

static int check_polys(const int* off, int n_poly, int n_pts)
{
    if (!off || n_poly < 0) return OBTG_ERR_ARG;
    if (off[0] != 0 || off[n_poly] != n_pts) return OBTG_ERR_ARG;
    for (int a = 0; a < n_poly; ++a) if (off[a + 1] - off[a] < 1) return OBTG_ERR_ARG;
    return OBTG_OK;
}

int obtg_gjk_pairs(obtg_ctx* c, const double* pts, int n_pts, const int* poly_off, int n_poly,
                   const int* pair_a, const int* pair_b, int n_pairs, int max_iter, int md_cap, int* flag,
                   double* p1, double* p2, double* dist, short* support_trace, int trace_cap,
                   int* n_support, int* status)
{
    if (!check_ctx(c) || !pts || !pair_a || !pair_b || !flag || !p1 || !p2 || !dist) return OBTG_ERR_ARG;
    if (n_pairs < 0 || max_iter < 1 || md_cap < 1 || trace_cap < 0) return OBTG_ERR_ARG;
    int rc = check_polys(poly_off, n_poly, n_pts);
    if (rc) return rc;
    for (int k = 0; k < n_pairs; ++k)
        if (pair_a[k] < 0 || pair_a[k] >= n_poly || pair_b[k] < 0 || pair_b[k] >= n_poly) return OBTG_ERR_ARG;
    if (n_pairs == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    auto soa = to_soa(pts, poly_off, n_poly);
    DevBuf* m = c->ws_misc;
    if ((rc = h2d(c, c->ws_in, soa.data(), soa.size() * sizeof(double)))) return rc;
    if ((rc = h2d(c, m[0], poly_off, sizeof(int) * (n_poly + 1)))) return rc;
    if ((rc = h2d(c, m[1], pair_a, sizeof(int) * n_pairs))) return rc;
    if ((rc = h2d(c, m[2], pair_b, sizeof(int) * n_pairs))) return rc;
    // outputs: flag | nsup | status (ints) ; p1 | p2 | dist (doubles) ; trace
    if ((rc = m[3].reserve(sizeof(int) * 3 * (size_t)n_pairs))) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * 7 * (size_t)n_pairs))) return rc;
    short* d_trace = nullptr;
    if (support_trace && trace_cap > 0) {
        if ((rc = m[4].reserve(sizeof(short) * 2 * (size_t)trace_cap * n_pairs))) return rc;
        d_trace = m[4].as<short>();
        OBTG_HIP(c, hipMemsetAsync(d_trace, 0, sizeof(short) * 2 * (size_t)trace_cap * n_pairs, c->stream));
    }
    int* d_flag = m[3].as<int>();
    int* d_nsup = d_flag + n_pairs;
    int* d_status = d_nsup + n_pairs;
    double* d_p1 = c->ws_out.as<double>();
    double* d_p2 = d_p1 + 3 * (size_t)n_pairs;
    double* d_dist = d_p2 + 3 * (size_t)n_pairs;
    bool planar = true;
    for (int k = 0; k < n_pts && planar; ++k) planar = pts[3 * (size_t)k + 2] == 0.0;
    rc = launch_gjk_pairs(c, c->ws_in.as<double>(), m[0].as<int>(), m[1].as<int>(), m[2].as<int>(), n_pairs,
                          max_iter, md_cap, d_flag, d_p1, d_p2, d_dist, d_trace, trace_cap, d_nsup, d_status,
                          planar);
    if (rc) return rc;
    if ((rc = d2h_copy(c, flag, d_flag, sizeof(int) * n_pairs))) return rc;
    if (n_support) if ((rc = d2h_copy(c, n_support, d_nsup, sizeof(int) * n_pairs))) return rc;
    if (status) if ((rc = d2h_copy(c, status, d_status, sizeof(int) * n_pairs))) return rc;
    if ((rc = d2h_copy(c, p1, d_p1, sizeof(double) * 3 * n_pairs))) return rc;
    if ((rc = d2h_copy(c, p2, d_p2, sizeof(double) * 3 * n_pairs))) return rc;
    if (d_trace)
        OBTG_HIP(c, hipMemcpyAsync(support_trace, d_trace, sizeof(short) * 2 * (size_t)trace_cap * n_pairs,
                                   hipMemcpyDeviceToHost, c->stream));
    return d2h(c, dist, d_dist, sizeof(double) * n_pairs);
}

int obtg_gjk_true_pairs(obtg_ctx* c, const double* pts, int n_pts, const int* poly_off, int n_poly, const int* pair_a,
                        const int* pair_b, int n_pairs, double eps, int max_iter, int* flag, double* p1, double* p2,
                        double* dist, double* lower, int* iters, int* status)
{
    if (!check_ctx(c) || !pts || !pair_a || !pair_b || !flag || !p1 || !p2 || !dist) return OBTG_ERR_ARG;
    if (n_pairs < 0 || max_iter < 1 || !(eps > 0)) return OBTG_ERR_ARG;
    int rc = check_polys(poly_off, n_poly, n_pts);
    if (rc) return rc;
    for (int k = 0; k < n_pairs; ++k)
        if (pair_a[k] < 0 || pair_a[k] >= n_poly || pair_b[k] < 0 || pair_b[k] >= n_poly) return OBTG_ERR_ARG;
    if (n_pairs == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    auto soa = to_soa(pts, poly_off, n_poly);
    DevBuf* m = c->ws_misc;
    if ((rc = h2d(c, c->ws_in, soa.data(), soa.size() * sizeof(double)))) return rc;
    if ((rc = h2d(c, m[0], poly_off, sizeof(int) * (n_poly + 1)))) return rc;
    if ((rc = h2d(c, m[1], pair_a, sizeof(int) * n_pairs))) return rc;
    if ((rc = h2d(c, m[2], pair_b, sizeof(int) * n_pairs))) return rc;
    if ((rc = m[3].reserve(sizeof(int) * 3 * (size_t)n_pairs))) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * 8 * (size_t)n_pairs))) return rc;
    int* d_flag = m[3].as<int>();
    int* d_iters = d_flag + n_pairs;
    int* d_status = d_iters + n_pairs;
    double* d_p1 = c->ws_out.as<double>();
    double* d_p2 = d_p1 + 3 * (size_t)n_pairs;
    double* d_dist = d_p2 + 3 * (size_t)n_pairs;
    double* d_lower = d_dist + n_pairs;
    rc = launch_gjk_true_pairs(c, c->ws_in.as<double>(), m[0].as<int>(), m[1].as<int>(), m[2].as<int>(), n_pairs, eps,
                               max_iter, d_flag, d_p1, d_p2, d_dist, d_lower, d_iters, d_status);
    if (rc) return rc;
    if ((rc = d2h_copy(c, flag, d_flag, sizeof(int) * n_pairs))) return rc;
    if (iters && (rc = d2h_copy(c, iters, d_iters, sizeof(int) * n_pairs))) return rc;
    if (status && (rc = d2h_copy(c, status, d_status, sizeof(int) * n_pairs))) return rc;
    if ((rc = d2h_copy(c, p1, d_p1, sizeof(double) * 3 * n_pairs))) return rc;
    if ((rc = d2h_copy(c, p2, d_p2, sizeof(double) * 3 * n_pairs))) return rc;
    if (lower && (rc = d2h_copy(c, lower, d_lower, sizeof(double) * n_pairs))) return rc;
    return d2h(c, dist, d_dist, sizeof(double) * n_pairs);
}

int obtg_min_dist2poly_robust(obtg_ctx* c, const double* curves, int n_curves, int K, const double* pts, int n_pts,
                              const int* poly_off, int n_poly, const int* pair_curve, const int* pair_poly, int n_pairs,
                              double eps, int max_nodes, double* res, int* info, int* status)
{
    if (!check_ctx(c) || !curves || !pts || !pair_curve || !pair_poly || !res || n_curves < 1 || n_pairs < 0)
        return OBTG_ERR_ARG;
    if (K < 2 || max_nodes < 1 || !(eps > 0)) return OBTG_ERR_ARG;
    int rc = check_polys(poly_off, n_poly, n_pts);
    if (rc) return rc;
    for (int k = 0; k < n_pairs; ++k)
        if (pair_curve[k] < 0 || pair_curve[k] >= n_curves || pair_poly[k] < 0 || pair_poly[k] >= n_poly)
            return OBTG_ERR_ARG;
    if (n_pairs == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    constexpr int kCap = 1024, kMaxLevel = 48;
    DevBuf* m = c->ws_misc;
    auto soa = to_soa(pts, poly_off, n_poly);
    int max_K = 0;
    for (int a = 0; a < n_poly; ++a) max_K = std::max(max_K, poly_off[a + 1] - poly_off[a]);
    if ((rc = h2d(c, c->ws_in, curves, sizeof(double) * 3 * (size_t)K * n_curves))) return rc;
    if ((rc = h2d(c, c->ws_in2, soa.data(), soa.size() * sizeof(double)))) return rc;
    if ((rc = h2d(c, m[0], poly_off, sizeof(int) * (n_poly + 1)))) return rc;
    if ((rc = h2d(c, m[1], pair_curve, sizeof(int) * n_pairs))) return rc;
    if ((rc = h2d(c, m[2], pair_poly, sizeof(int) * n_pairs))) return rc;
    if ((rc = m[5].reserve(sizeof(double) * 2 * kCap * 2 * (size_t)n_pairs))) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * 5 * (size_t)n_pairs))) return rc;
    if ((rc = m[3].reserve(sizeof(int) * 4 * (size_t)n_pairs))) return rc;
    rc = launch_min_dist2poly_robust(c, c->ws_in.as<double>(), K, c->ws_in2.as<double>(), m[0].as<int>(), m[1].as<int>(),
                                     m[2].as<int>(), n_pairs, eps, max_nodes, kMaxLevel, kCap, max_K, m[5].as<double>(),
                                     c->ws_out.as<double>(), m[3].as<int>());
    if (rc) return rc;
    std::vector<int> hinfo((size_t)4 * n_pairs);
    if ((rc = d2h_copy(c, hinfo.data(), m[3].p, sizeof(int) * 4 * n_pairs))) return rc;
    rc = d2h(c, res, c->ws_out.p, sizeof(double) * 5 * n_pairs);
    if (rc) return rc;
    if (info) std::memcpy(info, hinfo.data(), sizeof(int) * 4 * n_pairs);
    if (status) for (int k = 0; k < n_pairs; ++k) status[k] = hinfo[4 * k + 3];
    return OBTG_OK;
}

int obtg_ctx_set_polygons(obtg_ctx* c, const double* pts, int n_pts, const int* poly_off, int n_poly)
{
    if (!check_ctx(c) || n_poly < 0 || n_pts < 0) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    OBTG_HIP(c, hipStreamSynchronize(c->stream));
    if (n_poly == 0) {
        c->n_poly = 0; c->n_poly_pts = 0; c->polys_planar = true; c->max_poly_K = 0;
        c->n_hull_pairs = 0; c->hull_pairs_set = false; c->tile_valid = false; c->gjk_len_rows = 0;
        int zero = 0;
        return upload(c, c->d_poly_off, &zero, sizeof(int));
    }
    if (!pts) return OBTG_ERR_ARG;
    int rc = check_polys(poly_off, n_poly, n_pts);
    if (rc) return rc;
    auto soa = to_soa(pts, poly_off, n_poly);
    if ((rc = upload(c, c->d_poly_pts, soa.data(), soa.size() * sizeof(double)))) return rc;
    if ((rc = upload(c, c->d_poly_off, poly_off, sizeof(int) * (n_poly + 1)))) return rc;
    c->n_poly = n_poly; c->n_poly_pts = n_pts;
    c->polys_planar = true;
    for (int k = 0; k < n_pts && c->polys_planar; ++k) c->polys_planar = pts[3 * (size_t)k + 2] == 0.0;
    c->max_poly_K = 0;
    for (int a = 0; a < n_poly; ++a) c->max_poly_K = std::max(c->max_poly_K, poly_off[a + 1] - poly_off[a]);
    c->n_hull_pairs = 0;   // object ids may have changed meaning
    c->hull_pairs_set = false;
    c->tile_valid = false;
    c->gjk_len_rows = 0;
    return OBTG_OK;
}

int obtg_ctx_set_hull_pairs(obtg_ctx* c, const int* pair_a, const int* pair_b, int n_pairs)
{
    if (!check_ctx(c) || n_pairs < 0 || (n_pairs && (!pair_a || !pair_b))) return OBTG_ERR_ARG;
    const int n_objs = c->n_veh + c->n_poly;
    for (int k = 0; k < n_pairs; ++k)
        if (pair_a[k] < 0 || pair_a[k] >= n_objs || pair_b[k] < 0 || pair_b[k] >= n_objs) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    OBTG_HIP(c, hipStreamSynchronize(c->stream));
    int rc = upload(c, c->d_hp_a, pair_a, sizeof(int) * (size_t)n_pairs);
    if (rc) return rc;
    if ((rc = upload(c, c->d_hp_b, pair_b, sizeof(int) * (size_t)n_pairs))) return rc;
    c->n_hull_pairs = n_pairs;
    c->hull_pairs_set = false;             // (true again once EVERY table of the list is on the device: a failed upload below
                                           //  must not leave the previous list's per-vehicle index marked valid)
    c->h_hp_a.assign(pair_a, pair_a + n_pairs);
    c->h_hp_b.assign(pair_b, pair_b + n_pairs);
    {   // which pairs contain vehicle v (list order): what a finite-difference row that moves v has to re-evaluate
        std::vector<int> off(c->n_veh + 1, 0), idx;
        for (int k = 0; k < n_pairs; ++k) {
            if (pair_a[k] < c->n_veh) ++off[pair_a[k] + 1];
            if (pair_b[k] < c->n_veh && pair_b[k] != pair_a[k]) ++off[pair_b[k] + 1];
        }
        for (int v = 0; v < c->n_veh; ++v) off[v + 1] += off[v];
        idx.resize((size_t)off[c->n_veh] + 1);
        std::vector<int> fill(off.begin(), off.end() - 1);
        for (int k = 0; k < n_pairs; ++k) {
            if (pair_a[k] < c->n_veh) idx[fill[pair_a[k]]++] = k;
            if (pair_b[k] < c->n_veh && pair_b[k] != pair_a[k]) idx[fill[pair_b[k]]++] = k;
        }
        if ((rc = upload(c, c->d_vp_off, off.data(), sizeof(int) * off.size()))) return rc;
        if ((rc = upload(c, c->d_vp_idx, idx.data(), sizeof(int) * idx.size()))) return rc;
    }
    c->tile_valid = false;
    c->gjk_len_rows = 0;
    if (c->d_poly_off.p == nullptr) {
        int zero = 0;
        if ((rc = upload(c, c->d_poly_off, &zero, sizeof(int)))) return rc;
    }
    if (c->d_poly_pts.p == nullptr && (rc = c->d_poly_pts.reserve(8))) return rc;
    c->hull_pairs_set = true;
    return OBTG_OK;
}

int obtg_ctx_set_fd_dedup(obtg_ctx* c, int on)
{
    if (!check_ctx(c)) return OBTG_ERR_ARG;
    c->fd_dedup = on != 0;
    return OBTG_OK;
}

int obtg_pair_sweep_dev(obtg_ctx* c, const double* dY, int B, double max_sep, double* d_out_sep, int max_iter,
                        int md_cap, int* d_flag, double* d_p1, double* d_p2, double* d_dist, int* d_nsup,
                        int* d_status)
{
    if (!check_ctx(c) || !d_out_sep || !d_flag || !d_p1 || !d_p2 || !d_dist || B < 0 || max_iter < 1 ||
        md_cap < 1) return OBTG_ERR_ARG;
    if (!c->hull_pairs_set) return OBTG_ERR_ARG;   // no pair list registered (or invalidated by obtg_ctx_set_polygons)
    (void)hipSetDevice(c->device);
    return with_batch(c, dY, B, pair_sweep_can_fd(c), [&](const double* src) {
        return launch_pair_sweep(c, src, B, max_sep, d_out_sep, max_iter, md_cap, d_flag, d_p1, d_p2, d_dist, d_nsup, d_status); });
}

int obtg_constraint_sweep_dev(obtg_ctx* c, const double* dY, const double* d_tf, int B, double max_sep, double* d_out_sep,
                              double speed_bound, int speed_is_max, double max_rate, double* d_out_speed,
                              double* d_out_ang, int max_iter, int md_cap, int* d_flag, double* d_p1, double* d_p2,
                              double* d_dist, int* d_nsup, int* d_status)
{
    if (!check_ctx(c) || !d_tf || !d_out_sep || !d_out_speed || !d_flag || !d_p1 || !d_p2 || !d_dist || B < 0 ||
        max_iter < 1 || md_cap < 1) return OBTG_ERR_ARG;
    if (!c->hull_pairs_set) return OBTG_ERR_ARG;
    if (d_out_ang && c->dim != 2) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    return with_batch(c, dY, B, pair_sweep_can_fd(c) && dynamics_can_fd(c, true, d_out_ang != nullptr), [&](const double* src) {
        SweepFold sp;
        sp.d_tf = d_tf; sp.speed_bound = speed_bound; sp.speed_is_max = speed_is_max;
        sp.d_out_speed = d_out_speed; sp.d_out_ang = d_out_ang; sp.max_rate = max_rate;
        int rc = launch_pair_sweep(c, src, B, max_sep, d_out_sep, max_iter, md_cap, d_flag, d_p1, d_p2, d_dist, d_nsup, d_status, &sp);
        if (rc) return rc;
        if (sp.did_dynamics || (sp.did_speed && !d_out_ang)) return (int)OBTG_OK;
        return launch_dynamics(c, src, d_tf, B, speed_bound, speed_is_max, max_rate, d_out_speed, d_out_ang);
    });
}

int obtg_constraint_sweep_fd_structured_rows_dev(obtg_ctx* c, const double* dY0, int n_fixed_cols, double h, int row_begin,
                                                 const double* d_tf, int B, double max_sep, double* d_out_sep, double speed_bound,
                                                 int speed_is_max, double max_rate, double* d_out_speed, double* d_out_ang,
                                                 int max_iter, int md_cap, int* d_flag, double* d_p1, double* d_p2, double* d_dist,
                                                 int* d_nsup, int* d_status)
{
    if (!check_ctx(c) || !dY0 || !d_tf || !d_out_sep || !d_out_speed || !d_out_ang || !d_flag || !d_p1 || !d_p2 || !d_dist ||
        B < 1 || row_begin < 0 || max_iter < 1 || md_cap < 1) return OBTG_ERR_ARG;
    if (!c->hull_pairs_set || c->dim != 2) return OBTG_ERR_ARG;
    if (c->view.Y0) return OBTG_ERR_ARG;       // a view of the caller's is open: this call is a view of its own and would replace and close it
    int rc = obtg_fd_view_begin_rows(c, dY0, n_fixed_cols, h, row_begin, B);
    if (rc) return rc;
    (void)hipSetDevice(c->device);
    SweepFold sp;
    sp.d_tf = d_tf; sp.speed_bound = speed_bound; sp.speed_is_max = speed_is_max;
    sp.d_out_speed = d_out_speed; sp.d_out_ang = d_out_ang; sp.max_rate = max_rate;
    c->fd.Y0 = c->view.Y0; c->fd.h = c->view.h; c->fd.fixed = c->view.fixed; c->fd.row0 = c->view.row0;
    rc = launch_step_fd_structured(c, B, max_sep, d_out_sep, max_iter, md_cap, d_flag, d_p1, d_p2, d_dist, d_nsup, d_status, &sp);
    c->fd.Y0 = nullptr; c->fd.row0 = 0;
    (void)obtg_fd_view_end(c);
    return rc;
}

int obtg_constraint_sweep_fd_structured_dev(obtg_ctx* c, const double* dY0, int n_fixed_cols, double h, const double* d_tf, int B,
                                            double max_sep, double* d_out_sep, double speed_bound, int speed_is_max, double max_rate,
                                            double* d_out_speed, double* d_out_ang, int max_iter, int md_cap, int* d_flag,
                                            double* d_p1, double* d_p2, double* d_dist, int* d_nsup, int* d_status)
{
    return obtg_constraint_sweep_fd_structured_rows_dev(c, dY0, n_fixed_cols, h, 0, d_tf, B, max_sep, d_out_sep, speed_bound, speed_is_max,
                                                        max_rate, d_out_speed, d_out_ang, max_iter, md_cap, d_flag, d_p1, d_p2, d_dist,
                                                        d_nsup, d_status);
}

int obtg_ctx_set_gjk_history(obtg_ctx* c, int on)
{
    if (!check_ctx(c)) return OBTG_ERR_ARG;
    c->gjk_history = on != 0;
    c->gjk_len_rows = 0;
    return OBTG_OK;
}

int obtg_gjk_swarm_dev(obtg_ctx* c, const double* dY, int B, int max_iter, int md_cap, int* d_flag,
                       double* d_p1, double* d_p2, double* d_dist, int* d_nsup, int* d_status)
{
    if (!check_ctx(c) || !d_flag || !d_p1 || !d_p2 || !d_dist || B < 0) return OBTG_ERR_ARG;
    if (max_iter < 1 || md_cap < 1) return OBTG_ERR_ARG;
    if (c->dim < 2) return OBTG_ERR_ARG;   // bezier.py:847-851: curves must be 2-D or 3-D
    if (!c->hull_pairs_set) return OBTG_ERR_ARG;   // no pair list registered (or invalidated by obtg_ctx_set_polygons)
    (void)hipSetDevice(c->device);
    return with_batch(c, dY, B, true, [&](const double* src) {
        return launch_gjk_swarm(c, src, B, max_iter, md_cap, d_flag, d_p1, d_p2, d_dist, d_nsup, d_status); });
}

int obtg_gjk_swarm(obtg_ctx* c, const double* Y, int B, int max_iter, int md_cap, int* flag, double* p1,
                   double* p2, double* dist, int* nsup, int* status)
{
    if (!check_ctx(c) || !Y || !flag || !p1 || !p2 || !dist || B < 0) return OBTG_ERR_ARG;
    if (max_iter < 1 || md_cap < 1) return OBTG_ERR_ARG;
    if (c->dim < 2 || !c->hull_pairs_set) return OBTG_ERR_ARG;
    const size_t n = (size_t)B * c->n_hull_pairs;
    if (n == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    DevBuf* m = c->ws_misc;
    int rc = h2d(c, c->ws_in, Y, sizeof(double) * ysize(c) * B);
    if (rc) return rc;
    if ((rc = m[3].reserve(sizeof(int) * 3 * n))) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * 7 * n))) return rc;
    int* d_flag = m[3].as<int>();
    int* d_nsup = d_flag + n;
    int* d_status = d_nsup + n;
    double* d_p1 = c->ws_out.as<double>();
    double* d_p2 = d_p1 + 3 * n;
    double* d_dist = d_p2 + 3 * n;
    rc = launch_gjk_swarm(c, c->ws_in.as<double>(), B, max_iter, md_cap, d_flag, d_p1, d_p2, d_dist, d_nsup, d_status);
    if (rc) return rc;
    if ((rc = d2h_copy(c, flag, d_flag, sizeof(int) * n))) return rc;
    if (nsup) if ((rc = d2h_copy(c, nsup, d_nsup, sizeof(int) * n))) return rc;
    if (status) if ((rc = d2h_copy(c, status, d_status, sizeof(int) * n))) return rc;
    if ((rc = d2h_copy(c, p1, d_p1, sizeof(double) * 3 * n))) return rc;
    if ((rc = d2h_copy(c, p2, d_p2, sizeof(double) * 3 * n))) return rc;
    return d2h(c, dist, d_dist, sizeof(double) * n);
}

// ------------------------------------------------------------------ minDist
int obtg_min_dist(obtg_ctx* c, const double* curves, int n_curves, int K, const int* pair_a, const int* pair_b,
                  int n_pairs, double eps, int max_iter, int md_cap, int max_depth, int max_nodes, double* res,
                  int* info, int* status)
{
    if (!check_ctx(c) || !curves || !pair_a || !pair_b || !res || n_curves < 1 || n_pairs < 0) return OBTG_ERR_ARG;
    if (K < 2 || max_iter < 1 || md_cap < 1 || max_depth < 1 || max_nodes < 1) return OBTG_ERR_ARG;
    for (int k = 0; k < n_pairs; ++k)
        if (pair_a[k] < 0 || pair_a[k] >= n_curves || pair_b[k] < 0 || pair_b[k] >= n_curves) return OBTG_ERR_ARG;
    if (n_pairs == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    DevBuf* m = c->ws_misc;
    int rc = h2d(c, c->ws_in, curves, sizeof(double) * 3 * (size_t)K * n_curves);
    if (rc) return rc;
    if ((rc = h2d(c, m[1], pair_a, sizeof(int) * n_pairs))) return rc;
    if ((rc = h2d(c, m[2], pair_b, sizeof(int) * n_pairs))) return rc;
    if ((rc = m[5].reserve(sizeof(double) * min_dist_stack_doubles(c, K, max_depth, n_pairs)))) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * 3 * (size_t)n_pairs))) return rc;
    if ((rc = m[3].reserve(sizeof(int) * 4 * (size_t)n_pairs))) return rc;
    // the order the worker waves take the pairs in: by the previous evaluation's node counts, longest search first, when
    // this very pair list was evaluated before (an SLSQP run evaluates one list over and over at nearby x); list order else
    unsigned long long sig = 1469598103934665603ull ^ (unsigned long long)n_pairs;
    for (int k = 0; k < n_pairs; ++k) {
        sig = (sig ^ (unsigned)pair_a[k]) * 1099511628211ull;
        sig = (sig ^ (unsigned)pair_b[k]) * 1099511628211ull;
    }
    static const bool use_hist = !(getenv("OBTG_MD_HISTORY") && getenv("OBTG_MD_HISTORY")[0] == '0');
    std::vector<int> qbuf((size_t)n_pairs + 1, 0);           // [0] the queue counter, [1..] the order
    int slot = -1;
    for (size_t h = 0; h < c->md_hist.size(); ++h)
        if (c->md_hist[h].sig == sig && (int)c->md_hist[h].nodes.size() == n_pairs) slot = (int)h;
    const bool have = use_hist && slot >= 0;
    for (int k = 0; k < n_pairs; ++k) qbuf[1 + k] = k;
    if (have) {
        const std::vector<int>& hn = c->md_hist[slot].nodes;
        std::stable_sort(qbuf.begin() + 1, qbuf.end(), [&](int a, int b) { return hn[a] > hn[b]; });
    }
    if ((rc = h2d(c, m[6], qbuf.data(), sizeof(int) * qbuf.size()))) return rc;
    // 2-D curves arrive padded with a zero z row (bezier.py:1294-1308): then the planar gjkNew machine runs (the same bits)
    bool planar = true;
    for (int i = 0; i < n_curves && planar; ++i) {
        const double* z = curves + ((size_t)i * 3 + 2) * K;
        for (int j = 0; j < K; ++j) if (z[j] != 0.0 || std::signbit(z[j])) { planar = false; break; }      // (+0 only: a -0 would show in a returned closest point)
    }
    rc = launch_min_dist(c, c->ws_in.as<double>(), K, m[1].as<int>(), m[2].as<int>(), n_pairs, eps, max_iter,
                         md_cap, max_depth, max_nodes, m[5].as<double>(), c->ws_out.as<double>(), m[3].as<int>(),
                         have ? m[6].as<int>() + 1 : nullptr, m[6].as<int>(), planar);
    if (rc) return rc;
    std::vector<int> hinfo((size_t)4 * n_pairs);
    if ((rc = d2h_copy(c, hinfo.data(), m[3].p, sizeof(int) * 4 * n_pairs))) return rc;
    if (slot >= 0) c->md_hist.erase(c->md_hist.begin() + slot);
    if (c->md_hist.size() >= 4) c->md_hist.erase(c->md_hist.begin());
    c->md_hist.push_back({ sig, std::vector<int>((size_t)n_pairs) });
    for (int k = 0; k < n_pairs; ++k) c->md_hist.back().nodes[k] = hinfo[4 * k];
    rc = d2h(c, res, c->ws_out.p, sizeof(double) * 3 * n_pairs);
    if (rc) return rc;
    if (info) std::memcpy(info, hinfo.data(), sizeof(int) * 4 * n_pairs);
    if (status) for (int k = 0; k < n_pairs; ++k) status[k] = hinfo[4 * k + 3];
    return OBTG_OK;
}

int obtg_min_dist_robust(obtg_ctx* c, const double* curves, int n_curves, int K, const int* pair_a, const int* pair_b,
                         int n_pairs, double eps, int max_nodes, double* res, int* info, int* status)
{
    if (!check_ctx(c) || !curves || !pair_a || !pair_b || !res || n_curves < 1 || n_pairs < 0) return OBTG_ERR_ARG;
    if (K < 2 || max_nodes < 1 || !(eps > 0)) return OBTG_ERR_ARG;
    for (int k = 0; k < n_pairs; ++k)
        if (pair_a[k] < 0 || pair_a[k] >= n_curves || pair_b[k] < 0 || pair_b[k] >= n_curves) return OBTG_ERR_ARG;
    if (n_pairs == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    constexpr int kCap = 1024, kMaxLevel = 48;
    DevBuf* m = c->ws_misc;
    int rc = h2d(c, c->ws_in, curves, sizeof(double) * 3 * (size_t)K * n_curves);
    if (rc) return rc;
    if ((rc = h2d(c, m[1], pair_a, sizeof(int) * n_pairs))) return rc;
    if ((rc = h2d(c, m[2], pair_b, sizeof(int) * n_pairs))) return rc;
    if ((rc = m[5].reserve(sizeof(double) * 2 * kCap * 3 * (size_t)n_pairs))) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * 3 * (size_t)n_pairs))) return rc;
    if ((rc = m[3].reserve(sizeof(int) * 4 * (size_t)n_pairs))) return rc;
    rc = launch_min_dist_robust(c, c->ws_in.as<double>(), K, m[1].as<int>(), m[2].as<int>(), n_pairs, eps, max_nodes,
                                kMaxLevel, kCap, m[5].as<double>(), c->ws_out.as<double>(), m[3].as<int>());
    if (rc) return rc;
    std::vector<int> hinfo((size_t)4 * n_pairs);
    if ((rc = d2h_copy(c, hinfo.data(), m[3].p, sizeof(int) * 4 * n_pairs))) return rc;
    rc = d2h(c, res, c->ws_out.p, sizeof(double) * 3 * n_pairs);
    if (rc) return rc;
    if (info) std::memcpy(info, hinfo.data(), sizeof(int) * 4 * n_pairs);
    if (status) for (int k = 0; k < n_pairs; ++k) status[k] = hinfo[4 * k + 3];
    return OBTG_OK;
}

int obtg_min_dist2poly(obtg_ctx* c, const double* curves, int n_curves, int K, const double* pts, int n_pts,
                       const int* poly_off, int n_poly, const int* pair_curve, const int* pair_poly, int n_pairs,
                       double eps, int max_iter, int md_cap, int max_depth, int max_nodes, double* res, int* info,
                       int* status)
{
    if (!check_ctx(c) || !curves || !pts || !pair_curve || !pair_poly || !res || n_curves < 1 || n_pairs < 0)
        return OBTG_ERR_ARG;
    if (K < 2 || max_iter < 1 || md_cap < 1 || max_depth < 1 || max_nodes < 1) return OBTG_ERR_ARG;
    int rc = check_polys(poly_off, n_poly, n_pts);
    if (rc) return rc;
    for (int k = 0; k < n_pairs; ++k)
        if (pair_curve[k] < 0 || pair_curve[k] >= n_curves || pair_poly[k] < 0 || pair_poly[k] >= n_poly)
            return OBTG_ERR_ARG;
    if (n_pairs == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    DevBuf* m = c->ws_misc;
    auto soa = to_soa(pts, poly_off, n_poly);
    int max_K = 0;
    for (int a = 0; a < n_poly; ++a) max_K = std::max(max_K, poly_off[a + 1] - poly_off[a]);
    if ((rc = h2d(c, c->ws_in, curves, sizeof(double) * 3 * (size_t)K * n_curves))) return rc;
    if ((rc = h2d(c, c->ws_in2, soa.data(), soa.size() * sizeof(double)))) return rc;
    if ((rc = h2d(c, m[0], poly_off, sizeof(int) * (n_poly + 1)))) return rc;
    if ((rc = h2d(c, m[1], pair_curve, sizeof(int) * n_pairs))) return rc;
    if ((rc = h2d(c, m[2], pair_poly, sizeof(int) * n_pairs))) return rc;
    if ((rc = m[5].reserve(sizeof(double) * min_dist2poly_stack_doubles(K, max_depth) * n_pairs))) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * 5 * (size_t)n_pairs))) return rc;
    if ((rc = m[3].reserve(sizeof(int) * 4 * (size_t)n_pairs))) return rc;
    bool planar = true;             // 2-D curves against polygons in the plane (bezier.py:1416-1430 pads both with z = 0)
    for (int i = 0; i < n_curves && planar; ++i) {
        const double* z = curves + ((size_t)i * 3 + 2) * K;
        for (int j = 0; j < K; ++j) if (z[j] != 0.0 || std::signbit(z[j])) { planar = false; break; }      // (+0 only: a -0 would show in a returned closest point)
    }
    for (int i = 0; i < n_pts && planar; ++i) if (pts[3 * (size_t)i + 2] != 0.0 || std::signbit(pts[3 * (size_t)i + 2])) planar = false;
    rc = launch_min_dist2poly(c, c->ws_in.as<double>(), K, c->ws_in2.as<double>(), m[0].as<int>(), m[1].as<int>(),
                              m[2].as<int>(), n_pairs, eps, max_iter, md_cap, max_depth, max_nodes,
                              m[5].as<double>(), c->ws_out.as<double>(), m[3].as<int>(), max_K, planar);
    if (rc) return rc;
    std::vector<int> hinfo((size_t)4 * n_pairs);
    if ((rc = d2h_copy(c, hinfo.data(), m[3].p, sizeof(int) * 4 * n_pairs))) return rc;
    rc = d2h(c, res, c->ws_out.p, sizeof(double) * 5 * n_pairs);
    if (rc) return rc;
    if (info) std::memcpy(info, hinfo.data(), sizeof(int) * 4 * n_pairs);
    if (status) for (int k = 0; k < n_pairs; ++k) status[k] = hinfo[4 * k + 3];
    return OBTG_OK;
}

// ------------------------------------------------------------------ single-curve algebra
int obtg_bern_elev(obtg_ctx* c, const double* in, int rows, int n, int R, double* out)
{
    if (!check_ctx(c) || !in || !out || rows < 0 || n < 0 || R < 0) return OBTG_ERR_ARG;
    if (rows == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    int rc = h2d(c, c->ws_in, in, sizeof(double) * (size_t)rows * (n + 1));
    if (rc) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * (size_t)rows * (n + R + 1)))) return rc;
    if ((rc = launch_bern_elev(c, c->ws_in.as<double>(), rows, n, R, c->ws_out.as<double>()))) return rc;
    return d2h(c, out, c->ws_out.p, sizeof(double) * (size_t)rows * (n + R + 1));
}

int obtg_bern_diff(obtg_ctx* c, const double* in, int rows, int n, double T, double* out)
{
    if (!check_ctx(c) || !in || !out || rows < 0 || n < 1) return OBTG_ERR_ARG;
    if (rows == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    int rc = h2d(c, c->ws_in, in, sizeof(double) * (size_t)rows * (n + 1));
    if (rc) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * (size_t)rows * (n + 1)))) return rc;
    if ((rc = launch_bern_diff(c, c->ws_in.as<double>(), rows, n, T, c->ws_out.as<double>()))) return rc;
    return d2h(c, out, c->ws_out.p, sizeof(double) * (size_t)rows * (n + 1));
}

int obtg_bern_split(obtg_ctx* c, const double* in, int rows, int n, double z, double* left, double* right)
{
    if (!check_ctx(c) || !in || !left || !right || rows < 0 || n < 0) return OBTG_ERR_ARG;
    if (rows == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    const size_t len = (size_t)rows * (n + 1);
    int rc = h2d(c, c->ws_in, in, sizeof(double) * len);
    if (rc) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * 2 * len))) return rc;
    double* dl = c->ws_out.as<double>();
    if ((rc = launch_bern_split(c, c->ws_in.as<double>(), rows, n, z, dl, dl + len))) return rc;
    if ((rc = d2h_copy(c, left, dl, sizeof(double) * len))) return rc;
    return d2h(c, right, dl + len, sizeof(double) * len);
}

int obtg_bern_eval(obtg_ctx* c, const double* cpts, int rows, int n, const double* tau, int n_tau, double t0, double tf, double* out)
{
    if (!check_ctx(c) || !cpts || !tau || !out || rows < 0 || n < 0 || n_tau < 0) return OBTG_ERR_ARG;
    if (rows == 0 || n_tau == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    int rc = h2d(c, c->ws_in, cpts, sizeof(double) * (size_t)rows * (n + 1));
    if (rc) return rc;
    if ((rc = h2d(c, c->ws_in2, tau, sizeof(double) * (size_t)n_tau))) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * (size_t)rows * n_tau))) return rc;
    if ((rc = launch_bern_eval(c, c->ws_in.as<double>(), rows, n, c->ws_in2.as<double>(), n_tau, t0, tf, c->ws_out.as<double>())))
        return rc;
    return d2h(c, out, c->ws_out.p, sizeof(double) * (size_t)rows * n_tau);
}

int obtg_bern_mul(obtg_ctx* c, const double* a, const double* b, int rows, int m, int n, double* out)
{
    if (!check_ctx(c) || !a || !b || !out || rows < 0 || m < 0 || n < 0) return OBTG_ERR_ARG;
    if (rows == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    int rc = h2d(c, c->ws_in, a, sizeof(double) * (size_t)rows * (m + 1));
    if (rc) return rc;
    if ((rc = h2d(c, c->ws_in2, b, sizeof(double) * (size_t)rows * (n + 1)))) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * (size_t)rows * (m + n + 1)))) return rc;
    if ((rc = launch_bern_mul(c, c->ws_in.as<double>(), c->ws_in2.as<double>(), rows, m, n, c->ws_out.as<double>())))
        return rc;
    return d2h(c, out, c->ws_out.p, sizeof(double) * (size_t)rows * (m + n + 1));
}

int obtg_bern_normsq(obtg_ctx* c, const double* x, int d, int n, double* out)
{
    if (!check_ctx(c) || !x || !out || d < 1 || n < 0) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    int rc = h2d(c, c->ws_in, x, sizeof(double) * (size_t)d * (n + 1));
    if (rc) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * (size_t)(2 * n + 1)))) return rc;
    if ((rc = launch_bern_normsq(c, c->ws_in.as<double>(), d, n, c->ws_out.as<double>()))) return rc;
    return d2h(c, out, c->ws_out.p, sizeof(double) * (size_t)(2 * n + 1));
}

// ------------------------------------------------------------------ objectives
int obtg_euclidean_obj(obtg_ctx* c, const double* Y, int B, double* out)
{
    if (!check_ctx(c) || !Y || !out || B < 0) return OBTG_ERR_ARG;
    if (B == 0) return OBTG_OK;
    (void)hipSetDevice(c->device);
    int rc = h2d(c, c->ws_in, Y, sizeof(double) * ysize(c) * B, true);
    if (rc) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * B, true))) return rc;
    if ((rc = launch_euclidean_obj(c, c->ws_in.as<double>(), B, c->ws_out.as<double>()))) return rc;
    return d2h(c, out, c->ws_out.p, sizeof(double) * B);
}

static int host_deriv_obj(obtg_ctx* c, const double* Y, const double* tf, int B, int order, double* out)
{
    if (!check_ctx(c) || !Y || !tf || !out || B < 0) return OBTG_ERR_ARG;
    if (B == 0) return OBTG_OK;
    // one final time for the whole batch, as the reference's objectives have (optimization.py:294-308)
    for (int b = 1; b < B; ++b) if (!(tf[b] == tf[0])) return OBTG_ERR_ARG;
    (void)hipSetDevice(c->device);
    int rc = h2d(c, c->ws_in, Y, sizeof(double) * ysize(c) * B, true);
    if (rc) return rc;
    if ((rc = h2d(c, c->ws_in2, tf, sizeof(double) * B, true))) return rc;
    if ((rc = c->ws_out.reserve(sizeof(double) * B, true))) return rc;
    if ((rc = launch_deriv_energy_obj(c, c->ws_in.as<double>(), c->ws_in2.as<double>(), tf[0], B, order, c->ws_out.as<double>())))
        return rc;
    return d2h(c, out, c->ws_out.p, sizeof(double) * B);
}

int obtg_accel_obj(obtg_ctx* c, const double* Y, const double* tf, int B, double* out)
{
    return host_deriv_obj(c, Y, tf, B, 2, out);
}

int obtg_jerk_obj(obtg_ctx* c, const double* Y, const double* tf, int B, double* out)
{
    return host_deriv_obj(c, Y, tf, B, 3, out);
}

// ------------------------------------------------------------------ instrumentation
int obtg_set_profiling(obtg_ctx* c, int on)
{
    if (!check_ctx(c)) return OBTG_ERR_ARG;
    if ((on & OBTG_PROFILE_ONLY_FLAG) && (on & 0xff) >= OBTG_K_COUNT) return OBTG_ERR_ARG;
    OBTG_HIP(c, hipStreamSynchronize(c->stream));
    flush_pending_events(c);
    c->profiling = on != 0;
    c->profile_mask = (on & OBTG_PROFILE_ONLY_FLAG) ? (1u << (on & 0xff)) : ~0u;
    return OBTG_OK;
}

int obtg_set_profile_period(obtg_ctx* c, int every)
{
    if (!check_ctx(c) || every < 1) return OBTG_ERR_ARG;
    c->profile_period = every;
    for (auto& n : c->profile_seen) n = 0;
    return OBTG_OK;
}

int obtg_kernel_stats(obtg_ctx* c, int kernel_id, double* total_ms, long long* launches)
{
    if (!check_ctx(c) || kernel_id < 0 || kernel_id >= OBTG_K_COUNT) return OBTG_ERR_ARG;
    flush_pending_events(c);
    if (total_ms) *total_ms = c->stats[kernel_id].ms;
    if (launches) *launches = c->stats[kernel_id].launches;
    return OBTG_OK;
}

int obtg_reset_kernel_stats(obtg_ctx* c)
{
    if (!check_ctx(c)) return OBTG_ERR_ARG;
    flush_pending_events(c);
    for (auto& s : c->stats) s = KernelStat{};
    return OBTG_OK;
}

const char* obtg_kernel_name(int id)
{
    switch (id) {
        case OBTG_K_TEMPORAL_SEP: return "temporal_sep";
        case OBTG_K_SPEED: return "speed";
        case OBTG_K_ANG_RATE: return "ang_rate";
        case OBTG_K_GJK: return "gjk";
        case OBTG_K_MIN_DIST: return "min_dist";
        case OBTG_K_FD_BATCH: return "fd_batch";
        case OBTG_K_BERN: return "bern";
        case OBTG_K_PAIR_SWEEP: return "pair_sweep";
        default: return "?";
    }
}

}  // extern "C"
