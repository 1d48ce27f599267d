// Internal declarations shared by the translation units of libobtg_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/obtg.h"

namespace obtg {

// ---- Specialised control-point counts (degree + 1).  The batch kernels are templates on the count; these lists are the
// ONLY place that says which counts are instantiated -- every dispatch below expands one of them, so a new degree is one
// entry here (round 5: 9 = the degree-8 drivers, Examples/DubinsCarTimeOptimal.py:72, DubinsCarExample2.py:83).
//   OBTG_NC_SEP : separation / speed rows (k_normsq_elev, k_tsep_fd, k_one_vs_many; 2-D and 3-D), planar hull sweep
//   OBTG_NC_DYN : angular rate (k_dynamics*), 3-D sweeps, the one-launch steps (k_pair_sweep*, k_step_fd_structured)
//   OBTG_NC_ELEV: DEG_ELEV > 0 separation + dynamics in one launch (k_sep_dynamics_elev) and the elevated structured step
#define OBTG_NC_ELEV(X) X(4) X(6) X(8) X(9) X(11)
#define OBTG_NC_DYN(X)  OBTG_NC_ELEV(X) X(16)
#define OBTG_NC_SEP(X)  OBTG_NC_DYN(X) X(21)
#define OBTG_NC_EQ_(N) || nc == N
static inline bool nc_in_sep(int nc)  { return false OBTG_NC_SEP(OBTG_NC_EQ_); }
static inline bool nc_in_dyn(int nc)  { return false OBTG_NC_DYN(OBTG_NC_EQ_); }
static inline bool nc_in_elev(int nc) { return false OBTG_NC_ELEV(OBTG_NC_EQ_); }

constexpr int kWave = 64;
constexpr int kNeedBatch = 1077;      // internal launcher result: "this kernel needs the finite-difference batch in memory"
constexpr int kMaxGenericLen = 1024;  // longest Bernstein coefficient vector of the generic kernels

// ---------------------------------------------------------------- host-side tables (tables.cpp)
double binom(int n, int k);                         // scipy.special.binom on integers (0 outside range)
std::vector<double> binom_row(int n);               // C(n, 0..n)
std::vector<double> folded_product_weights(int n, int dim);  // [2n+1][n+1], see bern_kernels.hip
std::vector<double> elev_table_frag(int L_in, int R);   // the same matrix as v_mfma_f64_16x16x4 B fragments: [NT][KS][64]
std::vector<double> elev_table_T_ld(int L_in, int R);   // dense, transposed, zero padded: [L_in+R][L_in] (bit-equal to the fragments)
struct DdTables { int wn, w2n, w22n, ratio, row4, sc4; };     // offsets (doubles) of angrate_dd_tables' parts
DdTables angrate_dd_tables(int n, int R, std::vector<double>& t);
std::vector<double> elev_conv_tables(int L_in, int R);  // scale | padded C(R,.) | 1/C(N+R,.)
std::vector<double> elev_conv_padded(int L_in, int R, int extra, bool normalise, bool with_inv);

// ---------------------------------------------------------------- device buffers
// Device buffer that grows.  A staging buffer of the host entry points (`io`) serves requests of up to kZeroCopyBytes
// from mapped pinned HOST memory instead: the kernel reads its few hundred bytes of control points and writes its
// results (up to one 64-vehicle row of separation rows, 339 KB) across PCIe itself, and a one-row SLSQP callback is memcpy + ONE launch + synchronize + memcpy instead
// of two DMA transfers around the launch (tools/latency_probe.py: 30 -> 15 us per call at Example1's size).
constexpr size_t kZeroCopyBytes = 512u << 10;
struct DevBuf {
    void* p = nullptr;          // the address kernels use for the current request
    size_t cap = 0;             // capacity behind p
    bool io = false;            // a staging buffer of the host entry points (set once, at context creation)
    bool on_host = false;       // p is the mapped host block
    void* dev = nullptr;        // device allocation (grows, never shrinks)
    size_t dev_cap = 0;
    void* host = nullptr;       // CPU address of the mapped block (kZeroCopyBytes), host_dev its device address
    void* host_dev = nullptr;
    bool host_failed = false;
    // returns OBTG_* code.  zero_copy: this request may be served from the mapped host block -- only for data a kernel
    // touches once (control points staged to LDS or registers, results written once)
    int reserve(size_t bytes, bool zero_copy = false);
    void release();
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct KernelStat {
    double ms = 0.0;
    long long launches = 0;
};

}  // namespace obtg

struct obtg_ctx {
    int device = 0;
    int n_cus = 256;          // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    int n_veh = 0, dim = 0, deg = 0, R = 0, n_obs = 0;
    int n_obj = 0, n_pairs = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string last_error;

    // resident tables
    obtg::DevBuf d_pairs;     // int2[n_pairs]  (i, j) lexicographic, i < j < n_obj
    obtg::DevBuf d_obs;       // double[n_obs][dim]
    obtg::DevBuf d_w2;        // folded product weights for (deg, dim)
    obtg::DevBuf d_Tt;        // elevation (2*deg -> 2*deg+R) as convolution tables (elev_conv_tables)
    obtg::DevBuf d_Td;        // the same elevation as a dense transposed matrix (elev_table_T_ld): rows for the lane-per-item chains
    obtg::DevBuf d_Tf;        // ... and as matrix-instruction B fragments (elev_table_frag): the batch kernels
    obtg::DevBuf d_ang_w2n, d_ang_w22n, d_ang_wn;  // angular-rate fast path weights
    obtg::DevBuf d_ang_T4;    // angular rate, R > 0: elevation 4*deg -> 4*(deg+R) as a scaled convolution (elev_conv_padded)
    obtg::DevBuf d_ang_cv2;   // the same for the speed rows, 2*deg -> 2*deg+R, with the 1/C(2n+R, k) row
    bool ang_elevate_first = false;   // true: the reference's order (elevate, then products at degree n+R; generic kernel)
    bool ang_exact = false;           // obtg_ctx_set_ang_rate_order(2): the default order, then the rows of near-stop vehicles again in
                                      // double-double (k_angrate_dd); tables and the flag list below, made on first use
    obtg::DevBuf d_ang_dd, d_ang_flags;
    obtg::DdTables ang_dd_off{};
    int ang_dd_R = -1;
    // obtg_ctx_set_second_speed_bound: every dynamics pass that writes speed rows also writes the other bound's rows
    struct { double bound = 0.0; int is_max = 0; double* d_out = nullptr; } speed2;
    std::vector<int> h_pairs; // host copy of the pair table (2 ints per pair)
    std::vector<int> h_tiles; // row-window tiles of the current (pair_begin, pair_count)
    obtg::DevBuf d_tiles;
    int tiles_begin = -1, tiles_count = -1;
    std::vector<double> h_binrows;
    obtg::DevBuf d_binrows;   // concatenated binomial rows for the generic kernels
    std::vector<int> binrow_off;   // offset of row C(n, .) inside d_binrows, -1 if absent
    int tables_R = -1;

    // swarm GJK state
    obtg::DevBuf d_poly_pts;  // SoA per polygon: x[K], y[K], z[K]
    obtg::DevBuf d_poly_off;  // int[n_poly+1]
    int n_poly = 0, n_poly_pts = 0, max_poly_K = 0;
    bool polys_planar = true;   // every registered polygon vertex has z == 0
    bool fd_dedup = false;      // reuse row 0's GJK results for bit-identical hull pairs
    obtg::DevBuf d_hp_a, d_hp_b;  // hull pair list
    obtg::DevBuf d_vp_off, d_vp_idx;   // per vehicle: the positions of the hull pairs that contain it (CSR; structured FD step)
    std::vector<int> h_hp_a, h_hp_b;   // host copy (tile-major chunking of large rows)
    obtg::DevBuf d_tile_chunk_off, d_tile_order, d_tile_pslots, d_tile_cobj_off, d_tile_cobjs, d_tile_ij;
    bool tile_ts_ok = false;  // every chunk's tile origin is recorded and the hull pair list holds every vehicle pair: the tiled
                              // sweep can write the temporal-separation rows of its tiles (one-launch pair sweep of large rows)
    int tile_a = 8;           // tile height the chunks were cut for
    bool tile_valid = false;
    int tile_n_chunks = 0, tile_max_objs = 0, tile_max_pairs = 0;
    int n_hull_pairs = 0;
    bool hull_pairs_set = false;  // obtg_ctx_set_hull_pairs since the last obtg_ctx_set_polygons
    obtg::DevBuf d_gjk_len[2];   // per-pair support-scan counts of the last planar sweep (scheduling history)
    bool gjk_history = true;
    int gjk_len_cur = 0, gjk_len_rows = 0;   // buffer holding the latest counts, and how many rows of them

    // Virtual finite-difference batch (obtg_*_fd_dev): rows are formed on the fly from ONE row of control points,
    // row b >= 1 = Y0 with its (b-1)-th free control point advanced by h (exactly obtg_fd_batch_dev's rows).
    // `fd` is set for the duration of ONE launcher call; launchers whose kernel cannot form the rows return
    // obtg::kNeedBatch before launching anything and the caller materialises the batch (once per view) instead.
    struct FdView { const double* Y0 = nullptr; double h = 0.0; int fixed = 0; int row0 = 0; } fd;       // row0: batch row of local row 0
    struct { const double* Y0 = nullptr; double h = 0.0; int fixed = 0; int B = 0; int row0 = 0; bool materialised = false; } view;   // obtg_fd_view_begin[_rows] .. _end
    obtg::DevBuf ws_fd;                   // the view's batch, written only when some kernel needs it

    // scratch for host-buffer entry points
    obtg::DevBuf ws_in, ws_in2, ws_out, ws_misc[8];
    // obtg_min_dist: node counts of the previous evaluation of the SAME pair list (signature = count + hash of the lists):
    // the next evaluation hands its pairs to the worker waves in descending order of them (k_min_dist_wave)
    // (up to four lists, least recently used dropped: a driver alternates the constraint's list with its Jacobian's longer one)
    struct MdHist { unsigned long long sig; std::vector<int> nodes; };
    std::vector<MdHist> md_hist;
    void* ring = nullptr;                 // pinned staging ring for pageable caller buffers (capi.cpp h2d / d2h)
    hipEvent_t ring_ev[4] = {};
    bool ring_pending[4] = {};            // slot's event recorded and not waited for yet
    int ring_next = 0;                    // next slot of the rotation

    // instrumentation
    bool profiling = false;
    int profile_period = 1;               // events on every n-th eligible launch of a kernel id
    long long profile_seen[OBTG_K_COUNT] = {};
    unsigned profile_mask = ~0u;          // kernel ids that get events while profiling
    std::vector<hipEvent_t> event_pool;   // recycled events
    obtg::KernelStat stats[OBTG_K_COUNT];
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending_events;
};

namespace obtg {

int set_error(obtg_ctx* c, hipError_t e, const char* where);
#define OBTG_HIP(c, call)                                                   \
    do {                                                                    \
        hipError_t e__ = (call);                                            \
        if (e__ != hipSuccess) return ::obtg::set_error((c), e__, #call);   \
    } while (0)

// bracket a launch with events when profiling.  ext = true: the events are NOT recorded on the stream; the launch itself
// carries them (launch_timed -> hipExtLaunchKernelGGL: the dispatch's own start / stop timestamps, no barrier packets
// around the kernel, so events on every launch do not stretch the step).
struct ScopedKernelTimer {
    obtg_ctx* c;
    int id;
    bool ext;
    hipEvent_t a = nullptr, b = nullptr;
    ScopedKernelTimer(obtg_ctx* c_, int id_, bool ext_ = false);
    ~ScopedKernelTimer();
};
void flush_pending_events(obtg_ctx* c);

int ensure_tables(obtg_ctx* c);
int binrow_offset(obtg_ctx* c, int n);  // ensures row C(n,.) is resident; returns offset (doubles)

// ---------------------------------------------------------------- launchers (bern_kernels.hip)
// comm.cpp: this rank's block of the per-pair minima (contiguous balanced partition of the lexicographic pair list), ONE
// RCCL all-gather on the context's stream, rows [B][P] on every rank
int comm_gather_pair_minima(::obtg_comm* m, obtg_ctx* c, const double* dY, int B, double max_sep, double* d_min_all);
int launch_temporal_sep(obtg_ctx* c, const double* dY, int B, double max_sep, int pair_begin,
                        int pair_count, bool min_only, double* d_out, int sel_k = 0, int* d_sel_idx = nullptr);
struct NsParams;
// What launch_gjk_swarm may fold into its 3-D sweep launch (k_pair_sweep_3d): the row's temporal-separation block and,
// when d_out_speed is set, its speed rows.  did_* report what the launch took over.
struct SweepFold {
    double max_sep = 0.0;
    double* d_out_sep = nullptr;
    const double* d_tf = nullptr;
    double speed_bound = 0.0;
    int speed_is_max = 1;
    double* d_out_speed = nullptr;
    double max_rate = 0.0;          // planar rows: with d_out_ang the speed / angular-rate groups may join the sweep's grid
    double* d_out_ang = nullptr;
    bool did_sep = false, did_speed = false, did_dynamics = false;
};
int plan_temporal_sep(obtg_ctx* c, const double* dY, int B, int pair_begin, int pair_count, double* d_out, NsParams& p);
size_t temporal_sep_lds_bytes(const obtg_ctx* c, NsParams& p, size_t budget);
// speed (optional): the speed rows of the same batch, folded into the launch where a kernel does that (3-D sweeps);
// speed->did_speed tells the caller whether they still have to be launched
int launch_pair_sweep(obtg_ctx* c, const double* dY, int B, double max_sep, double* d_out_sep, int max_iter,
                      int md_cap, int* d_flag, double* d_p1, double* d_p2, double* d_dist, int* d_nsup, int* d_status,
                      SweepFold* speed = nullptr);
int launch_step_fd_structured(obtg_ctx* c, int B, double max_sep, double* d_out_sep, int max_iter, int md_cap, int* d_flag,
                              double* d_p1, double* d_p2, double* d_dist, int* d_nsup, int* d_status, SweepFold* speed);
int launch_temporal_sep_fd(obtg_ctx* c, const double* dY0, int n_pert, const int* d_prow, const int* d_pcol,
                           const double* d_pval, double max_sep, double* d_out, int min_only = 0, int fd_row0 = 0,
                           int fd_fixed = 0, double fd_h = 0.0);
int launch_one_vs_many_min(obtg_ctx* c, const double* d_one, int B, const double* d_many, int K, double max_sep, double* d_out);
int launch_speed(obtg_ctx* c, const double* dY, const double* d_tf, int B, double bound, int is_max,
                 double* d_out);
int launch_ang_rate(obtg_ctx* c, const double* dY, const double* d_tf, int B, double max_rate,
                    double* d_out);
// DEG_ELEV > 0, planar: separation rows + speed / angular-rate rows in one launch; OBTG_ERR_UNSUPPORTED = not this shape
int launch_sep_dynamics_elev(obtg_ctx* c, const double* dY, int B, double max_sep, double* d_out_sep, const SweepFold& f);
int launch_dynamics(obtg_ctx* c, const double* dY, const double* d_tf, int B, double bound, int is_max,
                    double max_rate, double* d_out_speed, double* d_out_ang);
int launch_fd_batch(obtg_ctx* c, const double* dY0, int n_fixed_cols, double h, int B, double* dY, int row0 = 0);   // rows row0 .. row0 + B - 1 of the batch
bool dynamics_fd_on_the_fly(const obtg_ctx* c, bool want_ang);
int ang_rate_order_in_effect(obtg_ctx* c);
bool bernstein_fd_on_the_fly(const obtg_ctx* c);      // the separate temporal-separation / speed kernels form a view's rows themselves
bool pair_sweep_is_one_launch(const obtg_ctx* c);
int launch_bern_elev(obtg_ctx* c, const double* d_in, int rows, int n, int R, double* d_out);
int launch_bern_diff(obtg_ctx* c, const double* d_in, int rows, int n, double T, double* d_out);
int launch_bern_split(obtg_ctx* c, const double* d_in, int rows, int n, double z, double* d_left, double* d_right);
int launch_bern_eval(obtg_ctx* c, const double* d_cpts, int rows, int n, const double* d_tau, int n_tau, double t0, double tf,
                     double* d_out);
int launch_bern_mul(obtg_ctx* c, const double* d_a, const double* d_b, int rows, int m, int n,
                    double* d_out);
int launch_bern_normsq(obtg_ctx* c, const double* d_x, int d, int n, double* d_out);
int launch_euclidean_obj(obtg_ctx* c, const double* dY, int B, double* d_out);
int launch_deriv_energy_obj(obtg_ctx* c, const double* dY, const double* d_tf, double tf0, int B, int order, double* d_out);

// ---------------------------------------------------------------- launchers (gjk_kernels.hip)
int launch_gjk_pairs(obtg_ctx* c, const double* d_soa, const int* d_off, const int* d_pa,
                     const int* d_pb, int n_pairs, int max_iter, int md_cap, int* d_flag,
                     double* d_p1, double* d_p2, double* d_dist, short* d_trace, int trace_cap,
                     int* d_nsup, int* d_status, bool planar);
int launch_gjk_swarm(obtg_ctx* c, const double* dY, int B, int max_iter, int md_cap, int* d_flag,
                     double* d_p1, double* d_p2, double* d_dist, int* d_nsup, int* d_status, SweepFold* fold = nullptr);
int launch_min_dist(obtg_ctx* c, const double* d_curves, int K, const int* d_pa, const int* d_pb,
                    int n_pairs, double eps, int max_iter, int md_cap, int max_depth, int max_nodes,
                    double* d_stack, double* d_res, int* d_info, const int* d_order = nullptr, int* d_queue = nullptr,
                    bool planar = false);      // planar: every control point of every curve has z == 0 (the caller has looked)
int launch_min_dist_robust(obtg_ctx* c, const double* d_curves, int K, const int* d_pa, const int* d_pb, int n_pairs,
                           double eps, int max_nodes, int max_level, int cap, double* d_frontier, double* d_res, int* d_info);
int launch_min_dist2poly(obtg_ctx* c, const double* d_curves, int K, const double* d_soa,
                         const int* d_off, const int* d_pc, const int* d_pp, int n_pairs, double eps,
                         int max_iter, int md_cap, int max_depth, int max_nodes, double* d_stack,
                         double* d_res, int* d_info, int max_poly_K, bool planar = false);     // planar: curves AND polygons have z == 0 throughout
int launch_gjk_true_pairs(obtg_ctx* c, const double* d_soa, const int* d_off, const int* d_pa, const int* d_pb, int n_pairs,
                          double eps, int max_iter, int* d_flag, double* d_p1, double* d_p2, double* d_dist, double* d_lower,
                          int* d_iters, int* d_status);
int launch_min_dist2poly_robust(obtg_ctx* c, const double* d_curves, int K, const double* d_soa, const int* d_off,
                                const int* d_pc, const int* d_pp, int n_pairs, double eps, int max_nodes, int max_level,
                                int cap, int max_poly_K, double* d_frontier, double* d_res, int* d_info);
// x**2 as Python forms it for the constraint offsets (optimization.py:343, 384, 422, 459: maxSep**2, minSpeed**2, maxSpeed**2,
// maxAngRate**2 on Python floats): the host libm's pow(x, 2.0), which is not always x * x (tables.cpp)
double square_as_python(double x);
double cube_as_python(double x);       // x**3 likewise (bezier.py:1468: eps**3)
size_t min_dist_stack_doubles(const obtg_ctx* c, int K, int max_depth, int n_pairs);   // whole launch
size_t min_dist2poly_stack_doubles(int K, int max_depth);

}  // namespace obtg
