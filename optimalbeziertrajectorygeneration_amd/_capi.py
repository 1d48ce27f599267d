"""ctypes binding of libobtg_hip.so (include/obtg.h).

This is the ONLY compute path of the package: there is no CPU fallback.  Importing
works without a GPU (so that host logic can be tested), but creating a `Context`
raises `RuntimeError` when the library is missing or no gfx950 device is usable.
"""
import ctypes as C
import os
import threading
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OBTG_LIB") or os.path.join(HERE, "libobtg_hip.so")   # OBTG_LIB: experimental builds (tools/build_variant.sh)

OK = 0
ST_OK, ST_MD_CAP, ST_MAXITER, ST_CYCLE = 0, 1, 2, 3
MD_OK, MD_NODE_CAP, MD_DEPTH_CAP, MD_GJK_CAP = 0, 1, 2, 3
K_TEMPORAL_SEP, K_SPEED, K_ANG_RATE, K_GJK, K_MIN_DIST, K_FD_BATCH, K_BERN, K_PAIR_SWEEP, K_COUNT = range(9)

_lib = None

_vp = C.c_void_p
_i = C.c_int
_d = C.c_double

# name -> (restype, argtypes); mirrors include/obtg.h one to one
_SIGNATURES = {
    "obtg_strerror": (C.c_char_p, [_i]),
    "obtg_last_error": (C.c_char_p, [_vp]),
    "obtg_abi_version": (_i, []),
    "obtg_source_hash": (C.c_char_p, [C.c_char_p]),
    "obtg_libm_pow_matches": (_i, []),
    "obtg_fast_kernels": (_i, [_i, _i]),
    "obtg_device_count": (_i, []),
    "obtg_abi_symbols": (_vp, []),
    "obtg_host_alloc": (_i, [C.c_size_t, C.POINTER(_vp)]),
    "obtg_host_free": (_i, [_vp]),
    "obtg_ctx_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _vp, _i]),
    "obtg_ctx_destroy": (None, [_vp]),
    "obtg_ctx_set_stream": (_i, [_vp, _vp]),
    "obtg_ctx_use_own_stream": (_i, [_vp]),
    "obtg_ctx_set_deg_elev": (_i, [_vp, _i]),
    "obtg_ctx_set_ang_rate_order": (_i, [_vp, _i]),
    "obtg_ctx_ang_rate_order_in_effect": (_i, [_vp]),
    "obtg_ctx_set_second_speed_bound": (_i, [_vp, _d, _i, _vp]),
    "obtg_constraint_sweep_fd_structured_dev": (_i, [_vp, _vp, _i, _d, _vp, _i, _d, _vp, _d, _i, _d, _vp, _vp, _i, _i, _vp, _vp,
                                                     _vp, _vp, _vp, _vp]),
    "obtg_constraint_sweep_fd_structured_rows_dev": (_i, [_vp, _vp, _i, _d, _i, _vp, _i, _d, _vp, _d, _i, _d, _vp, _vp, _i, _i, _vp,
                                                          _vp, _vp, _vp, _vp, _vp]),
    "obtg_one_vs_many_min": (_i, [_vp, _vp, _i, _vp, _i, _d, _vp]),
    "obtg_one_vs_many_min_dev": (_i, [_vp, _vp, _i, _vp, _i, _d, _vp]),
    "obtg_sync": (_i, [_vp]),
    "obtg_len_temporal_sep": (_i, [_vp]),
    "obtg_len_speed": (_i, [_vp]),
    "obtg_len_ang_rate": (_i, [_vp]),
    "obtg_num_pairs": (_i, [_vp]),
    "obtg_temporal_sep": (_i, [_vp, _vp, _i, _d, _vp]),
    "obtg_speed": (_i, [_vp, _vp, _vp, _i, _d, _i, _vp]),
    "obtg_ang_rate": (_i, [_vp, _vp, _vp, _i, _d, _vp]),
    "obtg_temporal_sep_min": (_i, [_vp, _vp, _i, _d, _vp]),
    "obtg_temporal_sep_active": (_i, [_vp, _vp, _i, _d, _i, _vp, _vp]),
    "obtg_temporal_sep_active_dev": (_i, [_vp, _vp, _i, _d, _i, _i, _i, _vp, _vp]),
    "obtg_temporal_sep_min_gather_dev": (_i, [_vp, _vp, _vp, _i, _d, _vp]),
    "obtg_temporal_sep_fd_min_rows_dev": (_i, [_vp, _vp, _i, _d, _i, _i, _d, _vp]),
    "obtg_comm_unique_id": (_i, [_vp]),
    "obtg_comm_create": (_i, [C.POINTER(_vp), _i, _i, _vp, _i]),
    "obtg_comm_destroy": (None, [_vp]),
    "obtg_comm_size": (_i, [_vp]),
    "obtg_comm_rank": (_i, [_vp]),
    "obtg_comm_last_error": (C.c_char_p, [_vp]),
    "obtg_comm_all_gather_dev": (_i, [_vp, _vp, _vp, _vp, C.c_size_t]),
    "obtg_pair_block": (_i, [_vp, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "obtg_unpack_pair_blocks_dev": (_i, [_vp, _vp, _i, _i, _vp]),
    "obtg_temporal_sep_min_range": (_i, [_vp, _vp, _i, _d, _i, _i, _vp]),
    "obtg_temporal_sep_fd": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _d, _vp]),
    "obtg_temporal_sep_fd_dev": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _d, _vp]),
    "obtg_temporal_sep_dev": (_i, [_vp, _vp, _i, _d, _i, _i, _vp]),
    "obtg_temporal_sep_min_dev": (_i, [_vp, _vp, _i, _d, _i, _i, _vp]),
    "obtg_speed_dev": (_i, [_vp, _vp, _vp, _i, _d, _i, _vp]),
    "obtg_ang_rate_dev": (_i, [_vp, _vp, _vp, _i, _d, _vp]),
    "obtg_dynamics_dev": (_i, [_vp, _vp, _vp, _i, _d, _i, _d, _vp, _vp]),
    "obtg_fd_batch_dev": (_i, [_vp, _vp, _i, _d, _i, _vp]),
    "obtg_fd_view_begin": (_i, [_vp, _vp, _i, _d, _i]),
    "obtg_fd_view_begin_rows": (_i, [_vp, _vp, _i, _d, _i, _i]),
    "obtg_fd_view_end": (_i, [_vp]),
    "obtg_fd_forms_on_the_fly": (_i, [_vp]),
    "obtg_pair_sweep_fd_dev": (_i, [_vp, _vp, _i, _d, _i, _d, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "obtg_dynamics_fd_dev": (_i, [_vp, _vp, _i, _d, _vp, _i, _d, _i, _d, _vp, _vp]),
    "obtg_gjk_pairs": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "obtg_ctx_set_polygons": (_i, [_vp, _vp, _i, _vp, _i]),
    "obtg_ctx_set_hull_pairs": (_i, [_vp, _vp, _vp, _i]),
    "obtg_ctx_set_fd_dedup": (_i, [_vp, _i]),
    "obtg_ctx_set_gjk_history": (_i, [_vp, _i]),
    "obtg_pair_sweep_dev": (_i, [_vp, _vp, _i, _d, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "obtg_gjk_swarm_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "obtg_constraint_sweep_dev": (_i, [_vp, _vp, _vp, _i, _d, _vp, _d, _i, _d, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "obtg_gjk_swarm": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "obtg_min_dist": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _d, _i, _i, _i, _i, _vp, _vp, _vp]),
    "obtg_min_dist_robust": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _d, _i, _vp, _vp, _vp]),
    "obtg_min_dist2poly": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _d, _i, _i, _i, _i, _vp, _vp, _vp]),
    "obtg_min_dist2poly_robust": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _d, _i, _vp, _vp, _vp]),
    "obtg_gjk_true_pairs": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i, _d, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "obtg_bern_elev": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "obtg_bern_diff": (_i, [_vp, _vp, _i, _i, _d, _vp]),
    "obtg_bern_mul": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "obtg_bern_normsq": (_i, [_vp, _vp, _i, _i, _vp]),
    "obtg_bern_split": (_i, [_vp, _vp, _i, _i, _d, _vp, _vp]),
    "obtg_bern_eval": (_i, [_vp, _vp, _i, _i, _vp, _i, _d, _d, _vp]),
    "obtg_euclidean_obj": (_i, [_vp, _vp, _i, _vp]),
    "obtg_accel_obj": (_i, [_vp, _vp, _vp, _i, _vp]),
    "obtg_jerk_obj": (_i, [_vp, _vp, _vp, _i, _vp]),
    "obtg_set_profiling": (_i, [_vp, _i]),
    "obtg_set_profile_period": (_i, [_vp, _i]),
    "obtg_kernel_stats": (_i, [_vp, _i, C.POINTER(_d), C.POINTER(C.c_longlong)]),
    "obtg_reset_kernel_stats": (_i, [_vp]),
    "obtg_kernel_name": (C.c_char_p, [_i]),
}


def abi_symbol_names():
    """The symbols include/obtg.h declares (used by the CPU-side load test)."""
    return sorted(_SIGNATURES)


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so
    (soname libamdhip64.so.7, the same soname /opt/rocm's carries).  If libobtg_hip.so pulled in
    the system copy and torch later loaded its bundled copy, the process would hold two HIP
    runtimes and the second would see no GPU.  Loading torch's copy first (by path, without
    importing torch) makes both libobtg_hip.so (NEEDED libamdhip64.so.7, matched by soname) and
    a later `import torch` (same file) share it.  Without torch installed the system runtime
    is used."""
    if os.environ.get("OBTG_USE_SYSTEM_HIP") == "1":
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    try:
        return C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError:
        return None


_hip_runtime = None


def load():
    """dlopen the in-tree library; loud failure when it has not been built."""
    global _lib, _hip_runtime
    if _lib is not None:
        return _lib
    _hip_runtime = _preload_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libobtg_hip.so is missing (%s). Build it with "
            "`python -m optimalbeziertrajectorygeneration_amd.build`; there is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here == ABI drift between header and library
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def device_count():
    return load().obtg_device_count()


def abi_version():
    return load().obtg_abi_version()


def source_hash(unit="all"):
    """obtg_source_hash: which sources the loaded library was built from (16 hex digits; None for an unknown unit name)."""
    h = load().obtg_source_hash(unit.encode() if unit is not None else None)
    return h.decode() if h is not None else None


def libm_pow_matches():
    """obtg_libm_pow_matches: does this host's pow(x, 2.0) round as the device's restatement of glibc 2.35's does."""
    return bool(load().obtg_libm_pow_matches())


def fast_kernels(dim, deg):
    """obtg_fast_kernels: bit 0 separation / speed rows, bit 1 angular rate + one-launch steps, bit 2 the DEG_ELEV > 0 forms."""
    return load().obtg_fast_kernels(int(dim), int(deg))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _prefer_torch_rccl():
    """One RCCL per process, as for the HIP runtime: PyTorch bundles a librccl.so; when it is there and the caller has not
    chosen a file, comm.cpp is told to open that one (OBTG_RCCL_LIB) so that obtg_comm_* and torch.distributed share it."""
    if os.environ.get("OBTG_RCCL_LIB"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
        if os.path.exists(cand):
            os.environ["OBTG_RCCL_LIB"] = cand


class Comm(object):
    """obtg_comm: this process's membership of a group of n_ranks processes, one per GPU (RCCL).  Rank 0 creates the id
    (`Comm.unique_id()`, 128 bytes) and hands it to the others by any means; every rank then constructs its Comm --
    collectively."""

    @staticmethod
    def unique_id():
        _prefer_torch_rccl()
        buf = (C.c_ubyte * 128)()
        rc = load().obtg_comm_unique_id(buf)
        if rc:
            raise ObtgError("obtg_comm_unique_id: %s" % load().obtg_strerror(rc).decode(), rc)
        return bytes(buf)

    def __init__(self, n_ranks, rank, unique_id, device=0):
        _prefer_torch_rccl()
        self._lib = load()
        self._h = _vp()
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        rc = self._lib.obtg_comm_create(C.byref(self._h), int(n_ranks), int(rank), buf, int(device))
        if rc:
            raise ObtgError("obtg_comm_create: %s" % self._lib.obtg_strerror(rc).decode(), rc)
        self.n_ranks, self.rank = int(n_ranks), int(rank)

    def all_gather_dev(self, ctx, d_send, d_recv, bytes_per_rank):
        rc = self._lib.obtg_comm_all_gather_dev(self._h, ctx._h, _vp(d_send), _vp(d_recv), int(bytes_per_rank))
        if rc:
            raise ObtgError("obtg_comm_all_gather_dev: %s (%s)" % (self._lib.obtg_strerror(rc).decode(),
                                                                     (self._lib.obtg_comm_last_error(self._h) or b"").decode()), rc)

    def close(self):
        if self._h:
            self._lib.obtg_comm_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ObtgError(RuntimeError):
    """A non-zero return of the C ABI; `.code` is the OBTG_ERR_* value (-5: the shape has no kernel of that kind)."""

    def __init__(self, msg, code=None):
        RuntimeError.__init__(self, msg)
        self.code = code


ERR_UNSUPPORTED = -5


class _PinnedPool(object):
    """Page-locked host blocks for the large result arrays of the host-buffer entry points.

    The device writes such an array by DMA at PCIe rate; a pageable NumPy array would be staged chunk by chunk
    (include/obtg.h, obtg_host_alloc).  Blocks are recycled: pinning costs milliseconds per 100 MB, an SLSQP
    driver asks for the same shapes on every iteration.  An array keeps its block alive through its buffer
    object; when the last view dies the block returns to the pool.  At most `cap` bytes stay cached -- 1 GiB by
    default (two Jacobian-sized result arrays of the 64-vehicle configuration), OBTG_PINNED_CACHE_MB overrides it --
    and `trim()` (module function `pinned_trim`) hands every cached block back to the driver."""

    MIN_BYTES = 1 << 20

    def __init__(self, cap=None):
        if cap is None:
            cap = int(os.environ.get("OBTG_PINNED_CACHE_MB", "1024")) << 20
        self.cap, self.cached, self.free = cap, 0, {}
        self._lock = threading.RLock()       # callers on several threads; re-entrant: a finaliser may run inside empty()

    def trim(self, keep_bytes=0):
        """Free cached blocks (largest first) until at most keep_bytes stay cached; returns the bytes released."""
        released = 0
        with self._lock:
            for size in sorted(self.free, reverse=True):
                blocks = self.free[size]
                while blocks and self.cached > keep_bytes:
                    ptr = blocks.pop()
                    self.cached -= size
                    released += size
                    if _lib is not None:
                        _lib.obtg_host_free(_vp(ptr))
            for size in [k for k, v in self.free.items() if not v]:
                del self.free[size]
        return released

    def _release(self, size, ptr):
        with self._lock:
            if self.cached + size <= self.cap:
                self.free.setdefault(size, []).append(ptr)
                self.cached += size
                return
        if _lib is not None:
            _lib.obtg_host_free(_vp(ptr))

    def empty(self, shape, dtype=np.float64):
        dtype = np.dtype(dtype)
        count = int(np.prod(shape))
        nbytes = count * dtype.itemsize
        if nbytes < self.MIN_BYTES:
            return np.empty(shape, dtype)
        size = 1 << (nbytes - 1).bit_length()
        if size > (1 << 28):                      # above 256 MB: 64 MB granularity instead of powers of two
            size = -(-nbytes // (64 << 20)) * (64 << 20)
        ptr = None
        with self._lock:
            blocks = self.free.get(size)
            if blocks:
                ptr = blocks.pop()
                self.cached -= size
        if ptr is None:
            h = _vp()
            if load().obtg_host_alloc(size, C.byref(h)) != OK or not h.value:
                return np.empty(shape, dtype)     # no pinned memory left: a pageable array still works (staged)
            ptr = h.value
        buf = (C.c_char * nbytes).from_address(ptr)
        weakref.finalize(buf, self._release, size, ptr)
        return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)


_pinned = _PinnedPool()


def pinned_trim(keep_bytes=0):
    """Release the page-locked blocks the result-array pool has cached (all of them by default)."""
    return _pinned.trim(keep_bytes)


def pinned_empty(shape, dtype=np.float64):
    """An uninitialised array in page-locked host memory (falls back to np.empty for small sizes)."""
    return _pinned.empty(shape, dtype)


def torch_stream():
    """HIP handle of torch's current stream, for Context.set_stream (0 for torch's default stream, which IS the null
    stream: obtg_ctx_set_stream takes the handle as given, so the library's launches are ordered with torch's own)."""
    import torch
    return torch.cuda.current_stream().cuda_stream


class Context(object):
    """One problem shape on one MI355X: wraps obtg_ctx."""

    def __init__(self, n_veh, dim, deg, deg_elev=0, point_obs=None, device=0):
        self._lib = load()
        self._h = _vp()
        if self._lib.obtg_device_count() <= 0:
            raise ObtgError("no usable gfx950 (MI355X) device: the HIP path is the only compute path")
        obs = None
        n_obs = 0
        if point_obs is not None and len(point_obs) > 0:
            obs = _f64(point_obs).reshape(-1, dim)
            n_obs = obs.shape[0]
        rc = self._lib.obtg_ctx_create(C.byref(self._h), n_veh, dim, deg, deg_elev, n_obs, _ptr(obs), device)
        if rc != OK:
            self._h = _vp()
            raise ObtgError("obtg_ctx_create: " + self._lib.obtg_strerror(rc).decode())
        self.n_veh, self.dim, self.deg, self.deg_elev, self.n_obs = n_veh, dim, deg, deg_elev, n_obs
        self.device = device
        self.n_poly = 0
        self.n_hull_pairs = None      # no hull pair list registered yet

    # -- plumbing
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.obtg_ctx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != OK:
            msg = self._lib.obtg_strerror(rc).decode()
            extra = self._lib.obtg_last_error(self._h)
            if extra:
                msg += " (" + extra.decode() + ")"
            raise ObtgError("%s: %s" % (what, msg), rc)

    @property
    def handle(self):
        return self._h

    def set_stream(self, stream_ptr):
        """Every later call goes to this HIP stream, the handle as given: 0 / None = the null stream, i.e. torch's default
        stream (torch.cuda.current_stream().cuda_stream is the right argument whichever stream is current)."""
        self._check(self._lib.obtg_ctx_set_stream(self._h, _vp(stream_ptr or None)), "obtg_ctx_set_stream")

    def use_own_stream(self):
        """Back to the context's private stream (created non-blocking: not ordered with the null stream)."""
        self._check(self._lib.obtg_ctx_use_own_stream(self._h), "obtg_ctx_use_own_stream")

    def set_deg_elev(self, R):
        self._check(self._lib.obtg_ctx_set_deg_elev(self._h, int(R)), "obtg_ctx_set_deg_elev")
        self.deg_elev = int(R)

    def set_ang_rate_order(self, elevate_first):
        """DEG_ELEV > 0: False / 0 (default) = products at degree 4n, then elevation by 4R; True / 1 = the reference's
        order of operations; 2 = the default order plus a double-double recompute of near-stop vehicles' rows
        (include/obtg.h obtg_ctx_set_ang_rate_order)."""
        self._check(self._lib.obtg_ctx_set_ang_rate_order(self._h, int(elevate_first)), "obtg_ctx_set_ang_rate_order")

    def ang_rate_order_in_effect(self):
        """0 / 1 / 2 = the order of operations the angular rate of this shape really runs in (a request for 'exact' or
        'fast' holds only where those kernels exist: include/obtg.h obtg_ctx_ang_rate_order_in_effect)."""
        rc = self._lib.obtg_ctx_ang_rate_order_in_effect(self._h)
        if rc < 0:
            self._check(rc, "obtg_ctx_ang_rate_order_in_effect")
        return rc

    def sync(self):
        self._check(self._lib.obtg_sync(self._h), "obtg_sync")

    @property
    def len_temporal_sep(self):
        return self._lib.obtg_len_temporal_sep(self._h)

    @property
    def len_speed(self):
        return self._lib.obtg_len_speed(self._h)

    @property
    def len_ang_rate(self):
        return self._lib.obtg_len_ang_rate(self._h)

    @property
    def num_pairs(self):
        return self._lib.obtg_num_pairs(self._h)

    def _rows(self, Y):
        Y = _f64(Y)
        per = self.n_veh * self.dim * (self.deg + 1)
        if Y.size % per != 0 or Y.size == 0:
            raise ValueError("Y must hold B x (%d x %d) doubles, got %s" % (self.n_veh * self.dim, self.deg + 1, Y.shape))
        return Y, Y.size // per

    def _tf(self, tf, B):
        return _f64(np.broadcast_to(np.asarray(tf, dtype=np.float64), (B,)))

    # -- host-buffer sweeps
    def temporal_sep(self, Y, max_sep):
        Y, B = self._rows(Y)
        out = pinned_empty((B, self.len_temporal_sep))
        self._check(self._lib.obtg_temporal_sep(self._h, _ptr(Y), B, float(max_sep), _ptr(out)), "obtg_temporal_sep")
        return out

    def temporal_sep_min(self, Y, max_sep, pair_begin=0, pair_count=None):
        Y, B = self._rows(Y)
        if pair_count is None:
            pair_count = self.num_pairs - pair_begin
        out = pinned_empty((B, pair_count))
        self._check(self._lib.obtg_temporal_sep_min_range(self._h, _ptr(Y), B, float(max_sep), int(pair_begin),
                                                          int(pair_count), _ptr(out)), "obtg_temporal_sep_min_range")
        return out

    def temporal_sep_active(self, Y, max_sep, k, with_index=False):
        """Per pair its k smallest elevated separation control points, in control-point order (obtg_temporal_sep_active): [B][P * k]
        (and, with_index, the int32 control-point indices of the same shape)."""
        Y, B = self._rows(Y)
        out = pinned_empty((B, self.num_pairs * int(k)))
        idx = np.empty((B, self.num_pairs * int(k)), np.int32) if with_index else None
        self._check(self._lib.obtg_temporal_sep_active(self._h, _ptr(Y), B, float(max_sep), int(k), _ptr(out), _ptr(idx)),
                    "obtg_temporal_sep_active")
        return (out, idx) if with_index else out

    def temporal_sep_active_dev(self, dY, B, max_sep, k, d_out_val, d_out_idx=None, pair_begin=0, pair_count=None):
        if pair_count is None:
            pair_count = self.num_pairs - pair_begin
        self._check(self._lib.obtg_temporal_sep_active_dev(self._h, _vp(dY), B, float(max_sep), int(k), pair_begin, pair_count,
                                                           _vp(d_out_val), _vp(d_out_idx)), "obtg_temporal_sep_active_dev")

    def temporal_sep_fd_min_rows_dev(self, dY0, n_fixed_cols, h, row_begin, n_rows, max_sep, d_out):
        """Per finite-difference row only the minima of the pairs its vehicle touches: d_out[n_rows][n_obj - 1]
        (obtg_temporal_sep_fd_min_rows_dev)."""
        self._check(self._lib.obtg_temporal_sep_fd_min_rows_dev(self._h, _vp(dY0), int(n_fixed_cols), float(h), int(row_begin),
                                                                int(n_rows), float(max_sep), _vp(d_out)),
                    "obtg_temporal_sep_fd_min_rows_dev")

    # ---- pair partition + collective behind the C ABI (include/obtg.h obtg_comm_*)
    def pair_block(self, n_ranks, rank):
        b, c = _i(0), _i(0)
        self._check(self._lib.obtg_pair_block(self._h, int(n_ranks), int(rank), C.byref(b), C.byref(c)), "obtg_pair_block")
        return b.value, c.value

    def unpack_pair_blocks_dev(self, d_blocks, B, n_ranks, d_rows):
        self._check(self._lib.obtg_unpack_pair_blocks_dev(self._h, _vp(d_blocks), int(B), int(n_ranks), _vp(d_rows)),
                    "obtg_unpack_pair_blocks_dev")

    def temporal_sep_min_gather_dev(self, comm, dY, B, max_sep, d_min_all):
        """This rank's block of the per-pair minima + ONE RCCL all-gather on the context's stream: d_min_all[B][P] on every
        rank (obtg_temporal_sep_min_gather_dev)."""
        rc = self._lib.obtg_temporal_sep_min_gather_dev(self._h, comm._h, _vp(dY), int(B), float(max_sep), _vp(d_min_all))
        if rc:
            raise ObtgError("obtg_temporal_sep_min_gather_dev: %s (%s)" % (self._lib.obtg_strerror(rc).decode(),
                            (self._lib.obtg_comm_last_error(comm._h) or self._lib.obtg_last_error(self._h) or b"").decode()), rc)

    def speed(self, Y, tf, bound, is_max):
        Y, B = self._rows(Y)
        tf = self._tf(tf, B)
        out = pinned_empty((B, self.len_speed))
        self._check(self._lib.obtg_speed(self._h, _ptr(Y), _ptr(tf), B, float(bound), int(bool(is_max)), _ptr(out)),
                    "obtg_speed")
        return out

    def ang_rate(self, Y, tf, max_rate):
        Y, B = self._rows(Y)
        tf = self._tf(tf, B)
        out = pinned_empty((B, self.len_ang_rate))
        self._check(self._lib.obtg_ang_rate(self._h, _ptr(Y), _ptr(tf), B, float(max_rate), _ptr(out)),
                    "obtg_ang_rate")
        return out

    def euclidean_obj(self, Y):
        Y, B = self._rows(Y)
        out = np.empty(B)
        self._check(self._lib.obtg_euclidean_obj(self._h, _ptr(Y), B, _ptr(out)), "obtg_euclidean_obj")
        return out

    def deriv_energy_obj(self, Y, tf, order):
        """order 2: _minAccelObjective, order 3: _minJerkObjective (optimization.py:503-539)."""
        Y, B = self._rows(Y)
        tf = self._tf(tf, B)
        out = np.empty(B)
        fn = {2: self._lib.obtg_accel_obj, 3: self._lib.obtg_jerk_obj}[int(order)]
        self._check(fn(self._h, _ptr(Y), _ptr(tf), B, _ptr(out)), "obtg_accel/jerk_obj")
        return out

    def accel_obj(self, Y, tf):
        return self.deriv_energy_obj(Y, tf, 2)

    # -- device-pointer sweeps (pointers are plain ints, e.g. torch.Tensor.data_ptr())
    def temporal_sep_dev(self, dY, B, max_sep, d_out, pair_begin=0, pair_count=None):
        if pair_count is None:
            pair_count = self.num_pairs - pair_begin
        self._check(self._lib.obtg_temporal_sep_dev(self._h, _vp(dY), B, float(max_sep), pair_begin, pair_count,
                                                    _vp(d_out)), "obtg_temporal_sep_dev")

    def temporal_sep_min_dev(self, dY, B, max_sep, d_out, pair_begin=0, pair_count=None):
        if pair_count is None:
            pair_count = self.num_pairs - pair_begin
        self._check(self._lib.obtg_temporal_sep_min_dev(self._h, _vp(dY), B, float(max_sep), pair_begin, pair_count,
                                                        _vp(d_out)), "obtg_temporal_sep_min_dev")

    def speed_dev(self, dY, d_tf, B, bound, is_max, d_out):
        self._check(self._lib.obtg_speed_dev(self._h, _vp(dY), _vp(d_tf), B, float(bound), int(bool(is_max)),
                                             _vp(d_out)), "obtg_speed_dev")

    def ang_rate_dev(self, dY, d_tf, B, max_rate, d_out):
        self._check(self._lib.obtg_ang_rate_dev(self._h, _vp(dY), _vp(d_tf), B, float(max_rate), _vp(d_out)),
                    "obtg_ang_rate_dev")

    def dynamics_dev(self, dY, d_tf, B, speed_bound, speed_is_max, max_rate, d_out_speed, d_out_ang):
        self._check(self._lib.obtg_dynamics_dev(self._h, _vp(dY), _vp(d_tf), B, float(speed_bound),
                                                int(bool(speed_is_max)), float(max_rate), _vp(d_out_speed),
                                                _vp(d_out_ang)), "obtg_dynamics_dev")

    def fd_batch_dev(self, dY0, n_fixed_cols, h, B, dY):
        self._check(self._lib.obtg_fd_batch_dev(self._h, _vp(dY0), int(n_fixed_cols), float(h), B, _vp(dY)),
                    "obtg_fd_batch_dev")

    def one_vs_many_min(self, one, many, max_sep):
        """Examples/SequentialSwarm.py:43-70: one[B][dim][deg+1] (or [dim][deg+1]) against many[K][dim][deg+1] ->
        out[B][K], the per-pair minimum of the elevated squared-distance control points minus max_sep^2
        (include/obtg.h obtg_one_vs_many_min)."""
        nc = self.deg + 1
        one = np.ascontiguousarray(one, dtype=np.float64).reshape(-1, self.dim, nc)
        many = np.ascontiguousarray(many, dtype=np.float64).reshape(-1, self.dim, nc)
        B, K = one.shape[0], many.shape[0]
        out = np.empty((B, K))
        self._check(self._lib.obtg_one_vs_many_min(self._h, _ptr(one), B, _ptr(many), K, float(max_sep), _ptr(out)),
                    "obtg_one_vs_many_min")
        return out

    def one_vs_many_min_dev(self, d_one, B, d_many, K, max_sep, d_out):
        self._check(self._lib.obtg_one_vs_many_min_dev(self._h, _vp(d_one), int(B), _vp(d_many), int(K), float(max_sep),
                                                       _vp(d_out)), "obtg_one_vs_many_min_dev")

    def set_second_speed_bound(self, bound, is_max, d_out2):
        """Both speed bounds from one dynamics pass (include/obtg.h obtg_ctx_set_second_speed_bound): while d_out2 (a
        device pointer, [B][N*(2n+R+1)]) is set, dynamics_dev / constraint_sweep_dev also write this bound's rows.
        d_out2 = None switches it off."""
        self._check(self._lib.obtg_ctx_set_second_speed_bound(self._h, float(bound), int(bool(is_max)),
                                                               _vp(d_out2) if d_out2 else None),
                    "obtg_ctx_set_second_speed_bound")

    def fd_view_begin(self, dY0, n_fixed_cols, h, B, row_begin=0):
        """Open a virtual finite-difference batch over the ONE device row dY0 (include/obtg.h obtg_fd_view_begin): until
        fd_view_end the `_dev` sweeps take dY = None.  row_begin > 0: the view is rows row_begin .. row_begin + B - 1 of
        the batch (obtg_fd_view_begin_rows: one rank's share of a row-sharded iteration)."""
        if row_begin:
            self._check(self._lib.obtg_fd_view_begin_rows(self._h, _vp(dY0), int(n_fixed_cols), float(h), int(row_begin), int(B)),
                        "obtg_fd_view_begin_rows")
        else:
            self._check(self._lib.obtg_fd_view_begin(self._h, _vp(dY0), int(n_fixed_cols), float(h), int(B)), "obtg_fd_view_begin")

    def fd_view_end(self):
        self._check(self._lib.obtg_fd_view_end(self._h), "obtg_fd_view_end")

    def fd_forms_on_the_fly(self):
        """(pair sweep, dynamics): does the _fd_dev form build the finite-difference rows while staging them?"""
        m = self._lib.obtg_fd_forms_on_the_fly(self._h)
        return bool(m & 1), bool(m & 2)

    def pair_sweep_fd_dev(self, dY0, n_fixed_cols, h, B, max_sep, d_out_sep, d_flag, d_p1, d_p2, d_dist, d_nsup=None,
                          d_status=None, max_iter=128, md_cap=4096):
        """obtg_pair_sweep_dev on the virtual FD batch of ONE row dY0 (include/obtg.h obtg_pair_sweep_fd_dev)."""
        self._need_hull_pairs("pair_sweep_fd_dev")
        self._check(self._lib.obtg_pair_sweep_fd_dev(self._h, _vp(dY0), int(n_fixed_cols), float(h), B, float(max_sep),
                                                     _vp(d_out_sep), max_iter, md_cap, _vp(d_flag), _vp(d_p1), _vp(d_p2),
                                                     _vp(d_dist), _vp(d_nsup) if d_nsup else None,
                                                     _vp(d_status) if d_status else None), "obtg_pair_sweep_fd_dev")

    def dynamics_fd_dev(self, dY0, n_fixed_cols, h, d_tf, B, speed_bound, speed_is_max, max_rate, d_out_speed, d_out_ang):
        self._check(self._lib.obtg_dynamics_fd_dev(self._h, _vp(dY0), int(n_fixed_cols), float(h), _vp(d_tf), B,
                                                   float(speed_bound), int(bool(speed_is_max)), float(max_rate),
                                                   _vp(d_out_speed), _vp(d_out_ang)), "obtg_dynamics_fd_dev")

    # -- GJK
    def gjk_pairs(self, pts, off, pair_a, pair_b, max_iter=128, md_cap=4096, trace_cap=0):
        pts = _f64(pts).reshape(-1, 3)
        off = _i32(off)
        pa, pb = _i32(pair_a), _i32(pair_b)
        n = pa.shape[0]
        flag = np.zeros(n, np.int32)
        nsup = np.zeros(n, np.int32)
        status = np.zeros(n, np.int32)
        p1 = np.empty((n, 3))
        p2 = np.empty((n, 3))
        dist = np.empty(n)
        trace = np.zeros((n, trace_cap, 2), np.int16) if trace_cap else None
        self._check(self._lib.obtg_gjk_pairs(self._h, _ptr(pts), pts.shape[0], _ptr(off), off.shape[0] - 1,
                                             _ptr(pa), _ptr(pb), n, max_iter, md_cap, _ptr(flag), _ptr(p1),
                                             _ptr(p2), _ptr(dist), _ptr(trace), trace_cap, _ptr(nsup),
                                             _ptr(status)), "obtg_gjk_pairs")
        return dict(flag=flag, c1=p1, c2=p2, dist=dist, trace=trace, n_support=nsup, status=status)

    def set_polygons(self, pts, off):
        if pts is None or len(off) <= 1:
            self._check(self._lib.obtg_ctx_set_polygons(self._h, None, 0, None, 0), "obtg_ctx_set_polygons")
            self.n_poly = 0
            self.n_hull_pairs = None      # object ids changed meaning: the library dropped the pair list too
            return
        pts = _f64(pts).reshape(-1, 3)
        off = _i32(off)
        self._check(self._lib.obtg_ctx_set_polygons(self._h, _ptr(pts), pts.shape[0], _ptr(off), off.shape[0] - 1),
                    "obtg_ctx_set_polygons")
        self.n_poly = off.shape[0] - 1
        self.n_hull_pairs = None

    def _need_hull_pairs(self, what):
        if self.n_hull_pairs is None:
            raise ObtgError("%s: no hull pair list registered -- call set_hull_pairs() (again) after set_polygons()" % what)

    def set_hull_pairs(self, pair_a, pair_b):
        pa, pb = _i32(pair_a), _i32(pair_b)
        self._check(self._lib.obtg_ctx_set_hull_pairs(self._h, _ptr(pa), _ptr(pb), pa.shape[0]),
                    "obtg_ctx_set_hull_pairs")
        self.n_hull_pairs = pa.shape[0]

    def set_fd_dedup(self, on):
        self._check(self._lib.obtg_ctx_set_fd_dedup(self._h, int(bool(on))), "obtg_ctx_set_fd_dedup")

    def set_gjk_history(self, on):
        self._check(self._lib.obtg_ctx_set_gjk_history(self._h, int(bool(on))), "obtg_ctx_set_gjk_history")

    def gjk_swarm(self, Y, max_iter=128, md_cap=4096):
        self._need_hull_pairs("gjk_swarm")
        Y, B = self._rows(Y)
        n = self.n_hull_pairs
        flag, nsup, status = (pinned_empty((B, n), np.int32) for _ in range(3))
        p1, p2, dist = pinned_empty((B, n, 3)), pinned_empty((B, n, 3)), pinned_empty((B, n))
        for a, v in ((flag, 0), (nsup, 0), (status, 0), (p1, np.nan), (p2, np.nan), (dist, np.nan)):
            a.fill(v)
        self._check(self._lib.obtg_gjk_swarm(self._h, _ptr(Y), B, max_iter, md_cap, _ptr(flag), _ptr(p1), _ptr(p2),
                                             _ptr(dist), _ptr(nsup), _ptr(status)), "obtg_gjk_swarm")
        return dict(flag=flag, c1=p1, c2=p2, dist=dist, n_support=nsup, status=status)

    def gjk_swarm_dev(self, dY, B, d_flag, d_p1, d_p2, d_dist, d_nsup=None, d_status=None, max_iter=128,
                      md_cap=4096):
        self._need_hull_pairs("gjk_swarm_dev")
        self._check(self._lib.obtg_gjk_swarm_dev(self._h, _vp(dY), B, max_iter, md_cap, _vp(d_flag), _vp(d_p1),
                                                 _vp(d_p2), _vp(d_dist), _vp(d_nsup), _vp(d_status)),
                    "obtg_gjk_swarm_dev")

    # -- minDist
    def pair_sweep_dev(self, dY, B, max_sep, d_out_sep, d_flag, d_p1, d_p2, d_dist, d_nsup=None, d_status=None,
                       max_iter=128, md_cap=4096):
        """temporal separation + gjkNew hull sweep of the same rows in one launch (obtg_pair_sweep_dev)."""
        self._need_hull_pairs("pair_sweep_dev")
        self._check(self._lib.obtg_pair_sweep_dev(self._h, _vp(dY), B, float(max_sep), _vp(d_out_sep), max_iter,
                                                  md_cap, _vp(d_flag), _vp(d_p1), _vp(d_p2), _vp(d_dist),
                                                  _vp(d_nsup) if d_nsup else None,
                                                  _vp(d_status) if d_status else None), "obtg_pair_sweep_dev")

    def constraint_sweep_dev(self, dY, d_tf, B, max_sep, d_out_sep, speed_bound, speed_is_max, max_rate, d_out_speed,
                             d_out_ang, d_flag, d_p1, d_p2, d_dist, d_nsup=None, d_status=None, max_iter=128, md_cap=4096):
        """Every constraint family of the batch in one call (obtg_constraint_sweep_dev)."""
        self._need_hull_pairs("constraint_sweep_dev")
        self._check(self._lib.obtg_constraint_sweep_dev(self._h, _vp(dY), _vp(d_tf), B, float(max_sep), _vp(d_out_sep),
                                                        float(speed_bound), int(bool(speed_is_max)), float(max_rate),
                                                        _vp(d_out_speed), _vp(d_out_ang), max_iter, md_cap, _vp(d_flag),
                                                        _vp(d_p1), _vp(d_p2), _vp(d_dist), _vp(d_nsup) if d_nsup else None,
                                                        _vp(d_status) if d_status else None), "obtg_constraint_sweep_dev")

    def constraint_sweep_fd_structured_dev(self, dY0, n_fixed_cols, h, d_tf, B, max_sep, d_out_sep, speed_bound, speed_is_max,
                                           max_rate, d_out_speed, d_out_ang, d_flag, d_p1, d_p2, d_dist, d_nsup=None,
                                           d_status=None, max_iter=128, md_cap=4096, row_begin=0):
        """The FD step as one launch that evaluates row 0 in full and per row only what its vehicle touches
        (obtg_constraint_sweep_fd_structured_dev); raises for shapes it does not cover (OBTG_ERR_UNSUPPORTED).
        row_begin > 0: the B rows [row_begin, row_begin + B) of the batch (obtg_constraint_sweep_fd_structured_rows_dev):
        d_tf and the outputs hold those B rows."""
        self._need_hull_pairs("constraint_sweep_fd_structured_dev")
        self._check(self._lib.obtg_constraint_sweep_fd_structured_rows_dev(
            self._h, _vp(dY0), int(n_fixed_cols), float(h), int(row_begin), _vp(d_tf), int(B), float(max_sep), _vp(d_out_sep),
            float(speed_bound), int(bool(speed_is_max)), float(max_rate), _vp(d_out_speed), _vp(d_out_ang), max_iter, md_cap,
            _vp(d_flag), _vp(d_p1), _vp(d_p2), _vp(d_dist), _vp(d_nsup) if d_nsup else None,
            _vp(d_status) if d_status else None), "obtg_constraint_sweep_fd_structured_rows_dev")

    def min_dist(self, curves, pair_a, pair_b, eps=1e-9, max_iter=128, md_cap=4096, max_depth=64,
                 max_nodes=200000):
        """curves[n][3][K] (2-D curves: pass a zero z row)."""
        curves = _f64(curves)
        n_curves, _, K = curves.shape
        pa, pb = _i32(pair_a), _i32(pair_b)
        n = pa.shape[0]
        res = np.empty((n, 3))
        info = np.zeros((n, 4), np.int32)
        status = np.zeros(n, np.int32)
        self._check(self._lib.obtg_min_dist(self._h, _ptr(curves), n_curves, K, _ptr(pa), _ptr(pb), n, float(eps),
                                            max_iter, md_cap, max_depth, max_nodes, _ptr(res), _ptr(info),
                                            _ptr(status)), "obtg_min_dist")
        return dict(res=res, nodes=info[:, 0], gjk_calls=info[:, 1], depth=info[:, 2], status=status)

    def min_dist_robust(self, curves, pair_a, pair_b, eps=1e-9, max_nodes=200000):
        """Robust branch & bound (obtg_min_dist_robust): true minimum within relative eps when status == MD_OK."""
        curves = _f64(curves)
        n_curves, _, K = curves.shape
        pa, pb = _i32(pair_a), _i32(pair_b)
        n = pa.shape[0]
        res = np.empty((n, 3))
        info = np.zeros((n, 4), np.int32)
        status = np.zeros(n, np.int32)
        self._check(self._lib.obtg_min_dist_robust(self._h, _ptr(curves), n_curves, K, _ptr(pa), _ptr(pb), n, float(eps),
                                                   int(max_nodes), _ptr(res), _ptr(info), _ptr(status)),
                    "obtg_min_dist_robust")
        return dict(res=res, nodes=info[:, 0], levels=info[:, 1], frontier=info[:, 2], status=status)

    def min_dist2poly(self, curves, pts, off, pair_curve, pair_poly, eps=1e-6, max_iter=128, md_cap=4096,
                      max_depth=64, max_nodes=200000):
        curves = _f64(curves)
        n_curves, _, K = curves.shape
        pts = _f64(pts).reshape(-1, 3)
        off = _i32(off)
        pc, pp = _i32(pair_curve), _i32(pair_poly)
        n = pc.shape[0]
        res = np.empty((n, 5))
        info = np.zeros((n, 4), np.int32)
        status = np.zeros(n, np.int32)
        self._check(self._lib.obtg_min_dist2poly(self._h, _ptr(curves), n_curves, K, _ptr(pts), pts.shape[0],
                                                 _ptr(off), off.shape[0] - 1, _ptr(pc), _ptr(pp), n, float(eps),
                                                 max_iter, md_cap, max_depth, max_nodes, _ptr(res), _ptr(info),
                                                 _ptr(status)), "obtg_min_dist2poly")
        return dict(res=res, nodes=info[:, 0], gjk_calls=info[:, 1], depth=info[:, 2], status=status)

    def min_dist2poly_robust(self, curves, pts, off, pair_curve, pair_poly, eps=1e-9, max_nodes=200000):
        """Robust curve <-> polygon distance (obtg_min_dist2poly_robust): true minimum within relative eps when MD_OK."""
        curves = _f64(curves)
        n_curves, _, K = curves.shape
        pts = _f64(pts).reshape(-1, 3)
        off = _i32(off)
        pc, pp = _i32(pair_curve), _i32(pair_poly)
        n = pc.shape[0]
        res = np.empty((n, 5))
        info = np.zeros((n, 4), np.int32)
        status = np.zeros(n, np.int32)
        self._check(self._lib.obtg_min_dist2poly_robust(self._h, _ptr(curves), n_curves, K, _ptr(pts), pts.shape[0], _ptr(off),
                                                        off.shape[0] - 1, _ptr(pc), _ptr(pp), n, float(eps), int(max_nodes),
                                                        _ptr(res), _ptr(info), _ptr(status)), "obtg_min_dist2poly_robust")
        return dict(res=res, nodes=info[:, 0], levels=info[:, 1], frontier=info[:, 2], status=status)

    def gjk_true_pairs(self, pts, off, pair_a, pair_b, eps=1e-10, max_iter=64):
        """True hull distances (obtg_gjk_true_pairs; not gjkNew).  status 0: converged with the certificate
        dist - lower <= eps * dist; 1: iteration cap; 2: stalled at rounding level before the certificate closed -- dist is
        then still a distance between hull points and `lower` a proven lower bound: check `lower` (or status)."""
        pts = _f64(pts).reshape(-1, 3)
        off = _i32(off)
        pa, pb = _i32(pair_a), _i32(pair_b)
        n = pa.shape[0]
        flag, iters, status = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        p1, p2, dist, lower = np.empty((n, 3)), np.empty((n, 3)), np.empty(n), np.empty(n)
        self._check(self._lib.obtg_gjk_true_pairs(self._h, _ptr(pts), pts.shape[0], _ptr(off), off.shape[0] - 1, _ptr(pa), _ptr(pb),
                                                  n, float(eps), int(max_iter), _ptr(flag), _ptr(p1), _ptr(p2), _ptr(dist),
                                                  _ptr(lower), _ptr(iters), _ptr(status)), "obtg_gjk_true_pairs")
        return dict(flag=flag, c1=p1, c2=p2, dist=dist, lower=lower, iters=iters, status=status)

    # -- single-curve algebra
    def bern_elev(self, cpts, R):
        a = np.atleast_2d(_f64(cpts))
        rows, nc = a.shape
        out = np.empty((rows, nc + R))
        self._check(self._lib.obtg_bern_elev(self._h, _ptr(a), rows, nc - 1, int(R), _ptr(out)), "obtg_bern_elev")
        return out

    def bern_diff(self, cpts, T):
        a = np.atleast_2d(_f64(cpts))
        rows, nc = a.shape
        out = np.empty((rows, nc))
        self._check(self._lib.obtg_bern_diff(self._h, _ptr(a), rows, nc - 1, float(T), _ptr(out)), "obtg_bern_diff")
        return out

    def bern_mul(self, a, b):
        a = np.atleast_2d(_f64(a))
        b = np.atleast_2d(_f64(b))
        rows, mc = a.shape
        nc = b.shape[1]
        out = np.empty((rows, mc + nc - 1))
        self._check(self._lib.obtg_bern_mul(self._h, _ptr(a), _ptr(b), rows, mc - 1, nc - 1, _ptr(out)),
                    "obtg_bern_mul")
        return out

    def bern_normsq(self, x):
        x = np.atleast_2d(_f64(x))
        d, nc = x.shape
        out = np.empty((1, 2 * nc - 1))
        self._check(self._lib.obtg_bern_normsq(self._h, _ptr(x), d, nc - 1, _ptr(out)), "obtg_bern_normsq")
        return out

    def bern_split(self, cpts, z):
        """de Casteljau split of every row at parameter z in [0, 1] -> (left, right), each rows x (n+1)."""
        a = np.atleast_2d(_f64(cpts))
        rows, nc = a.shape
        left, right = np.empty((rows, nc)), np.empty((rows, nc))
        self._check(self._lib.obtg_bern_split(self._h, _ptr(a), rows, nc - 1, float(z), _ptr(left), _ptr(right)),
                    "obtg_bern_split")
        return left, right

    def bern_eval(self, cpts, tau, t0, tf):
        """Every row of control points at every tau (obtg_bern_eval: de Casteljau per sample) -> rows x len(tau)."""
        a = np.atleast_2d(_f64(cpts))
        tau = np.atleast_1d(_f64(tau)).reshape(-1)
        rows, nc = a.shape
        out = np.empty((rows, tau.size))
        self._check(self._lib.obtg_bern_eval(self._h, _ptr(a), rows, nc - 1, _ptr(tau), tau.size, float(t0), float(tf), _ptr(out)),
                    "obtg_bern_eval")
        return out

    # -- instrumentation
    def temporal_sep_fd(self, Y0, pert_row, pert_col, pert_val, max_sep):
        """Structured finite differences: -> blk[n_pert][n_obj-1][2n+R+1] (include/obtg.h obtg_temporal_sep_fd)."""
        Y0 = _f64(Y0).reshape(self.n_veh * self.dim, self.deg + 1)
        pr = np.ascontiguousarray(pert_row, dtype=np.int32)
        pc = np.ascontiguousarray(pert_col, dtype=np.int32)
        pv = _f64(pert_val)
        n = pr.shape[0]
        out = pinned_empty((n, max(self.n_veh + self.n_obs - 1, 0), 2 * self.deg + self.deg_elev + 1))
        self._check(self._lib.obtg_temporal_sep_fd(self._h, _ptr(Y0), n, _ptr(pr), _ptr(pc), _ptr(pv),
                                                   float(max_sep), _ptr(out)), "obtg_temporal_sep_fd")
        return out

    def temporal_sep_fd_dev(self, dY0, n_pert, d_row, d_col, d_val, max_sep, d_out):
        self._check(self._lib.obtg_temporal_sep_fd_dev(self._h, _vp(dY0), int(n_pert), _vp(d_row), _vp(d_col),
                                                       _vp(d_val), float(max_sep), _vp(d_out)),
                    "obtg_temporal_sep_fd_dev")

    def set_profiling(self, on, only=None):
        """on: events around every kernel launch; only='gjk' (a kernel name): around that kernel alone."""
        mode = int(bool(on))
        if on and only is not None:
            names = [self._lib.obtg_kernel_name(k).decode() for k in range(K_COUNT)]
            mode = 0x100 | names.index(only)
        self._check(self._lib.obtg_set_profiling(self._h, mode), "obtg_set_profiling")

    def set_profile_period(self, every):
        self._check(self._lib.obtg_set_profile_period(self._h, int(every)), "obtg_set_profile_period")

    def reset_kernel_stats(self):
        self._check(self._lib.obtg_reset_kernel_stats(self._h), "obtg_reset_kernel_stats")

    def kernel_stats(self):
        out = {}
        for k in range(K_COUNT):
            ms = _d(0)
            n = C.c_longlong(0)
            self._check(self._lib.obtg_kernel_stats(self._h, k, C.byref(ms), C.byref(n)), "obtg_kernel_stats")
            out[self._lib.obtg_kernel_name(k).decode()] = (ms.value, n.value)
        return out


_scratch_ctx = None


def scratch_context():
    """A shape-agnostic context for the calls that do not depend on the problem shape
    (gjkNew on raw point sets, minDist, single-curve Bernstein algebra)."""
    global _scratch_ctx
    if _scratch_ctx is None:
        _scratch_ctx = Context(1, 2, 1, 0, device=default_device())
    return _scratch_ctx


def default_device():
    """The GPU of this process when nothing names one: OBTG_DEVICE, else the launcher's LOCAL_RANK (one process per GPU),
    wrapped to the devices present (a rehearsal of several ranks on one card), else 0."""
    n = max(load().obtg_device_count(), 1)
    for key in ("OBTG_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(key)
        if v is not None and v.strip().lstrip("-").isdigit():
            return int(v) % n
    return 0
