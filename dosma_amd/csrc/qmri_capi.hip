// qmri_capi.hip -- the C ABI declared in include/qmri.h (extern "C", plain pointers and sizes).
//
// Boundary it implements: the reference's Python seam  _Fitter._fit / curve_fit
//   /root/reference/dosma/core/fitting.py:422-435, 755-870
// i.e. "(E, N) echo-major samples + x + p0  ->  (N, 2) popt + (N,) r2", for func = monoexponential.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "fp64_fast.h"
#include "qmri_internal.h"

namespace {

thread_local char g_err[512] = "";
thread_local int g_timing = 0;
thread_local float g_last_ms = 0.f;

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(QMRI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                  \
    } while (0)

constexpr int kMaxDevices = 16;

struct DeviceCtx {
    std::once_flag once;
    hipError_t init_err = hipSuccess;
    int num_cu = 0;
};
DeviceCtx g_ctx[kMaxDevices];

hipError_t ctx_get(int device, DeviceCtx **out) {
    DeviceCtx &c = g_ctx[device];
    std::call_once(c.once, [&] {
        hipError_t e = hipSetDevice(device);
        hipDeviceProp_t prop;
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
        if (e == hipSuccess) c.num_cu = prop.multiProcessorCount;
        c.init_err = e;
    });
    *out = &c;
    return c.init_err;
}

size_t dtype_size(int dt) {
    switch (dt) {
        case QMRI_F32: return 4;
        case QMRI_F64: return 8;
        case QMRI_I16:
        case QMRI_U16: return 2;
        default: return 0;
    }
}

int validate(const qmri_monoexp_args *a) {
    if (!a) return fail(QMRI_ERR_ARG, "args is NULL");
    if ((!a->y && !a->y_rows) || !a->x || !a->r2 || (!a->popt && !a->tc))
        return fail(QMRI_ERR_ARG, "y, x, r2 and popt (or tc) are required");
    if (a->y_rows)
        for (int e = 0; e < a->E && e < QMRI_MAX_ECHOES; ++e)
            if (!a->y_rows[e]) return fail(QMRI_ERR_ARG, "y_rows[%d] is NULL", e);
    if (dtype_size(a->y_dtype) == 0) return fail(QMRI_ERR_ARG, "unknown y_dtype %d", a->y_dtype);
    if (a->out_dtype != QMRI_F32 && a->out_dtype != QMRI_F64)
        return fail(QMRI_ERR_ARG, "out_dtype must be QMRI_F32 or QMRI_F64");
    if (a->E < 2)
        return fail(QMRI_ERR_ARG, "E=%d: need at least as many samples as parameters (2)", a->E);
    if (a->E > QMRI_MAX_ECHOES)
        return fail(QMRI_ERR_UNSUPPORTED, "E=%d exceeds QMRI_MAX_ECHOES=%d", a->E, QMRI_MAX_ECHOES);
    if (a->N < 0 || a->ld < a->N) return fail(QMRI_ERR_ARG, "need 0 <= N <= ld");
    // the result ring carries a 40-bit voxel index, and tile indices / the claim counter / the tile list are 32-bit at 128 voxels
    // per tile (monoexp_tile_voxels()): below 2^38 voxels both hold with room for the guided claims' overshoot
    if (a->N >= (1ll << 38)) return fail(QMRI_ERR_UNSUPPORTED, "N must be below 2^38 voxels per call");
    if (a->init < QMRI_INIT_SCALAR || a->init > QMRI_INIT_LOGLIN)
        return fail(QMRI_ERR_ARG, "unknown init mode %d", a->init);
    if (a->maxfev <= 0 || a->ftol < 0 || a->xtol < 0 || a->gtol < 0 || a->factor <= 0)
        return fail(QMRI_ERR_ARG, "bad solver constants (MINPACK 'improper input parameters')");
    if (a->device < 0 || a->device >= kMaxDevices) return fail(QMRI_ERR_ARG, "bad device %d", a->device);
    if (a->post.enable && a->post.decimals > 300) return fail(QMRI_ERR_ARG, "bad decimals");
    for (int i = 0; i < a->E; ++i)
        if (!std::isfinite(a->x[i]))
            return fail(QMRI_ERR_NONFINITE, "x holds a non-finite value");
    return QMRI_OK;
}

void fill_kargs(const qmri_monoexp_args *a, qmri::FitKArgs &k) {
    std::memset(&k, 0, sizeof(k));
    k.y = a->y;
    k.ld = a->ld;
    k.N = a->N;
    k.E = a->E;
    k.y_dtype = a->y_dtype;
    const size_t es = dtype_size(a->y_dtype);
    k.vec_ok = (reinterpret_cast<uintptr_t>(a->y) % (4 * es) == 0) && (a->ld % 4 == 0);
    k.init = a->init;
    k.use_y_bounds = a->use_y_bounds;
    {
        const char *env = std::getenv("QMRI_REFILL_IDLE");
        k.refill_idle = env ? std::atoi(env) : 8;
        if (k.refill_idle < 1) k.refill_idle = 1;
        if (k.refill_idle > 64) k.refill_idle = 64;
    }
    k.y_lo = a->y_lo;
    k.y_hi = a->y_hi;
    k.mask = a->mask;
    k.a0v = a->init == QMRI_INIT_PER_VOXEL ? a->a0v : nullptr;
    k.b0v = a->init == QMRI_INIT_PER_VOXEL ? a->b0v : nullptr;
    k.a0 = a->a0;
    k.b0 = a->b0;
    k.ftol = a->ftol;
    k.xtol = a->xtol;
    k.gtol = a->gtol;
    k.factor = a->factor;
    k.r2_eps = a->r2_eps;
    k.maxfev = a->maxfev;
    k.out_f64 = a->out_dtype == QMRI_F64;
    k.post = a->post;
    if (!k.post.enable || k.post.decimals < -300) k.post.decimals = QMRI_NO_ROUND;  // also: nothing below 1e-300
    k.p10 = k.post.decimals == QMRI_NO_ROUND ? 1.0 : std::pow(10.0, std::abs(k.post.decimals));
    k.popt = a->popt;
    k.r2 = a->r2;
    k.tc = a->tc;
    k.info = a->info;
    k.nfev = a->nfev;
    double xm = 0.0;
    for (int i = 0; i < a->E; ++i) {
        k.x[i] = a->x[i];
        xm += a->x[i];
    }
    xm /= a->E;
    double sxx = 0.0;
    for (int i = 0; i < a->E; ++i) sxx += (a->x[i] - xm) * (a->x[i] - xm);
    k.xmean = xm;
    k.sxx = sxx;
    // equally spaced echo / spin-lock times (the usual multi-echo acquisition): see FitKArgs::uniform_x
    static const int uni_ok = [] { const char *e = std::getenv("QMRI_FIT_UNIFORM_X"); return e ? std::atoi(e) : 1; }();
    static const int lmpar_cf = [] { const char *e = std::getenv("QMRI_FIT_LMPAR_CF"); return e ? std::atoi(e) : 1; }();
    k.lmpar_closed_form = lmpar_cf;
    k.uniform_x = 0;
    k.x0_pow = -1;
    k.x_step = 0.0;
    if (uni_ok && a->E >= 3 && a->x[0] >= 0.0 && a->x[1] > a->x[0]) {
        const double dx = a->x[1] - a->x[0];
        bool uni = std::isfinite(dx);
        for (int i = 2; i < a->E && uni; ++i)
            uni = std::fabs(a->x[i] - (a->x[0] + i * dx)) <= 8.0 * 2.220446049250313e-16 * std::fabs(a->x[i]);
        if (uni) {
            k.uniform_x = 1;
            k.x_step = dx;
            const double kk = std::nearbyint(a->x[0] / dx);
            if (kk >= 0.0 && kk <= 4.0 && std::fabs(a->x[0] - kk * dx) <= 8.0 * 2.220446049250313e-16 * std::fabs(a->x[0]))
                k.x0_pow = (int)kk;
        }
    }
}

// stream-ordered scratch that is returned to the pool on every exit path (the HIP_TRY early returns included)
struct AsyncScratch {
    void *p = nullptr;
    hipStream_t stream = nullptr;
    hipError_t alloc(size_t bytes, hipStream_t s) {
        stream = s;
        return hipMallocAsync(&p, bytes, s);
    }
    ~AsyncScratch() {
        if (p) (void)hipFreeAsync(p, stream);
    }
};

// issue one fit launch on `stream`; flag_out (device) receives the non-finite flag if non-NULL
int launch_fit(const qmri_monoexp_args *a, int32_t *flag_out) {
    DeviceCtx *ctx = nullptr;
    HIP_TRY(hipSetDevice(a->device));
    HIP_TRY(ctx_get(a->device, &ctx));
    hipStream_t stream = static_cast<hipStream_t>(a->stream);
    if (a->N == 0) return QMRI_OK;

    qmri::FitKArgs k;
    fill_kargs(a, k);
    // the tile counter + non-finite flag of THIS launch: stream-ordered scratch, so any number of fits can be in flight
    // on one device (a fixed ring of slots handed two concurrent kernels the same counter once it wrapped)
    AsyncScratch cnt_scratch;
    HIP_TRY(cnt_scratch.alloc(64, stream));
    unsigned int *cnt = static_cast<unsigned int *>(cnt_scratch.p);
    k.tile_counter = cnt;
    k.nonfinite = flag_out ? flag_out : reinterpret_cast<int *>(cnt + 1);
    HIP_TRY(hipMemsetAsync(cnt, 0, 8, stream));

    const long long tiles = (a->N + qmri::monoexp_tile_voxels() - 1) / qmri::monoexp_tile_voxels();
    // masked volume: fill the tiles without a selected voxel and compact the others first (stream-ordered scratch)
    unsigned int *tile_list = nullptr;
    AsyncScratch tile_scratch;
    if (a->mask && tiles >= 4096) {
        HIP_TRY(tile_scratch.alloc((size_t)(tiles + 1) * 4, stream));
        tile_list = static_cast<unsigned int *>(tile_scratch.p);
        HIP_TRY(hipMemsetAsync(tile_list + tiles, 0, 4, stream));
        HIP_TRY(qmri::monoexp_mask_prepass(k, tile_list, tile_list + tiles, ctx->num_cu, stream));
        k.tile_list = tile_list;
        k.tile_list_count = tile_list + tiles;
    }
    const int per_cu = qmri::monoexp_blocks_per_cu(k);
    long long grid = (long long)ctx->num_cu * per_cu;
    const int wpb = qmri::monoexp_waves_per_block(k);
    const long long need = (tiles + wpb - 1) / wpb;  // one tile per wave at a time
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (g_timing) {
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
        HIP_TRY(hipEventRecord(ev0, stream));
    }
    HIP_TRY(qmri::monoexp_launch(k, static_cast<int>(grid), stream));
    if (g_timing) {
        HIP_TRY(hipEventRecord(ev1, stream));
        HIP_TRY(hipEventSynchronize(ev1));
        HIP_TRY(hipEventElapsedTime(&g_last_ms, ev0, ev1));
        (void)hipEventDestroy(ev0);
        (void)hipEventDestroy(ev1);
    }
    return QMRI_OK;
}

}  // namespace

namespace qmri {
void set_last_error(const char *msg) {
    std::strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
}  // namespace qmri

extern "C" {

int qmri_version(void) { return QMRI_VERSION; }

const char *qmri_last_error(void) { return g_err; }

int qmri_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int qmri_device_mem_info(int32_t device, uint64_t *free_bytes, uint64_t *total_bytes) {
    if (!free_bytes || !total_bytes) return fail(QMRI_ERR_ARG, "qmri_device_mem_info: NULL result pointer");
    HIP_TRY(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    *free_bytes = f;
    *total_bytes = t;
    return QMRI_OK;
}

// ---- self-test of the kernels' own exp / log (fp64_fast.h) ------------------------------------------------------------
namespace {
__global__ void fp64_selftest_kernel(const double *x, long long n, double *e_sk, double *e_lib, double *l_sk, double *l_lib) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    e_sk[i] = qmri::exp_sk(v);
    e_lib[i] = exp(v);
    l_sk[i] = qmri::log_sk(v);
    l_lib[i] = log(v);
}
}  // namespace

int qmri_selftest_fp64(int32_t device, const double *x, int64_t n, double *exp_sk_out, double *exp_lib_out,
                       double *log_sk_out, double *log_lib_out) {
    if (!x || !exp_sk_out || !exp_lib_out || !log_sk_out || !log_lib_out || n < 0)
        return fail(QMRI_ERR_ARG, "qmri_selftest_fp64: NULL pointer or negative n");
    if (n == 0) return QMRI_OK;
    HIP_TRY(hipSetDevice(device));
    AsyncScratch buf;
    const size_t bytes = (size_t)n * sizeof(double);
    HIP_TRY(buf.alloc(5 * bytes, nullptr));
    double *d = static_cast<double *>(buf.p);
    HIP_TRY(hipMemcpy(d, x, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fp64_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d, (long long)n, d + n,
                       d + 2 * n, d + 3 * n, d + 4 * n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(exp_sk_out, d + n, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(exp_lib_out, d + 2 * n, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(log_sk_out, d + 3 * n, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(log_lib_out, d + 4 * n, bytes, hipMemcpyDeviceToHost));
    return QMRI_OK;
}

int qmri_host_alloc(uint64_t bytes, void **out) {
    if (!out || bytes == 0) return fail(QMRI_ERR_ARG, "qmri_host_alloc: NULL result pointer or zero bytes");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, (size_t)bytes, hipHostMallocPortable));  // (usable from every device of the process)
    // ROCm marks page-locked host allocations MADV_DONTFORK: a fork()ed child (a multiprocessing worker -- the reference's own
    // parallelism, dosma/core/fitting.py:860-868) that touches a result array it inherited would fault.  Results handed to
    // Python live in these blocks (dosma_amd/_hostpool.py), so inheritance is switched back on: the child gets ordinary
    // copy-on-write pages of the parent's data (it never owns, frees or DMA-targets the block).  Best effort: a platform that
    // refuses leaves the block as the runtime made it.
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(*out) & ~(uintptr_t)4095;
    const uintptr_t a1 = (reinterpret_cast<uintptr_t>(*out) + bytes + 4095) & ~(uintptr_t)4095;
    (void)madvise(reinterpret_cast<void *>(a0), (size_t)(a1 - a0), MADV_DOFORK);
    return QMRI_OK;
}

int qmri_host_free(void *p) {
    if (p) HIP_TRY(hipHostFree(p));
    return QMRI_OK;
}

void qmri_set_timing(int enable) { g_timing = enable; }
float qmri_last_kernel_ms(void) { return g_last_ms; }

void qmri_monoexp_defaults(qmri_monoexp_args *a) {
    if (!a) return;
    // dosma/core/fitting.py:761-763 (maxfev, ftol, eps) + scipy.optimize.leastsq defaults
    a->ftol = 1e-5;
    a->xtol = 1.49012e-8;
    a->gtol = 0.0;
    a->factor = 100.0;
    a->r2_eps = 1e-8;
    a->maxfev = 100;
    a->init = QMRI_INIT_SCALAR;
    a->use_y_bounds = 0;
    a->y_lo = -INFINITY;
    a->y_hi = INFINITY;
    a->a0 = 1.0;  // scipy: p0 = ones(n) when p0 is None
    a->b0 = 1.0;
    std::memset(&a->post, 0, sizeof(a->post));
    a->post.decimals = QMRI_NO_ROUND;
    a->post.lb[0] = a->post.lb[1] = -INFINITY;
    a->post.ub[0] = a->post.ub[1] = INFINITY;
    a->out_dtype = QMRI_F64;
}

const char *qmri_monoexp_kernel_name(const qmri_monoexp_args *a) {
    if (!a) return "";
    return qmri::monoexp_variant_name(a->E, a->y_dtype);
}

int qmri_monoexp_fit_device(const qmri_monoexp_args *a, int32_t *nonfinite_flag) {
    const int rc = validate(a);
    if (rc != QMRI_OK) return rc;
    if (!a->y) return fail(QMRI_ERR_ARG, "the device entry needs the (E, ld) array y (y_rows is for the host entry)");
    return launch_fit(a, nonfinite_flag);
}

// ---- host entry: slab pipeline -------------------------------------------------------------------------
// The caller's thread uploads slab i + 1 (pageable hipMemcpy: ~56 GB/s on this platform, measured, pinned or
// not) while the kernel of slab i runs on its stream and a helper thread downloads slab i - 1: PCIe is used in
// both directions at once.  Two things used to dominate the call (scripts/pin_probe.py, scripts/host_rate.py):
// per-call hipMalloc / hipFree of the slab buffers (now cached per device), and the first touch of the freshly
// allocated OUTPUT arrays -- the D2H copy faulted in and zeroed 1 GB of pages at 17 GB/s; now 8 threads touch the
// output pages while the first slabs upload and compute.
namespace {

struct SlabBuf {
    void *y = nullptr, *popt = nullptr, *r2 = nullptr, *tc = nullptr;
    uint8_t *mask = nullptr;
    double *a0v = nullptr, *b0v = nullptr;
    int8_t *info = nullptr;
    int16_t *nfev = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;  // kernel finished
    size_t cap[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
};
struct HostPipe {
    std::mutex mu;  // one host call at a time per device uses the cached buffers
    SlabBuf buf[2];
    int32_t *flag = nullptr;
    hipStream_t d2h = nullptr;
};
HostPipe g_pipe[kMaxDevices];

hipError_t ensure(void **p, size_t *cap, size_t need) {
    if (need <= *cap) return hipSuccess;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc(p, need);
    if (e == hipSuccess) *cap = need;
    return e;
}

// first touch of part `part` of `parts` of [base, base + bytes): MADV_POPULATE_WRITE (one call, no per-page trap)
// on the page-aligned interior, a plain write per page otherwise
void prefault(char *base, size_t bytes, int part, int parts) {
    if (!base || !bytes) return;
    const size_t share = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
    size_t lo = (size_t)part * share, hi = lo + share < bytes ? lo + share : bytes;
    if (lo >= hi) return;
    char *p = base + lo, *end = base + hi;
    char *ap = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(p) + 4095) & ~(uintptr_t)4095);
    char *ae = reinterpret_cast<char *>(reinterpret_cast<uintptr_t>(end) & ~(uintptr_t)4095);
#ifdef MADV_POPULATE_WRITE
    if (ae > ap && madvise(ap, (size_t)(ae - ap), MADV_POPULATE_WRITE) == 0) {
        *reinterpret_cast<volatile char *>(p) = 0;
        *reinterpret_cast<volatile char *>(end - 1) = 0;
        return;
    }
#endif
    for (char *q = p; q < end; q += 4096) *reinterpret_cast<volatile char *>(q) = 0;
    *reinterpret_cast<volatile char *>(end - 1) = 0;
}

}  // namespace

int qmri_monoexp_fit_host(const qmri_monoexp_args *a) {
    const int rc = validate(a);
    if (rc != QMRI_OK) return rc;
    if (a->N == 0) return QMRI_OK;
    HIP_TRY(hipSetDevice(a->device));

    const size_t es = dtype_size(a->y_dtype);
    const size_t os = a->out_dtype == QMRI_F64 ? 8 : 4;
    const long long kSlab = 1LL << 22;  // 4 Mi voxels per slab
    const long long S = a->N < kSlab ? ((a->N + 255) / 256) * 256 : kSlab;
    const int nbuf = a->N > S ? 2 : 1;
    const bool per_voxel = a->init == QMRI_INIT_PER_VOXEL;

    HostPipe &P = g_pipe[a->device];
    std::lock_guard<std::mutex> lock(P.mu);
#define HIP_TRY_C(expr)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(QMRI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
    if (!P.flag) HIP_TRY_C(hipMalloc(reinterpret_cast<void **>(&P.flag), 4));
    if (!P.d2h) HIP_TRY_C(hipStreamCreateWithFlags(&P.d2h, hipStreamNonBlocking));
    HIP_TRY_C(hipMemset(P.flag, 0, 4));
    for (int b = 0; b < nbuf; ++b) {
        SlabBuf &B = P.buf[b];
        if (!B.stream) HIP_TRY_C(hipStreamCreateWithFlags(&B.stream, hipStreamNonBlocking));
        if (!B.done) HIP_TRY_C(hipEventCreateWithFlags(&B.done, hipEventDisableTiming));
        HIP_TRY_C(ensure(&B.y, &B.cap[0], (size_t)a->E * S * es));
        if (a->popt) HIP_TRY_C(ensure(&B.popt, &B.cap[1], (size_t)S * 2 * os));
        HIP_TRY_C(ensure(&B.r2, &B.cap[2], (size_t)S * os));
        if (a->tc) HIP_TRY_C(ensure(&B.tc, &B.cap[3], (size_t)S * os));
        if (a->mask) HIP_TRY_C(ensure(reinterpret_cast<void **>(&B.mask), &B.cap[4], (size_t)S));
        if (per_voxel && a->a0v) HIP_TRY_C(ensure(reinterpret_cast<void **>(&B.a0v), &B.cap[5], (size_t)S * 8));
        if (per_voxel && a->b0v) HIP_TRY_C(ensure(reinterpret_cast<void **>(&B.b0v), &B.cap[6], (size_t)S * 8));
        if (a->info) HIP_TRY_C(ensure(reinterpret_cast<void **>(&B.info), &B.cap[7], (size_t)S));
        if (a->nfev) HIP_TRY_C(ensure(reinterpret_cast<void **>(&B.nfev), &B.cap[8], (size_t)S * 2));
    }

    // first touch of the output arrays, slab by slab in parallel, while the first slabs upload and compute
    const long long nslabs = (a->N + S - 1) / S;
    constexpr int kMaxTouchers = 16;
    static const int kTouchers = [] {
        const char *e = std::getenv("QMRI_HOST_TOUCHERS");
        const int n = e ? std::atoi(e) : 4;  // 4, 8, 16 measured equal: the OS zero-fill does not scale further
        return n < 0 ? 0 : (n > kMaxTouchers ? kMaxTouchers : n);
    }();
    std::thread touchers[kMaxTouchers];
    const bool big = kTouchers > 0 && (size_t)a->N * 2 * os >= (64u << 20);
    std::vector<std::atomic<int>> touched(big ? (size_t)nslabs : 0);
    for (auto &t : touched) t.store(0);
    if (big) {
        for (int t = 0; t < kTouchers; ++t)
            touchers[t] = std::thread([&, t] {
                for (long long j = 0; j < nslabs; ++j) {
                    const size_t s0 = (size_t)(j * S);
                    const size_t cnt = (size_t)((a->N - j * S) < S ? (a->N - j * S) : S);
                    if (a->popt) prefault(static_cast<char *>(a->popt) + s0 * 2 * os, cnt * 2 * os, t, kTouchers);
                    prefault(static_cast<char *>(a->r2) + s0 * os, cnt * os, t, kTouchers);
                    if (a->tc) prefault(static_cast<char *>(a->tc) + s0 * os, cnt * os, t, kTouchers);
                    if (a->info) prefault(reinterpret_cast<char *>(a->info) + s0, cnt, t, kTouchers);
                    if (a->nfev) prefault(reinterpret_cast<char *>(a->nfev) + s0 * 2, cnt * 2, t, kTouchers);
                    touched[(size_t)j].fetch_add(1, std::memory_order_release);
                }
            });
    }

    // downloader: slab jobs in order; job i may start when its kernel is done and its output pages are touched;
    // a device buffer is free again when its download has finished
    std::mutex qmu;
    std::condition_variable qcv;
    long long submitted = 0, downloaded = 0;
    hipError_t dl_err = hipSuccess;
    bool abort_dl = false;
    std::thread downloader([&] {
        (void)hipSetDevice(a->device);
        for (long long i = 0; i < nslabs; ++i) {
            {
                std::unique_lock<std::mutex> lk(qmu);
                qcv.wait(lk, [&] { return submitted > i || abort_dl; });
                if (abort_dl) return;
            }
            SlabBuf &B = P.buf[nbuf == 1 ? 0 : (i & 1)];
            const long long s0 = i * S;
            const long long cnt = (a->N - s0) < S ? (a->N - s0) : S;
            hipError_t e = hipEventSynchronize(B.done);
            if (big)
                while (touched[(size_t)i].load(std::memory_order_acquire) < kTouchers) std::this_thread::yield();
            if (e == hipSuccess && a->popt)
                e = hipMemcpyAsync(static_cast<char *>(a->popt) + (size_t)s0 * 2 * os, B.popt, (size_t)cnt * 2 * os,
                                   hipMemcpyDeviceToHost, P.d2h);
            if (e == hipSuccess)
                e = hipMemcpyAsync(static_cast<char *>(a->r2) + (size_t)s0 * os, B.r2, (size_t)cnt * os,
                                   hipMemcpyDeviceToHost, P.d2h);
            if (e == hipSuccess && a->tc)
                e = hipMemcpyAsync(static_cast<char *>(a->tc) + (size_t)s0 * os, B.tc, (size_t)cnt * os,
                                   hipMemcpyDeviceToHost, P.d2h);
            if (e == hipSuccess && a->info) e = hipMemcpyAsync(a->info + s0, B.info, (size_t)cnt, hipMemcpyDeviceToHost, P.d2h);
            if (e == hipSuccess && a->nfev) e = hipMemcpyAsync(a->nfev + s0, B.nfev, (size_t)cnt * 2, hipMemcpyDeviceToHost, P.d2h);
            if (e == hipSuccess) e = hipStreamSynchronize(P.d2h);
            {
                std::lock_guard<std::mutex> lk(qmu);
                if (e != hipSuccess && dl_err == hipSuccess) dl_err = e;
                downloaded = i + 1;
            }
            qcv.notify_all();
        }
    });

    int status = QMRI_OK;
    hipError_t up_err = hipSuccess;
    for (long long i = 0; i < nslabs && status == QMRI_OK && up_err == hipSuccess; ++i) {
        SlabBuf &B = P.buf[nbuf == 1 ? 0 : (i & 1)];
        const long long s0 = i * S;
        const long long cnt = (a->N - s0) < S ? (a->N - s0) : S;
        {  // this buffer's previous slab (i - nbuf) must have been downloaded
            std::unique_lock<std::mutex> lk(qmu);
            qcv.wait(lk, [&] { return downloaded >= i - nbuf + 1; });
            if (dl_err != hipSuccess) break;
        }
        hipStream_t st = B.stream;
        for (int e = 0; e < a->E && up_err == hipSuccess; ++e)
            up_err = hipMemcpyAsync(static_cast<char *>(B.y) + (size_t)e * S * es,
                                    a->y_rows ? static_cast<const char *>(a->y_rows[e]) + (size_t)s0 * es
                                              : static_cast<const char *>(a->y) + ((size_t)e * a->ld + s0) * es,
                                    (size_t)cnt * es, hipMemcpyHostToDevice, st);
        if (up_err == hipSuccess && a->mask) up_err = hipMemcpyAsync(B.mask, a->mask + s0, (size_t)cnt, hipMemcpyHostToDevice, st);
        if (up_err == hipSuccess && B.a0v && per_voxel && a->a0v)
            up_err = hipMemcpyAsync(B.a0v, a->a0v + s0, (size_t)cnt * 8, hipMemcpyHostToDevice, st);
        if (up_err == hipSuccess && B.b0v && per_voxel && a->b0v)
            up_err = hipMemcpyAsync(B.b0v, a->b0v + s0, (size_t)cnt * 8, hipMemcpyHostToDevice, st);
        if (up_err != hipSuccess) break;

        qmri_monoexp_args d = *a;
        d.y = B.y;
        d.y_rows = nullptr;
        d.ld = S;
        d.N = cnt;
        d.mask = a->mask ? B.mask : nullptr;
        d.a0v = (per_voxel && a->a0v) ? B.a0v : nullptr;
        d.b0v = (per_voxel && a->b0v) ? B.b0v : nullptr;
        d.popt = a->popt ? B.popt : nullptr;
        d.r2 = B.r2;
        d.tc = a->tc ? B.tc : nullptr;
        d.info = a->info ? B.info : nullptr;
        d.nfev = a->nfev ? B.nfev : nullptr;
        d.stream = st;
        status = launch_fit(&d, P.flag);
        if (status != QMRI_OK) break;
        up_err = hipEventRecord(B.done, st);
        if (up_err != hipSuccess) break;
        {
            std::lock_guard<std::mutex> lk(qmu);
            submitted = i + 1;
        }
        qcv.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(qmu);
        if (status != QMRI_OK || up_err != hipSuccess || dl_err != hipSuccess) abort_dl = true;
    }
    qcv.notify_all();
    downloader.join();
    if (big)
        for (int t = 0; t < kTouchers; ++t) touchers[t].join();
    if (status != QMRI_OK) {
        (void)hipDeviceSynchronize();
        return status;
    }
    if (up_err != hipSuccess || dl_err != hipSuccess) {
        (void)hipDeviceSynchronize();
        return fail(QMRI_ERR_HIP, "fit_host: %s", hipGetErrorString(up_err != hipSuccess ? up_err : dl_err));
    }
    int32_t hflag = 0;
    HIP_TRY_C(hipMemcpy(&hflag, P.flag, 4, hipMemcpyDeviceToHost));
    if (hflag) return fail(QMRI_ERR_NONFINITE, "array must not contain infs or NaNs");
    return QMRI_OK;
#undef HIP_TRY_C
}


static int linfit_validate(const qmri_linfit_args *a) {
    if (!a) return fail(QMRI_ERR_ARG, "args is NULL");
    if (!a->y || !a->x || !a->popt || !a->r2) return fail(QMRI_ERR_ARG, "y, x, popt and r2 are required");
    if (dtype_size(a->y_dtype) == 0) return fail(QMRI_ERR_ARG, "unknown y_dtype %d", a->y_dtype);
    if (a->out_dtype != QMRI_F32 && a->out_dtype != QMRI_F64)
        return fail(QMRI_ERR_ARG, "out_dtype must be QMRI_F32 or QMRI_F64");
    if (a->E < 2) return fail(QMRI_ERR_ARG, "E=%d: a line needs at least 2 samples", a->E);
    if (a->E > QMRI_MAX_ECHOES)
        return fail(QMRI_ERR_UNSUPPORTED, "E=%d exceeds QMRI_MAX_ECHOES=%d", a->E, QMRI_MAX_ECHOES);
    if (a->N < 0 || a->ld < a->N) return fail(QMRI_ERR_ARG, "need 0 <= N <= ld");
    if (a->device < 0 || a->device >= kMaxDevices) return fail(QMRI_ERR_ARG, "bad device %d", a->device);
    return QMRI_OK;
}

int qmri_linfit_device(const qmri_linfit_args *a) {
    const int rc = linfit_validate(a);
    if (rc != QMRI_OK) return rc;
    if (a->N == 0) return QMRI_OK;
    DeviceCtx *ctx = nullptr;
    HIP_TRY(hipSetDevice(a->device));
    HIP_TRY(ctx_get(a->device, &ctx));
    qmri::LinfitKArgs k;
    std::memset(&k, 0, sizeof(k));
    k.y = a->y;
    k.ld = a->ld;
    k.N = a->N;
    k.E = a->E;
    k.y_dtype = a->y_dtype;
    k.log_transform = a->log_transform;
    k.skip_rules = a->skip_rules;
    k.use_y_bounds = a->use_y_bounds;
    k.out_f64 = a->out_dtype == QMRI_F64;
    k.y_lo = a->y_lo;
    k.y_hi = a->y_hi;
    k.r2_eps = a->r2_eps;
    k.popt = a->popt;
    k.r2 = a->r2;
    double xm = 0.0;
    for (int i = 0; i < a->E; ++i) {
        k.x[i] = a->x[i];
        xm += a->x[i];
    }
    xm /= a->E;
    double sxx = 0.0;
    for (int i = 0; i < a->E; ++i) sxx += (a->x[i] - xm) * (a->x[i] - xm);
    k.xmean = xm;
    k.sxx = sxx;
    HIP_TRY(qmri::linfit_launch(k, ctx->num_cu, static_cast<hipStream_t>(a->stream)));
    return QMRI_OK;
}

int qmri_linfit_host(const qmri_linfit_args *a) {
    const int rc = linfit_validate(a);
    if (rc != QMRI_OK) return rc;
    if (a->N == 0) return QMRI_OK;
    HIP_TRY(hipSetDevice(a->device));
    const size_t es = dtype_size(a->y_dtype);
    const size_t os = a->out_dtype == QMRI_F64 ? 8 : 4;
    void *dy = nullptr, *dp = nullptr, *dr = nullptr;
    int status = QMRI_OK;
    hipError_t e = hipMalloc(&dy, (size_t)a->E * a->N * es);
    if (e == hipSuccess) e = hipMalloc(&dp, (size_t)a->N * 2 * os);
    if (e == hipSuccess) e = hipMalloc(&dr, (size_t)a->N * os);
    if (e == hipSuccess)
        e = hipMemcpy2D(dy, (size_t)a->N * es, a->y, (size_t)a->ld * es, (size_t)a->N * es, (size_t)a->E,
                        hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        qmri_linfit_args d = *a;
        d.y = dy;
        d.ld = a->N;
        d.popt = dp;
        d.r2 = dr;
        d.stream = nullptr;
        status = qmri_linfit_device(&d);
        if (status == QMRI_OK) {
            e = hipMemcpy(a->popt, dp, (size_t)a->N * 2 * os, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(a->r2, dr, (size_t)a->N * os, hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(dy);
    (void)hipFree(dp);
    (void)hipFree(dr);
    if (e != hipSuccess) return fail(QMRI_ERR_HIP, "linfit_host: %s", hipGetErrorString(e));
    return status;
}


// ---- general polynomial least squares (linfit.hip: polyls_kernel) ---------------------------------------------------
static int polyls_validate(const qmri_polyls_args *a) {
    if (!a) return fail(QMRI_ERR_ARG, "args is NULL");
    if (!a->y || !a->solve || !a->design || !a->popt || !a->r2) return fail(QMRI_ERR_ARG, "y, solve, design, popt and r2 are required");
    if (dtype_size(a->y_dtype) == 0) return fail(QMRI_ERR_ARG, "unknown y_dtype %d", a->y_dtype);
    // (beyond QMRI_POLY_MAX_PARAMS parameters / QMRI_MAX_ECHOES samples the streaming variant of the kernel runs)
    if (a->P < 1 || a->P > 4096) return fail(QMRI_ERR_ARG, "deg + 1 must be in [1, 4096]");
    if (a->E < 1 || a->E > (1 << 20)) return fail(QMRI_ERR_ARG, "E must be in [1, 2^20]");
    if (a->N < 0 || a->ld < a->N) return fail(QMRI_ERR_ARG, "need 0 <= N <= ld");
    return QMRI_OK;
}

int qmri_polyls_device(const qmri_polyls_args *a) {
    const int rc = polyls_validate(a);
    if (rc != QMRI_OK) return rc;
    if (a->N == 0) return QMRI_OK;
    DeviceCtx *ctx = nullptr;
    HIP_TRY(hipSetDevice(a->device));
    HIP_TRY(ctx_get(a->device, &ctx));
    hipStream_t stream = static_cast<hipStream_t>(a->stream);
    const int P = a->P, E = a->E;
    std::vector<double> ops((size_t)2 * P * E + E);
    std::memcpy(ops.data(), a->solve, sizeof(double) * P * E);
    std::memcpy(ops.data() + (size_t)P * E, a->design, sizeof(double) * P * E);
    for (int e = 0; e < E; ++e) ops[(size_t)2 * P * E + e] = a->w ? a->w[e] : 1.0;
    AsyncScratch dops;
    HIP_TRY(dops.alloc(ops.size() * sizeof(double), stream));
    HIP_TRY(hipMemcpyAsync(dops.p, ops.data(), ops.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));  // `ops` is a pageable stack-lifetime buffer: the copy must have read it
    qmri::PolylsKArgs k;
    std::memset(&k, 0, sizeof(k));
    k.y = a->y; k.ld = a->ld; k.N = a->N; k.E = E; k.P = P; k.y_dtype = a->y_dtype;
    k.skip_rules = a->skip_rules; k.use_y_bounds = a->use_y_bounds; k.y_lo = a->y_lo; k.y_hi = a->y_hi;
    k.r2_eps = a->r2_eps;
    k.ops = static_cast<const double *>(dops.p);
    k.popt = a->popt; k.r2 = a->r2; k.resid = a->resid;
    HIP_TRY(qmri::polyls_launch(k, ctx->num_cu, stream));
    return QMRI_OK;
}

int qmri_polyls_host(const qmri_polyls_args *a) {
    const int rc = polyls_validate(a);
    if (rc != QMRI_OK) return rc;
    if (a->N == 0) return QMRI_OK;
    HIP_TRY(hipSetDevice(a->device));
    const size_t es = dtype_size(a->y_dtype);
    const size_t N = (size_t)a->N;
    void *dy = nullptr;
    double *dp = nullptr, *dr = nullptr, *ds = nullptr;
    int status = QMRI_OK;
    hipError_t e = hipMalloc(&dy, (size_t)a->E * N * es);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dp), N * a->P * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dr), N * 8);
    if (e == hipSuccess && a->resid) e = hipMalloc(reinterpret_cast<void **>(&ds), N * 8);
    if (e == hipSuccess)
        e = hipMemcpy2D(dy, N * es, a->y, (size_t)a->ld * es, N * es, (size_t)a->E, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        qmri_polyls_args d = *a;
        d.y = dy; d.ld = a->N; d.popt = dp; d.r2 = dr; d.resid = ds; d.stream = nullptr;
        status = qmri_polyls_device(&d);
        if (status == QMRI_OK) {
            e = hipMemcpy(a->popt, dp, N * a->P * 8, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(a->r2, dr, N * 8, hipMemcpyDeviceToHost);
            if (e == hipSuccess && a->resid) e = hipMemcpy(a->resid, ds, N * 8, hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(dy);
    (void)hipFree(dp);
    (void)hipFree(dr);
    (void)hipFree(ds);
    if (e != hipSuccess) return fail(QMRI_ERR_HIP, "polyls_host: %s", hipGetErrorString(e));
    return status;
}

// ---- general lmdif (lm_generic.hip) ---------------------------------------------------------------------
void qmri_lmfit_defaults(qmri_lmfit_args *a) {
    if (!a) return;
    std::memset(a, 0, sizeof(*a));
    a->model = QMRI_MODEL_BIEXP;
    a->y_dtype = QMRI_F32;
    a->ftol = 1e-5;  // dosma/core/fitting.py:761-763
    a->xtol = 1.49012e-8;
    a->gtol = 0.0;
    a->factor = 100.0;
    a->epsfcn = 2.220446049250313e-16;
    a->r2_eps = 1e-8;
    a->maxfev = 100;
    a->y_lo = -INFINITY;
    a->y_hi = INFINITY;
    for (int j = 0; j < QMRI_LM_MAX_PARAMS; ++j) a->p0[j] = 1.0;  // scipy: ones(n) when p0 is None
}

static int lmfit_validate(const qmri_lmfit_args *a) {
    if (!a) return fail(QMRI_ERR_ARG, "args is NULL");
    const int np = qmri::lm_generic_nparams(a->model);
    if (np == 0) return fail(QMRI_ERR_UNSUPPORTED, "model %d is not implemented", a->model);
    if (!a->y || !a->x || !a->popt || !a->r2) return fail(QMRI_ERR_ARG, "y, x, popt and r2 are required");
    if (dtype_size(a->y_dtype) == 0) return fail(QMRI_ERR_ARG, "unknown y_dtype %d", a->y_dtype);
    if (a->E < np)  // scipy: "The number of func parameters must not exceed the number of data points"
        return fail(QMRI_ERR_ARG, "E=%d: need at least as many samples as parameters (%d)", a->E, np);
    if (a->E > QMRI_LM_MAX_ECHOES || (np + 3) * a->E > 320)  // the kernel's lane-private columns: (n + 3) * E * 512 B of LDS <= 160 KB
        return fail(QMRI_ERR_UNSUPPORTED, "E=%d exceeds what the general kernel holds for %d parameters (E <= %d)", a->E, np,
                    320 / (np + 3) < QMRI_LM_MAX_ECHOES ? 320 / (np + 3) : QMRI_LM_MAX_ECHOES);
    if (a->N < 0 || a->ld < a->N) return fail(QMRI_ERR_ARG, "need 0 <= N <= ld");
    if (a->maxfev <= 0 || !(a->ftol >= 0) || !(a->xtol >= 0) || !(a->gtol >= 0) || !(a->factor > 0))
        return fail(QMRI_ERR_ARG, "maxfev > 0, ftol/xtol/gtol >= 0 and factor > 0 are required");
    if (a->device < 0 || a->device >= kMaxDevices) return fail(QMRI_ERR_ARG, "bad device %d", a->device);
    return QMRI_OK;
}

int qmri_lmfit_device(const qmri_lmfit_args *a, int32_t *nonfinite_flag) {
    const int rc = lmfit_validate(a);
    if (rc != QMRI_OK) return rc;
    if (a->N == 0) return QMRI_OK;
    DeviceCtx *ctx = nullptr;
    HIP_TRY(hipSetDevice(a->device));
    HIP_TRY(ctx_get(a->device, &ctx));
    hipStream_t stream = static_cast<hipStream_t>(a->stream);
    qmri::LmKArgs k;
    std::memset(&k, 0, sizeof(k));
    k.y = a->y;
    k.ld = a->ld;
    k.N = a->N;
    k.E = a->E;
    k.y_dtype = a->y_dtype;
    k.maxfev = a->maxfev;
    k.use_y_bounds = a->use_y_bounds;
    k.y_lo = a->y_lo;
    k.y_hi = a->y_hi;
    for (int j = 0; j < QMRI_LM_MAX_PARAMS; ++j) {
        k.p0[j] = a->p0[j];
        k.p0v[j] = a->p0v[j];
    }
    k.ftol = a->ftol;
    k.xtol = a->xtol;
    k.gtol = a->gtol;
    k.factor = a->factor;
    k.epsfcn = a->epsfcn;
    k.r2_eps = a->r2_eps;
    k.popt = a->popt;
    k.r2 = a->r2;
    k.info = a->info;
    k.nfev = a->nfev;
    for (int i = 0; i < a->E; ++i) k.x[i] = a->x[i];
    AsyncScratch scratch;  // per-launch: [0, 8) the pull counter, [8, 12) the flag word when the caller does not collect it
    HIP_TRY(scratch.alloc(64, stream));
    HIP_TRY(hipMemsetAsync(scratch.p, 0, 16, stream));
    unsigned long long *counter = static_cast<unsigned long long *>(scratch.p);
    k.nonfinite = nonfinite_flag ? nonfinite_flag : reinterpret_cast<int *>(counter + 1);
    HIP_TRY(qmri::lm_generic_launch(k, a->model, ctx->num_cu, counter, stream));
    return QMRI_OK;
}

int qmri_lmfit_host(const qmri_lmfit_args *a) {
    const int rc = lmfit_validate(a);
    if (rc != QMRI_OK) return rc;
    if (a->N == 0) return QMRI_OK;
    HIP_TRY(hipSetDevice(a->device));
    const int np = qmri::lm_generic_nparams(a->model);
    const size_t es = dtype_size(a->y_dtype);
    const size_t N = (size_t)a->N;
    void *dy = nullptr;
    double *dp = nullptr, *dr = nullptr, *dv[QMRI_LM_MAX_PARAMS] = {nullptr, nullptr, nullptr, nullptr};
    int8_t *di = nullptr;
    int16_t *dn = nullptr;
    int32_t *dflag = nullptr;
    int status = QMRI_OK;
    int32_t hflag = 0;
    hipError_t e = hipMalloc(&dy, (size_t)a->E * N * es);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dp), N * np * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dr), N * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dflag), 4);
    if (e == hipSuccess) e = hipMemset(dflag, 0, 4);
    if (e == hipSuccess && a->info) e = hipMalloc(reinterpret_cast<void **>(&di), N);
    if (e == hipSuccess && a->nfev) e = hipMalloc(reinterpret_cast<void **>(&dn), N * 2);
    for (int j = 0; j < np && e == hipSuccess; ++j) {
        if (!a->p0v[j]) continue;
        e = hipMalloc(reinterpret_cast<void **>(&dv[j]), N * 8);
        if (e == hipSuccess) e = hipMemcpy(dv[j], a->p0v[j], N * 8, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess)
        e = hipMemcpy2D(dy, N * es, a->y, (size_t)a->ld * es, N * es, (size_t)a->E, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        qmri_lmfit_args d = *a;
        d.y = dy;
        d.ld = a->N;
        d.popt = dp;
        d.r2 = dr;
        d.info = di;
        d.nfev = dn;
        d.stream = nullptr;
        for (int j = 0; j < QMRI_LM_MAX_PARAMS; ++j) d.p0v[j] = dv[j];
        status = qmri_lmfit_device(&d, dflag);
        if (status == QMRI_OK) {
            e = hipMemcpy(a->popt, dp, N * np * 8, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(a->r2, dr, N * 8, hipMemcpyDeviceToHost);
            if (e == hipSuccess && a->info) e = hipMemcpy(a->info, di, N, hipMemcpyDeviceToHost);
            if (e == hipSuccess && a->nfev) e = hipMemcpy(a->nfev, dn, N * 2, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(&hflag, dflag, 4, hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(dy);
    (void)hipFree(dp);
    (void)hipFree(dr);
    (void)hipFree(di);
    (void)hipFree(dn);
    (void)hipFree(dflag);
    for (int j = 0; j < QMRI_LM_MAX_PARAMS; ++j) (void)hipFree(dv[j]);
    if (e != hipSuccess) return fail(QMRI_ERR_HIP, "lmfit_host: %s", hipGetErrorString(e));
    if (status == QMRI_OK && hflag) return fail(QMRI_ERR_NONFINITE, "array must not contain infs or NaNs");
    return status;
}


static int dess_fill(const qmri_dess_args *a, qmri::DessKArgs &k) {
    if (!a) return fail(QMRI_ERR_ARG, "args is NULL");
    if (!a->echo1 || !a->echo2 || !a->t2) return fail(QMRI_ERR_ARG, "echo1, echo2 and t2 are required");
    if (dtype_size(a->dtype) == 0) return fail(QMRI_ERR_ARG, "unknown dtype %d", a->dtype);
    if (a->out_dtype != QMRI_F32 && a->out_dtype != QMRI_F64) return fail(QMRI_ERR_ARG, "bad out_dtype");
    if (a->N < 0) return fail(QMRI_ERR_ARG, "N < 0");
    if (a->device < 0 || a->device >= kMaxDevices) return fail(QMRI_ERR_ARG, "bad device %d", a->device);
    std::memset(&k, 0, sizeof(k));
    k.echo1 = a->echo1;
    k.echo2 = a->echo2;
    k.N = a->N;
    k.c0 = a->c0;
    k.k = a->k;
    k.c1 = a->c1;
    k.use_bounds = a->use_bounds;
    k.use_nan_to_num = a->use_nan_to_num;
    k.lo = a->lo;
    k.hi = a->hi;
    k.nan_value = a->nan_value;
    k.decimals = a->decimals <= -1000000 ? QMRI_NO_ROUND : a->decimals;
    k.p10 = k.decimals == QMRI_NO_ROUND ? 1.0 : std::pow(10.0, std::abs(k.decimals));
    k.ip10 = 1.0 / k.p10;
    k.suppress_fat = a->suppress_fat;
    k.suppress_fluid = a->suppress_fluid;
    k.out_f64 = a->out_dtype == QMRI_F64;
    k.beta = a->beta;
    k.t2 = a->t2;
    k.vec_ok = ((reinterpret_cast<uintptr_t>(a->echo1) | reinterpret_cast<uintptr_t>(a->echo2) |
                 reinterpret_cast<uintptr_t>(a->t2)) % 32) == 0;
    return QMRI_OK;
}

int qmri_dess_t2_device(const qmri_dess_args *a) {
    qmri::DessKArgs k;
    const int rc = dess_fill(a, k);
    if (rc != QMRI_OK) return rc;
    if (a->N == 0) return QMRI_OK;
    DeviceCtx *ctx = nullptr;
    HIP_TRY(hipSetDevice(a->device));
    HIP_TRY(ctx_get(a->device, &ctx));
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    AsyncScratch scratch;
    if (a->suppress_fat || a->suppress_fluid) HIP_TRY(scratch.alloc((2 * 1024 + 2) * 8, st));
    HIP_TRY(qmri::dess_t2_launch(k, a->dtype, ctx->num_cu, static_cast<double *>(scratch.p), st));
    return QMRI_OK;
}

int qmri_dess_t2_host(const qmri_dess_args *a) {
    qmri::DessKArgs k;
    const int rc = dess_fill(a, k);
    if (rc != QMRI_OK) return rc;
    if (a->N == 0) return QMRI_OK;
    HIP_TRY(hipSetDevice(a->device));
    const size_t es = dtype_size(a->dtype), os = a->out_dtype == QMRI_F64 ? 8 : 4;
    void *d1 = nullptr, *d2 = nullptr, *dt = nullptr;
    hipError_t e = hipMalloc(&d1, (size_t)a->N * es);
    if (e == hipSuccess) e = hipMalloc(&d2, (size_t)a->N * es);
    if (e == hipSuccess) e = hipMalloc(&dt, (size_t)a->N * os);
    if (e == hipSuccess) e = hipMemcpy(d1, a->echo1, (size_t)a->N * es, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d2, a->echo2, (size_t)a->N * es, hipMemcpyHostToDevice);
    int status = QMRI_OK;
    if (e == hipSuccess) {
        qmri_dess_args d = *a;
        d.echo1 = d1;
        d.echo2 = d2;
        d.t2 = dt;
        d.stream = nullptr;
        status = qmri_dess_t2_device(&d);
        if (status == QMRI_OK) e = hipDeviceSynchronize();
        if (status == QMRI_OK && e == hipSuccess) e = hipMemcpy(a->t2, dt, (size_t)a->N * os, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d1);
    (void)hipFree(d2);
    (void)hipFree(dt);
    if (e != hipSuccess) return fail(QMRI_ERR_HIP, "dess_t2_host: %s", hipGetErrorString(e));
    return status;
}

int qmri_rss_host(const void *echo1, const void *echo2, int32_t dtype, int64_t N, int32_t mode, double *out,
                  int32_t device) {
    if (!echo1 || !echo2 || !out) return fail(QMRI_ERR_ARG, "NULL argument");
    if (dtype_size(dtype) == 0) return fail(QMRI_ERR_ARG, "unknown dtype %d", dtype);
    if (mode != 0 && mode != 1) return fail(QMRI_ERR_ARG, "mode must be 0 (rss) or 1 (rms)");
    if (N <= 0) return N == 0 ? QMRI_OK : fail(QMRI_ERR_ARG, "N < 0");
    DeviceCtx *ctx = nullptr;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(ctx_get(device, &ctx));
    const size_t es = dtype_size(dtype);
    void *d1 = nullptr, *d2 = nullptr;
    double *dout = nullptr;
    hipError_t e = hipMalloc(&d1, (size_t)N * es);
    if (e == hipSuccess) e = hipMalloc(&d2, (size_t)N * es);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dout), (size_t)N * 8);
    if (e == hipSuccess) e = hipMemcpy(d1, echo1, (size_t)N * es, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d2, echo2, (size_t)N * es, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = qmri::rss_launch(d1, d2, dtype, N, mode, dout, ctx->num_cu, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)N * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d1);
    (void)hipFree(d2);
    (void)hipFree(dout);
    if (e != hipSuccess) return fail(QMRI_ERR_HIP, "rss_host: %s", hipGetErrorString(e));
    return QMRI_OK;
}

int qmri_region_stats_host(const qmri_region_stats_args *a) {
    if (!a || !a->values || !a->out) return fail(QMRI_ERR_ARG, "NULL argument");
    if (a->v_dtype != QMRI_F32 && a->v_dtype != QMRI_F64) return fail(QMRI_ERR_ARG, "values must be f32 or f64");
    if (a->N < 0) return fail(QMRI_ERR_ARG, "N < 0");
    const int nkeys = a->labels ? a->nkeys : 0;
    if (nkeys < 0 || nkeys > QMRI_MAX_REGIONS - 1) return fail(QMRI_ERR_ARG, "nkeys must be 0..%d", QMRI_MAX_REGIONS - 1);
    if (nkeys > 0 && !a->label_keys) return fail(QMRI_ERR_ARG, "label_keys is NULL");
    if (a->use_bounds && (a->closed < 0 || a->closed > 3)) return fail(QMRI_ERR_ARG, "closed must be 0..3");
    if (a->labels && (a->l_kind < 0 || a->l_kind > 2)) return fail(QMRI_ERR_ARG, "l_kind must be 0 (int32), 1 (uint8) or 2 (int16)");
    const size_t ls = a->l_kind == 0 ? 4 : (a->l_kind == 1 ? 1 : 2);
    DeviceCtx *ctx = nullptr;
    HIP_TRY(hipSetDevice(a->device));
    HIP_TRY(ctx_get(a->device, &ctx));
    const size_t es = a->v_dtype == QMRI_F64 ? 8 : 4;
    const size_t n = (size_t)(a->N > 0 ? a->N : 1);
    void *dv = nullptr, *dstate = nullptr, *dl = nullptr;
    double *dout = nullptr;
    hipError_t e = hipMalloc(&dv, n * es);
    if (e == hipSuccess && a->labels) e = hipMalloc(&dl, n * ls);
    if (e == hipSuccess) e = hipMalloc(&dstate, qmri::region_stats_state_bytes());
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dout), (size_t)(nkeys + 1) * 4 * 8);
    if (e == hipSuccess && a->N > 0) e = hipMemcpy(dv, a->values, (size_t)a->N * es, hipMemcpyHostToDevice);
    if (e == hipSuccess && a->labels && a->N > 0) e = hipMemcpy(dl, a->labels, (size_t)a->N * ls, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = qmri::region_stats_launch(dv, a->v_dtype == QMRI_F64, dl, a->l_kind, a->N, nkeys, a->label_keys, a->use_bounds, a->lo, a->hi,
                                      a->closed, dstate, dout, ctx->num_cu, nullptr);
    if (e == hipSuccess) e = hipMemcpy(a->out, dout, (size_t)(nkeys + 1) * 4 * 8, hipMemcpyDeviceToHost);
    (void)hipFree(dv);
    (void)hipFree(dl);
    (void)hipFree(dstate);
    (void)hipFree(dout);
    if (e != hipSuccess) return fail(QMRI_ERR_HIP, "region_stats_host: %s", hipGetErrorString(e));
    return QMRI_OK;
}

int qmri_region_stats_device(const qmri_region_stats_args *a, void *hip_stream) {
    if (!a || !a->values || !a->out) return fail(QMRI_ERR_ARG, "NULL argument");
    if (a->v_dtype != QMRI_F32 && a->v_dtype != QMRI_F64) return fail(QMRI_ERR_ARG, "values must be f32 or f64");
    if (a->N < 0) return fail(QMRI_ERR_ARG, "N < 0");
    const int nkeys = a->labels ? a->nkeys : 0;
    if (nkeys < 0 || nkeys > QMRI_MAX_REGIONS - 1) return fail(QMRI_ERR_ARG, "nkeys must be 0..%d", QMRI_MAX_REGIONS - 1);
    if (nkeys > 0 && !a->label_keys) return fail(QMRI_ERR_ARG, "label_keys is NULL");
    if (a->use_bounds && (a->closed < 0 || a->closed > 3)) return fail(QMRI_ERR_ARG, "closed must be 0..3");
    if (a->labels && (a->l_kind < 0 || a->l_kind > 2)) return fail(QMRI_ERR_ARG, "l_kind must be 0 (int32), 1 (uint8) or 2 (int16)");
    DeviceCtx *ctx = nullptr;
    HIP_TRY(hipSetDevice(a->device));
    HIP_TRY(ctx_get(a->device, &ctx));
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    AsyncScratch state, dout;
    HIP_TRY(state.alloc(qmri::region_stats_state_bytes(), st));
    HIP_TRY(dout.alloc((size_t)(nkeys + 1) * 4 * 8, st));
    HIP_TRY(qmri::region_stats_launch(a->values, a->v_dtype == QMRI_F64, a->labels, a->l_kind, a->N, nkeys, a->label_keys,
                                      a->use_bounds, a->lo, a->hi, a->closed, state.p, static_cast<double *>(dout.p),
                                      ctx->num_cu, st));
    HIP_TRY(hipMemcpyAsync(a->out, dout.p, (size_t)(nkeys + 1) * 4 * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return QMRI_OK;
}

}  // extern "C"
