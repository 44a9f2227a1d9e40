// csr_host_single.inl -- part of csr_lib.hip (one translation unit; included in this order): reference-shaped single-chain entry points on host buffers (default context)
// clang-format off is NOT needed; this file is plain C++/HIP host code.

// ---------------------------------------------------------------------------------------------------------------
// (1) reference-shaped single-chain entry points on host buffers (default context, device from
//     CONSENRICH_AMD_DEVICE or 0)
// ---------------------------------------------------------------------------------------------------------------
static csr_ctx *g_default = nullptr;
// The reference-shaped entry points share ONE default context (device buffers cached across calls): they serialise on
// this mutex, so calls from several host threads are safe (the reference's callers are single-threaded per chromosome,
// core.py:3290, but run beside thread pools).  Batch contexts are independent; one thread per context.
static std::recursive_mutex g_defaultMutex;
#define DEFAULT_CTX_GUARD std::lock_guard<std::recursive_mutex> guard_(g_defaultMutex)
static csr_ctx *default_ctx() {
    DEFAULT_CTX_GUARD;          // creation is serialised too (csr_set_validation(NULL) / csr_profile_enable(NULL) come here unlocked)
    if (!g_default) {
        int dev = 0;
        if (const char *e = getenv("CONSENRICH_AMD_DEVICE")) dev = atoi(e);
        g_default = csr_create(dev);
        // the drop-in single-chain callables favour parity: bit-exact sequential semantics unless told otherwise
        if (g_default && !getenv("CONSENRICH_AMD_XTOL_ULPS")) {
            g_default->xTolUlps = 0;
            mode_warm_defaults(g_default);
        }
    }
    return g_default;
}

// Freshly allocated NumPy outputs are unmapped pages: a D2H copy into them runs at the first-touch page-fault rate
// (~10 GB/s on one thread, scripts/ubench/pcie.hip) -- five times slower than the PCIe copy itself, and at chr1 x 32 the
// residual matrix alone is 159 MB.  A few host threads touch the pages (read a byte, write it back: contents unchanged)
// WHILE the upload and the kernels run, so the downloads land in mapped memory.  CONSENRICH_AMD_PREFAULT_THREADS=0 disables.
struct Prefault {
    std::vector<std::thread> th;
    int nthreads = 8;
    Prefault() {
        if (const char *e = getenv("CONSENRICH_AMD_PREFAULT_THREADS")) nthreads = atoi(e);
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && nthreads > hw) nthreads = hw;
    }
    void add(void *ptr, size_t bytes) {
        if (!ptr || nthreads <= 0 || bytes < ((size_t)4 << 20)) return;
        char *base = static_cast<char *>(ptr);
        const int nt = (int)std::min<size_t>((size_t)nthreads, bytes / ((size_t)8 << 20) + 1);      // >= 8 MB per thread
        const size_t per = ((bytes + nt - 1) / nt + 4095) / 4096 * 4096;
        for (int t = 0; t < nt; ++t) {
            const size_t lo = per * (size_t)t;
            if (lo >= bytes) break;
            const size_t hi = std::min(bytes, lo + per);
            th.emplace_back([base, lo, hi] {
                for (size_t o = lo; o < hi; o += 4096) {
                    volatile char *q = base + o;
                    const char v = *q;
                    *q = v;
                }
                volatile char *last = base + (hi - 1);
                const char v = *last;
                *last = v;
            });
        }
    }
    void join() {
        for (auto &t : th) t.join();
        th.clear();
    }
    ~Prefault() { join(); }
};

static int configure_single(csr_ctx *c, const csr_model *mdl, int64_t m, int64_t n) {
    const bool reuse = c->configured && c->chains.size() == 1 && c->chains[0].n == n && c->m == m &&
                       c->mdl.state_dim == mdl->state_dim;
    if (reuse) return csr_batch_set_model(c, mdl);
    return csr_batch_configure(c, mdl, m, 1, &n);
}

static int import_nat(csr_ctx *c, const float *host, int ncomp, int64_t rows, int64_t rowShift, float *dst, int dstStride) {
    // host natural (rows, ncomp) -> blocked float slots (dstStride floats per slot), bins [rowShift, rowShift+rows)
    float *scr;
    CHECK(nat_array(c, CSR_ARR_PS, &scr));   // 4 (or 1) floats per bin of natural scratch
    if (rows > 0)
        HIPOK(hipMemcpyAsync(scr + (c->chains[0].off + rowShift) * ncomp, host, sizeof(float) * ncomp * rows,
                             hipMemcpyHostToDevice, c->stream));
    for (int k = 0; k < ncomp; ++k) {
        hipLaunchKernelGGL(k_import_f32, dim3(grid_slots(c)), dim3(256), 0, c->stream, c->p, scr, ncomp, k, dst, dstStride, k);
        LAUNCH_CHECK("k_import_f32");
    }
    return 0;
}

extern "C" int csr_forward_pass(const csr_model *mdl, const csr_fwd_io *io, csr_fwd_out *out) {
    DEFAULT_CTX_GUARD;
    if (!mdl || !io || !out) return fail("null argument");
    if (io->m <= 0 || io->n <= 0) return fail("empty input must be handled by the caller (pyx:6494-6501)");
    if (!io->data || !io->munc || !io->D) return fail("null host buffer");
    if ((io->flags & CSR_USE_LAMBDA) && !io->lambda) return fail("CSR_USE_LAMBDA without lambda");
    if ((io->flags & CSR_USE_KAPPA) && !io->kappa) return fail("CSR_USE_KAPPA without kappa");
    if ((io->flags & CSR_USE_QSCALE) && !io->qscale) return fail("CSR_USE_QSCALE without qscale");
    csr_ctx *c = default_ctx();
    if (!c) return -1;
    const int dd = mdl->state_dim == 1 ? 1 : 2;
    Prefault pf;
    pf.add(io->D, sizeof(float) * (size_t)io->n);
    if (io->xf && io->Pf && io->pnoise) {
        pf.add(io->xf, sizeof(float) * (size_t)io->n * dd);
        pf.add(io->Pf, sizeof(float) * (size_t)io->n * dd * dd);
        if (io->n > 1) pf.add(io->pnoise, sizeof(float) * (size_t)(io->n - 1) * dd * dd);
    }
    CHECK(configure_single(c, mdl, io->m, io->n));
    CHECK(csr_batch_upload(c, 0, io->data, io->munc));
    CHECK(csr_batch_upload_multipliers(c, 0, (io->flags & CSR_USE_LAMBDA) ? io->lambda : nullptr,
                                       (io->flags & CSR_USE_KAPPA) ? io->kappa : nullptr,
                                       (io->flags & CSR_USE_QSCALE) ? io->qscale : nullptr));
    CHECK(csr_batch_stats(c));
    CHECK(csr_batch_forward(c, io->flags, &out->sum_d, &out->sum_nll));
    CHECK(csr_batch_export(c, CSR_EXPORT_FORWARD));
    pf.join();
    CHECK(csr_batch_download(c, 0, CSR_ARR_D, io->D));
    if (io->xf) {
        if (!io->Pf || !io->pnoise) return fail("xf/Pf/pnoise must be given together");
        CHECK(csr_batch_download(c, 0, CSR_ARR_XF, io->xf));
        CHECK(csr_batch_download(c, 0, CSR_ARR_PF, io->Pf));
        CHECK(csr_batch_download(c, 0, CSR_ARR_PNOISE, io->pnoise));
    }
    return 0;
}

extern "C" int csr_backward_pass(const csr_model *mdl, int64_t m, int64_t n, const float *data, const float *xf,
                                 const float *Pf, const float *pnoise, float *xs, float *Ps, float *lag,
                                 int64_t lag_rows, float *resid) {
    DEFAULT_CTX_GUARD;
    if (!mdl || !data || !xf || !Pf || !pnoise || !xs || !Ps || !lag || !resid) return fail("null argument");
    if (m <= 0 || n <= 0) return fail("empty input must be handled by the caller (pyx:6737)");
    if (lag_rows < std::max<int64_t>(n - 1, 1)) return fail("lagCovSmoothed too small");
    csr_ctx *c = default_ctx();
    if (!c) return -1;
    const int dd = mdl->state_dim == 1 ? 1 : 2;
    Prefault pf;
    pf.add(resid, sizeof(float) * (size_t)n * (size_t)m);
    pf.add(xs, sizeof(float) * (size_t)n * dd);
    pf.add(Ps, sizeof(float) * (size_t)n * dd * dd);
    if (n > 1) pf.add(lag, sizeof(float) * (size_t)(n - 1) * dd * dd);
    CHECK(configure_single(c, mdl, m, n));
    // only `data` matters for the smoother (residuals); munc is not an input of cbackwardPass
    const ChainInfo &ci = c->chains[0];
    HIPOK(hipMemcpy2DAsync(const_cast<float *>(c->p.data) + ci.off, sizeof(float) * c->Npad, data, sizeof(float) * n,
                           sizeof(float) * n, (size_t)m, hipMemcpyHostToDevice, c->stream));
    c->statsValid = false;
    const int d = mdl->state_dim;
    CHECK(import_nat(c, xf, d, n, 0, (float *)c->p.tXf, 2));
    CHECK(import_nat(c, Pf, d * d, n, 0, (float *)c->p.tPf, 4));
    if (n > 1) CHECK(import_nat(c, pnoise, d * d, n - 1, 0, (float *)c->p.tQ, 4));
    c->haveFwd = true;
    c->fwdInternal = false;
    c->fwdQCompact = false;
    // the caller's xf / Pf ARE the blocked copies now: whatever a previous resident pass left in the reference layout is not theirs
    c->fwdBlockedStale = c->pfBlockedStale = false;
    c->fwdNat = c->xfNat = c->pfNat = c->pnNat = false;
    c->fwdFlags = 0;
    CHECK(backward_impl(c, true, nullptr));
    CHECK(csr_batch_export(c, CSR_EXPORT_SMOOTH | CSR_EXPORT_RESID));
    pf.join();
    CHECK(csr_batch_download(c, 0, CSR_ARR_XS, xs));
    CHECK(csr_batch_download(c, 0, CSR_ARR_PS, Ps));
    CHECK(csr_batch_download(c, 0, CSR_ARR_LAG, lag));
    CHECK(csr_batch_download(c, 0, CSR_ARR_RESID, resid));
    c->haveFwd = c->haveBwd = false;   // imported filter results are not a reusable forward pass
    return 0;
}

extern "C" int csr_fixed_background_ecm(const csr_model *mdl, const csr_ecm_cfg *cfg, int64_t m, int64_t n,
                                        const float *data, const float *munc, const float *qscale, float *lambda,
                                        float *kappa, float *xs, float *Ps, float *lag, float *resid,
                                        double *nll_path, csr_ecm_out *out) {
    DEFAULT_CTX_GUARD;
    if (!mdl || !cfg || !data || !munc || !xs || !Ps || !lag || !resid || !out) return fail("null argument");
    if (m <= 0 || n <= 0) return fail("empty input must be handled by the caller (pyx:7999)");
    if (cfg->use_lambda && !lambda) return fail("use_lambda without lambda buffer");
    if (cfg->use_kappa && !kappa) return fail("use_kappa without kappa buffer");
    csr_ctx *c = default_ctx();
    if (!c) return -1;
    const int dd = mdl->state_dim == 1 ? 1 : 2;
    Prefault pf;
    pf.add(resid, sizeof(float) * (size_t)n * (size_t)m);
    pf.add(xs, sizeof(float) * (size_t)n * dd);
    pf.add(Ps, sizeof(float) * (size_t)n * dd * dd);
    if (n > 1) pf.add(lag, sizeof(float) * (size_t)(n - 1) * dd * dd);
    CHECK(configure_single(c, mdl, m, n));
    CHECK(csr_batch_upload(c, 0, data, munc));
    CHECK(csr_batch_upload_multipliers(c, 0, cfg->use_lambda ? lambda : nullptr, cfg->use_kappa ? kappa : nullptr, qscale));
    CHECK(csr_batch_stats(c));
    CHECK(csr_batch_ecm(c, cfg, qscale ? CSR_USE_QSCALE : 0u, out, nll_path));
    CHECK(csr_batch_export(c, CSR_EXPORT_SMOOTH | CSR_EXPORT_RESID | CSR_EXPORT_MULT));
    pf.join();
    CHECK(csr_batch_download(c, 0, CSR_ARR_XS, xs));
    CHECK(csr_batch_download(c, 0, CSR_ARR_PS, Ps));
    CHECK(csr_batch_download(c, 0, CSR_ARR_LAG, lag));
    CHECK(csr_batch_download(c, 0, CSR_ARR_RESID, resid));
    if (cfg->use_lambda) CHECK(csr_batch_download(c, 0, CSR_ARR_LAMBDA, lambda));
    if (cfg->use_kappa) CHECK(csr_batch_download(c, 0, CSR_ARR_KAPPA, kappa));
    return 0;
}

extern "C" int csr_expected_transition_residual_sums(int32_t state_dim, int64_t n, const double *xs, const double *Ps,
                                                     const double *lag, const double *F, double *sum_level,
                                                     double *sum_trend, int64_t *count) {
    DEFAULT_CTX_GUARD;
    if (!sum_level || !sum_trend || !count) return fail("null argument");
    *sum_level = 0.0; *sum_trend = 0.0;
    *count = n - 1 > 0 ? n - 1 : 0;
    if (n - 1 <= 0) return 0;
    if (!xs || !Ps || !lag) return fail("null host buffer");
    if (state_dim == 2 && !F) return fail("matrixF required for the levelTrend model");
    csr_ctx *c = default_ctx();
    if (!c) return -1;
    CHECK(ctx_select(c));
    const int d = state_dim;
    double *dxs = nullptr, *dPs = nullptr, *dlag = nullptr, *dpart = nullptr;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 512);
    auto cleanup = [&]() { (void)hipFree(dxs); (void)hipFree(dPs); (void)hipFree(dlag); (void)hipFree(dpart); };
    if (hipMalloc((void **)&dxs, sizeof(double) * n * d) != hipSuccess || hipMalloc((void **)&dPs, sizeof(double) * n * d * d) != hipSuccess ||
        hipMalloc((void **)&dlag, sizeof(double) * (n - 1) * d * d) != hipSuccess ||
        hipMalloc((void **)&dpart, sizeof(double) * 2 * grid) != hipSuccess) {
        cleanup();
        return fail("hipMalloc failed in transition sums");
    }
    if (hipMemcpyAsync(dxs, xs, sizeof(double) * n * d, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(dPs, Ps, sizeof(double) * n * d * d, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(dlag, lag, sizeof(double) * (n - 1) * d * d, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
        cleanup();
        return fail("hipMemcpyAsync (transition sums inputs) failed");
    }
    {
        Scope sc(c, "transition_sums");
        hipLaunchKernelGGL(k_tsums, dim3(grid), dim3(256), 0, c->stream, d, n, dxs, dPs, dlag, d == 2 ? F[0] : 1.0,
                           d == 2 ? F[1] : 0.0, d == 2 ? F[2] : 0.0, d == 2 ? F[3] : 1.0, dpart, dpart + grid);
    }
    std::vector<double> part(2 * grid);
    hipError_t e = hipMemcpyAsync(part.data(), dpart, sizeof(double) * 2 * grid, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return fail("transition sums failed: %s", hipGetErrorString(e));
    double aL = 0.0, aT = 0.0;
    for (int i = 0; i < grid; ++i) { aL += part[i]; aT += part[grid + i]; }
    *sum_level = aL;
    *sum_trend = (d == 2) ? aT : 0.0;
    return 0;
}

