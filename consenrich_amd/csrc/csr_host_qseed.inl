// csr_host_qseed.inl -- part of csr_lib.hip: SURVEY 8(f) rank 4, the initial process-noise (Q0) seed
// (cconsenrich.pyx:1441-2146, caller core.py:3621-3780).  Device: csr_qseed.h (gather + per-transition reductions over the
// sampled columns of the resident matrices) and csr_qseed_post.h (order statistics of the precision sample by bitwise
// bisection; weighted quantiles by bitonic sort; the grid posterior).  Host, below: orchestration, the compaction of the
// selected transitions in scan order and the stable ordering of the signal levels for the 2048-transition panel.

// ---------------------------------------------------------------------------------------------------------------
// host tail
// ---------------------------------------------------------------------------------------------------------------
static void qs_stable_argsort(const double *v, int64_t n, std::vector<int64_t> &idx) {   // np.argsort(kind="mergesort")
    idx.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [v](int64_t a, int64_t b) { return v[a] < v[b]; });
}
static inline int64_t qs_sample_index_h(int64_t i, int64_t items, int64_t samples) {   // pyx:1431-1438
    return (int64_t)std::floor((((double)i + 0.5) * (double)items) / (double)samples);
}

// Grid posteriors of a set of transition lists (pyx:1998-2146), all in ONE launch of k_qs_posterior (csr_qseed_post.h):
// weighted quantiles by a bitonic sort + a walk of the cumulative weights, one thread per grid point for the likelihood.
struct QpHostJob {
    const double *d, *s2, *w;       // host
    int64_t n;
};
static const char *const QP_ERR[] = {"", "deltas must be finite", "samplingVariances must be nonnegative finite",
                                     "transitionWeights must be positive finite",
                                     "q seed posterior produced a nonfinite score", "q seed posterior normalization failed"};
static int qs_posterior_device(csr_ctx *c, const std::vector<QpHostJob> &in, const csr_qseed_post_cfg &cf,
                               csr_qseed_post *outs) {
    const size_t nj = in.size();
    if (nj == 0) return 0;
    const int64_t G = std::max<int64_t>(cf.grid_size, 1);
    size_t need_ = 0;
    auto take = [&](size_t b) { const size_t o = need_; need_ += (b + 255) / 256 * 256; return o; };
    struct Off { size_t d, s2, w, key, ord, lp; int64_t P; };
    std::vector<Off> off(nj);
    const size_t oJobs = take(sizeof(QpJob) * nj), oOut = take(sizeof(csr_qseed_post) * nj), oSt = take(sizeof(int) * nj);
    for (size_t j = 0; j < nj; ++j) {
        const size_t n = (size_t)std::max<int64_t>(in[j].n, 1);
        int64_t P = 1;
        while (P < (int64_t)n) P <<= 1;
        off[j] = Off{take(8 * n), take(8 * n), take(8 * n), take(8 * (size_t)P), take(4 * (size_t)P), take(8 * 3 * (size_t)G), P};
    }
    CHECK(c->qpBuf.reserve(need_));
    char *base = (char *)c->qpBuf.ptr;
    std::vector<QpJob> jobs(nj);
    for (size_t j = 0; j < nj; ++j) {
        const size_t n = (size_t)in[j].n;
        if (n) {
            HIPOK(hipMemcpyAsync(base + off[j].d, in[j].d, 8 * n, hipMemcpyHostToDevice, c->stream));
            HIPOK(hipMemcpyAsync(base + off[j].s2, in[j].s2, 8 * n, hipMemcpyHostToDevice, c->stream));
            HIPOK(hipMemcpyAsync(base + off[j].w, in[j].w, 8 * n, hipMemcpyHostToDevice, c->stream));
        }
        QpJob &q = jobs[j];
        q.n = in[j].n; q.P = off[j].P;
        q.d = (const double *)(base + off[j].d); q.s2 = (const double *)(base + off[j].s2); q.w = (const double *)(base + off[j].w);
        q.key = (double *)(base + off[j].key); q.work = nullptr; q.ord = (int *)(base + off[j].ord);
        q.logPost = (double *)(base + off[j].lp);
        q.out = (csr_qseed_post *)(base + oOut) + j;
        q.status = (int *)(base + oSt) + j;
    }
    HIPOK(hipMemcpyAsync(base + oJobs, jobs.data(), sizeof(QpJob) * nj, hipMemcpyHostToDevice, c->stream));
    QpCfg k;
    k.q_floor = cf.q_floor; k.q_cap = cf.q_cap; k.robust_t_nu = cf.robust_t_nu; k.q_seed_prior_level = cf.q_seed_prior_level;
    k.prior_log_sd = cf.prior_log_sd; k.default_t_nu = cf.default_t_nu; k.min_transitions = cf.min_transitions; k.grid_size = G;
    {
        Scope sc_(c, "qseed_posterior");
        hipLaunchKernelGGL(k_qs_posterior, dim3((unsigned)nj), dim3(1024), 0, c->stream, (const QpJob *)(base + oJobs), k);
    }
    LAUNCH_CHECK("k_qs_posterior");
    std::vector<int> st(nj);
    HIPOK(hipMemcpyAsync(outs, base + oOut, sizeof(csr_qseed_post) * nj, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipMemcpyAsync(st.data(), base + oSt, sizeof(int) * nj, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    for (size_t j = 0; j < nj; ++j)
        if (st[j] != 0) return fail("%s", QP_ERR[st[j] < 6 ? st[j] : 0]);
    return 0;
}

// independent per-chain host work (quantiles, orderings, grid posteriors) on a few threads
template <class F>
static void qs_parallel_for(int n, F fn) {
    const int T = std::max(1, std::min<int>({n, 16, (int)std::thread::hardware_concurrency()}));
    auto worker = [&](int t) {
        for (int i = t; i < n; i += T) fn(i);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < T; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (std::thread &th : pool) th.join();
}

// ---------------------------------------------------------------------------------------------------------------
// device part
// ---------------------------------------------------------------------------------------------------------------
static const char *const QS_ERR[] = {"", "active matrixData values must be finite",
                                     "active obsVar values must be positive finite",
                                     "active transition values must be finite",
                                     "active transition precision must be positive finite", "",
                                     "active pooled observations must be finite with positive variance"};

struct QsJob {
    QsArgs a{};                    // source fields filled by the caller (m, n, stride, src64, pointers, pad)
    bool empty = false;
    int64_t rawCap = 0;
    size_t oRaw = 0, oVec = 0;     // offsets into qsBuf
    std::vector<double> deltas, svar, weights;   // selected transitions
    csr_qseed_sample_diag dg{};
};

static int qs_same_track(csr_ctx *c, std::vector<QsJob> &jobs, const csr_qseed_sample_cfg &cf) {
    size_t need_ = 0, workBytes = 0;
    auto take = [&](size_t b) { const size_t o = need_; need_ += (b + 255) / 256 * 256; return o; };
    for (QsJob &jb : jobs) {
        QsArgs &a = jb.a;
        memset(&jb.dg, 0, sizeof(jb.dg));
        jb.dg.precision_cap = NAN;
        jb.dg.transition_sample_fraction = 1.0;
        jb.empty = a.n < 2 || a.m <= 0;
        if (jb.empty) continue;
        a.maxT = a.n - 1;
        a.scanCount = a.maxT;
        a.capped = 0;
        if (cf.max_transition_samples > 0 && cf.max_transition_samples < a.maxT) {
            a.capped = 1;
            a.scanCount = cf.max_transition_samples;
            jb.dg.transition_sample_fraction = (double)a.scanCount / (double)a.maxT;
            if (cf.precision_sample_cap <= 0) return fail("precisionSampleCap must be positive");
        }
        jb.dg.capped_mode = a.capped;
        jb.dg.scan_count = a.scanCount;
        const size_t sc = (size_t)a.scanCount;
        jb.rawCap = a.capped ? std::min<int64_t>(cf.precision_sample_cap, a.m * a.scanCount) : a.m * a.maxT;
        jb.oVec = take(sc * (4 + 4 + 8 + 8 * 4) + 64);
        jb.oRaw = take(8 * (size_t)jb.rawCap);
        workBytes = std::max(workBytes, 24 * (size_t)a.m * sc);
    }
    const size_t oWork = take(workBytes);
    CHECK(c->qsBuf.reserve(need_ + 512 + (sizeof(QsSelJob) + 32) * jobs.size()));
    char *base = (char *)c->qsBuf.ptr;
    struct Scal { int64_t pairs; unsigned long long err; };
    std::vector<Scal> scal(jobs.size());
    // phase A: pair counts, validation, prefix
    {
        Scope sc_(c, "qseed_count");
        for (size_t i = 0; i < jobs.size(); ++i) {
            QsJob &jb = jobs[i];
            if (jb.empty) continue;
            QsArgs &a = jb.a;
            const size_t sc = (size_t)a.scanCount;
            char *v = base + jb.oVec;
            a.pairCount = (int64_t *)v;
            a.firstErr = (unsigned long long *)(v + 8);
            a.prefix = (int64_t *)(v + 64);
            a.deltas = (double *)(v + 64 + 8 * sc);
            a.svar = a.deltas + sc;
            a.weights = a.svar + sc;
            a.sig = a.weights + sc;
            a.cnt = (int32_t *)(a.sig + sc);
            a.cappedCnt = a.cnt + sc;
            a.raw = (double *)(base + jb.oRaw);
            a.work = (double *)(base + oWork);
            HIPOK(hipMemsetAsync(a.firstErr, 0xFF, 8, c->stream));
            hipLaunchKernelGGL(k_qs_count, dim3((unsigned)((sc + 255) / 256)), dim3(256), 0, c->stream, a);
            hipLaunchKernelGGL(k_qs_scan, dim3(1), dim3(1024), 0, c->stream, a);
            HIPOK(hipMemcpyAsync(&scal[i], v, 16, hipMemcpyDeviceToHost, c->stream));
        }
    }
    LAUNCH_CHECK("k_qs_count");
    HIPOK(hipStreamSynchronize(c->stream));
    // phase B: the precision sample stays on the device; its two quantiles (pyx:1658-1662: median and cap quantile, linear
    // interpolation between order statistics) come from exact rank selection (k_qs_select), four ranks per chain
    std::vector<QsSelJob> sel(jobs.size());
    std::vector<double> frac(2 * jobs.size(), 0.0);
    size_t nSel = 0;
    {
        Scope sc_(c, "qseed_sample");
        for (size_t i = 0; i < jobs.size(); ++i) {
            QsJob &jb = jobs[i];
            memset(&sel[i], 0, sizeof(sel[i]));
            for (int q = 0; q < 4; ++q) sel[i].rank[q] = -1;
            if (jb.empty) continue;
            if (scal[i].err != ~0ull) return fail("%s", QS_ERR[scal[i].err & 0xFF]);
            QsArgs &a = jb.a;
            a.nPairs = scal[i].pairs;
            a.sampleCount = a.capped ? std::min<int64_t>(a.nPairs, cf.precision_sample_cap) : a.nPairs;
            if (a.sampleCount > jb.rawCap) return fail("internal: precision sample exceeds its buffer");
            jb.dg.pair_count = jb.dg.sampled_pair_count = a.nPairs;
            jb.dg.precision_sample_count = a.sampleCount;
            if (a.sampleCount <= 0) continue;
            hipLaunchKernelGGL(k_qs_sample, dim3((unsigned)((a.sampleCount + 255) / 256)), dim3(256), 0, c->stream, a);
            sel[i].v = a.raw;
            sel[i].n = a.sampleCount;
            const double qs[2] = {0.5, cf.precision_cap_quantile};
            for (int k = 0; k < 2; ++k) {
                const int64_t ns = a.sampleCount;
                const double pos = qs[k] <= 0.0 ? 0.0 : (qs[k] >= 1.0 ? (double)(ns - 1) : qs[k] * (double)(ns - 1));
                const int64_t lo = (int64_t)std::floor(pos);
                sel[i].rank[2 * k] = lo;
                sel[i].rank[2 * k + 1] = std::min<int64_t>(lo + 1, ns - 1);
                frac[2 * i + k] = pos - (double)lo;
            }
            ++nSel;
        }
    }
    LAUNCH_CHECK("k_qs_sample");
    std::vector<double> caps(jobs.size(), NAN);
    if (nSel > 0) {
        const size_t oSel = (need_ + 255) / 256 * 256;          // behind everything laid out above (qsBuf was sized with it)
        char *sb = base + oSel;
        for (size_t i = 0; i < jobs.size(); ++i) sel[i].out = (double *)(sb + sizeof(QsSelJob) * jobs.size()) + 4 * i;
        HIPOK(hipMemcpyAsync(sb, sel.data(), sizeof(QsSelJob) * jobs.size(), hipMemcpyHostToDevice, c->stream));
        {
            Scope sc_(c, "qseed_select");
            hipLaunchKernelGGL(k_qs_select, dim3((unsigned)jobs.size()), dim3(1024), 0, c->stream, (const QsSelJob *)sb);
        }
        LAUNCH_CHECK("k_qs_select");
        std::vector<double> osel(4 * jobs.size(), 0.0);
        HIPOK(hipMemcpyAsync(osel.data(), sb + sizeof(QsSelJob) * jobs.size(), 8 * osel.size(), hipMemcpyDeviceToHost, c->stream));
        HIPOK(hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < jobs.size(); ++i) {
            if (jobs[i].empty || jobs[i].a.sampleCount <= 0) continue;
            const double *v = &osel[4 * i];
            const double med = sel[i].rank[1] == sel[i].rank[0] ? v[0] : v[0] + frac[2 * i] * (v[1] - v[0]);
            const double qp = sel[i].rank[3] == sel[i].rank[2] ? v[2] : v[2] + frac[2 * i + 1] * (v[3] - v[2]);
            caps[i] = std::fmin(qp, cf.precision_cap_multiplier * med);
        }
    } else {
        HIPOK(hipStreamSynchronize(c->stream));
    }
    // phase C: the per-transition reductions with that cap
    struct Down { std::vector<char> buf; };
    std::vector<Down> down(jobs.size());
    {
        Scope sc_(c, "qseed_transitions");
        for (size_t i = 0; i < jobs.size(); ++i) {
            QsJob &jb = jobs[i];
            if (jb.empty) continue;
            QsArgs &a = jb.a;
            a.cap = caps[i];
            jb.dg.precision_cap = caps[i];
            const size_t sc = (size_t)a.scanCount;
            if ((size_t)a.m * 3 * 64 * 8 <= 65536)      // default dynamic-LDS limit of a workgroup
                hipLaunchKernelGGL(k_qs_transitions<true>, dim3((unsigned)((sc + 63) / 64)), dim3(64),
                                   (size_t)a.m * 3 * 64 * 8, c->stream, a);
            else
                hipLaunchKernelGGL(k_qs_transitions<false>, dim3((unsigned)((sc + 255) / 256)), dim3(256), 0, c->stream, a);
            // deltas | svar | weights | sig | cnt | cappedCnt are contiguous on the device: one copy per chain
            down[i].buf.resize(40 * sc);
            HIPOK(hipMemcpyAsync(down[i].buf.data(), a.deltas, 40 * sc, hipMemcpyDeviceToHost, c->stream));
        }
    }
    LAUNCH_CHECK("k_qs_transitions");
    HIPOK(hipStreamSynchronize(c->stream));
    qs_parallel_for((int)jobs.size(), [&](int i) {
        QsJob &jb = jobs[i];
        jb.deltas.clear(); jb.svar.clear(); jb.weights.clear();
        if (jb.empty) return;
        const int64_t sc = jb.a.scanCount;
        const double *dD = (const double *)down[i].buf.data(), *dS = dD + sc, *dW = dS + sc, *dG = dW + sc;
        const int32_t *dCnt = (const int32_t *)(dG + sc), *dCap = dCnt + sc;
        std::vector<double> sig;
        int64_t cappedPairs = 0;
        for (int64_t s = 0; s < sc; ++s) {                       // compaction in scan order (pyx:1709-1731)
            cappedPairs += dCap[s];
            if (dCnt[s] <= 0) continue;
            jb.deltas.push_back(dD[s]);
            jb.svar.push_back(dS[s]);
            jb.weights.push_back(dW[s]);
            sig.push_back(dG[s]);
        }
        int64_t outCount = (int64_t)jb.deltas.size();
        jb.dg.candidate_count = jb.dg.selected_count = outCount;
        if (jb.a.nPairs > 0) jb.dg.precision_cap_fraction = (double)cappedPairs / (double)jb.a.nPairs;
        if (cf.signal_panel_size > 0 && outCount > cf.signal_panel_size) {   // pyx:1734-1768
            std::vector<int64_t> order;
            qs_stable_argsort(sig.data(), outCount, order);
            const int64_t P = cf.signal_panel_size;
            std::vector<double> d2((size_t)P), s2((size_t)P), w2((size_t)P);
            for (int64_t pi = 0; pi < P; ++pi) {
                const int64_t ci = order[qs_sample_index_h(pi, outCount, P)];
                d2[pi] = jb.deltas[ci];
                s2[pi] = jb.svar[ci];
                w2[pi] = jb.weights[ci];
            }
            jb.deltas.swap(d2); jb.svar.swap(s2); jb.weights.swap(w2);
            jb.dg.selected_count = P;
        }
    });
    return 0;
}

// pyx:1845-1902
static int qs_pooled(csr_ctx *c, QsArgs a, std::vector<double> &deltas, std::vector<double> &svar,
                     std::vector<double> &weights) {
    deltas.clear(); svar.clear(); weights.clear();
    if (a.n < 2 || a.m <= 0) return 0;
    const size_t n = (size_t)a.n;
    CHECK(c->qsBuf.reserve(16 * n + 256));
    char *base = (char *)c->qsBuf.ptr;
    a.firstErr = (unsigned long long *)base;
    a.pooledMean = (double *)(base + 256);
    a.pooledVar = a.pooledMean + n;
    HIPOK(hipMemsetAsync(a.firstErr, 0xFF, 8, c->stream));
    {
        Scope sc_(c, "qseed_pooled");
        hipLaunchKernelGGL(k_qs_pooled, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, a);
    }
    LAUNCH_CHECK("k_qs_pooled");
    std::vector<double> pm(n), pv(n);
    unsigned long long err = 0;
    HIPOK(hipMemcpyAsync(pm.data(), a.pooledMean, 8 * n, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipMemcpyAsync(pv.data(), a.pooledVar, 8 * n, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipMemcpyAsync(&err, a.firstErr, 8, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    if (err != ~0ull) return fail("%s", QS_ERR[6]);
    for (size_t i = 0; i + 1 < n; ++i)
        if (std::isfinite(pm[i]) && std::isfinite(pm[i + 1]) && std::isfinite(pv[i]) && std::isfinite(pv[i + 1])) {
            const double s2 = pv[i] + pv[i + 1];
            deltas.push_back(pm[i + 1] - pm[i]);
            svar.push_back(s2);
            weights.push_back(s2 > 0.0 ? 1.0 / std::fmax(s2, DBL_MIN) : 1.0);
        }
    return 0;
}

// core.py:3705-3709: np.median of the active observation variances.  obsVar = max(float64(munc) + pad, 1e-12) is a
// non-decreasing function of the float32 variance, so its order statistics are those of the float32 keys: exact radix
// selection (4 passes of 8 bits per order statistic), no sort, no copy of the matrix.
static int qs_median_obsvar(csr_ctx *c, QsArgs a, double *median, int64_t *count) {
    *median = NAN;
    *count = 0;
    if (a.n < 1 || a.m < 1) return 0;
    CHECK(c->qsBuf.reserve(2048));
    unsigned long long *hist = (unsigned long long *)c->qsBuf.ptr;
    const int64_t cells = a.m * a.n;
    const unsigned grid = (unsigned)std::min<int64_t>((cells + 255) / 256, 4096);
    unsigned long long h[256];
    auto select = [&](int64_t rank, float *out) -> int {
        unsigned prefix = 0, mask = 0;
        for (int shift = 24; shift >= 0; shift -= 8) {
            HIPOK(hipMemsetAsync(hist, 0, sizeof(h), c->stream));
            hipLaunchKernelGGL(k_qs_hist, dim3(grid), dim3(256), 0, c->stream, a, shift, mask, prefix, hist);
            HIPOK(hipMemcpyAsync(h, hist, sizeof(h), hipMemcpyDeviceToHost, c->stream));
            HIPOK(hipStreamSynchronize(c->stream));
            if (shift == 24) {
                int64_t tot = 0;
                for (int d = 0; d < 256; ++d) tot += (int64_t)h[d];
                *count = tot;
                if (tot == 0) return 0;
                if (rank < 0) rank = 0;
            }
            int d = 0;
            for (; d < 255; ++d) {
                if (rank < (int64_t)h[d]) break;
                rank -= (int64_t)h[d];
            }
            prefix |= (unsigned)d << shift;
            mask |= 0xFFu << shift;
        }
        const unsigned u = (prefix & 0x80000000u) ? (prefix & 0x7FFFFFFFu) : ~prefix;
        memcpy(out, &u, 4);
        return 0;
    };
    Scope sc_(c, "qseed_median");
    // count first (rank is unknown until the first histogram): select rank 0 pass yields the count, then the real ranks
    float v0 = 0.f;
    CHECK(select(0, &v0));
    if (*count == 0) return 0;
    const int64_t N = *count, rlo = (N - 1) / 2, rhi = N / 2;
    float vlo, vhi;
    CHECK(select(rlo, &vlo));
    vhi = vlo;
    if (rhi != rlo) CHECK(select(rhi, &vhi));
    auto obs = [&](float v) { const double o = (double)v + a.pad; return o > 1.0e-12 ? o : 1.0e-12; };
    *median = (rhi == rlo) ? obs(vlo) : (obs(vlo) + obs(vhi)) / 2.0;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// entry points
// ---------------------------------------------------------------------------------------------------------------
static void qs_src64(QsArgs &a, int64_t m, int64_t n, char *base, size_t oD, size_t oO, size_t oA) {
    memset(&a, 0, sizeof(a));
    a.m = m; a.n = n; a.stride = n; a.src64 = 1;
    a.data64 = (const double *)(base + oD); a.obs64 = (const double *)(base + oO); a.act = (const uint8_t *)(base + oA);
}
static int qs_upload64(csr_ctx *c, int64_t m, int64_t n, const double *data, const double *obs, const uint8_t *active,
                       QsArgs &a) {
    const size_t mn = (size_t)m * n;
    size_t need_ = 0;
    auto take = [&](size_t b) { const size_t o = need_; need_ += (b + 255) / 256 * 256; return o; };
    const size_t oD = take(8 * mn), oO = take(8 * mn), oA = take(mn);
    char *base;
    CHECK(fold_stage(c, need_, &base));
    HIPOK(hipMemcpyAsync(base + oD, data, 8 * mn, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipMemcpyAsync(base + oO, obs, 8 * mn, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipMemcpyAsync(base + oA, active, mn, hipMemcpyHostToDevice, c->stream));
    qs_src64(a, m, n, base, oD, oO, oA);
    return 0;
}

extern "C" int csr_qseed_same_track(int64_t m, int64_t n, const double *data, const double *obs_var,
                                    const uint8_t *active, const csr_qseed_sample_cfg *cfg, double *deltas,
                                    double *sampling_var, double *weights, int64_t *out_count,
                                    csr_qseed_sample_diag *diag) {
    DEFAULT_CTX_GUARD;
    if (!cfg || !out_count || !diag) return fail("null argument");
    csr_ctx *c = default_ctx();
    if (!c) return -1;
    CHECK(ctx_select(c));
    std::vector<QsJob> jobs(1);
    if (n >= 2 && m > 0) {
        if (!data || !obs_var || !active || !deltas || !sampling_var || !weights) return fail("null argument");
        CHECK(qs_upload64(c, m, n, data, obs_var, active, jobs[0].a));
    } else {
        jobs[0].a.m = m; jobs[0].a.n = n;
    }
    CHECK(qs_same_track(c, jobs, *cfg));
    const size_t cnt = jobs[0].deltas.size();
    if (cnt) {
        memcpy(deltas, jobs[0].deltas.data(), 8 * cnt);
        memcpy(sampling_var, jobs[0].svar.data(), 8 * cnt);
        memcpy(weights, jobs[0].weights.data(), 8 * cnt);
    }
    *out_count = (int64_t)cnt;
    *diag = jobs[0].dg;
    return 0;
}

extern "C" int csr_qseed_pooled(int64_t m, int64_t n, const double *data, const double *obs_var, const uint8_t *active,
                                double *deltas, double *sampling_var, double *weights, int64_t *out_count) {
    DEFAULT_CTX_GUARD;
    if (!out_count) return fail("null argument");
    *out_count = 0;
    if (n < 2 || m <= 0) return 0;
    if (!data || !obs_var || !active || !deltas || !sampling_var || !weights) return fail("null argument");
    csr_ctx *c = default_ctx();
    if (!c) return -1;
    CHECK(ctx_select(c));
    QsArgs a;
    CHECK(qs_upload64(c, m, n, data, obs_var, active, a));
    std::vector<double> d, s, w;
    CHECK(qs_pooled(c, a, d, s, w));
    if (!d.empty()) {
        memcpy(deltas, d.data(), 8 * d.size());
        memcpy(sampling_var, s.data(), 8 * d.size());
        memcpy(weights, w.data(), 8 * d.size());
    }
    *out_count = (int64_t)d.size();
    return 0;
}

extern "C" int csr_qseed_posterior(int64_t count, const double *deltas, const double *sampling_var, const double *weights,
                                   const csr_qseed_post_cfg *cfg, csr_qseed_post *out) {
    DEFAULT_CTX_GUARD;
    if (!cfg || !out || (count > 0 && (!deltas || !sampling_var || !weights))) return fail("null argument");
    csr_ctx *c = default_ctx();
    if (!c) return -1;
    CHECK(ctx_select(c));
    std::vector<QpHostJob> in(1, QpHostJob{deltas, sampling_var, weights, count});
    return qs_posterior_device(c, in, *cfg, out);
}

// core.py:3621-3780 for every chain of the batch
extern "C" int csr_batch_qseed(csr_ctx *c, const csr_qseed_cfg *cfg, csr_qseed_out *out) {
    CHECK(need(c));
    CHECK(settle(c));
    if (!cfg || !out) return fail("null argument");
    if (!std::isfinite(cfg->min_q) || cfg->min_q <= 0.0) return fail("`minQ` must be positive and finite");
    if (!std::isfinite(cfg->q_seed_prior_level) || cfg->q_seed_prior_level <= 0.0)
        return fail("`minQ` must be positive and finite");                      // core.py:3635 checks it under that name
    const double qFloor = cfg->min_q;
    const double qCap = (cfg->max_q < 0.0 || !std::isfinite(cfg->max_q)) ? INFINITY : std::max(cfg->max_q, qFloor);
    if (std::isfinite(qCap) && cfg->q_seed_prior_level > qCap) return fail("`qSeedPriorLevel` must not exceed `maxQ`");
    const int nc = (int)c->chains.size();
    std::vector<QsJob> jobs((size_t)nc);
    for (int i = 0; i < nc; ++i) {
        QsArgs &a = jobs[i].a;
        a.m = c->m; a.n = c->chains[i].n; a.stride = c->Npad; a.src64 = 0;
        a.data = c->p.data + c->chains[i].off;
        a.munc = c->p.munc + c->chains[i].off;
        a.pad = cfg->pad;
        a.maskedHalf = 0.5 * (double)1.0e30f;                                    // constants.py:387, core.py:3000
    }
    CHECK(qs_same_track(c, jobs, cfg->sample));
    csr_qseed_post_cfg pc;
    pc.q_floor = qFloor; pc.q_cap = qCap;
    pc.robust_t_nu = std::isfinite(cfg->robust_t_nu) ? cfg->robust_t_nu : cfg->default_t_nu;   // core.py:3650-3654
    pc.q_seed_prior_level = cfg->q_seed_prior_level;
    pc.min_transitions = cfg->min_transitions; pc.prior_log_sd = cfg->prior_log_sd; pc.default_t_nu = cfg->default_t_nu;
    pc.grid_size = cfg->grid_size;
    auto clampq = [&](double v) {                                                 // core.py:3514-3522
        if (!std::isfinite(v)) v = qFloor;
        v = std::max(v, qFloor);
        if (std::isfinite(qCap)) v = std::min(v, qCap);
        return v;
    };
    // the grid posteriors of all chains in one launch (csr_qseed_post.h)
    std::vector<csr_qseed_post> ests((size_t)nc);
    {
        std::vector<QpHostJob> in((size_t)nc);
        for (int i = 0; i < nc; ++i)
            in[i] = QpHostJob{jobs[i].deltas.data(), jobs[i].svar.data(), jobs[i].weights.data(), (int64_t)jobs[i].deltas.size()};
        CHECK(qs_posterior_device(c, in, pc, ests.data()));
    }
    for (int i = 0; i < nc; ++i) {
        QsJob &jb = jobs[i];
        csr_qseed_out &o = out[i];
        memset(&o, 0, sizeof(o));
        o.sample = jb.dg;
        csr_qseed_post est = ests[i];
        int source = 0;
        if (!est.ok) {                                                            // core.py:3676-3697
            std::vector<double> d, s, w;
            CHECK(qs_pooled(c, jb.a, d, s, w));
            csr_qseed_post pooled;
            {
                std::vector<QpHostJob> in(1, QpHostJob{d.data(), s.data(), w.data(), (int64_t)d.size()});
                CHECK(qs_posterior_device(c, in, pc, &pooled));
            }
            if (pooled.ok) { est = pooled; source = 1; }
        }
        int reason = est.ok ? 0 : 3;
        double qBefore = est.ok ? est.posterior_median : NAN;
        if (!std::isfinite(qBefore) || qBefore <= 0.0) {                          // core.py:3704-3724
            double fvar;
            int64_t cnt;
            CHECK(qs_median_obsvar(c, jb.a, &fvar, &cnt));
            const bool ok = std::isfinite(fvar) && fvar > 0.0;
            qBefore = ok ? 1.0e-4 * fvar : qFloor;
            source = std::isfinite(fvar) ? 2 : 3;
            reason = std::isfinite(fvar) ? 1 : 2;
        }
        const double qInit = clampq(qBefore);
        double qTrend = qInit, qTrendRaw = qInit;
        if (cfg->state_dim == 2) {                                                // core.py:3726-3737
            const double df = std::max(cfg->delta_f, 1.0e-12);
            qTrendRaw = qInit / (df * df);
            qTrend = clampq(qTrendRaw);
        }
        o.q_level = qInit; o.q_trend = qTrend; o.level_pre_clamp = qBefore; o.trend_pre_clamp = qTrendRaw;
        o.source = source; o.reason = reason; o.post = est;
    }
    return 0;
}
