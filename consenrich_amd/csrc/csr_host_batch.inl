// csr_host_batch.inl -- part of csr_lib.hip (one translation unit; included in this order): batch configuration, uploads, natural-layout scratch arrays
// clang-format off is NOT needed; this file is plain C++/HIP host code.

// ---------------------------------------------------------------------------------------------------------------
// batch configuration
// ---------------------------------------------------------------------------------------------------------------
static void fill_model(csr_ctx *c) {
    Prm &p = c->p;
    const csr_model &m = c->mdl;
    p.d = m.state_dim;
    p.F00 = m.F[0]; p.F01 = m.F[1]; p.F10 = m.F[2]; p.F11 = m.F[3];
    p.Q00 = m.Q0[0]; p.Q01 = m.Q0[1]; p.Q10 = m.Q0[2]; p.Q11 = m.Q0[3];
    if (m.state_dim == 1) { p.F00 = 1; p.F01 = 0; p.F10 = 0; p.F11 = 1; p.Q01 = p.Q10 = p.Q11 = 0; }
    p.init = m.state_init; p.cinit = m.state_covar_init; p.pad = m.pad;
    p.wMin = m.w_min; p.wMax = m.w_max; p.kMin = m.k_min; p.kMax = m.k_max;
    p.apnMinQ = m.apn_min_q; p.apnMaxQ = m.apn_max_q; p.apnThresh = m.apn_thresh;
    p.apnScale = m.apn_scale; p.apnPC = m.apn_pc;
    p.qDiag = 0.5 * (m.Q0[0] + m.Q0[3]);
    p.nu = 8.0;
    c->modelQDiagonal = p.Q01 == 0.0 && p.Q10 == 0.0;
    c->qDiagonal = c->modelQDiagonal && (p.chainQ == nullptr || c->chainQDiagonal);
}

extern "C" int csr_batch_set_model(csr_ctx *c, const csr_model *mdl) {
    if (!c || !mdl) return fail("null argument");
    if (!c->configured) return fail("batch not configured");
    if (mdl->state_dim != c->mdl.state_dim) return fail("state_dim cannot change without reconfiguring the batch");
    CHECK(ctx_select(c));
    CHECK(settle(c));
    if (mdl->pad != c->mdl.pad) c->statsValid = false;
    c->mdl = *mdl;
    fill_model(c);
    c->haveFwd = c->haveBwd = false;
    return 0;
}

// Per-chain base process noise: q = n_chains x 4 doubles (row-major Q0 of every chain, float32 values widened like
// csr_model.Q0) or NULL = every chain uses the model's Q0 again.  The reference seeds Q0 per chromosome (core.py:5667).
extern "C" int csr_batch_set_chain_q(csr_ctx *c, const double *q) {
    if (!c) return fail("null argument");
    if (!c->configured) return fail("batch not configured");
    CHECK(ctx_select(c));
    CHECK(settle(c));
    const int nc = (int)c->chains.size();
    if (q == nullptr) {
        c->p.chainQ = nullptr;
        c->qDiagonal = c->modelQDiagonal;
    } else {
        std::vector<double> h(q, q + 4 * (size_t)nc);
        for (int i = 0; i < nc; ++i) {
            double *t = &h[4 * (size_t)i];
            for (int k = 0; k < 4; ++k)
                if (!std::isfinite(t[k])) return fail("chain %d: Q0 must be finite", i);
            if (c->mdl.state_dim == 1) t[1] = t[2] = t[3] = 0.0;          // level model: only Q00 exists (fill_model)
        }
        if (!c->dChainQ) CHECK(dalloc(c, &c->dChainQ, 4 * (int64_t)nc));
        HIPOK(hipMemcpyAsync(c->dChainQ, h.data(), 8 * h.size(), hipMemcpyHostToDevice, c->stream));
        HIPOK(hipStreamSynchronize(c->stream));
        c->p.chainQ = c->dChainQ;
        c->chainQDiagonal = true;
        for (int i = 0; i < nc; ++i)
            if (h[4 * (size_t)i + 1] != 0.0 || h[4 * (size_t)i + 2] != 0.0) c->chainQDiagonal = false;
        c->qDiagonal = c->chainQDiagonal;       // with a per-chain table the model's Q0 is not used
    }
    c->haveFwd = c->haveBwd = false;
    return 0;
}

extern "C" int csr_batch_configure(csr_ctx *c, const csr_model *mdl, int64_t m, int32_t n_chains,
                                   const int64_t *chain_len) {
    if (!c || !mdl || !chain_len) return fail("null argument");
    if (mdl->state_dim != 1 && mdl->state_dim != 2) return fail("state_dim must be 1 or 2");
    if (m <= 0 || n_chains <= 0) return fail("m and n_chains must be positive");
    CHECK(ctx_select(c));
    HIPOK(hipStreamSynchronize(c->stream));
    free_batch(c);
    c->mdl = *mdl;
    c->m = m;
    c->chains.clear();
    if (!c->Bfixed || c->B == 0) {
        // enough blocks to occupy the chip (>= ~16k lanes) without inflating the warm-up share more than needed
        int64_t total = 0;
        for (int i = 0; i < n_chains; ++i) total += chain_len[i];
        // with the 80-bin windows of the tolerant mode 128-bin blocks give ~1.7 waves/SIMD at genome scale, which hides
        // the chains' load latency better than the smaller warm-up share of 256-bin blocks pays (0.36 -> 0.24 ms)
        // below 2 M bins the chains are purely latency-bound (< 1 wave per SIMD): 32-bin blocks shorten every lane's walk
        // (80 + 32 instead of 80 + 64 steps; 0.436 -> 0.414 ms on a 1/8-genome shard).  Bit-exact mode keeps 64: its
        // state chain repairs one block per validation pass, shorter blocks mean more passes.
        // 2 M .. 6 M bins (a quarter-genome shard): 64-bin blocks, 0.77 vs 0.82 ms with 128; above that 64 = 128.
        c->B = total >= (int64_t)24000000 ? 256
             : total >= (int64_t)6000000  ? 128
             : total >= (int64_t)2000000  ? 64
                                          : (c->xTolUlps > 0 ? 32 : 64);
    }
    const int B = c->B;
    int64_t off = 0, nb = 0;
    for (int i = 0; i < n_chains; ++i) {
        if (chain_len[i] <= 0) return fail("chain %d has non-positive length", i);
        ChainInfo ci;
        ci.n = chain_len[i];
        ci.off = off;
        ci.b0 = nb;
        ci.nb = (ci.n + B - 1) / B;
        c->chains.push_back(ci);
        off += (ci.n + 63) / 64 * 64;
        nb += ci.nb;
    }
    if (off >= (int64_t)1 << 31) return fail("batch too large: %lld bins (limit 2^31)", (long long)off);
    c->Npad = off;
    c->NB = nb;
    c->NG = (nb + 63) / 64;
    c->TN = c->NG * (int64_t)B * 64;

    Prm &p = c->p;
    memset(&p, 0, sizeof(p));
    fill_model(c);
    p.B = B; p.m = (int)m; p.nchains = n_chains; p.NB = c->NB; p.NG = c->NG; p.Npad = c->Npad;

    // block table
    std::vector<int4> blk((size_t)nb);
    std::vector<int> bch((size_t)nb);
    std::vector<int64_t> cf(n_chains), cn(n_chains);
    for (int i = 0; i < n_chains; ++i) {
        const ChainInfo &ci = c->chains[i];
        cf[i] = ci.b0; cn[i] = ci.nb;
        for (int64_t k = 0; k < ci.nb; ++k) {
            int4 e;
            e.x = (int)(ci.off + k * B);
            e.y = (int)std::min<int64_t>(B, ci.n - k * B);
            e.z = (int)ci.b0;
            e.w = (int)(ci.b0 + ci.nb - 1);
            blk[(size_t)(ci.b0 + k)] = e;
            bch[(size_t)(ci.b0 + k)] = i;
        }
    }
    int4 *dblk; int *dbch;
    CHECK(dalloc(c, &dblk, nb));
    CHECK(dalloc(c, &dbch, nb));
    CHECK(dalloc(c, &c->dChainFirst, n_chains));
    CHECK(dalloc(c, &c->dChainNb, n_chains));
    CHECK(dalloc(c, &c->dActive, n_chains));
    HIPOK(hipMemcpy(dblk, blk.data(), sizeof(int4) * nb, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(dbch, bch.data(), sizeof(int) * nb, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(c->dChainFirst, cf.data(), sizeof(int64_t) * n_chains, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(c->dChainNb, cn.data(), sizeof(int64_t) * n_chains, hipMemcpyHostToDevice));
    {
        std::vector<int64_t> co(n_chains), cl(n_chains);
        for (int i = 0; i < n_chains; ++i) { co[i] = c->chains[i].off; cl[i] = c->chains[i].n; }
        CHECK(dalloc(c, &c->dChainOff, n_chains));
        CHECK(dalloc(c, &c->dChainLen, n_chains));
        HIPOK(hipMemcpy(c->dChainOff, co.data(), sizeof(int64_t) * n_chains, hipMemcpyHostToDevice));
        HIPOK(hipMemcpy(c->dChainLen, cl.data(), sizeof(int64_t) * n_chains, hipMemcpyHostToDevice));
    }
    HIPOK(hipMemset(c->dActive, 1, n_chains));
    p.blk = dblk; p.blkChain = dbch; p.chainActive = nullptr; p.bg = nullptr;

    float *dd, *dm;
    CHECK(dalloc(c, &dd, m * c->Npad));
    CHECK(dalloc(c, &dm, m * c->Npad));
    HIPOK(hipMemsetAsync(dd, 0, sizeof(float) * m * c->Npad, c->stream));
    HIPOK(hipMemsetAsync(dm, 0, sizeof(float) * m * c->Npad, c->stream));
    p.data = dd; p.munc = dm;

    const int64_t T = c->TN;
    CHECK(dalloc(c, &p.tSZ, T)); CHECK(dalloc(c, &p.tS2c, T)); CHECK(dalloc(c, &p.tLogR, T)); CHECK(dalloc(c, &p.tS2L, T));
    p.statsF32 = 0; p.nisInChain = 0; p.rM = 1.0 / (double)m;
    // (+ one padding wave-group: the smoother's look-ahead at the multipliers of bin k+1 may touch the slot after the batch's
    // last block, csr_device.h BwdTrend::load)
    const int64_t TP = T + (int64_t)B * 64;
    CHECK(dalloc(c, &p.tLam, TP)); CHECK(dalloc(c, &p.tKap, TP)); CHECK(dalloc(c, &p.tQs, TP));
    p.tKapOut = p.tKap;
    p.storePP = 1;
    c->kapScratch[0] = c->kapScratch[1] = nullptr;
    c->kapIn = c->kapOut = nullptr;
    CHECK(dalloc(c, &p.tXin, T)); CHECK(dalloc(c, &p.tPf, T)); CHECK(dalloc(c, &p.tQ, T)); CHECK(dalloc(c, &p.tQ2, T));
    CHECK(dalloc(c, &p.tXf, T)); CHECK(dalloc(c, &p.tD, T)); CHECK(dalloc(c, &p.tPP, T));
    CHECK(dalloc(c, &p.tXs, T)); CHECK(dalloc(c, &p.tPs, T)); CHECK(dalloc(c, &p.tLag, T));
    if (mdl->state_dim == 1) { CHECK(dalloc(c, &p.tXd, T)); }
    // multipliers default to 1 (the reference's cold start, pyx:7901/7914) until csr_batch_upload_multipliers
    HIPOK(hipMemsetD32Async((hipDeviceptr_t)p.tLam, 0x3f800000, (size_t)TP, c->stream));
    HIPOK(hipMemsetD32Async((hipDeviceptr_t)p.tKap, 0x3f800000, (size_t)TP, c->stream));
    HIPOK(hipMemsetD32Async((hipDeviceptr_t)p.tQs, 0x3f800000, (size_t)TP, c->stream));
    // defined contents for slots no kernel writes (pNoise/lag tails, padding)
    HIPOK(hipMemsetAsync(p.tQ, 0, sizeof(float4) * T, c->stream));
    HIPOK(hipMemsetAsync(p.tQ2, 0, sizeof(float2) * T, c->stream));
    HIPOK(hipMemsetAsync(p.tLag, 0, sizeof(float4) * T, c->stream));
    HIPOK(hipMemsetAsync(p.tD, 0, sizeof(float) * T, c->stream));
    CHECK(dalloc(c, &p.blkSumD, nb)); CHECK(dalloc(c, &p.blkSumNLL, nb));
    c->mailBytes = MAIL_HDR + sizeof(double) * 2 * (size_t)n_chains;
    CHECK(dalloc(c, &c->dMail, (int64_t)c->mailBytes));
    HIPOK(hipMemsetAsync(c->dMail, 0, c->mailBytes, c->stream));
    p.rerunCount = reinterpret_cast<unsigned int *>(c->dMail);
    p.rerunCountPass = reinterpret_cast<unsigned int *>(c->dMail) + MAIL_DUMMY;
    p.chainSumD = reinterpret_cast<double *>(c->dMail + MAIL_HDR);
    p.chainSumNLL = p.chainSumD + n_chains;
    for (unsigned int &v : c->lastCnt) v = 0;
    for (int &v : c->nPasses) v = 1;
    for (int &v : c->cleanRuns) v = 0;
    for (DevBuf *b : {&c->bgBuf, &c->wrBuf, &c->textBuf})
        if (b->ptr) { (void)hipFree(b->ptr); b->ptr = nullptr; b->cap = 0; }
    if (c->hMail) (void)hipHostFree(c->hMail);
    c->hMail = nullptr;
    HIPOK(hipHostMalloc((void **)&c->hMail, c->mailBytes));
    memset(c->hMail, 0, c->mailBytes);
    char *ci_, *coa, *cob;
    CHECK(dalloc(c, &ci_, nb * 32)); CHECK(dalloc(c, &coa, nb * 32)); CHECK(dalloc(c, &cob, nb * 32));
    p.carryIn = ci_; p.carryOutA = coa; p.carryOutB = cob;
    {
        // second (carry-in, carry-out) set: consecutive stages alternate, so that a stage's carries survive until the next
        // speculative kernel has checked them (folded validation, run_chain)
        char *b0, *b1;
        CHECK(dalloc(c, &b0, nb * 32)); CHECK(dalloc(c, &b1, nb * 32));
        c->carrySet[0][0] = ci_; c->carrySet[0][1] = coa; c->carrySet[1][0] = b0; c->carrySet[1][1] = b1;
        c->pendChk = csr_ctx::PendingCheck{};
    }
    c->configured = true;
    c->rs = csr_run_stats{};
    return 0;
}

static int need(csr_ctx *c) {
    if (!c) return fail("null context");
    if (!c->configured) return fail("batch not configured");
    return ctx_select(c);
}

extern "C" int64_t csr_batch_chain_offset(csr_ctx *c, int32_t chain) {
    if (!c || !c->configured || chain < 0 || chain >= (int)c->chains.size()) return -1;
    return c->chains[chain].off;
}

extern "C" int csr_batch_upload(csr_ctx *c, int32_t chain, const float *data, const float *munc) {
    CHECK(need(c));
    CHECK(settle(c));
    if (chain < 0 || chain >= (int)c->chains.size()) return fail("chain index out of range");
    if (!data || !munc) return fail("null host buffer");
    const ChainInfo &ci = c->chains[chain];
    HIPOK(hipMemcpy2DAsync(const_cast<float *>(c->p.data) + ci.off, sizeof(float) * c->Npad, data, sizeof(float) * ci.n,
                           sizeof(float) * ci.n, (size_t)c->m, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipMemcpy2DAsync(const_cast<float *>(c->p.munc) + ci.off, sizeof(float) * c->Npad, munc, sizeof(float) * ci.n,
                           sizeof(float) * ci.n, (size_t)c->m, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    c->statsValid = c->haveFwd = c->haveBwd = false;
    return 0;
}

// D2H of a chain's resident inputs ((m, n) float32 each, like csr_batch_upload takes them): lets a caller check what a
// device-side generator (csr_batch_synthesize) or fold builder (csr_batch_make_fold) put there.
extern "C" int csr_batch_download_inputs(csr_ctx *c, int32_t chain, float *data, float *munc) {
    CHECK(need(c));
    CHECK(settle(c));
    if (chain < 0 || chain >= (int)c->chains.size()) return fail("chain index out of range");
    const ChainInfo &ci = c->chains[chain];
    if (data)
        HIPOK(hipMemcpy2DAsync(data, sizeof(float) * ci.n, c->p.data + ci.off, sizeof(float) * c->Npad, sizeof(float) * ci.n,
                               (size_t)c->m, hipMemcpyDeviceToHost, c->stream));
    if (munc)
        HIPOK(hipMemcpy2DAsync(munc, sizeof(float) * ci.n, c->p.munc + ci.off, sizeof(float) * c->Npad, sizeof(float) * ci.n,
                               (size_t)c->m, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    return 0;
}

static int grid_slots(csr_ctx *c) { return (int)((c->TN + 255) / 256); }

static int64_t arr_comps(csr_ctx *c, int id);
// Reference-layout ("natural") device arrays of per-bin floats (import / export), allocated at first use.
// First use zeroes the array on the context's own zeroing stream and WAITS for it on the host before the pointer is handed
// out: the zeroing is ordered against no caller's stream, so whichever stream `c->stream` is at that moment (step_pipelined
// swaps it to a tail group's stream) nothing queued later -- on any stream -- can be overtaken by it.  (Round 4 zeroed on
// `c->stream`: a first use inside a tail group wiped what the next group had already written from the other stream.)
static int nat_array(csr_ctx *c, int id, float **out) {
    if (!c->nat[id]) {
        const int64_t per = arr_comps(c, id);
        // (+ 64 spare bins: the bit-exact state chain's ring DMAs fetch the stored trajectory 128 bins at a time)
        CHECK(dalloc(c, &c->nat[id], per * (c->Npad + 64)));
        HIPOK(hipMemsetAsync(c->nat[id], 0, sizeof(float) * per * (c->Npad + 64), c->zeroStream));
        HIPOK(hipStreamSynchronize(c->zeroStream));
        if (c->mainStream && c->stream != c->mainStream) ++c->rs.nat_first_use_off_main;
    }
    *out = c->nat[id];
    return 0;
}
static int64_t arr_comps(csr_ctx *c, int id);
static int64_t arr_comps_impl(csr_ctx *c, int id) {
    const int d = c->mdl.state_dim;
    switch (id) {
        case CSR_ARR_D: case CSR_ARR_LAMBDA: case CSR_ARR_KAPPA: case CSR_ARR_QSCALE: case CSR_ARR_SUMGAIN0:
        case CSR_ARR_SUMGAIN1: case CSR_ARR_EFFQ_LEVEL: case CSR_ARR_EFFQ_TREND: case CSR_ARR_MUNCTRACE:
        case CSR_ARR_BACKGROUND: case CSR_ARR_BACKGROUND_NEXT: return 1;
        case CSR_ARR_XF: case CSR_ARR_XS: return d;
        case CSR_ARR_RESID: return c->m;
        default: return d * d;
    }
}

static int64_t arr_comps(csr_ctx *c, int id) { return arr_comps_impl(c, id); }

extern "C" int csr_batch_upload_multipliers(csr_ctx *c, int32_t chain, const float *lambda, const float *kappa,
                                            const float *qscale) {
    CHECK(need(c));
    CHECK(settle(c));
    if (chain < 0 || chain >= (int)c->chains.size()) return fail("chain index out of range");
    const float *src[3] = {lambda, kappa, qscale};
    float *dst[3] = {c->p.tLam, c->p.tKap, c->p.tQs};
    // staged through a scratch buffer of its own (the exported arrays stay what the last export made them)
    CHECK(c->stageBuf.reserve(sizeof(float) * (size_t)c->Npad));
    float *scr = reinterpret_cast<float *>(c->stageBuf.ptr);
    // the scatter is restricted to this chain (k_import_f32 skips inactive chains): the others keep their multipliers
    std::vector<unsigned char> act(c->chains.size(), 0);
    act[chain] = 1;
    HIPOK(hipMemcpyAsync(c->dActive, act.data(), act.size(), hipMemcpyHostToDevice, c->stream));
    const ChainInfo &ci = c->chains[chain];
    for (int k = 0; k < 3; ++k) {
        if (!src[k]) continue;
        HIPOK(hipMemcpyAsync(scr + ci.off, src[k], sizeof(float) * ci.n, hipMemcpyHostToDevice, c->stream));
        Prm p = c->p;
        p.chainActive = c->dActive;
        {
            Scope sc(c, "import_f32");
            hipLaunchKernelGGL(k_import_f32, dim3(grid_slots(c)), dim3(256), 0, c->stream, p, scr, 1, 0, dst[k], 1, 0);
        }
        LAUNCH_CHECK("k_import_f32");
        HIPOK(hipStreamSynchronize(c->stream));
    }
    c->multGen += 1;
    c->haveFwd = c->haveBwd = false;
    return 0;
}

