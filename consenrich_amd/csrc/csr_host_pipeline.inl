// csr_host_pipeline.inl -- part of csr_lib.hip (one translation unit; included in this order): statistics, speculative chains with deferred validation, forward / backward / ECM, export
// clang-format off is NOT needed; this file is plain C++/HIP host code.

// ---------------------------------------------------------------------------------------------------------------
// compute
// ---------------------------------------------------------------------------------------------------------------
extern "C" int csr_batch_stats(csr_ctx *c) {
    CHECK(need(c));
    CHECK(settle(c));
    // 2-ulp throughput mode: {S2c, log R} as one float32 pair (csr_device.h Prm::tS2L); the form is a property of the RESIDENT
    // statistics (a later csr_set_validation does not invalidate them: every reader goes through load_s2c / load_logr / load_s2l)
    c->p.statsF32 = (c->statsF32Enabled && c->xTolUlps > 0) ? 1 : 0;
    Prm p = c->p;
    // default mode with the superblock state chain: that chain reads {S0u, zbar} in the reference layout -- written here directly
    const bool natSZ = CSR_GAIN_NAT && CSR_STATS_NATSZ && c->xTolUlps == 0 && c->mdl.state_dim == 2 && c->sbState && !c->seqState;
    p.natSZ = nullptr;
    if (natSZ) {
        if (!c->sbNatGain) { CHECK(dalloc(c, &c->sbNatGain, c->Npad)); CHECK(dalloc(c, &c->sbNatSZ, c->Npad)); }
        p.natSZ = reinterpret_cast<double2 *>(c->sbNatSZ);
    }
    {
        // 16-byte loads, four bins per thread; 64-bin tiles (32 when the block length is not a multiple of 64)
        Scope sc(c, "stats");
        const int ts = (c->B % 64 == 0) ? 64 : 32;
        const int grid = (int)(c->NG * (c->B / ts) * (ts == 64 ? 4 : 2));
        if (ts == 64) hipLaunchKernelGGL((k_stats_v4<64, 4>), dim3(grid), dim3(256), 0, c->stream, p);
        else hipLaunchKernelGGL((k_stats_v4<32, 4>), dim3(grid), dim3(256), 0, c->stream, p);
    }
    LAUNCH_CHECK("k_stats");
    c->statsValid = true;
    c->natSZValid = natSZ;
    c->haveFwd = c->haveBwd = false;
    return 0;
}

enum { ST_P = 0, ST_X = 1, ST_B = 2, ST_DEBUG = 3 };

// Host wait for the library's stream.  The waits on the pipeline's critical path are short (tens of microseconds at
// 1/8-genome batch sizes), where the wake-up latency of a blocking hipStreamSynchronize is a measurable share of the
// step: poll first, block only if the stream is still busy after ~200 us.
// OPT-IN (CONSENRICH_AMD_TIMER_SLACK=1; round 6: off by default -- a library call should not change an attribute of its caller's
// thread for 0.05 ms of a benchmark step): the calling thread's timer slack, lowered for the length of a wait loop (the kernel's
// default of 50 us turns a 20-us sleep into ~75); the caller's setting is put back on the way out.
static const bool g_timerSlackOptIn = [] { const char *e = getenv("CONSENRICH_AMD_TIMER_SLACK"); return e && atoi(e) != 0; }();
struct TimerSlack {
    long old = 0;
    TimerSlack() {
        if (!g_timerSlackOptIn) return;
        old = prctl(PR_GET_TIMERSLACK, 0UL, 0UL, 0UL, 0UL);
        if (old > 1000) (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
    }
    ~TimerSlack() { if (g_timerSlackOptIn && old > 1000) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)old, 0UL, 0UL, 0UL); }
};
// site: which wait of the pipeline this is (0 = the settle point's mailbox read, 1 = the state chain's verdict): each keeps its own
// estimate -- a step has both, of very different lengths
static hipError_t wait_stream(csr_ctx *c, int site = 0) {
    if (c->spinWait) {
        // a short burst of polls (the waits on a shard's critical path are tens of microseconds), then polls 20 us apart, then
        // a blocking wait: a rank never burns a core for the length of a latency-bound launch (8 ranks per node).  Two details
        // keep the wake-up from costing the step tens of microseconds: the calling thread's timer slack is lowered FOR THE LENGTH OF
        // THE WAIT (the kernel's default of 50 us turns a 20-us sleep into ~75; the caller's setting is put back on the way out),
        // and around the moment the PREVIOUS wait of this context ended (steps repeat) the loop polls without sleeping -- at most
        // ~0.4 ms of spinning per wait.
        TimerSlack slack;
        const auto t0 = std::chrono::steady_clock::now();
        auto elapsed_us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
        for (int i = 0; i < 128; ++i) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) { c->lastWaitUs[site] = elapsed_us(); return hipSuccess; }
            if (q != hipErrorNotReady) return q;
        }
        const double expect = c->lastWaitUs[site];
        for (;;) {
            const double el = elapsed_us();
            if (el > 8000.0) break;
            const bool nearEnd = expect > 0.0 && el > expect - 150.0 && el < expect + 250.0;
            if (!nearEnd) std::this_thread::sleep_for(std::chrono::microseconds(20));
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) { c->lastWaitUs[site] = elapsed_us(); return hipSuccess; }
            if (q != hipErrorNotReady) return q;
        }
        const hipError_t r = hipStreamSynchronize(c->stream);
        c->lastWaitUs[site] = elapsed_us();
        return r;
    }
    return hipStreamSynchronize(c->stream);
}
// hand a pending folded check to the kernel about to be launched with `p` (or clear the fields)
static void take_pending_check(csr_ctx *c, Prm &p) {
    p.prevKind = CK_NONE;
    if (!c->pendChk.valid) return;
    unsigned int *cnt = reinterpret_cast<unsigned int *>(c->dMail);
    p.prevKind = c->pendChk.kind;
    p.prevCarryIn = c->pendChk.cin;
    p.prevCarryOut = c->pendChk.cout;
    p.prevCount = cnt + c->pendChk.stage;
    p.prevCountPass = cnt + MAIL_PASS0 + 4 * c->pendChk.stage;
    p.prevActive = c->pendChk.active;
    c->pendChk.valid = false;
}
// a stage that no speculative kernel followed: its check runs as a kernel of its own
static int flush_pending_check(csr_ctx *c) {
    if (!c->pendChk.valid) return 0;
    Prm p = c->p;
    p.xTolUlps = c->xTolUlps;           // the acceptance rule of the stage that is being checked
    take_pending_check(c, p);
    Scope sc(c, "chain_check");
    hipLaunchKernelGGL(k_chain_check, dim3((int)c->NG), dim3(64), 0, c->stream, p);
    LAUNCH_CHECK("k_chain_check");
    c->rs.fix_launches++;
    return 0;
}
static int read_mail(csr_ctx *c, size_t bytes) {
    CHECK(flush_pending_check(c));
    HIPOK(hipMemcpyAsync(c->hMail, c->dMail, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPOK(wait_stream(c));
    return 0;
}
static unsigned int take_fresh(csr_ctx *c, int stage) {
    const unsigned int now = reinterpret_cast<const unsigned int *>(c->hMail)[stage];
    const unsigned int fresh = now - c->lastCnt[stage];
    c->lastCnt[stage] = now;
    return fresh;
}
// adaptive warm-up: many first-pass mismatches mean the speculation window is too short for this data (longer filter
// memory); lengthen it for the following sweeps.  Results do not depend on it.
static void grow_warm(csr_ctx *c, int &warmRef, unsigned int fresh) {
    if (c->adaptWarm && (int64_t)fresh > std::max<int64_t>(4, c->NB / 256) && warmRef < 8192)
        warmRef = std::min(8192, warmRef * 2);
}
static int &stage_warm(csr_ctx *c, int stage) {
    if (stage == ST_P && c->fwdWindow) return *c->fwdWindow;        // fused forward chain with its own window
    if (stage == ST_B && c->bwdWindow) return *c->bwdWindow;        // warm-started smoother sweep
    return stage == ST_P ? c->warmP : (stage == ST_X ? c->warmX : c->warmB);
}
static int64_t &stage_reruns(csr_ctx *c, int stage) {
    return stage == ST_P ? c->rs.reruns_p : (stage == ST_X ? c->rs.reruns_x : c->rs.reruns_b);
}

// F = [[1, f], [0, 1]] (constructMatrixF, core.py:2164-2176): the UF instances of the levelTrend policies (csr_device.h)
static bool unit_f(const csr_ctx *c, const Prm &p) {
    return c->unitFEnabled && p.F00 == 1.0 && p.F10 == 0.0 && p.F11 == 1.0;
}

// Speculative pass + validation/fix-up.  defer = true: launch the speculative pass and ONE validation pass and return
// without a host round trip (the stage's monotonic counter is checked at the next settle point); otherwise iterate
// validation passes to the fixed point here.
template <class CH>
static int run_chain(csr_ctx *c, Prm p, const char *name, const char *fixName, int stage, bool defer) {
    static_assert(sizeof(typename CH::Carry) <= 32, "carry buffers are sized for 32 bytes per block");
    int &warmRef = stage_warm(c, stage);
    p.warm = warmRef;
    p.xTolUlps = c->xTolUlps;
    p.rerunCount = reinterpret_cast<unsigned int *>(c->dMail) + stage;
    const int grid = (int)c->NG;
    const bool pcq = CH::USES_Q && p.chainQ != nullptr;      // per-chain base process noise: per-lane model copy
    // consecutive stages alternate between two carry sets; the previous stage's pending check (if any) rides in this
    // stage's speculative kernel
    const int cset = (c->carryToggle ^= 1);
    p.carryIn = c->carrySet[cset][0];
    p.carryOutA = c->carrySet[cset][1];
    take_pending_check(c, p);
    {
        Scope sc(c, name);
        if constexpr (CH::DMA) {
            if (c->useDma)
                hipLaunchKernelGGL(k_chain_spec_dma<CH>, dim3(grid), dim3(64), sizeof(unsigned) * DMA_R * CH::NW * 64,
                                   c->stream, p);
            else hipLaunchKernelGGL(k_chain_spec<CH>, dim3(grid), dim3(64), 0, c->stream, p);
        } else {
            bool launched = false;
            // (round 6: the LDS-DMA ring and the reference-layout-input forms exist for F = [[1, f], [0, 1]] only -- what the reference's
            // constructMatrixF builds; any other `matrixF` runs the plain-load general instances: same results, fewer kernels to build)
            if constexpr (CH::FAMILY == FAM_FWD_FUSED && CH::UNITF) {
                // ECM sweeps and other passes without reference-layout outputs: inputs through the LDS-DMA ring
                if (c->useDmaFused && c->useDmaWarm && p.natOut && !pcq && p.warm > 0) {
                    // reference-layout outputs: ring for the warm-up only, tile walker for the main phase
                    const uint32_t mm = p.flags & (F_LAMBDA | F_KAPPA | F_QSCALE);
                    const size_t tl = sizeof(NatTilesFwd);
                    if (mm == 0)
                        hipLaunchKernelGGL((k_chain_spec_dmawarm_natfwd<FwdTrendFusedDma<0, CH::UNITF>>), dim3(grid), dim3(64),
                                           std::max(sizeof(unsigned) * DMA_R * 4 * 64, tl), c->stream, p);
                    else if (mm == F_KAPPA)
                        hipLaunchKernelGGL((k_chain_spec_dmawarm_natfwd<FwdTrendFusedDma<1, CH::UNITF>>), dim3(grid), dim3(64),
                                           std::max(sizeof(unsigned) * DMA_R * 5 * 64, tl), c->stream, p);
                    else
                        hipLaunchKernelGGL((k_chain_spec_dmawarm_natfwd<FwdTrendFusedDma<2, CH::UNITF>>), dim3(grid), dim3(64),
                                           std::max(sizeof(unsigned) * DMA_R * 7 * 64, tl), c->stream, p);
                    launched = true;
                } else if (c->useDmaFused && !p.natOut && !pcq && p.ckptOut != nullptr) {
                    // warm-started ECM sweep (kappa only is the reference's default loop)
                    const uint32_t mm = p.flags & (F_LAMBDA | F_KAPPA | F_QSCALE);
                    if (mm == F_KAPPA)
                        hipLaunchKernelGGL((k_chain_spec_dma<FwdTrendFusedDma<1, CH::UNITF>, true>), dim3(grid), dim3(64),
                                           sizeof(unsigned) * DMA_R * 5 * 64, c->stream, p);
                    else
                        hipLaunchKernelGGL((k_chain_spec_dma<FwdTrendFusedDma<2, CH::UNITF>, true>), dim3(grid), dim3(64),
                                           sizeof(unsigned) * DMA_R * 7 * 64, c->stream, p);
                    launched = true;
                } else if (c->useDmaFused && !p.natOut && !pcq) {
                    const uint32_t mm = p.flags & (F_LAMBDA | F_KAPPA | F_QSCALE);
                    if (mm == 0)
                        hipLaunchKernelGGL((k_chain_spec_dma<FwdTrendFusedDma<0, CH::UNITF>>), dim3(grid), dim3(64),
                                           sizeof(unsigned) * DMA_R * 4 * 64, c->stream, p);
                    else if (mm == F_KAPPA)
                        hipLaunchKernelGGL((k_chain_spec_dma<FwdTrendFusedDma<1, CH::UNITF>>), dim3(grid), dim3(64),
                                           sizeof(unsigned) * DMA_R * 5 * 64, c->stream, p);
                    else
                        hipLaunchKernelGGL((k_chain_spec_dma<FwdTrendFusedDma<2, CH::UNITF>>), dim3(grid), dim3(64),
                                           sizeof(unsigned) * DMA_R * 7 * 64, c->stream, p);
                    launched = true;
                }
            }
            if constexpr (CH::FAMILY == FAM_BWD_TREND && CH::UNITF) {
                if (p.natIn && p.natOut && !pcq) {
                    // reference-layout inputs AND outputs, both through the LDS tiles (the forward pass wrote no blocked xf / Pf)
                    if (p.qFromMult) hipLaunchKernelGGL((k_smooth_natin<CH, false>), dim3(grid), dim3(64), sizeof(NatTiles), c->stream, p);
                    else hipLaunchKernelGGL((k_smooth_natin<CH, true>), dim3(grid), dim3(64), sizeof(NatTiles), c->stream, p);
                    launched = true;
                }
                // smoother with reference-layout outputs: warm-up through the ring
                // (32-bin blocks: measured slower, 0.063 vs 0.057 ms -- the ring's fill and drain weigh more than they hide)
                if (!launched && c->useDmaFused && c->useDmaWarm && p.natOut && !pcq && p.warm > 0 && c->B >= 64) {
                    if (p.qFromMult)
                        hipLaunchKernelGGL((k_chain_spec_dmawarm_natbwd<BwdTrendDma<false, CH::UNITF>>), dim3(grid), dim3(64),
                                           std::max(sizeof(unsigned) * DMA_R * 6 * 64, sizeof(NatTiles)), c->stream, p);
                    else
                        hipLaunchKernelGGL((k_chain_spec_dmawarm_natbwd<BwdTrendDma<true, CH::UNITF>>), dim3(grid), dim3(64),
                                           std::max(sizeof(unsigned) * DMA_R * 10 * 64, sizeof(NatTiles)), c->stream, p);
                    launched = true;
                }
            }
            if constexpr (CH::NATOUT || CH::NATOUT_FWD) {
                if (!launched && p.natOut) {
                    if (pcq) hipLaunchKernelGGL((k_chain_spec<CH, true, true>), dim3(grid), dim3(64), sizeof(NatTiles), c->stream, p);
                    else hipLaunchKernelGGL((k_chain_spec<CH, true, false>), dim3(grid), dim3(64), sizeof(NatTiles), c->stream, p);
                    launched = true;
                }
            }
            if (!launched) {
                if (pcq) hipLaunchKernelGGL((k_chain_spec<CH, false, true>), dim3(grid), dim3(64), 0, c->stream, p);
                else if (p.ckptOut != nullptr) hipLaunchKernelGGL((k_chain_spec<CH, false, false, true>), dim3(grid), dim3(64), 0, c->stream, p);
                else hipLaunchKernelGGL((k_chain_spec<CH, false, false>), dim3(grid), dim3(64), 0, c->stream, p);
            }
        }
    }
    LAUNCH_CHECK(name);
    int which = 0;
    unsigned int *const cnt = reinterpret_cast<unsigned int *>(c->dMail);
    auto launch_fix = [&](unsigned int *passCounter) {
        Scope sc(c, fixName);
        p.rerunCountPass = passCounter;
        bool launched = false;
        if constexpr (CH::NATOUT || CH::NATOUT_FWD) {
            if (p.natOut) {
                if (pcq) hipLaunchKernelGGL((k_chain_fix<CH, true, true>), dim3(grid), dim3(64), 0, c->stream, p, which);
                else hipLaunchKernelGGL((k_chain_fix<CH, true, false>), dim3(grid), dim3(64), 0, c->stream, p, which);
                launched = true;
            }
        }
        if (!launched) {
            if (pcq) hipLaunchKernelGGL((k_chain_fix<CH, false, true>), dim3(grid), dim3(64), 0, c->stream, p, which);
            else hipLaunchKernelGGL((k_chain_fix<CH, false, false>), dim3(grid), dim3(64), 0, c->stream, p, which);
        }
        c->rs.fix_launches++;
        which ^= 1;
    };
    if (defer && c->nPasses[stage] <= 1) {
        // clean so far: no validation kernel -- the next speculative kernel (or read_mail) checks this stage's carries
        c->pendChk.valid = true;
        c->pendChk.kind = CH::KIND;
        c->pendChk.stage = stage;
        c->pendChk.cin = p.carryIn;
        c->pendChk.cout = p.carryOutA;
        c->pendChk.active = p.chainActive;
        c->launchedPasses[stage] = 1;
        return 0;
    }
    if (defer) {
        // optimistic: nPasses validation passes back to back, no host round trip; the stage stands iff the last one re-ran
        // nothing (checked at the next settle point through its own counter)
        p.debugForce = 0;
        const int np = std::max(1, std::min(MAX_DEFER_PASSES, c->nPasses[stage]));
        for (int j = 0; j < np; ++j) launch_fix(cnt + MAIL_PASS0 + 4 * stage + j);
        LAUNCH_CHECK(fixName);
        c->launchedPasses[stage] = np;
        return 0;
    }
    // Validation passes are launched in bursts once the first one has re-run blocks: a correction travels one block per
    // pass (bit-exact state chains need hundreds of passes), and reading the counter after every pass costs a host round
    // trip each.  A burst whose passes re-ran nothing at all is the fixed point (a pass without re-runs copies the
    // carries unchanged, so all later ones are empty too); at most burst-1 empty passes are wasted.
    int burst = 1;
    for (int64_t it = 0; it <= c->NB + 1; ++it) {
        p.debugForce = 0;
        for (int rep = 0; rep < burst; ++rep) launch_fix(cnt + MAIL_DUMMY);
        LAUNCH_CHECK(fixName);
        CHECK(read_mail(c, MAIL_HDR));
        const unsigned int fresh = take_fresh(c, stage);
        if (c->dbgLog) fprintf(stderr, "[csr] %s iter %lld reruns %u\n", fixName, (long long)it, fresh);
        if (fresh == 0) {
            // re-arm optimistic launches after a few clean synchronous runs in a row (a stage that fails every other time
            // would otherwise pay a pipeline replay every other time)
            if (it == 0 && (stage != ST_X || c->xTolUlps > 0)) {
                if (++c->cleanRuns[stage] >= 4 || c->nPasses[stage] < MAX_DEFER_PASSES) c->optimistic[stage] = true;
            } else if (it > 0) {
                c->cleanRuns[stage] = 0;
            }
            return 0;
        }
        stage_reruns(c, stage) += fresh;
        if (it == 0) grow_warm(c, warmRef, fresh);
        burst = it == 0 ? 2 : std::min(32, burst * 2);
    }
    return fail("%s: speculative fix-up did not reach a fixed point", name);
}

// ---- warm-started speculation of the ECM sweeps (Prm::ckptIn) -----------------------------------------------------
// Fills the checkpoint fields of `p` for the chain about to be launched in direction fwd / !fwd and returns the window
// variable to use (nullptr: the cold one).  Call ws_launched() after the launch.
static int ws_prepare(csr_ctx *c, Prm &p, bool fwd, int **window) {
    *window = nullptr;
    p.ckptIn = nullptr; p.ckptOut = nullptr; p.ckptSaveWarm = 0;
    if (!c->wsActive || !c->wsEnabled || c->xTolUlps == 0 || p.chainQ != nullptr) return 0;
    // Only the latency-bound batches gain: those the block-length rule gives 32-bin blocks (< 2 M bins: a 1/8-genome shard,
    // 0.64 -> 0.53 ms per ECM iteration).  Larger ones are bandwidth-bound, walk mostly their own block (64 .. 256 bins) and
    // have thousands of wavefront edges -- an edge block that fails costs a replay of the iteration: measured 0.88 -> 1.05 ms
    // on a quarter genome, 2.31 -> 2.46 ms on the genome.
    if (c->B > c->wsMaxBlock) return 0;
    int &warm = fwd ? c->wsWarmF : c->wsWarmB;
    if (warm >= (fwd ? c->warmFM : c->warmB)) return 0;       // widened up to the cold window: nothing left to gain
    void **ck = fwd ? c->ckF : c->ckB;
    for (int k = 0; k < 2; ++k)
        if (!ck[k]) { char *q; CHECK(dalloc(c, &q, c->NB * 32)); ck[k] = q; }
    const int saved = fwd ? c->wsSavedF : c->wsSavedB;
    const int sweep = fwd ? c->wsSweepF : c->wsSweepB;
    p.ckptOut = ck[(sweep + 1) & 1];
    p.ckptSaveWarm = warm;
    p.localFixCount = reinterpret_cast<unsigned int *>(c->dMail) + MAIL_LOCAL;
    if (!c->wsCold && saved == warm && saved > 0) {
        p.ckptIn = ck[sweep & 1];
        *window = &warm;
    }
    return 0;
}
static void ws_launched(csr_ctx *c, const Prm &p, bool fwd) {
    if (p.ckptOut == nullptr) return;
    if (fwd) { c->wsSavedF = p.ckptSaveWarm; c->wsSweepF += 1; }
    else { c->wsSavedB = p.ckptSaveWarm; c->wsSweepB += 1; }
}

// ---- superblock view of the batch (bit-exact state chain) -------------------------------------------------------
static int ensure_sb_view(csr_ctx *c) {
    csr_ctx::SbView &v = c->sb;
    if (v.ready) return 0;
    int B = c->sbBins;
    const int nc = (int)c->chains.size();
    if (!c->sbBinsPinned) {
        // one wavefront per superblock, at most one wavefront per SIMD: the shortest superblock (a multiple of 8192 bins) that
        // leaves no more superblocks than 5/8 of the device's SIMDs (measured at genome scale: 16 384 bins 7.2 ms, 24 576 / 32 768
        // bins 7.0 ms, 8 192 bins 8.3 ms per step -- fewer, longer repair passes win slightly while a pass costs the slowest
        // superblock's walk)
        hipDeviceProp_t prop;
        int simds = 1024;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) simds = 4 * prop.multiProcessorCount * 5 / 8;
        for (;; B += 8192) {
            int64_t cnt = 0;
            for (int i = 0; i < nc; ++i) cnt += (c->chains[i].n + B - 1) / B;
            if (cnt <= simds || B >= (1 << 20)) break;
        }
    }
    std::vector<int64_t> first((size_t)nc);
    int64_t nb = 0;
    for (int i = 0; i < nc; ++i) {
        first[(size_t)i] = nb;
        nb += (c->chains[i].n + B - 1) / B;
    }
    std::vector<int4> blk((size_t)nb);
    std::vector<int> bch((size_t)nb);
    for (int i = 0; i < nc; ++i) {
        const ChainInfo &ci = c->chains[i];
        const int64_t cnt = (ci.n + B - 1) / B;
        for (int64_t k = 0; k < cnt; ++k) {
            int4 e;
            e.x = (int)(ci.off + k * B);
            e.y = (int)std::min<int64_t>(B, ci.n - k * B);
            e.z = (int)first[(size_t)i];
            e.w = (int)(first[(size_t)i] + cnt - 1);
            blk[(size_t)(first[(size_t)i] + k)] = e;
            bch[(size_t)(first[(size_t)i] + k)] = i;
        }
    }
    v.B = B; v.NB = nb; v.NG = (nb + 63) / 64; v.TN = v.NG * (int64_t)B * 64;
    CHECK(dalloc(c, &v.blk, nb)); CHECK(dalloc(c, &v.blkChain, nb)); CHECK(dalloc(c, &v.chainFirst, nc));
    HIPOK(hipMemcpyAsync(v.blk, blk.data(), sizeof(int4) * nb, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipMemcpyAsync(v.blkChain, bch.data(), sizeof(int) * nb, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipMemcpyAsync(v.chainFirst, first.data(), sizeof(int64_t) * nc, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));           // the host vectors go out of scope
    char *q[3];
    for (char *&x : q) CHECK(dalloc(c, &x, nb * 32));
    v.carryIn = q[0]; v.carryOutA = q[1]; v.carryOutB = q[2];
    v.ready = true;
    return 0;
}

// Bit-exact state chain, systolic form (k_sb_sys): the gain / statistics records go to the natural layout (one tiled
// conversion launch), one wavefront per superblock walks 64 bins per batch as a shift register and writes the filtered state
// straight into the reference-layout xf array; superblocks start from the cold prior and the validation / repair passes run
// to the fixed point (= the sequential recursion, whatever the superblock length); one tiled launch brings the filtered state
// back into the batch's blocked layout for the epilogue and the smoother.
static int early_cov_exports(csr_ctx *c, const Prm &p, uint32_t flags, bool withPf);
static int ensure_blocked_fwd(csr_ctx *c, const unsigned char *active);
// phase 0: the whole chain.  phase 1 (a step that pipelines its tail per chain, step_pipelined): stop right after the launch of
// the barrier-free kernel, with per-chain "done" words the host can watch -- c->sbp.active says that this happened (otherwise
// the whole chain ran, as in phase 0).  phase 2: wait for that launch (or run the pass form if it bailed out); the filtered
// state is NOT brought back into the blocked layout (the caller does that per group of chains).
static int state_chain_systolic(csr_ctx *c, const Prm &p, bool earlyExports = false, uint32_t flags = 0, int phase = 0) {
    const bool resume = phase == 2;
    CHECK(ensure_sb_view(c));
    if (!resume) CHECK(flush_pending_check(c));
    csr_ctx::SbView &v = c->sb;
    if (!c->sbNatGain) { CHECK(dalloc(c, &c->sbNatGain, c->Npad)); CHECK(dalloc(c, &c->sbNatSZ, c->Npad)); }
    float *natXf;
    CHECK(nat_array(c, CSR_ARR_XF, &natXf));
    c->sbp.active = false;
    if (!resume && !(c->gainNat && c->natSZValid)) {
        Scope sc(c, "state_records_natural");
        ExpList L;
        memset(&L, 0, sizeof(L));
        // the gain records of this pass unless the covariance chain wrote them in the reference layout itself (forward_impl); the
        // statistics records {S0, zbar} only when the statistics changed since they were last converted (csr_batch_stats): the
        // sweeps of an ECM iteration share them
        L.count = 0;
        if (!c->gainNat) {
            L.count = 1;
            L.d[0].src = reinterpret_cast<const float *>(p.tXin); L.d[0].dst = reinterpret_cast<float *>(c->sbNatGain); L.d[0].E = 4; L.d[0].n = 4;
        }
        Prm pe = p;
        if (!c->natSZValid) {
            ExpDesc &e = L.d[L.count++];
            e.src = reinterpret_cast<const float *>(p.tSZ); e.dst = reinterpret_cast<float *>(c->sbNatSZ); e.E = 4; e.n = 4;
            pe.chainActive = nullptr;       // the statistics of EVERY chain (csr_batch_stats computed them all), whatever this pass masks
            c->natSZValid = true;
        }
        hipLaunchKernelGGL(k_export_tiled, dim3((int)(c->NG * (c->B / 32))), dim3(256), 0, c->stream, pe, L);
    }
    LAUNCH_CHECK("k_export_tiled (state records)");
    if (earlyExports && !resume) CHECK(early_cov_exports(c, p, flags, true));
    Prm q = p;
    q.B = v.B; q.NB = v.NB; q.NG = v.NG; q.blk = v.blk; q.blkChain = v.blkChain;
    q.carryIn = v.carryIn; q.carryOutA = v.carryOutA; q.carryOutB = v.carryOutB;
    unsigned int *const cnt = reinterpret_cast<unsigned int *>(c->dMail);
    q.rerunCount = cnt + ST_X;
    q.rerunCountPass = cnt + MAIL_DUMMY;
    q.prevKind = CK_NONE;
    const int mode = unit_f(c, p) ? ((p.F01 == 1.0 && c->unitF1Enabled) ? 2 : 1) : 0;
    const int grid = (int)((v.NB + 3) / 4);
    q.sbDbg = nullptr;          // (the instrumented instances of round 4's studies, CONSENRICH_AMD_SB_DEBUG, were retired in round 6)
    auto launch = [&](int which, int fix) {
        float2 *xf = reinterpret_cast<float2 *>(natXf);
        if (fix) {        // repair passes in delta form (k_sb_delta)
            const int spec = (c->sbAdvMin << 8) | (c->sbAdvFrom << 16);
            if (mode == 2) hipLaunchKernelGGL(k_sb_delta<2>, dim3(grid), dim3(256), 0, c->stream, q, c->sbNatGain, c->sbNatSZ, xf, which, spec);
            else if (mode == 1) hipLaunchKernelGGL(k_sb_delta<1>, dim3(grid), dim3(256), 0, c->stream, q, c->sbNatGain, c->sbNatSZ, xf, which, spec);
            else hipLaunchKernelGGL(k_sb_delta<0>, dim3(grid), dim3(256), 0, c->stream, q, c->sbNatGain, c->sbNatSZ, xf, which, spec);
            return;
        }
        if (mode == 2) hipLaunchKernelGGL(k_sb_sys<2>, dim3(grid), dim3(256), 0, c->stream, q, c->sbNatGain, c->sbNatSZ, xf);
        else if (mode == 1) hipLaunchKernelGGL(k_sb_sys<1>, dim3(grid), dim3(256), 0, c->stream, q, c->sbNatGain, c->sbNatSZ, xf);
        else hipLaunchKernelGGL(k_sb_sys<0>, dim3(grid), dim3(256), 0, c->stream, q, c->sbNatGain, c->sbNatSZ, xf);
    };
    bool done = false;
    if (c->sbAsync) {
        // the whole chain in one launch, no barrier between passes (k_sb_async); a bail-out (a bounded wait ran out) falls
        // through to the pass form below, which starts over from the cold prior
        if (!v.pub) { CHECK(dalloc(c, &v.pub, 2 * v.NB + 2)); }
        SbAsync a;
        a.carry = v.pub; a.vf = v.pub + v.NB; a.ctl = reinterpret_cast<unsigned int *>(v.pub + 2 * v.NB);
        a.spinLimit = c->sbSpinLimit;
        a.hostDone = nullptr;
        if (phase == 1) {
            // (nothing of this context is in flight that writes these words: the previous launch was waited for)
            if (!c->hDone) {
                // (coherent = fine-grained: a system-scope store of the running kernel is visible to the polling host at once,
                // whatever HIP_HOST_COHERENT says)
                HIPOK(hipHostMalloc((void **)&c->hDone, sizeof(unsigned int) * c->chains.size(), hipHostMallocCoherent | hipHostMallocMapped));
                HIPOK(hipHostGetDevicePointer((void **)&c->dDone, c->hDone, 0));
            }
            for (size_t i = 0; i < c->chains.size(); ++i) c->hDone[i] = 0u;
            a.hostDone = c->dDone;
        }
        if (!resume) {
            HIPOK(hipMemsetAsync(v.pub, 0, sizeof(unsigned long long) * (size_t)(2 * v.NB + 2), c->stream));
            Scope sc(c, "fwd_state_chain");
            float2 *xf = reinterpret_cast<float2 *>(natXf);
            // (the repair runs' LDS ring: > 64 KB of dynamic LDS has to be asked for once per kernel)
            using KFn = void (*)(Prm, const float4 *, const float4 *, float2 *, SbAsync);
            const KFn fns[3] = {&k_sb_async<0, false>, &k_sb_async<1, false>, &k_sb_async<2, false>};
            const int which = mode;
            if (!c->sbAsyncLdsRaised[which]) {
                HIPOK(hipFuncSetAttribute(reinterpret_cast<const void *>(fns[which]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SB_ASYNC_LDS));
                c->sbAsyncLdsRaised[which] = true;
            }
            hipLaunchKernelGGL(fns[which], dim3(grid), dim3(256), SB_ASYNC_LDS, c->stream, q, c->sbNatGain, c->sbNatSZ, xf, a);
        }
        LAUNCH_CHECK("k_sb_async");
        if (phase == 1) {
            c->sbp.active = true;
            c->sbp.p = p;
            return 0;
        }
        unsigned int ctl[4];
        HIPOK(hipMemcpyAsync(ctl, a.ctl, sizeof(ctl), hipMemcpyDeviceToHost, c->stream));
        HIPOK(wait_stream(c, 1));
        c->rs.fix_launches++;
        if (ctl[1] == 0) {
            done = true;
            c->rs.reruns_x += ctl[2];
            if (c->dbgLog) fprintf(stderr, "[csr] fwd_state_chain (barrier-free superblocks): %u delta runs, %u abandoned; superblock %d bins, %lld superblocks\n",
                                   ctl[2], ctl[3], v.B, (long long)v.NB);
        } else {
            c->rs.sb_bailouts++;
            if (c->dbgLog) fprintf(stderr, "[csr] fwd_state_chain (barrier-free superblocks): bailed out, running the pass form\n");
            // (a pipelined step may have tails of finished chains in flight that read the track the pass form rewrites)
            if (resume && c->tail) HIPOK(hipStreamSynchronize(c->tail));
        }
    }
    if (!done) {
        Scope sc(c, "fwd_state_chain");
        launch(0, 0);
    }
    LAUNCH_CHECK("k_sb_sys");
    int which = 0, burst = 2;
    for (int64_t it = 0; it <= v.NB + 1 && !done; ++it) {
        {
            Scope sc(c, "fwd_state_fix");
            for (int rep = 0; rep < burst; ++rep) {
                launch(which, 1);
                which ^= 1;
                c->rs.fix_launches++;
            }
        }
        LAUNCH_CHECK("k_sb_sys (repair)");
        CHECK(read_mail(c, MAIL_HDR));
        const unsigned int fresh = take_fresh(c, ST_X);
        if (c->dbgLog) fprintf(stderr, "[csr] fwd_state_fix (systolic superblocks) iter %lld reruns %u\n", (long long)it, fresh);
        if (fresh == 0) done = true;
        c->rs.reruns_x += fresh;
        burst = c->dbgLog ? 1 : std::min(32, burst * 2);
    }
    if (!done) return fail("fwd_state_chain (systolic superblocks): fix-up did not reach a fixed point");
    if (!resume) {
        Scope sc(c, "state_reblock_out");
        hipLaunchKernelGGL(k_import_tiled<float2>, dim3((int)(c->NG * (c->B / 32))), dim3(256), 0, c->stream, p,
                           reinterpret_cast<const float2 *>(natXf), p.tXf, (int64_t)0);
    }
    LAUNCH_CHECK("k_import_tiled_f2");
    c->xfNat = true;
    return 0;
}

// Join the side stream.  The per-chain sums (22 workgroups of 1024 threads, ~10 us) run on the main stream after the
// join: left on the side stream behind the epilogue they starve for whole-CU slots while a bandwidth-bound kernel of the
// main stream keeps the chip full (measured 0.64 ms instead of 0.01).
static void join_side(csr_ctx *c) {
    if (c->sidePending) {
        (void)hipStreamWaitEvent(c->stream, c->evJoin, 0);
        c->sidePending = false;
        if (c->sideSumsDone) { c->sideSumsDone = false; return; }    // the side stream ran the per-chain sums itself
        Scope sc(c, "chain_sums");
        hipLaunchKernelGGL(k_chain_sums, dim3((int)c->chains.size()), dim3(1024), 0, c->stream, c->sidePrm, c->dChainFirst,
                           c->dChainNb);
    }
}

// NIS/NLL epilogue; side = true runs it on the side stream (forked after the state chain) so that it overlaps the
// latency-bound smoother chain.
static int forward_epilogue(csr_ctx *c, const Prm &p, bool side) {
    hipStream_t st = c->stream;
    if (side) {
        join_side(c);
        HIPOK(hipEventRecord(c->evFork, c->stream));
        HIPOK(hipStreamWaitEvent(c->side, c->evFork, 0));
        st = c->side;
    }
    {
        Scope sc(c, "fwd_dstat", st);
        if (p.natD) hipLaunchKernelGGL(k_fwd_dstat<true>, dim3((int)c->NG), dim3(256), sizeof(float) * 64 * (c->B + 1), st, p);
        else hipLaunchKernelGGL(k_fwd_dstat<false>, dim3((int)c->NG), dim3(256), 0, st, p);
    }
    LAUNCH_CHECK("k_fwd_dstat");
    if (side) {
        HIPOK(hipEventRecord(c->evJoin, c->side));
        c->sidePending = true;
        c->sidePrm = p;                 // k_chain_sums follows on the main stream at the join
        return 0;
    }
    {
        Scope sc(c, "chain_sums", st);
        hipLaunchKernelGGL(k_chain_sums, dim3((int)c->chains.size()), dim3(1024), 0, st, p, c->dChainFirst, c->dChainNb);
    }
    LAUNCH_CHECK("k_chain_sums");
    return 0;
}

// main stream waits for the early covariance exports of the side stream (early_cov_exports)
static void join_pf(csr_ctx *c) {
    if (c->pfPending) {
        (void)hipStreamWaitEvent(c->stream, c->evPf, 0);
        c->pfPending = false;
    }
}
// constant process noise as rows of the reference-layout array (skipped when the array already holds exactly this fill)
static void pn_fill_values(const csr_ctx *c, const Prm &p, float q[4]) {
    q[0] = (float)p.Q00;
    q[1] = c->mdl.state_dim == 2 ? (float)p.Q01 : 0.f;
    q[2] = c->mdl.state_dim == 2 ? (float)p.Q10 : 0.f;
    q[3] = c->mdl.state_dim == 2 ? (float)p.Q11 : 0.f;
}
static bool pn_fill_current(const csr_ctx *c, const Prm &p) {
    float q[4];
    pn_fill_values(c, p, q);
    return c->pnFillValid && c->nat[CSR_ARR_PNOISE] != nullptr && memcmp(q, c->pnFillQ, sizeof(q)) == 0;
}
static int launch_pn_fill(csr_ctx *c, const Prm &p, float *dst, hipStream_t st) {
    const int nm = c->mdl.state_dim * c->mdl.state_dim;
    const unsigned grid = (unsigned)std::min<int64_t>((c->Npad + 255) / 256, 8192);
    float q[4];
    pn_fill_values(c, p, q);
    Scope sc(c, "export_natural", st);
    if (nm == 4) hipLaunchKernelGGL(k_fill_rows<4>, dim3(grid), dim3(256), 0, st, dst, c->Npad, q[0], q[1], q[2], q[3]);
    else hipLaunchKernelGGL(k_fill_rows<1>, dim3(grid), dim3(256), 0, st, dst, c->Npad, q[0], 0.f, 0.f, 0.f);
    LAUNCH_CHECK("k_fill_rows");
    memcpy(c->pnFillQ, q, sizeof(q));
    c->pnFillValid = true;
    return 0;
}

// Bit-exact mode: the state chain keeps at most 5/8 of the SIMDs busy for milliseconds and is bound by latency, not by
// bandwidth.  The reference-layout outputs that depend on the covariance chain alone -- Pf, and the process noise when it
// is one constant matrix -- are written on the side stream underneath it instead of after the smoother.
static int early_cov_exports(csr_ctx *c, const Prm &p, uint32_t flags, bool withPf = true) {
    const int nm = c->mdl.state_dim * c->mdl.state_dim;
    const bool constFlags = !(flags & (F_APN | F_QSCALE | F_KAPPA));
    const bool constQ = constFlags && p.chainQ == nullptr;
    // (round 4: with Pf also the process noise that is NOT one constant matrix -- per-bin multipliers: the covariance chain
    // stored it; per-chain base matrices: a table -- so that a step with multipliers pipelines its tail as well)
    const bool convQ = withPf && !constQ;
    if (!withPf && !constQ) return 0;
    const bool doPf = withPf && !c->pfNat;          // (the covariance chain may have written Pf in the reference layout itself)
    const bool fill = constQ && !pn_fill_current(c, p);
    if (constQ && !fill) c->pnNat = true;           // the array already holds this constant fill
    if (!doPf && !fill && !convQ) { if (withPf) c->pfNat = true; return 0; }
    // (a reference-layout array is allocated -- and zeroed ON THE MAIN STREAM -- at its first use: before the fork, so that the
    // side stream's writes are ordered behind the zeroing)
    float *dstPf = nullptr, *dstPn = nullptr;
    if (doPf) CHECK(nat_array(c, CSR_ARR_PF, &dstPf));
    if (fill || convQ) CHECK(nat_array(c, CSR_ARR_PNOISE, &dstPn));
    HIPOK(hipEventRecord(c->evFork2, c->stream));
    HIPOK(hipStreamWaitEvent(c->side, c->evFork2, 0));
    if (doPf || convQ) {
        ExpList L;
        memset(&L, 0, sizeof(L));
        L.count = 0;
        if (doPf) {
            L.count = 1;
            L.d[0].src = reinterpret_cast<const float *>(p.tPf); L.d[0].dst = dstPf; L.d[0].E = 4; L.d[0].n = nm;
        }
        if (convQ) {            // (as export_impl describes it)
            ExpDesc &e = L.d[L.count++];
            e.src = constFlags ? nullptr : reinterpret_cast<const float *>(p.tQ); e.dst = dstPn; e.E = 4; e.n = nm; e.skipLast = 1;
            if (constFlags) {
                e.cval[0] = (float)p.Q00;
                e.cval[1] = c->mdl.state_dim == 2 ? (float)p.Q01 : 0.f;
                e.cval[2] = c->mdl.state_dim == 2 ? (float)p.Q10 : 0.f;
                e.cval[3] = c->mdl.state_dim == 2 ? (float)p.Q11 : 0.f;
            }
        }
        Scope sc(c, "export_natural", c->side);
        hipLaunchKernelGGL(k_export_tiled, dim3((int)(c->NG * (c->B / 32))), dim3(256), 0, c->side, p, L);
    }
    LAUNCH_CHECK("k_export_tiled (early Pf)");
    if (withPf) c->pfNat = true;
    if (convQ) { c->pnNat = true; c->pnFillValid = false; }
    if (fill) {
        CHECK(launch_pn_fill(c, p, dstPn, c->side));
        c->pnNat = true;
    }
    HIPOK(hipEventRecord(c->evPf, c->side));
    c->pfPending = true;
    return 0;
}

// split: the caller pipelines everything behind the bit-exact state chain per group of chains (step_pipelined); if that chain
// went out as one barrier-free launch the pass returns right behind it (c->sbp.active) without the NIS / NLL epilogue.
static int forward_impl(csr_ctx *c, uint32_t flags, bool wantD, const unsigned char *active, bool defer = false,
                        bool side = false, bool natOut = false, bool split = false) {
    if (!c->statsValid) return fail("csr_batch_stats must run before the forward pass");
    // the resident statistics carry {S2c, log R} as a float32 pair (computed in the 2-ulp mode) and the context is back in the
    // bit-exact mode: that mode's D / NLL / lambda E-step promise float64 statistics -- recompute them
    if (c->p.statsF32 && c->xTolUlps == 0) CHECK(csr_batch_stats(c));
    // a MASKED pass rewrites the blocked xf / Pf of its own chains only: when the resident pass left them in the reference layout
    // alone, the chains outside the mask get their blocked copies back first (they keep their resident results)
    // (whether or not new statistics have invalidated those results since: the masked chains get new ones, the others keep the old)
    if (active != nullptr && (c->fwdBlockedStale || c->pfBlockedStale)) CHECK(ensure_blocked_fwd(c, nullptr));
    c->sbp.active = false;
    join_pf(c);         // (an early export nobody asked for afterwards still reads the arrays this pass overwrites)
    c->pfNat = c->pnNat = false;
    c->gainNat = false;
    Prm p = c->p;
    p.flags = flags;
    p.chainActive = active;
    if (c->kapIn) p.tKap = c->kapIn;                                  // ECM sweep: the kappa of the previous sweep's E-step
    p.qFromMult = (flags & (F_APN | F_QSCALE | F_KAPPA)) ? 0 : 1;     // constant process noise: pNoise is not stored
    // inner ECM sweeps: only the smoother reads this pass's pNoise (diagonal base process noise: its two diagonal entries,
    // 8 B instead of 16) and nobody its predicted variance (no NIS/NLL epilogue)
    p.qFromKappa = (c->sweepSkipQ && !p.qFromMult && !(flags & F_APN) && c->qDiagonal) ? 1 : 0;
    p.storePP = (wantD || !c->sweepSkipQ) ? 1 : 0;
    c->fwdQCompact = p.qFromKappa != 0;
    defer = defer && c->deferEnabled;
    c->fwdNat = false;
    c->xfNat = false;
    c->dNat = false;
    c->fwdBlockedStale = false;
    c->pfBlockedStale = false;
    c->pendFwdNat = natOut;
    const bool seq = (flags & F_APN) && !(flags & F_QSCALE);
    if (seq) {
        Scope sc(c, "fwd_apn_sequential");
        hipLaunchKernelGGL(k_fwd_apn, dim3(((int)c->chains.size() + 63) / 64), dim3(64), 0, c->stream, p, c->dChainFirst,
                           c->dChainNb);
        LAUNCH_CHECK("k_fwd_apn");
    } else {
        bool dP = defer && c->optimistic[ST_P], dX = defer && c->optimistic[ST_X];
        bool nisInChain = false, natOnly = false;
        if (wantD && natOut && c->natOutEnabled && c->natOutD) {        // D straight into the reference layout
            const size_t tileBytes = sizeof(float) * 64 * (size_t)(c->B + 1);
            bool ok = true;
            if (tileBytes > 48 * 1024 && !c->dstatLdsRaised) {      // 256-bin blocks: 65.8 KB of dynamic LDS
                ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fwd_dstat<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)tileBytes) == hipSuccess;
                (void)hipGetLastError();
                c->dstatLdsRaised = ok;
            }
            if (ok) {
                CHECK(nat_array(c, CSR_ARR_D, &p.natD));
                c->dNat = true;
            }
        }
        // Fused chain (tolerant mode).  Its state recursion warms up on SPECULATIVE gains (the split state chain reads the
        // validated ones), so with per-bin multipliers (the ECM loop: kappa per bin) a few blocks need about covariance-window +
        // state-window bins.  Round 1 ran every block with a 160-bin window for them (a failed validation cost a whole
        // pipeline).  With up to four confirmation passes per stage (check_stages) the few blocks that need more are simply
        // repaired: the window starts 16 bins above the plain one (96: ECM iteration 3.01 -> 2.85 ms at genome scale, 1.02 ->
        // 0.88 ms on a 1/8-genome shard) and widens itself like the others when many blocks fail.
        if (c->fuseFwd && c->xTolUlps > 0) {
            // one stage (counter of the covariance stage; the window covers the state chain's needs too)
            const bool mult = (flags & (F_KAPPA | F_LAMBDA | F_QSCALE)) != 0;
            if (c->warmP < c->warmX) c->warmP = c->warmX;
            if (c->warmFM < c->warmP + 16 && !c->pinFM) c->warmFM = c->warmP + 16;
            c->fwdWindow = mult ? &c->warmFM : &c->warmP;
            dX = false;
            p.predCompact = c->mdl.state_dim == 2 ? 1 : 0;
            if (!natOut && mult && c->mdl.state_dim == 2) {        // ECM sweep: window from the previous sweep's carries
                int *w = nullptr;
                CHECK(ws_prepare(c, p, true, &w));
                if (w) c->fwdWindow = w;
            }
            if (natOut && c->natOutEnabled && c->natOutFwd && c->mdl.state_dim == 2) {     // xf / Pf also in the reference layout
                CHECK(nat_array(c, CSR_ARR_XF, &p.natXs));
                CHECK(nat_array(c, CSR_ARR_PF, &p.natPs));
                p.natOut = 1;
                c->fwdNat = true;
                // the constant process noise depends on nothing: its reference-layout rows are filled on the side stream beside
                // the (latency-bound) forward chain instead of after the smoother
                if (c->earlyPf && active == nullptr) CHECK(early_cov_exports(c, p, flags, false));
                // NIS / NLL inside the chain's tile walker (csr_device.h FwdTrendFusedT::step_nis): no epilogue kernel, no
                // predicted-variance track.  (Per-bin NLL in D keeps the epilogue: its log terms belong off the serial path.)
                if (c->nisInChainEnabled && wantD && p.natD != nullptr && !(flags & F_NLL_IN_D)) {
                    p.nisInChain = 1;
                    p.storePP = 0;
                    nisInChain = true;
                }
                // constant process noise and nobody left to read the blocked xf / Pf but the smoother (which then reads the
                // reference-layout arrays through its tiles): the tile walker stores nothing in the blocked layout
                if (c->natOnlyEnabled && p.qFromMult && p.chainQ == nullptr && (nisInChain || !wantD) && (c->B % 8) == 0 && unit_f(c, p)) {
                    p.natOnly = 1;
                    natOnly = true;
                }
            }
            if (c->mdl.state_dim == 2 && unit_f(c, p)) CHECK(run_chain<FwdTrendFusedT<true>>(c, p, "fwd_chain", "fwd_fix", ST_P, dP));
            else if (c->mdl.state_dim == 2) CHECK(run_chain<FwdTrendFused>(c, p, "fwd_chain", "fwd_fix", ST_P, dP));
            else CHECK(run_chain<FwdLevelFused>(c, p, "fwd_chain", "fwd_fix", ST_P, dP));
            ws_launched(c, p, true);
            p.ckptIn = nullptr; p.ckptOut = nullptr; p.ckptSaveWarm = 0;
            c->lastFwdWindow = c->fwdWindow;
            c->fwdWindow = nullptr;
            c->fwdBlockedStale = natOnly;
        } else if (c->mdl.state_dim == 2) {
            c->lastFwdWindow = nullptr;
            const bool sbX = c->xTolUlps == 0 && c->sbState && !c->seqState;
            const bool seqX = c->xTolUlps == 0 && c->seqState;
            // (round 4: the covariance chain is validated optimistically here too.  Rounds 1-3 read its counters back before the
            // millisecond state chains -- one host round trip, ~0.1 ms per forward pass -- although a failed validation is as
            // rare here as in the 2-ulp mode and costs the same replay of the pass.  The sequential yardstick keeps the round trip.)
            if (seqX) dP = false;
            c->gainNat = false;
            Prm pc = p;
            if (sbX && CSR_GAIN_NAT) {
                // the superblock state chain reads its records in the reference layout: the covariance chain writes the gain
                // records there itself (and Pf, when this pass's Pf is an output), through LDS tiles -- no conversion launch
                // between the two chains; the NIS epilogue reads P00pred from the compact track instead of the blocked record
                if (!c->sbNatGain) { CHECK(dalloc(c, &c->sbNatGain, c->Npad)); CHECK(dalloc(c, &c->sbNatSZ, c->Npad)); }
                p.predCompact = 1;
                pc = p;
                pc.natOut = 1;
                pc.natLag = reinterpret_cast<float *>(c->sbNatGain);
                pc.natPs = nullptr;
                if (natOut && c->earlyPf && active == nullptr && c->natOutEnabled) {
                    CHECK(nat_array(c, CSR_ARR_PF, &pc.natPs));
                    c->pfNat = true;
                    // a pipelined step with one constant process noise: the smoother of every group reads xf / Pf in the reference
                    // layout (k_smooth_natin) -- no blocked copy of Pf
                    if (c->natOnlyEnabled && split && p.qFromMult && p.chainQ == nullptr && (c->B % 8) == 0 && unit_f(c, p)) {
                        pc.natOnly = 1;
                        c->pfBlockedStale = true;
                    }
                }
                c->gainNat = true;
            }
            if (unit_f(c, p)) CHECK(run_chain<FwdPTrendT<true>>(c, pc, "fwd_cov_chain", "fwd_cov_fix", ST_P, dP));
            else CHECK(run_chain<FwdPTrend>(c, pc, "fwd_cov_chain", "fwd_cov_fix", ST_P, dP));
            if (seqX) {
                // bit-exact mode: the state recursion cannot be validated speculatively in reasonable time (see
                // k_state_seq_trend) -- one wavefront per chain runs it sequentially on the validated gains
                Scope sc(c, "fwd_state_seq");
                if (p.F00 == 1.0 && p.F10 == 0.0 && p.F11 == 1.0)
                    hipLaunchKernelGGL(k_state_seq_trend<true>, dim3((unsigned)c->chains.size()), dim3(64), 0, c->stream, p,
                                       c->dChainFirst, c->dChainNb);
                else
                    hipLaunchKernelGGL(k_state_seq_trend<false>, dim3((unsigned)c->chains.size()), dim3(64), 0, c->stream, p,
                                       c->dChainFirst, c->dChainNb);
                LAUNCH_CHECK("k_state_seq_trend");
                dX = false;
            } else if (sbX) {
                // (the early covariance exports fork off behind the state chain's own record conversion)
                const bool early = natOut && c->earlyPf && active == nullptr && c->natOutEnabled;
                CHECK(state_chain_systolic(c, p, early, flags, split ? 1 : 0));
                dX = false;
            } else {
                // (round 6: the form that validated the levelTrend state chain on the batch's own blocks -- it only ever ran under the
                // retired switches CONSENRICH_AMD_FUSE=0 / CONSENRICH_AMD_SB_STATE=0 -- is gone)
                return fail("internal: no state chain for this mode");
            }
        } else {
            CHECK(run_chain<FwdPLevel>(c, p, "fwd_cov_chain", "fwd_cov_fix", ST_P, dP));
            CHECK(run_chain<FwdXLevel>(c, p, "fwd_state_chain", "fwd_state_fix", ST_X, dX));
        }
        if (wantD && nisInChain) {
            // the chain kernels left D and the per-block sums: only the per-chain reduction follows -- beside the smoother (side
            // stream) when one follows, so that its few microseconds leave the critical path
            join_side(c);
            hipStream_t st = c->stream;
            if (side && c->deferEnabled) {
                HIPOK(hipEventRecord(c->evFork, c->stream));
                HIPOK(hipStreamWaitEvent(c->side, c->evFork, 0));
                st = c->side;
            }
            {
                Scope sc(c, "chain_sums", st);
                hipLaunchKernelGGL(k_chain_sums, dim3((int)c->chains.size()), dim3(1024), 0, st, p, c->dChainFirst, c->dChainNb);
            }
            LAUNCH_CHECK("k_chain_sums");
            if (st == c->side) {
                HIPOK(hipEventRecord(c->evJoin, c->side));
                c->sidePending = true;
                c->sideSumsDone = true;         // join_side only waits
            }
        } else
        if (wantD && !c->sbp.active) CHECK(forward_epilogue(c, p, side && c->deferEnabled));
        if (dP || dX) {
            c->pendFwd = true;
            c->pendFlags = flags;
            c->pendWantD = wantD;
            c->pendActiveF = active;
        }
    }
    c->haveFwd = true;
    c->haveBwd = false;
    c->fwdInternal = true;
    c->fwdFlags = flags;
    return 0;
}

// The resident forward pass left xf / Pf in the reference layout only and a reader needs the blocked copies after all (a
// smoother pass without reference-layout outputs, per-chain base matrices): bring them back through LDS tiles.
static int ensure_blocked_fwd(csr_ctx *c, const unsigned char *active) {
    if (!c->fwdBlockedStale && !c->pfBlockedStale) return 0;
    float *natXf, *natPf;
    CHECK(nat_array(c, CSR_ARR_XF, &natXf));
    CHECK(nat_array(c, CSR_ARR_PF, &natPf));
    Prm p = c->p;
    p.chainActive = active;
    Scope sc(c, "state_reblock_out");
    if (c->fwdBlockedStale)
        hipLaunchKernelGGL(k_import_tiled<float2>, dim3((int)(c->NG * (c->B / 32))), dim3(256), 0, c->stream, p,
                           reinterpret_cast<const float2 *>(natXf), p.tXf, (int64_t)0);
    hipLaunchKernelGGL(k_import_tiled<float4>, dim3((int)(c->NG * (c->B / 32))), dim3(256), 0, c->stream, p,
                       reinterpret_cast<const float4 *>(natPf), p.tPf, (int64_t)0);
    LAUNCH_CHECK("k_import_tiled");
    if (active == nullptr) c->fwdBlockedStale = c->pfBlockedStale = false;      // (a masked import leaves the other chains stale)
    return 0;
}

// The resident smoother wrote xs / Ps / lag in the reference layout only (a step, forward_backward) and a MASKED ECM call is about
// to rewrite the blocked copies of its own chains: the chains outside the mask get theirs back first, so that the conversions
// that follow the call (exports, per-phase tracks, the background update: all from the blocked copies) return what is resident.
static int ensure_blocked_smooth(csr_ctx *c) {
    if (!c->smoothNat || c->mdl.state_dim != 2 || !c->nat[CSR_ARR_XS] || !c->nat[CSR_ARR_PS] || !c->nat[CSR_ARR_LAG]) return 0;
    float *xs, *Ps, *lag;
    CHECK(nat_array(c, CSR_ARR_XS, &xs));
    CHECK(nat_array(c, CSR_ARR_PS, &Ps));
    CHECK(nat_array(c, CSR_ARR_LAG, &lag));
    Prm p = c->p;
    p.chainActive = nullptr;
    const dim3 grid((int)(c->NG * (c->B / 32)));
    Scope sc(c, "state_reblock_out");
    hipLaunchKernelGGL(k_import_tiled<float2>, grid, dim3(256), 0, c->stream, p, reinterpret_cast<const float2 *>(xs), p.tXs, (int64_t)0);
    hipLaunchKernelGGL(k_import_tiled<float4>, grid, dim3(256), 0, c->stream, p, reinterpret_cast<const float4 *>(Ps), p.tPs, (int64_t)0);
    hipLaunchKernelGGL(k_import_tiled<float4>, grid, dim3(256), 0, c->stream, p, reinterpret_cast<const float4 *>(lag), p.tLag, (int64_t)0);
    LAUNCH_CHECK("k_import_tiled (smoothed)");
    return 0;
}

// estep: 0 = plain smoother; 1 = ECM sweep whose kappa E-step is evaluated inside the smoother chain, moments stored;
//        2 = same, but the smoothed moments are not stored (an inner sweep nobody reads them from)
static int backward_impl(csr_ctx *c, bool wantLag, const unsigned char *active, bool defer = false, bool natOut = false,
                         int estep = 0, bool preferNatIn = false) {
    if (!c->haveFwd) return fail("forward results are not resident: run csr_batch_forward first");
    Prm p = c->p;
    p.flags = c->fwdFlags;
    p.chainActive = active;
    p.estepKappa = estep != 0 ? 1 : 0;
    p.storeMoments = estep == 2 ? 0 : 1;
    if (c->kapIn) p.tKap = c->kapIn;
    p.tKapOut = c->kapOut ? c->kapOut : p.tKap;
    p.qFromKappa = c->fwdQCompact ? 1 : 0;       // the resident forward pass stored qf instead of pNoise
    c->pendEstep = estep;
    natOut = natOut && c->natOutEnabled && c->mdl.state_dim == 2;
    if (natOut) {
        CHECK(nat_array(c, CSR_ARR_XS, &p.natXs));
        CHECK(nat_array(c, CSR_ARR_PS, &p.natPs));
        CHECK(nat_array(c, CSR_ARR_LAG, &p.natLag));
        p.natOut = 1;
    }
    // preferNatIn (a group's tail of a pipelined step): the reference-layout xf / Pf of these chains stand -- read them there
    // even though a blocked xf is on its way for the epilogue
    if (c->fwdBlockedStale || c->pfBlockedStale || preferNatIn) {
        const bool pcq = p.chainQ != nullptr;
        const bool constQ = c->fwdInternal && !(c->fwdFlags & (F_APN | F_QSCALE | F_KAPPA));
        const bool natValid = c->fwdNat || (c->pfNat && (c->xfNat || preferNatIn));
        const bool need = c->fwdBlockedStale || c->pfBlockedStale;
        if (c->natInEnabled && natOut && !pcq && !p.qFromKappa && natValid && (need || constQ) && (stage_warm(c, ST_B) % 8) == 0 &&
            unit_f(c, p)) {
            float *natXf, *natPf;
            CHECK(nat_array(c, CSR_ARR_XF, &natXf));
            CHECK(nat_array(c, CSR_ARR_PF, &natPf));
            p.natIn = 1;
            p.natXfIn = reinterpret_cast<const float2 *>(natXf);
            p.natPfIn = reinterpret_cast<const float4 *>(natPf);
        } else {
            CHECK(ensure_blocked_fwd(c, active));
        }
    }
    c->smoothNat = natOut;
    c->pendNatOut = natOut;
    c->fitGen += 1;
    if (natOut) c->natSmoothGen = c->fitGen;        // this smoother writes xs / Ps in the reference layout itself
    // constant process noise (no kappa / qScale / adaptive noise): the smoother need not read pNoise at all
    p.qFromMult = (c->fwdInternal && !(c->fwdFlags & (F_APN | F_QSCALE | F_KAPPA))) ? 1 : 0;
    (void)wantLag;      // the lag-one covariance is produced by the smoother's own main phase
    const bool dB = defer && c->deferEnabled && c->optimistic[ST_B];
    if (!natOut && c->mdl.state_dim == 2 && estep != 0) {       // ECM sweep: window from the previous sweep's carries
        int *w = nullptr;
        CHECK(ws_prepare(c, p, false, &w));
        c->bwdWindow = w;
    }
    struct BwdWindowReset { csr_ctx *c; ~BwdWindowReset() { c->bwdWindow = nullptr; } } bwdWindowReset{c};
    if (p.qFromKappa && !natOut) {
        if (c->mdl.state_dim == 2 && unit_f(c, p)) CHECK(run_chain<BwdTrendQ2T<true>>(c, p, "bwd_chain", "bwd_fix", ST_B, dB));
        else if (c->mdl.state_dim == 2) CHECK(run_chain<BwdTrendQ2>(c, p, "bwd_chain", "bwd_fix", ST_B, dB));
        else CHECK(run_chain<BwdLevelQ2>(c, p, "bwd_chain", "bwd_fix", ST_B, dB));
    } else if (p.qFromKappa) {
        return fail("internal: compact process noise is only produced by ECM sweeps (no reference-layout outputs)");
    } else if (c->mdl.state_dim == 2 && unit_f(c, p)) CHECK(run_chain<BwdTrendT<true>>(c, p, "bwd_chain", "bwd_fix", ST_B, dB));
    else if (c->mdl.state_dim == 2) CHECK(run_chain<BwdTrend>(c, p, "bwd_chain", "bwd_fix", ST_B, dB));
    else CHECK(run_chain<BwdLevel>(c, p, "bwd_chain", "bwd_fix", ST_B, dB));
    ws_launched(c, p, false);
    c->lastBwdWindow = c->bwdWindow;
    if (dB) {
        c->pendBwd = true;
        c->pendActiveB = active;
    }
    c->haveBwd = true;
    return 0;
}

// Settle point: every optimistically launched stage is checked (one mailbox copy, one host sync).  If a stage re-ran
// blocks in its single validation pass, its results -- and everything computed from them -- are not yet the fixed
// point: the pipeline is re-run synchronously from that stage and the stage goes back to synchronous validation
// until a clean pass re-arms it.  After settle() the mailbox mirror holds the current per-chain sums.
// Consumes the stage counters of the mailbox mirror (which the caller has just refreshed).  Returns the first stage whose
// single optimistic validation pass re-ran blocks (-1: none did); such a stage goes back to synchronous validation and
// gets a wider window.
static int check_stages(csr_ctx *c) {
    int firstFail = -1;
    for (int stg = ST_P; stg <= ST_B; ++stg) {
        const unsigned int fresh = take_fresh(c, stg);
        unsigned int pass[MAX_DEFER_PASSES];
        for (int j = 0; j < MAX_DEFER_PASSES; ++j) pass[j] = take_fresh(c, MAIL_PASS0 + 4 * stg + j);
        const int np = c->launchedPasses[stg];
        if (fresh == 0) {
            // a long clean streak lets an extra confirmation pass go again
            if (np > 1 && ++c->cleanRuns[stg] >= 64) { c->nPasses[stg] = np - 1; c->cleanRuns[stg] = 0; }
            continue;
        }
        stage_reruns(c, stg) += fresh;
        int &wstage = (stg == ST_P && c->lastFwdWindow) ? *c->lastFwdWindow
                    : (stg == ST_B && c->lastBwdWindow) ? *c->lastBwdWindow : stage_warm(c, stg);
        if (&wstage == &c->wsWarmF || &wstage == &c->wsWarmB) {
            // a warm-started window (ECM sweeps) was too short at a wavefront's edge: widen IT (up to the cold window, where
            // warm starting switches itself off) and leave the stage's own policy alone
            const int npw = c->launchedPasses[stg];
            const bool stoodWs = npw > 1 && pass[npw - 1] == 0;
            if (c->dbgLog) fprintf(stderr, "[csr] settle: warm-started stage %d re-ran %u blocks, window %d -> %d\n", stg, fresh, wstage, wstage * 2);
            wstage = (wstage * 2 + 15) / 16 * 16;
            if (!stoodWs && firstFail < 0) firstFail = stg;
            continue;
        }
        c->cleanRuns[stg] = 0;
        grow_warm(c, wstage, pass[0]);
        const bool stands = np >= 1 && pass[np - 1] == 0;      // the last validation pass re-ran nothing: a fixed point
        if (c->dbgLog)
            fprintf(stderr, "[csr] settle: stage %d re-ran %u blocks (passes %u %u %u %u of %d) -> %s\n", stg, fresh, pass[0],
                    pass[1], pass[2], pass[3], np, stands ? "stands" : "replay");
        if (stands) continue;
        // failed: one more confirmation pass next time; at the limit the stage goes back to synchronous validation until a
        // few clean runs re-arm it, and its window widens by half (up to 4x the mode's default; beyond that the data simply
        // has long memory and repairing the few failing blocks is cheaper than a longer walk for every block)
        if (c->nPasses[stg] < MAX_DEFER_PASSES) c->nPasses[stg] += 1;
        else c->optimistic[stg] = false;
        if (c->adaptWarm) {
            int &w = wstage;
            const int cap = 4 * (c->xTolUlps > 0 ? 80 : 256);
            if (w < cap) w = std::min(cap, (w + w / 2 + 15) / 16 * 16);
        }
        if (firstFail < 0) firstFail = stg;
    }
    return firstFail;
}

static int settle(csr_ctx *c) {
    join_side(c);
    join_pf(c);
    if (!c->pendFwd && !c->pendBwd) return 0;
    CHECK(read_mail(c, c->mailBytes));
    const bool pf = c->pendFwd, pb = c->pendBwd;
    const uint32_t pe = c->pendExport;
    c->pendFwd = c->pendBwd = false;
    c->pendExport = 0;
    const int firstFail = check_stages(c);
    if (firstFail < 0) return 0;
    c->rs.pipeline_redos += 1;     // pipelines re-run after a failed optimistic validation
    if (firstFail <= ST_X && pf) {
        const bool bwdToo = pb || c->haveBwd;
        CHECK(forward_impl(c, c->pendFlags, c->pendWantD, c->pendActiveF, false, false, c->pendFwdNat));
        const bool nat = c->pendNatOut;
        const int es = c->pendEstep;
        if (bwdToo) CHECK(backward_impl(c, true, pb ? c->pendActiveB : c->pendActiveF, false, nat, es));
    } else if (pb) {
        CHECK(backward_impl(c, true, c->pendActiveB, false, c->pendNatOut, c->pendEstep));
    }
    if (pe) CHECK(export_impl(c, pe));      // arrays exported from the unvalidated results
    CHECK(read_mail(c, c->mailBytes));
    for (int stg = ST_P; stg <= ST_B; ++stg) (void)take_fresh(c, stg);
    return 0;
}

static int read_sums(csr_ctx *c, double *sum_d, double *sum_nll) {
    const size_t nc = c->chains.size();
    const bool pending = c->pendFwd || c->pendBwd;
    CHECK(settle(c));
    if (!pending) CHECK(read_mail(c, c->mailBytes));
    const double *hs = reinterpret_cast<const double *>(c->hMail + MAIL_HDR);
    if (sum_d) memcpy(sum_d, hs, sizeof(double) * nc);
    if (sum_nll) memcpy(sum_nll, hs + nc, sizeof(double) * nc);
    return 0;
}

extern "C" int csr_batch_forward(csr_ctx *c, uint32_t flags, double *sum_d, double *sum_nll) {
    CHECK(need(c));
    CHECK(settle(c));
    CHECK(forward_impl(c, flags, true, nullptr, true, false, true));
    if (sum_d || sum_nll) CHECK(read_sums(c, sum_d, sum_nll));
    return 0;
}

// chain_mask[c] == 0: chain c's resident forward results stay untouched (its sums are not updated)
extern "C" int csr_batch_forward_masked(csr_ctx *c, uint32_t flags, const unsigned char *chain_mask, double *sum_d,
                                        double *sum_nll) {
    if (!chain_mask) return csr_batch_forward(c, flags, sum_d, sum_nll);
    CHECK(need(c));
    CHECK(settle(c));
    const int nc = (int)c->chains.size();
    std::vector<unsigned char> act(chain_mask, chain_mask + nc);
    HIPOK(hipMemcpyAsync(c->dActive, act.data(), nc, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    CHECK(forward_impl(c, flags, true, c->dActive, true, false, false));
    if (sum_d || sum_nll) CHECK(read_sums(c, sum_d, sum_nll));
    return 0;
}

extern "C" int csr_batch_backward(csr_ctx *c) {
    CHECK(need(c));
    CHECK(backward_impl(c, true, nullptr, true, true));
    return 0;       // validated at the next settle point
}

// forward (NIS, optional NLL) + backward as one pipeline: one host synchronisation, the NIS/NLL epilogue overlapped
// with the smoother chain.  Equivalent to csr_batch_forward followed by csr_batch_backward.
extern "C" int csr_batch_forward_backward(csr_ctx *c, uint32_t flags, double *sum_d, double *sum_nll) {
    CHECK(need(c));
    CHECK(settle(c));
    CHECK(forward_impl(c, flags, true, nullptr, true, true, true));
    CHECK(backward_impl(c, true, nullptr, true, true));
    if (sum_d || sum_nll) return read_sums(c, sum_d, sum_nll);
    return 0;       // validation stays pending until the next settle point (sums, download, synchronize, new inputs)
}

// ---------------------------------------------------------------------------------------------------------------
// ECM (pyx:7660-8442 / 7153-7657) over all chains in lock-step; converged chains are masked out
// ---------------------------------------------------------------------------------------------------------------
struct EcmState {
    double prev = 1.0e16, cur = 0.0;
    bool haveInit = false, done = false;
};

extern "C" int csr_batch_ecm_masked(csr_ctx *c, const csr_ecm_cfg *cfg, uint32_t flags, const unsigned char *chain_mask,
                                    csr_ecm_out *out, double *nll_path);
extern "C" int csr_batch_ecm(csr_ctx *c, const csr_ecm_cfg *cfg, uint32_t flags, csr_ecm_out *out, double *nll_path) {
    return csr_batch_ecm_masked(c, cfg, flags, nullptr, out, nll_path);
}

// chain_mask[c] == 0: chain c is left exactly as it is (results of its last fit stay resident); out[c].skipped = 2
extern "C" int csr_batch_ecm_masked(csr_ctx *c, const csr_ecm_cfg *cfg, uint32_t flags, const unsigned char *chain_mask,
                                    csr_ecm_out *out, double *nll_path) {
    CHECK(need(c));
    CHECK(settle(c));
    if (!cfg || !out) return fail("null argument");
    if (!c->statsValid) CHECK(csr_batch_stats(c));
    c->multGen += 1;        // the E-steps rewrite the resident multipliers
    const int nc = (int)c->chains.size();
    uint32_t fl = flags & (F_QSCALE);
    if (cfg->use_lambda) fl |= F_LAMBDA;
    if (cfg->use_kappa) fl |= F_KAPPA;
    if (cfg->use_apn) fl |= F_APN;
    c->p.nu = cfg->nu;
    std::vector<EcmState> st(nc);
    std::vector<unsigned char> act(nc, 0);
    std::vector<double> nll(nc);
    for (int i = 0; i < nc; ++i) {
        csr_ecm_out &o = out[i];
        memset(&o, 0, sizeof(o));
    }
    auto push_active = [&]() -> int {
        HIPOK(hipMemcpyAsync(c->dActive, act.data(), nc, hipMemcpyHostToDevice, c->stream));
        HIPOK(hipStreamSynchronize(c->stream));
        return 0;
    };
    // tiny chains: filter + smoother + NLL only (pyx:7998-8129)
    bool anyTiny = false, anyBig = false;
    auto masked = [&](int i) { return chain_mask != nullptr && chain_mask[i] == 0; };
    bool anyMasked = false;
    for (int i = 0; i < nc; ++i) {
        if (masked(i)) { out[i].skipped = 2; anyMasked = true; continue; }
        if (c->chains[i].n <= 5) { act[i] = 1; anyTiny = true; out[i].skipped = 1; }
        else anyBig = true;
    }
    if (anyMasked && (anyTiny || anyBig)) CHECK(ensure_blocked_smooth(c));
    if (anyTiny) {
        CHECK(push_active());
        CHECK(forward_impl(c, fl | F_NLL, true, c->dActive, true));
        CHECK(backward_impl(c, true, c->dActive, true));
        CHECK(read_sums(c, nullptr, nll.data()));
        for (int i = 0; i < nc; ++i)
            if (act[i]) { out[i].final_nll = out[i].initial_nll = nll[i]; st[i].done = true; }
    }
    if (anyBig) {
        for (int i = 0; i < nc; ++i) act[i] = (c->chains[i].n > 5 && !masked(i)) ? 1 : 0;
        CHECK(push_active());
        bool fwdFresh = false;   // forward results already match the current multipliers
        struct SweepStateReset {     // also on the error paths
            csr_ctx *c;
            ~SweepStateReset() { c->sweepSkipQ = false; c->kapIn = c->kapOut = nullptr; c->wsActive = c->wsCold = false; }
        } sweepStateReset{c};
        c->wsSavedF = c->wsSavedB = 0;      // the first sweep of a call starts cold (the data / background may have changed)
        // kappa only (the reference CLI's default, constants.py:270-271): the smoother chain holds the moments of bins k
        // and k+1 and the lag covariance when it finishes bin k, so it evaluates the E-step itself; only the last inner
        // sweep's moments can become the result of this iteration, the others are not even stored.
        const bool fusedE = c->fuseEstep && cfg->use_kappa && !cfg->use_lambda && !cfg->use_apn && c->mdl.state_dim == 2;
        if (fusedE) {
            // The fused E-step must not overwrite the kappa its own sweep's forward pass ran with: were that pass's
            // deferred validation to fail, its re-run would read the NEXT sweep's kappa (one E-step ahead of pyx:8222-8300).
            // Sweeps therefore ping-pong two scratch buffers; the resident kappa changes only when an iteration is validated.
            for (float *&q : c->kapScratch)
                if (!q) CHECK(dalloc(c, &q, c->TN + (int64_t)c->B * 64));      // + the look-ahead padding group
        }
        // One ECM iteration (pyx:8156-8300) as launches only: t_inner x [forward, smoother + kappa E-step] + NLL forward.
        // Returns the buffer holding the iteration's final kappa (nullptr: no sweep ran, the resident one is unchanged).
        float *iterKappa = nullptr;
        auto launch_iteration = [&](bool defer, bool skipFirstForward) -> int {
            c->wsActive = true;
            float *cur = nullptr;                   // nullptr = the resident kappa (the one this iteration starts from)
            for (int64_t inner = 0; inner < cfg->inner_iters; ++inner) {
                c->kapIn = cur;
                c->sweepSkipQ = true;
                if (!(inner == 0 && skipFirstForward)) CHECK(forward_impl(c, fl, false, c->dActive, defer));
                c->sweepSkipQ = false;
                c->kapOut = c->kapScratch[inner & 1];
                CHECK(backward_impl(c, true, c->dActive, defer, false, inner + 1 == cfg->inner_iters ? 1 : 2));
                cur = c->kapOut;
                c->kapOut = nullptr;
            }
            c->kapIn = cur;
            CHECK(forward_impl(c, fl | F_NLL, true, c->dActive, defer));      // pyx:8300 (stores everything: it is the
            c->kapIn = nullptr;                                               // forward pass that stays resident)
            c->wsActive = false;
            iterKappa = cur;
            return 0;
        };
        const double *mailSums = reinterpret_cast<const double *>(c->hMail + MAIL_HDR);
        for (int64_t it = 0; it < cfg->max_iters; ++it) {
            if (fusedE) {
                // ONE settle point per iteration: every stage of every sweep is validated optimistically; if any of them
                // re-ran blocks, the whole iteration is replayed with synchronous validation from the kappa it started with
                // (still resident: it is replaced only below, once the iteration stands).
                const bool defer = c->deferIteration && c->deferEnabled;
                CHECK(launch_iteration(defer, fwdFresh));
                CHECK(read_mail(c, c->mailBytes));
                const bool hadPending = c->pendFwd || c->pendBwd;
                c->pendFwd = c->pendBwd = false;
                c->pendExport = 0;
                if (hadPending && check_stages(c) >= 0) {
                    c->rs.pipeline_redos += 1;
                    c->wsCold = true;               // the replay starts every window cold (and records fresh checkpoints)
                    const int rcReplay = launch_iteration(false, false);
                    c->wsCold = false;
                    CHECK(rcReplay);
                    CHECK(read_mail(c, c->mailBytes));
                    c->pendFwd = c->pendBwd = false;
                    for (int stg = ST_P; stg <= ST_B; ++stg) (void)take_fresh(c, stg);
                }
                if (iterKappa) {      // the iteration's kappa becomes the resident one (for the chains of this iteration)
                    bool everyChain = true;
                    for (int i = 0; i < nc; ++i) everyChain = everyChain && act[i] != 0;
                    if (everyChain) {
                        // every chain of the batch took part: the buffer the last sweep wrote BECOMES the resident one and the
                        // old resident buffer takes its place among the scratch buffers (no copy: 58 MB less per iteration at
                        // genome scale, one launch less on a shard)
                        float *old = c->p.tKap;
                        for (float *&q : c->kapScratch)
                            if (q == iterKappa) q = old;
                        c->p.tKap = iterKappa;
                        c->p.tKapOut = iterKappa;
                    } else {
                        Prm p = c->p;
                        p.chainActive = c->dActive;
                        Scope sc(c, "ecm_commit_kappa");
                        hipLaunchKernelGGL(k_copy_active_f32, dim3(grid_slots(c)), dim3(256), 0, c->stream, p, iterKappa, c->p.tKap);
                        LAUNCH_CHECK("k_copy_active_f32");
                    }
                }
                for (int i = 0; i < nc; ++i) nll[i] = mailSums[nc + i];
            } else {
                for (int64_t inner = 0; inner < cfg->inner_iters; ++inner) {
                    c->sweepSkipQ = true;
                    if (!fwdFresh) CHECK(forward_impl(c, fl, false, c->dActive, true));
                    fwdFresh = false;
                    CHECK(backward_impl(c, true, c->dActive, true, false, 0));
                    CHECK(settle(c));          // the E-step kernels consume validated results and update in place
                    c->sweepSkipQ = false;
                    Prm p = c->p;
                    p.flags = fl;
                    p.chainActive = c->dActive;
                    if (cfg->use_lambda) {
                        Scope sc(c, "estep_lambda");
                        hipLaunchKernelGGL(k_estep_lambda, dim3(grid_slots(c)), dim3(256), 0, c->stream, p);
                        LAUNCH_CHECK("k_estep_lambda");
                    }
                    if (cfg->use_kappa) {
                        Scope sc(c, "estep_kappa");
                        hipLaunchKernelGGL(k_estep_kappa, dim3(grid_slots(c)), dim3(256), 0, c->stream, p);
                        LAUNCH_CHECK("k_estep_kappa");
                    }
                }
                CHECK(forward_impl(c, fl | F_NLL, true, c->dActive, true));      // pyx:8300
                CHECK(read_sums(c, nullptr, nll.data()));
            }
            // the multipliers do not change until the next E-step: the next sweep may reuse this forward pass,
            // unless adaptive process noise made it depend on returnNLL-independent state only (it does not)
            fwdFresh = (cfg->inner_iters > 0);
            bool anyLeft = false, changed = false;
            for (int i = 0; i < nc; ++i) {
                if (!act[i]) continue;
                EcmState &s = st[i];
                csr_ecm_out &o = out[i];
                o.iters_done = it + 1;
                s.cur = nll[i];
                if (nll_path) nll_path[(int64_t)i * cfg->max_iters + it] = s.cur;
                const bool havePrev = s.haveInit;       // pyx:8337-8407
                if (!havePrev) { o.initial_nll = s.cur; s.haveInit = true; }
                else if (s.cur > s.prev + (1.0e-12 * std::fmax(std::fabs(s.prev), 1.0))) o.nll_increase_count += 1;
                double delta, scale;
                if (havePrev) { delta = std::fabs(s.cur - s.prev); scale = std::fabs(s.prev); }
                else { delta = 0.0; scale = std::fabs(s.cur); }
                if (std::fabs(s.cur) > scale) scale = std::fabs(s.cur);
                if (scale < 1.0) scale = 1.0;
                if (havePrev) { o.rel_improvement = (s.prev - s.cur) / scale; o.abs_rel_change = delta / scale; }
                else { o.rel_improvement = 0.0; o.abs_rel_change = 0.0; }
                const double tol = cfg->rtol * scale;
                s.prev = s.cur;
                if (havePrev && delta <= tol) o.stable_iters += 1; else o.stable_iters = 0;
                if (o.stable_iters >= 2) { o.converged = 1; s.done = true; act[i] = 0; changed = true; }
                else anyLeft = true;
            }
            if (!anyLeft) break;
            if (changed) CHECK(push_active());
        }
        for (int i = 0; i < nc; ++i) {
            if (c->chains[i].n <= 5 || masked(i)) continue;
            out[i].has_initial_nll = st[i].haveInit ? 1 : 0;
            out[i].final_nll = st[i].prev;
        }
    }
    c->sweepSkipQ = false;
    c->fwdFlags = fl;
    c->haveFwd = c->haveBwd = true;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// export / download
// ---------------------------------------------------------------------------------------------------------------

// Exports may be queued behind an optimistically validated pipeline: they are re-issued by settle() if it fails.
extern "C" int csr_batch_export(csr_ctx *c, uint32_t what) {
    CHECK(need(c));
    if (c->pendFwd || c->pendBwd) c->pendExport |= what;
    return export_impl(c, what);
}

extern "C" int csr_batch_sums(csr_ctx *c, double *sum_d, double *sum_nll) {
    CHECK(need(c));
    if (!c->haveFwd) return fail("no forward results");
    return read_sums(c, sum_d, sum_nll);
}

static int flush_export(csr_ctx *c, ExpList &L) {
    if (L.count == 0) return 0;
    {
        Scope sc(c, "export_natural");
        hipLaunchKernelGGL(k_export_tiled, dim3((int)(c->NG * (c->B / 32))), dim3(256), 0, c->stream, c->p, L);
    }
    LAUNCH_CHECK("k_export_tiled");
    L.count = 0;
    return 0;
}
static int add_export(csr_ctx *c, ExpList &L, int id, const float *src, int E, int n, int skipLast) {
    if (L.count == 8) CHECK(flush_export(c, L));       // one launch converts up to eight arrays
    float *dst;
    CHECK(nat_array(c, id, &dst));
    ExpDesc &d = L.d[L.count++];
    memset(&d, 0, sizeof(d));
    d.src = src; d.dst = dst; d.E = E; d.n = n; d.skipLast = skipLast;
    return 0;
}

// the smoothed state / one multiplier array into the reference layout unless the array there already is this fit's / these
// multipliers' (csr_ctx::natXsStamp, natMultStamp)
static int add_export_xs(csr_ctx *c, ExpList &L) {
    if (c->smoothNat || c->natXsStamp == c->fitGen) return 0;
    CHECK(add_export(c, L, CSR_ARR_XS, (const float *)c->p.tXs, 2, c->mdl.state_dim, 0));
    c->natXsStamp = c->fitGen;
    return 0;
}
static int add_export_mult(csr_ctx *c, ExpList &L, int id) {
    const int k = id == CSR_ARR_LAMBDA ? 0 : (id == CSR_ARR_KAPPA ? 1 : 2);
    if (c->nat[id] && c->natMultStamp[k] == c->multGen) return 0;
    const float *src = k == 0 ? c->p.tLam : (k == 1 ? c->p.tKap : c->p.tQs);
    CHECK(add_export(c, L, id, src, 1, 1, 0));
    c->natMultStamp[k] = c->multGen;
    return 0;
}

// residuals of the bins [off, off + nb) of the batch's natural layout (whole chains: off and nb are multiples of 64)
static int launch_resid(csr_ctx *c, int64_t off, int64_t nb, bool foldCheck) {
    const int d = c->mdl.state_dim;
    float *xs, *res;
    CHECK(nat_array(c, CSR_ARR_XS, &xs));
    CHECK(nat_array(c, CSR_ARR_RESID, &res));
    Scope sc(c, "residuals");
    Prm pr = c->p;
    pr.xTolUlps = c->xTolUlps;
    pr.prevKind = CK_NONE;
    // grid x 256 threads >= blocks: Npad / (K * 64) workgroups, K <= 4, block length >= 32
    if (foldCheck && off == 0 && (int64_t)((nb + 255) / 256) * 256 >= c->NB) take_pending_check(c, pr);
    pr.data += off;
    if (pr.bg) pr.bg += off;
    xs += off * d;
    res += off * c->m;
    if ((c->m & 3) == 0) {
        constexpr int K = 2;        // 64-bin sub-tiles per workgroup: 2 measured best (0.665 vs 0.684 ms with 1 or 4, round 3)
        const size_t lds = sizeof(float) * (size_t)(K * 64 + 4) * c->m;
        const dim3 grid((unsigned)((nb + K * 64 - 1) / (K * 64)));
        if (lds <= 65536)
            hipLaunchKernelGGL(k_resid_v4<2>, grid, dim3(256), lds, c->stream, pr, xs, d, res, nb);
        else
            hipLaunchKernelGGL(k_resid_v4<1>, dim3((unsigned)((nb + 63) / 64)), dim3(256), sizeof(float) * 68 * c->m,
                               c->stream, pr, xs, d, res, nb);
    } else
        hipLaunchKernelGGL(k_resid, dim3((int)((nb + 63) / 64)), dim3(256), sizeof(float) * 65 * c->m, c->stream,
                           pr, xs, d, res, nb);
    LAUNCH_CHECK("k_resid");
    return 0;
}

static int export_impl(csr_ctx *c, uint32_t what) {
    const int d = c->mdl.state_dim;
    const Prm &p = c->p;
    const int nv = d, nm = d * d;      // exported components of state vectors / covariance matrices
    ExpList L;
    memset(&L, 0, sizeof(L));
    // The NIS/NLL track is the only export that depends on the side stream's epilogue: when that is still running it is
    // converted last, after the (long, bandwidth-bound) residual kernel, so the epilogue leaves the critical path.
    // (Small batches are launch-bound: there the extra conversion launch costs more than the overlap saves.)
    const bool lateD = (what & CSR_EXPORT_FORWARD) && c->sidePending && (what & CSR_EXPORT_RESID) && !c->dNat &&
                       c->Npad >= ((int64_t)4 << 20);
    // dNat: the epilogue writes D in the reference layout itself -- nothing in this export depends on the side stream; it
    // is joined at the next settle point (sums, download, device_array, synchronize)
    if (!lateD && !c->dNat) join_side(c);
    if (what & CSR_EXPORT_FORWARD) {
        if (!c->haveFwd) return fail("no forward results to export");
        join_pf(c);
        if (!lateD && !c->dNat) CHECK(add_export(c, L, CSR_ARR_D, p.tD, 1, 1, 0));   // dNat: the epilogue wrote it already
        if (!c->fwdNat) {       // fwdNat: the forward chain already wrote both in the reference layout
            if (!c->xfNat) CHECK(add_export(c, L, CSR_ARR_XF, (const float *)p.tXf, 2, nv, 0));    // xfNat: the systolic state chain wrote it
            if (!c->pfNat) CHECK(add_export(c, L, CSR_ARR_PF, (const float *)p.tPf, 4, nm, 0));    // pfNat: written underneath the state chain
        }
        const bool constQ = c->fwdInternal && !(c->fwdFlags & (F_APN | F_QSCALE | F_KAPPA));
        if (c->pnNat) {
            // (the constant process noise was filled underneath the state chain as well)
        } else if (constQ && p.chainQ == nullptr) {
            // one Q0 for every bin of every chain: a streaming fill (rows past a chain's n-1 are padding nobody reads)
            if (!pn_fill_current(c, p)) {
                float *dst;
                CHECK(nat_array(c, CSR_ARR_PNOISE, &dst));
                CHECK(launch_pn_fill(c, p, dst, c->stream));
            }
        } else {
            CHECK(add_export(c, L, CSR_ARR_PNOISE, constQ ? nullptr : (const float *)p.tQ, 4, nm, 1));
            c->pnFillValid = false;
        }
        if (constQ && p.chainQ != nullptr) {
            ExpDesc &e = L.d[L.count - 1];
            e.cval[0] = (float)p.Q00;
            e.cval[1] = d == 2 ? (float)p.Q01 : 0.f;
            e.cval[2] = d == 2 ? (float)p.Q10 : 0.f;
            e.cval[3] = d == 2 ? (float)p.Q11 : 0.f;
        }
    }
    if (what & (CSR_EXPORT_SMOOTH | CSR_EXPORT_RESID)) {
        if (!c->haveBwd) return fail("no smoothed results to export");
        CHECK(add_export_xs(c, L));
    }
    if ((what & CSR_EXPORT_SMOOTH) && !c->smoothNat && c->natSmoothGen != c->fitGen) {      // smoothNat: the smoother already wrote the natural arrays; natSmoothGen: converted before
        CHECK(add_export(c, L, CSR_ARR_PS, (const float *)p.tPs, 4, nm, 0));
        CHECK(add_export(c, L, CSR_ARR_LAG, (const float *)p.tLag, 4, nm, 1));
    }
    if (what & CSR_EXPORT_SMOOTH) c->natSmoothGen = c->fitGen;
    if (what & CSR_EXPORT_MULT) {
        CHECK(add_export_mult(c, L, CSR_ARR_LAMBDA));
        CHECK(add_export_mult(c, L, CSR_ARR_KAPPA));
        CHECK(add_export_mult(c, L, CSR_ARR_QSCALE));
    }
    CHECK(flush_export(c, L));
    if (what & CSR_EXPORT_RESID) CHECK(launch_resid(c, 0, c->Npad, true));
    if (lateD) {
        join_side(c);
        CHECK(add_export(c, L, CSR_ARR_D, p.tD, 1, 1, 0));
        CHECK(flush_export(c, L));
    }
    return 0;
}

// One pass of the hot path in one call: statistics + forward (NIS / NLL) + backward + export of `what` + per-chain sums.
// Same launches as csr_batch_stats / csr_batch_forward_backward / csr_batch_export / csr_batch_sums in sequence -- a
// convenience for callers; the four separate calls cost the same (0.411 vs 0.412 ms on a 1/8-genome shard: the host runs
// ahead of the device, only the final read-back is a round trip).
// Everything of a step behind the filtered state -- blocked copy of xf, NIS / NLL epilogue, smoother, residuals -- for the
// chains of `dmask` (device bytes, nullptr = all), on c->stream.  `runs`: the maximal runs of those chains as bin ranges of the
// natural layout.
struct ChainRun { int64_t off, len, b0, b1; };     // bins [off, off + len) of the natural layout = blocks [b0, b1) of the batch
static int step_tail(csr_ctx *c, const Prm &pf, const unsigned char *dmask, const std::vector<ChainRun> &runs, uint32_t what) {
    Prm pt = pf;
    pt.chainActive = dmask;
    pt.prevKind = CK_NONE;
    float *natXf;
    CHECK(nat_array(c, CSR_ARR_XF, &natXf));
    // one constant process noise: the smoother reads xf / Pf of these chains in the reference layout (k_smooth_natin) and starts
    // at once; the blocked copy of xf only the NIS / NLL epilogue needs is made on the side stream in front of it
    const bool constQ = c->fwdInternal && !(c->fwdFlags & (F_APN | F_QSCALE | F_KAPPA));
    const bool natTail = c->natOnlyEnabled && c->natInEnabled && c->pfNat && constQ && pf.chainQ == nullptr && (c->B % 8) == 0 &&
                         (stage_warm(c, ST_B) % 8) == 0 && unit_f(c, pf);
    HIPOK(hipEventRecord(c->evFork, c->stream));
    HIPOK(hipStreamWaitEvent(c->side, c->evFork, 0));
    hipStream_t imp = natTail ? c->side : c->stream;
    // (the smoother is what the group's residuals wait for: when it reads the reference layout it is launched FIRST, the
    // epilogue's kernels of the side stream after it -- they were forked above and do not wait for it)
    if (natTail) CHECK(backward_impl(c, true, dmask, true, true, 0, true));
    if (!runs.empty()) {
        // ONE launch over the wavefront-groups from the first to the last block of these chains: the mask trims what lies between
        // (round 4: a launch per run of chains serialised four 60-us launches behind the state chain for a scattered last group)
        int64_t g0 = runs.front().b0 / 64, g1 = (runs.front().b1 + 63) / 64;
        for (const ChainRun &r : runs) { g0 = std::min(g0, r.b0 / 64); g1 = std::max(g1, (r.b1 + 63) / 64); }
        Scope sc(c, "state_reblock_out", imp);
        hipLaunchKernelGGL(k_import_tiled<float2>, dim3((int)((g1 - g0) * (c->B / 32))), dim3(256), 0, imp, pt,
                           reinterpret_cast<const float2 *>(natXf), pt.tXf, g0);
    }
    LAUNCH_CHECK("k_import_tiled_f2");
    // the NIS / NLL epilogue beside the (latency-bound) smoother chain, on the side stream
    if (!natTail) {
        HIPOK(hipEventRecord(c->evFork, c->stream));
        HIPOK(hipStreamWaitEvent(c->side, c->evFork, 0));
    }
    {
        Scope sc(c, "fwd_dstat", c->side);
        if (pt.natD) hipLaunchKernelGGL(k_fwd_dstat<true>, dim3((int)c->NG), dim3(256), sizeof(float) * 64 * (c->B + 1), c->side, pt);
        else hipLaunchKernelGGL(k_fwd_dstat<false>, dim3((int)c->NG), dim3(256), 0, c->side, pt);
    }
    LAUNCH_CHECK("k_fwd_dstat");
    HIPOK(hipEventRecord(c->evJoin, c->side));
    if (!natTail) CHECK(backward_impl(c, true, dmask, true, true, 0, false));
    CHECK(flush_pending_check(c));              // (the residual launches below cover a part of the batch each: no folded check)
    if (what & CSR_EXPORT_RESID)
        for (const ChainRun &r : runs) CHECK(launch_resid(c, r.off, r.len, false));
    HIPOK(hipStreamWaitEvent(c->stream, c->evJoin, 0));
    return 0;
}

// A step of the bit-exact mode whose tail is PIPELINED PER CHAIN behind the barrier-free state chain.  That launch lasts as
// long as the slowest chain needs (3.7 ms at genome scale) while most chains are final a millisecond earlier, and it leaves the
// memory system idle.  The kernel sets a host-visible word per chain when the chain is final; the host watches those words and,
// whenever the newly finished chains make up an eighth of the batch, launches their tail (blocked copy of xf, NIS / NLL
// epilogue, smoother, residuals; every kernel under a chain mask) on a second stream; the remainder follows when the state
// chain has ended.  Same kernels, same results; a chain's tail simply starts when ITS filtered state stands.
static int step_pipelined(csr_ctx *c, uint32_t flags, uint32_t what, bool *handled) {
    *handled = false;
    // (round 4: per-bin multipliers and per-chain base matrices pipeline as well -- their process noise goes to the reference
    // layout underneath the state chain with Pf, early_cov_exports; the sequential APN pass has no state chain to hide behind)
    if (!(c->tailSplit && c->xTolUlps == 0 && c->mdl.state_dim == 2 && c->sbState && !c->seqState && c->sbAsync &&
          c->deferEnabled && !((flags & F_APN) && !(flags & F_QSCALE)) &&
          !(what & CSR_EXPORT_MULT) && c->chains.size() >= 2 && c->chains.size() <= 4096))
        return 0;
    CHECK(settle(c));
    CHECK(forward_impl(c, flags, true, nullptr, true, false, true, true));
    if (!c->sbp.active) {               // the state chain did not go out as one launch: the pass is complete, carry on as usual
        CHECK(backward_impl(c, true, nullptr, true, true));
        if (what) CHECK(csr_batch_export(c, what));
        *handled = true;
        return 0;
    }
    const Prm pf = c->sbp.p;
    if (!(c->dNat && c->pfNat && c->pnNat)) {
        // an output of this pass still needs a conversion launch of its own (export_impl): no pipelining, finish in order
        CHECK(state_chain_systolic(c, pf, false, flags, 2));
        Prm pt = pf;
        float *natXf;
        CHECK(nat_array(c, CSR_ARR_XF, &natXf));
        hipLaunchKernelGGL(k_import_tiled<float2>, dim3((int)(c->NG * (c->B / 32))), dim3(256), 0, c->stream, pt,
                           reinterpret_cast<const float2 *>(natXf), pt.tXf, (int64_t)0);
        LAUNCH_CHECK("k_import_tiled_f2");
        CHECK(forward_epilogue(c, pf, true));
        CHECK(backward_impl(c, true, nullptr, true, true));
        if (what) CHECK(csr_batch_export(c, what));
        *handled = true;
        return 0;
    }
    const int nc = (int)c->chains.size();
    int64_t total = 0;
    for (const ChainInfo &ci : c->chains) total += ci.n;
    if (!c->dMask[0]) for (auto &m : c->dMask) CHECK(dalloc(c, &m, nc));
    if (c->hMaskPinChains < (size_t)nc) {
        if (c->hMaskPin) HIPOK(hipHostFree(c->hMaskPin));
        c->hMaskPin = nullptr;
        HIPOK(hipHostMalloc((void **)&c->hMaskPin, 8 * (size_t)nc, hipHostMallocDefault));
        c->hMaskPinChains = (size_t)nc;
    }
    std::vector<unsigned char> tailed((size_t)nc, 0);
    int phase = 0;
    bool any = false;
    hipStream_t mainStream = c->stream;
    auto launch_group = [&](const std::vector<unsigned char> &grp) -> int {
        // runs of consecutive chains -> bin ranges (a chain occupies [off, off + its length rounded up to 64))
        std::vector<ChainRun> runs;
        for (int i = 0; i < nc; ++i) {
            if (!grp[(size_t)i]) continue;
            const ChainInfo &ci = c->chains[(size_t)i];
            const int64_t l = (ci.n + 63) / 64 * 64;
            if (!runs.empty() && runs.back().off + runs.back().len == ci.off && runs.back().b1 == ci.b0) {
                runs.back().len += l;
                runs.back().b1 = ci.b0 + ci.nb;
            } else runs.push_back(ChainRun{ci.off, l, ci.b0, ci.b0 + ci.nb});
        }
        unsigned char *dm = c->dMask[phase];
        unsigned char *hm = c->hMaskPin + (size_t)phase * (size_t)nc;
        std::copy(grp.begin(), grp.end(), hm);
        HIPOK(hipMemcpyAsync(dm, hm, (size_t)nc, hipMemcpyHostToDevice, c->stream));
        CHECK(step_tail(c, pf, dm, runs, what));
        ++phase;
        any = true;
        return 0;
    };
    // ---- while the state chain runs: tails of the chains that are final, an eighth of the batch at a time, on the tail stream
    c->stream = c->tail;
    int rc = 0;
    // (the end of the launch is on the step's critical path -- the last tail starts behind it: the loop's sleeps run with a lowered
    // timer slack and, around the moment the PREVIOUS launch of this context ended, it polls without sleeping)
    TimerSlack slack;
    const auto tLoop = std::chrono::steady_clock::now();
    const double expectEnd = c->lastSbLoopUs;
    for (;;) {
        const hipError_t qs = hipStreamQuery(mainStream);
        const double el = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tLoop).count();
        if (qs != hipErrorNotReady) { if (qs != hipSuccess) rc = fail("state chain: %s", hipGetErrorString(qs)); else c->lastSbLoopUs = el; break; }
        // (bounded sleep-poll: the launch lasts milliseconds and a tail group is worth launching 20-50 us late; round 3 spun here)
        if (!(expectEnd > 0.0 && el > expectEnd - 100.0 && el < expectEnd + 200.0)) std::this_thread::sleep_for(std::chrono::microseconds(20));
        if (phase >= 6) continue;
        std::vector<unsigned char> grp((size_t)nc, 0);
        int64_t bins = 0;
        for (int i = 0; i < nc; ++i)
            if (!tailed[(size_t)i] && __atomic_load_n(&c->hDone[i], __ATOMIC_ACQUIRE) != 0u) { grp[(size_t)i] = 1; bins += c->chains[(size_t)i].n; }
        if (bins * 100 >= total * (phase == 0 ? c->tailFirstPct : c->tailNextPct)) {
            rc = launch_group(grp);
            if (rc) break;
            for (int i = 0; i < nc; ++i) if (grp[(size_t)i]) tailed[(size_t)i] = 1;
        }
    }
    c->stream = mainStream;
    if (rc) return rc;
    // ---- the launch has ended: its verdict (a bail-out runs the pass form and invalidates nothing that was final, but the
    // pass form rewrites the whole track from the cold prior: the tails already launched are waited for and everything is redone)
    const int64_t bail0 = c->rs.sb_bailouts;
    CHECK(state_chain_systolic(c, pf, false, flags, 2));
    if (c->rs.sb_bailouts != bail0 && any) {
        HIPOK(hipStreamSynchronize(c->tail));
        std::fill(tailed.begin(), tailed.end(), (unsigned char)0);
    }
    // ---- the remaining chains: on the main stream, behind the state chain and beside whatever the tail stream still has to do
    // (groups are disjoint sets of chains, hence of blocks: their kernels share no data; every group leaves no folded check behind)
    std::vector<unsigned char> rest((size_t)nc, 0);
    bool anyRest = false;
    for (int i = 0; i < nc; ++i) if (!tailed[(size_t)i]) { rest[(size_t)i] = 1; anyRest = true; }
    if (anyRest) CHECK(launch_group(rest));
    HIPOK(hipEventRecord(c->evTailJoin, c->tail));
    HIPOK(hipStreamWaitEvent(mainStream, c->evTailJoin, 0));
    {
        Scope sc(c, "chain_sums");
        Prm ps = pf;
        ps.chainActive = nullptr;
        hipLaunchKernelGGL(k_chain_sums, dim3((int)c->chains.size()), dim3(1024), 0, c->stream, ps, c->dChainFirst, c->dChainNb);
    }
    LAUNCH_CHECK("k_chain_sums");
    c->pendActiveB = nullptr;           // a replay after a failed optimistic validation covers every chain
    if (c->pendBwd) c->pendExport |= what;
    if (what & CSR_EXPORT_SMOOTH) c->natSmoothGen = c->fitGen;
    c->rs.tail_groups += phase;
    *handled = true;
    return 0;
}

extern "C" int csr_batch_step(csr_ctx *c, uint32_t flags, uint32_t what, double *sum_d, double *sum_nll) {
    CHECK(csr_batch_stats(c));
    bool handled = false;
    CHECK(step_pipelined(c, flags, what, &handled));
    if (!handled) {
        CHECK(csr_batch_forward_backward(c, flags, nullptr, nullptr));
        if (what) CHECK(csr_batch_export(c, what));
    }
    if (sum_d || sum_nll) return csr_batch_sums(c, sum_d, sum_nll);
    return 0;
}

extern "C" int csr_batch_step_forward(csr_ctx *c, uint32_t flags, uint32_t what, double *sum_d, double *sum_nll) {
    CHECK(csr_batch_stats(c));
    CHECK(settle(c));
    CHECK(forward_impl(c, flags, true, nullptr, true, false, true));
    if (what & CSR_EXPORT_FORWARD) CHECK(csr_batch_export(c, CSR_EXPORT_FORWARD));
    if (sum_d || sum_nll) return csr_batch_sums(c, sum_d, sum_nll);
    return 0;
}

extern "C" int csr_batch_device_array(csr_ctx *c, int32_t id, void **dev_ptr, int64_t *n_elems) {
    CHECK(need(c));
    CHECK(settle(c));
    if (id < 0 || id >= CSR_ARR_COUNT) return fail("bad array id");
    float *ptr;
    CHECK(nat_array(c, id, &ptr));
    if (dev_ptr) *dev_ptr = ptr;
    if (n_elems) *n_elems = arr_comps(c, id) * c->Npad;
    return 0;
}

extern "C" int csr_batch_download(csr_ctx *c, int32_t chain, int32_t id, void *host_dst) {
    CHECK(need(c));
    CHECK(settle(c));
    if (chain < 0 || chain >= (int)c->chains.size()) return fail("chain index out of range");
    if (id < 0 || id >= CSR_ARR_COUNT) return fail("bad array id");
    if (!host_dst) return fail("null host buffer");
    if (id == CSR_ARR_BACKGROUND && !c->nat[id]) {       // no background set yet: it is identically zero
        float *q;
        CHECK(nat_array(c, id, &q));
    }
    if (!c->nat[id]) return fail("array %d was not exported", id);
    const ChainInfo &ci = c->chains[chain];
    const int64_t per = arr_comps(c, id);
    int64_t rows = ci.n;
    if (id == CSR_ARR_PNOISE || id == CSR_ARR_LAG) rows = ci.n - 1;
    if (rows > 0)
        HIPOK(hipMemcpyAsync(host_dst, c->nat[id] + ci.off * per, sizeof(float) * per * rows, hipMemcpyDeviceToHost,
                             c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// synthetic fill (bench / scale tests)
// ---------------------------------------------------------------------------------------------------------------
extern "C" int csr_batch_synthesize(csr_ctx *c, uint64_t seed) {
    CHECK(need(c));
    CHECK(settle(c));
    if (!c->dLatent) CHECK(dalloc(c, &c->dLatent, c->Npad));
    std::vector<float> lat((size_t)c->Npad, 0.f);
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 12345;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t ch = 0; ch < c->chains.size(); ++ch) {
        const ChainInfo &ci = c->chains[ch];
        double x = 0.0;
        for (int64_t k = 0; k < ci.n; ++k) {
            // Irwin-Hall(12) normal approximation is plenty for a synthetic random walk
            double acc = 0.0;
            const uint64_t a = next(), b = next();
            for (int q = 0; q < 6; ++q) acc += (double)((a >> (q * 10)) & 1023) / 1024.0;
            for (int q = 0; q < 6; ++q) acc += (double)((b >> (q * 10)) & 1023) / 1024.0;
            x += 0.03 * (acc - 6.0);
            lat[(size_t)(ci.off + k)] = (float)x;
        }
    }
    HIPOK(hipMemcpy(c->dLatent, lat.data(), sizeof(float) * c->Npad, hipMemcpyHostToDevice));
    {
        Scope sc(c, "synthesize");
        hipLaunchKernelGGL(k_synth, dim3((int)((c->Npad + 255) / 256)), dim3(256), 0, c->stream, c->p, c->dLatent,
                           const_cast<float *>(c->p.data), const_cast<float *>(c->p.munc), seed, c->Npad);
    }
    LAUNCH_CHECK("k_synth");
    HIPOK(hipStreamSynchronize(c->stream));
    c->statsValid = c->haveFwd = c->haveBwd = false;
    return 0;
}

