// csr_lib.hip -- host side of libconsenrich_amd.so: context, device memory, launch orchestration and the C ABI
// declared in include/consenrich_amd.h.  gfx950 only.  No CPU compute path exists here: without a GPU every entry
// point fails loudly.
#include "../../include/consenrich_amd.h"
#include "csr_device.h"
#include "csr_background.h"
#include "csr_writers.h"
#include "csr_folds.h"
#include "csr_qseed.h"
#include "csr_qseed_post.h"
#include "csr_objective.h"
#include "csr_gain.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <chrono>
#include <sys/prctl.h>
#ifndef CSR_GAIN_NAT
#define CSR_GAIN_NAT 1       // 0: gain records through the blocked layout and a conversion launch (yardstick; scripts/ sweeps)
#endif
#ifndef CSR_STATS_NATSZ
#define CSR_STATS_NATSZ 1    // 0: the statistics records reach the reference layout through the conversion launch
#endif
#include <thread>
#include <type_traits>
#include <vector>

using namespace csr;

// ---------------------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}
#define HIPOK(expr)                                                                                       \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define CHECK(expr)                \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != 0) return rc_;  \
    } while (0)

extern "C" const char *csr_last_error(void) { return g_err; }
extern "C" int csr_abi_version(void) { return CSR_ABI_VERSION; }
#ifndef CSR_SOURCE_HASH
#define CSR_SOURCE_HASH "unknown"
#endif
#ifndef CSR_BUILD_FLAGS
#define CSR_BUILD_FLAGS ""
#endif
#define CSR_STR2(x) #x
#define CSR_STR(x) CSR_STR2(x)
// (the marker in front lets consenrich_amd/build.py find the record in the file without loading the library)
static const char g_buildId[] = "CSR_BUILD_ID:abi " CSR_STR(CSR_ABI_VERSION) " src " CSR_SOURCE_HASH " " CSR_BUILD_FLAGS;
extern "C" const char *csr_build_id(void) { return g_buildId + 13; }
extern "C" int csr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------------------
constexpr size_t MAIL_HDR = 80;     // 20 x u32 counters
constexpr int MAIL_PASS0 = 4, MAIL_DUMMY = 16, MAIL_LOCAL = 17, MAX_DEFER_PASSES = 4;    // MAIL_LOCAL: blocks repaired inside speculative kernels

struct ChainInfo {
    int64_t n;      // bins
    int64_t off;    // natural offset (multiple of 64)
    int64_t b0;     // first block
    int64_t nb;     // number of blocks
};

struct ProfEntry {
    int64_t launches = 0;
    double total_ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

// growable device work space owned by a context (freed with it)
struct DevBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&ptr, want);
        if (e != hipSuccess) return fail("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        cap = want;
        return 0;
    }
};

struct csr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // tuning
    int B = 0;                 // block length; 0 = chosen from the batch size at configure time
    // speculative warm-up in bins (multiples of 16).  Defaults follow the validation mode (mode_warm_defaults): bitwise
    // coalescence of float32-rounded trajectories needs ~4x the window that k-ulp agreement does.
    int warmP = 256, warmX = 256, warmB = 128;
    bool pinP = false, pinX = false, pinB = false, pinFM = false;
    int warmFM = 96;            // fused forward chain with per-bin multipliers (see forward_impl)
    int *fwdWindow = nullptr;   // window variable of the forward stage being launched (see stage_warm)
    int *bwdWindow = nullptr;   // ... of the smoother stage (warm-started ECM sweeps)
    int *lastBwdWindow = nullptr;
    // Warm-started speculation inside the ECM loop (Prm::ckptIn): a sweep's chains start their windows from the carries the
    // previous sweep recorded (double-buffered per direction), with windows of wsWarmF / wsWarmB bins instead of the cold
    // ones.  A failed validation at a wavefront's edge widens them; a window that changed since the checkpoints were
    // recorded makes the next sweep start cold once.
    bool wsEnabled = true;      // CONSENRICH_AMD_WARMSTART=0: off
    bool wsActive = false;      // set by the ECM loop around its sweeps
    bool wsCold = false;        // replay of a failed iteration: record checkpoints, do not start from them
    int wsWarmF = 32, wsWarmB = 32;         // warm-started windows (widen themselves when an edge block fails)
    static constexpr int wsMaxBlock = 32;   // largest block length (= batch size class) that warm-starts
    int wsSavedF = 0, wsSavedB = 0;         // window length the resident checkpoints were recorded for (0: none)
    int wsSweepF = 0, wsSweepB = 0;         // parity of the double buffers
    void *ckF[2] = {nullptr, nullptr}, *ckB[2] = {nullptr, nullptr};
    int *lastFwdWindow = nullptr;   // ... of the last forward stage launched (a failed settle widens that one)
    bool Bfixed = false;
    bool adaptWarm = true;
    bool useDmaWarm = true;    // ... and, with reference-layout outputs, for its warm-up phase
    bool useDmaFused = true;   // fused forward chain without reference-layout outputs: LDS-DMA ring
    bool useDma = true;        // LDS-DMA speculative kernels for the chains that provide them
    int xTolUlps = 0;           // carry validation: 0 = bit-exact sequential semantics (DEFAULT of every context since round 3: the
                                // only mode that holds the parity gate through the ECM loop on ill-conditioned data, tests/test_hard_data.py);
                                // k > 0 = k-ulp acceptance, the opt-in throughput mode (csr_set_validation / CONSENRICH_AMD_XTOL_ULPS)
    // batch
    bool configured = false;
    csr_model mdl{};
    int64_t m = 0;
    std::vector<ChainInfo> chains;
    int64_t Npad = 0, NB = 0, NG = 0, TN = 0;
    bool statsValid = false;
    bool haveFwd = false, haveBwd = false;
    uint64_t fitGen = 0, natSmoothGen = ~0ull;     // generation of the resident smoothed fit / of the natural xs + Ps arrays (csr_batch_gather_tracks)
    // Conversions of the smoothed state / the multipliers into the reference layout are remembered: a caller that asks chain by
    // chain (the per-phase run diagnostics: 22 chains x every ECM phase) converts the batch once per phase, not once per chain.
    // multGen counts changes of the resident multipliers (uploads, ECM calls); the stamps say what the arrays were converted from.
    uint64_t multGen = 0, natMultStamp[3] = {~0ull, ~0ull, ~0ull}, natXsStamp = ~0ull;
    bool smoothNat = false;     // the last smoother pass wrote xs / Ps / lag straight into the natural arrays (the
                                // block-transposed copies are stale; nothing but the ECM E-steps reads those)
    bool pendNatOut = false;
    bool fwdNat = false, pendFwdNat = false;   // the last forward pass wrote xf / Pf in the reference layout too
    bool dNat = false;          // ... and its NIS/NLL epilogue wrote D there (nothing left to convert)
    static constexpr bool natOutD = true;       // the NIS / NLL epilogue writes D in the reference layout itself
    bool dstatLdsRaised = false;
    // 2-ulp throughput mode, the byte cuts of round 5 (their A/B switch CONSENRICH_AMD_LEAN was retired in round 6; each of them
    // still has its own conditions at the call site -- per-bin NLL in D, per-bin process noise, per-chain Q):
    static constexpr bool statsF32Enabled = true;       // {S2c, log R} as one float32 pair
    static constexpr bool nisInChainEnabled = true;     // NIS / NLL terms inside the fused forward chain's tile walker (no epilogue kernel)
    static constexpr bool natOnlyEnabled = true;        // constant process noise: xf / Pf only in the reference layout, the smoother reads them there
    bool natInEnabled = true;           // CONSENRICH_AMD_NATIN=0 (tests): the smoother never reads the reference layout -- blocked copies
                                        // a forward pass did not write are brought back first (ensure_blocked_fwd)
    bool fwdBlockedStale = false;       // the resident forward pass wrote xf / Pf in the reference layout ONLY (blocked tXf / tPf are stale)
    bool pfBlockedStale = false;        // ... Pf alone (default mode: the covariance chain of a pipelined step writes it in the reference layout only)
    bool sideSumsDone = false;          // the pending side-stream work already includes the per-chain sums (join_side only waits)
    double lastSbLoopUs = 0.0;          // how long the host watched the previous single launch of the state chain (step_pipelined)
    double lastWaitUs[2] = {0.0, 0.0};  // how long the previous host wait for the stream lasted, per wait site (wait_stream polls around that moment)
    bool sbAsyncLdsRaised[3] = {false, false, false};   // per context = per device (HIP keeps the attribute per device)
    int pendEstep = 0;
    static constexpr bool fuseEstep = true;     // ECM: kappa E-step inside the smoother chain (levelTrend, no lambda re-weighting)
    bool fwdInternal = false;   // forward results were produced by this library (vs imported through csr_backward_pass)
    uint32_t fwdFlags = 0;
    Prm p{};
    std::vector<void *> allocs;
    // device arrays not in Prm
    int64_t *dChainFirst = nullptr, *dChainNb = nullptr;
    int64_t *dChainOff = nullptr, *dChainLen = nullptr;     // natural offset / length of every chain (bins)
    unsigned char *dActive = nullptr;
    float *dLatent = nullptr;
    float *nat[CSR_ARR_COUNT] = {nullptr};
    // mailbox: [20 x u32 monotonic re-run counters: 0-3 cumulative per stage (covariance, state, smoother, debug), 4-15 per
    // stage and validation-pass index, 16 scratch | sumD[nchains] | sumNLL[nchains]] in device memory, mirrored into pinned
    // host memory with ONE copy per settle point
    char *dMail = nullptr, *hMail = nullptr;
    size_t mailBytes = 0;
    unsigned int lastCnt[20] = {0};
    // deferred validation launches nPasses[stage] validation passes back to back; the stage stands iff the LAST of them
    // re-ran nothing (then it was a fixed point).  Isolated speculation failures -- the normal case when the data has a
    // longer memory than the window (small Q0) -- are repaired by pass 1 and confirmed by pass 2 without a pipeline replay.
    int nPasses[3] = {1, 1, 1};
    int cleanRuns[3] = {0, 0, 0};
    int launchedPasses[3] = {1, 1, 1};
    // deferred validation: a stage whose last synchronous run needed no re-run is launched optimistically (speculative
    // pass + one validation pass, no host round trip); the counters are checked at the next settle point and the
    // pipeline is re-run synchronously from the first stage that did re-run blocks.
    bool deferEnabled = true;
    static constexpr bool spinWait = true;
    static constexpr bool fuseFwd = true;       // tolerant validation: covariance and state chains advance in one kernel
    static constexpr bool unitF1Enabled = true; // F01 == 1: the superblock walker's predicted level is one float32 add
    static constexpr bool unitFEnabled = true;  // F = [[1, f], [0, 1]] (constructMatrixF) runs the UF instances of the levelTrend chains; any other F the general ones
    bool seqState = false;      // bit-exact validation, levelTrend: one wavefront per chain walks the state chain sequentially (CONSENRICH_AMD_SEQ_STATE=1)
    // bit-exact validation, levelTrend (default): the state chain speculates on SUPERBLOCKS of sbBins bins with an sbWarm-bin
    // window -- two float32-rounded state trajectories need ~10^4 bins to coincide bit for bit (scripts/ubench/merge_time.c),
    // so the batch's own 32..256-bin blocks never validate; the gain / statistics records are re-blocked into a second view
    // of the batch (own block table and carries) for this one chain and the filtered state is re-blocked back
    static constexpr bool sbState = true;       // (round 6: the form that speculated on the batch's own blocks is gone; CONSENRICH_AMD_SEQ_STATE=1 is the sequential yardstick)
    int sbBins = 8192;          // CONSENRICH_AMD_SB_BINS (default: chosen from the batch, ensure_sb_view)
    // k_sb_delta's fallback rule (CONSENRICH_AMD_SB_ADV = "min,from"): walk the rest of a batch when, from round `from` on, the
    // rounds have settled fewer than `min` bins each.  Measured flat between "give up after 20 rounds" (4,20) and "walk as soon
    // as a round is worth less than its six steps" (6,2): 3.75-3.95 ms of repairs either way (profiles/r03_sb_sweeps.txt) -- where
    // the levels flip densely a round and the steps it replaces cost the same.
    int sbAdvMin = 4, sbAdvFrom = 20;
    bool sbBinsPinned = false;  // CONSENRICH_AMD_SB_BINS given: no automatic choice of the superblock length
    // CONSENRICH_AMD_SB_ASYNC=0: speculative pass + repair passes as separate launches (k_sb_sys / k_sb_delta) instead of the
    // barrier-free single launch (k_sb_async); sbSpinLimit bounds every wait inside it (polls of ~2 us; then: bail out to the pass form)
    bool sbAsync = true;
    int sbSpinLimit = 1 << 19;
    bool natSZValid = false;    // sbNatSZ holds the current statistics of every chain
    bool gainNat = false;       // this forward pass's covariance chain wrote sbNatGain itself (walk_nat_gain)
    float4 *sbNatGain = nullptr, *sbNatSZ = nullptr;    // natural-layout records of the systolic walker (freed with the batch)
    bool xfNat = false;         // the resident forward pass left xf in the reference layout already (systolic walker)
    struct SbView {
        bool ready = false;
        int B = 0;
        int64_t NB = 0, NG = 0, TN = 0;
        int4 *blk = nullptr;
        int *blkChain = nullptr;
        int64_t *chainFirst = nullptr;
        void *carryIn = nullptr, *carryOutA = nullptr, *carryOutB = nullptr;
        unsigned long long *pub = nullptr;      // k_sb_async: carry[NB], {version, final}[NB], control words
    } sb;
    static constexpr bool natOutEnabled = true; // the smoother writes the reference layout directly
    static constexpr bool natOutFwd = true;     // ... and so does the fused forward chain
    // debugging switches, read once from the environment at creation (never on the launch path)
    bool dbgLog = false;
    bool optimistic[3] = {true, true, true};
    bool pendFwd = false, pendBwd = false, sidePending = false;
    double *dChainQ = nullptr;  // per-chain base process noise (csr_batch_set_chain_q), freed with the batch
    // ECM with the kappa E-step inside the smoother: a sweep's smoother writes kappa into a scratch buffer (kapOut), the next
    // sweep's forward pass reads it (kapIn); the resident tKap only changes when a whole iteration has been validated
    // (nullptr = the resident array).  Allocated on first use, freed with the batch.
    float *kapScratch[2] = {nullptr, nullptr};
    float *kapIn = nullptr, *kapOut = nullptr;
    static constexpr bool deferIteration = true;    // ECM (fused E-step): one settle point per iteration, replay on a failed validation
    bool sweepSkipQ = false;    // the forward pass being launched is an inner ECM sweep (Prm::qFromKappa, storePP)
    bool fwdQCompact = false;   // the resident forward pass stored the diagonal of pNoise (tQ2) instead of pNoise (tQ)
    bool qDiagonal = true;      // the base process noise in use (model's, or every chain's) is diagonal
    bool modelQDiagonal = true, chainQDiagonal = true;
    Prm sidePrm{};              // parameters of the epilogue running on the side stream (its sums follow at the join)
    uint32_t pendFlags = 0, pendExport = 0;
    bool pendWantD = false;
    const unsigned char *pendActiveF = nullptr, *pendActiveB = nullptr;
    // device-resident background update (allocated on first use, freed with the batch)
    struct BgState {
        bool ready = false, haveCur = false;
        int Bp = 0;
        BgPrm prm{};
        BgBatch bat{};
        int *dGroupChain = nullptr;
        double *out1 = nullptr;
        unsigned char *dActive = nullptr, *dHasSup = nullptr;
        double *dPen = nullptr;
        long long *dSelRank = nullptr;
    } bg;
    DevBuf qsBuf, qpBuf;                // Q0-seed work space (sampling / posterior)
    DevBuf stageBuf;                    // host -> device staging of per-bin vectors (csr_batch_upload_multipliers)
    DevBuf bgBuf, wrBuf, textBuf;       // host-buffer background solver / bedGraph writer work space (this device)
    hipStream_t side = nullptr;         // NIS/NLL epilogue runs here, concurrently with the smoother chain
    // Folded validation (Prm::prevKind): a clean optimistic stage leaves its check to the next speculative kernel; the two use
    // different carry sets.  pendChk = the check that has not been handed to a kernel yet (flushed by read_mail).
    struct PendingCheck {
        bool valid = false;
        int kind = 0, stage = 0;
        const void *cin = nullptr, *cout = nullptr;
        const unsigned char *active = nullptr;
    } pendChk;
    void *carrySet[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [set][carryIn, carryOutA]
    int carryToggle = 0;
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    hipEvent_t evFork2 = nullptr, evPf = nullptr;      // early covariance exports on the side stream (bit-exact mode)
    // step_pipelined: tails of the chains whose filtered state stands, on a stream of their own while the state chain runs
    hipStream_t tail = nullptr;
    hipEvent_t evTailJoin = nullptr;
    // first-use zeroing of a reference-layout array (nat_array) runs HERE and is waited for by the host before the array is
    // handed out: it is ordered against no other stream, so it does not matter which stream the caller is on at that moment
    hipStream_t zeroStream = nullptr;
    hipStream_t mainStream = nullptr;   // what `stream` is outside step_pipelined's tail groups (csr_run_stats.nat_first_use_off_main)
    struct SbPending { bool active = false; Prm p{}; } sbp;     // a state chain launched and not yet waited for (step_pipelined)
    unsigned int *hDone = nullptr, *dDone = nullptr;    // host-visible "chain is final" words (pinned; device alias)
    unsigned char *dMask[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    unsigned char *hMaskPin = nullptr;          // pinned staging of the eight masks (an upload from pageable memory may hold the host)
    size_t hMaskPinChains = 0;
    // CONSENRICH_AMD_TAIL_PCT="first,next": share of the batch's bins a group of finished chains must reach.  Round 4: 60 / 40 (round 3:
    // 50 / 15) -- tail kernels take issue slots from the walking wavefronts, so fewer, later groups win (profiles/r04_tail_sweep.txt)
    int tailFirstPct = 60, tailNextPct = 40;
    bool tailSplit = true;      // CONSENRICH_AMD_TAIL_SPLIT=0: a step's tail follows the state chain for all chains at once
    bool pfPending = false, pfNat = false, pnNat = false;
    // the reference-layout process-noise array holds the constant fill of THESE values in every row (k_fill_rows): a step with the
    // same constant process noise does not write it again (a 1/8-genome step: one side-stream launch and two stream waits less)
    bool pnFillValid = false;
    float pnFillQ[4] = {0.f, 0.f, 0.f, 0.f};
    static constexpr bool earlyPf = true;       // Pf / constant pNoise are exported underneath the state chain
    // profiling
    bool profiling = false;
    std::map<std::string, ProfEntry> prof;
    std::vector<hipEvent_t> eventPool;
    // stats
    csr_run_stats rs{};
};

static int settle(csr_ctx *c);
static int export_impl(csr_ctx *c, uint32_t what);

static int ctx_select(csr_ctx *c) {
    HIPOK(hipSetDevice(c->device));
    return 0;
}

template <class T>
static int dalloc(csr_ctx *c, T **ptr, int64_t count) {
    void *q = nullptr;
    const size_t bytes = (size_t)std::max<int64_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    c->allocs.push_back(q);
    *ptr = reinterpret_cast<T *>(q);
    return 0;
}

static void free_batch(csr_ctx *c) {
    for (void *q : c->allocs) (void)hipFree(q);
    c->allocs.clear();
    c->configured = false;
    c->statsValid = c->haveFwd = c->haveBwd = false;
    c->pendFwd = c->pendBwd = c->sidePending = false;
    c->pendExport = 0;
    c->dMail = nullptr;
    c->dChainQ = nullptr;
    c->pendChk = csr_ctx::PendingCheck{};
    c->carrySet[0][0] = c->carrySet[0][1] = c->carrySet[1][0] = c->carrySet[1][1] = nullptr;
    c->kapScratch[0] = c->kapScratch[1] = nullptr;
    c->kapIn = c->kapOut = nullptr;
    c->bg = csr_ctx::BgState{};
    c->dActive = nullptr;
    c->sb = csr_ctx::SbView{};
    c->sbNatGain = c->sbNatSZ = nullptr;
    c->natSZValid = false;
    if (c->tail) (void)hipStreamSynchronize(c->tail);
    if (c->hDone) { (void)hipHostFree(c->hDone); c->hDone = nullptr; c->dDone = nullptr; }
    if (c->hMaskPin) { (void)hipHostFree(c->hMaskPin); c->hMaskPin = nullptr; c->hMaskPinChains = 0; }
    for (auto &m : c->dMask) m = nullptr;
    c->sbp.active = false;
    c->pfPending = false;
    c->xfNat = false;
    c->fwdNat = c->pfNat = c->pnNat = c->dNat = c->smoothNat = false;
    c->fwdBlockedStale = c->pfBlockedStale = false;     // (the reference-layout arrays they point at are gone)
    c->pnFillValid = false;
    c->ckF[0] = c->ckF[1] = c->ckB[0] = c->ckB[1] = nullptr;
    c->wsSavedF = c->wsSavedB = 0;
    c->wsActive = c->wsCold = false;
    for (auto &n : c->nat) n = nullptr;
    c->natXsStamp = c->natMultStamp[0] = c->natMultStamp[1] = c->natMultStamp[2] = ~0ull;
}

// Warm-up windows that gave zero re-runs on the bench workload with margin (hg38 x 32 synthetic: the state chain needs
// 64 bins at k = 2, the covariance chain 64, the smoother 48; exact mode 256 / 256 / 128).  run_chain lengthens them
// when the data has a longer filter memory; results never depend on them.
static void mode_warm_defaults(csr_ctx *c) {
    const bool tol = c->xTolUlps > 0;
    c->optimistic[1] = tol;     // bit-exact state chains re-run blocks on most calls: validate them synchronously
    if (!c->pinP) c->warmP = tol ? 80 : 256;
    if (!c->pinX) c->warmX = tol ? 80 : 256;
    if (!c->pinB) c->warmB = tol ? 64 : 128;
}

extern "C" csr_ctx *csr_create(int device_ordinal) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        fail("no HIP device visible: consenrich_amd has no CPU fallback");
        return nullptr;
    }
    if (device_ordinal < 0 || device_ordinal >= n) {
        fail("device ordinal %d out of range (0..%d)", device_ordinal, n - 1);
        return nullptr;
    }
    csr_ctx *c = new csr_ctx();
    c->device = device_ordinal;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) {
        fail("cannot initialise device %d", device_ordinal);
        delete c;
        return nullptr;
    }
    c->mainStream = c->stream;
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evFork2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evPf, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&c->tail, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->zeroStream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->evTailJoin, hipEventDisableTiming) != hipSuccess) {
        fail("cannot create the side stream of device %d", device_ordinal);
        delete c;
        return nullptr;
    }
    const char *e;
    if ((e = getenv("CONSENRICH_AMD_BLOCK"))) { c->B = atoi(e); c->Bfixed = c->B != 0; }
    if ((e = getenv("CONSENRICH_AMD_WARM"))) {
        // "p,x,b[,fm]": warm-up windows (bins) of the covariance / state / smoother chains and of the fused forward chain with
        // per-bin multipliers; -1 leaves one at its default (csr_set_tuning is the API for the first three)
        int w[4] = {-1, -1, -1, -1};
        (void)sscanf(e, "%d,%d,%d,%d", &w[0], &w[1], &w[2], &w[3]);
        if (w[0] >= 0) { c->warmP = w[0]; c->pinP = true; }
        if (w[1] >= 0) { c->warmX = w[1]; c->pinX = true; }
        if (w[2] >= 0) { c->warmB = w[2]; c->pinB = true; }
        if (w[3] >= 0) { c->warmFM = w[3]; c->pinFM = true; }
    }
    if ((e = getenv("CONSENRICH_AMD_XTOL_ULPS"))) c->xTolUlps = atoi(e);
    if ((e = getenv("CONSENRICH_AMD_DEFER"))) c->deferEnabled = atoi(e) != 0;
    if ((e = getenv("CONSENRICH_AMD_NATIN"))) c->natInEnabled = atoi(e) != 0;
    if ((e = getenv("CONSENRICH_AMD_SEQ_STATE"))) c->seqState = atoi(e) != 0;
    if ((e = getenv("CONSENRICH_AMD_WARMSTART"))) c->wsEnabled = atoi(e) != 0;
    if ((e = getenv("CONSENRICH_AMD_SB_ADV"))) {
        int a = 4, f = 20;
        if (sscanf(e, "%d,%d", &a, &f) >= 1) { c->sbAdvMin = std::min(255, std::max(0, a)); c->sbAdvFrom = std::min(255, std::max(1, f)); }
    }
    if ((e = getenv("CONSENRICH_AMD_SB_BINS"))) { c->sbBins = std::max(64, (atoi(e) + 63) / 64 * 64); c->sbBinsPinned = true; }
    if ((e = getenv("CONSENRICH_AMD_SB_ASYNC"))) c->sbAsync = atoi(e) != 0;
    if ((e = getenv("CONSENRICH_AMD_TAIL_SPLIT"))) c->tailSplit = atoi(e) != 0;
    if ((e = getenv("CONSENRICH_AMD_TAIL_PCT"))) {
        int a = 60, b = 40;
        if (sscanf(e, "%d,%d", &a, &b) >= 1) { c->tailFirstPct = std::min(100, std::max(1, a)); c->tailNextPct = std::min(100, std::max(1, b)); }
    }
    if ((e = getenv("CONSENRICH_AMD_SB_SPIN_LIMIT"))) c->sbSpinLimit = std::max(1, atoi(e));
    c->dbgLog = getenv("CONSENRICH_AMD_DEBUG") != nullptr;
    mode_warm_defaults(c);
    if ((e = getenv("CONSENRICH_AMD_DMA"))) c->useDma = c->useDmaFused = c->useDmaWarm = atoi(e) != 0;     // 0: the plain-load forms of the chains (yardstick of the LDS-DMA ring tests)
    if (c->B != 0 && (c->B < 32 || (c->B % 32) != 0)) c->B = 0;
    return c;
}

extern "C" void csr_destroy(csr_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->side) (void)hipStreamSynchronize(c->side);
    free_batch(c);
    for (auto &kv : c->prof)
        for (auto &pr : kv.second.pending) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    for (hipEvent_t ev : c->eventPool) (void)hipEventDestroy(ev);
    for (DevBuf *b : {&c->bgBuf, &c->wrBuf, &c->textBuf, &c->qsBuf, &c->qpBuf, &c->stageBuf})
        if (b->ptr) { (void)hipFree(b->ptr); b->ptr = nullptr; b->cap = 0; }
    if (c->hMail) (void)hipHostFree(c->hMail);
    if (c->evFork) (void)hipEventDestroy(c->evFork);
    if (c->evJoin) (void)hipEventDestroy(c->evJoin);
    if (c->evFork2) (void)hipEventDestroy(c->evFork2);
    if (c->evPf) (void)hipEventDestroy(c->evPf);
    if (c->evTailJoin) (void)hipEventDestroy(c->evTailJoin);
    if (c->tail) { (void)hipStreamSynchronize(c->tail); (void)hipStreamDestroy(c->tail); }
    if (c->hDone) (void)hipHostFree(c->hDone);
    if (c->hMaskPin) (void)hipHostFree(c->hMaskPin);
    if (c->zeroStream) (void)hipStreamDestroy(c->zeroStream);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int csr_set_tuning(csr_ctx *c, int32_t block_len, int32_t warm_p, int32_t warm_x, int32_t warm_b) {
    if (!c) return fail("null context");
    if (block_len != 0) {
        if (block_len < 32 || (block_len % 32) != 0) return fail("block_len must be a positive multiple of 32");
        if (c->configured && block_len != c->B) return fail("block_len cannot change after csr_batch_configure");
        c->B = block_len;
        c->Bfixed = true;
    }
    if (warm_p >= 0 || warm_x >= 0 || warm_b >= 0) c->adaptWarm = false;   // explicit tuning pins the windows
    if (warm_p >= 0) { c->warmP = (warm_p + 15) / 16 * 16; c->pinP = true; }
    if (warm_x >= 0) { c->warmX = (warm_x + 15) / 16 * 16; c->pinX = true; }
    if (warm_b >= 0) { c->warmB = (warm_b + 15) / 16 * 16; c->pinB = true; }
    return 0;
}

static csr_ctx *default_ctx();
extern "C" int csr_set_validation(csr_ctx *c, int32_t x_tol_ulps) {
    if (!c) c = default_ctx();      // NULL addresses the default context of the reference-shaped entry points
    if (!c) return -1;
    if (x_tol_ulps < 0 || x_tol_ulps > 64) return fail("x_tol_ulps must be in [0, 64]");
    c->xTolUlps = x_tol_ulps;
    mode_warm_defaults(c);
    return 0;
}

extern "C" int csr_synchronize(csr_ctx *c) {
    if (!c) return fail("null context");
    CHECK(ctx_select(c));
    if (c->configured) CHECK(settle(c));
    HIPOK(hipStreamSynchronize(c->stream));
    HIPOK(hipDeviceSynchronize());      // side stream, other contexts of this process: nothing is in flight on the device
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// profiling helpers: HIP events on the library's stream around every kernel launch
// ---------------------------------------------------------------------------------------------------------------
static hipEvent_t get_event(csr_ctx *c) {
    if (!c->eventPool.empty()) {
        hipEvent_t e = c->eventPool.back();
        c->eventPool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
struct Scope {
    csr_ctx *c;
    ProfEntry *pe = nullptr;
    hipEvent_t a{}, b{};
    hipStream_t st;
    Scope(csr_ctx *c_, const char *name, hipStream_t st_ = nullptr) : c(c_), st(st_ ? st_ : c_->stream) {
        if (c->profiling) {
            pe = &c->prof[name];
            a = get_event(c);
            b = get_event(c);
            (void)hipEventRecord(a, st);
        }
    }
    ~Scope() {
        if (pe) {
            (void)hipEventRecord(b, st);
            pe->pending.emplace_back(a, b);
            pe->launches++;
        }
    }
};
static void prof_collect(csr_ctx *c) {
    (void)hipStreamSynchronize(c->stream);
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->tail) (void)hipStreamSynchronize(c->tail);
    for (auto &kv : c->prof) {
        for (auto &pr : kv.second.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) kv.second.total_ms += ms;
            c->eventPool.push_back(pr.first);
            c->eventPool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}
extern "C" int csr_profile_enable(csr_ctx *c, int32_t on) {
    if (!c) c = default_ctx();      // NULL addresses the default context of the host-buffer entry points
    if (!c) return -1;
    CHECK(ctx_select(c));
    prof_collect(c);
    c->prof.clear();
    c->profiling = on != 0;
    return 0;
}
extern "C" int csr_profile_read(csr_ctx *c, csr_kernel_time *out, int32_t capacity, int32_t *n_out) {
    if (!c) c = default_ctx();
    if (!c) return -1;
    CHECK(ctx_select(c));
    prof_collect(c);
    int32_t k = 0;
    for (auto &kv : c->prof) {
        if (k < capacity && out) {
            memset(&out[k], 0, sizeof(out[k]));
            strncpy(out[k].name, kv.first.c_str(), sizeof(out[k].name) - 1);
            out[k].launches = kv.second.launches;
            out[k].total_ms = kv.second.total_ms;
        }
        ++k;
    }
    if (n_out) *n_out = k;
    return 0;
}
extern "C" int csr_get_run_stats(csr_ctx *c, csr_run_stats *out) {
    if (!c || !out) return fail("null argument");
    *out = c->rs;
    out->blocks = c->NB;
    out->block_len = c->B;
    out->warm_p = c->warmP;
    out->warm_x = c->warmX;
    out->warm_b = c->warmB;
    out->x_tol_ulps = c->xTolUlps;
    out->local_repairs = c->hMail ? (int64_t)reinterpret_cast<const unsigned int *>(c->hMail)[MAIL_LOCAL] : 0;
    out->ws_warm_f = c->wsWarmF;
    out->ws_warm_b = c->wsWarmB;
    return 0;
}

#define LAUNCH_CHECK(name)                                                              \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) return fail("launch %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

// The rest of the host library, split by concern (same translation unit, order matters: later parts use earlier ones)
#include "csr_host_batch.inl"
#include "csr_host_pipeline.inl"
#include "csr_host_single.inl"
#include "csr_host_rows.inl"
#include "csr_host_qseed.inl"
#include "csr_host_comm.inl"
