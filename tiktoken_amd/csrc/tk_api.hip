// C ABI + host orchestration of the MI355X BPE encode path (see include/tiktoken_amd.h).
// Host code here only builds tables, moves buffers and launches kernels; every byte of
// pre-tokenisation and merging is done by the kernels in tk_fused.h / tk_kernels.h.  There is no CPU path.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <functional>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tiktoken_amd.h"
#include "tk_decode.h"
#include "tk_fused.h"
#include "tk_mid_plan.h"
#include "tk_tables.h"
#include "tk_unicode_tables.inc"
#include "tk_regex_kernels.h"

static thread_local std::string g_err;
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fail(TK_RUNTIME_ERROR, std::string("HIP error: ") + hipGetErrorString(e__) + " at " #expr); \
    } while (0)

struct Buf {
    void* p = nullptr;
    size_t cap = 0;
    template <class T>
    T* as() const { return (T*)p; }
};
static int ensure(Buf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return TK_OK;
    if (b.p) HIPCHK(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    HIPCHK(hipMalloc(&b.p, want));
    b.cap = want;
    return TK_OK;
}
static void release(Buf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

struct KernelStat {
    double ms = 0;
    uint64_t launches = 0;
};

#define TK_NAUX 6  // side streams of the merge kernels
#define TK_NSET 4  // chunks in flight (work sets): the front kernel of chunk k + 1 runs while chunk k is merged and its tokens are placed

// Work buffers of ONE chunk in flight.
struct WorkSet {
    Buf text_al, tile_sum, wide_ws, scan_sums, row_base, brk, docb, cand, ss, si, starts, blockcnt, pstart, res, staging, listB, listC, counters, total, g_id, g_rk,
        g_nx, g_pv, g_lv, tile_np, tile_nt, mt_keys, mtab, movf, mcnt, wbin, deferred, big, rx_spec, rx_gst, rx_lnk, rx_exit, merge_work;
    hipStream_t sb = nullptr;        // the set's back stage in a multi-chunk batch: back stages of different chunks overlap each other too
                                     // (they are chains of short latency-bound kernels, ~2 ms however small the chunk)
    hipEvent_t ev_front = nullptr;   // the front kernel is done
    hipEvent_t ev_cnt = nullptr;     // ... and the deferred tiles: the counters are in h_counters
    hipEvent_t ev_tot = nullptr;     // the chunk's token base and total are known (tok_bases[k + 1] written)
    hipEvent_t ev_done = nullptr;    // the back stage is done: totals and counters are in h_total / h_counters + TK_CNT_N, the buffers are free
    hipEvent_t ev_fork = nullptr, ev_join[TK_NAUX] = {};  // fork / join of the merge kernels on the side streams
    uint32_t* h_counters = nullptr;  // pinned [2][TK_CNT_N]
    uint64_t* h_total = nullptr;     // pinned [2]: tokens, pieces of the chunk
    std::vector<Buf*> all() {
        return {&text_al, &tile_sum, &wide_ws, &scan_sums, &row_base, &brk, &docb, &cand, &ss, &si, &starts, &blockcnt, &pstart, &res, &staging, &listB,
                &listC, &counters, &total, &g_id, &g_rk, &g_nx, &g_pv, &g_lv, &tile_np, &tile_nt, &mt_keys, &mtab, &movf, &mcnt, &wbin, &deferred, &big, &rx_spec,
                &rx_gst, &rx_lnk, &rx_exit, &merge_work};
    }
};
// What the front stage of a chunk leaves for its back stage.
struct ChunkJob {
    const uint8_t* d_text = nullptr;
    uint64_t n = 0, n_docs = 0, base = 0, ntiles = 0;
    const uint64_t* d_doc_off = nullptr;
    uint64_t* d_tok_off = nullptr;
    bool single_piece = false, spec = false, pretok = false;
    uint32_t index = 0;  // position of the chunk in its batch
    uint32_t mt_bits = 14;
    TkMissKey* mt = nullptr;   // the in-call miss table's keys (null: no table -- every missed piece gets an overflow entry)
    uint32_t ovf_base = 0;     // slots of the table = index of the first overflow entry of the miss data
    uint32_t ovf_cap = 0;      // overflow entries there is room for
    bool optimistic = false;   // the host has not waited for the deferred tiles' counters (stage_deferred): chunk_finish looks at them
};

// the chunk's entries of distinct missed pieces, as the kernels take them
static TkMiss miss_of(WorkSet& w, const ChunkJob& job) { return TkMiss{w.mtab.as<TkMissTab>(), w.movf.as<TkMissOvf>(), job.ovf_base, w.mcnt.as<uint8_t>()}; }

struct tk_core {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux[TK_NAUX] = {};  // side streams: the merge kernels are independent of each other
    hipEvent_t ev_start = nullptr;
    hipStream_t cs_h2d = nullptr, cs_d2h = nullptr;  // copy streams of the host-buffer entry point (created on first use)
    void* stage[2] = {nullptr, nullptr};             // page-locked staging buffers
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    TkHostTables H;
    TkTables D;  // device view
    Buf t_stage1, t_stage2, t_bmp, t_byte_tab, t_short, t_mid, t_dec, t_piece, t_piece_off, t_tok_bytes, t_pair, t_pair2, t_byte_rank, t_xl, t_xfilter, t_spec_bytes, t_spec_off, t_spec_id, t_spec_head;
    uint32_t spec_max_len = 0;
    bool has_rx = false;  // the pat_str runs on the generic engine (tk_regex_kernels.h)
    bool has_rx_fb = false;  // a pat_str of the scanner families, compiled for the generic engine as well: the way out of stretches without certain starts (stage_deferred)
    TkRxCompiled rx_fb;
    uint64_t st_fallbacks = 0;  // chunks that took that way
    uint64_t st_regrown = 0;    // batches repeated with a larger miss data (encode_device_locked)
    uint64_t st_resynced = 0;   // batches repeated because a deferred tile gave up its walk while the host was not waiting for the counters (stage_deferred)
    bool defer_sync = false;    // ... from then on the host waits for them in every chunk, as it did up to round 5
    uint32_t defer_ppm = 1u << 12;  // deferred tiles per 2^20 tiles of the last chunk (the grid of the kernel that finishes them; first guess: one in 256 -- 4400 empty workgroups of that kernel were 0.06 ms of a first 1 GiB call)
    TkRxDev rx{};
    Buf t_rx_ins, t_rx_sets, t_rx_ranges, t_rx_first, t_rx_s1, t_rx_s2, t_rx_dtrans, t_rx_dascii, t_rx_ds1, t_rx_ds2;
    bool rx_staged = true;  // the speculative pass over text staged in LDS where the pattern's DFA allows it ($TIKTOKEN_AMD_RX_STAGED=0: never)
    int rx_form = TK_RX_FORM_PROGRAM;  // how the generic engine's kernels match: the pattern's DFA where it has one ($TIKTOKEN_AMD_RX_MATCHER)
    std::mutex mu;
    // workspace: per chunk in flight, and what a whole call shares
    WorkSet ws[TK_NSET];
    Buf text, doc_off, out_tokens, out_tok_off, allowed, tok_bases;  // tok_bases[k]: tokens of the chunks before chunk k (on the device)
    // Streams of the back stages of a multi-chunk batch.  HIP multiplexes its streams onto a few hardware queues (four by default), and
    // two streams that share a queue run one after the other: the back stage of chunk k, queued behind the front kernel of chunk
    // k + 1, then waits for that kernel to END instead of running beside it (seen in the kernel timeline of round 4: no overlap at all).
    // Which streams share a queue cannot be asked; it is found out once per caller's stream (pick_back_streams).
    hipStream_t back_for = nullptr;  // the front stream the choice was made for
    struct BackChoice {
        hipStream_t s[3];
        int n;
    };
    std::map<hipStream_t, BackChoice> back_known;  // ... and the choices made for other front streams before
    bool back_probed = false;
    hipStream_t back_s[3] = {};      // streams that share a queue neither with back_for nor with each other (as far as the pool has any)
    int n_back = 0;
    uint32_t* h_probe = nullptr;     // page-locked words of the probe: [0] the gate, [1 ..] one per candidate
    Buf out_tokens_alt, out_tok_off_alt;  // the other pair of result buffers of tk_encode_batch_device (tk_set_output_buffers(core, 2))
    uint32_t out_bufs = 1;
    bool ovf_full = false;  // a batch has asked for more overflow entries of the miss data than the default: room for the worst case from then on
    uint64_t chunk_bytes = 1ull << 30;  // one chunk per GiB: smaller chunks pipeline (stage_front / stage_back) but pay the merge kernels' fixed latency per chunk
    int dbg = 0;
    uint32_t n_cu = 256;             // compute units of the device
    uint32_t rx_grid_cap = 65536;    // most workgroups of tk_k_rx_speculate_staged (each walks the stretches with the stride of the grid: a chunk of more than 2 GiB, or $TIKTOKEN_AMD_RX_GRID_CAP)
    uint32_t rx_seg_shift = 0;       // 0: by chunk size (tk_regex_split.h)
    uint32_t rx_ahead = 0;           // 0: TK_RX_AHEAD / TK_RX_AHEAD_DFA by the kernels' form ($TIKTOKEN_AMD_RX_AHEAD: bytes)
    uint32_t front_wgs = TKF_OCC;    // workgroups per CU of the persistent front kernel ($TIKTOKEN_AMD_FRONT_WGS)
    uint32_t n_dec = 0;  // entries of the device decode table (0: ids too sparse for a direct table -- decode stays on the host)
    Buf d_tok, d_lens, d_bsum, d_tboff, d_bytes, d_bytes_alt, d_boff;  // decode workspace (d_bytes_alt: the other range's bytes on their way to the host)
    // Small calls (tk_k_small) do not take `mu`: the reference's normal use is several threads on one Encoding (core.py:175, a thread pool
    // over encode; lib.rs:232-238 keeps a regex per thread for it), and a small call needs nothing of the shared workspace -- a slot of its
    // own (page-locked text and result buffers the kernel reads and writes directly, merge scratch, a stream) is all.  A caller takes a
    // free slot (one atomic exchange), launches, watches the slot's completion word.
    struct SmallSlot {
        std::atomic<int> busy{0};
        std::atomic<int> state{0};  // 0 idle, 1 ready (input written, waiting for a launch), 2 launched
        uint8_t* in = nullptr;    // page-locked, device-visible: text of the call
        uint32_t* out = nullptr;  // page-locked, device-visible: its result
        void *d_in = nullptr, *d_out = nullptr;
        uint32_t seq = 0, n = 0;
        Buf ws;
        hipStream_t s = nullptr;
        bool ready = false;       // every piece above has been made (set last: a first use that failed half-way is repeated by the next caller)
    };
    SmallSlot small[TK_SMALL_SLOTS];
    // Callers of small calls that arrive together go out in ONE launch (flat combining): whoever gets this mutex -- try_lock: nobody waits for
    // it -- launches every slot that is ready, his own included or not (someone else may have taken it along already); the others watch
    // their completion words.  The streams of the launches take turns so that consecutive batches overlap on the device.
    std::mutex small_launch_mu;
    std::atomic<int> small_active{0};  // callers inside encode_small / encode_mid
    std::atomic<int> mid_skip{0};     // calls that skip encode_mid (set when an attempt found the text unfit for the small kernel)
    std::atomic<int> mid_fail_run{0};  // such attempts in a row
    bool mid_cut = false;             // an ASCII letter followed by a space is a certain piece start of this pattern: documents of 2 .. 128 KiB are cut there
    uint64_t st_mid_calls = 0;
    hipStream_t small_s[4] = {};
    uint32_t small_turn = 0;
    uint64_t st_small_launches = 0, st_small_calls = 0;
    std::vector<uint8_t> sorted_blob;  // token_byte_values(), packed (built on first use)
    std::vector<uint64_t> sorted_off;
    // instrumentation
    bool profiling = false;
    std::map<std::string, KernelStat> stats;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
    uint64_t st_bytes = 0, st_pieces = 0, st_tokens = 0, st_docs = 0, st_medium = 0, st_long = 0;
    uint64_t st_chunks = 0;
    double host_us[6] = {0, 0, 0, 0, 0, 0};  // host time of the last call: front stages, back stages (of which: waiting for the front kernel), finish waits, total
};

// ------------------------------------------------------------------------------------------
// Pinned host memory, pooled.  Results of the host-buffer entry points are returned in page-locked buffers (the D2H copy runs at PCIe
// speed and the caller reads them in place: no second copy); pinning a gigabyte costs far more than filling it, so released buffers are
// kept for the next call.  tk_free() recognises them.
// ------------------------------------------------------------------------------------------
struct PinnedBuf {
    void* p;
    size_t cap;
    bool used;
};
static std::mutex g_pin_mu;
static std::vector<PinnedBuf> g_pin;
static void* pinned_get(size_t bytes) {
    if (bytes < 64) bytes = 64;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    PinnedBuf* best = nullptr;
    for (auto& b : g_pin)
        if (!b.used && b.cap >= bytes && (!best || b.cap < best->cap)) best = &b;
    if (best && best->cap <= 4 * bytes + (64u << 20)) {
        best->used = true;
        return best->p;
    }
    void* p = nullptr;
    const size_t cap = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&p, cap, hipHostMallocPortable) != hipSuccess) return nullptr;
    size_t free_bytes = 0;  // keep the pool bounded: drop idle buffers beyond 8 GiB
    for (size_t i = 0; i < g_pin.size();) {
        if (!g_pin[i].used && (free_bytes += g_pin[i].cap) > (8ull << 30)) {
            (void)hipHostFree(g_pin[i].p);
            g_pin.erase(g_pin.begin() + i);
        } else {
            ++i;
        }
    }
    g_pin.push_back(PinnedBuf{p, cap, true});
    return p;
}
static bool pinned_release(void* p) {
    if (!p) return false;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (auto& b : g_pin)
        if (b.p == p) {
            b.used = false;
            return true;
        }
    return false;
}

template <class F>
static int timed(tk_core* c, hipStream_t s, const char* name, F&& f) {
    if (!c->profiling) {
        f();
        HIPCHK(hipGetLastError());
        return TK_OK;
    }
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    if (e == hipSuccess) e = hipEventRecord(a, s);
    if (e == hipSuccess) {
        f();
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(b, s);
    if (e != hipSuccess) {  // (no event pair is left behind by a failed launch)
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
        return fail(TK_RUNTIME_ERROR, std::string("HIP error: ") + hipGetErrorString(e) + " in " + name);
    }
    c->pending.push_back({name, {a, b}});
    return TK_OK;
}
static int drain_events(tk_core* c) {
    for (auto& pe : c->pending) {
        float ms = 0;
        HIPCHK(hipEventSynchronize(pe.second.second));
        HIPCHK(hipEventElapsedTime(&ms, pe.second.first, pe.second.second));
        KernelStat& ks = c->stats[pe.first];
        ks.ms += ms;
        ks.launches += 1;
        (void)hipEventDestroy(pe.second.first);
        (void)hipEventDestroy(pe.second.second);
    }
    c->pending.clear();
    return TK_OK;
}
#define TRY(x)                      \
    do {                            \
        int rc__ = (x);             \
        if (rc__ != TK_OK) return rc__; \
    } while (0)

static int upload(Buf& b, const void* src, size_t bytes) {
    TRY(ensure(b, bytes ? bytes : 16));
    if (bytes) HIPCHK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return TK_OK;
}

extern "C" void tk_free(void* p);
extern "C" const char* tk_last_error(void) { return g_err.c_str(); }

extern "C" int tk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int tk_create(const uint8_t* ranks_blob, const uint64_t* ranks_off, const uint32_t* ranks_ids, uint64_t n_ranks,
                         const uint8_t* spec_blob, const uint64_t* spec_off, const uint32_t* spec_ids, uint64_t n_spec,
                         const char* pat_str, int device, tk_core** out) {
    if (!out) return fail(TK_VALUE_ERROR, "out is null");
    *out = nullptr;
    {
        TkPat pp;
        TkRxCompiled prx;
        const std::string perr = tk_compile_pattern(pat_str, &pp, nullptr, &prx);
        if (!perr.empty()) return fail(TK_UNSUPPORTED, perr);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(TK_RUNTIME_ERROR, "no HIP device available: tiktoken_amd has no CPU path");
    if (device < 0 || device >= ndev) return fail(TK_VALUE_ERROR, "device ordinal out of range");
    tk_core* c = new tk_core();
    c->device = device;
    std::string err = tk_build_tables(ranks_blob, ranks_off, ranks_ids, n_ranks, spec_blob, spec_off, spec_ids, n_spec, pat_str, &c->H);
    if (!err.empty()) {
        delete c;
        return fail(TK_VALUE_ERROR, err);
    }
    auto bail = [&](int rc) {
        tk_destroy(c);
        return rc;
    };
    if (hipSetDevice(device) != hipSuccess) return bail(fail(TK_RUNTIME_ERROR, "hipSetDevice failed"));
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(TK_RUNTIME_ERROR, "hipStreamCreate failed"));
    // All streams at the default priority.  (Measured, profiles/r03_stream_priority.txt: the mere existence of high-priority streams
    // in the process slows the front kernel from 5.76 to 6.28 ms per GiB even while nothing runs on them.  $TIKTOKEN_AMD_PRIO=1 gives the
    // back stages' streams the highest priority for experiments.)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (!getenv("TIKTOKEN_AMD_PRIO")) prio_hi = 0;
    for (int i = 0; i < TK_NAUX; ++i)
        if (hipStreamCreateWithPriority(&c->aux[i], hipStreamNonBlocking, prio_hi) != hipSuccess) return bail(fail(TK_RUNTIME_ERROR, "hipStreamCreate failed"));
    if (hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming) != hipSuccess) return bail(fail(TK_RUNTIME_ERROR, "hipEventCreate failed"));
    for (WorkSet& w : c->ws) {
        if (hipStreamCreateWithPriority(&w.sb, hipStreamNonBlocking, prio_hi) != hipSuccess) return bail(fail(TK_RUNTIME_ERROR, "hipStreamCreate failed"));
        for (hipEvent_t* e : {&w.ev_front, &w.ev_cnt, &w.ev_tot, &w.ev_done, &w.ev_fork})
            if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return bail(fail(TK_RUNTIME_ERROR, "hipEventCreate failed"));
        for (int i = 0; i < TK_NAUX; ++i)
            if (hipEventCreateWithFlags(&w.ev_join[i], hipEventDisableTiming) != hipSuccess) return bail(fail(TK_RUNTIME_ERROR, "hipEventCreate failed"));
        if (hipHostMalloc((void**)&w.h_counters, 2 * TK_CNT_N * 4 + 64, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void**)&w.h_total, 64, hipHostMallocDefault) != hipSuccess)
            return bail(fail(TK_RUNTIME_ERROR, "hipHostMalloc failed"));
    }
    const TkHostTables& H = c->H;
    int rc;
    auto upload_rx = [&](const TkRxCompiled& X) -> int {  // the program of the generic engine
        TRY(upload(c->t_rx_ins, X.ins.data(), X.ins.size() * sizeof(TkRxIns)));
        TRY(upload(c->t_rx_sets, X.sets.data(), X.sets.size() * sizeof(TkRxSet)));
        TRY(upload(c->t_rx_ranges, X.ranges.data(), X.ranges.size() * 4));
        TRY(upload(c->t_rx_first, X.first.data(), X.first.size() * 4));
        TRY(upload(c->t_rx_s1, tk_rx_props_stage1(), 0x1100));
        TRY(upload(c->t_rx_s2, tk_rx_props_stage2(), (size_t)tk_rx_props_blocks() * 256));
        c->rx = TkRxDev{c->t_rx_ins.as<TkRxIns>(), c->t_rx_sets.as<TkRxSet>(), c->t_rx_ranges.as<uint32_t>(), c->t_rx_s1.as<uint8_t>(),
                        c->t_rx_s2.as<uint8_t>(), (uint32_t)X.ins.size(), (uint32_t)X.sets.size(), (uint32_t)X.ranges.size() / 2,
                        c->t_rx_first.as<uint32_t>(), (uint32_t)X.first.size() / 8, nullptr, nullptr, nullptr, nullptr, 0u, 0u, 0u};
        c->rx_form = TK_RX_FORM_PROGRAM;
        // the pattern's DFA (tk_regex_dfa.inc), where it has one: $TIKTOKEN_AMD_RX_MATCHER = program | dfa | flat (the default) chooses the
        // kernels' form -- "dfa" keeps the piece-by-piece speculative lane, "program" interprets the backtracking program as before
        const char* want = getenv("TIKTOKEN_AMD_RX_MATCHER");
        if (const char* st = getenv("TIKTOKEN_AMD_RX_STAGED")) c->rx_staged = strcmp(st, "0") != 0;
        if (X.has_dfa() && !(want && !strcmp(want, "program"))) {
            std::vector<uint16_t> tr(X.dfa_trans);
            tr.resize((tr.size() + 1) & ~(size_t)1, 0);  // (whole 32-bit words: the kernels copy it to LDS word by word)
            TRY(upload(c->t_rx_dtrans, tr.data(), tr.size() * 2));
            TRY(upload(c->t_rx_dascii, X.dfa_ascii.data(), 384));
            TRY(upload(c->t_rx_ds1, X.dfa_s1.data(), X.dfa_s1.size() * 2));
            TRY(upload(c->t_rx_ds2, X.dfa_s2.data(), X.dfa_s2.size()));
            c->rx.dfa_trans = c->t_rx_dtrans.as<uint16_t>();
            c->rx.dfa_ascii = c->t_rx_dascii.as<uint8_t>();
            c->rx.dfa_s1 = c->t_rx_ds1.as<uint16_t>();
            c->rx.dfa_s2 = c->t_rx_ds2.as<uint8_t>();
            c->rx.dfa_ncls = X.dfa_ncls;
            c->rx.dfa_nstates = X.dfa_nstates;
            c->rx.dfa_flags = X.dfa_flags;
            c->rx_form = (want && !strcmp(want, "dfa")) ? TK_RX_FORM_DFA : TK_RX_FORM_DFA_FLAT;
            if (X.dfa_flags & 1u) c->rx_form += TK_RX_FORM_DFA_PREV - TK_RX_FORM_DFA;  // (a pattern that looks behind: the instantiations that read the char in front of a match)
        }
        return TK_OK;
    };
    // the class of every code point below U+10000 in one table (TkTables::uc_bmp), from the two stages
    auto upload_bmp = [&](const uint8_t* s1, const uint8_t* s2) -> int {
        std::vector<uint8_t> bmp(65536);
        for (uint32_t cp = 0; cp < 65536u; ++cp) bmp[cp] = s2[(uint32_t)s1[cp >> 8] * 256u + (cp & 255u)];
        return upload(c->t_bmp, bmp.data(), bmp.size());
    };
    if (H.rx.empty()) {
        // a pattern of the scanner families: compiled for the generic engine as well, for stretches of text without certain starts
        // (stage_deferred); a family member the generic compiler cannot take keeps its scanners alone
        if (tk_rx_compile(pat_str, &c->rx_fb).empty()) {
            if ((rc = upload_rx(c->rx_fb))) return bail(rc);
            c->has_rx_fb = true;
        }
        if ((rc = upload(c->t_stage1, tk_uc_stage1, sizeof tk_uc_stage1))) return bail(rc);
        if ((rc = upload(c->t_stage2, tk_uc_stage2, sizeof tk_uc_stage2))) return bail(rc);
        if ((rc = upload_bmp(tk_uc_stage1, tk_uc_stage2))) return bail(rc);
        uint32_t bt[256 * 2];
        tk_build_byte_table(tk_uc_stage1, tk_uc_stage2, bt);
        if ((rc = upload(c->t_byte_tab, bt, sizeof bt))) return bail(rc);
    } else {
        // The generic engine splits (tk_regex_kernels.h) and hands every piece start to the front kernel as a hard start; the scanners
        // then run over a class table in which every char is a lower-case letter: a piece is a run of letters up to the next hard start.
        std::vector<uint8_t> s1(0x1100, 0), s2(256, (uint8_t)TK_C_LL);
        if ((rc = upload(c->t_stage1, s1.data(), s1.size()))) return bail(rc);
        if ((rc = upload(c->t_stage2, s2.data(), s2.size()))) return bail(rc);
        if ((rc = upload_bmp(s1.data(), s2.data()))) return bail(rc);
        uint32_t bt[256 * 2];
        tk_build_byte_table(s1.data(), s2.data(), bt);
        if ((rc = upload(c->t_byte_tab, bt, sizeof bt))) return bail(rc);
        if ((rc = upload_rx(H.rx))) return bail(rc);
        c->has_rx = true;
    }
    if ((rc = upload(c->t_short, H.short_tab.data(), H.short_tab.size() * sizeof(TkShortSlot)))) return bail(rc);
    if ((rc = upload(c->t_mid, H.mid_tab.data(), H.mid_tab.size() * sizeof(TkPieceSlot)))) return bail(rc);
    if ((rc = upload(c->t_piece, H.piece.data(), H.piece.size() * sizeof(TkPieceSlot)))) return bail(rc);
    if ((rc = upload(c->t_piece_off, H.piece_off.data(), H.piece_off.size() * 4))) return bail(rc);
    if ((rc = upload(c->t_tok_bytes, H.tok_bytes.data(), H.tok_bytes.size()))) return bail(rc);
    if ((rc = upload(c->t_pair, H.pair8.empty() ? (const void*)H.pair.data() : (const void*)H.pair8.data(),
                     H.pair8.empty() ? H.pair.size() * sizeof(TkPairSlot) : H.pair8.size() * 8))) return bail(rc);
    if ((rc = upload(c->t_pair2, H.pair2.data(), H.pair2.size() * 4))) return bail(rc);
    if ((rc = upload(c->t_byte_rank, H.byte_rank, sizeof H.byte_rank))) return bail(rc);
    if ((rc = upload(c->t_xl, H.xl.data(), H.xl.size() * sizeof(TkXlSlot)))) return bail(rc);
    if ((rc = upload(c->t_xfilter, H.xfilter.data(), H.xfilter.size() * 4))) return bail(rc);
    if ((rc = upload(c->t_spec_bytes, H.spec_bytes.data(), H.spec_bytes.size()))) return bail(rc);
    if ((rc = upload(c->t_spec_off, H.spec_off.data(), H.spec_off.size() * 4))) return bail(rc);
    if ((rc = upload(c->t_spec_id, H.spec_id.data(), H.spec_id.size() * 4))) return bail(rc);
    TkTables& D = c->D;
    D.uc_stage1 = c->t_stage1.as<uint8_t>();
    D.uc_stage2 = c->t_stage2.as<uint8_t>();
    D.uc_bmp = c->t_bmp.as<uint8_t>();
    D.byte_tab = c->t_byte_tab.as<uint32_t>();
    D.short_tab = H.short_tab.empty() ? nullptr : c->t_short.as<TkShortSlot>();
    D.short_mask = H.short_mask;
    D.short_shift = H.short_shift;
    D.mid_tab = c->t_mid.as<TkPieceSlot>();
    D.mid_mask = H.mid_mask;
    D.mid_shift = H.mid_shift;
    D.piece = c->t_piece.as<TkPieceSlot>();
    D.piece_off = c->t_piece_off.as<uint32_t>();
    D.piece_mask = H.piece_mask;
    D.max_token_len = H.max_token_len;
    D.tok_bytes = c->t_tok_bytes.as<uint8_t>();
    D.pair = H.pair8.empty() ? c->t_pair.as<TkPairSlot>() : nullptr;
    D.pair8 = H.pair8.empty() ? nullptr : c->t_pair.as<uint64_t>();
    D.pair_mask = H.pair_mask;
    D.pair2 = c->t_pair2.as<uint32_t>();
    D.byte_rank = c->t_byte_rank.as<uint32_t>();
    D.xl = c->t_xl.as<TkXlSlot>();
    D.xl_mask = H.xl_mask;
    D.xfilter = c->t_xfilter.as<uint32_t>();
    D.spec_bytes = c->t_spec_bytes.as<uint8_t>();
    D.spec_off = c->t_spec_off.as<uint32_t>();
    D.spec_id = c->t_spec_id.as<uint32_t>();
    {
        std::vector<uint32_t> head(4 * (H.spec_id.size() + 1), 0u);
        for (size_t k = 0; k < H.spec_id.size(); ++k) {
            const uint32_t o = H.spec_off[k], len = H.spec_off[k + 1] - o;
            uint64_t h8 = 0;
            for (uint32_t i = 0; i < len && i < 8u; ++i) h8 |= (uint64_t)H.spec_bytes[o + i] << (8u * i);
            head[4 * k] = (uint32_t)h8;
            head[4 * k + 1] = (uint32_t)(h8 >> 32);
            head[4 * k + 2] = len;
            head[4 * k + 3] = o;
        }
        if ((rc = upload(c->t_spec_head, head.data(), head.size() * 4))) return bail(rc);
        D.spec_head = c->t_spec_head.as<uint32_t>();
    }
    D.n_spec = (uint32_t)H.spec_id.size();
    memcpy(D.spec_first, H.spec_first, sizeof D.spec_first);
    D.spec_fb = 0;
    D.n_spec_fb = 0;
    for (uint32_t b = 0; b < 256; ++b)
        if ((H.spec_first[b >> 5] >> (b & 31)) & 1u) {
            if (D.n_spec_fb < 4) D.spec_fb |= b << (8 * D.n_spec_fb);
            D.n_spec_fb += 1;
        }
    if (D.n_spec_fb > 4) D.n_spec_fb = 0xFF;
    memset(D.spec_second, 0, sizeof D.spec_second);
    for (size_t k = 0; k + 1 < H.spec_off.size(); ++k) {
        const uint32_t o = H.spec_off[k], len = H.spec_off[k + 1] - o;
        if (len < 2) memset(D.spec_second, 0xFF, sizeof D.spec_second);
        else D.spec_second[H.spec_bytes[o + 1] >> 5] |= 1u << (H.spec_bytes[o + 1] & 31);
    }
    D.pattern = H.pattern;
    D.pat = H.pat;
    memcpy(D.cert, H.cert, sizeof D.cert);
    // a document of a few KiB is cut at "letter, then space" when that is a certain piece start of the pattern (encode_mid): both cases of letter,
    // in the family's table and in what was derived for this pattern
    c->mid_cut = tk_mid_cut_certain(H.cert) && !(c->dbg & 0x4000000);  // (debug bit 0x4000000: never)
    for (size_t k = 0; k + 1 < H.spec_off.size(); ++k) c->spec_max_len = std::max(c->spec_max_len, H.spec_off[k + 1] - H.spec_off[k]);
    {  // decode table: id -> {offset into the token / special blob, length}
        uint32_t max_id = 0;
        max_id = H.max_rank;
        for (const auto& kv : H.spec_decoder) max_id = std::max(max_id, kv.first);
        if (max_id < (1u << 26)) {
            std::vector<uint2> dec((size_t)max_id + 1, make_uint2(0, 0));
            for (const auto& kv : H.spec_decoder) dec[kv.first] = make_uint2(kv.second.first | TK_DEC_SPEC, kv.second.second);
            H.for_each_token([&](uint32_t r, uint32_t o, uint32_t l) { dec[r] = make_uint2(o, l); });  // (lib.rs:347-351: decoder first)
            if ((rc = upload(c->t_dec, dec.data(), dec.size() * sizeof(uint2)))) return bail(rc);
            c->n_dec = max_id + 1;
        }
    }
    if (const char* e = getenv("TIKTOKEN_AMD_CHUNK_BYTES")) {
        uint64_t v = strtoull(e, nullptr, 10);
        if (v >= 4096 && v <= (3ull << 30)) c->chunk_bytes = v;
    }
    if (const char* e = getenv("TIKTOKEN_AMD_DEBUG")) c->dbg = atoi(e);
    if (const char* e = getenv("TIKTOKEN_AMD_DEFER_SYNC")) c->defer_sync = atoi(e) != 0;  // (experiments: the host waits for the deferred tiles' counters in every chunk)
    if (const char* e = getenv("TIKTOKEN_AMD_RX_AHEAD")) {  // (experiments: bytes a speculative match may look beyond its segment)
        const int k = atoi(e);
        if (k >= 16 && k <= (1 << 20)) c->rx_ahead = (uint32_t)k;
    }
    if (const char* e = getenv("TIKTOKEN_AMD_RX_GRID_CAP")) {  // (tests: most workgroups of the staged speculative pass, so that a small input makes every workgroup take several stretches)
        const int k = atoi(e);
        if (k >= 1 && k <= 65536) c->rx_grid_cap = (uint32_t)k;
    }
    if (const char* e = getenv("TIKTOKEN_AMD_RX_SEG_SHIFT")) {  // (experiments: segment size of the generic engine's speculative pass, 2^k bytes)
        const int k = atoi(e);
        if (k >= 5 && k <= 14) c->rx_seg_shift = (uint32_t)k;
    }
    {
        int cu = 0;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) c->n_cu = (uint32_t)cu;
        if (const char* e = getenv("TIKTOKEN_AMD_FRONT_WGS")) {
            const int v = atoi(e);
            if (v >= 1 && v <= 8) c->front_wgs = (uint32_t)v;
        }
    }
    {
        // tk_k_front reads some of its arguments from the kernarg segment again, at offsets taken from TkFrontArgs (tk_fused.h, TKF_PARK_ARGS): one launch
        // of a kernel with the same parameter list and arguments of distinct values checks that the compiler lays the segment out that way
        TkTables Tt{};
        uint64_t v = 0x1000;
        auto nextp = [&]() { v += 0x1010; return (uintptr_t)v; };
        Tt.short_tab = (const TkShortSlot*)nextp(); Tt.short_mask = (uint32_t)nextp(); Tt.short_shift = (uint32_t)nextp();
        Tt.mid_tab = (const TkPieceSlot*)nextp(); Tt.mid_mask = (uint32_t)nextp(); Tt.mid_shift = (uint32_t)nextp();
        Tt.xl = (decltype(Tt.xl))nextp(); Tt.xl_mask = (uint32_t)nextp(); Tt.max_token_len = (uint32_t)nextp();
        Tt.tok_bytes = (const uint8_t*)nextp(); Tt.piece = (const TkPieceSlot*)nextp(); Tt.piece_off = (const uint32_t*)nextp(); Tt.piece_mask = nextp();
        Tt.n_spec = (uint32_t)nextp(); Tt.spec_bytes = (const uint8_t*)nextp(); Tt.spec_id = (const uint32_t*)nextp(); Tt.spec_off = (const uint32_t*)nextp();
        TkFrontOut fo{};
        fo.starts = (uint32_t*)nextp(); fo.tile_np = (uint32_t*)nextp(); fo.res = (uint32_t*)nextp(); fo.tile_sum = (uint8_t*)nextp();
        fo.data.tab = (TkMissTab*)nextp(); fo.data.ovf = (TkMissOvf*)nextp(); fo.data.ovf_base = (uint32_t)nextp(); fo.ovf_cap = (uint32_t)nextp();
        fo.listC = (uint32_t*)nextp(); fo.counters = (uint32_t*)nextp();
        Buf okb;
        if (ensure(okb, 64) != TK_OK) { tk_destroy(c); return TK_RUNTIME_ERROR; }
        (void)hipMemsetAsync(okb.p, 0, 4, c->stream);
        hipLaunchKernelGGL(tk_k_front_args_check, dim3(1), dim3(64), 0, c->stream, Tt, (const uint8_t*)nextp(), (uint64_t)nextp(), (uint64_t)nextp(), (const uint32_t*)nextp(),
                           (const uint32_t*)nextp(), (const uint32_t*)nextp(), (const uint32_t*)nextp(), fo, (TkMissKey*)nextp(), (uint32_t)nextp(), (uint32_t*)nextp(),
                           (const uint32_t*)nextp(), (int)nextp(), okb.as<uint32_t>());
        uint32_t ok = 0;
        const bool copied = hipMemcpyAsync(&ok, okb.p, 4, hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
        release(okb);
        if (!copied || ok != 1u) {
            tk_destroy(c);
            return fail(TK_RUNTIME_ERROR, "internal error: the front kernel's arguments do not lie in the kernarg segment as TkFrontArgs says");
        }
    }
    *out = c;
    return TK_OK;
}

extern "C" void tk_destroy(tk_core* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (Buf* b : {&c->t_rx_ins, &c->t_rx_sets, &c->t_rx_ranges, &c->t_rx_first, &c->t_rx_s1, &c->t_rx_s2, &c->t_rx_dtrans, &c->t_rx_dascii, &c->t_rx_ds1, &c->t_rx_ds2}) release(*b);
    for (Buf* b : {&c->t_stage1, &c->t_stage2, &c->t_bmp, &c->t_byte_tab, &c->t_short, &c->t_mid, &c->t_dec, &c->d_tok, &c->d_lens, &c->d_bsum, &c->d_tboff, &c->d_bytes, &c->d_bytes_alt, &c->d_boff, &c->t_piece,
                   &c->t_piece_off, &c->t_tok_bytes, &c->t_pair, &c->t_pair2, &c->t_byte_rank, &c->t_xl, &c->t_xfilter, &c->t_spec_bytes, &c->t_spec_off, &c->t_spec_id, &c->t_spec_head, &c->text, &c->doc_off,
                   &c->out_tokens, &c->out_tok_off, &c->out_tokens_alt, &c->out_tok_off_alt, &c->allowed, &c->tok_bases})
        release(*b);
    if (c->h_probe) (void)hipHostFree(c->h_probe);
    for (WorkSet& w : c->ws) {
        for (Buf* b : w.all()) release(*b);
        for (hipEvent_t e : {w.ev_front, w.ev_cnt, w.ev_tot, w.ev_done, w.ev_fork})
            if (e) (void)hipEventDestroy(e);
        for (int i = 0; i < TK_NAUX; ++i)
            if (w.ev_join[i]) (void)hipEventDestroy(w.ev_join[i]);
        if (w.sb) (void)hipStreamDestroy(w.sb);
        if (w.h_counters) (void)hipHostFree(w.h_counters);
        if (w.h_total) (void)hipHostFree(w.h_total);
    }
    for (auto& sl : c->small) {
        if (sl.in) (void)hipHostFree(sl.in);
        if (sl.out) (void)hipHostFree(sl.out);
        if (sl.ws.p) (void)hipFree(sl.ws.p);
        if (sl.s) (void)hipStreamDestroy(sl.s);
    }
    for (hipStream_t& ls : c->small_s) {
        if (ls) (void)hipStreamDestroy(ls);
        ls = nullptr;
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (int i = 0; i < TK_NAUX; ++i)
        if (c->aux[i]) (void)hipStreamDestroy(c->aux[i]);
    if (c->cs_h2d) (void)hipStreamDestroy(c->cs_h2d);
    if (c->cs_d2h) (void)hipStreamDestroy(c->cs_d2h);
    for (int i = 0; i < 2; ++i) {
        if (c->stage[i]) (void)hipHostFree(c->stage[i]);
        if (c->ev_stage[i]) (void)hipEventDestroy(c->ev_stage[i]);
    }
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    delete c;
}

// entries of the piece-id space of an n-byte chunk: TKF_CAP per tile (a tile's pieces form a run at tile * TKF_CAP)
static uint64_t tk_pid_cap(uint64_t n) { return (n / TK_TILE + 1) * TKF_CAP + 64; }

static uint32_t grid_for(uint64_t items, uint32_t per_block, uint32_t cap) {
    uint64_t g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (uint32_t)g;
}

template <int MODE, class... A>
static void launch_front(int pattern, bool spec, dim3 grid, hipStream_t s, A... a) {
    if (pattern == TK_PAT_R50K) {
        if (spec) hipLaunchKernelGGL((tk_k_front<TK_PAT_R50K, true, MODE>), grid, dim3(256), 0, s, a...);
        else hipLaunchKernelGGL((tk_k_front<TK_PAT_R50K, false, MODE>), grid, dim3(256), 0, s, a...);
    } else if (pattern == TK_PAT_CL100K) {
        if (spec) hipLaunchKernelGGL((tk_k_front<TK_PAT_CL100K, true, MODE>), grid, dim3(256), 0, s, a...);
        else hipLaunchKernelGGL((tk_k_front<TK_PAT_CL100K, false, MODE>), grid, dim3(256), 0, s, a...);
    } else if (pattern == TK_PAT_O200K) {
        if (spec) hipLaunchKernelGGL((tk_k_front<TK_PAT_O200K, true, MODE>), grid, dim3(256), 0, s, a...);
        else hipLaunchKernelGGL((tk_k_front<TK_PAT_O200K, false, MODE>), grid, dim3(256), 0, s, a...);
    } else {  // a pattern of the family that is not one of the stock three: family and parameters are run-time values
        if (spec) hipLaunchKernelGGL((tk_k_front<TK_PAT_GENERIC, true, MODE>), grid, dim3(256), 0, s, a...);
        else hipLaunchKernelGGL((tk_k_front<TK_PAT_GENERIC, false, MODE>), grid, dim3(256), 0, s, a...);
    }
}

// exclusive prefix sum of a uint32 array in place, total -> total_out[0]
static int scan_u32(tk_core* c, WorkSet& w, hipStream_t s, uint32_t* a, uint64_t n, uint64_t* total_out) {
    if (n <= 2 * (uint64_t)TK_SCAN_BLOCK) {  // (one workgroup walks 8 Ki values in ~7 us: cheaper than three launches of 6-8 us each; 17.5 Ki -- the tiles of 64 MiB -- took it 27 us)
        TRY(timed(c, s, "tk_k_scan_small", [&] { hipLaunchKernelGGL(tk_k_scan_small, dim3(1), dim3(TK_SCAN_THREADS), 0, s, a, n, total_out); }));
        return TK_OK;
    }
    const uint64_t nb = (n + TK_SCAN_BLOCK - 1) / TK_SCAN_BLOCK;
    TRY(ensure(w.scan_sums, (nb + 2) * 4));
    uint32_t* sums = w.scan_sums.as<uint32_t>();
    TRY(timed(c, s, "tk_k_scan_sums", [&] { hipLaunchKernelGGL(tk_k_scan_sums, dim3((uint32_t)nb), dim3(TK_SCAN_THREADS), 0, s, a, n, sums); }));
    TRY(timed(c, s, "tk_k_scan_small", [&] { hipLaunchKernelGGL(tk_k_scan_small, dim3(1), dim3(TK_SCAN_THREADS), 0, s, sums, nb, total_out); }));
    TRY(timed(c, s, "tk_k_scan_apply", [&] { hipLaunchKernelGGL(tk_k_scan_apply, dim3((uint32_t)nb), dim3(TK_SCAN_THREADS), 0, s, a, n, sums); }));
    return TK_OK;
}

// the generic pat_str engine gave up on a piece (tk_regex_split.h): which way, and where
static int rx_failure(const uint32_t* counters, uint64_t base) {
    const std::string at = std::to_string(base + (uint64_t)(~counters[TK_CNT_RXPOS]));
    if (counters[TK_CNT_ERR] & TK_RX_ERR_STACK)
        return fail(TK_VALUE_ERROR, "pat_str: a repeated group needs more backtracking state than the matcher keeps (piece at byte " + at +
                                        " of the batch); make the group possessive, e.g. (?:...)++");
    return fail(TK_VALUE_ERROR, "pat_str: backtrack limit exceeded at byte " + at +
                                    " of the batch (nested quantifiers; the reference's fancy-regex gives up after 1 000 000 backtracks as well)");
}

// The generic engine's split of a chunk (tk_regex_kernels.h): speculate, link, resolve, then brk |= the true piece starts.  The three
// pairs of bitmaps (rx_spec, rx_lnk, rx_gst: starts and gap chars) are zero on entry.
static int rx_split(tk_core* c, WorkSet& w, hipStream_t s, const uint8_t* d_text, uint64_t n, uint32_t* brk, const uint32_t* ss, const uint32_t* si,
                    const uint64_t* d_doc_off, uint64_t n_docs, uint64_t base) {
    const uint64_t nwords = (n + 31) / 32;
    uint32_t* counters = w.counters.as<uint32_t>();
    const uint32_t seg_shift = c->rx_seg_shift ? c->rx_seg_shift : (n < TK_RX_SEG_SMALL_BELOW ? TK_RX_SEG_SHIFT_SMALL : TK_RX_SEG_SHIFT_LARGE);
    const uint64_t nseg = (n + (1ull << seg_shift) - 1) >> seg_shift;
    TRY(ensure(w.rx_exit, 3 * (nseg + 2) * 4));  // exit of every segment's chain; where the link met it; where the link left the segment
    uint32_t *spec = w.rx_spec.as<uint32_t>(), *gst = w.rx_gst.as<uint32_t>(), *xexit = w.rx_exit.as<uint32_t>();
    uint32_t *lnk = w.rx_lnk.as<uint32_t>(), *lmerge = xexit + nseg + 2, *lexit = xexit + 2 * (nseg + 2);
    // (the kernels' form: the pattern's DFA in LDS -- its speculative pass as one loop -- or the backtracking program; tk_regex_kernels.h)
    const uint32_t lds = c->rx_form == TK_RX_FORM_PROGRAM ? 0u : tk_rx_dfa_lds_bytes(c->rx);
    const uint32_t ahead = c->rx_ahead ? c->rx_ahead : (c->rx_form == TK_RX_FORM_PROGRAM ? TK_RX_AHEAD : TK_RX_AHEAD_DFA);
    auto by_form = [&](auto&& launch) {
        if (c->rx_form == TK_RX_FORM_DFA_FLAT) launch(std::integral_constant<int, TK_RX_FORM_DFA_FLAT>{});
        else if (c->rx_form == TK_RX_FORM_DFA) launch(std::integral_constant<int, TK_RX_FORM_DFA>{});
        else if (c->rx_form == TK_RX_FORM_DFA_FLAT_PREV) launch(std::integral_constant<int, TK_RX_FORM_DFA_FLAT_PREV>{});
        else if (c->rx_form == TK_RX_FORM_DFA_PREV) launch(std::integral_constant<int, TK_RX_FORM_DFA_PREV>{});
        else launch(std::integral_constant<int, TK_RX_FORM_PROGRAM>{});
    };
    // (the pattern's DFA without look-behind over 128-byte segments: the lanes walk codes staged in LDS -- tk_k_rx_speculate_staged; $TIKTOKEN_AMD_RX_STAGED=0: the one-loop lanes over global memory)
    const bool staged = c->rx_staged && c->rx_form == TK_RX_FORM_DFA_FLAT && seg_shift == TK_RX_SEG_SHIFT_SMALL && tk_rx_staged_fits(c->rx);
    TRY(timed(c, s, "tk_k_rx_speculate", [&] {
        if (staged)
            hipLaunchKernelGGL(tk_k_rx_speculate_staged, dim3(grid_for(nseg, TK_RX_STAGE_SEGS, c->rx_grid_cap)), dim3(TK_RX_STAGE_SEGS), tk_rx_staged_lds_bytes(c->rx), s, c->rx, d_text, (uint32_t)n, brk, ss,
                               si, ahead, spec, spec + nwords + 2, xexit);
        else
            by_form([&](auto form) {
                hipLaunchKernelGGL(tk_k_rx_speculate<decltype(form)::value>, dim3(grid_for(nseg, 256, 65536)), dim3(256), lds, s, c->rx, d_text, (uint32_t)n, brk, ss, si, seg_shift,
                                   ahead, spec, spec + nwords + 2, xexit);
            });
    }));
    const bool links = !(c->dbg & 0x20000);  // (debug bit 0x20000: no link pass -- the resolving pass matches its way from one chain to the next)
    if (links) {
        TRY(timed(c, s, "tk_k_rx_link", [&] {
            by_form([&](auto form) {
                hipLaunchKernelGGL(tk_k_rx_link<decltype(form)::value>, dim3(grid_for(nseg, 256, 65536)), dim3(256), lds, s, c->rx, d_text, (uint32_t)n, brk, ss, si, seg_shift,
                                   ahead, spec, xexit, lnk, lnk + nwords + 2, lmerge, lexit);
            });
        }));
    }
    const TkRxMaps maps{spec, spec + nwords + 2, xexit, links ? lnk : (const uint32_t*)nullptr, lnk + nwords + 2, lmerge, lexit, seg_shift};
    TRY(timed(c, s, "tk_k_rx_resolve", [&] {
        by_form([&](auto form) {
            if (c->dbg & 0x40000)  // (debug bit 0x40000: one lane per document instead of one wavefront)
                hipLaunchKernelGGL(tk_k_rx_resolve<decltype(form)::value>, dim3(grid_for(n_docs, 256, 65536)), dim3(256), lds, s, c->rx, d_text, (uint32_t)n, brk, ss, si,
                                   d_doc_off, n_docs, base, maps, gst, gst + nwords + 2, counters);
            else
                hipLaunchKernelGGL(tk_k_rx_resolve_wave<decltype(form)::value>, dim3(grid_for(n_docs, 4, 65536)), dim3(256), lds, s, c->rx, d_text, (uint32_t)n, brk, ss, si,
                                   d_doc_off, n_docs, base, maps, gst, gst + nwords + 2, counters);
        });
    }));
    TRY(timed(c, s, "tk_k_rx_merge", [&] { hipLaunchKernelGGL(tk_k_rx_merge, dim3(grid_for(nwords, 256, 4096)), dim3(256), 0, s, brk, gst, nwords); }));
    return TK_OK;
}

// ------------------------------------------------------------------------------------------
// The production pipeline (kernels of tk_fused.h) on one chunk (n < 4 GiB bytes) of packed documents, everything device resident, in two
// stages so that chunks can overlap: while chunk k is merged and its tokens are placed (stage_back: latency- and memory-bound kernels,
// on their own stream), the front kernel of chunk k + 1 (bound by the vector ALU) already runs on the caller's stream.  Each chunk in
// flight has its own WorkSet.  The host never needs a chunk's token count to queue the next one: the running total stays on the device
// (c->tok_bases[k]: written by chunk k - 1's back stage as soon as it knows its token count).
//   d_text: chunk text (readable 64 bytes past n); d_doc_off: uint64 offsets of the chunk's documents (n_docs + 1 entries, absolute;
//   `base` is subtracted); single_piece: the whole buffer is one piece (encode_single_piece), no pre-tokenisation.
// ------------------------------------------------------------------------------------------
// The tiles the front kernel has deferred (they need the workgroup-wide scanner: long pieces, far-away piece starts; their number stays
// on the device), then the counters of both kernels -- pieces for the tree kernel, errors of the generic engine -- on their way to the host.
// A kernel of a few hundred workgroups that each take ~0.3 ms: it belongs to the back stage, beside the next chunk's front kernel.
static int stage_deferred(tk_core* c, WorkSet& w, ChunkJob& job, hipStream_t s) {
    const TkTables& T = c->D;
    // A stretch without certain starts ("x'llx'll...": whether 'll ends a piece depends on everything before it) makes every deferred tile
    // inside it walk from the stretch's start -- quadratic in its length, seconds for 10 MB.  A tile whose walk exceeds TKF_WALK_BUDGET
    // windows gives up instead (second list); if any did, the generic engine -- linear on exactly such text: the pieces are short, every
    // segment's guess is taken -- splits the chunk under the same pat_str, its piece starts become hard starts, and the tiles that gave up
    // run again: every piece start is certain now.  (Pieces that are already final are what they were: a hard start at the start of a
    // piece changes nothing, and the stock patterns match a piece the same way when the text ends behind it.)
    const bool can_fall_back = c->has_rx_fb && job.n >= (256u << 10) && !(c->dbg & 0x400000);  // (debug bit 0x400000: never)
    if (job.n > 0 && !job.single_piece) {
        TkFrontOut fo{w.starts.as<uint32_t>(), w.tile_np.as<uint32_t>(), w.res.as<uint32_t>(), w.tile_sum.as<uint8_t>(), miss_of(w, job), job.ovf_cap,
                      w.listC.as<uint32_t>(), w.counters.as<uint32_t>()};
        uint32_t *ss = job.spec ? w.ss.as<uint32_t>() : nullptr, *si = job.spec ? w.si.as<uint32_t>() : nullptr, *docb = job.spec ? w.docb.as<uint32_t>() : nullptr;
        // (the grid: what is resident -- or, where the host does not wait for the counters, as many workgroups as the chunks before had deferred tiles (at
        // least 64: they take their tiles from a counter).  On ordinary text no tile is deferred since round 6, and an empty grid of 768 such workgroups
        // -- 168 registers, 37 KiB of LDS, scratch -- costs 12.6 us against the 7 of 64.)
        const uint64_t defer_guess = ((job.ntiles * (uint64_t)c->defer_ppm) >> 20) * 5 / 4 + 64;
        const bool sync_now = can_fall_back && (c->defer_sync || job.pretok);  // ($TIKTOKEN_AMD_DEFER_SYNC=1: from the start; the piece-offsets entry has no chunk_finish)
        job.optimistic = can_fall_back && !sync_now;
        uint64_t slow_wgs = 256u * TKF_SLOW_OCC;
        if (job.optimistic && defer_guess < slow_wgs) slow_wgs = defer_guess;
        const dim3 grid((uint32_t)(job.ntiles < slow_wgs ? job.ntiles : slow_wgs));
        const int pat_id = T.pat.generic() ? TK_PAT_GENERIC : T.pattern;
        TkMissKey* mt_arg = (c->dbg & 256) ? (TkMissKey*)nullptr : job.mt;
        const uint32_t* gapb = c->has_rx ? w.rx_gst.as<uint32_t>() + (job.n + 31) / 32 + 2 : (const uint32_t*)nullptr;
        int fdbg = c->dbg | (job.pretok ? 8 : 0);  // (piece starts only: every probe counts as a hit, nothing is listed for the merges)
        // The deferred tiles in two kernels: the deferred-tile instance finds a tile's piece starts (the workgroup-wide scanner: 128 registers,
        // four workgroups per CU), the one-tile-per-workgroup instance does the rest from the starts it is given (phases E and F, at eight
        // workgroups per CU).  Its grid is the list's length where the host reads the counters (inputs of 256 KiB and more); otherwise one
        // workgroup per tile of the chunk, of which all but the list's length return at once.
        TRY(timed(c, s, "tk_k_front_slow", [&] {
            launch_front<TKF_MODE_STARTS>(pat_id, job.spec, grid, s, T, job.d_text, job.n, job.base, w.brk.as<uint32_t>(), docb, ss, si, fo, mt_arg, (1u << job.mt_bits) - 1u,
                               w.deferred.as<uint32_t>(), gapb, (fdbg & ~TKF_DBG_SECOND) | (can_fall_back ? TKF_DBG_MAY_GIVE_UP : 0));
        }));
        uint64_t n_given = job.ntiles;
        // (round 6) The host does not wait for the counters here any more: the wait cost every chunk ~25 us of an idle device between the two kernels
        // (3 % of a 64 MiB batch) and kept the host from queueing the next chunk -- for a decision that ordinary text never needs.  The kernel that
        // finishes the deferred tiles walks the list with a grid sized from the chunk before (the same kind of text: a quarter more than it needed),
        // and chunk_finish looks at the counter of tiles that gave up: if there is one the batch is repeated (encode_device_locked) and the core waits
        // here from then on (c->defer_sync), as it did up to round 5.  Small inputs, a pat_str without that way out: one workgroup per tile.
        if (job.optimistic) n_given = defer_guess < job.ntiles ? defer_guess : job.ntiles;
        if (sync_now) {
            HIPCHK(hipMemcpyAsync(w.h_counters, w.counters.p, TK_CNT_N * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(hipEventRecord(w.ev_cnt, s));
            HIPCHK(hipEventSynchronize(w.ev_cnt));
            if (can_fall_back && w.h_counters[TK_CNT_DEFER2]) {
                const uint64_t nwords = (job.n + 31) / 32;
                for (Buf* b : {&w.rx_spec, &w.rx_gst, &w.rx_lnk}) {
                    TRY(ensure(*b, 2 * (nwords + 2) * 4));
                    HIPCHK(hipMemsetAsync(b->p, 0, 2 * (nwords + 2) * 4, s));
                }
                TRY(rx_split(c, w, s, job.d_text, job.n, w.brk.as<uint32_t>(), ss, si, job.d_doc_off, job.n_docs, job.base));
                TRY(timed(c, s, "tk_k_front_slow", [&] {
                    launch_front<TKF_MODE_STARTS>(pat_id, job.spec, grid, s, T, job.d_text, job.n, job.base, w.brk.as<uint32_t>(), docb, ss, si, fo, mt_arg, (1u << job.mt_bits) - 1u,
                                       w.deferred.as<uint32_t>() + job.ntiles + 2, gapb, (fdbg & ~TKF_DBG_MAY_GIVE_UP) | TKF_DBG_SECOND);
                }));
                c->st_fallbacks += 1;
            }
            n_given = w.h_counters[TK_CNT_DEFER];
        }
        if (n_given && !(c->dbg & 0x1F000)) {  // (debug bits 0x1000 .. 0x10000: the kernels stop after a phase, there are no starts to go on from)
            TRY(timed(c, s, "tk_k_front_given", [&] {
                launch_front<TKF_MODE_GIVEN>(pat_id, job.spec, dim3((uint32_t)n_given), s, T, job.d_text, job.n, job.base, w.brk.as<uint32_t>(), docb, ss, si, fo, mt_arg,
                                    (1u << job.mt_bits) - 1u, w.deferred.as<uint32_t>(), gapb, fdbg & ~(TKF_DBG_SECOND | TKF_DBG_MAY_GIVE_UP));
            }));
        }
    }
    HIPCHK(hipMemcpyAsync(w.h_counters, w.counters.p, TK_CNT_N * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(w.ev_cnt, s));
    return TK_OK;
}
#define TK_RESYNC (-1001)  // (internal) a deferred tile gave up while the host was not waiting: the batch is repeated, encode_device_locked

static int stage_front(tk_core* c, WorkSet& w, ChunkJob& job, hipStream_t s, const uint8_t* d_text, uint64_t n, const uint64_t* d_doc_off, uint64_t n_docs,
                       uint64_t base, bool use_special, bool single_piece, uint64_t* d_tok_off, bool pretok_only, bool no_lookup,
                       uint64_t* pretok_count_out) {
    const TkTables& T = c->D;
    const uint64_t nwords = (n + 31) / 32;
    const uint64_t nblk = (nwords + 255) / 256;
    const uint64_t ntiles = (n && !single_piece) ? (n + TK_TILE - 1) / TK_TILE : 1;  // (single piece: one run of one piece)
    job = ChunkJob();
    job.d_text = d_text;
    job.n = n;
    job.n_docs = n_docs;
    job.base = base;
    job.ntiles = ntiles;
    job.d_doc_off = d_doc_off;
    job.d_tok_off = d_tok_off;
    job.single_piece = single_piece;
    TRY(ensure(w.brk, (nwords + 2) * 4));
    TRY(ensure(w.starts, (nwords + 2) * 4));
    TRY(ensure(w.blockcnt, (nblk + 2) * 4));
    TRY(ensure(w.counters, TK_CNT_N * 4));
    TRY(ensure(w.total, 32));
    TRY(ensure(w.tile_np, (ntiles + 2) * 4));
    TRY(ensure(w.tile_nt, (ntiles + 2) * 4));
    TRY(ensure(w.wbin, (TK_NBIN * TKD_WAVES + 2) * 4));
    const uint64_t pid_cap = tk_pid_cap(n);
    TRY(ensure(w.res, pid_cap * 4));
    TRY(ensure(w.staging, (n + 64) * 4));
    // The entries of the distinct missed pieces (tk_fused.h, TkMiss): one per slot of the in-call de-duplication table -- which pays for
    // its reset (of the 16-byte keys: the 64-byte entries are written by whoever claims a slot, never cleared) only on real batches -- and
    // the overflow entries behind them.  The table takes what repeats; the overflow entries are sized for a sixteenth of the worst case
    // (every piece a distinct two-byte piece that is not a token) unless the chunk is small or a batch has already asked for more
    // (c->ovf_full: encode_device_locked repeats such a batch once, with room for the worst case).
    job.pretok = pretok_only;
    // (small chunks too: without the table every missed piece has an overflow entry, whose tokens tk_k_place copies from the staging area two
    // pieces at a time -- 92 us for a 4 KiB call, against 4 us for clearing 16 Ki keys)
    if (n >= 512 && !single_piece && !pretok_only) {
#ifndef TK_MT_DIV
#define TK_MT_DIV 128
#endif
        while (job.mt_bits < TK_MT_BITS && (1ull << job.mt_bits) < n / TK_MT_DIV) ++job.mt_bits;  // 4 Mi slots from 512 MiB up
        job.ovf_base = 1u << job.mt_bits;
    }
    {
        const uint64_t worst = n / 2 + 64;
        job.ovf_cap = (uint32_t)((n <= (1u << 20) || c->ovf_full) ? worst : std::min<uint64_t>(worst, std::max<uint64_t>(n >> 4, 1u << 16)));
        if (pretok_only) job.ovf_cap = 64;
    }
    const uint64_t n_entries = (uint64_t)job.ovf_base + job.ovf_cap;
    TRY(ensure(w.mtab, (uint64_t)job.ovf_base * sizeof(TkMissTab)));
    TRY(ensure(w.mcnt, (uint64_t)job.ovf_base + 16));
    TRY(ensure(w.movf, ((uint64_t)job.ovf_cap + 1) * sizeof(TkMissOvf)));
    TRY(ensure(w.listB, (n_entries + 64) * 4));
    TRY(ensure(w.listC, (n / 1025 + 64) * 20));
    TRY(ensure(w.big, (1 + 3 * TK_BIGCOPY_CAP) * 4));
    TkClearArgs clr;
    clr.n = 0;
    auto clear = [&](Buf& b, uint64_t bytes, uint32_t v) {  // (every Buf has at least 256 bytes of slack behind what was asked for)
        clr.p[clr.n] = b.as<uint4>();
        clr.n16[clr.n] = (bytes + 15) / 16;
        clr.v[clr.n] = v;
        ++clr.n;
    };
    clear(w.brk, (nwords + 2) * 4, 0u);
    clear(w.counters, TK_CNT_N * 4, 0u);
    clear(w.big, 4, 0u);
    TRY(ensure(w.merge_work, 16 * TKM_WORK_STRIDE * 4));
    clear(w.merge_work, 16 * TKM_WORK_STRIDE * 4, 0u);
    clear(w.total, 32, 0u);
    uint32_t *brk = w.brk.as<uint32_t>(), *starts = w.starts.as<uint32_t>();
    uint32_t *ss = nullptr, *si = nullptr, *docb = nullptr;
    uint32_t* counters = w.counters.as<uint32_t>();
    TRY(ensure(w.tile_sum, ntiles + 16));
    // (round 6) the first document that starts in every tile (tk_k_mark_docs), for tk_k_place, which writes the documents' token offsets as it
    // passes their pieces; one piece without pre-tokenisation and empty chunks keep tk_k_docoff
    const bool docs_in_place = n > 0 && !single_piece && d_tok_off != nullptr;
    TRY(ensure(w.row_base, (ntiles + 4) * 4));
    if (docs_in_place) clear(w.row_base, (ntiles + 2) * 4, 0xFFFFFFFFu);
    clear(w.tile_sum, ntiles + 16, 0xFFFFFFFFu);
    if (job.ovf_base) {
        TRY(ensure(w.mt_keys, sizeof(TkMissKey) * job.ovf_base));
        clear(w.mt_keys, sizeof(TkMissKey) * job.ovf_base, 0xFFFFFFFFu);
        job.mt = w.mt_keys.as<TkMissKey>();
    }
    TkFrontOut fo{starts, w.tile_np.as<uint32_t>(), w.res.as<uint32_t>(), w.tile_sum.as<uint8_t>(), miss_of(w, job), job.ovf_cap, w.listC.as<uint32_t>(), counters};
    if (n > 0 && !single_piece) {
        if (use_special) {
            TRY(ensure(w.docb, (nwords + 4) * 4));  // (tk_bits64 reads two words past the one a position lies in)
            TRY(ensure(w.cand, (nwords + 4) * 4));
            TRY(ensure(w.ss, (nwords + 2) * 4));
            TRY(ensure(w.si, (nwords + 2) * 4));
            for (Buf* b : {&w.docb, &w.cand}) clear(*b, (nwords + 4) * 4, 0u);
            for (Buf* b : {&w.ss, &w.si}) clear(*b, (nwords + 2) * 4, 0u);
            docb = w.docb.as<uint32_t>();
            ss = w.ss.as<uint32_t>();
            si = w.si.as<uint32_t>();
        }
        if (c->has_rx) {
            // (two bitmaps each: the starts, and behind them the gap chars among the starts)
            TRY(ensure(w.rx_spec, 2 * (nwords + 2) * 4));
            TRY(ensure(w.rx_gst, 2 * (nwords + 2) * 4));
            TRY(ensure(w.rx_lnk, 2 * (nwords + 2) * 4));
            clear(w.rx_spec, 2 * (nwords + 2) * 4, 0u);
            clear(w.rx_gst, 2 * (nwords + 2) * 4, 0u);
            clear(w.rx_lnk, 2 * (nwords + 2) * 4, 0u);
        }
        hipLaunchKernelGGL(tk_k_chunk_clear, dim3(grid_for(n / 64 + 1, 256, 2048)), dim3(256), 0, s, clr);
        clr.n = 0;
        TRY(timed(c, s, "tk_k_mark_docs", [&] {
            hipLaunchKernelGGL(tk_k_mark_docs, dim3(grid_for(n_docs, 256, 4096)), dim3(256), 0, s, d_doc_off, n_docs, base, n, brk, docb,
                               docs_in_place ? w.row_base.as<uint32_t>() : (uint32_t*)nullptr, ntiles);
        }));
        if (use_special) {
            const uint8_t* allowed = c->allowed.as<uint8_t>();
            uint32_t* cand = w.cand.as<uint32_t>();
            TRY(timed(c, s, "tk_k_spec_cand", [&] {
                hipLaunchKernelGGL(tk_k_spec_cand, dim3(grid_for(n / 16 + 1, 256, 65536)), dim3(256), 0, s, T, d_text, n, allowed, docb, cand);
            }));
            TRY(timed(c, s, "tk_k_spec_resolve", [&] {
                hipLaunchKernelGGL(tk_k_spec_resolve, dim3(grid_for(nwords, 256, 65536)), dim3(256), 0, s, T, d_text, n, allowed, docb, cand,
                                   c->spec_max_len, ss, si, brk);
            }));
        }
        if (c->has_rx) TRY(rx_split(c, w, s, d_text, n, brk, ss, si, d_doc_off, n_docs, base));  // the generic engine finds the piece starts; they join the hard starts in `brk`
        TRY(ensure(w.deferred, 2 * (ntiles + 2) * 4));  // (behind the list of deferred tiles: those that gave up their walk, stage_deferred)
        uint32_t* deferred = w.deferred.as<uint32_t>();
        TkMissKey* mt_arg = (c->dbg & 256) ? (TkMissKey*)nullptr : job.mt;
        TRY(timed(c, s, "tk_k_front", [&] {
            const dim3 grid((uint32_t)ntiles);
            launch_front<TKF_MODE_TILE>(T.pat.generic() ? TK_PAT_GENERIC : T.pattern, ss != nullptr, grid, s, T, d_text, n, base, brk, docb, ss, si, fo,
                                mt_arg, (1u << job.mt_bits) - 1u, deferred, c->has_rx ? w.rx_gst.as<uint32_t>() + nwords + 2 : (const uint32_t*)nullptr,
                                c->dbg | (pretok_only ? 8 : 0) | ((c->has_rx && !(c->dbg & 4)) ? TKF_DBG_HARD_ONLY : 0));  // (debug bit 4: the scanners run even so)
        }));
    } else if (n > 0) {
        hipLaunchKernelGGL(tk_k_chunk_clear, dim3(1), dim3(256), 0, s, clr);
        clr.n = 0;
        TRY(timed(c, s, "tk_k_single_front", [&] { hipLaunchKernelGGL(tk_k_single_front, dim3(1), dim3(64), 0, s, T, d_text, (uint32_t)n, fo, no_lookup ? 1 : 0); }));
    }
    if (clr.n) hipLaunchKernelGGL(tk_k_chunk_clear, dim3(1), dim3(256), 0, s, clr);  // (an empty chunk)
    job.spec = ss != nullptr;
    HIPCHK(hipEventRecord(w.ev_front, s));
    if (pretok_only) {  // debugging / test entry: piece offsets only
        TRY(stage_deferred(c, w, job, s));
        uint64_t P = 0;
        TRY(ensure(w.pstart, 16));
        if (n > 0) {
            TRY(timed(c, s, "tk_k_count", [&] {
                hipLaunchKernelGGL(tk_k_count, dim3((uint32_t)nblk), dim3(256), 0, s, starts, nwords, w.blockcnt.as<uint32_t>());
            }));
            TRY(timed(c, s, "tk_k_scan_small", [&] {
                hipLaunchKernelGGL(tk_k_scan_small, dim3(1), dim3(TK_SCAN_THREADS), 0, s, w.blockcnt.as<uint32_t>(), nblk, w.total.as<uint64_t>());
            }));
            HIPCHK(hipMemcpyAsync(&P, w.total.p, 8, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (w.h_counters[TK_CNT_ERR] & (TK_RX_ERR_GAP | TK_RX_ERR_STACK | TK_RX_ERR_LIMIT)) return rx_failure(w.h_counters, base);
            TRY(ensure(w.pstart, (P + 2) * 4));
            TRY(timed(c, s, "tk_k_emit", [&] {
                hipLaunchKernelGGL(tk_k_emit, dim3((uint32_t)nblk), dim3(256), 0, s, starts, nwords, w.blockcnt.as<uint32_t>(),
                                   w.pstart.as<uint32_t>(), P, n, c->has_rx ? w.rx_gst.as<uint32_t>() + nwords + 2 : (const uint32_t*)nullptr);
            }));
        } else {
            HIPCHK(hipMemsetAsync(w.pstart.p, 0, 4, s));
        }
        *pretok_count_out = P;
    }
    return TK_OK;
}

// Second stage of a chunk, on stream s (the front stage's stream for a single chunk, the set's own otherwise; the caller has made it wait
// for w.ev_front).  d_out: the batch's token buffer; the chunk's tokens go behind tok_bases[job.index] of them, which the previous chunk's
// back stage writes (prev_tot: the event to wait for, null for the first chunk or on a single stream).
static int stage_back(tk_core* c, WorkSet& w, ChunkJob& job, hipStream_t s, uint32_t* d_out, hipEvent_t prev_tot) {
    const TkTables& T = c->D;
    const uint64_t n = job.n, ntiles = job.ntiles;
    const uint8_t* d_text = job.d_text;
    uint32_t* counters = w.counters.as<uint32_t>();
    uint32_t *res = w.res.as<uint32_t>(), *stg = w.staging.as<uint32_t>();
    const TkMiss data = miss_of(w, job);
    uint32_t *tile_np = w.tile_np.as<uint32_t>(), *tile_nt = w.tile_nt.as<uint32_t>();
    const unsigned long long* tok_base = c->tok_bases.as<unsigned long long>() + job.index;
    uint64_t nC = 0;
    TRY(stage_deferred(c, w, job, s));
    // (perf experiments, tools/gpu_phases.sh: debug bits 0x1000 .. 0x10000 stop the front kernel after one of its phases -- its outputs are
    // incomplete, so nothing behind it runs: the call returns zero tokens and offsets that mean nothing)
    const bool front_only = (c->dbg & 0x1F000) != 0;
    if (front_only) {
        HIPCHK(hipMemsetAsync(w.total.p, 0, 32, s));
        if (prev_tot) HIPCHK(hipStreamWaitEvent(s, prev_tot, 0));
        hipLaunchKernelGGL(tk_k_advance, dim3(1), dim3(64), 0, s, c->tok_bases.as<unsigned long long>(), job.index, w.total.as<uint64_t>());
        HIPCHK(hipEventRecord(w.ev_tot, s));
        HIPCHK(hipMemcpyAsync(w.h_total, w.total.p, 16, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(w.h_counters + TK_CNT_N, counters, TK_CNT_N * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipEventRecord(w.ev_done, s));
        return TK_OK;
    }
    if (n > 0) {
        uint32_t* wbin = w.wbin.as<uint32_t>();
        uint32_t* listB = w.listB.as<uint32_t>();
        // the two list passes walk the miss data (table slots + overflow entries): as many wavefronts as it has rows of 64 entries for
        // (small calls are latency-bound: a workgroup or two)
        const uint64_t n_entries = (uint64_t)job.ovf_base + job.ovf_cap;
        const uint32_t dd_blocks = grid_for(n_entries, 4 * 256, TKD_WAVES / 4);
        TRY(timed(c, s, "tk_k_bincount", [&] {
            hipLaunchKernelGGL(tk_k_bincount, dim3(dd_blocks), dim3(256), 0, s, T, d_text, data, job.mt, job.ovf_cap, counters, wbin, (c->dbg & 512) ? 0 : 1);
        }));
        TRY(scan_u32(c, w, s, wbin, (uint64_t)TK_NBIN * dd_blocks * 4 + 1, w.total.as<uint64_t>()));
        TRY(timed(c, s, "tk_k_binfill", [&] {
            hipLaunchKernelGGL(tk_k_binfill, dim3(dd_blocks), dim3(256), 0, s, T, d_text, data, job.mt, job.ovf_cap, wbin, listB, counters);
        }));
        if (T.pair8 && !(c->dbg & 0x800000)) {
            // every bin in one launch (tk_k_merge_all); debug bit 0x800000: the kernel-per-bin form below
            uint64_t most_units = 0;
            for (int b = 0; b < TK_NBIN; ++b)
                if (n >= tk_bin_lo(b)) most_units += std::min<uint64_t>(n / tk_bin_lo(b), n_entries) / (64u >> (b == 0 ? 0 : (b <= 2 ? 1 : (b <= 4 ? 2 : b - 2)))) + 1;
            uint32_t wgs = (uint32_t)std::min<uint64_t>((most_units + TKM_WAVES - 1) / TKM_WAVES, (uint64_t)c->n_cu * TKM_WGS_PER_CU);
            wgs = std::max(16u / TKM_WAVES, (wgs + 16u / TKM_WAVES - 1u) / (16u / TKM_WAVES) * (16u / TKM_WAVES));  // (wavefronts: a multiple of 16, tk_k_merge_all's work counters rely on it)
            TRY(timed(c, s, "tk_k_merge_all", [&] {
                hipLaunchKernelGGL(tk_k_merge_all, dim3(wgs), dim3(64 * TKM_WAVES), TKM_LDS_BYTES, s, T, d_text, listB, counters, data, stg, w.merge_work.as<uint32_t>(), c->dbg);
            }));
        } else
        {
            // The bins are independent: spread them over the side streams, longest-tailed kernels first.  List starts and lengths are
            // read on the device, so nothing waits for the host here; grids are sized by the most a bin can hold.
            static const char* const names[TK_NBIN] = {"tk_k_merge_llane_16", "tk_k_merge_llane_24", "tk_k_merge_llane_32", "tk_k_merge_llane_48", "tk_k_merge_llane_64",
                                                       "tk_k_merge_group_8", "tk_k_merge_group_16", "tk_k_merge_group_32", "tk_k_merge_group_64"};
            uint32_t small_counts[TK_CNT_N];
            const bool small = n <= (4u << 20);  // small calls: a round trip is cheaper than launching kernels over empty lists
            if (small) {
                HIPCHK(hipMemcpyAsync(small_counts, counters, sizeof small_counts, hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
            }
            HIPCHK(hipEventRecord(w.ev_fork, s));
            for (int i = 0; i < TK_NAUX; ++i) HIPCHK(hipStreamWaitEvent(c->aux[i], w.ev_fork, 0));
            // (longest first; the four lane-group kernels have a stream each -- their run time is the longest piece's chain of
            // merges --, the five lane-per-piece kernels share two)
            static const int order[TK_NBIN] = {6, 8, 7, 5, 2, 4, 1, 3, 0};
            static const int stream_of[TK_NBIN] = {5, 5, 4, 4, 5, 3, 0, 2, 1};
            for (int oi = 0; oi < TK_NBIN; ++oi) {
                const int b = order[oi];
                if (n < tk_bin_lo(b) || (small && !small_counts[TK_CNT_BIN0 + b])) continue;
                const uint64_t most = std::min<uint64_t>(n / tk_bin_lo(b), n_entries);
                hipStream_t sa = c->aux[stream_of[b]];
                TRY(timed(c, sa, names[b], [&] {
                    switch (b) {
                        case 0: hipLaunchKernelGGL((tk_k_merge_llane<16, 256>), dim3(grid_for(most, 256, 8192)), dim3(256), 0, sa, T, d_text, listB, counters, b, data, stg); break;
                        case 1: hipLaunchKernelGGL((tk_k_merge_llane<24, 256>), dim3(grid_for(most, 256, 8192)), dim3(256), 0, sa, T, d_text, listB, counters, b, data, stg); break;
                        case 2: hipLaunchKernelGGL((tk_k_merge_llane<32, 256>), dim3(grid_for(most, 256, 8192)), dim3(256), 0, sa, T, d_text, listB, counters, b, data, stg); break;
                        case 3: hipLaunchKernelGGL((tk_k_merge_llane<48, 128>), dim3(grid_for(most, 128, 8192)), dim3(128), 0, sa, T, d_text, listB, counters, b, data, stg); break;
                        case 4: hipLaunchKernelGGL((tk_k_merge_llane<64, 128>), dim3(grid_for(most, 128, 8192)), dim3(128), 0, sa, T, d_text, listB, counters, b, data, stg); break;
                        case 5: hipLaunchKernelGGL((tk_k_merge_group<8>), dim3(grid_for(most, 32, 8192)), dim3(256), 0, sa, T, d_text, listB, counters, b, data, stg); break;
                        case 6: hipLaunchKernelGGL((tk_k_merge_group<16>), dim3(grid_for(most, 16, 8192)), dim3(256), 0, sa, T, d_text, listB, counters, b, data, stg); break;
                        case 7: hipLaunchKernelGGL((tk_k_merge_group<32>), dim3(grid_for(most, 8, 8192)), dim3(256), 0, sa, T, d_text, listB, counters, b, data, stg); break;
                        default: hipLaunchKernelGGL((tk_k_merge_group<64>), dim3(grid_for(most, 4, 8192)), dim3(256), 0, sa, T, d_text, listB, counters, b, data, stg); break;
                    }
                }));
            }
            for (int i = 0; i < TK_NAUX; ++i) {
                HIPCHK(hipEventRecord(w.ev_join[i], c->aux[i]));
                HIPCHK(hipStreamWaitEvent(s, w.ev_join[i], 0));
            }
        }
        // pieces longer than TK_GLANE_MAX were listed by the front kernel: their scratch is sized from its counters,
        // which were copied back while the kernels above were being queued
        {
            const double t0 = now_us();
            HIPCHK(hipEventSynchronize(w.ev_cnt));
            c->host_us[2] += now_us() - t0;
        }
        const uint32_t* hc = w.h_counters;
        nC = hc[TK_CNT_C];
        if (nC) {
            const uint64_t lb = (uint64_t)hc[TK_CNT_CBYTES] + 4 * nC, lvls = hc[TK_CNT_CLEVELS];
            TRY(ensure(w.g_id, (lb + 64) * 4));
            TRY(ensure(w.g_rk, (lb + 64) * 4));
            TRY(ensure(w.g_nx, (lb + 64) * 4));
            TRY(ensure(w.g_pv, (lb + 64) * 4));
            TRY(ensure(w.g_lv, (lvls + 64) * 8));
            const bool rounds = !(c->dbg & 1024);  // (debug bit 1024: one merge at a time for every long piece)
            if (rounds) {
                TRY(timed(c, s, "tk_k_merge_rounds", [&] {
                    hipLaunchKernelGGL(tk_k_merge_rounds, dim3(grid_for(nC, 1, 1024)), dim3(TKB_THREADS), 0, s, T, d_text, w.listC.as<uint32_t>(),
                                       (uint32_t)nC, w.g_id.as<uint32_t>(), w.g_rk.as<uint32_t>(), w.g_nx.as<uint32_t>(), w.g_pv.as<uint32_t>(),
                                       data, stg);
                }));
            }
            if (rounds && n >= TK_WIDE_MIN) {  // (pieces of TK_WIDE_MIN bytes and more, if there are any: the whole grid on each)
                TRY(ensure(w.wide_ws, sizeof(TkWideWs)));
                HIPCHK(hipMemsetAsync(w.wide_ws.p, 0, sizeof(TkWideWs), s));
                TRY(timed(c, s, "tk_k_merge_rounds_wide", [&] {
                    hipLaunchKernelGGL(tk_k_merge_rounds_wide, dim3(TK_WIDE_BLOCKS), dim3(TKB_THREADS), 0, s, T, d_text, w.listC.as<uint32_t>(),
                                       (uint32_t)nC, w.g_id.as<uint32_t>(), w.g_rk.as<uint32_t>(), w.g_nx.as<uint32_t>(), w.g_pv.as<uint32_t>(),
                                       data, stg, w.wide_ws.as<TkWideWs>());
                }));
            }
            TRY(timed(c, s, "tk_k_merge_long", [&] {
                hipLaunchKernelGGL(tk_k_merge_long, dim3(grid_for(nC, 4, 8192)), dim3(256), 0, s, T, d_text, w.listC.as<uint32_t>(), (uint32_t)nC,
                                   w.g_id.as<uint32_t>(), w.g_rk.as<uint32_t>(), w.g_nx.as<uint32_t>(), w.g_pv.as<uint32_t>(),
                                   w.g_lv.as<uint64_t>(), data, stg, rounds ? 1 : 0);
            }));
        }
    }
    const bool small_rows = n < (8u << 20);  // (the one-row instances: a third of the code to fetch for a kernel that runs over a few tiles)
    if (n > 0) {
        // token count per tile (a missed piece's count from its entry), then the tiles' places (tk_fused.h: back end)
        TRY(timed(c, s, "tk_k_count_tiles", [&] {
            if (small_rows) hipLaunchKernelGGL(tk_k_count_tiles<1>, dim3(grid_for(ntiles, 4, 2048)), dim3(256), 0, s, ntiles, tile_np, res, data, tile_nt,
                               w.total.as<unsigned long long>());
            else hipLaunchKernelGGL(tk_k_count_tiles<TKP_ROWS_COUNT>, dim3(grid_for(ntiles, 4, 2048)), dim3(256), 0, s, ntiles, tile_np, res, data, tile_nt,
                               w.total.as<unsigned long long>());
        }));
        TRY(scan_u32(c, w, s, tile_nt, ntiles, w.total.as<uint64_t>()));
    } else {
        HIPCHK(hipMemsetAsync(w.total.p, 0, 32, s));  // (the scan of the bin counts is not run for an empty chunk; be explicit)
    }
    // the chunk's token count is known: the next chunk's base.  Its tokens go behind those of the chunk before it (whose back stage may be
    // running beside this one).
    if (prev_tot) HIPCHK(hipStreamWaitEvent(s, prev_tot, 0));
    hipLaunchKernelGGL(tk_k_advance, dim3(1), dim3(64), 0, s, c->tok_bases.as<unsigned long long>(), job.index, w.total.as<uint64_t>());
    HIPCHK(hipEventRecord(w.ev_tot, s));
    if (n > 0) {
        TRY(timed(c, s, "tk_k_place", [&] {
            // (one instance for every size: three rows per step for inputs of a few tiles measured slower -- C1 0.082 ms against 0.060 --, profiles/r05_place_experiments.txt)
            const bool docs_in_place = !job.single_piece && job.d_tok_off != nullptr;
            const TkPlaceDocs docs{job.d_doc_off, job.base, n, job.n_docs, w.row_base.as<uint32_t>(), w.starts.as<uint32_t>(), w.total.as<uint64_t>(), docs_in_place ? job.d_tok_off : (uint64_t*)nullptr};
            // (six workgroups per CU: 80 registers, 26 688 B of LDS; a grid of four times what is resident -- 0.84 ms with 4096 workgroups, 0.81 with 1536 or 3072,
            // 0.80 with 6144 or 12 288, round 6)
            hipLaunchKernelGGL(tk_k_place<TKP_ROWS_PLACE>, dim3(grid_for(ntiles, 4, 6144)), dim3(256), 0, s, ntiles, tile_np, tile_nt, res, data, stg, d_out, tok_base, w.big.as<uint32_t>(), docs);
        }));
    }
    if (n > TK_BIGCOPY)  // (a token run of TK_BIGCOPY tokens needs at least as many bytes)
        hipLaunchKernelGGL(tk_k_bigcopy, dim3(1024), dim3(256), 0, s, w.big.as<uint32_t>(), stg, d_out, tok_base);
    // (the document offsets need the tile counts only, but beside tk_k_place on a second stream the two take as long as one after the
    // other: both are bound by the rate of random accesses -- measured in round 4)
    if (job.d_tok_off && (n == 0 || job.single_piece)) {  // (else tk_k_place has written them)
        TRY(timed(c, s, "tk_k_docoff", [&] {
            hipLaunchKernelGGL(tk_k_docoff, dim3(grid_for(job.n_docs + 1, 16, 4096)), dim3(256), 0, s, job.n_docs, job.d_doc_off, job.base, n, w.starts.as<uint32_t>(), tile_nt, res, data,
                               (n > 0 && !job.single_piece) ? w.row_base.as<uint32_t>() : (const uint32_t*)nullptr, w.total.as<uint64_t>(), tok_base, job.d_tok_off);
        }));
    }
    HIPCHK(hipMemcpyAsync(w.h_total, w.total.p, 16, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(w.h_counters + TK_CNT_N, counters, TK_CNT_N * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(w.ev_done, s));
    c->st_long += nC;
    return TK_OK;
}

#define TK_GROW (-1000)  // (internal) the batch has to be repeated with a larger miss data: encode_device_locked
// the chunk of `w` is complete: its totals, statistics and error flags (waits for its back stage)
static int chunk_finish(tk_core* c, WorkSet& w, const ChunkJob& job, uint64_t* n_tokens_out) {
    HIPCHK(hipEventSynchronize(w.ev_done));
    const uint32_t* hb = w.h_counters + TK_CNT_N;
    uint64_t nB = 0;
    for (int b = 0; b < TK_NBIN; ++b) {
        nB += hb[TK_CNT_BIN0 + b];
        if ((c->dbg & 64) && hb[TK_CNT_BIN0 + b]) fprintf(stderr, "bin %d (%u..%u bytes): %u pieces\n", b, tk_bin_lo(b), tk_bin_hi(b), hb[TK_CNT_BIN0 + b]);
    }
    if (hb[TK_CNT_ERR] & (TK_RX_ERR_GAP | TK_RX_ERR_STACK | TK_RX_ERR_LIMIT)) return rx_failure(hb, job.base);
    if (hb[TK_CNT_ERR]) return fail(TK_RUNTIME_ERROR, "internal error in the front kernel (scanner list overflow, code " + std::to_string(hb[TK_CNT_ERR]) + ")");
    if (job.optimistic) {  // (stage_deferred did not wait for these)
        if (hb[TK_CNT_DEFER2]) {
            c->defer_sync = true;
            return TK_RESYNC;
        }
        // (at once upwards, to a quarter per chunk downwards: chunks of two kinds of text in turn keep the larger grid)
        const uint32_t ppm = (uint32_t)std::min<uint64_t>(((uint64_t)hb[TK_CNT_DEFER] << 20) / (job.ntiles ? job.ntiles : 1), 1u << 20);
        c->defer_ppm = std::max(ppm, c->defer_ppm / 4);
    }
    if (hb[TK_CNT_OVF] > job.ovf_cap && !job.pretok) {  // more distinct missed pieces than the miss data has room for: the batch is repeated with room for the worst case
        c->ovf_full = true;
        return TK_GROW;
    }
    c->st_bytes += job.n;
    c->st_pieces += w.h_total[1];
    c->st_tokens += w.h_total[0];
    c->st_medium += nB;
    c->st_chunks += 1;
    *n_tokens_out = w.h_total[0];
    return TK_OK;
}

// one chunk, both stages on one stream, waited for: the single-chunk entries (pre-tokenise only, single piece)
static int run_chunk(tk_core* c, hipStream_t s, const uint8_t* d_text, uint64_t n, const uint64_t* d_doc_off, uint64_t n_docs,
                     uint64_t base, bool use_special, bool single_piece, uint32_t* d_out, uint64_t* d_tok_off, uint64_t* n_tokens_out,
                     bool pretok_only = false, bool no_lookup = false) {
    WorkSet& w = c->ws[0];
    ChunkJob job;
    TRY(ensure(c->tok_bases, 64));
    HIPCHK(hipMemsetAsync(c->tok_bases.p, 0, 16, s));
    TRY(stage_front(c, w, job, s, d_text, n, d_doc_off, n_docs, base, use_special, single_piece, d_tok_off, pretok_only, no_lookup, n_tokens_out));
    if (pretok_only) return TK_OK;
    TRY(stage_back(c, w, job, s, d_out, nullptr));
    return chunk_finish(c, w, job, n_tokens_out);
}

static int prepare_allowed(tk_core* c, hipStream_t s, const uint32_t* allowed_ids, uint64_t n_allowed, bool* any) {
    const TkHostTables& H = c->H;
    std::vector<uint8_t> a(H.spec_id.size() + 16, 0);
    *any = false;
    for (size_t k = 0; k < H.spec_id.size(); ++k)
        for (uint64_t j = 0; j < n_allowed; ++j)
            if (H.spec_id[k] == allowed_ids[j]) {
                a[k] = 1;
                *any = true;
            }
    TRY(ensure(c->allowed, a.size()));
    HIPCHK(hipMemcpyAsync(c->allowed.p, a.data(), a.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));  // `a` goes out of scope
    return TK_OK;
}

// ---- which of the library's streams run BESIDE a given stream (see tk_core::back_s) ----
// A gate kernel spins on `on` until the host opens the gate (or 4 ms have passed: 100 MHz counter); a one-thread kernel on every
// candidate stream sets a word of its own.  Candidates whose word arrives while the gate is closed do not share `on`'s hardware queue.
// (the probe's words live in mapped host memory: system-scope atomics, so that neither side looks at a cached copy)
__global__ void tk_k_gate(volatile uint32_t* gate, uint64_t max_ticks) {
    const uint64_t t0 = wall_clock64();
    while (!__hip_atomic_load((const uint32_t*)gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(16);
}
__global__ void tk_k_touch(uint32_t* p) { __hip_atomic_store(p, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

static int streams_beside(tk_core* c, hipStream_t on, const std::vector<hipStream_t>& cand, std::vector<hipStream_t>* beside) {
    beside->clear();
    if (cand.empty()) return TK_OK;
    if (!c->h_probe) HIPCHK(hipHostMalloc((void**)&c->h_probe, 64 * 4, hipHostMallocCoherent | hipHostMallocMapped));
    volatile uint32_t* h = c->h_probe;
    uint32_t* d = nullptr;
    HIPCHK(hipHostGetDevicePointer((void**)&d, c->h_probe, 0));
    for (int i = 0; i < 64; ++i) h[i] = 0;
    hipLaunchKernelGGL(tk_k_gate, dim3(1), dim3(1), 0, on, (volatile uint32_t*)d, (uint64_t)400000);
    for (size_t i = 0; i < cand.size(); ++i) hipLaunchKernelGGL(tk_k_touch, dim3(1), dim3(1), 0, cand[i], d + 1 + i);
    // wait until the words have stopped arriving (a launch reaches the device within tens of microseconds)
    const double t0 = now_us();
    size_t seen = 0;
    double t_last = t0;
    for (;;) {
        size_t n = 0;
        for (size_t i = 0; i < cand.size(); ++i) n += h[1 + i] ? 1 : 0;
        const double t = now_us();
        if (n != seen) {
            seen = n;
            t_last = t;
        }
        if (n == cand.size() || (t - t0 > 200.0 && t - t_last > 100.0) || t - t0 > 2000.0) break;
    }
    std::vector<bool> ok(cand.size());
    for (size_t i = 0; i < cand.size(); ++i) ok[i] = h[1 + i] != 0;
    h[0] = 1u;  // open the gate
    HIPCHK(hipStreamSynchronize(on));
    for (hipStream_t q : cand) HIPCHK(hipStreamSynchronize(q));
    for (size_t i = 0; i < cand.size(); ++i)
        if (ok[i]) beside->push_back(cand[i]);
    return TK_OK;
}

// back-stage streams for front stream s: first those that run beside s, among them those that run beside each other
static int pick_back_streams(tk_core* c, hipStream_t s) {
    if (c->back_probed && c->back_for == s) return TK_OK;
    {  // (a caller that alternates between streams: what was found for a stream is kept)
        auto it = c->back_known.find(s);
        if (it != c->back_known.end()) {
            c->n_back = it->second.n;
            for (int i = 0; i < 3; ++i) c->back_s[i] = it->second.s[i];
            c->back_for = s;
            c->back_probed = true;
            return TK_OK;
        }
    }
    std::vector<hipStream_t> pool, f;
    for (WorkSet& w : c->ws) pool.push_back(w.sb);
    for (int i = 0; i < TK_NAUX; ++i) pool.push_back(c->aux[i]);
    c->n_back = 0;
    hipStream_t on = s;
    for (int r = 0; r < 3 && !pool.empty(); ++r) {
        TRY(streams_beside(c, on, pool, &f));
        if (f.empty()) break;
        c->back_s[c->n_back++] = on = f[0];
        pool.assign(f.begin() + 1, f.end());
    }
    if (c->n_back == 0) c->back_s[c->n_back++] = c->ws[0].sb;  // (nothing runs beside s: the stages take turns, as before)
    c->back_for = s;
    c->back_probed = true;
    tk_core::BackChoice bc{};
    bc.n = c->n_back;
    for (int i = 0; i < 3; ++i) bc.s[i] = c->back_s[i];
    if (c->back_known.size() < 64) c->back_known[s] = bc;
    return TK_OK;
}

// Hooks of the host-buffer entry point: the text of a chunk must have arrived before its kernels start, and its tokens can start
// their way back while the next chunk is being encoded.
struct ChunkHooks {
    std::function<int(uint64_t /*byte_end*/)> before;                                   // text [0, byte_end) has to be on the device
    std::function<int(uint64_t /*tok_begin*/, uint64_t /*n_tok*/, bool /*last*/)> after;  // a chunk's tokens are final (stream idle)
};

// Device-resident batch: cut into chunks at document boundaries and pipelined -- the front stage of chunk k + 1 is queued (on the
// caller's stream) before the back stage of chunk k (on the back-stage stream), so the two overlap on the device; TK_NSET work sets.
static int encode_device_pass(tk_core* c, hipStream_t s, const uint8_t* d_utf8, uint64_t n_bytes, const uint64_t* d_doc_off,
                              const uint64_t* h_doc_off, uint64_t n_docs, bool use_special, uint64_t* n_tokens_out,
                              uint64_t chunk_bytes, const ChunkHooks* hooks) {
    if (!chunk_bytes) chunk_bytes = c->chunk_bytes;
    c->st_bytes = c->st_pieces = c->st_tokens = c->st_medium = c->st_long = c->st_chunks = 0;
    c->st_docs = n_docs;
    TRY(ensure(c->out_tokens, (n_bytes + 64) * 4));  // (a token is at least one byte of text)
    TRY(ensure(c->out_tok_off, (n_docs + 2) * 8));
    uint32_t* d_out = c->out_tokens.as<uint32_t>();
    uint64_t* d_tok_off = c->out_tok_off.as<uint64_t>();
    // chunks: document ranges [d0, d1) of at most chunk_bytes (a longer document is a chunk of its own); cuts prefer documents that
    // start at a 16-byte aligned address (the kernels read the text with aligned 4- and 16-byte loads, some of them through the scalar
    // unit, which ignores the low address bits: any other chunk is first copied to an aligned buffer, ~0.1 ms per 128 MiB)
    struct Cut {
        uint64_t d0, d1, b, nn;
    };
    std::vector<Cut> cuts;
    if (n_bytes <= chunk_bytes || !h_doc_off) {
        if (n_bytes > chunk_bytes && n_bytes >= (3ull << 30)) return fail(TK_VALUE_ERROR, "h_doc_off is required when n_bytes exceeds the chunk size");
        cuts.push_back(Cut{0, n_docs, 0, n_bytes});
    } else {
        // (chunks of equal size, cut at the document boundaries next to j * n / N: a last chunk of a few KiB would still pay the back
        // stage's fixed latencies, about a millisecond; a chunk may exceed chunk_bytes by less than a document)
        const uint64_t n_cuts = (n_bytes + chunk_bytes - 1) / chunk_bytes;
        uint64_t d0 = 0;
        while (d0 < n_docs) {
            uint64_t d1 = d0 + 1;
            const uint64_t j = cuts.size() + 1;
            const uint64_t until = j >= n_cuts ? n_bytes : (uint64_t)((unsigned __int128)n_bytes * j / n_cuts);
            while (d1 < n_docs && h_doc_off[d1] < until && h_doc_off[d1 + 1] - h_doc_off[d0] <= chunk_bytes + (chunk_bytes >> 3)) ++d1;
            if (d1 < n_docs)  // an aligned cut a little earlier saves the next chunk its copy
                for (uint64_t q = d1; q > d0 + 1 && d1 - q < 256; --q)
                    if ((((uintptr_t)d_utf8 + h_doc_off[q]) & 15u) == 0) {
                        d1 = q;
                        break;
                    }
            const uint64_t b = h_doc_off[d0], nn = h_doc_off[d1] - b;
            if (nn >= (4ull << 30) - 65536) return fail(TK_VALUE_ERROR, "a single document of 4 GiB or more is not supported");
            cuts.push_back(Cut{d0, d1, b, nn});
            d0 = d1;
        }
        if (cuts.empty()) cuts.push_back(Cut{0, 0, 0, 0});
    }
    const size_t N = cuts.size();
    if (N > 1) TRY(pick_back_streams(c, s));
    TRY(ensure(c->tok_bases, (N + 2) * 8));
    HIPCHK(hipMemsetAsync(c->tok_bases.p, 0, 8, s));  // (before the first front stage on the same stream, which every back stage waits for)
    ChunkJob jobs[TK_NSET];
    auto front = [&](size_t k) -> int {
        WorkSet& w = c->ws[k % TK_NSET];
        const Cut& q = cuts[k];
        if (k >= TK_NSET) HIPCHK(hipStreamWaitEvent(s, w.ev_done, 0));  // the set's previous chunk has left its buffers
        if (hooks && hooks->before) TRY(hooks->before(q.b + q.nn));
        const uint8_t* tx = d_utf8 + q.b;
        if (((uintptr_t)tx & 15u) != 0 && q.nn) {
            TRY(ensure(w.text_al, q.nn + 256));
            HIPCHK(hipMemcpyAsync(w.text_al.p, tx, q.nn, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemsetAsync((uint8_t*)w.text_al.p + q.nn, 0, 128, s));
            tx = w.text_al.as<uint8_t>();
        }
        TRY(stage_front(c, w, jobs[k % TK_NSET], s, tx, q.nn, d_doc_off + q.d0, q.d1 - q.d0, q.b, use_special, false, d_tok_off + q.d0, false, false, nullptr));
        jobs[k % TK_NSET].index = (uint32_t)k;
        return TK_OK;
    };
    uint64_t total = 0;
    auto finish = [&](size_t k) -> int {
        uint64_t t = 0;
        TRY(chunk_finish(c, c->ws[k % TK_NSET], jobs[k % TK_NSET], &t));
        if (hooks && hooks->after) TRY(hooks->after(total, t, k + 1 == N));
        total += t;
        return TK_OK;
    };
    for (double& x : c->host_us) x = 0;
    const double t_call = now_us();
    TRY(front(0));
    for (size_t k = 0; k < N; ++k) {
        if (k + 1 < N) {
            double t0 = now_us();
            if (k + 1 >= TK_NSET) TRY(finish(k + 1 - TK_NSET));  // (its totals are read before the set is handed to chunk k + 1)
            c->host_us[3] += now_us() - t0;
            t0 = now_us();
            TRY(front(k + 1));
            c->host_us[0] += now_us() - t0;
        }
        const double tb0 = now_us();
        // a single chunk keeps both stages on the caller's stream; otherwise every set's back stage has a stream of its own
        WorkSet& w = c->ws[k % TK_NSET];
        hipStream_t sb = N > 1 ? c->back_s[k % (size_t)c->n_back] : s;
        if (N > 1) HIPCHK(hipStreamWaitEvent(sb, w.ev_front, 0));
        TRY(stage_back(c, w, jobs[k % TK_NSET], sb, d_out, (N > 1 && k > 0) ? c->ws[(k - 1) % TK_NSET].ev_tot : (hipEvent_t) nullptr));
        c->host_us[1] += now_us() - tb0;
    }
    const double t_tail = now_us();
    for (size_t k = N > TK_NSET ? N - TK_NSET : 0; k < N; ++k) TRY(finish(k));
    HIPCHK(hipStreamSynchronize(s));
    c->host_us[4] = now_us() - t_tail;
    c->host_us[5] = now_us() - t_call;
    TRY(drain_events(c));
    *n_tokens_out = total;
    return TK_OK;
}

// The miss data's overflow entries are sized for ordinary text (stage_front).  A batch with more distinct missed pieces than that -- the
// counter on the device says so when a chunk is finished -- is run once more from its first chunk, with room for the worst case from
// then on (c->ovf_full).  Everything a pass writes is written again by the next one, and the hooks are idempotent (text already sent
// is not sent again; token ranges are copied again).
static int encode_device_locked(tk_core* c, hipStream_t s, const uint8_t* d_utf8, uint64_t n_bytes, const uint64_t* d_doc_off,
                                const uint64_t* h_doc_off, uint64_t n_docs, bool use_special, uint64_t* n_tokens_out,
                                uint64_t chunk_bytes = 0, const ChunkHooks* hooks = nullptr) {
    // (round 6) ... and so is a batch in which a deferred tile gave up its walk while the host was not waiting for the counters (TK_RESYNC,
    // stage_deferred): once more, waiting -- at most one repeat of either kind.
    int rc = encode_device_pass(c, s, d_utf8, n_bytes, d_doc_off, h_doc_off, n_docs, use_special, n_tokens_out, chunk_bytes, hooks);
    for (int again = 0; (rc == TK_GROW || rc == TK_RESYNC) && again < 2; ++again) {
        HIPCHK(hipDeviceSynchronize());  // (chunks of the abandoned pass may still be in flight on the sets' streams)
        (void)drain_events(c);
        if (rc == TK_GROW) c->st_regrown += 1;
        else c->st_resynced += 1;
        rc = encode_device_pass(c, s, d_utf8, n_bytes, d_doc_off, h_doc_off, n_docs, use_special, n_tokens_out, chunk_bytes, hooks);
    }
    if (rc == TK_GROW) return fail(TK_RUNTIME_ERROR, "internal error: the miss data overflowed at its largest size");
    if (rc == TK_RESYNC) return fail(TK_RUNTIME_ERROR, "internal error: a deferred tile gave up although the host was waiting");
    return rc;
}

extern "C" int tk_encode_batch_device(tk_core* c, const void* d_utf8, uint64_t n_bytes, const void* d_doc_off,
                                      const uint64_t* h_doc_off, uint64_t n_docs, int use_special, const uint32_t* allowed_ids,
                                      uint64_t n_allowed, void* stream, const uint32_t** d_tokens_out, uint64_t* n_tokens_out,
                                      const uint64_t** d_tok_off_out) {
    if (!c) return fail(TK_VALUE_ERROR, "core is null");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    bool any = false;
    if (use_special) TRY(prepare_allowed(c, s, allowed_ids, n_allowed, &any));
    uint64_t total = 0;
    if (c->out_bufs == 2) {  // (the previous call's result stays where it is: a consumer on another stream may still be reading it)
        std::swap(c->out_tokens, c->out_tokens_alt);
        std::swap(c->out_tok_off, c->out_tok_off_alt);
    }
    TRY(encode_device_locked(c, s, (const uint8_t*)d_utf8, n_bytes, (const uint64_t*)d_doc_off, h_doc_off, n_docs, use_special && any, &total));
    if (d_tokens_out) *d_tokens_out = c->out_tokens.as<uint32_t>();
    if (d_tok_off_out) *d_tok_off_out = c->out_tok_off.as<uint64_t>();
    if (n_tokens_out) *n_tokens_out = total;
    return TK_OK;
}

// Host-buffer batches: the text goes to the device through two page-locked staging buffers (filled by a few host threads, sent by DMA on
// a copy stream) while earlier chunks are being encoded, and every chunk's tokens start their way back to a page-locked result buffer on a
// second copy stream while the next chunk is being encoded.  PCIe is the ceiling of this path (about 50 GB/s per direction).
#define TK_STAGE_BYTES (64ull << 20)
#define TK_HOST_CHUNK (128ull << 20)

static uint64_t host_chunk_bytes(bool direct) {  // ($TIKTOKEN_AMD_HOST_CHUNK_MIB: experiments)
    if (const char* e = getenv("TIKTOKEN_AMD_HOST_CHUNK_MIB")) {
        const long v = atol(e);
        if (v >= 8 && v <= 2048) return (uint64_t)v << 20;
    }
    return direct ? (32ull << 20) : TK_HOST_CHUNK;
}
// host threads of the staging copies (text into / ids out of the page-locked buffers): the copy, not the link, bounds the host-buffer
// entries (round 4: 8 threads, 25-34 GB/s of text; a GPU box gives the container 16 cores); $TIKTOKEN_AMD_COPY_THREADS overrides
static unsigned copy_threads(unsigned hw) {
    if (const char* e = getenv("TIKTOKEN_AMD_COPY_THREADS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 64) return (unsigned)v;
    }
    return hw == 0 ? 4u : (hw > 16u ? 16u : hw);
}
static void parallel_memcpy(void* dst, const void* src, size_t n, unsigned nth) {
    if (n < (8u << 20) || nth <= 1) {
        memcpy(dst, src, n);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = ((n + nth - 1) / nth + 4095) & ~(size_t)4095;
    for (unsigned t = 0; t < nth; ++t) {
        const size_t a = (size_t)t * per;
        if (a >= n) break;
        const size_t len = a + per < n ? per : n - a;
        th.emplace_back([=]() { memcpy((uint8_t*)dst + a, (const uint8_t*)src + a, len); });
    }
    for (auto& t : th) t.join();
}

static int encode_batch_impl(tk_core* c, const uint8_t* utf8, const uint64_t* doc_off, uint64_t n_docs, int use_special, const uint32_t* allowed_ids,
                             uint64_t n_allowed, uint32_t** tokens_out, uint64_t* n_tokens_out, uint64_t* tok_off_out, bool device_result, bool no_small);
// ---- the slots of the small-call path (tk_core::SmallSlot) ----
static int small_slot_init(tk_core* c, tk_core::SmallSlot* sl) {
    if (sl->ready) return TK_OK;  // (a first use that failed: what it did make is kept, the rest is made now)
    if (!sl->in) {
        HIPCHK(hipHostMalloc((void**)&sl->in, TK_SMALL_MAX + 64, hipHostMallocCoherent | hipHostMallocMapped));
        memset(sl->in, 0, TK_SMALL_MAX + 64);
    }
    if (!sl->out) {
        HIPCHK(hipHostMalloc((void**)&sl->out, (TK_SMALL_HDR + TK_SMALL_MAX + 16) * 4, hipHostMallocCoherent | hipHostMallocMapped));
        memset(sl->out, 0, (TK_SMALL_HDR + TK_SMALL_MAX + 16) * 4);
    }
    TRY(ensure(sl->ws, 256 * TK_SMALL_PIECE * 4));
    HIPCHK(hipHostGetDevicePointer(&sl->d_in, sl->in, 0));
    HIPCHK(hipHostGetDevicePointer(&sl->d_out, sl->out, 0));
    sl->ready = true;
    return TK_OK;
}
// text -> the slot, marked ready: whoever launches next takes it along
static void small_slot_submit(tk_core::SmallSlot* sl, const uint8_t* utf8, uint32_t n, bool no_long = false) {
    memcpy(sl->in, utf8, n);
    memset(sl->in + n, 0, 8);
    sl->seq = ++sl->seq ? sl->seq : ++sl->seq;  // (never 0: the buffer starts zeroed)
    sl->n = n | (no_long ? TK_SMALL_NO_LONG : 0u);
    sl->state.store(1, std::memory_order_release);
}
// Hands a slot back on any exit path.  A slot that was submitted and never launched goes from "ready" to idle by compare-and-swap: a launcher
// that is taking it along at this very moment (1 -> 2) either loses that race or is seen.  A slot in state 2 whose kernel has not written
// the call's sequence number yet is IN FLIGHT on this call's buffers (the owner left early: its time-out); the next owner's text must not
// meet that kernel, so the slot is waited for (bounded) and, if the kernel never completes, stays busy for good (quarantined: the
// other slots remain).
static void small_slot_release(tk_core::SmallSlot* sl) {
    int st = 1;
    if (!sl->state.compare_exchange_strong(st, 0, std::memory_order_acq_rel)) {
        if (st == 2 && sl->out && __atomic_load_n(&sl->out[2], __ATOMIC_ACQUIRE) != sl->seq) {
            const auto t0 = std::chrono::steady_clock::now();
            while (__atomic_load_n(&sl->out[2], __ATOMIC_ACQUIRE) != sl->seq) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return;  // (quarantined: busy stays set)
                std::this_thread::yield();
            }
        }
        sl->state.store(0, std::memory_order_release);
    }
    sl->busy.store(0, std::memory_order_release);
}
// Waits until every one of the caller's slots has completed (the kernel's last store is the slot's sequence number, system scope: watched
// instead of a stream); while one of them has not been launched, tries to be the one who launches -- EVERY ready slot of the core, the
// caller's or not (flat combining: try_lock, nobody waits for the mutex).
static int small_wait(tk_core* c, tk_core::SmallSlot* const* mine, uint32_t k) {
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    for (;;) {
        bool all = true, unlaunched = false;
        for (uint32_t i = 0; i < k; ++i) {
            if (__atomic_load_n(&mine[i]->out[2], __ATOMIC_ACQUIRE) != mine[i]->seq) all = false;
            if (mine[i]->state.load(std::memory_order_acquire) == 1) unlaunched = true;
        }
        if (all) break;
        if (unlaunched && c->small_launch_mu.try_lock()) {
            std::lock_guard<std::mutex> lk(c->small_launch_mu, std::adopt_lock);
            TkSmallReqs R{};
            uint32_t cnt = 0;
            for (uint32_t j = 0; j < TK_SMALL_SLOTS && cnt < TK_SMALL_BATCH; ++j) {
                tk_core::SmallSlot& q = c->small[j];
                int expect = 1;
                if (q.state.load(std::memory_order_acquire) == 1 && q.state.compare_exchange_strong(expect, 2, std::memory_order_acq_rel))
                    R.r[cnt++] = TkSmallReq{(const uint8_t*)q.d_in, (uint32_t*)q.d_out, q.ws.as<uint32_t>(), q.n, q.seq};
            }
            if (cnt) {
                hipStream_t& ls = c->small_s[c->small_turn++ & 3u];
                if (!ls) HIPCHK(hipStreamCreateWithFlags(&ls, hipStreamNonBlocking));
                hipLaunchKernelGGL(tk_k_small, dim3(cnt), dim3(256), 0, ls, c->D, R);
                const hipError_t le = hipGetLastError();
                if (le != hipSuccess) {
                    // (the slots this launcher took go back to "ready": their owners try the launch themselves instead of waiting two seconds)
                    for (uint32_t j = 0; j < TK_SMALL_SLOTS; ++j) {
                        int two = 2;
                        for (uint32_t q = 0; q < cnt; ++q)
                            if (R.r[q].out == (uint32_t*)c->small[j].d_out) c->small[j].state.compare_exchange_strong(two, 1, std::memory_order_acq_rel);
                    }
                    return fail(TK_RUNTIME_ERROR, std::string("HIP error: ") + hipGetErrorString(le) + " in tk_k_small");
                }
                c->st_small_launches += 1;
                c->st_small_calls += cnt;
            }
            continue;
        }
        if ((++spins & 0xFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            (void)hipDeviceSynchronize();
            for (uint32_t i = 0; i < k; ++i)
                if (__atomic_load_n(&mine[i]->out[2], __ATOMIC_ACQUIRE) != mine[i]->seq) return fail(TK_RUNTIME_ERROR, "the small-call kernel did not complete");
            break;
        }
        if ((spins & 127u) == 127u && c->small_active.load(std::memory_order_relaxed) > 8) std::this_thread::yield();  // (many callers, maybe more than cores: a spinning waiter must not keep the launcher off its core; a lone caller never yields)
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    for (uint32_t i = 0; i < k; ++i) mine[i]->state.store(0, std::memory_order_release);
    return TK_OK;
}

// One document of 2 .. 128 KiB without special tokens: cut into segments of at most TK_SMALL_MAX bytes at piece starts that are certain whatever
// stands on either side (an ASCII letter followed by a space, where the pattern's table says so: c->mid_cut), the segments encoded as so
// many small calls in ONE launch (a workgroup each), their tokens put together on the host.  The general pipeline costs a dozen dependent
// launches -- 0.15 ms for 4 KiB; this is one.  *handled = false: no cut where one is needed, not enough free slots, or a segment
// the small kernel does not do (a long piece that is not a token): the general path takes the call.
static int encode_mid(tk_core* c, const uint8_t* utf8, uint32_t n, uint32_t** tokens_out, uint64_t* n_tokens_out, bool* handled) {
    *handled = false;
    auto why = [&](const char* r) {
        if (c->dbg & 64) fprintf(stderr, "encode_mid: %u bytes not taken: %s\n", n, r);
        return TK_OK;
    };
    if (!c->mid_cut) return why("letter -> space is not a certain start of this pattern");
    // (text full of long pieces that are not tokens -- URLs, runs of a script without spaces -- is not for the small kernel: after a call that
    // found that out, the next 16 .. 64 go straight to the general pipeline instead of paying for a launch first)
    if (c->mid_skip.load(std::memory_order_relaxed) > 0) {
        c->mid_skip.fetch_sub(1, std::memory_order_relaxed);
        return why("the last attempt met a long piece that is not a token");
    }
    // cuts: about equal segments, each from one certain piece start to the next (tk_mid_plan.h)
    static_assert(TK_MID_SEGMENT_MAX == TK_SMALL_MAX && TK_SMALL_SLOTS == TK_SMALL_BATCH, "a segment is one small call; one launch carries every slot");
    uint32_t cuts[TK_SMALL_SLOTS + 1];
    const char* reason = nullptr;
    const uint32_t k = tk_mid_plan(utf8, n, cuts, &reason);
    if (!k) return why(reason);
    // k free slots, or none
    tk_core::SmallSlot* mine[TK_SMALL_SLOTS];
    uint32_t got = 0;
    for (uint32_t j = 0; j < TK_SMALL_SLOTS && got < k; ++j) {
        tk_core::SmallSlot& cand = c->small[j];
        if (!cand.busy.load(std::memory_order_relaxed) && !cand.busy.exchange(1, std::memory_order_acquire)) mine[got++] = &cand;
    }
    struct Release {
        tk_core::SmallSlot** s;
        uint32_t* n;
        std::atomic<int>* active;
        ~Release() {
            for (uint32_t i = 0; i < *n; ++i) small_slot_release(s[i]);  // (also when small_wait left early -- a failed launch, its time-out: the next owner must not be launched with this call's text)
            active->fetch_sub(1, std::memory_order_relaxed);
        }
    } release_slots{mine, &got, &c->small_active};
    c->small_active.fetch_add(1, std::memory_order_relaxed);
    if (got < k) return why("not enough free slots");  // (other callers hold them: the general path)
    HIPCHK(hipSetDevice(c->device));
    for (uint32_t i = 0; i < k; ++i) TRY(small_slot_init(c, mine[i]));
    // (segments leave EVERY piece of more than TK_SMALL_PIECE bytes that is not a token to the general pipeline: sixty-four workgroups each
    // waiting for its longest chain of merges cost more than the pipeline, whose merge kernel runs all the chains of the document side by side
    // -- measured on web text, profiles/r04_mid_calls_corpus.txt)
    // (documents of up to 6 KiB: the segments merge their long pieces themselves -- 4 KiB of web text 140 us against 162 when a segment gives the
    // document up at such a piece, and 170 through the general pipeline; from 16 KiB on the pipeline wins on such text either way:
    // profiles/r05_small_variants.txt.  Debug bit 0x20000000: always.)
    const bool keep_long = n <= 6144u || (c->dbg & 0x20000000);
    for (uint32_t i = 0; i < k; ++i) small_slot_submit(mine[i], utf8 + cuts[i], cuts[i + 1] - cuts[i], !keep_long);
    TRY(small_wait(c, mine, k));
    // the segments' tokens, one after the other.  A segment the small kernel did not do sends the WHOLE document to the general pipeline (one pass
    // over 64 KiB costs it little more than one over 2 KiB; segment by segment it was 1.4x the pipeline on web text), and the next calls do not
    // even try: 16 of them after the first such document, 32 and then 64 after further ones in a row, 16 again after a success
    uint64_t nt = 0;
    bool all_done = true;
    for (uint32_t i = 0; i < k; ++i) {
        if (mine[i]->out[0] != 1u) all_done = false;
        else nt += mine[i]->out[1];
    }
    if (!all_done) {
        const int run = std::min(std::max(c->mid_fail_run.load(std::memory_order_relaxed), 0), 2);  // (a lost update between two threads costs a launch, no more)
        c->mid_fail_run.store(std::min(run + 1, 2), std::memory_order_relaxed);
        c->mid_skip.store(16 << run, std::memory_order_relaxed);
        return why("a segment with a long piece that is not a token");
    }
    c->mid_fail_run.store(0, std::memory_order_relaxed);
    uint32_t* host = (uint32_t*)malloc((nt ? nt : 1) * 4);
    if (!host) return fail(TK_RUNTIME_ERROR, "out of host memory");
    uint64_t at = 0;
    for (uint32_t i = 0; i < k; ++i) {
        const uint32_t cnt = mine[i]->out[1];
        if (cnt) memcpy(host + at, mine[i]->out + TK_SMALL_HDR, (size_t)cnt * 4);
        at += cnt;
    }
    __atomic_store_n(&c->st_bytes, (uint64_t)n, __ATOMIC_RELAXED);
    __atomic_store_n(&c->st_tokens, nt, __ATOMIC_RELAXED);
    __atomic_store_n(&c->st_docs, (uint64_t)1, __ATOMIC_RELAXED);
    __atomic_store_n(&c->st_pieces, (uint64_t)0, __ATOMIC_RELAXED);
    __atomic_fetch_add(&c->st_mid_calls, (uint64_t)1, __ATOMIC_RELAXED);
    *tokens_out = host;
    *n_tokens_out = nt;
    *handled = true;
    return TK_OK;
}

// One short document without special tokens: one launch, no copies, no stream synchronisation (tk_k_small, tk_fused.h), and no lock: the
// call runs on a slot of its own (tk_core::SmallSlot), so the threads of a caller's pool overlap (core.py:175).
// Returns TK_OK with *handled = false when the call has to take the general path.
static int encode_small(tk_core* c, const uint8_t* utf8, uint32_t n, uint32_t** tokens_out, uint64_t* n_tokens_out, bool* handled) {
    *handled = false;
    // a free slot: start at a place of this thread's own, so that a pool of callers does not fight over slot 0
    static std::atomic<uint32_t> next_thread{0};
    static thread_local uint32_t home = next_thread.fetch_add(1, std::memory_order_relaxed);
    tk_core::SmallSlot* sl = nullptr;
    for (uint32_t spins = 0; !sl; ++spins) {
        for (uint32_t k = 0; k < TK_SMALL_SLOTS && !sl; ++k) {
            tk_core::SmallSlot& cand = c->small[(home + k) % TK_SMALL_SLOTS];
            if (!cand.busy.load(std::memory_order_relaxed) && !cand.busy.exchange(1, std::memory_order_acquire)) sl = &cand;
        }
        if (!sl) {
            if (spins > 64) std::this_thread::yield();  // (more callers than slots: a slot is held for some tens of microseconds)
#if defined(__x86_64__)
            else __builtin_ia32_pause();
#endif
        }
    }
    struct Release {
        tk_core::SmallSlot* s;
        std::atomic<int>* active;
        ~Release() {
            small_slot_release(s);  // (see encode_mid's guard)
            active->fetch_sub(1, std::memory_order_relaxed);
        }
    } release_slot{sl, &c->small_active};
    c->small_active.fetch_add(1, std::memory_order_relaxed);
    HIPCHK(hipSetDevice(c->device));
    TRY(small_slot_init(c, sl));
    if (c->profiling) {  // (kernel times are collected in the core's shared list: one caller at a time then, every call a launch of its own)
        std::lock_guard<std::mutex> lk(c->mu);
        memcpy(sl->in, utf8, n);
        memset(sl->in + n, 0, 8);
        sl->seq = ++sl->seq ? sl->seq : ++sl->seq;
        sl->n = n;
        if (!sl->s) HIPCHK(hipStreamCreateWithFlags(&sl->s, hipStreamNonBlocking));
        TkSmallReqs R{};
        R.r[0] = TkSmallReq{(const uint8_t*)sl->d_in, (uint32_t*)sl->d_out, sl->ws.as<uint32_t>(), n, sl->seq};
        TRY(timed(c, sl->s, "tk_k_small", [&] { hipLaunchKernelGGL(tk_k_small, dim3(1), dim3(256), 0, sl->s, c->D, R); }));
        HIPCHK(hipStreamSynchronize(sl->s));
        TRY(drain_events(c));
    } else {
        small_slot_submit(sl, utf8, n);
    }
    TRY(small_wait(c, &sl, 1));
    if (sl->out[0] != 1u) return TK_OK;  // a long piece that is not a token: general path
    const uint64_t nt = sl->out[1];
    uint32_t* host = (uint32_t*)malloc((nt ? nt : 1) * 4);
    if (!host) return fail(TK_RUNTIME_ERROR, "out of host memory");
    memcpy(host, sl->out + TK_SMALL_HDR, nt * 4);
    // (statistics of the last call: plain words, written without the lock -- with several callers "the last call" is whoever came last)
    __atomic_store_n(&c->st_bytes, (uint64_t)n, __ATOMIC_RELAXED);
    __atomic_store_n(&c->st_tokens, nt, __ATOMIC_RELAXED);
    __atomic_store_n(&c->st_docs, (uint64_t)1, __ATOMIC_RELAXED);
    __atomic_store_n(&c->st_pieces, (uint64_t)0, __ATOMIC_RELAXED);
    __atomic_store_n(&c->st_medium, (uint64_t)0, __ATOMIC_RELAXED);
    __atomic_store_n(&c->st_long, (uint64_t)0, __ATOMIC_RELAXED);
    *tokens_out = host;
    *n_tokens_out = nt;
    *handled = true;
    return TK_OK;
}

// Host text in.  device_result = false: the contract of tk_encode_batch (ids in a host buffer the caller frees).  device_result = true: the ids
// stay on the device (c->out_tokens, c->out_tok_off: valid until the core's next call) and only the text crosses PCIe -- what the
// several-GPU gather needs; the one-launch small path (which writes straight to host memory) is not taken then.
static int encode_batch_impl(tk_core* c, const uint8_t* utf8, const uint64_t* doc_off, uint64_t n_docs, int use_special,
                             const uint32_t* allowed_ids, uint64_t n_allowed, uint32_t** tokens_out, uint64_t* n_tokens_out,
                             uint64_t* tok_off_out, bool device_result, bool no_small) {
    if (!c) return fail(TK_VALUE_ERROR, "core is null");
    if (!doc_off || (!device_result && !tokens_out) || !n_tokens_out) return fail(TK_VALUE_ERROR, "null argument");
    if (doc_off[0] != 0) return fail(TK_VALUE_ERROR, "doc_off[0] must be 0");
    for (uint64_t d = 0; d < n_docs; ++d)
        if (doc_off[d + 1] < doc_off[d]) return fail(TK_VALUE_ERROR, "doc_off must be non-decreasing");
    const uint64_t n_bytes = doc_off[n_docs];
    if (!no_small && !device_result && n_docs == 1 && n_bytes > 0 && n_bytes <= (uint64_t)TK_SMALL_MAX * TK_MID_SEGMENTS && !(use_special && n_allowed) && !(c->dbg & 2048) && !c->has_rx &&
        (n_bytes <= TK_SMALL_MAX || !c->profiling)) {
        bool handled = false;  // (before the lock: small calls of several threads run side by side)
        if (n_bytes <= TK_SMALL_MAX) TRY(encode_small(c, utf8, (uint32_t)n_bytes, tokens_out, n_tokens_out, &handled));
        else TRY(encode_mid(c, utf8, (uint32_t)n_bytes, tokens_out, n_tokens_out, &handled));
        if (handled) {
            if (tok_off_out) {
                tok_off_out[0] = 0;
                tok_off_out[1] = *n_tokens_out;
            }
            return TK_OK;
        }
    }
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    TRY(ensure(c->text, n_bytes + 256));
    TRY(ensure(c->doc_off, (n_docs + 2) * 8));
    HIPCHK(hipMemsetAsync((uint8_t*)c->text.p + n_bytes, 0, 128, s));
    HIPCHK(hipMemcpyAsync(c->doc_off.p, doc_off, (n_docs + 1) * 8, hipMemcpyHostToDevice, s));
    bool any = false;
    if (use_special) TRY(prepare_allowed(c, s, allowed_ids, n_allowed, &any));
    uint64_t total = 0;
    if (n_bytes < 2 * TK_STAGE_BYTES) {
        // small batches: one copy each way (latency matters more than overlap)
        if (n_bytes) HIPCHK(hipMemcpyAsync(c->text.p, utf8, n_bytes, hipMemcpyHostToDevice, s));
        TRY(encode_device_locked(c, s, c->text.as<uint8_t>(), n_bytes, c->doc_off.as<uint64_t>(), doc_off, n_docs, use_special && any, &total));
        if (device_result) {
            *n_tokens_out = total;
            return TK_OK;
        }
        uint32_t* host = (uint32_t*)(total * 4 >= (1u << 20) ? pinned_get((total ? total : 1) * 4) : malloc((total ? total : 1) * 4));
        if (!host) return fail(TK_RUNTIME_ERROR, "out of host memory");
        hipError_t e = hipSuccess;
        if (total) e = hipMemcpy(host, c->out_tokens.p, total * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && tok_off_out) e = hipMemcpy(tok_off_out, c->out_tok_off.p, (n_docs + 1) * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            tk_free(host);
            return fail(TK_RUNTIME_ERROR, std::string("HIP error: ") + hipGetErrorString(e));
        }
        *tokens_out = host;
        *n_tokens_out = total;
        return TK_OK;
    }
    // ---- pipelined
    if (!c->cs_h2d) {
        HIPCHK(hipStreamCreateWithFlags(&c->cs_h2d, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&c->cs_d2h, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipHostMalloc(&c->stage[i], TK_STAGE_BYTES, hipHostMallocPortable));
            HIPCHK(hipEventCreateWithFlags(&c->ev_stage[i], hipEventDisableTiming));
        }
    }
    // How the text gets to the device.  Round 6 measured the link on the bench's box (tools/ubench/pcie_rates.hip, profiles/r06_pcie_link.txt):
    // 57 GB/s either way alone, 48 GB/s each way with both directions busy -- and hipMemcpyAsync straight from PAGEABLE memory at 56.5 GB/s,
    // the runtime's own staging.  So the default is that copy, in blocks of 64 MiB (smaller ones cost the copy its rate: 28 ms with 16 MiB), and
    // chunks of 32 MiB: 25 ms per GiB = 0.88 of what the link gives in both directions at once, where the staging buffers of rounds 2-5
    // (filled by host threads, 64 MiB at a time, chunks of 128 MiB) took 32 ms.  $TIKTOKEN_AMD_H2D_DIRECT=0 brings those back.
    const bool h2d_direct = !(getenv("TIKTOKEN_AMD_H2D_DIRECT") && atoi(getenv("TIKTOKEN_AMD_H2D_DIRECT")) == 0);
    uint64_t block_bytes = TK_STAGE_BYTES;
    if (const char* e = getenv("TIKTOKEN_AMD_H2D_BLOCK_MIB"))  // (experiments; the staging buffers bound it)
        if (h2d_direct && atol(e) >= 4 && atol(e) <= 1024) block_bytes = (uint64_t)atol(e) << 20;
    const uint64_t n_blocks = (n_bytes + block_bytes - 1) / block_bytes;
    std::vector<hipEvent_t> ev_block(n_blocks, nullptr);
    for (auto& e : ev_block) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    auto drop_events = [&]() {
        for (auto e : ev_block)
            if (e) (void)hipEventDestroy(e);
    };
    // producer: stage + send the text, block by block
    std::atomic<int> h2d_rc{TK_OK};
    std::atomic<uint64_t> blocks_sent{0};
    const int dev = c->device;
    unsigned nth = std::thread::hardware_concurrency();
    nth = copy_threads(nth);
    std::thread producer([&]() {
        (void)hipSetDevice(dev);
        for (uint64_t b = 0; b < n_blocks; ++b) {
            const int slot = (int)(b & 1);
            const uint64_t a = b * block_bytes, len = a + block_bytes < n_bytes ? block_bytes : n_bytes - a;
            if (h2d_direct) {  // (the runtime's own way from pageable memory: measured at the link's rate on this platform, tools/ubench/pcie_rates.hip)
                if (hipMemcpyAsync((uint8_t*)c->text.p + a, utf8 + a, len, hipMemcpyHostToDevice, c->cs_h2d) != hipSuccess) h2d_rc = TK_RUNTIME_ERROR;
            } else {
                if (b >= 2 && hipEventSynchronize(c->ev_stage[slot]) != hipSuccess) h2d_rc = TK_RUNTIME_ERROR;  // the slot's previous DMA is done
                parallel_memcpy(c->stage[slot], utf8 + a, len, nth);
                if (hipMemcpyAsync((uint8_t*)c->text.p + a, c->stage[slot], len, hipMemcpyHostToDevice, c->cs_h2d) != hipSuccess) h2d_rc = TK_RUNTIME_ERROR;
            }
            (void)hipEventRecord(c->ev_stage[slot], c->cs_h2d);
            (void)hipEventRecord(ev_block[b], c->cs_h2d);
            blocks_sent.store(b + 1, std::memory_order_release);
        }
    });
    // consumer side
    uint32_t* host = nullptr;
    uint64_t host_cap = 0;  // tokens
    ChunkHooks hooks;
    hooks.before = [&](uint64_t byte_end) -> int {
        const uint64_t need = (byte_end + block_bytes - 1) / block_bytes;  // blocks [0, need) must have been sent
        while (blocks_sent.load(std::memory_order_acquire) < need) std::this_thread::yield();
        if (need) HIPCHK(hipStreamWaitEvent(s, ev_block[need - 1], 0));
        return h2d_rc.load() == TK_OK ? TK_OK : fail(TK_RUNTIME_ERROR, "host-to-device copy failed");
    };
    uint64_t bytes_done = 0;
    if (!device_result) hooks.after = [&](uint64_t tok_begin, uint64_t n_tok, bool last) -> int {
        bytes_done = c->st_bytes;  // (bytes encoded so far: the density of the chunks seen sizes the result buffer)
        const uint64_t need = tok_begin + n_tok;
        if (need > host_cap || !host) {
            uint64_t est = last ? need : (uint64_t)((double)need / (double)(bytes_done ? bytes_done : 1) * (double)n_bytes * 1.08) + 4096;
            if (est < need) est = need;
            uint32_t* nh = (uint32_t*)pinned_get((est ? est : 1) * 4);
            if (!nh) return fail(TK_RUNTIME_ERROR, "out of page-locked host memory");
            if (host) {
                HIPCHK(hipStreamSynchronize(c->cs_d2h));
                if (tok_begin) parallel_memcpy(nh, host, tok_begin * 4, nth);
                tk_free(host);
            }
            host = nh;
            host_cap = est;
        }
        if (n_tok) HIPCHK(hipMemcpyAsync(host + tok_begin, c->out_tokens.as<uint32_t>() + tok_begin, n_tok * 4, hipMemcpyDeviceToHost, c->cs_d2h));
        return TK_OK;
    };
    int rc = encode_device_locked(c, s, c->text.as<uint8_t>(), n_bytes, c->doc_off.as<uint64_t>(), doc_off, n_docs, use_special && any, &total,
                                  host_chunk_bytes(h2d_direct), &hooks);
    producer.join();
    hipError_t e = hipStreamSynchronize(c->cs_h2d);
    if (e == hipSuccess) e = hipStreamSynchronize(c->cs_d2h);
    drop_events();
    if (rc == TK_OK && e != hipSuccess) rc = fail(TK_RUNTIME_ERROR, std::string("HIP error: ") + hipGetErrorString(e));
    if (rc == TK_OK && tok_off_out && !device_result) {
        e = hipMemcpy(tok_off_out, c->out_tok_off.p, (n_docs + 1) * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(TK_RUNTIME_ERROR, std::string("HIP error: ") + hipGetErrorString(e));
    }
    if (rc != TK_OK) {
        tk_free(host);
        return rc;
    }
    *n_tokens_out = total;
    if (device_result) return TK_OK;
    if (!host) host = (uint32_t*)pinned_get(64);
    *tokens_out = host;
    return TK_OK;
}

extern "C" int tk_encode_batch(tk_core* c, const uint8_t* utf8, const uint64_t* doc_off, uint64_t n_docs, int use_special,
                               const uint32_t* allowed_ids, uint64_t n_allowed, uint32_t** tokens_out, uint64_t* n_tokens_out,
                               uint64_t* tok_off_out) {
    return encode_batch_impl(c, utf8, doc_off, n_docs, use_special, allowed_ids, n_allowed, tokens_out, n_tokens_out, tok_off_out, false, false);
}

// Debug / test entry: the piece-start offsets the GPU pre-tokeniser produces for a packed batch
// (what regex.find_iter yields at src/lib.rs:365 and :405).  *starts_out gets n_pieces+1 uint32
// values (ascending piece starts, then the total byte count); release with tk_free.
extern "C" int tk_pretokenize_batch(tk_core* c, const uint8_t* utf8, const uint64_t* doc_off, uint64_t n_docs, int use_special,
                                    const uint32_t* allowed_ids, uint64_t n_allowed, uint32_t** starts_out, uint64_t* n_out) {
    if (!c) return fail(TK_VALUE_ERROR, "core is null");
    if (!doc_off || !starts_out || !n_out) return fail(TK_VALUE_ERROR, "null argument");
    if (doc_off[0] != 0) return fail(TK_VALUE_ERROR, "doc_off[0] must be 0");
    for (uint64_t d = 0; d < n_docs; ++d)
        if (doc_off[d + 1] < doc_off[d]) return fail(TK_VALUE_ERROR, "doc_off must be non-decreasing");
    const uint64_t n_bytes = doc_off[n_docs];
    if (n_bytes > c->chunk_bytes) return fail(TK_VALUE_ERROR, "tk_pretokenize_batch handles a single chunk only");
    if (n_bytes >= (1ull << 31)) return fail(TK_VALUE_ERROR, "tk_pretokenize_batch: less than 2 GiB per call (bit 31 of an offset marks a gap char)");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    TRY(ensure(c->text, n_bytes + 256));
    TRY(ensure(c->doc_off, (n_docs + 2) * 8));
    if (n_bytes) HIPCHK(hipMemcpyAsync(c->text.p, utf8, n_bytes, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync((uint8_t*)c->text.p + n_bytes, 0, 128, s));
    HIPCHK(hipMemcpyAsync(c->doc_off.p, doc_off, (n_docs + 1) * 8, hipMemcpyHostToDevice, s));
    bool any = false;
    if (use_special) TRY(prepare_allowed(c, s, allowed_ids, n_allowed, &any));
    uint64_t P = 0;
    TRY(run_chunk(c, s, c->text.as<uint8_t>(), n_bytes, c->doc_off.as<uint64_t>(), n_docs, 0, use_special && any, false, nullptr, nullptr, &P, true));
    HIPCHK(hipStreamSynchronize(s));
    TRY(drain_events(c));
    uint32_t* host = (uint32_t*)malloc((P + 1) * 4);
    if (!host) return fail(TK_RUNTIME_ERROR, "out of host memory");
    {
        hipError_t e = hipMemcpy(host, c->ws[0].pstart.p, (P + 1) * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            free(host);
            return fail(TK_RUNTIME_ERROR, std::string("HIP error: ") + hipGetErrorString(e));
        }
    }
    *starts_out = host;
    *n_out = P + 1;
    return TK_OK;
}

extern "C" int tk_encode_ordinary(tk_core* c, const uint8_t* utf8, uint64_t len, uint32_t** tokens_out, uint64_t* n_tokens_out) {
    uint64_t off[2] = {0, len};
    return tk_encode_batch(c, utf8, off, 1, 0, nullptr, 0, tokens_out, n_tokens_out, nullptr);
}

extern "C" int tk_encode(tk_core* c, const uint8_t* utf8, uint64_t len, const uint32_t* allowed_ids, uint64_t n_allowed,
                         uint32_t** tokens_out, uint64_t* n_tokens_out) {
    uint64_t off[2] = {0, len};
    return tk_encode_batch(c, utf8, off, 1, 1, allowed_ids, n_allowed, tokens_out, n_tokens_out, nullptr);
}

static int single_piece(tk_core* c, const uint8_t* piece, uint64_t len, bool no_lookup, uint32_t** tokens_out, uint64_t* n_tokens_out) {
    if (!c) return fail(TK_VALUE_ERROR, "core is null");
    if (!tokens_out || !n_tokens_out) return fail(TK_VALUE_ERROR, "null argument");
    if (len >= (4ull << 30) - 65536) return fail(TK_VALUE_ERROR, "piece too long");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    c->st_bytes = c->st_pieces = c->st_tokens = c->st_medium = c->st_long = c->st_chunks = 0;
    TRY(ensure(c->text, len + 256));
    if (len) HIPCHK(hipMemcpyAsync(c->text.p, piece, len, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync((uint8_t*)c->text.p + len, 0, 128, s));
    TRY(ensure(c->out_tokens, (len + 64) * 4));
    uint64_t total = 0;
    TRY(run_chunk(c, s, c->text.as<uint8_t>(), len, nullptr, 0, 0, false, true, c->out_tokens.as<uint32_t>(), nullptr, &total, false, no_lookup));
    HIPCHK(hipStreamSynchronize(s));
    TRY(drain_events(c));
    uint32_t* host = (uint32_t*)malloc((total ? total : 1) * 4);
    if (!host) return fail(TK_RUNTIME_ERROR, "out of host memory");
    if (total) {
        hipError_t e = hipMemcpy(host, c->out_tokens.p, total * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            free(host);
            return fail(TK_RUNTIME_ERROR, std::string("HIP error: ") + hipGetErrorString(e));
        }
    }
    *tokens_out = host;
    *n_tokens_out = total;
    return TK_OK;
}

extern "C" int tk_encode_single_piece(tk_core* c, const uint8_t* piece, uint64_t len, uint32_t** tokens_out, uint64_t* n_tokens_out) {
    return single_piece(c, piece, len, false, tokens_out, n_tokens_out);
}

extern "C" int tk_byte_pair_encode(tk_core* c, const uint8_t* piece, uint64_t len, uint32_t** tokens_out, uint64_t* n_tokens_out) {
    return single_piece(c, piece, len, true, tokens_out, n_tokens_out);
}

extern "C" int tk_encode_single_token(tk_core* c, const uint8_t* piece, uint64_t len, uint32_t* token_out) {
    if (!c) return fail(TK_VALUE_ERROR, "core is null");
    uint32_t r = len < 0xFFFFFFFFull ? c->H.lookup_piece(piece, (uint32_t)len) : TK_RANK_MAX;
    if (r == TK_RANK_MAX) {
        const TkHostTables& H = c->H;
        for (size_t k = 0; k + 1 < H.spec_off.size(); ++k)
            if (H.spec_off[k + 1] - H.spec_off[k] == len && memcmp(H.spec_bytes.data() + H.spec_off[k], piece, len) == 0) r = H.spec_id[k];
    }
    if (r == TK_RANK_MAX) return fail(TK_KEY_ERROR, "token not found");
    *token_out = r;
    return TK_OK;
}

extern "C" int tk_decode_single_token_bytes(tk_core* c, uint32_t token, const uint8_t** bytes_out, uint64_t* len_out) {
    if (!c) return fail(TK_VALUE_ERROR, "core is null");
    if (const auto* e = c->H.find_token(token)) {
        *bytes_out = c->H.tok_bytes.data() + e->first;
        *len_out = e->second;
        return TK_OK;
    }
    auto it = c->H.spec_decoder.find(token);
    if (it != c->H.spec_decoder.end()) {
        *bytes_out = c->H.spec_bytes.data() + it->second.first;
        *len_out = it->second.second;
        return TK_OK;
    }
    return fail(TK_KEY_ERROR, std::to_string(token));
}

extern "C" int tk_decode_bytes(tk_core* c, const uint32_t* tokens, uint64_t n, uint8_t** bytes_out, uint64_t* len_out) {
    if (!c) return fail(TK_VALUE_ERROR, "core is null");
    if (n >= 8192 && c->n_dec) {  // long inputs: on the device (short ones are quicker on the host than a launch)
        const uint64_t off[2] = {0, n};
        return tk_decode_batch(c, tokens, off, 1, bytes_out, len_out, nullptr);
    }
    std::string acc;
    acc.reserve(n * 4);
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t* p;
        uint64_t l;
        if (tk_decode_single_token_bytes(c, tokens[i], &p, &l) != TK_OK)
            return fail(TK_KEY_ERROR, "Invalid token for decoding: " + std::to_string(tokens[i]));
        acc.append((const char*)p, l);
    }
    uint8_t* host = (uint8_t*)malloc(acc.size() ? acc.size() : 1);
    memcpy(host, acc.data(), acc.size());
    *bytes_out = host;
    *len_out = acc.size();
    return TK_OK;
}

// CoreBPE.decode_bytes over a packed batch (Encoding.decode_bytes_batch / decode_batch, tiktoken/core.py:331-350), on the device.
// One range of a packed batch on the device: lengths and their block sums, the scan, then (decode_range_copy) the bytes.
//   d_tok: the batch's ids on the device; [a, a + cnt) the range (a: a multiple of TK_DEC_BLOCK); `tot`: two device words {bytes of the
//   range, first invalid position (the caller sets it to ~0 once)}
static int decode_range_len(tk_core* c, hipStream_t s, const uint32_t* d_tok, uint64_t a, uint64_t cnt, unsigned long long* tot) {
    const uint64_t nb = (cnt + TK_DEC_BLOCK - 1) / TK_DEC_BLOCK;
    unsigned long long* bsum = c->d_bsum.as<unsigned long long>() + a / TK_DEC_BLOCK;
    TRY(timed(c, s, "tk_k_dec_len", [&] {
        hipLaunchKernelGGL(tk_k_dec_len, dim3((uint32_t)nb), dim3(256), 0, s, d_tok + a, cnt, c->t_dec.as<uint2>(), c->n_dec, c->d_lens.as<uint32_t>() + a, bsum, tot + 1, a);
    }));
    hipLaunchKernelGGL(tk_k_dec_scan64, dim3(1), dim3(1024), 0, s, bsum, nb, tot);
    return TK_OK;
}
static int decode_range_copy(tk_core* c, hipStream_t s, const uint32_t* d_tok, uint64_t a, uint64_t cnt, uint8_t* d_out, unsigned long long* d_tboff,
                             uint64_t byte_base) {
    const uint64_t nb = (cnt + TK_DEC_BLOCK - 1) / TK_DEC_BLOCK;
    TRY(timed(c, s, "tk_k_dec_copy", [&] {
        hipLaunchKernelGGL(tk_k_dec_copy, dim3((uint32_t)nb), dim3(256), 0, s, d_tok + a, cnt, c->t_dec.as<uint2>(), c->d_lens.as<uint32_t>() + a,
                           c->d_bsum.as<unsigned long long>() + a / TK_DEC_BLOCK, c->D.tok_bytes, c->D.spec_bytes, d_out, d_tboff ? d_tboff + a : nullptr,
                           (unsigned long long)byte_base);
    }));
    return TK_OK;
}

// Device-resident decode: ids and token offsets in HBM in, bytes and byte offsets in HBM out (library-owned buffers, valid until the core's
// next decode call).  The host learns the byte count (one 16-byte copy): the output buffer is sized by it.
extern "C" int tk_decode_batch_device(tk_core* c, const void* d_tokens, uint64_t n_tokens, const void* d_tok_off, uint64_t n_docs, void* stream,
                                      const uint8_t** d_bytes_out, uint64_t* n_bytes_out, const uint64_t** d_byte_off_out) {
    if (!c) return fail(TK_VALUE_ERROR, "core is null");
    if (!n_bytes_out || (n_tokens && !d_tokens)) return fail(TK_VALUE_ERROR, "null argument");
    if (!c->n_dec) return fail(TK_UNSUPPORTED, "token ids are too sparse for the device decode table");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const uint64_t n = n_tokens, nb = (n + TK_DEC_BLOCK - 1) / TK_DEC_BLOCK;
    TRY(ensure(c->d_lens, (n + 1) * 4));
    TRY(ensure(c->d_bsum, (nb + 8) * 8));
    if (d_tok_off) TRY(ensure(c->d_tboff, (n + 1) * 8));
    TRY(ensure(c->d_boff, (n_docs + 2) * 8 * 2));
    unsigned long long* tot = c->d_bsum.as<unsigned long long>() + nb + 2;
    unsigned long long h_tot[2] = {0, ~0ull};
    HIPCHK(hipMemcpyAsync(tot, h_tot, 16, hipMemcpyHostToDevice, s));
    if (n) TRY(decode_range_len(c, s, (const uint32_t*)d_tokens, 0, n, tot));
    HIPCHK(hipMemcpyAsync(h_tot, tot, 16, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (h_tot[1] != ~0ull) {
        uint32_t bad_tok = 0;
        (void)hipMemcpy(&bad_tok, (const uint32_t*)d_tokens + h_tot[1], 4, hipMemcpyDeviceToHost);
        return fail(TK_KEY_ERROR, "Invalid token for decoding: " + std::to_string(bad_tok));
    }
    const uint64_t nbytes = h_tot[0];
    TRY(ensure(c->d_bytes, nbytes + 16));
    if (n) TRY(decode_range_copy(c, s, (const uint32_t*)d_tokens, 0, n, c->d_bytes.as<uint8_t>(), d_tok_off ? c->d_tboff.as<unsigned long long>() : nullptr, 0));
    uint64_t* d_byte_off = c->d_boff.as<uint64_t>() + n_docs + 1;
    if (d_tok_off)
        hipLaunchKernelGGL(tk_k_dec_docoff, dim3(grid_for(n_docs + 1, 256, 4096)), dim3(256), 0, s, (const uint64_t*)d_tok_off, n_docs, n,
                           c->d_tboff.as<unsigned long long>(), tot, d_byte_off);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    TRY(drain_events(c));
    if (d_bytes_out) *d_bytes_out = c->d_bytes.as<uint8_t>();
    if (d_byte_off_out) *d_byte_off_out = d_tok_off ? d_byte_off : nullptr;
    *n_bytes_out = nbytes;
    return TK_OK;
}

// Host buffers in, host buffers out.  The batch is decoded in ranges of 16 Mi ids: the ids of range k + 1 travel to the device (through
// two page-locked staging buffers filled by a few host threads, or straight from the caller's buffer when that is page-locked itself:
// the result of an encode call is) while range k is decoded and the bytes of range k - 1 travel back -- both directions of the link at
// once.  The result buffer is page-locked and sized by the density of the ranges seen so far (grown by a copy if a later range is
// denser).  A batch of less than two ranges takes one copy each way.
extern "C" int tk_decode_batch(tk_core* c, const uint32_t* tokens, const uint64_t* tok_off, uint64_t n_docs, uint8_t** bytes_out,
                               uint64_t* n_bytes_out, uint64_t* byte_off_out) {
    if (!c) return fail(TK_VALUE_ERROR, "core is null");
    if (!tok_off || !bytes_out || !n_bytes_out) return fail(TK_VALUE_ERROR, "null argument");
    if (tok_off[0] != 0) return fail(TK_VALUE_ERROR, "tok_off[0] must be 0");
    for (uint64_t d = 0; d < n_docs; ++d)
        if (tok_off[d + 1] < tok_off[d]) return fail(TK_VALUE_ERROR, "tok_off must be non-decreasing");
    if (!c->n_dec) return fail(TK_UNSUPPORTED, "token ids are too sparse for the device decode table");
    const uint64_t n = tok_off[n_docs];
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const uint64_t nb = (n + TK_DEC_BLOCK - 1) / TK_DEC_BLOCK;
    constexpr uint64_t RANGE = TK_STAGE_BYTES / 4;  // ids per range (a multiple of TK_DEC_BLOCK)
    const uint64_t n_ranges = n ? (n + RANGE - 1) / RANGE : 0;
    TRY(ensure(c->d_tok, (n + 1) * 4));
    TRY(ensure(c->d_lens, (n + 1) * 4));
    TRY(ensure(c->d_bsum, (nb + 8 + 2 * n_ranges) * 8));
    if (byte_off_out) TRY(ensure(c->d_tboff, (n + 1) * 8));
    TRY(ensure(c->d_boff, (n_docs + 2) * 8 * 2));
    unsigned long long* tots = c->d_bsum.as<unsigned long long>() + nb + 2;  // per range {bytes, first invalid position}; [0 .. 1] also the batch's
    unsigned long long* d_tboff = byte_off_out ? c->d_tboff.as<unsigned long long>() : nullptr;
    if (!c->cs_h2d) {
        HIPCHK(hipStreamCreateWithFlags(&c->cs_h2d, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&c->cs_d2h, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipHostMalloc(&c->stage[i], TK_STAGE_BYTES, hipHostMallocPortable));
            HIPCHK(hipEventCreateWithFlags(&c->ev_stage[i], hipEventDisableTiming));
        }
    }
    // is the caller's buffer page-locked (then the DMA engine reads it directly)?
    bool src_pinned = false;
    {
        hipPointerAttribute_t at;
        if (n && hipPointerGetAttributes(&at, tokens) == hipSuccess) src_pinned = at.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
    }
    std::vector<hipEvent_t> ev_in(n_ranges, nullptr), ev_out(2, nullptr);
    for (auto& e : ev_in) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : ev_out) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t ev_copy = nullptr;
    HIPCHK(hipEventCreateWithFlags(&ev_copy, hipEventDisableTiming));
    auto drop_events = [&]() {
        for (auto e : ev_in) (void)hipEventDestroy(e);
        for (auto e : ev_out) (void)hipEventDestroy(e);
        (void)hipEventDestroy(ev_copy);
    };
    std::atomic<int> h2d_rc{TK_OK};
    std::atomic<uint64_t> sent{0};
    const int dev = c->device;
    unsigned nth = std::thread::hardware_concurrency();
    nth = copy_threads(nth);
    std::thread producer([&]() {
        (void)hipSetDevice(dev);
        for (uint64_t k = 0; k < n_ranges; ++k) {
            const uint64_t a = k * RANGE, cnt = a + RANGE < n ? RANGE : n - a;
            const void* src = tokens + a;
            if (!src_pinned) {
                const int slot = (int)(k & 1);
                if (k >= 2 && hipEventSynchronize(c->ev_stage[slot]) != hipSuccess) h2d_rc = TK_RUNTIME_ERROR;
                parallel_memcpy(c->stage[slot], tokens + a, cnt * 4, nth);
                src = c->stage[slot];
                if (hipMemcpyAsync(c->d_tok.as<uint32_t>() + a, src, cnt * 4, hipMemcpyHostToDevice, c->cs_h2d) != hipSuccess) h2d_rc = TK_RUNTIME_ERROR;
                (void)hipEventRecord(c->ev_stage[slot], c->cs_h2d);
            } else if (hipMemcpyAsync(c->d_tok.as<uint32_t>() + a, src, cnt * 4, hipMemcpyHostToDevice, c->cs_h2d) != hipSuccess) {
                h2d_rc = TK_RUNTIME_ERROR;
            }
            (void)hipEventRecord(ev_in[k], c->cs_h2d);
            sent.store(k + 1, std::memory_order_release);
        }
    });
    uint8_t* host = nullptr;
    uint64_t host_cap = 0, base = 0;
    Buf* dout[2] = {&c->d_bytes, &c->d_bytes_alt};
    int rc = TK_OK;
    uint64_t bad_pos = ~0ull;
    {
        unsigned long long init[2] = {0, ~0ull};
        std::vector<unsigned long long> initv(2 * (n_ranges + 1));
        for (size_t i = 0; i < initv.size(); i += 2) initv[i] = init[0], initv[i + 1] = init[1];
        hipError_t e = hipMemcpyAsync(tots, initv.data(), initv.size() * 8, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);  // (initv leaves scope)
        if (e != hipSuccess) rc = fail(TK_RUNTIME_ERROR, std::string("HIP error: ") + hipGetErrorString(e));
    }
    const double t_call = now_us();
    auto step = [&](uint64_t k) -> int {
        const uint64_t a = k * RANGE, cnt = a + RANGE < n ? RANGE : n - a;
        const double t_0 = now_us();
        while (sent.load(std::memory_order_acquire) <= k) std::this_thread::yield();
        const double t_1 = now_us();
        if (h2d_rc.load() != TK_OK) return fail(TK_RUNTIME_ERROR, "host-to-device copy failed");
        HIPCHK(hipStreamWaitEvent(s, ev_in[k], 0));
        unsigned long long* tot = tots + 2 * (k + 1);
        TRY(decode_range_len(c, s, c->d_tok.as<uint32_t>(), a, cnt, tot));
        unsigned long long h_tot[2] = {0, ~0ull};
        HIPCHK(hipMemcpyAsync(h_tot, tot, 16, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (h_tot[1] != ~0ull) {
            bad_pos = h_tot[1];
            return TK_KEY_ERROR;
        }
        const uint64_t nbk = h_tot[0];
        const double t_2 = now_us();
        Buf& d = *dout[k & 1];
        if (k >= 2) HIPCHK(hipEventSynchronize(ev_out[k & 1]));  // the buffer's previous bytes have left
        TRY(ensure(d, nbk + 16));
        TRY(decode_range_copy(c, s, c->d_tok.as<uint32_t>(), a, cnt, d.as<uint8_t>(), d_tboff, base));
        HIPCHK(hipEventRecord(ev_copy, s));
        // the result buffer: sized by the density so far
        const uint64_t need = base + nbk, done = a + cnt;
        if (need > host_cap || !host) {
            uint64_t est = done == n ? need : (uint64_t)((double)need / (double)done * (double)n * 1.06) + 4096;
            if (est < need) est = need;
            uint8_t* nh = (uint8_t*)(est >= (1u << 20) || n_ranges > 1 ? pinned_get(est ? est : 1) : malloc(est ? est : 1));
            if (!nh) return fail(TK_RUNTIME_ERROR, "out of host memory");
            if (host) {
                HIPCHK(hipStreamSynchronize(c->cs_d2h));
                if (base) parallel_memcpy(nh, host, base, nth);
                tk_free(host);
            }
            host = nh;
            host_cap = est;
        }
        const double t_3 = now_us();
        if (c->dbg & 64)
            fprintf(stderr, "decode range %llu: at %.0f us; ids waited %.0f us, lengths %.0f us, buffers %.0f us\n", (unsigned long long)k, t_0 - t_call, t_1 - t_0,
                    t_2 - t_1, t_3 - t_2);
        HIPCHK(hipStreamWaitEvent(c->cs_d2h, ev_copy, 0));
        if (nbk) HIPCHK(hipMemcpyAsync(host + base, d.p, nbk, hipMemcpyDeviceToHost, c->cs_d2h));
        HIPCHK(hipEventRecord(ev_out[k & 1], c->cs_d2h));
        base = need;
        return TK_OK;
    };
    for (uint64_t k = 0; k < n_ranges && rc == TK_OK; ++k) rc = step(k);
    producer.join();
    hipError_t e = hipStreamSynchronize(c->cs_h2d);
    if (rc == TK_OK && byte_off_out) {
        uint64_t* d_tok_off = c->d_boff.as<uint64_t>();
        uint64_t* d_byte_off = d_tok_off + n_docs + 1;
        unsigned long long h_total = base;
        if (e == hipSuccess) e = hipMemcpyAsync(tots, &h_total, 8, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_tok_off, tok_off, (n_docs + 1) * 8, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(tk_k_dec_docoff, dim3(grid_for(n_docs + 1, 256, 4096)), dim3(256), 0, s, d_tok_off, n_docs, n, d_tboff, tots, d_byte_off);
            e = hipMemcpyAsync(byte_off_out, d_byte_off, (n_docs + 1) * 8, hipMemcpyDeviceToHost, s);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipStreamSynchronize(c->cs_d2h);
    if (e == hipSuccess) e = hipGetLastError();
    drop_events();
    if (c->dbg & 64) fprintf(stderr, "decode: %llu ranges, ids %s, %.0f us\n", (unsigned long long)n_ranges, src_pinned ? "page-locked" : "staged", now_us() - t_call);
    if (rc == TK_KEY_ERROR && bad_pos != ~0ull) rc = fail(TK_KEY_ERROR, "Invalid token for decoding: " + std::to_string(tokens[bad_pos]));
    if (rc == TK_OK && e != hipSuccess) rc = fail(TK_RUNTIME_ERROR, std::string("HIP error: ") + hipGetErrorString(e));
    if (rc == TK_OK) rc = drain_events(c);  // (before the outputs are published: a caller that gets an error owns nothing)
    else (void)drain_events(c);
    if (rc != TK_OK) {
        if (host) tk_free(host);
        return rc;
    }
    if (!host) host = (uint8_t*)malloc(1);
    *bytes_out = host;
    *n_bytes_out = base;
    return TK_OK;
}

// ------------------------------------------------------------------------------------------
// Several GPUs of one node, one process: the documents of a batch are split into contiguous ranges of about equal
// byte counts (documents never interact: tiktoken/core.py:174-176 maps a pure function), every core encodes its range on its own
// device from its own host thread, and the token ids are gathered in document order -- on the host, or on the first core's device
// through peer copies over xGMI.  No collective on the data path.
// ------------------------------------------------------------------------------------------
// RCCL, loaded on demand (a 570 MB library nobody pays for who uses one GPU): the gather of the several-GPU entry runs on it when the
// group's devices are pairwise distinct -- grouped ncclSend / ncclRecv of the shards' id buffers to the first core's device
// (rccl.h:700-722; a gather with per-rank counts).  Virtual ranks (a device named twice) and a missing library use peer copies.
struct TkRccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool tried = false, ok = false;
    bool load() {  // (the answer of the first call, whatever it was: a library without one of the symbols is not asked again)
        if (tried) return ok;
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Send = (decltype(Send))dlsym(lib, "ncclSend");
        Recv = (decltype(Recv))dlsym(lib, "ncclRecv");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv;
        if (!ok) {
            dlclose(lib);
            lib = nullptr;
        }
        return ok;
    }
};
static TkRccl g_rccl;
static std::mutex g_rccl_mu;

struct tk_group {
    std::vector<tk_core*> cores;
    std::mutex mu;
    Buf root_tokens, root_off, root_raw;  // gathered results on cores[0]'s device; root_raw: the shards' own offsets before rebasing
    std::vector<hipStream_t> gs;          // one copy stream per core, on the core's device
    std::vector<hipEvent_t> ge;
    std::vector<ncclComm_t> comms;        // RCCL communicators (one per core), empty: peer copies
    bool rccl_tried = false;
    uint64_t gathers_rccl = 0, gathers_peer = 0;
};

extern "C" int tk_group_create(tk_core** cores, uint32_t n, tk_group** out) {
    if (!cores || !out || n == 0) return fail(TK_VALUE_ERROR, "tk_group_create needs at least one core");
    for (uint32_t i = 0; i < n; ++i)
        if (!cores[i]) return fail(TK_VALUE_ERROR, "null core");
    tk_group* g = new tk_group();
    g->cores.assign(cores, cores + n);
    *out = g;
    return TK_OK;
}
extern "C" void tk_group_destroy(tk_group* g) {
    if (!g) return;
    for (size_t r = 0; r < g->gs.size(); ++r) {
        (void)hipSetDevice(g->cores[r]->device);
        if (g->gs[r]) (void)hipStreamDestroy(g->gs[r]);
        if (g->ge[r]) (void)hipEventDestroy(g->ge[r]);
    }
    for (ncclComm_t cm : g->comms)
        if (cm) (void)g_rccl.CommDestroy(cm);
    if (!g->cores.empty()) (void)hipSetDevice(g->cores[0]->device);
    release(g->root_tokens);
    release(g->root_off);
    release(g->root_raw);
    delete g;
}
extern "C" uint32_t tk_group_size(tk_group* g) { return g ? (uint32_t)g->cores.size() : 0; }
// 1: the device gather of the last tk_group_encode_batch_device ran on RCCL, 0: on peer copies
extern "C" uint64_t tk_group_stat(tk_group* g, const char* name) {
    if (!g || !name) return 0;
    if (!strcmp(name, "gathers_rccl")) return g->gathers_rccl;
    if (!strcmp(name, "gathers_peer")) return g->gathers_peer;
    return 0;
}

// document ranges [first[r], first[r + 1]) of about equal byte counts (contiguous: the order of the results is the order of the input)
static std::vector<uint64_t> partition_by_bytes(const uint64_t* doc_off, uint64_t n_docs, uint32_t parts) {
    std::vector<uint64_t> first(parts + 1, n_docs);
    first[0] = 0;
    const uint64_t total = doc_off[n_docs];
    uint64_t d = 0;
    for (uint32_t r = 1; r < parts; ++r) {
        const uint64_t want = total / parts * r + (total % parts) * r / parts;
        while (d < n_docs && doc_off[d] < want) ++d;
        first[r] = d;
    }
    return first;
}

struct ShardResult {
    int rc = TK_OK;
    std::string err;
    uint32_t* tokens = nullptr;
    uint64_t n_tokens = 0;
    std::vector<uint64_t> tok_off;
};

// every core encodes its document range from its own host thread; on_device: the ids stay in each core's out_tokens / out_tok_off
static int group_encode(tk_group* g, const uint8_t* utf8, const uint64_t* doc_off, uint64_t n_docs, int use_special, const uint32_t* allowed_ids,
                        uint64_t n_allowed, std::vector<ShardResult>& res, std::vector<uint64_t>& first, bool on_device) {
    if (!g) return fail(TK_VALUE_ERROR, "group is null");
    if (!doc_off) return fail(TK_VALUE_ERROR, "null argument");
    if (doc_off[0] != 0) return fail(TK_VALUE_ERROR, "doc_off[0] must be 0");
    for (uint64_t d = 0; d < n_docs; ++d)
        if (doc_off[d + 1] < doc_off[d]) return fail(TK_VALUE_ERROR, "doc_off must be non-decreasing");
    const uint32_t R = (uint32_t)g->cores.size();
    first = partition_by_bytes(doc_off, n_docs, R);
    res.assign(R, ShardResult());
    std::vector<std::thread> th;
    for (uint32_t r = 0; r < R; ++r) {
        th.emplace_back([&, r]() {
            const uint64_t d0 = first[r], nd = first[r + 1] - d0;
            std::vector<uint64_t> off(nd + 1);
            for (uint64_t k = 0; k <= nd; ++k) off[k] = doc_off[d0 + k] - doc_off[d0];
            ShardResult& o = res[r];
            if (!on_device) o.tok_off.assign(nd + 1, 0);
            o.rc = encode_batch_impl(g->cores[r], utf8 + doc_off[d0], off.data(), nd, use_special, allowed_ids, n_allowed, &o.tokens, &o.n_tokens,
                                     on_device ? nullptr : o.tok_off.data(), on_device, false);
            if (o.rc != TK_OK) o.err = tk_last_error();  // (thread-local message of this worker)
        });
    }
    for (auto& t : th) t.join();
    for (uint32_t r = 0; r < R; ++r)
        if (res[r].rc != TK_OK) {
            const int rc = res[r].rc;
            const std::string msg = "device " + std::to_string(g->cores[r]->device) + ": " + res[r].err;
            for (auto& o : res) tk_free(o.tokens);
            return fail(rc, msg);
        }
    return TK_OK;
}

// Encoding.encode_ordinary_batch / encode_batch over several GPUs; same contract as tk_encode_batch.
extern "C" int tk_group_encode_batch(tk_group* g, const uint8_t* utf8, const uint64_t* doc_off, uint64_t n_docs, int use_special,
                                     const uint32_t* allowed_ids, uint64_t n_allowed, uint32_t** tokens_out, uint64_t* n_tokens_out,
                                     uint64_t* tok_off_out) {
    if (!tokens_out || !n_tokens_out) return fail(TK_VALUE_ERROR, "null argument");
    if (!g) return fail(TK_VALUE_ERROR, "group is null");
    std::lock_guard<std::mutex> lk(g->mu);
    std::vector<ShardResult> res;
    std::vector<uint64_t> first;
    TRY(group_encode(g, utf8, doc_off, n_docs, use_special, allowed_ids, n_allowed, res, first, false));
    uint64_t total = 0;
    std::vector<uint64_t> base(res.size() + 1, 0);
    for (size_t r = 0; r < res.size(); ++r) {
        base[r] = total;
        total += res[r].n_tokens;
    }
    uint32_t* host = (uint32_t*)malloc((total ? total : 1) * 4);
    if (!host) {
        for (auto& o : res) tk_free(o.tokens);
        return fail(TK_RUNTIME_ERROR, "out of host memory");
    }
    std::vector<std::thread> th;
    for (size_t r = 0; r < res.size(); ++r)
        th.emplace_back([&, r]() {
            if (res[r].n_tokens) memcpy(host + base[r], res[r].tokens, res[r].n_tokens * 4);
            if (tok_off_out)
                for (uint64_t k = 0; k + 1 < res[r].tok_off.size() || (r + 1 == res.size() && k < res[r].tok_off.size()); ++k)
                    tok_off_out[first[r] + k] = base[r] + res[r].tok_off[k];
            tk_free(res[r].tokens);
        });
    for (auto& t : th) t.join();
    *tokens_out = host;
    *n_tokens_out = total;
    return TK_OK;
}

// raw[0 .. n]: a shard's own token offsets; out[k] = raw[k] + base for its documents (k < n), and out[n] as well when `last`
__global__ __launch_bounds__(256) void tk_k_group_rebase(const uint64_t* __restrict__ raw, uint64_t n, uint64_t base, uint64_t* __restrict__ out, int last) {
    const uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (k < n || (last && k == n)) out[k] = raw[k] + base;
}

// The same with the results gathered on the FIRST core's device and nothing but the text crossing PCIe: every core encodes its shard with
// the ids left on its own device, then all shards travel to the first device at the same time -- RCCL send / recv over xGMI when the
// devices are distinct, concurrent peer copies (one stream per source device) otherwise -- and their document offsets are rebased
// there.  *d_tokens_out / *d_tok_off_out are owned by the group (valid until its next call).
extern "C" int tk_group_encode_batch_device(tk_group* g, const uint8_t* utf8, const uint64_t* doc_off, uint64_t n_docs, int use_special,
                                            const uint32_t* allowed_ids, uint64_t n_allowed, const uint32_t** d_tokens_out,
                                            uint64_t* n_tokens_out, const uint64_t** d_tok_off_out) {
    if (!d_tokens_out || !n_tokens_out) return fail(TK_VALUE_ERROR, "null argument");
    if (!g) return fail(TK_VALUE_ERROR, "group is null");
    std::lock_guard<std::mutex> lk(g->mu);
    std::vector<ShardResult> res;
    std::vector<uint64_t> first;
    TRY(group_encode(g, utf8, doc_off, n_docs, use_special, allowed_ids, n_allowed, res, first, true));
    const size_t R = res.size();
    uint64_t total = 0;
    std::vector<uint64_t> base(R + 1, 0);
    for (size_t r = 0; r < R; ++r) {
        base[r] = total;
        total += res[r].n_tokens;
    }
    tk_core* root = g->cores[0];
    if (g->gs.empty()) {
        g->gs.assign(R, nullptr);
        g->ge.assign(R, nullptr);
        for (size_t r = 0; r < R; ++r) {
            HIPCHK(hipSetDevice(g->cores[r]->device));
            HIPCHK(hipStreamCreateWithFlags(&g->gs[r], hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&g->ge[r], hipEventDisableTiming));
        }
    }
    if (!g->rccl_tried) {  // communicators once per group, when no device is named twice
        g->rccl_tried = true;
        std::vector<int> devs;
        bool distinct = R > 1 && !getenv("TIKTOKEN_AMD_NO_RCCL");
        for (size_t r = 0; r < R; ++r) {
            for (int d : devs) distinct = distinct && d != g->cores[r]->device;
            devs.push_back(g->cores[r]->device);
        }
        if (distinct) {
            std::lock_guard<std::mutex> lr(g_rccl_mu);
            if (g_rccl.load()) {
                g->comms.assign(R, nullptr);
                if (g_rccl.CommInitAll(g->comms.data(), (int)R, devs.data()) != ncclSuccess) g->comms.clear();
            }
        }
    }
    HIPCHK(hipSetDevice(root->device));
    TRY(ensure(g->root_tokens, (total + 1) * 4));
    TRY(ensure(g->root_off, (n_docs + 2) * 8));
    TRY(ensure(g->root_raw, (n_docs + R + 2) * 8));
    uint32_t* rt = g->root_tokens.as<uint32_t>();
    uint64_t* raw = g->root_raw.as<uint64_t>();
    bool by_rccl = !g->comms.empty();
    if (by_rccl) {
        ncclResult_t nr = g_rccl.GroupStart();
        for (size_t r = 1; r < R && nr == ncclSuccess; ++r) {
            if (!res[r].n_tokens) continue;
            (void)hipSetDevice(g->cores[r]->device);
            nr = g_rccl.Send(g->cores[r]->out_tokens.p, res[r].n_tokens, ncclUint32, 0, g->comms[r], g->gs[r]);
            (void)hipSetDevice(root->device);
            if (nr == ncclSuccess) nr = g_rccl.Recv(rt + base[r], res[r].n_tokens, ncclUint32, (int)r, g->comms[0], g->gs[0]);
        }
        const ncclResult_t ne = g_rccl.GroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess)
            return fail(TK_RUNTIME_ERROR, std::string("RCCL gather failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(nr) : "?"));
        HIPCHK(hipSetDevice(root->device));
        if (res[0].n_tokens) HIPCHK(hipMemcpyAsync(rt, root->out_tokens.p, res[0].n_tokens * 4, hipMemcpyDeviceToDevice, g->gs[0]));
        ++g->gathers_rccl;
    }
    for (size_t r = 0; r < R; ++r) {  // offsets (and, without RCCL, the ids): one peer copy per shard, each on its source device's stream
        tk_core* c = g->cores[r];
        const uint64_t nd = first[r + 1] - first[r];
        HIPCHK(hipSetDevice(c->device));
        if (!by_rccl && res[r].n_tokens)
            HIPCHK(hipMemcpyPeerAsync(rt + base[r], root->device, c->out_tokens.p, c->device, res[r].n_tokens * 4, g->gs[r]));
        HIPCHK(hipMemcpyPeerAsync(raw + first[r] + r, root->device, c->out_tok_off.p, c->device, (nd + 1) * 8, g->gs[r]));
        HIPCHK(hipEventRecord(g->ge[r], g->gs[r]));
    }
    if (!by_rccl) ++g->gathers_peer;
    HIPCHK(hipSetDevice(root->device));
    hipStream_t s0 = g->gs[0];
    for (size_t r = 1; r < R; ++r) HIPCHK(hipStreamWaitEvent(s0, g->ge[r], 0));
    for (size_t r = 0; r < R; ++r) {
        const uint64_t nd = first[r + 1] - first[r];
        hipLaunchKernelGGL(tk_k_group_rebase, dim3((uint32_t)((nd + 1 + 255) / 256)), dim3(256), 0, s0, raw + first[r] + r, nd, base[r],
                           g->root_off.as<uint64_t>() + first[r], r + 1 == R ? 1 : 0);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s0));
    *d_tokens_out = rt;
    if (d_tok_off_out) *d_tok_off_out = g->root_off.as<uint64_t>();
    *n_tokens_out = total;
    return TK_OK;
}

extern "C" uint64_t tk_n_tokens(tk_core* c) { return c ? c->H.n_ranks : 0; }

extern "C" int tk_sorted_token(tk_core* c, uint64_t i, const uint8_t** bytes_out, uint64_t* len_out, uint32_t* rank_out) {
    if (!c || i >= c->H.sorted_ranks().size()) return fail(TK_VALUE_ERROR, "index out of range");
    uint32_t r = c->H.sorted_ranks()[i];
    if (rank_out) *rank_out = r;
    return tk_decode_single_token_bytes(c, r, bytes_out, len_out);
}

// token_byte_values() in one call: all token byte strings in lexicographic order, packed (src/lib.rs:648-650, py.rs:178-183).
// The arrays are owned by the core (built on first use) and stay valid until tk_destroy.
extern "C" int tk_sorted_tokens_packed(tk_core* c, const uint8_t** blob_out, const uint64_t** off_out, uint64_t* n_out) {
    if (!c || !blob_out || !off_out || !n_out) return fail(TK_VALUE_ERROR, "null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->sorted_off.empty()) {
        const TkHostTables& H = c->H;
        c->sorted_off.reserve(H.sorted_ranks().size() + 1);
        c->sorted_off.push_back(0);
        for (uint32_t r : H.sorted_ranks()) {
            const auto& e = *H.find_token(r);
            c->sorted_blob.insert(c->sorted_blob.end(), H.tok_bytes.begin() + e.first, H.tok_bytes.begin() + e.first + e.second);
            c->sorted_off.push_back(c->sorted_blob.size());
        }
        if (c->sorted_blob.empty()) c->sorted_blob.push_back(0);
    }
    *blob_out = c->sorted_blob.data();
    *off_out = c->sorted_off.data();
    *n_out = c->sorted_off.size() - 1;
    return TK_OK;
}

// Vocabulary wire format: the contents of a `.tiktoken` file -> packed arrays in the layout tk_create takes (release each with tk_free).
extern "C" int tk_parse_tiktoken_bpe(const uint8_t* text, uint64_t len, uint8_t** blob_out, uint64_t** off_out, uint32_t** ids_out,
                                     uint64_t* n_out) {
    if (!blob_out || !off_out || !ids_out || !n_out || (!text && len)) return fail(TK_VALUE_ERROR, "null argument");
    std::vector<uint8_t> blob;
    std::vector<uint64_t> off;
    std::vector<uint32_t> ids;
    const std::string err = tk_parse_tiktoken(text, len, &blob, &off, &ids);
    if (!err.empty()) return fail(TK_VALUE_ERROR, err);
    uint8_t* b = (uint8_t*)malloc(blob.size() ? blob.size() : 1);
    uint64_t* o = (uint64_t*)malloc(off.size() * 8);
    uint32_t* r = (uint32_t*)malloc(ids.size() ? ids.size() * 4 : 4);
    if (!b || !o || !r) {
        free(b);
        free(o);
        free(r);
        return fail(TK_RUNTIME_ERROR, "out of host memory");
    }
    if (!blob.empty()) memcpy(b, blob.data(), blob.size());
    memcpy(o, off.data(), off.size() * 8);
    if (!ids.empty()) memcpy(r, ids.data(), ids.size() * 4);
    *blob_out = b;
    *off_out = o;
    *ids_out = r;
    *n_out = ids.size();
    return TK_OK;
}

extern "C" void tk_free(void* p) {
    if (!pinned_release(p)) free(p);
}

// Well-formed UTF-8 (Unicode 15, table 3-7): what the reference gets for free from &str.  Eight ASCII bytes per step where there are any.
extern "C" int tk_validate_utf8(const uint8_t* s, uint64_t n, uint64_t* bad_pos) {
    if (!s && n) return fail(TK_VALUE_ERROR, "null argument");
    uint64_t i = 0;
    auto bad = [&](uint64_t at) {
        if (bad_pos) *bad_pos = at;
        return fail(TK_VALUE_ERROR, "invalid UTF-8 at byte " + std::to_string(at));
    };
    while (i < n) {
        if (i + 8 <= n) {
            uint64_t w;
            memcpy(&w, s + i, 8);
            if (!(w & 0x8080808080808080ull)) {
                i += 8;
                continue;
            }
        }
        const uint8_t b = s[i];
        if (b < 0x80) {
            ++i;
            continue;
        }
        uint32_t need;
        uint8_t lo = 0x80, hi = 0xBF;  // bounds of the second byte
        if (b >= 0xC2 && b <= 0xDF) need = 1;
        else if (b >= 0xE0 && b <= 0xEF) {
            need = 2;
            if (b == 0xE0) lo = 0xA0;       // no overlong three-byte forms
            else if (b == 0xED) hi = 0x9F;  // no surrogates
        } else if (b >= 0xF0 && b <= 0xF4) {
            need = 3;
            if (b == 0xF0) lo = 0x90;       // no overlong four-byte forms
            else if (b == 0xF4) hi = 0x8F;  // nothing above U+10FFFF
        } else return bad(i);               // a continuation byte, 0xC0, 0xC1, 0xF5..0xFF
        if (i + need >= n) return bad(i);  // truncated
        if (s[i + 1] < lo || s[i + 1] > hi) return bad(i);
        for (uint32_t k = 2; k <= need; ++k)
            if ((s[i + k] & 0xC0) != 0x80) return bad(i);
        i += need + 1;
    }
    return TK_OK;
}

extern "C" int tk_set_output_buffers(tk_core* c, uint32_t n) {
    if (!c) return fail(TK_VALUE_ERROR, "core is null");
    if (n != 1 && n != 2) return fail(TK_VALUE_ERROR, "tk_set_output_buffers: 1 or 2");
    std::lock_guard<std::mutex> lk(c->mu);
    c->out_bufs = n;
    return TK_OK;
}

extern "C" void tk_set_profiling(tk_core* c, int enabled) {
    if (c) c->profiling = enabled != 0;
}
extern "C" void tk_reset_kernel_ms(tk_core* c) {
    if (c) c->stats.clear();
}
extern "C" int tk_get_kernel_ms(tk_core* c, const char* name, double* ms_out, uint64_t* launches_out) {
    if (!c) return TK_VALUE_ERROR;
    auto it = c->stats.find(name);
    if (it == c->stats.end()) {
        if (ms_out) *ms_out = 0;
        if (launches_out) *launches_out = 0;
        return TK_KEY_ERROR;
    }
    if (ms_out) *ms_out = it->second.ms;
    if (launches_out) *launches_out = it->second.launches;
    return TK_OK;
}
extern "C" void tk_last_stats(tk_core* c, uint64_t* n_bytes, uint64_t* n_pieces, uint64_t* n_tokens, uint64_t* n_docs,
                              uint64_t* n_medium, uint64_t* n_long) {
    if (!c) return;
    if (n_bytes) *n_bytes = c->st_bytes;
    if (n_pieces) *n_pieces = c->st_pieces;
    if (n_tokens) *n_tokens = c->st_tokens;
    if (n_docs) *n_docs = c->st_docs;
    if (n_medium) *n_medium = c->st_medium;
    if (n_long) *n_long = c->st_long;
}
extern "C" uint64_t tk_stat(tk_core* c, const char* name) {
    if (!c || !name) return 0;
    const std::string k(name);
    if (k == "front_wgs_per_cu") return c->front_wgs;
    if (k == "compute_units") return c->n_cu;
    if (k == "chunks") return c->st_chunks;
    if (k == "small_launches") return c->st_small_launches;  // launches of tk_k_small and the calls they carried (several callers share a launch)
    if (k == "small_calls") return c->st_small_calls;
    if (k == "mid_calls") return c->st_mid_calls;  // documents of 2 .. 128 KiB encoded as segments in one launch
    if (k == "back_streams") return (uint64_t)c->n_back;  // streams found to run beside the front stream (0: no multi-chunk batch yet)
    if (k == "resynced") return c->st_resynced;  // batches repeated because a deferred tile gave up while the host was not waiting (stage_deferred)
    if (k == "regrown") return c->st_regrown;  // batches repeated with a larger miss data since the core was made (encode_device_locked)
    if (k == "workspace_bytes") {             // device memory of the work sets (everything but the text, the tables and the outputs)
        uint64_t t = 0;
        for (auto& w : c->ws)
            for (Buf* b : w.all()) t += b->cap;
        return t;
    }
    // (experiments) the deferred tiles of the last chunk of work set 0: "deferred_count", "deferred_tile_<i>" (read from the device: the caller has waited for the call)
    if (k == "deferred_count") return c->ws[0].h_counters ? c->ws[0].h_counters[TK_CNT_N + TK_CNT_DEFER] : 0;
    if (k.rfind("deferred_tile_", 0) == 0) {
        uint32_t v = 0;
        const uint64_t i = strtoull(k.c_str() + 14, nullptr, 10);
        if (c->ws[0].deferred.p && (i + 1) * 4 <= c->ws[0].deferred.cap && hipMemcpy(&v, c->ws[0].deferred.as<uint32_t>() + i, 4, hipMemcpyDeviceToHost) != hipSuccess) v = 0xFFFFFFFFu;
        return v;
    }
    if (k == "fallbacks") return c->st_fallbacks;  // chunks re-split by the generic engine since the core was made (stage_deferred)
    if (k == "host_front_us") return (uint64_t)c->host_us[0];
    if (k == "host_back_us") return (uint64_t)c->host_us[1];
    if (k == "host_back_wait_us") return (uint64_t)c->host_us[2];
    if (k == "host_finish_us") return (uint64_t)c->host_us[3];
    if (k == "host_tail_us") return (uint64_t)c->host_us[4];
    if (k == "host_total_us") return (uint64_t)c->host_us[5];
    if (k == "chunk_bytes") return c->chunk_bytes;
#ifdef TKF_TIMING
    if (k.rfind("time_", 0) == 0) {  // (experiments: tk_fused.h, TKT)
        static unsigned long long acc[2 * 1024 * 16];
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        if (k == "time_reset") {
            memset(acc, 0, sizeof(acc));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(tk_time_acc), acc, sizeof(acc));
            return 0;
        }
        (void)hipMemcpyFromSymbol(acc, HIP_SYMBOL(tk_time_acc), sizeof(acc));
        const bool starts = k.rfind("time_s", 0) == 0;  // ("time_s<i>": the deferred-tile instance)
        const int i = atoi(k.c_str() + (starts ? 6 : 5));
        unsigned long long sum = 0;
        for (int b = 0; b < 1024 && i >= 0 && i < 16; ++b) sum += acc[(starts ? 16384 : 0) + b * 16 + i];
        return sum;
    }
#endif
    return 0;
}
