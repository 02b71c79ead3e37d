// Host runtime + C ABI of the MI355X Stage-1 engine (include/mi355sw.h).
//
// Replaces the host side of the reference's CUDA extension:
//   X/CUDAligner.cpp   (setSequences :229-265, allocate/free :611-671, processDiagonal :202-209,
//                       getSpecialRow/LastRow/LastColumn/BlockScores :355-452, setFirstRow/Column :461-504)
//   X/cuda_util.cpp    (device selection :191-287)
//   M/libmasa/aligners/AbstractDiagonalAligner.cpp (call order of the IManager hooks, :59-159, :286-456)
// with one persistent strip kernel per partition instead of a launch pair + sync per external
// diagonal (X/CUDAligner.cu:1261-1277), and streamed border columns instead of per-diagonal copies.
//
// There is NO CPU fallback: without a gfx950 device mi355sw_create() fails with MI355SW_ENOGPU.
#include "../../include/mi355sw.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "sw_kernel.h"

using namespace mi355sw;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

// host-pinned, device-mapped buffer: the running strip kernel reads/writes it directly over PCIe, so
// streaming the border columns needs NO copy queued behind (or beside) the persistent kernel.
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
};

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct mi355sw_handle {
    mi355sw_config cfg{};
    int device = 0;
    int compute_units = 256;
    hipStream_t stream = nullptr;   // strip kernel
    hipStream_t copy = nullptr;     // border traffic while the kernel runs
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;

    // sequences (X/CUDAligner.cpp:229-265)
    bool have_seq = false;
    int len0 = 0, len1 = 0;
    DevBuf d_seq0, d_seq1;
    bool profile = false;           // coded sequences with <= 7 codes: the int32 nibble-profile kernel applies
    bool packed_ok = false;         // coded sequences with <= 14 codes: the packed 16-bit kernel applies
    int n_match_codes = 0, pad_code = 0, seq0_shift = 0;

    // work buffers
    DevBuf d_bus, d_first_col, d_special, d_last_row, d_progress, d_strip_best, d_ctrl, d_kargs, d_trace, d_ckpt, d_seed;
    PinBuf p_first_col, p_last_col;  // streamed first column / last column (zero-copy)
    bool first_col_pinned = false;
    int* h_pinned = nullptr;        // [0] strips_done (kernel->host) [16] first_col_ready (host->kernel) [32] abort (host->kernel)
                                    // [48] error mirror (kernel->host, written before [0] moves past the failing strip)
                                    // [64] best-score hint (host->kernel, T domain) [80] running best (kernel->host, T domain)
    // column ports (xGMI boundary column): inbound = fine-grained HBM of this GPU, outbound = the next band's inbound
    // port mapped here (hipIpc / peer access).  Layout: 256 control bytes (int32 row counter at +0), then m+1 cells.
    void* in_port = nullptr; size_t in_port_bytes = 0; int in_port_rows = 0;
    void* out_port = nullptr; int out_port_rows = 0; bool out_port_ipc = false;
    bool first_col_port = false;    // active stream reads its first column from in_port
    int clean_rows = 0;             // rows reported by the last poll that saw no kernel error
    bool overflow_seen = false;
    std::vector<int4> strip_best_host;

    // stream state
    bool active = false;
    mi355sw_partition part{};
    mi355sw_stream_params sp{};
    int m = 0, n = 0, R = 8, SH = 512, strips = 0, waves = 0;
    bool use16 = false;             // packed 16-bit SW kernel selected for the active stream
    bool two_phase = false;         // value-only tracking in the main pass + exact re-run of the winning strip
    int ckpt_interval = 0, n_ckpt = 0;
    long long ckpt_pitch = 0;
    KernelArgs kargs{};             // arguments of the main pass (re-used by the exact pass)
    mi355sw_score final_best{};
    double exact_ms = 0;
    int special_interval_strips = 0, n_special = 0;
    long long special_pitch = 0;
    int fed_rows = 0;
    bool finished = false;
    mi355sw_stats stats{};
    std::atomic<long long> processed_total{0};
    int abort_strips = -1;          // host stop: strips complete or in flight at that moment (the rest were skipped)
    bool exact_records = false;     // every strip record must carry its cell (block-score pass): no two-phase tracking
    std::atomic<int> prog_strips{0}, prog_total{0};
};

#define FAIL(h, code, ...)                                   \
    do {                                                     \
        char _b[512];                                        \
        snprintf(_b, sizeof(_b), __VA_ARGS__);               \
        (h)->err = _b;                                       \
        return (code);                                       \
    } while (0)

#define HIPCHK(h, expr)                                                                      \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) FAIL(h, MI355SW_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

static int ensure(mi355sw_handle* h, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return MI355SW_OK;
    if (b.p) { (void) hipFree(b.p); b.p = nullptr; b.cap = 0; }
    size_t want = std::max<size_t>(bytes, 256);
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) FAIL(h, MI355SW_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    b.cap = want;
    return MI355SW_OK;
}

static int ensure_pinned(mi355sw_handle* h, PinBuf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return MI355SW_OK;
    if (b.p) { (void) hipHostFree(b.p); b.p = nullptr; b.cap = 0; }
    size_t want = std::max<size_t>(bytes, 4096);
    hipError_t e = hipHostMalloc(&b.p, want, hipHostMallocMapped);
    if (e != hipSuccess) FAIL(h, MI355SW_ENOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    b.cap = want;
    return MI355SW_OK;
}

static void release(DevBuf& b) {
    if (b.p) (void) hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

extern "C" {

int mi355sw_abi_version(void) { return MI355SW_ABI_VERSION; }

int mi355sw_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mi355sw_device_info(int32_t device, char* name, size_t name_len, int32_t* cus, int32_t* clock_mhz,
                        int64_t* hbm_bytes) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MI355SW_EHIP;
    if (name && name_len) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (cus) *cus = prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = prop.clockRate / 1000;
    if (hbm_bytes) *hbm_bytes = (int64_t) prop.totalGlobalMem;
    return MI355SW_OK;
}

int mi355sw_create(const mi355sw_config* config, mi355sw_handle** out) {
    if (!out) return MI355SW_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MI355SW_ENOGPU;
    mi355sw_handle* h = new mi355sw_handle();
    if (config) h->cfg = *config;
    else { h->cfg.device = -1; }
    int dev = h->cfg.device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= count) { delete h; return MI355SW_ENOGPU; }
    if (hipSetDevice(dev) != hipSuccess) { delete h; return MI355SW_EHIP; }
    h->device = dev;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { delete h; return MI355SW_EHIP; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        // the code object only carries gfx950 ISA (CDNA4); anything else cannot run it
        delete h;
        return MI355SW_ENOGPU;
    }
    h->compute_units = prop.multiProcessorCount;
    // The persistent strip kernel owns its hardware queue for seconds.  HIP multiplexes streams onto a
    // few HSA queues per priority level; a copy queued on a stream that shares the kernel's queue would
    // sit behind it forever.  The kernel stream therefore lives alone in the high-priority pool.
    int prio_lo = 0, prio_hi = 0;
    (void) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    // MI355SW_STREAM_PRIO=low|normal: experiments with two engines in one process (tools/concurrency_probe.py)
    if (const char* sp = getenv("MI355SW_STREAM_PRIO")) {
        if (!strcmp(sp, "low")) prio_hi = prio_lo;
        else if (!strcmp(sp, "normal")) prio_hi = (prio_lo + prio_hi) / 2;
    }
    if (hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipStreamCreateWithFlags(&h->copy, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
        hipHostMalloc((void**) &h->h_pinned, 512, hipHostMallocMapped) != hipSuccess) {
        mi355sw_destroy(h);
        return MI355SW_EHIP;
    }
    memset(h->h_pinned, 0, 512);
    h->h_pinned[64] = h->h_pinned[80] = -MI355SW_INF;
    *out = h;
    return MI355SW_OK;
}

void mi355sw_destroy(mi355sw_handle* h) {
    if (!h) return;
    (void) hipSetDevice(h->device);
    if (h->active) { mi355sw_stream_abort(h); mi355sw_stream_end(h, nullptr, nullptr); }
    mi355sw_port_close(h);
    release(h->d_seq0); release(h->d_seq1); release(h->d_bus); release(h->d_first_col);
    if (h->p_first_col.p) (void) hipHostFree(h->p_first_col.p);
    if (h->p_last_col.p) (void) hipHostFree(h->p_last_col.p);
    release(h->d_special); release(h->d_last_row); release(h->d_progress);
    release(h->d_strip_best); release(h->d_ctrl); release(h->d_kargs); release(h->d_ckpt); release(h->d_trace); release(h->d_seed);
    if (h->h_pinned) (void) hipHostFree(h->h_pinned);
    if (h->ev0) (void) hipEventDestroy(h->ev0);
    if (h->ev1) (void) hipEventDestroy(h->ev1);
    if (h->stream) (void) hipStreamDestroy(h->stream);
    if (h->copy) (void) hipStreamDestroy(h->copy);
    delete h;
}

const char* mi355sw_last_error(mi355sw_handle* h) { return h ? h->err.c_str() : "null handle"; }

int mi355sw_get_capabilities(mi355sw_handle* h, mi355sw_capabilities* c) {
    if (!h || !c) return MI355SW_EINVAL;
    memset(c, 0, sizeof(*c));
    // X/CUDAligner.cpp:87-111 : everything but special columns / variable penalties;
    // no texture limit on MI355X => no maximum sequence length (M/stage1/sw_stage1.cpp:362-375)
    c->dispatch_last_cell = 1; c->dispatch_last_row = 1; c->dispatch_last_column = 1;
    c->dispatch_special_row = 1; c->dispatch_special_column = 0;
    c->dispatch_scores = 1; c->dispatch_block_scores = 1; c->dispatch_best_score = 1;   // block scores: config.block_score_columns
    c->customize_first_row = 1; c->customize_first_column = 1;
    c->process_partition = 1; c->variable_penalties = 0; c->block_pruning = 1;
    c->needleman_wunsch = 1; c->smith_waterman = 1; c->fork_processes = 1;
    c->maximum_seq0_len = 0; c->maximum_seq1_len = 0;
    return MI355SW_OK;
}

int mi355sw_get_score_parameters(mi355sw_handle* h, mi355sw_score_params* p) {
    if (!p) return MI355SW_EINVAL;
    p->match = 1; p->mismatch = -3; p->gap_open = 3; p->gap_ext = 2;   // X/CUDAligner.hpp:77-98
    return MI355SW_OK;
}

int mi355sw_set_sequences(mi355sw_handle* h, const char* seq0, const char* seq1, int32_t len0, int32_t len1) {
    if (!h || !seq0 || !seq1 || len0 < 0 || len1 < 0) return MI355SW_EINVAL;
    if (h->active) FAIL(h, MI355SW_ESTATE, "set_sequences while a stream is active");
    HIPCHK(h, hipSetDevice(h->device));
    const unsigned char* s0 = (const unsigned char*) seq0;
    const unsigned char* s1 = (const unsigned char*) seq1;
    // byte histogram of both sequences: bytes present in BOTH can match each other (raw byte equality,
    // X/CUDAligner.cu:276-289), every other byte never matches anything
    long long cnt0[256] = {0}, cnt1[256] = {0};
    for (int k = 0; k < len0; k++) cnt0[s0[k]]++;
    for (int k = 0; k < len1; k++) cnt1[s1[k]]++;
    int common = 0;
    for (int b = 0; b < 256; b++) common += (cnt0[b] > 0 && cnt1[b] > 0);
    // Coded form ("profile"): common bytes become codes 0..K-1, most frequent first -- the packed kernel scores a
    // chunk with one byte permute when every column in reach carries a code < 4, so the four commonest letters
    // (A, C, G, T of a genome, whatever else it contains) must be the ones below 4.
    //   K <= 14: packed 16-bit kernel (one-hot bit 2+code in a 16-bit half)
    //   K <=  7: int32 nibble-profile kernel as its overflow fallback; 8..14: int32 byte-compare kernel on the codes
    h->packed_ok = (common <= 14) && !(h->cfg.flags & MI355SW_F_FORCE_GENERIC_COMPARE);
    h->profile = h->packed_ok && common <= 7;
    std::vector<unsigned char> c0((size_t) len0 + 64), c1((size_t) len1 + 64);
    if (h->packed_ok) {
        int order[256], k = 0;
        for (int b = 0; b < 256; b++) if (cnt0[b] > 0 && cnt1[b] > 0) order[k++] = b;
        std::stable_sort(order, order + k, [&](int x, int y) { return cnt0[x] + cnt1[x] > cnt0[y] + cnt1[y]; });
        // bytes of one sequence only: a code no byte of the other sequence can carry
        const int foreign0 = (k <= 7) ? 7 : 14, foreign1 = (k <= 7) ? 7 : 15;
        unsigned char lut0[256], lut1[256];
        for (int b = 0; b < 256; b++) { lut0[b] = (unsigned char) foreign0; lut1[b] = (unsigned char) foreign1; }
        for (int q = 0; q < k; q++) { lut0[order[q]] = (unsigned char) q; lut1[order[q]] = (unsigned char) q; }
        h->n_match_codes = k;
        h->pad_code = foreign0;
        h->seq0_shift = 2;
        for (int i = 0; i < len0; i++) c0[i] = lut0[s0[i]];
        for (int j = 0; j < len1; j++) c1[j] = (unsigned char) (lut1[s1[j]] * 4);   // v_bfe_i32 bit offset / code << 2
    } else {
        h->n_match_codes = 256;
        h->pad_code = 256;     // rows beyond m never equal any byte
        h->seq0_shift = 0;
        memcpy(c0.data(), s0, (size_t) len0);
        memcpy(c1.data(), s1, (size_t) len1);
    }
    int rc;
    if ((rc = ensure(h, h->d_seq0, c0.size())) || (rc = ensure(h, h->d_seq1, c1.size()))) return rc;
    HIPCHK(h, hipMemcpy(h->d_seq0.p, c0.data(), c0.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_seq1.p, c1.data(), c1.size(), hipMemcpyHostToDevice));
    h->len0 = len0;
    h->len1 = len1;
    h->have_seq = true;
    return MI355SW_OK;
}

int mi355sw_unset_sequences(mi355sw_handle* h) {
    if (!h) return MI355SW_EINVAL;
    if (h->active) FAIL(h, MI355SW_ESTATE, "unset_sequences while a stream is active");
    h->have_seq = false;
    return MI355SW_OK;
}

// ------------------------------------------------------------------------------------------------
// streaming form
// ------------------------------------------------------------------------------------------------
// Strip geometry (measured on MI355X, tools/gpu_perf.py sweeps, packed kernel, one wavefront per SIMD):
// one systolic step of a 64*R-row strip costs step_ns(R) (hand-off, LDS traffic and the per-chunk
// publish/poll are per step, the cell arithmetic per row), a strip follows its predecessor ~280 steps
// behind, and the W wavefronts sweep the partition in ceil(strips/W) rounds of n steps each -- a partly
// filled last round costs a full one.  Pick the height with the smallest estimate.
static double step_ns(int R) {
    switch (R) {                       // 2 rounds x 1 M columns, 1024 wavefronts: T / (2e6 + 280*1024)
    case 4: return 97; case 8: return 138; case 12: return 181; case 16: return 222;
    case 24: return 291; case 32: return 368;
    default: return 60.0 + 10.5 * R;
    }
}

static double estimate_ns(int m, int n, int R, int W) {
    const double step = step_ns(R);
    const long long strips = ((long long) m + 64 * R - 1) / (64 * R);
    const long long rounds = (strips + W - 1) / W;
    const long long hops = strips < W ? strips : W;
    return step * ((double) rounds * n + 280.0 * hops);
}

static int pick_rows_per_lane(const mi355sw_handle* h, int m, int n, bool packed, bool special_rows) {
    if (h->cfg.rows_per_lane == 4 || h->cfg.rows_per_lane == 8 || h->cfg.rows_per_lane == 12 ||
        h->cfg.rows_per_lane == 16 || h->cfg.rows_per_lane == 24 || h->cfg.rows_per_lane == 32)
        return h->cfg.rows_per_lane;
    // Special rows sit on multiples of the strip height.  With a height that divides 8192 they fall on CUDAlign's
    // own grid (multiples of MINIMUM_FLUSH_INTERVAL, AbstractDiagonalAligner.cpp:35), so an area written here can be
    // continued or read by a run of the reference and vice versa; 768- and 1536-row strips are only chosen when no
    // special rows are asked for (they are 1-2 % faster on some shapes).
    static const int cand16[] = {4, 8, 12, 16, 24, 32}, cand32[] = {4, 8, 16}, cand16p2[] = {4, 8, 16, 32};
    // block scores: the grid's block height is the strip height, and it must survive a rerun on the int32 kernels
    const bool common_heights = !packed || h->cfg.block_score_columns > 0;
    const int* cand = common_heights ? cand32 : (special_rows ? cand16p2 : cand16);
    const int nc = common_heights ? 3 : (special_rows ? 4 : 6);
    const int W = h->cfg.waves > 0 ? h->cfg.waves : h->compute_units * 4;
    int best = cand[0];
    double tb = estimate_ns(m, n, best, W);
    for (int k = 1; k < nc; k++) {
        const double t = estimate_ns(m, n, cand[k], W);
        if (t < tb) { tb = t; best = cand[k]; }
    }
    return best;
}

static int pick_waves(const mi355sw_handle* h, int m, int strips) {
    int waves = h->cfg.waves;
    if (waves <= 0) waves = h->compute_units * 4;     // one per SIMD (enforced by the kernels' register allocation)
    if (waves > strips) waves = strips;
    return waves < 1 ? 1 : waves;
}

int mi355sw_stream_begin(mi355sw_handle* h, const mi355sw_partition* part, const mi355sw_stream_params* p) {
    if (!h || !part || !p) return MI355SW_EINVAL;
    if (!h->have_seq) FAIL(h, MI355SW_ESTATE, "stream_begin before set_sequences");
    if (h->active) FAIL(h, MI355SW_ESTATE, "stream already active");
    if (part->i0 < 0 || part->j0 < 0 || part->i1 > h->len0 || part->j1 > h->len1 || part->i1 <= part->i0 ||
        part->j1 <= part->j0)
        FAIL(h, MI355SW_EINVAL, "bad partition (%d,%d)-(%d,%d) for sequences %d x %d", part->i0, part->j0,
             part->i1, part->j1, h->len0, h->len1);
    HIPCHK(h, hipSetDevice(h->device));
    h->part = *part;
    h->sp = *p;
    const int m = part->i1 - part->i0, n = part->j1 - part->j0;
    h->m = m; h->n = n;
    {
        const bool will16 = h->packed_ok && !p->force_int32 &&
                            !(h->cfg.flags & MI355SW_F_FORCE_INT32);
        h->R = pick_rows_per_lane(h, m, n, will16, p->special_row_interval > 0);
        // the int32 kernels are instantiated for R in {4,8,16}
        if (!will16 && h->R != 4 && h->R != 8 && h->R != 16) h->R = h->R > 16 ? 16 : 8;
    }
    h->SH = 64 * h->R;
    h->strips = (m + h->SH - 1) / h->SH;
    const int waves = pick_waves(h, m, h->strips);
    h->waves = waves;
    h->finished = false;
    h->fed_rows = 0;

    // special rows: AbstractDiagonalAligner::isSpecialRow (:466-478): every K-th strip boundary,
    // K = max(ceil(interval/bh), MINIMUM_FLUSH_INTERVAL/bh), never row 0 nor rows >= height
    h->special_interval_strips = 0;
    h->n_special = 0;
    if (p->special_row_interval > 0) {
        int K = (p->special_row_interval + h->SH - 1) / h->SH;
        if (K <= 0) K = 1;
        if (K < (8192 + h->SH - 1) / h->SH) K = (8192 + h->SH - 1) / h->SH;
        if (K < 1) K = 1;
        h->special_interval_strips = K;
        h->n_special = (int) (((long long) m - 1) / ((long long) K * h->SH));   // rows K*SH*k < m
    }
    h->special_pitch = ((long long) n + 63) / 64 * 64;
    h->use16 = h->packed_ok && !p->force_int32 &&
               !(h->cfg.flags & MI355SW_F_FORCE_INT32);
    // Two-phase best: the main pass keeps only each strip's best VALUE (no per-step position test, no
    // rare path on the start-up critical path of every strip); the canonical cell is then recomputed
    // for the first strip that holds the global maximum, from the nearest checkpoint row.
    // The exact pass costs one strip sweep (n steps of a short pipeline); it pays off once the main pass
    // is hundreds of sweeps long.  Smaller partitions keep exact tracking in the main pass, seeded with the
    // running global best so that it stays off the start-up path.
    h->two_phase = h->use16 && p->track_best && !h->exact_records && (m >= (32 << 20) || getenv("MI355SW_TWO_PHASE"));
    h->ckpt_interval = 0; h->n_ckpt = 0; h->ckpt_pitch = h->special_pitch;
    if (h->two_phase) {
        const int64_t budget = h->cfg.max_special_bytes > 0 ? h->cfg.max_special_bytes : (8LL << 30);   // checkpoints: 64 rows at most
        long long max_ck = budget / 2 / (long long) (sizeof(int2) * h->ckpt_pitch);
        if (max_ck > 64) max_ck = 64;
        if (max_ck < 1) max_ck = 1;
        int K = (int) ((h->strips + max_ck - 1) / max_ck);
        if (K < 1) K = 1;
        h->ckpt_interval = K;
        h->n_ckpt = (h->strips - 1) / K + 1;      // slots 0..(strips-1)/K
    }

    int rc;
    if ((rc = ensure(h, h->d_bus, sizeof(int2) * ((size_t) n + 64)))) return rc;
    if ((rc = ensure(h, h->d_progress, sizeof(int) * ((size_t) h->strips + 1)))) return rc;
    if ((rc = ensure(h, h->d_strip_best, sizeof(int4) * (size_t) h->strips))) return rc;
    if ((rc = ensure(h, h->d_ctrl, 256))) return rc;
    if ((rc = ensure(h, h->d_kargs, sizeof(KernelArgs)))) return rc;
    if (h->two_phase && (rc = ensure(h, h->d_ckpt, sizeof(int2) * (size_t) h->ckpt_pitch * h->n_ckpt))) return rc;
    const bool need_first_col = (p->first_column_init_type != MI355SW_INIT_WITH_ZEROES);
    h->first_col_port = false;
    if (p->first_column_port) {
        if (p->first_column_init_type != MI355SW_INIT_WITH_CUSTOM_DATA) FAIL(h, MI355SW_EINVAL, "first_column_port needs INIT_WITH_CUSTOM_DATA");
        if (!h->in_port || h->in_port_rows < m) FAIL(h, MI355SW_ESTATE, "first_column_port without an inbound port of >= %d rows (mi355sw_port_create)", m);
        h->first_col_port = true;
    }
    if (p->last_column_port) {
        if (p->want_last_column) FAIL(h, MI355SW_EINVAL, "last_column_port excludes want_last_column");
        if (!h->out_port || h->out_port_rows < m) FAIL(h, MI355SW_ESTATE, "last_column_port without an outbound port of >= %d rows (mi355sw_port_open/attach)", m);
    }
    h->first_col_pinned = need_first_col && !h->first_col_port && p->first_column_init_type == MI355SW_INIT_WITH_CUSTOM_DATA &&
                          p->stream_first_column;
    if (need_first_col && !h->first_col_pinned && !h->first_col_port && (rc = ensure(h, h->d_first_col, sizeof(int2) * ((size_t) m + 1)))) return rc;
    if (h->first_col_pinned && (rc = ensure_pinned(h, h->p_first_col, sizeof(int2) * ((size_t) m + 1)))) return rc;
    if (p->want_last_column && (rc = ensure_pinned(h, h->p_last_col, sizeof(int2) * ((size_t) m + 1)))) return rc;
    if (p->want_last_row && (rc = ensure(h, h->d_last_row, sizeof(int2) * ((size_t) n + 64)))) return rc;
    if (h->n_special > 0) {
        const size_t bytes = sizeof(int2) * (size_t) h->special_pitch * h->n_special;
        // default: 60 % of the HBM that is free right now (288 GB per MI355X: C3's 69 rows of 368 MB are 25 GB)
        int64_t budget = h->cfg.max_special_bytes;
        if (budget <= 0) {
            size_t free_b = 0, total_b = 0;
            budget = (hipMemGetInfo(&free_b, &total_b) == hipSuccess) ? (int64_t) ((double) free_b * 0.6) : (8LL << 30);
            budget += (int64_t) h->d_special.cap;        // what this handle already holds for the purpose is reusable
        }
        if ((int64_t) bytes > budget)
            FAIL(h, MI355SW_ENOMEM, "%d special rows need %zu bytes > budget %lld", h->n_special, bytes,
                 (long long) budget);
        if ((rc = ensure(h, h->d_special, bytes))) return rc;
    }

    // ---- borders ----
    // first row -> bus (AbstractDiagonalAligner::loadFirstRow :409-426 / CUDAligner::setFirstRow :461-465)
    if (p->first_row_init_type == MI355SW_INIT_WITH_CUSTOM_DATA) {
        if (!p->first_row) FAIL(h, MI355SW_EINVAL, "custom first row without data");
        HIPCHK(h, hipMemcpyAsync(h->d_bus.p, p->first_row + 1, sizeof(int2) * (size_t) n, hipMemcpyHostToDevice,
                                 h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));   // caller's buffer is only borrowed
    } else {
        HIPCHK(h, launch_fill_bus((int2*) h->d_bus.p, n, p->first_row_init_type, p->first_row_start_offset,
                                  h->stream));
    }
    // first column
    h->h_pinned[16] = 0;
    if (need_first_col) {
        if (h->first_col_port) {
            // the rows come from the previous band's GPU; only the corner cell is this side's to write.  The
            // counter is NOT touched: the neighbour may have started publishing before this call.
            mi355sw_cell corner = {0, -MI355SW_INF};
            if (p->first_column) corner = p->first_column[0];
            HIPCHK(h, hipMemcpyAsync((char*) h->in_port + 256, &corner, sizeof(corner), hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
        } else if (p->first_column_init_type == MI355SW_INIT_WITH_CUSTOM_DATA) {
            if (!p->stream_first_column) {
                if (!p->first_column) FAIL(h, MI355SW_EINVAL, "custom first column without data");
                HIPCHK(h, hipMemcpyAsync(h->d_first_col.p, p->first_column, sizeof(int2) * ((size_t) m + 1),
                                         hipMemcpyHostToDevice, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                h->h_pinned[16] = m;
                h->fed_rows = m;
            } else {
                // only the corner is known up front; rows arrive through mi355sw_stream_feed_column()
                mi355sw_cell corner = {0, -MI355SW_INF};
                if (p->first_column) corner = p->first_column[0];
                memcpy(h->p_first_col.p, &corner, sizeof(corner));
                if (p->first_column_resume_rows > 0) {
                    if (p->first_column_resume_rows > m) FAIL(h, MI355SW_EINVAL, "first_column_resume_rows > rows");
                    h->fed_rows = p->first_column_resume_rows;
                    h->h_pinned[16] = p->first_column_resume_rows;
                }
            }
        } else {
            // InitialCellsReader (InitialCellsReader.cpp:84-108) generated on the host once
            std::vector<mi355sw_cell> col((size_t) m + 1);
            const int open = (p->first_column_init_type == MI355SW_INIT_WITH_GAPS) ? 3 : 0;
            for (int k = 0; k <= m; k++) {
                const long long pos = (long long) p->first_column_start_offset + k;
                col[k].h = (pos == 0) ? 0 : (int32_t) (-2 * pos - open);
                col[k].f = -MI355SW_INF;
            }
            HIPCHK(h, hipMemcpyAsync(h->d_first_col.p, col.data(), sizeof(int2) * ((size_t) m + 1),
                                     hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            h->h_pinned[16] = m;
            h->fed_rows = m;
        }
    }
    if (h->two_phase)   // checkpoint slot 0 = the first row
        HIPCHK(h, hipMemcpyAsync(h->d_ckpt.p, h->d_bus.p, sizeof(int2) * (size_t) n, hipMemcpyDeviceToDevice, h->stream));
    // synchronisation words
    HIPCHK(h, hipMemsetAsync(h->d_progress.p, 0, sizeof(int) * ((size_t) h->strips + 1), h->stream));
    HIPCHK(h, launch_fill_int((int*) h->d_progress.p, 1, n, h->stream));   // virtual strip above: all columns ready
    HIPCHK(h, hipMemsetAsync(h->d_ctrl.p, 0, 256, h->stream));
    HIPCHK(h, launch_fill_int((int*) h->d_ctrl.p + 52, 2, -MI355SW_INF, h->stream));   // running global best (+ its never-written twin)
    HIPCHK(h, hipMemsetAsync(h->d_strip_best.p, 0, sizeof(int4) * (size_t) h->strips, h->stream));
    h->h_pinned[0] = 0;

    KernelArgs a{};
    a.seq0 = (const unsigned char*) h->d_seq0.p + part->i0;
    a.seq1 = (const unsigned char*) h->d_seq1.p + part->j0;
    a.m = m; a.n = n;
    a.n_match_codes = h->n_match_codes;
    a.pad_code = h->pad_code;
    a.seq0_shift = h->seq0_shift;
    a.num_strips = h->strips;
    a.strip_row0 = 0;
    a.strip_index0 = 0;
    a.bus = (int2*) h->d_bus.p;
    a.first_col = need_first_col ? (const int2*) (h->first_col_port ? (void*) ((char*) h->in_port + 256) : (h->first_col_pinned ? h->p_first_col.p : h->d_first_col.p)) : nullptr;
    a.last_col = p->want_last_column ? (int2*) h->p_last_col.p : (p->last_column_port ? (int2*) ((char*) h->out_port + 256) : nullptr);
    a.peer_ready = p->last_column_port ? (int*) h->out_port : nullptr;
    a.special_rows = h->n_special > 0 ? (int2*) h->d_special.p : nullptr;
    a.special_pitch = h->special_pitch;
    a.special_interval_strips = h->n_special > 0 ? h->special_interval_strips : 0;
    a.last_row = p->want_last_row ? (int2*) h->d_last_row.p : nullptr;
    a.ckpt_rows = h->two_phase ? (int2*) h->d_ckpt.p : nullptr;
    a.ckpt_pitch = h->ckpt_pitch;
    a.ckpt_interval_strips = h->ckpt_interval;
    a.progress = (int*) h->d_progress.p;
    int* ctrl = (int*) h->d_ctrl.p;
    a.ticket = ctrl + 0;
    a.abort_flag = ctrl + 16;
    a.error_flag = ctrl + 32;
    a.strips_done_dev = ctrl + 48;
    a.strips_done_host = getenv("MI355SW_NOHOST") ? nullptr : h->h_pinned + 0;
    a.first_col_ready = h->first_col_port ? (const int*) h->in_port : ((need_first_col && p->stream_first_column) ? h->h_pinned + 16 : nullptr);
    __atomic_store_n(&h->h_pinned[48], 0, __ATOMIC_RELEASE);
    a.host_error = h->h_pinned + 48;
    a.fault_strip = -1;
    if (const char* e = getenv("MI355SW_FAULT_OVERFLOW_STRIP")) a.fault_strip = atoi(e);   // test knob, packed kernel only
    h->clean_rows = 0;
    h->overflow_seen = false;
    {
        // waits on data another GPU or the host delivers: wall-time budget (10 ns ticks), MI355SW_WAIT_S seconds
        double wait_s = 3600.0;
        if (const char* e = getenv("MI355SW_WAIT_S")) { const double v = atof(e); if (v > 0) wait_s = v; }
        a.wait_ticks = (long long) (wait_s * 1e8);
    }
    __atomic_store_n(&h->h_pinned[32], 0, __ATOMIC_RELEASE);
    h->abort_strips = -1;
    a.host_abort = h->h_pinned + 32;
    a.gbest = ctrl + 52;
    a.gbest_in = h->exact_records ? ctrl + 53 : ctrl + 52;
    // block pruning: packed SW kernel only (the int32 fallback and NW simply compute everything)
    a.prune = (p->prune_blocks && h->use16 && p->recurrence_type == MI355SW_SMITH_WATERMAN) ? 1 : 0;
    a.prune_rows = p->prune_rows > 0 ? p->prune_rows : m;
    a.prune_cols = p->prune_cols > 0 ? p->prune_cols : n;
    a.pruned_slabs = (unsigned long long*) (ctrl + 40);
    a.strip_best = (int4*) h->d_strip_best.p;
    a.dbg = getenv("MI355SW_DEBUG") ? ctrl + 56 : nullptr;
    // running best shared along the band chain / with the host (packed kernels only: the int32 family keeps no gbest)
    __atomic_store_n(&h->h_pinned[64], -MI355SW_INF, __ATOMIC_RELEASE);
    __atomic_store_n(&h->h_pinned[80], -MI355SW_INF, __ATOMIC_RELEASE);
    if (p->share_best && h->use16 && !getenv("MI355SW_NO_SHARED_BEST")) {
        if (h->first_col_port) { a.chain_down_in = (const int*) ((char*) h->in_port + 64); a.chain_up_pub = (int*) ((char*) h->in_port + 128); }
        if (p->last_column_port) { a.chain_down_pub = (int*) ((char*) h->out_port + 64); a.chain_up_in = (const int*) ((char*) h->out_port + 128); }
        a.host_best_hint = h->h_pinned + 64;
        a.host_best_report = h->h_pinned + 80;
    }
    a.trace = nullptr;
    if (getenv("MI355SW_TRACE")) {
        if ((rc = ensure(h, h->d_trace, sizeof(long long) * 4 * (size_t) h->strips))) return rc;
        HIPCHK(h, hipMemsetAsync(h->d_trace.p, 0, sizeof(long long) * 4 * (size_t) h->strips, h->stream));
        a.trace = (long long*) h->d_trace.p;
    }

    h->stats = mi355sw_stats{};
    h->stats.cells = (int64_t) m * n;
    h->stats.strips = h->strips;
    h->stats.strip_rows = h->SH;
    h->stats.waves = waves;
    h->stats.profile_kernel = h->profile ? 1 : 0;
    h->stats.kernel_launches = 1;
    h->stats.total_ms = now_ms();
    h->prog_strips = 0;
    h->prog_total = h->strips;

    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    h->kargs = a;
    h->exact_ms = 0;
    // Seed pass (local alignment with exact tracking only).  Every strip starts its sweep with the exact
    // bookkeeping switched on until the running best has risen above the background level, and in the first
    // round nobody has found anything yet: the start-up delay of every hop of the strip pipeline is ~50 %
    // longer than later on.  A throw-away launch first lets the same wavefronts sweep SEED_COLS columns of
    // their own rows with all dependencies pre-satisfied (zero top border, or whatever the strip above has
    // just written: either way a lower bound of the true cells, so every score it sees is the score of a
    // real local alignment) and publish the maximum; 0.5 ms for ~3e9 cells.
    if (h->use16 && p->recurrence_type == MI355SW_SMITH_WATERMAN && p->track_best && !h->two_phase &&
        h->strips >= 64 && n >= 16384 && !h->exact_records && !getenv("MI355SW_NOSEED")) {
        const int SEED_COLS = 2048;
        const int ws = std::min(h->strips, waves);
        // layout of the seed pass's scratch: [argument block | control words | progress | strip records | bus]
        static_assert(sizeof(KernelArgs) <= 512, "the seed pass keeps its argument block in the first 512 bytes");
        const size_t o_ctrl = 512, o_prog = 768, o_sb = o_prog + (((size_t) ws + 1) * 8 + 255) / 256 * 256;
        const size_t o_bus = o_sb + (size_t) ws * sizeof(int4);
        if ((rc = ensure(h, h->d_seed, o_bus + sizeof(int2) * (SEED_COLS + 64)))) return rc;
        char* base = (char*) h->d_seed.p;
        int* ctrl2 = (int*) (base + o_ctrl);
        KernelArgs b = a;
        b.n = SEED_COLS;
        b.m = (int) std::min<long long>(m, (long long) ws * h->SH);
        b.num_strips = (b.m + h->SH - 1) / h->SH;
        b.bus = (int2*) (base + o_bus);
        b.first_col = nullptr; b.last_col = nullptr; b.special_rows = nullptr; b.special_interval_strips = 0;
        b.last_row = nullptr; b.ckpt_rows = nullptr; b.ckpt_interval_strips = 0;
        b.progress = (int*) (base + o_prog);
        b.ticket = ctrl2 + 0; b.abort_flag = ctrl2 + 16; b.error_flag = ctrl2 + 32; b.strips_done_dev = ctrl2 + 48;
        b.strips_done_host = nullptr; b.first_col_ready = nullptr; b.host_abort = nullptr;
        b.peer_ready = nullptr; b.host_error = nullptr; b.fault_strip = -1;
        b.chain_down_in = nullptr; b.chain_up_pub = nullptr; b.chain_down_pub = nullptr; b.chain_up_in = nullptr;
        b.host_best_hint = nullptr; b.host_best_report = nullptr;
        b.strip_best = (int4*) (base + o_sb);
        b.dbg = nullptr; b.trace = nullptr;
        b.independent = 1;                             // nobody waits for anybody: progress[0..ws] stays "all columns ready"
        b.prune = 1;                                   // selects the variant that publishes the running best without tracking positions
        b.prune_rows = m; b.prune_cols = n;            // ... and never prunes: the bound sees the whole matrix ahead
        b.pruned_slabs = (unsigned long long*) (ctrl2 + 40);
        HIPCHK(h, hipMemsetAsync(base + o_ctrl, 0, 256, h->stream));
        HIPCHK(h, launch_fill_int(b.progress, 2 * ((long long) ws + 1), SEED_COLS, h->stream));
        HIPCHK(h, launch_fill_bus(b.bus, SEED_COLS, MI355SW_INIT_WITH_ZEROES, 0, h->stream));
        HIPCHK(h, launch_strip_kernel_pk16(b, (KernelArgs*) base, h->R / 2, ws, h->stream, false, true));
    }
    if (h->use16) {
        h->stats.profile_kernel = 2;
        HIPCHK(h, launch_strip_kernel_pk16(a, (KernelArgs*) h->d_kargs.p, h->R / 2, waves, h->stream,
                                           p->track_best != 0 && !h->two_phase, p->recurrence_type == MI355SW_SMITH_WATERMAN));
    } else {
        HIPCHK(h, launch_strip_kernel(a, (KernelArgs*) h->d_kargs.p, h->R, waves, h->stream, p->recurrence_type == MI355SW_SMITH_WATERMAN,
                                      h->profile, p->track_best != 0));
    }
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    h->active = true;
    return MI355SW_OK;
}

int mi355sw_stream_feed_column(mi355sw_handle* h, int32_t row, const mi355sw_cell* cells, int32_t len) {
    if (!h || !h->active) return MI355SW_ESTATE;
    if (!h->first_col_pinned) FAIL(h, MI355SW_ESTATE, "feed_column without a streamed first column");
    if (row != h->fed_rows || len < 0 || row + len > h->m) FAIL(h, MI355SW_EINVAL, "feed_column out of order");
    if (len == 0) return MI355SW_OK;
    // zero-copy: the kernel reads these cells straight from pinned host memory (system-scope loads)
    memcpy((mi355sw_cell*) h->p_first_col.p + 1 + row, cells, sizeof(mi355sw_cell) * (size_t) len);
    h->fed_rows += len;
    __atomic_store_n(&h->h_pinned[16], h->fed_rows, __ATOMIC_RELEASE);
    return MI355SW_OK;
}

// Re-publish rows [0, rows) of the pinned first column that a previous attempt of the same partition already
// received (overflow rerun: the manager's stream is sequential and cannot be rewound).
static void republish_first_column(mi355sw_handle* h, int rows) {
    h->fed_rows = rows;
    __atomic_store_n(&h->h_pinned[16], rows, __ATOMIC_RELEASE);
}

// ------------------------------------------------------------------------------------------------
// column ports
// ------------------------------------------------------------------------------------------------
// control block of a fresh port: row counter 0 at +0, the two running-best words (+64 pushed by the previous band,
// +128 published by the owner) at "nothing known yet"
static const int* port_control_block() {
    static int block[64];
    block[0] = 0; block[16] = -MI355SW_INF; block[32] = -MI355SW_INF;
    return block;
}

int mi355sw_port_create(mi355sw_handle* h, int32_t rows, mi355sw_port_handle* out) {
    if (!h || rows <= 0) return MI355SW_EINVAL;
    if (h->active) FAIL(h, MI355SW_ESTATE, "port_create while a stream is active");
    HIPCHK(h, hipSetDevice(h->device));
    if (h->in_port) { (void) hipFree(h->in_port); h->in_port = nullptr; }
    const size_t bytes = 256 + sizeof(int2) * ((size_t) rows + 1);
    // fine-grained: stores arriving over xGMI from the neighbour's kernel and this GPU's system-scope loads
    // meet in memory, not in a cache that only one side looks at
    hipError_t e = hipExtMallocWithFlags(&h->in_port, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { h->in_port = nullptr; FAIL(h, MI355SW_ENOMEM, "hipExtMallocWithFlags(%zu, finegrained) failed: %s", bytes, hipGetErrorString(e)); }
    h->in_port_bytes = bytes;
    h->in_port_rows = rows;
    HIPCHK(h, hipMemcpy(h->in_port, port_control_block(), 256, hipMemcpyHostToDevice));
    HIPCHK(h, hipDeviceSynchronize());
    if (out) {
        memset(out, 0, sizeof(*out));
        hipIpcMemHandle_t ipc;
        HIPCHK(h, hipIpcGetMemHandle(&ipc, h->in_port));
        static_assert(sizeof(ipc) <= sizeof(out->ipc), "hipIpcMemHandle_t does not fit");
        memcpy(out->ipc, &ipc, sizeof(ipc));
        out->bytes = (int64_t) bytes;
        out->rows = rows;
        out->device = h->device;
    }
    return MI355SW_OK;
}

static void drop_out_port(mi355sw_handle* h) {
    if (h->out_port && h->out_port_ipc) (void) hipIpcCloseMemHandle(h->out_port);
    h->out_port = nullptr; h->out_port_rows = 0; h->out_port_ipc = false;
}

int mi355sw_port_open(mi355sw_handle* h, const mi355sw_port_handle* remote) {
    if (!h || !remote) return MI355SW_EINVAL;
    if (h->active) FAIL(h, MI355SW_ESTATE, "port_open while a stream is active");
    HIPCHK(h, hipSetDevice(h->device));
    drop_out_port(h);
    hipIpcMemHandle_t ipc;
    memcpy(&ipc, remote->ipc, sizeof(ipc));
    void* ptr = nullptr;
    HIPCHK(h, hipIpcOpenMemHandle(&ptr, ipc, hipIpcMemLazyEnablePeerAccess));
    h->out_port = ptr; h->out_port_rows = remote->rows; h->out_port_ipc = true;
    return MI355SW_OK;
}

int mi355sw_port_attach(mi355sw_handle* h, mi355sw_handle* down) {
    if (!h || !down) return MI355SW_EINVAL;
    if (h->active) FAIL(h, MI355SW_ESTATE, "port_attach while a stream is active");
    if (!down->in_port) FAIL(h, MI355SW_ESTATE, "port_attach: the downstream handle has no inbound port");
    HIPCHK(h, hipSetDevice(h->device));
    drop_out_port(h);
    if (down->device != h->device) {
        int can = 0;
        HIPCHK(h, hipDeviceCanAccessPeer(&can, h->device, down->device));
        if (!can) FAIL(h, MI355SW_EHIP, "GPU %d cannot access GPU %d", h->device, down->device);
        hipError_t e = hipDeviceEnablePeerAccess(down->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) FAIL(h, MI355SW_EHIP, "hipDeviceEnablePeerAccess: %s", hipGetErrorString(e));
        (void) hipGetLastError();
    }
    h->out_port = down->in_port; h->out_port_rows = down->in_port_rows; h->out_port_ipc = false;
    return MI355SW_OK;
}

int mi355sw_port_reset(mi355sw_handle* h) {
    if (!h) return MI355SW_EINVAL;
    if (h->active) FAIL(h, MI355SW_ESTATE, "port_reset while a stream is active");
    if (!h->in_port) FAIL(h, MI355SW_ESTATE, "no inbound port");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(h->in_port, port_control_block(), 256, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MI355SW_OK;
}

int mi355sw_port_rows_ready(mi355sw_handle* h, int32_t* rows) {
    if (!h || !rows) return MI355SW_EINVAL;
    if (!h->in_port) FAIL(h, MI355SW_ESTATE, "no inbound port");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(rows, h->in_port, sizeof(int32_t), hipMemcpyDeviceToHost, h->copy));
    HIPCHK(h, hipStreamSynchronize(h->copy));
    return MI355SW_OK;
}

int mi355sw_port_read(mi355sw_handle* h, int32_t row, mi355sw_cell* cells, int32_t len) {
    if (!h || !cells) return MI355SW_EINVAL;
    if (!h->in_port) FAIL(h, MI355SW_ESTATE, "no inbound port");
    if (row < 0 || len < 0 || row + len > h->in_port_rows) FAIL(h, MI355SW_EINVAL, "port_read range");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(cells, (char*) h->in_port + 256 + sizeof(int2) * ((size_t) row + 1), sizeof(int2) * (size_t) len,
                             hipMemcpyDeviceToHost, h->copy));
    HIPCHK(h, hipStreamSynchronize(h->copy));
    return MI355SW_OK;
}

int mi355sw_port_local_pointers(mi355sw_handle* h, void** cells, void** counter) {
    if (!h) return MI355SW_EINVAL;
    if (!h->in_port) FAIL(h, MI355SW_ESTATE, "no inbound port");
    if (cells) *cells = (char*) h->in_port + 256;
    if (counter) *counter = h->in_port;
    return MI355SW_OK;
}

int mi355sw_port_close(mi355sw_handle* h) {
    if (!h) return MI355SW_EINVAL;
    if (h->active) FAIL(h, MI355SW_ESTATE, "port_close while a stream is active");
    (void) hipSetDevice(h->device);
    drop_out_port(h);
    if (h->in_port) (void) hipFree(h->in_port);
    h->in_port = nullptr; h->in_port_bytes = 0; h->in_port_rows = 0;
    return MI355SW_OK;
}

int mi355sw_stream_poll(mi355sw_handle* h, int32_t* rows_done, int32_t* finished) {
    if (!h || !h->active) return MI355SW_ESTATE;
    // completion FIRST, the strip counter second: a kernel that ends between the two reads must not be reported as
    // "finished with rows missing" (the caller takes that for an aborted run and drops the rest of the last column)
    if (!h->finished) {
        hipError_t e = hipEventQuery(h->ev1);
        if (e == hipSuccess) h->finished = true;
        else if (e != hipErrorNotReady) FAIL(h, MI355SW_EHIP, "kernel failed: %s", hipGetErrorString(e));
    }
    // strip counter FIRST, error mirror second: the kernel writes the mirror before the counter moves past the
    // failing strip, so a counter value read while the mirror is still clean only covers exact strips
    const int done = __atomic_load_n(&h->h_pinned[0], __ATOMIC_ACQUIRE);
    const int err = __atomic_load_n(&h->h_pinned[48], __ATOMIC_ACQUIRE);
    h->prog_strips = done;
    long long rows = (long long) done * h->SH;
    if (rows > h->m) rows = h->m;
    if (err == 0) h->clean_rows = (int) rows;
    else rows = h->clean_rows;
    if (rows_done) *rows_done = (int32_t) rows;
    if (finished) *finished = h->finished ? 1 : 0;
    if (err == 16 && __atomic_load_n(&h->h_pinned[32], __ATOMIC_ACQUIRE) == 0) {
        h->overflow_seen = true;
        FAIL(h, MI355SW_EOVERFLOW16, "packed 16-bit kernel left its exact range after %lld rows; rerun with force_int32", rows);
    }
    return MI355SW_OK;
}

int mi355sw_stream_read_column(mi355sw_handle* h, int32_t row, mi355sw_cell* cells, int32_t len) {
    if (!h || !h->active || !h->sp.want_last_column) return MI355SW_ESTATE;
    if (row < 0 || len < 0 || row + len > h->m) FAIL(h, MI355SW_EINVAL, "read_column range");
    // rows below strips_done were written by the kernel (system-scope release) straight into this
    // pinned buffer: no copy, no queue
    const int done = __atomic_load_n(&h->h_pinned[0], __ATOMIC_ACQUIRE);
    const int err = __atomic_load_n(&h->h_pinned[48], __ATOMIC_ACQUIRE);
    long long valid = std::min<long long>((long long) done * h->SH, h->m);
    if (h->finished) valid = h->m;
    if (err != 0) valid = h->clean_rows;     // rows of the failing strip and below are void
    if (row + len > valid) FAIL(h, MI355SW_EINVAL, "read_column beyond completed rows (%d+%d > %lld)", row, len, valid);
    memcpy(cells, (mi355sw_cell*) h->p_last_col.p + 1 + row, sizeof(mi355sw_cell) * (size_t) len);
    return MI355SW_OK;
}

int mi355sw_stream_read_special_row(mi355sw_handle* h, int32_t k, int32_t* dp_row, mi355sw_cell* cells,
                                    int32_t col, int32_t len) {
    if (!h || !h->active) return MI355SW_ESTATE;
    if (k < 0 || k >= h->n_special || col < 0 || len < 0 || col + len > h->n) FAIL(h, MI355SW_EINVAL, "special row range");
    if (dp_row) *dp_row = (k + 1) * h->special_interval_strips * h->SH;
    if (len > 0 && cells) {
        HIPCHK(h, hipMemcpyAsync(cells, (int2*) h->d_special.p + (size_t) k * h->special_pitch + col,
                                 sizeof(int2) * (size_t) len, hipMemcpyDeviceToHost, h->copy));
        HIPCHK(h, hipStreamSynchronize(h->copy));
    }
    return MI355SW_OK;
}

int mi355sw_stream_read_last_row(mi355sw_handle* h, mi355sw_cell* cells, int32_t col, int32_t len) {
    if (!h || !h->active || !h->sp.want_last_row) return MI355SW_ESTATE;
    if (col < 0 || len < 0 || col + len > h->n) FAIL(h, MI355SW_EINVAL, "last row range");
    HIPCHK(h, hipMemcpyAsync(cells, (int2*) h->d_last_row.p + col, sizeof(int2) * (size_t) len, hipMemcpyDeviceToHost,
                             h->copy));
    HIPCHK(h, hipStreamSynchronize(h->copy));
    return MI355SW_OK;
}

int mi355sw_stream_best_hint(mi355sw_handle* h, int32_t score) {
    if (!h || !h->active) return MI355SW_ESTATE;
    // the kernel keeps T = H - 3; the word only ever grows (one writer: this thread)
    const int t = score - 3;
    if (t > __atomic_load_n(&h->h_pinned[64], __ATOMIC_RELAXED)) __atomic_store_n(&h->h_pinned[64], t, __ATOMIC_RELEASE);
    return MI355SW_OK;
}

int mi355sw_stream_running_best(mi355sw_handle* h, int32_t* score) {
    if (!h || !score) return MI355SW_EINVAL;
    const int t = __atomic_load_n(&h->h_pinned[80], __ATOMIC_ACQUIRE);
    *score = (t <= -MI355SW_INF + 3) ? -MI355SW_INF : t + 3;
    return MI355SW_OK;
}

int mi355sw_stream_abort(mi355sw_handle* h) {
    if (!h || !h->active) return MI355SW_ESTATE;
    // a store into pinned memory the kernel polls with system scope: a copy, whatever its stream, may be held
    // back until the persistent kernel has left the queue it was mapped to -- and then stops nothing
    if (h->abort_strips < 0) {
        const int done = __atomic_load_n(&h->h_pinned[0], __ATOMIC_ACQUIRE);
        h->abort_strips = std::min(h->strips, done + h->waves);
    }
    __atomic_store_n(&h->h_pinned[32], 1, __ATOMIC_RELEASE);
    // unblock strips waiting for first-column rows that will never come
    __atomic_store_n(&h->h_pinned[16], h->m, __ATOMIC_RELEASE);
    return MI355SW_OK;
}

// Exact-position pass of the two-phase scheme: re-run strips [ck*K, s_star] with exact canonical tracking,
// starting from checkpoint row ck (the bus row above strip ck*K); returns the canonical cell of s_star.
static int run_exact_pass(mi355sw_handle* h, int s_star, int want_score, mi355sw_score* out) {
    const int K = h->ckpt_interval;
    const int ck = s_star / K;
    const int start = ck * K;
    const int count = s_star - start + 1;
    HIPCHK(h, hipMemcpyAsync(h->d_bus.p, (int2*) h->d_ckpt.p + (size_t) ck * h->ckpt_pitch, sizeof(int2) * (size_t) h->n,
                             hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_progress.p, 0, sizeof(int) * ((size_t) count + 1), h->stream));
    HIPCHK(h, launch_fill_int((int*) h->d_progress.p, 1, h->n, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_ctrl.p, 0, 256, h->stream));
    HIPCHK(h, launch_fill_int((int*) h->d_ctrl.p + 52, 2, -MI355SW_INF, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_strip_best.p, 0, sizeof(int4) * (size_t) count, h->stream));
    h->h_pinned[0] = 0;
    KernelArgs b = h->kargs;
    b.num_strips = count;
    b.strip_row0 = start * h->SH;
    b.strip_index0 = start;
    b.special_rows = nullptr; b.special_interval_strips = 0;
    b.last_col = nullptr; b.last_row = nullptr; b.ckpt_rows = nullptr; b.ckpt_interval_strips = 0;
    b.first_col_ready = nullptr;        // every row of a streamed first column has arrived by now
    b.peer_ready = nullptr;             // the boundary column was published by the main pass
    b.chain_down_in = nullptr; b.chain_up_pub = nullptr; b.chain_down_pub = nullptr; b.chain_up_in = nullptr;
    b.host_best_hint = nullptr; b.host_best_report = nullptr;
    b.fault_strip = -1;
    b.trace = nullptr;
    int waves = std::min(count, h->waves);
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0));
    HIPCHK(h, hipEventCreate(&e1));
    HIPCHK(h, hipEventRecord(e0, h->stream));
    HIPCHK(h, launch_strip_kernel_pk16(b, (KernelArgs*) h->d_kargs.p, h->R / 2, waves, h->stream, true,
                                       h->sp.recurrence_type == MI355SW_SMITH_WATERMAN));
    HIPCHK(h, hipEventRecord(e1, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    h->exact_ms = ms;
    int ctrl[64];
    HIPCHK(h, hipMemcpy(ctrl, h->d_ctrl.p, sizeof(ctrl), hipMemcpyDeviceToHost));
    if (ctrl[32] == 16) FAIL(h, MI355SW_EOVERFLOW16, "packed 16-bit kernel left its exact range in the exact pass");
    if (ctrl[32] != 0) FAIL(h, MI355SW_ETIMEOUT, "in-kernel wait timed out in the exact pass (code %d)", ctrl[32]);
    int4 r;
    HIPCHK(h, hipMemcpy(&r, (int4*) h->d_strip_best.p + (count - 1), sizeof(int4), hipMemcpyDeviceToHost));
    if (r.w != 1 || r.z < 0 || r.x != want_score)
        FAIL(h, MI355SW_ESTATE, "exact pass disagrees with the main pass (strip %d: %d vs %d)", s_star, r.x, want_score);
    out->score = r.x; out->i = r.y; out->j = r.z;
    return MI355SW_OK;
}

int mi355sw_stream_end(mi355sw_handle* h, mi355sw_score* best, int32_t* n_special_rows) {
    if (!h || !h->active) return MI355SW_ESTATE;
    hipError_t e = hipStreamSynchronize(h->stream);
    h->active = false;
    h->finished = true;
    if (e != hipSuccess) FAIL(h, MI355SW_EHIP, "strip kernel failed: %s", hipGetErrorString(e));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->stats.kernel_ms = ms;
    h->stats.total_ms = now_ms() - h->stats.total_ms;
    int ctrl[64];
    HIPCHK(h, hipMemcpy(ctrl, h->d_ctrl.p, sizeof(ctrl), hipMemcpyDeviceToHost));
    const bool aborted = ctrl[16] != 0 || __atomic_load_n(&h->h_pinned[32], __ATOMIC_ACQUIRE) != 0;
    // after a host stop the strips still in flight give up mid-sweep and their followers read stale bus cells:
    // whatever those raise is void like the rest of their output (a genuine overflow stops the kernel by itself,
    // the host then never sees the goal and never says stop)
    const bool host_stopped = __atomic_load_n(&h->h_pinned[32], __ATOMIC_ACQUIRE) != 0;
    if (ctrl[32] == 16 && !host_stopped) FAIL(h, MI355SW_EOVERFLOW16, "packed 16-bit kernel left its exact range; rerun with force_int32");
    if (ctrl[32] != 0 && !host_stopped) FAIL(h, MI355SW_ETIMEOUT, "in-kernel wait timed out (code %d)", ctrl[32]);
    if (getenv("MI355SW_TRACE") && h->d_trace.p) {
        std::vector<long long> tr((size_t) h->strips * 4);
        HIPCHK(h, hipMemcpy(tr.data(), h->d_trace.p, tr.size() * sizeof(long long), hipMemcpyDeviceToHost));
        FILE* f = fopen(getenv("MI355SW_TRACE"), "wb");
        if (f) { fwrite(tr.data(), sizeof(long long), tr.size(), f); fclose(f); }
    }
    h->strip_best_host.resize((size_t) h->strips);
    HIPCHK(h, hipMemcpy(h->strip_best_host.data(), h->d_strip_best.p, sizeof(int4) * (size_t) h->strips,
                        hipMemcpyDeviceToHost));
    mi355sw_score b;
    b.i = -1; b.j = -1; b.score = -MI355SW_INF;
    if (h->sp.track_best && h->two_phase && !aborted) {
        // main pass recorded values only: first strip holding the maximum, then its exact canonical cell
        int S = -MI355SW_INF, s_star = -1;
        for (int s = 0; s < h->strips; s++) {
            const int4 r = h->strip_best_host[(size_t) s];
            if (r.w != 0 && r.x > S) { S = r.x; s_star = s; }
        }
        if (s_star >= 0) {
            int rc2 = run_exact_pass(h, s_star, S, &b);
            if (rc2) return rc2;
            b.i += h->part.i0; b.j += h->part.j0;
        }
        h->stats.kernel_ms += h->exact_ms;
        h->stats.kernel_launches = 2;
    } else if (h->sp.track_best) {
        // canonical order of BestScoreList (M/common/BestScoreList.hpp:30-38): score desc, i asc, j asc
        for (int s = 0; s < h->strips; s++) {
            const int4 r = h->strip_best_host[(size_t) s];
            if (r.w == 0 || r.z < 0) continue;
            if (r.x > b.score || (r.x == b.score && (r.y < b.i || (r.y == b.i && r.z < b.j)))) {
                b.score = r.x; b.i = r.y; b.j = r.z;
            }
        }
        if (b.j >= 0) { b.i += h->part.i0; b.j += h->part.j0; }
    }
    h->final_best = b;
    if (best) *best = b;
    if (n_special_rows) *n_special_rows = h->n_special;
    // after a host stop the strips claimed later were skipped (they still count in ctrl[48]): an upper bound of
    // what was computed is what was complete or in flight at that moment
    const int done_strips = (aborted && h->abort_strips >= 0) ? h->abort_strips : (aborted ? ctrl[48] : h->strips);
    h->stats.processed_cells = (int64_t) std::min<long long>((long long) done_strips * h->SH, h->m) * h->n;
    {
        unsigned long long slabs = 0;
        memcpy(&slabs, ctrl + 40, sizeof(slabs));
        h->stats.pruned_cells = (int64_t) slabs * 64 * h->SH;
        if (h->stats.pruned_cells > h->stats.processed_cells) h->stats.pruned_cells = h->stats.processed_cells;
        h->stats.processed_cells -= h->stats.pruned_cells;
    }
    h->processed_total += h->stats.processed_cells;
    // 17 B per column per strip + seq0 once + flushed rows
    h->stats.algorithmic_bytes = 17LL * h->n * h->strips + h->m + 8LL * (h->n + 1) * (h->n_special + (h->sp.want_last_row ? 1 : 0));
    h->prog_strips = h->strips;
    return MI355SW_OK;
}

int mi355sw_stream_strip_scores(mi355sw_handle* h, mi355sw_score* out, int32_t max_count) {
    if (!h || !out) return MI355SW_EINVAL;
    if (h->two_phase) {   // only the global best has an exact position in the two-phase scheme
        if (max_count < 1 || h->final_best.j < 0) return 0;
        out[0] = h->final_best;
        return 1;
    }
    int cnt = 0;
    for (size_t s = 0; s < h->strip_best_host.size() && cnt < max_count; s++) {
        const int4 r = h->strip_best_host[s];
        if (r.w == 0 || r.z < 0) continue;
        out[cnt].score = r.x; out[cnt].i = r.y + h->part.i0; out[cnt].j = r.z + h->part.j0;
        cnt++;
    }
    return cnt;
}

// Best-score records of the strips [*rows_sent / SH, upto_strip) of the ACTIVE stream, handed to the manager while
// the kernel is still running (AbstractDiagonalAligner::flushBlockScores, :392-403, does it once per diagonal): a
// caller that checkpoints needs the best of everything above a special row at the moment it stores that row.
static int flush_strip_scores(mi355sw_handle* h, const mi355sw_partition* part, const mi355sw_manager* mg, void* user,
                              int upto_strip, int* rows_sent) {
    const int first = *rows_sent / h->SH;
    if (upto_strip > h->strips) upto_strip = h->strips;
    if (upto_strip <= first) return MI355SW_OK;
    std::vector<int4> rec((size_t) (upto_strip - first));
    HIPCHK(h, hipMemcpyAsync(rec.data(), (int4*) h->d_strip_best.p + first, sizeof(int4) * rec.size(), hipMemcpyDeviceToHost, h->copy));
    HIPCHK(h, hipStreamSynchronize(h->copy));
    for (int s = first; s < upto_strip; s++) {
        const int4 r = rec[(size_t) (s - first)];
        if (r.w == 1 && r.z >= 0) {
            mi355sw_score sc;
            sc.score = r.x; sc.i = r.y + part->i0; sc.j = r.z + part->j0;
            mg->dispatch_score(user, sc, -1, -1);
        } else if (r.w == 2 && r.x > -MI355SW_INF && mg->dispatch_strip_value) {
            const long long lo = (long long) s * h->SH, hi = std::min<long long>(lo + h->SH, h->m);
            mg->dispatch_strip_value(user, part->i0 + (int) lo, part->i0 + (int) hi, r.x);
        }
    }
    *rows_sent = (int) std::min<long long>((long long) upto_strip * h->SH, h->m);
    return MI355SW_OK;
}

// Block scores (config.block_score_columns = W > 0): the best cell of every block of the grid "strip x W columns",
// reported as dispatch_score(score, bx, by) -- what CUDAligner::getBlockScores + AbstractDiagonalAligner::
// flushBlockScores deliver (X/CUDAligner.cpp:441-452, AbstractDiagonalAligner.cpp:392-403; consumer: --dump-blocks).
// The strip kernel keeps ONE canonical best per strip, so the partition is swept once more as a chain of W-column
// bands -- band bx's per-strip records ARE the records of the blocks (bx, by) -- each band taking the last column of
// the band before it as its first column.  Exact rectangles, exact canonical cells, no pruning.
static int block_scores_pass(mi355sw_handle* h, const mi355sw_partition* part, const mi355sw_manager* mg, void* user,
                             int recurrence, const std::vector<mi355sw_cell>& row_host, bool custom_col, int rows_per_lane) {
    const int W = h->cfg.block_score_columns;
    const int m = part->i1 - part->i0, n = part->j1 - part->j0;
    const int B = (n + W - 1) / W;
    std::vector<mi355sw_cell> colbuf((size_t) m + 1), firstcol;
    if (custom_col) firstcol.assign((const mi355sw_cell*) h->p_first_col.p, (const mi355sw_cell*) h->p_first_col.p + m + 1);
    const mi355sw_stats main_stats = h->stats;
    const int saved_R = h->cfg.rows_per_lane;
    h->cfg.rows_per_lane = rows_per_lane;          // same grid rows as the main pass
    h->exact_records = true;
    double extra_ms = 0;
    int rc = MI355SW_OK, launches = 0;
    for (int bx = 0; bx < B && rc == MI355SW_OK; bx++) {
        mi355sw_partition bp = {part->i0, part->j0 + bx * W, part->i1, std::min(part->j0 + (bx + 1) * W, part->j1)};
        for (int attempt = 0; attempt < 2; attempt++) {
            mi355sw_stream_params sp{};
            sp.recurrence_type = recurrence;
            sp.track_best = 1;
            sp.force_int32 = attempt;
            if (!row_host.empty()) { sp.first_row_init_type = MI355SW_INIT_WITH_CUSTOM_DATA; sp.first_row = row_host.data() + (bp.j0 - part->j0); }
            if (bx == 0) {
                if (custom_col) { sp.first_column_init_type = MI355SW_INIT_WITH_CUSTOM_DATA; sp.first_column = firstcol.data(); }
            } else {
                colbuf[0].h = row_host.empty() ? 0 : row_host[(size_t) (bp.j0 - part->j0)].h;   // H(i0, band's first column - 1)
                colbuf[0].f = -MI355SW_INF;
                sp.first_column_init_type = MI355SW_INIT_WITH_CUSTOM_DATA;
                sp.first_column = colbuf.data();
            }
            sp.want_last_column = bx < B - 1;
            if ((rc = mi355sw_stream_begin(h, &bp, &sp))) break;
            hipError_t e = hipStreamSynchronize(h->stream);
            int rows_done = 0, fin = 0;
            rc = (e == hipSuccess) ? mi355sw_stream_poll(h, &rows_done, &fin) : MI355SW_EHIP;
            if (!rc && sp.want_last_column) rc = mi355sw_stream_read_column(h, 0, colbuf.data() + 1, m);
            mi355sw_score b;
            const int rc2 = mi355sw_stream_end(h, &b, nullptr);
            if (!rc) rc = rc2;
            extra_ms += h->stats.kernel_ms;
            launches += h->stats.kernel_launches;
            if (rc == MI355SW_EOVERFLOW16 && attempt == 0) { rc = MI355SW_OK; continue; }
            break;
        }
        if (rc) break;
        for (int s = 0; s < h->strips; s++) {
            const int4 r = h->strip_best_host[(size_t) s];
            mi355sw_score sc;
            if (r.w == 1 && r.z >= 0) { sc.score = r.x; sc.i = r.y + bp.i0; sc.j = r.z + bp.j0; }
            else { sc.score = -MI355SW_INF; sc.i = -1; sc.j = -1; }
            mg->dispatch_score(user, sc, bx, s);
        }
    }
    h->cfg.rows_per_lane = saved_R;
    h->exact_records = false;
    h->stats = main_stats;
    h->stats.kernel_ms += extra_ms;
    h->stats.kernel_launches += launches;
    return rc;
}

// ------------------------------------------------------------------------------------------------
// IAligner::alignPartition on top of the streaming form
// ------------------------------------------------------------------------------------------------
int mi355sw_align_partition(mi355sw_handle* h, const mi355sw_partition* part, const mi355sw_manager* mg, void* user) {
    if (!h || !part || !mg) return MI355SW_EINVAL;
    const int m = part->i1 - part->i0, n = part->j1 - part->j0;
    if (m <= 0 || n <= 0) return MI355SW_OK;   // AlignerManager.cpp:96-99: zero-area partition skipped
    mi355sw_stream_params sp{};
    sp.recurrence_type = mg->get_recurrence_type(user);
    sp.first_row_init_type = mg->get_first_row_init_type(user);
    sp.first_column_init_type = mg->get_first_column_init_type(user);
    const bool want_special = mg->must_dispatch_special_rows && mg->must_dispatch_special_rows(user);
    sp.special_row_interval = want_special ? mg->get_special_row_interval(user) : 0;
    sp.want_last_row = mg->must_dispatch_last_row && mg->must_dispatch_last_row(user);
    sp.want_last_column = mg->must_dispatch_last_column && mg->must_dispatch_last_column(user);
    const bool want_scores = mg->must_dispatch_scores && mg->must_dispatch_scores(user);
    const bool want_last_cell = mg->must_dispatch_last_cell && mg->must_dispatch_last_cell(user);
    sp.track_best = want_scores;
    if (want_last_cell) sp.want_last_row = 1;
    if (mg->must_prune_blocks && mg->must_prune_blocks(user)) {
        // AbstractBlockPruning: the bound uses the extents of the SUPER-partition (M3)
        mi355sw_partition sup = *part;
        if (mg->get_super_partition) mg->get_super_partition(user, &sup);
        sp.prune_blocks = 1;
        sp.prune_rows = std::max(sup.i1, part->i1) - part->i0;
        sp.prune_cols = std::max(sup.j1, part->j1) - part->j0;
    }

    // AbstractDiagonalAligner::prepareIterations (:76-104): corner from both borders, then the row
    mi355sw_cell corner_c, corner_r;
    mg->receive_first_column(user, &corner_c, 1);
    mg->receive_first_row(user, &corner_r, 1);
    mi355sw_cell first_col_tail = corner_c, first_row_tail = corner_r;

    std::vector<mi355sw_cell> row_host, col_host;
    if (sp.first_row_init_type != MI355SW_INIT_WITH_ZEROES) {
        // generic path: take whatever the manager streams (gaps with any start offset, custom data)
        row_host.resize((size_t) n + 1);
        row_host[0] = corner_r;
        const int CH = 1 << 20;
        for (int j = 0; j < n; j += CH) mg->receive_first_row(user, row_host.data() + 1 + j, std::min(CH, n - j));
        first_row_tail = row_host[(size_t) n];
        sp.first_row_init_type = MI355SW_INIT_WITH_CUSTOM_DATA;
        sp.first_row = row_host.data();
    } else {
        first_row_tail.h = 0; first_row_tail.f = -MI355SW_INF;
    }
    // AbstractDiagonalAligner::loadFirstRow (:419-422): first cell of the last column
    {
        mi355sw_cell c = first_row_tail;
        c.f = -MI355SW_INF;
        mg->dispatch_column(user, part->j1, &c, 1);
    }
    const int orig_col_type = sp.first_column_init_type;
    if (orig_col_type != MI355SW_INIT_WITH_ZEROES) {
        sp.first_column_init_type = MI355SW_INIT_WITH_CUSTOM_DATA;
        sp.stream_first_column = 1;
        col_host.resize(1);
        col_host[0] = corner_c;
        sp.first_column = col_host.data();
    }
    if (mg->must_continue && !mg->must_continue(user)) return MI355SW_OK;

    // The packed 16-bit kernel re-centres its window on the wavefront, but should it ever leave its exact
    // range this is reported, never silent, and the partition is re-run with the int32 kernel.  Rows are only
    // handed to the manager while the kernel's error mirror is clean (mi355sw_stream_poll), so everything
    // dispatched before the report is exact and stays dispatched: the rerun REPLAYS -- the first-column cells
    // already received are still in the pinned column and are re-published at once (the manager's stream is
    // sequential and is only asked for the rows that are still missing), last-column chunks and special rows
    // already handed over are not sent again.
    bool force32 = false;
    int rc = MI355SW_OK;
    // (the int32 kernels have fewer strip heights than the packed one, so a rerun may use another height: all
    //  replay state is kept in DP rows, not in strips)
    int fed = 0, col_sent = 0;
    int special_last_row = 0;             // DP row of the last special row handed over
    int scores_rows_sent = 0;             // rows whose strip records have been handed over
    bool stopped = false;
    int stop_rows = 0;                    // rows that were complete (and dispatched) when the manager said stop
    std::vector<mi355sw_cell> rowbuf;
    for (int attempt = 0; attempt < 2; attempt++) {
    sp.force_int32 = force32 ? 1 : 0;
    rc = mi355sw_stream_begin(h, part, &sp);
    if (rc) return rc;
    if (attempt > 0 && fed > 0) republish_first_column(h, fed);
    // nothing to hand over before the end: wait for the kernel without touching the runtime
    const bool quiet = (orig_col_type == MI355SW_INIT_WITH_ZEROES) && !sp.want_last_column && h->n_special == 0;
    const int SH = h->SH;
    int special_sent = 0;                 // slots of THIS attempt's special-row buffer already dealt with
    std::vector<mi355sw_cell> buf((size_t) std::max(SH, 1 << 16));
    bool overflow = false;
    for (;;) {
        // feed the first column in strip-sized chunks (AbstractDiagonalAligner::loadFirstColumn :433-456)
        // (nothing more is fed once the manager has said stop: AbstractDiagonalAligner leaves its iteration loop at the
        //  first mustContinue() == false, M/libmasa/aligners/AbstractDiagonalAligner.cpp:64; special rows above the
        //  rows already dispatched are still handed over -- the reference flushes them before the column of the same
        //  iteration, :286-372 -- rows below are void and stay here)
        if (orig_col_type != MI355SW_INIT_WITH_ZEROES && !stopped) {
            int budget = 64;   // chunks per poll round, keeps the stream ahead without starving dispatches
            while (fed < m && budget-- > 0) {
                const int len = std::min(SH, m - fed);
                mg->receive_first_column(user, buf.data(), len);
                first_col_tail = buf[(size_t) len - 1];
                if ((rc = mi355sw_stream_feed_column(h, fed, buf.data(), len))) { mi355sw_stream_abort(h); mi355sw_stream_end(h, nullptr, nullptr); return rc; }
                fed += len;
            }
        }
        int rows_done = 0, fin = 0;
        rc = mi355sw_stream_poll(h, &rows_done, &fin);
        if (rc == MI355SW_EOVERFLOW16 && !force32) { overflow = true; break; }
        if (rc) { mi355sw_stream_abort(h); mi355sw_stream_end(h, nullptr, nullptr); return rc; }
        if (quiet && !fin) { struct timespec ts = {0, 200000}; nanosleep(&ts, nullptr); if (!(mg->must_continue && !mg->must_continue(user))) continue; }
        // special rows that are complete (AbstractDiagonalAligner::flushSpecialRows :286-317)
        // (reads of device-resident rows go through the copy stream: only once nothing more has to be
        //  fed, so that a copy delayed by the running kernel can never starve the kernel of its column)
        while ((!stopped || fin) && special_sent < h->n_special && (fin || fed >= m || orig_col_type == MI355SW_INIT_WITH_ZEROES)) {
            const int dp_row = (special_sent + 1) * h->special_interval_strips * SH;
            if (dp_row > (stopped ? stop_rows : rows_done)) break;
            if (dp_row <= special_last_row) { special_sent++; continue; }   // handed over by the attempt before
            // the scores of everything above the row first: whoever stores the row as a checkpoint stores them with it
            if (want_scores && (rc = flush_strip_scores(h, part, mg, user, dp_row / SH, &scores_rows_sent))) { mi355sw_stream_abort(h); mi355sw_stream_end(h, nullptr, nullptr); return rc; }
            mi355sw_cell c;
            if (orig_col_type == MI355SW_INIT_WITH_ZEROES) { c.h = 0; }
            else c = ((const mi355sw_cell*) h->p_first_col.p)[dp_row];      // cell of DP row dp_row (index 0 = corner)
            c.f = -MI355SW_INF;
            mg->dispatch_row(user, part->i0 + dp_row, &c, 1);
            rowbuf.resize((size_t) n);
            if ((rc = mi355sw_stream_read_special_row(h, special_sent, nullptr, rowbuf.data(), 0, n))) { mi355sw_stream_abort(h); mi355sw_stream_end(h, nullptr, nullptr); return rc; }
            const int CH = 1 << 20;
            for (int j = 0; j < n; j += CH) mg->dispatch_row(user, part->i0 + dp_row, rowbuf.data() + j, std::min(CH, n - j));
            special_sent++;
            special_last_row = dp_row;
        }
        // last column chunks (AbstractDiagonalAligner::flushLastColumn :361-372)
        if (sp.want_last_column) {
            while (col_sent < rows_done && !stopped) {
                const int len = std::min(SH, rows_done - col_sent);
                if ((rc = mi355sw_stream_read_column(h, col_sent, buf.data(), len))) { mi355sw_stream_abort(h); mi355sw_stream_end(h, nullptr, nullptr); return rc; }
                mg->dispatch_column(user, part->j1, buf.data(), len);
                col_sent += len;
                if (mg->must_continue && !mg->must_continue(user)) { stopped = true; stop_rows = col_sent; }
            }
        }
        if (!stopped && mg->must_continue && !mg->must_continue(user)) { stopped = true; stop_rows = sp.want_last_column ? col_sent : rows_done; }
        if (stopped && !fin) { mi355sw_stream_abort(h); }
        if (fin && (stopped || (rows_done >= m && special_sent >= h->n_special && (!sp.want_last_column || col_sent >= m)))) break;
        if (fin && rows_done < m) break;   // aborted kernel
        if (!fin && fed >= m) {
            // nothing to feed: sleep a little instead of hammering the runtime
            struct timespec ts = {0, 200000};
            nanosleep(&ts, nullptr);
        }
    }
    if (overflow) {
        (void) mi355sw_stream_abort(h);
        (void) mi355sw_stream_end(h, nullptr, nullptr);
        force32 = true;
        continue;
    }
    // last row (AbstractDiagonalAligner::flushLastRow :325-353) and last cell (:378-386)
    mi355sw_score best;
    int nsp = 0;
    std::vector<mi355sw_cell> lastrow;
    if (!stopped && sp.want_last_row) {
        lastrow.resize((size_t) n);
        if ((rc = mi355sw_stream_read_last_row(h, lastrow.data(), 0, n))) { mi355sw_stream_end(h, nullptr, nullptr); return rc; }
    }
    rc = mi355sw_stream_end(h, &best, &nsp);
    if (rc == MI355SW_EOVERFLOW16 && !force32) { force32 = true; continue; }   // e.g. reported by the exact pass
    if (rc) return rc;
    if (stopped) return MI355SW_OK;
    if (mg->must_dispatch_last_row && mg->must_dispatch_last_row(user)) {
        mi355sw_cell c = first_col_tail;
        if (orig_col_type == MI355SW_INIT_WITH_ZEROES) c.h = 0;
        c.f = -MI355SW_INF;
        mg->dispatch_row(user, part->i1, &c, 1);
        const int CH = 1 << 20;
        for (int j = 0; j < n; j += CH) mg->dispatch_row(user, part->i1, lastrow.data() + j, std::min(CH, n - j));
    }
    if (want_scores) {
        // AbstractDiagonalAligner::flushBlockScores (:392-403): one score per strip instead of per block.  What was
        // not handed over next to a special row follows now; the two-phase scheme only knows its one exact cell.
        if (h->two_phase) {
            if (best.j >= 0) mg->dispatch_score(user, best, -1, -1);
        } else {
            const int first = scores_rows_sent / h->SH;
            for (int sidx = first; sidx < h->strips; sidx++) {
                const int4 r = h->strip_best_host[(size_t) sidx];
                if (r.w == 0 || r.z < 0) continue;
                mi355sw_score sc;
                sc.score = r.x; sc.i = r.y + part->i0; sc.j = r.z + part->j0;
                mg->dispatch_score(user, sc, -1, -1);
            }
        }
    }
    if (want_last_cell) {
        mi355sw_score s;
        s.i = part->i1 - 1; s.j = part->j1 - 1; s.score = lastrow[(size_t) n - 1].h;
        mg->dispatch_score(user, s, -1, -1);
    }
    if (h->cfg.block_score_columns > 0 && want_scores)
        return block_scores_pass(h, part, mg, user, sp.recurrence_type, row_host, orig_col_type != MI355SW_INIT_WITH_ZEROES, h->R);
    return MI355SW_OK;
    }   // attempt
    return rc;
}

// AbstractBlockProcessor::processBlock seam (S3): one partition-shaped call with explicit borders.
int mi355sw_process_block(mi355sw_handle* h, mi355sw_cell* row, mi355sw_cell* col, int32_t i0, int32_t j0,
                          int32_t i1, int32_t j1, int32_t recurrence, mi355sw_score* best) {
    if (!h || !row || !col) return MI355SW_EINVAL;
    const int m = i1 - i0, n = j1 - j0;
    if (m <= 0 || n <= 0) FAIL(h, MI355SW_EINVAL, "empty block");
    mi355sw_partition part = {i0, j0, i1, j1};
    mi355sw_stream_params sp{};
    sp.recurrence_type = recurrence;
    std::vector<mi355sw_cell> frow((size_t) n + 1);
    frow[0].h = col[0].h; frow[0].f = -MI355SW_INF;
    memcpy(frow.data() + 1, row, sizeof(mi355sw_cell) * (size_t) n);
    sp.first_row_init_type = MI355SW_INIT_WITH_CUSTOM_DATA;
    sp.first_row = frow.data();
    sp.first_column_init_type = MI355SW_INIT_WITH_CUSTOM_DATA;
    sp.first_column = col;
    sp.want_last_column = 1;
    sp.want_last_row = 1;
    sp.track_best = 1;
    mi355sw_score b;
    std::vector<mi355sw_cell> lr((size_t) n), lc((size_t) m);
    int rc = MI355SW_OK;
    // every input is in hand, so a block the packed kernel cannot hold is simply computed again in int32
    for (int attempt = 0; attempt < 2; attempt++) {
        sp.force_int32 = attempt;
        if ((rc = mi355sw_stream_begin(h, &part, &sp))) return rc;
        // the kernel must be complete before the borders are read back
        HIPCHK(h, hipStreamSynchronize(h->stream));
        int rows_done = 0, fin = 0;
        rc = mi355sw_stream_poll(h, &rows_done, &fin);
        if (!rc) rc = mi355sw_stream_read_last_row(h, lr.data(), 0, n);
        if (!rc) rc = mi355sw_stream_read_column(h, 0, lc.data(), m);
        if (rc) { (void) mi355sw_stream_end(h, nullptr, nullptr); if (rc == MI355SW_EOVERFLOW16 && attempt == 0) continue; return rc; }
        rc = mi355sw_stream_end(h, &b, nullptr);
        if (rc == MI355SW_EOVERFLOW16 && attempt == 0) continue;
        if (rc) return rc;
        break;
    }
    // CPUBlockProcessor.cpp:159-165: col[0] <- H(i0-1, j1-1) (diagonal for the block to the right)
    col[0].h = row[n - 1].h;
    memcpy(row, lr.data(), sizeof(mi355sw_cell) * (size_t) n);
    memcpy(col + 1, lc.data(), sizeof(mi355sw_cell) * (size_t) m);
    if (best) {
        if (b.j < 0) { best->i = -1; best->j = -1; best->score = -MI355SW_INF; }
        else *best = b;
    }
    return MI355SW_OK;
}

int mi355sw_match_last_column(mi355sw_handle* h, const mi355sw_cell* buffer, const mi355sw_cell* base, int32_t len,
                              int32_t goal, mi355sw_match_result* out) {
    if (!buffer || !base || !out) return MI355SW_EINVAL;
    // AlignerUtils::matchColumn (M/libmasa/utils/AlignerUtils.cpp:50-107), gap_open = 3
    out->found = 0; out->k = -1; out->score = 0; out->type = 0;
    for (int k = 0; k < len; k++) {
        const int sum_match = base[k].h + buffer[k].h;
        const int sum_gap = base[k].f + buffer[k].f + 3;
        if (sum_match == goal) { out->found = 1; out->k = k; out->score = base[k].h; out->type = 0; return MI355SW_OK; }
        if (sum_gap == goal) { out->found = 1; out->k = k; out->score = base[k].f; out->type = 1; return MI355SW_OK; }
        if (sum_match > goal || sum_gap > goal) { out->k = k; out->type = sum_match > goal ? -1 : -2; return MI355SW_OK; }
    }
    return MI355SW_OK;
}

int mi355sw_stage4(mi355sw_handle* h, const mi355sw_crosspoint* in, int32_t count, int32_t max_size, mi355sw_crosspoint** out,
                   int32_t* out_count, mi355sw_stage4_stats* stats) {
    if (!h || !in || !out || !out_count || count < 1 || max_size < 1) return MI355SW_EINVAL;
    if (!h->have_seq) FAIL(h, MI355SW_ESTATE, "stage4 before set_sequences");
    if (h->active) FAIL(h, MI355SW_ESTATE, "stage4 while a stream is active");
    for (int k = 0; k < count; k++)
        if (in[k].i < 0 || in[k].j < 0 || in[k].i > h->len0 || in[k].j > h->len1 || in[k].type < 0 || in[k].type > 2 ||
            (k > 0 && (in[k].i < in[k - 1].i || in[k].j < in[k - 1].j)))
            FAIL(h, MI355SW_EINVAL, "crosspoint %d (%d,%d,%d) out of order or out of range", k, in[k].type, in[k].i, in[k].j);
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<Stage4Crosspoint> list((size_t) count);
    for (int k = 0; k < count; k++) { list[(size_t) k].type = in[k].type; list[(size_t) k].i = in[k].i; list[(size_t) k].j = in[k].j; list[(size_t) k].score = in[k].score; }
    Stage4Stats st{};
    hipError_t he = hipSuccess;
    const int rc = stage4_refine((const unsigned char*) h->d_seq0.p, h->len0, (const unsigned char*) h->d_seq1.p, h->len1,
                                 h->seq0_shift, h->stream, list, max_size, &st, &he);
    if (rc == -1) FAIL(h, MI355SW_EHIP, "stage 4: %s", hipGetErrorString(he));
    if (rc == -2) FAIL(h, MI355SW_ETOOLARGE, "stage 4: a partition exceeds 131072 columns; store more special rows in stages 1-3");
    if (rc < 0) FAIL(h, MI355SW_ETRACEBACK, "stage 4: %s", rc == -3 ? "a partition has no matching column (backtrace lost)" : "a column's scores exceed the partition's score difference");
    mi355sw_crosspoint* o = (mi355sw_crosspoint*) malloc(sizeof(mi355sw_crosspoint) * list.size());
    if (!o) FAIL(h, MI355SW_ENOMEM, "stage 4: out of host memory");
    for (size_t k = 0; k < list.size(); k++) { o[k].type = list[k].type; o[k].i = list[k].i; o[k].j = list[k].j; o[k].score = list[k].score; }
    *out = o;
    *out_count = (int32_t) list.size();
    if (stats) { stats->steps = st.steps; stats->kernel_ms = st.kernel_ms; stats->dp_cells = st.dp_cells; stats->partitions = st.partitions; }
    return MI355SW_OK;
}

void mi355sw_free(void* p) { free(p); }

int mi355sw_progress(mi355sw_handle* h, char* buf, size_t len) {
    if (!h || !buf || !len) return MI355SW_EINVAL;
    int done = h->h_pinned ? __atomic_load_n(&h->h_pinned[0], __ATOMIC_RELAXED) : 0;
    if (getenv("MI355SW_DEBUG") && h->d_ctrl.p && h->copy) {
        int ctrl[64];
        if (hipMemcpyAsync(ctrl, h->d_ctrl.p, sizeof(ctrl), hipMemcpyDeviceToHost, h->copy) == hipSuccess &&
            hipStreamSynchronize(h->copy) == hipSuccess) {
            snprintf(buf, len, "PROGRESS: %d/%d strips ticket=%d abort=%d err=%d done_dev=%d dbg=[%d %d %d %d | %d %d %d]", done,
                     h->prog_total.load(), ctrl[0], ctrl[16], ctrl[32], ctrl[48], ctrl[56], ctrl[57], ctrl[58], ctrl[59], ctrl[60], ctrl[61], ctrl[62]);
            return MI355SW_OK;
        }
    }
    snprintf(buf, len, "PROGRESS: %d/%d strips", done, h->prog_total.load());
    return MI355SW_OK;
}

long long mi355sw_processed_cells(mi355sw_handle* h) { return h ? h->processed_total.load() : 0; }

int mi355sw_get_stats(mi355sw_handle* h, mi355sw_stats* out) {
    if (!h || !out) return MI355SW_EINVAL;
    *out = h->stats;
    return MI355SW_OK;
}

}  // extern "C"
