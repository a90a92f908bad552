// ggml-mi355x.cpp — ggml backend plugin for MI355X: reg / device / buffer type / buffer / backend vtables
// (contract: ggml/src/ggml-backend-impl.h:17-207) over the kernel C-ABI of include/ggml_mi355x_qmm.h (quantized MUL_MAT /
// MUL_MAT_ID, the hot path) and include/ggml_mi355x_ops.h (the other ops of a transformer layer).
//
// Mirrors the structure of the reference's "backend per ggml spec" section
// (ggml/src/ggml-hexagon/ggml-hexagon.cpp:5417-5427 buffer iface, 5708-5737 buffer type, 5818-5834 device iface,
// 5555-5574 graph_compute node loop, 5065-5115 supports_op gate, 5941-6007 reg, 6066-6125 init, 6127 DL_IMPL),
// with four deliberate differences:
//   * buffers are real HBM (is_host = false): weights are uploaded once by set_tensor and never re-marshalled
//     (the reference re-maps tensor memory through FastRPC per call, ggml-hexagon.cpp:4975-5060);
//   * op failures are reported as GGML_STATUS_FAILED (the reference logs and continues, :5053-5056);
//   * graph_compute schedules a split instead of walking it: MUL_MATs that share src1 run as one launch although llama.cpp's
//     graph order separates them, and chains of nodes (norm -> weight -> projections, projection -> residual add, gate / up ->
//     SwiGLU, rope -> KV-cache stores, KQ -> soft_max -> KQV -> head merge) run as single launches;
//   * pinned host buffers, asynchronous copies and events, so llama.cpp's pipelined loader applies.
// Row split over the devices of this process (llama.cpp -sm row) is the split buffer type at the end of the file.
//
// This file includes only ggml headers and the C-ABI; all HIP lives in libggml_mi355x_qmm.so.

#include "ggml-mi355x.h"
#include "ggml-backend-impl.h"
#include "ggml-impl.h"
#include "ggml_mi355x_qmm.h"
#include "ggml_mi355x_ops.h"

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <atomic>
#include <map>
#include <set>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct mi355x_device_ctx {
    int         ordinal = 0;
    qmm_ctx *   qmm = nullptr;
    std::string name;
    std::string description;
    ggml_backend_buffer_type buft{};
    std::string buft_name;
    // row split: staging for the copy of src1 and for this device's slice of dst when another device is the root of the op
    void *      stage_x = nullptr;
    size_t      stage_x_bytes = 0;
    void *      stage_d = nullptr;
    size_t      stage_d_bytes = 0;
    qmm_event * ev_done = nullptr;       // this device's slice has landed in the root's dst
    qmm_event * ev_ready = nullptr;      // (as root) src1 is ready on the root's stream
    std::vector<void *> retired;         // staging blocks outgrown while queued work may still read them: freed behind the next synchronize
    // Small set_tensor calls (the per-token inputs llama.cpp writes in front of every graph: token ids, positions, the KQ mask, the
    // output ids) go through a pinned ring and an asynchronous copy on the device's stream instead of a blocking copy each: the call
    // returns when the bytes are in the ring; everything queued later on the stream sees them, and every other way to the memory
    // (get_tensor, cpy_tensor, memset, clear: they use the NULL stream) settles the stream first.  GGML_MI355X_STAGED_SET=0: off.
    char *      ring = nullptr;
    size_t      ring_bytes = 0, ring_pos = 0;
    bool        staged_pending = false;
    std::mutex  ring_mu;
    // what this module has queued on the device's stream / what a synchronize has waited for: ggml_backend_sched synchronizes a backend
    // in front of every split input and behind every graph (six calls per generated token, five of them with nothing queued since the
    // last one); a wait on an idle stream still costs ~9 us of host time (round 3, GGML_MI355X_TIMING), so those return at once
    std::atomic<uint64_t> enq{0}, enq_synced{0};
    // prompt batches already run in QMM_PREC_BF16 (GGML_MI355X_PREC=bf16, or a first prompt met a weight block beyond the f16 range)
    bool        prefill_bf16 = [] { const char * e = getenv("GGML_MI355X_PREC"); return e && (!strcmp(e, "bf16") || !strcmp(e, "0")); }();
};

// SURVEY 8f-2, weight repack: Q4_0 / Q8_0 / Q6_K weight tensors are re-laid into aligned planes (qmm_repack_rows) the first time
// a MUL_MAT / MUL_MAT_ID reads them, in place, row by row: sizes and strides do not change, so nothing of ggml-alloc's view moves.
// (The AMX buffer type converts inside set_tensor and has no get_tensor, ggml/src/ggml-cpu/amx/amx.cpp; here llama.cpp's pipelined
// loader writes tensors in arbitrary byte chunks, so the conversion waits for the first use instead, and get_tensor, cpy_tensor and
// partial writes convert a tensor BACK to GGUF wire layout first: what leaves the buffer is always wire bytes.)
struct planar_rec { int wire_type; int64_t K, rows, row_bytes; size_t bytes; };
struct mi355x_buffer_ctx {
    mi355x_device_ctx * dev;
    void *              base;
    std::mutex                          mu;
    std::map<const char *, planar_rec>  planar;     // by the tensor's first byte
    std::set<const char *>              wire_only;  // weights that went back to wire layout for good (a view cut their rows)
};

struct mi355x_backend_ctx {
    mi355x_device_ctx * dev;
    std::string         name;
    qmm_event *         ev_copy = nullptr;   // cpy_tensor_async: "src is ready" on the source backend's stream
    // per-graph reader analysis (graph_compute): for every candidate tensor, who reads its memory in this graph
    struct reader_info { const ggml_tensor * t; int uses; int last_reader; bool glue_only; };
    std::vector<reader_info>         readers;
    uint64_t                         readers_sig = 0;        // signature of the graph `readers` was computed for (analyze_readers)
    int                              readers_sig_nodes = -1;
    // results of hoisted MUL_MATs that could not be written in place (their block of the compute buffer is still in use at
    // the earlier point): they live in `hoist_buf` and every reader gets the pointer swapped in to_qt
    struct redirect { const ggml_tensor * t; char * data; int last_reader; };
    std::vector<redirect>            redirects;
    void *                           hoist_buf = nullptr;
    size_t                           hoist_bytes = 0, hoist_used = 0;
    std::map<uintptr_t, uintptr_t>   later_ranges;   // analyze_readers: union of the byte ranges of the nodes behind the one looked at
    std::vector<const ggml_tensor *> skipped;
    std::vector<char>                done;
    std::vector<const ggml_tensor *> deferred;       // per node: the SILU whose result this MUL consumes in the same launch
    // RMS_NORM -> MUL(w) held back for the MUL_MATs that read it (few-token batches): they form the normed row while staging
    struct swiglu_src { const float * gate = nullptr; const float * up = nullptr; int64_t ld_gate = 0, ld_up = 0; };
    std::vector<swiglu_src>          swiglu_in;      // per node: this ffn_down forms silu(gate) * up in its activation prep (prompt batches)
    struct norm_req { const ggml_tensor * rn = nullptr, * mul = nullptr, * w = nullptr; int readers = 0; const ggml_tensor * add = nullptr; };   // add: the residual ADD in front of the norm (prompt batches)
    norm_req                         pending_norm;
    // GGML_MI355X_TIMING=1: stream time of every graph (an event pair around its launches), summed per kind of graph
    qmm_event *                      ev_t0 = nullptr, * ev_t1 = nullptr;
    double                           ms_tg = 0, ms_pp = 0, ms_pp_min = 0;      // (min: the warm-up pass also repacks weights at their first use)
    int64_t                          graphs_tg = 0, graphs_pp = 0, tokens_pp = 0;
    // ... and where the HOST's time goes around one-token graphs (wall clock, us): between two graph_compute calls (libllama: graph
    // build, scheduler, input copies, sampling-free bookkeeping), in the reader analysis, in the issue loop, waiting in synchronize
    double                           us_outside = 0, us_analyze = 0, us_issue = 0, us_wait = 0, t_exit = 0;
    double                           pp_outside = 0, pp_analyze = 0, pp_issue = 0, pp_wait = 0, t_exit_pp = 0;   // the same for prompt graphs
    int64_t                          pp_outside_n = 0;
};

// GGML_MI355X_GLUE=0: offload the quantized MUL_MAT / MUL_MAT_ID only (the round-1 surface); GGML_MI355X_FUSE=0: no fused pairs
bool GGML_MI355X_GLUE_OFF() {
    static const bool off = [] { const char * e = getenv("GGML_MI355X_GLUE"); return e && atoi(e) == 0; }();
    return off;
}
bool GGML_MI355X_ATTN_ROPE() {
    static const bool on = [] { const char * e = getenv("GGML_MI355X_ATTN_ROPE"); return !(e && atoi(e) == 0); }();
    return on;
}
bool GGML_MI355X_TIMING() {
    static const bool on = [] { const char * e = getenv("GGML_MI355X_TIMING"); return e && atoi(e) != 0; }();
    return on;
}
// GGML_MI355X_CHAIN=1: one-token MUL_MAT groups with nothing between them (wo -> ffn_gate/up -> ffn_down -> next wq/wk/wv once
// norm, residual and SwiGLU are folded in) go out as one persistent launch (qmm_chain_*); measured slower than launches so far
bool GGML_MI355X_CHAIN() {
    static const bool on = [] { const char * e = getenv("GGML_MI355X_CHAIN"); return e && atoi(e) != 0; }();
    return on;
}
bool GGML_MI355X_REPACK() {
    static const bool on = [] { const char * e = getenv("GGML_MI355X_REPACK"); return !(e && atoi(e) == 0); }();
    return on;
}
int g_fuse_override = -1;            // -1: the environment decides; 0 / 1: set through the "ggml_backend_mi355x_set_fuse" proc address (tests)
bool GGML_MI355X_FUSE_OFF() {
    static const bool off = [] { const char * e = getenv("GGML_MI355X_FUSE"); return e && atoi(e) == 0; }();
    return g_fuse_override >= 0 ? g_fuse_override == 0 : off;
}
void set_fuse(int on) { g_fuse_override = on < 0 ? -1 : on != 0; }

mi355x_device_ctx      g_devs[GGML_MI355X_MAX_DEVICES];
ggml_backend_device    g_devices[GGML_MI355X_MAX_DEVICES];
int                    g_ndev = 0;

bool type_supported(enum ggml_type t) {
    // the north-star's five formats plus SURVEY 8f-4's (round 2): any stock Q4_1 / Q5_0 / Q5_1 / Q2_K / Q3_K_* / IQ4_NL GGUF keeps its
    // matmul weights on the device; the kernel library answers the same question through qmm_row_size() != 0
    return t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q8_0 || t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q5_K || t == GGML_TYPE_Q6_K ||
           t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q5_1 || t == GGML_TYPE_Q2_K || t == GGML_TYPE_Q3_K || t == GGML_TYPE_IQ4_NL || t == GGML_TYPE_IQ4_XS;
}


// ----------------------------------------------------------------------------------------------- planar weights

// every planar tensor that overlaps [p, p + size) goes back to wire layout (unless the range covers it and `overwritten`)
void planar_release(mi355x_buffer_ctx * bc, const char * p, size_t size, bool overwritten, void * stream) {
    std::lock_guard<std::mutex> lock(bc->mu);
    for (auto it = bc->planar.begin(); it != bc->planar.end();) {
        const char * t0 = it->first;
        const planar_rec & r = it->second;
        if (t0 < p + size && p < t0 + r.bytes) {
            const bool covered = overwritten && p <= t0 && t0 + r.bytes <= p + size;
            if (!covered) {
                // the buffer-level callers pass the NULL stream, and the device's own stream is non-blocking: launches queued
                // there that still read the planar rows (or the to-planar repack itself) must be through before the rows are re-laid
                void * own = qmm_stream(bc->dev->qmm);
                if (stream != own && qmm_synchronize(bc->dev->qmm, own)) GGML_LOG_ERROR("MI355X: %s\n", qmm_last_error());
                if (qmm_repack_rows(bc->dev->qmm, r.wire_type, (void *) t0, r.row_bytes, r.rows, r.K, 0, stream))
                    GGML_ABORT("MI355X: converting a planar weight back to wire layout failed: %s", qmm_last_error());
                // (a failing synchronize here reports conditions of OTHER work as well: a non-finite prefill, a bad expert id; they are
                // logged where they belong, not turned into an abort of this copy)
                if (qmm_synchronize(bc->dev->qmm, stream)) GGML_LOG_ERROR("MI355X: %s\n", qmm_last_error());
            }
            it = bc->planar.erase(it);
        } else {
            ++it;
        }
    }
}
const char * buft_get_name(ggml_backend_buffer_type_t buft);
// the tensor's buffer context when it lives in an ordinary (not split, not host) MI355X buffer
mi355x_buffer_ctx * our_buffer_ctx(const ggml_tensor * t) {
    ggml_backend_buffer_t b = t->view_src ? t->view_src->buffer : t->buffer;
    return b && b->buft->iface.get_name == buft_get_name ? (mi355x_buffer_ctx *) b->context : nullptr;
}
// type code for the kernel library: the planar code when the tensor's rows are in planar form (views: same rows as the root)
int dev_type(const ggml_tensor * t) {
    const ggml_tensor * root = t->view_src ? t->view_src : t;
    if (root->type != GGML_TYPE_Q4_0 && root->type != GGML_TYPE_Q8_0 && root->type != GGML_TYPE_Q6_K) return (int) t->type;
    mi355x_buffer_ctx * bc = our_buffer_ctx(root);
    if (!bc) return (int) t->type;
    std::lock_guard<std::mutex> lock(bc->mu);
    auto it = bc->planar.find((const char *) root->data);
    if (it == bc->planar.end()) return (int) t->type;
    if (t->ne[0] != root->ne[0] || t->nb[1] != root->nb[1]) {
        // a view that cuts rows of a planar weight (nothing in llama.cpp builds one; ADVICE r2: no abort): the weight goes back to wire
        // layout for good and the view is served from that
        GGML_LOG_WARN("MI355X: %s views part of the rows of %s; the weight returns to wire layout\n", t->name, root->name);
        const planar_rec r = it->second;
        void * own = qmm_stream(bc->dev->qmm);
        if (qmm_repack_rows(bc->dev->qmm, r.wire_type, (void *) it->first, r.row_bytes, r.rows, r.K, 0, own))
            GGML_ABORT("MI355X: converting a planar weight back to wire layout failed: %s", qmm_last_error());
        bc->wire_only.insert(it->first);
        bc->planar.erase(it);
        return (int) t->type;
    }
    return (int) t->type + 100;
}
// first use as a MUL_MAT / MUL_MAT_ID weight: convert the whole (root) tensor, on the compute stream, in front of the launch
int weight_type(mi355x_backend_ctx * ctx, const ggml_tensor * t) {
    const ggml_tensor * root = t->view_src ? t->view_src : t;
    mi355x_buffer_ctx * bc = GGML_MI355X_REPACK() ? our_buffer_ctx(root) : nullptr;
    // weights only: a quantized tensor in a compute or KV buffer (a Q8_0 K cache) is rewritten all the time; re-laying it
    // in place behind its writers' backs would corrupt it
    if (bc && root->buffer && root->buffer->usage != GGML_BACKEND_BUFFER_USAGE_WEIGHTS) bc = nullptr;
    const int pt = bc ? qmm_planar_type((int) root->type, root->ne[0], (int64_t) root->nb[1]) : 0;
    if (!pt || !ggml_is_contiguous(root) || (uintptr_t) root->data % 16 || t->ne[0] != root->ne[0] || t->nb[1] != root->nb[1]) return dev_type(t);
    {
        std::lock_guard<std::mutex> lock(bc->mu);
        if (bc->planar.count((const char *) root->data)) return pt;
        if (bc->wire_only.count((const char *) root->data)) return (int) t->type;
        const int64_t rows = ggml_nrows(root);
        if (qmm_repack_rows(ctx->dev->qmm, (int) root->type, root->data, (int64_t) root->nb[1], rows, root->ne[0], 1, qmm_stream(ctx->dev->qmm))) {
            GGML_LOG_WARN("MI355X: repack of %s refused (%s); it stays in wire layout\n", root->name, qmm_last_error());
            return (int) t->type;
        }
        bc->planar[(const char *) root->data] = planar_rec{ (int) root->type, root->ne[0], rows, (int64_t) root->nb[1], ggml_nbytes(root) };
    }
    return pt;
}

// ----------------------------------------------------------------------------------------------- buffer

void settle(mi355x_device_ctx * d);
void buffer_free(ggml_backend_buffer_t buffer) {
    auto * ctx = (mi355x_buffer_ctx *) buffer->context;
    settle(ctx->dev);
    qmm_free(ctx->dev->qmm, ctx->base);
    delete ctx;
}
void * buffer_get_base(ggml_backend_buffer_t buffer) { return ((mi355x_buffer_ctx *) buffer->context)->base; }

enum ggml_status buffer_init_tensor(ggml_backend_buffer_t, struct ggml_tensor *) { return GGML_STATUS_SUCCESS; }

void buffer_memset_tensor(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor, uint8_t value, size_t offset, size_t size) {
    auto * ctx = (mi355x_buffer_ctx *) buffer->context;
    settle(ctx->dev);
    planar_release(ctx, (const char *) tensor->data + offset, size, true, nullptr);
    if (qmm_memset(ctx->dev->qmm, (char *) tensor->data + offset, value, size, nullptr) || qmm_synchronize(ctx->dev->qmm, nullptr))
        GGML_ABORT("MI355X memset_tensor: %s", qmm_last_error());
}
// GGML_MI355X_TIMING=1: host wall time inside the module's transfer / synchronize entry points (what of libllama's time between two
// graphs is spent here), summed per entry point
bool GGML_MI355X_TIMING();
struct host_timer {
    static constexpr int N = 8;
    static double us[N], pend_us[N], tg_us[N], pp_us[N]; // all calls; calls since the last graph_compute; calls in front of one-token / prompt graphs
    static long long calls[N], pend_calls[N], tg_calls[N], pp_calls[N];
    static int pp_seen;
    static void flush(bool one_token) {
        for (int i = 0; i < N; ++i) {
            if (one_token) { tg_us[i] += pend_us[i]; tg_calls[i] += pend_calls[i]; }
            else if (pp_seen >= 2) { pp_us[i] += pend_us[i]; pp_calls[i] += pend_calls[i]; }      // (in front of the first two prompt graphs: the model's upload, first-use initialisations)
            pend_us[i] = 0; pend_calls[i] = 0;
        }
        if (!one_token) ++pp_seen;
    }
    static const char * name(int i) { static const char * n[N] = { "set_tensor", "get_tensor", "cpy_tensor", "set_tensor_async", "get_tensor_async", "cpy_tensor_async", "synchronize", "event" }; return n[i]; }
    int slot; double t0;
    explicit host_timer(int s) : slot(s), t0(GGML_MI355X_TIMING() ? std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0) {}
    ~host_timer() {
        if (t0 > 0) {
            const double dt = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
            us[slot] += dt; calls[slot]++; pend_us[slot] += dt; pend_calls[slot]++;
        }
    }
};
int host_timer::pp_seen = 0;
double host_timer::us[host_timer::N] = { 0 }, host_timer::pend_us[host_timer::N] = { 0 }, host_timer::tg_us[host_timer::N] = { 0 }, host_timer::pp_us[host_timer::N] = { 0 };
long long host_timer::calls[host_timer::N] = { 0 }, host_timer::pend_calls[host_timer::N] = { 0 }, host_timer::tg_calls[host_timer::N] = { 0 }, host_timer::pp_calls[host_timer::N] = { 0 };

constexpr size_t STAGED_SET_MAX = (size_t) 256 << 10, STAGED_RING = (size_t) 4 << 20;
bool staged_set_on() { static const bool on = [] { const char * e = getenv("GGML_MI355X_STAGED_SET"); return !(e && atoi(e) == 0); }(); return on; }
// every path to device memory that does not run on the device's stream waits for the staged copies first
void settle(mi355x_device_ctx * d) {
    std::lock_guard<std::mutex> lock(d->ring_mu);
    if (d->staged_pending) {
        if (qmm_synchronize(d->qmm, qmm_stream(d->qmm))) GGML_LOG_ERROR("MI355X: %s\n", qmm_last_error());
        d->staged_pending = false;
    }
}
bool staged_set(mi355x_buffer_ctx * ctx, void * dst, const void * data, size_t size) {
    mi355x_device_ctx * d = ctx->dev;
    std::lock_guard<std::mutex> lock(d->ring_mu);
    if (!d->ring) {
        d->ring = (char *) qmm_host_malloc(d->qmm, STAGED_RING);
        if (!d->ring) return false;
        d->ring_bytes = STAGED_RING;
    }
    const size_t need = (size + 255) & ~(size_t) 255;
    if (d->ring_pos + need > d->ring_bytes) {                  // wrap: the slots in front are reused only behind the copies that read them
        if (qmm_synchronize(d->qmm, qmm_stream(d->qmm))) return false;
        d->ring_pos = 0;
    }
    memcpy(d->ring + d->ring_pos, data, size);
    if (qmm_memcpy_h2d_async(d->qmm, dst, d->ring + d->ring_pos, size, qmm_stream(d->qmm))) return false;
    d->ring_pos += need;
    d->staged_pending = true;       // (not counted in enq: the copy is ordered in front of everything queued later, its source is the ring, and every
    return true;                    //  other way to the destination settles the stream first: a synchronize need not wait for it)
}

void buffer_set_tensor(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    host_timer timer_(0);
    auto * ctx = (mi355x_buffer_ctx *) buffer->context;
    if (size <= STAGED_SET_MAX && buffer->usage != GGML_BACKEND_BUFFER_USAGE_WEIGHTS && staged_set_on()) {
        bool empty;
        { std::lock_guard<std::mutex> lock(ctx->mu); empty = ctx->planar.empty(); }
        if (empty && staged_set(ctx, (char *) tensor->data + offset, data, size)) return;
    }
    settle(ctx->dev);
    planar_release(ctx, (const char *) tensor->data + offset, size, true, nullptr);       // wire bytes come in: the tensor is wire again
    if (qmm_memcpy_h2d(ctx->dev->qmm, (char *) tensor->data + offset, data, size, nullptr))
        GGML_ABORT("MI355X set_tensor: %s", qmm_last_error());
}
void buffer_get_tensor(ggml_backend_buffer_t buffer, const struct ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    host_timer timer_(1);
    auto * ctx = (mi355x_buffer_ctx *) buffer->context;
    settle(ctx->dev);
    planar_release(ctx, (const char *) tensor->data + offset, size, false, nullptr);      // wire bytes go out (converted again at the next use)
    if (qmm_memcpy_d2h(ctx->dev->qmm, data, (const char *) tensor->data + offset, size, nullptr))
        GGML_ABORT("MI355X get_tensor: %s", qmm_last_error());
}
const char * buft_get_name(ggml_backend_buffer_type_t buft);
bool buffer_cpy_tensor(ggml_backend_buffer_t buffer, const struct ggml_tensor * src, struct ggml_tensor * dst) {
    host_timer timer_(2);
    auto * ctx = (mi355x_buffer_ctx *) buffer->context;
    settle(ctx->dev);
    ggml_backend_buffer_t sb = src->view_src ? src->view_src->buffer : src->buffer;
    if (!sb || sb->buft->iface.get_name != buft_get_name) return false;           // not one of ours: let ggml stage through the host
    auto * sctx = (mi355x_buffer_ctx *) sb->context;
    if (sctx->dev != ctx->dev) return false;
    settle(sctx->dev);
    planar_release(sctx, (const char *) src->data, ggml_nbytes(src), false, nullptr);
    planar_release(ctx, (const char *) dst->data, ggml_nbytes(src), true, nullptr);
    if (qmm_memcpy_d2d(ctx->dev->qmm, dst->data, src->data, ggml_nbytes(src), nullptr) || qmm_synchronize(ctx->dev->qmm, nullptr))
        GGML_ABORT("MI355X cpy_tensor: %s", qmm_last_error());
    return true;
}
void buffer_clear(ggml_backend_buffer_t buffer, uint8_t value) {
    auto * ctx = (mi355x_buffer_ctx *) buffer->context;
    settle(ctx->dev);
    { std::lock_guard<std::mutex> lock(ctx->mu); ctx->planar.clear(); }
    if (qmm_memset(ctx->dev->qmm, ctx->base, value, buffer->size, nullptr) || qmm_synchronize(ctx->dev->qmm, nullptr))
        GGML_ABORT("MI355X clear: %s", qmm_last_error());
}

const ggml_backend_buffer_i buffer_iface = {
    /* .free_buffer   = */ buffer_free,
    /* .get_base      = */ buffer_get_base,
    /* .init_tensor   = */ buffer_init_tensor,
    /* .memset_tensor = */ buffer_memset_tensor,
    /* .set_tensor    = */ buffer_set_tensor,
    /* .get_tensor    = */ buffer_get_tensor,
    /* .cpy_tensor    = */ buffer_cpy_tensor,
    /* .clear         = */ buffer_clear,
    /* .reset         = */ nullptr,
};

// ----------------------------------------------------------------------------------------------- buffer type

const char * buft_get_name(ggml_backend_buffer_type_t buft) { return ((mi355x_device_ctx *) buft->context)->buft_name.c_str(); }

ggml_backend_buffer_t buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    auto * dev = (mi355x_device_ctx *) buft->context;
    void * p = qmm_malloc(dev->qmm, size + 256);        // +256: vector loads of the last weight row never leave the allocation
    if (!p) {
        GGML_LOG_ERROR("%s: allocating %.2f MiB on %s failed: %s\n", __func__, size / 1048576.0, dev->name.c_str(), qmm_last_error());
        return nullptr;
    }
    auto * bc = new mi355x_buffer_ctx;
    bc->dev = dev;
    bc->base = p;
    return ggml_backend_buffer_init(buft, buffer_iface, bc, size);
}
size_t buft_get_alignment(ggml_backend_buffer_type_t) { return 256; }
size_t buft_get_max_size(ggml_backend_buffer_type_t buft) {
    auto * dev = (mi355x_device_ctx *) buft->context;
    size_t f = 0, t = 0;
    qmm_device_info(dev->qmm, nullptr, 0, &f, &t, nullptr);
    return t;
}
bool buft_is_host(ggml_backend_buffer_type_t) { return false; }

const ggml_backend_buffer_type_i buft_iface = {
    /* .get_name       = */ buft_get_name,
    /* .alloc_buffer   = */ buft_alloc_buffer,
    /* .get_alignment  = */ buft_get_alignment,
    /* .get_max_size   = */ buft_get_max_size,
    /* .get_alloc_size = */ nullptr,
    /* .is_host        = */ buft_is_host,
};

// ----------------------------------------------------------------------------------------------- host (pinned) buffer type
// with caps {async, host_buffer, events} llama.cpp's loader uploads through four pinned staging buffers and
// events (src/llama-model-loader.cpp:904-1079) and the scheduler gives the CPU backend's compute buffer this type, so the
// per-token inputs leave from page-locked memory.  The buffer itself is an ordinary CPU buffer over hipHostMalloc'ed memory.

const char * host_buft_get_name(ggml_backend_buffer_type_t) { return GGML_MI355X_BACKEND_NAME "_Host"; }
void host_buffer_free(ggml_backend_buffer_t buffer) { qmm_host_free(g_devs[0].qmm, buffer->context); }
ggml_backend_buffer_t host_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    void * p = getenv("GGML_MI355X_NO_PINNED") ? nullptr : qmm_host_malloc(g_devs[0].qmm, size);
    if (!p) return ggml_backend_buft_alloc_buffer(ggml_backend_cpu_buffer_type(), size);      // pageable memory still works, just slower
    ggml_backend_buffer_t buffer = ggml_backend_cpu_buffer_from_ptr(p, size);
    buffer->buft = buft;
    buffer->iface.free_buffer = host_buffer_free;
    return buffer;
}
ggml_backend_buffer_type_t host_buffer_type() {
    static ggml_backend_buffer_type buft = {
        { host_buft_get_name, host_buft_alloc_buffer, ggml_backend_cpu_buffer_type()->iface.get_alignment, nullptr,
          ggml_backend_cpu_buffer_type()->iface.get_alloc_size, ggml_backend_cpu_buffer_type()->iface.is_host },
        &g_devices[0], nullptr };
    return &buft;
}

// ----------------------------------------------------------------------------------------------- split buffer type
// ggml row split inside one process (ggml-cuda.cu:727-1052): a weight tensor's rows are divided over the devices by the
// cumulative fractions of `tensor_split`, boundaries rounded down to 64 rows, the last device takes the remainder
// (get_row_split :740-753; ggml-hexagon_amd/rowsplit.py is the same rule for the one-process-per-GPU path).  The buffer
// owns one allocation per (tensor, device); tensor->data is a dummy and tensor->extra points at the slices.
// Only whole-tensor set_tensor / get_tensor, as in the reference (:800-885).

constexpr int64_t SPLIT_ROW_ROUNDING = 256;    // the prefill kernels' row tile (every device's slice starts on a tile; round 1 used 64: ragged tiles on every device)

struct split_buft_ctx {
    int         main_device;
    std::array<float, GGML_MI355X_MAX_DEVICES> split;     // cumulative start fraction per device
    std::string name;
};
struct split_extra {
    void *  data[GGML_MI355X_MAX_DEVICES] = {};
    int64_t lo[GGML_MI355X_MAX_DEVICES] = {}, hi[GGML_MI355X_MAX_DEVICES] = {};
};
struct split_buffer_ctx {
    std::vector<split_extra *> extras;
};

// the rounding of one matrix: SPLIT_ROW_ROUNDING where every device still gets rows under an even split, else the largest power of two
// (>= 32) that leaves none empty: a 1024-row wk / wv over 8 devices would otherwise come out as 0 / 256 / 0 / 256 ... rows (ADVICE r2:
// half the devices idle on it); ggml-hexagon_amd/rowsplit.py rounding_for is the same rule for the one-process-per-GPU path
int64_t split_rounding(int64_t nrows) {
    int64_t r = SPLIT_ROW_ROUNDING;
    while (r > 32 && nrows / std::max(g_ndev, 1) < r) r /= 2;
    return r;
}
void split_row_range(const split_buft_ctx * c, int64_t nrows, int id, int64_t * lo, int64_t * hi) {
    const int64_t rounding = split_rounding(nrows);
    *lo = id == 0 ? 0 : (int64_t) (nrows * c->split[id]);
    *lo -= *lo % rounding;
    if (id == g_ndev - 1) {
        *hi = nrows;
    } else {
        *hi = (int64_t) (nrows * c->split[id + 1]);
        *hi -= *hi % rounding;
    }
    if (*hi < *lo) *hi = *lo;
}

const char * split_buft_get_name(ggml_backend_buffer_type_t buft) { return ((split_buft_ctx *) buft->context)->name.c_str(); }
bool buft_is_split(ggml_backend_buffer_type_t buft) { return buft->iface.get_name == split_buft_get_name; }

void split_buffer_free(ggml_backend_buffer_t buffer) {
    auto * ctx = (split_buffer_ctx *) buffer->context;
    for (split_extra * e : ctx->extras) {
        for (int id = 0; id < g_ndev; ++id)
            if (e->data[id]) qmm_free(g_devs[id].qmm, e->data[id]);
        delete e;
    }
    delete ctx;
}
void * split_buffer_get_base(ggml_backend_buffer_t) { return (void *) 0x1000; }      // never dereferenced (ggml-cuda.cu:793-798)

enum ggml_status split_buffer_init_tensor(ggml_backend_buffer_t buffer, struct ggml_tensor * tensor) {
    GGML_ASSERT(tensor->view_src == nullptr);                      // views of split tensors are not supported (:803)
    GGML_ASSERT(ggml_is_contiguous(tensor) && tensor->ne[2] == 1 && tensor->ne[3] == 1);
    auto * ctx  = (split_buffer_ctx *) buffer->context;
    auto * bctx = (split_buft_ctx *) buffer->buft->context;
    auto * e = new split_extra;
    ctx->extras.push_back(e);
    const size_t row_bytes = ggml_row_size(tensor->type, tensor->ne[0]);
    for (int id = 0; id < g_ndev; ++id) {
        split_row_range(bctx, tensor->ne[1], id, &e->lo[id], &e->hi[id]);
        const int64_t rows = e->hi[id] - e->lo[id];
        if (rows == 0) continue;
        e->data[id] = qmm_malloc(g_devs[id].qmm, rows * row_bytes + 256);
        if (!e->data[id]) {
            GGML_LOG_ERROR("%s: %s: %.2f MiB on %s: %s\n", __func__, tensor->name, rows * row_bytes / 1048576.0, g_devs[id].name.c_str(), qmm_last_error());
            return GGML_STATUS_ALLOC_FAILED;
        }
    }
    tensor->extra = e;
    return GGML_STATUS_SUCCESS;
}
void split_buffer_set_tensor(ggml_backend_buffer_t, struct ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    GGML_ASSERT(offset == 0 && size == ggml_nbytes(tensor));      // split tensors are set in one go (:839-841)
    auto * e = (split_extra *) tensor->extra;
    const size_t row_bytes = ggml_row_size(tensor->type, tensor->ne[0]);
    for (int id = 0; id < g_ndev; ++id) {
        if (!e->data[id]) continue;
        if (qmm_memcpy_h2d(g_devs[id].qmm, e->data[id], (const char *) data + e->lo[id] * row_bytes, (e->hi[id] - e->lo[id]) * row_bytes, nullptr))
            GGML_ABORT("MI355X split set_tensor: %s", qmm_last_error());
    }
}
void split_buffer_get_tensor(ggml_backend_buffer_t, const struct ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    GGML_ASSERT(offset == 0 && size == ggml_nbytes(tensor));
    auto * e = (split_extra *) tensor->extra;
    const size_t row_bytes = ggml_row_size(tensor->type, tensor->ne[0]);
    for (int id = 0; id < g_ndev; ++id) {
        if (!e->data[id]) continue;
        if (qmm_memcpy_d2h(g_devs[id].qmm, (char *) data + e->lo[id] * row_bytes, e->data[id], (e->hi[id] - e->lo[id]) * row_bytes, nullptr))
            GGML_ABORT("MI355X split get_tensor: %s", qmm_last_error());
    }
}
void split_buffer_clear(ggml_backend_buffer_t, uint8_t) {}

const ggml_backend_buffer_i split_buffer_iface = {
    /* .free_buffer   = */ split_buffer_free,
    /* .get_base      = */ split_buffer_get_base,
    /* .init_tensor   = */ split_buffer_init_tensor,
    /* .memset_tensor = */ nullptr,
    /* .set_tensor    = */ split_buffer_set_tensor,
    /* .get_tensor    = */ split_buffer_get_tensor,
    /* .cpy_tensor    = */ nullptr,
    /* .clear         = */ split_buffer_clear,
    /* .reset         = */ nullptr,
};

ggml_backend_buffer_t split_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    // the slices are allocated per tensor in init_tensor; `size` only keeps ggml-alloc's offsets apart (:948-956)
    return ggml_backend_buffer_init(buft, split_buffer_iface, new split_buffer_ctx, size);
}
size_t split_buft_get_alignment(ggml_backend_buffer_type_t) { return 128; }
size_t split_buft_get_alloc_size(ggml_backend_buffer_type_t, const struct ggml_tensor * tensor) {
    return ggml_nbytes(tensor) + (size_t) 256 * GGML_MI355X_MAX_DEVICES;
}
bool split_buft_is_host(ggml_backend_buffer_type_t) { return false; }

const ggml_backend_buffer_type_i split_buft_iface = {
    /* .get_name       = */ split_buft_get_name,
    /* .alloc_buffer   = */ split_buft_alloc_buffer,
    /* .get_alignment  = */ split_buft_get_alignment,
    /* .get_max_size   = */ nullptr,
    /* .get_alloc_size = */ split_buft_get_alloc_size,
    /* .is_host        = */ split_buft_is_host,
};

// ----------------------------------------------------------------------------------------------- ops

bool is_ours(const struct ggml_tensor * t) {
    ggml_backend_buffer_t b = t->view_src ? t->view_src->buffer : t->buffer;
    return b && b->buft->iface.get_name == buft_get_name;
}

// src0 rows must be whole rows of blocks, src1/dst dense f32 rows
bool mul_mat_shape_ok(const struct ggml_tensor * op) {
    const ggml_tensor * a = op->src[0], * b = op->src[1];
    if (!a || !b || !type_supported(a->type) || b->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32) return false;
    if (a->ne[0] % ggml_blck_size(a->type) || a->ne[0] != b->ne[0]) return false;
    if (a->nb[0] != ggml_type_size(a->type) || a->nb[1] < ggml_row_size(a->type, a->ne[0])) return false;
    if (a->nb[1] % 2 || a->nb[2] % 2 || a->nb[3] % 2) return false;
    if (b->nb[0] != sizeof(float) || b->nb[1] % 16 || b->nb[2] % 16 || b->nb[3] % 16) return false;
    if (op->nb[0] != sizeof(float) || op->nb[1] % 4 || op->nb[1] < op->ne[0] * sizeof(float)) return false;
    return true;
}

bool is_split(const struct ggml_tensor * t) { return t->buffer && buft_is_split(t->buffer->buft); }

bool supports_mul_mat(const struct ggml_tensor * op) {
    if (!mul_mat_shape_ok(op)) return false;
    const ggml_tensor * a = op->src[0], * b = op->src[1];
    if (is_split(a) && (a->ne[2] != 1 || a->ne[3] != 1 || b->ne[2] != 1 || b->ne[3] != 1)) return false;
    if (b->nb[1] < b->ne[0] * sizeof(float)) return false;                          // transposed / permuted src1
    if (b->ne[2] % a->ne[2] || b->ne[3] % a->ne[3]) return false;
    if (a->nb[2] < a->nb[1] * (size_t) a->ne[1] || (a->ne[3] > 1 && a->nb[3] < a->nb[2] * (size_t) a->ne[2])) return false;   // permuted src0
    return true;
}

bool supports_mul_mat_id(const struct ggml_tensor * op) {
    const ggml_tensor * as = op->src[0], * b = op->src[1], * ids = op->src[2];
    if (!mul_mat_shape_ok(op) || !ids || ids->type != GGML_TYPE_I32) return false;
    if (is_split(as)) return false;                                                // as in the reference tree (ggml-cuda.cu:1973)
    if (as->ne[3] != 1 || b->ne[3] != 1 || ids->ne[2] != 1 || ids->ne[3] != 1) return false;
    if (ids->nb[0] != sizeof(int32_t) || ids->nb[1] % 4) return false;
    if (b->ne[1] != 1 && b->ne[1] != ids->ne[0]) return false;                     // ne11 broadcast rule (ggml.c:2781-2808)
    if (as->nb[2] < as->nb[1] * (size_t) as->ne[1]) return false;
    if (op->nb[2] % 4) return false;
    return true;
}

// A staging block that has to grow: queued work may still read the old one, so it is retired, not freed (until round 3 this
// synchronized the stream in the middle of a graph: VERDICT r2 item 6); free_retired runs behind graph_compute's own synchronize.
bool grow(mi355x_device_ctx * d, void *& p, size_t & have, size_t need) {
    if (need <= have) return true;
    need = (need + ((size_t) 8 << 20) - 1) & ~(((size_t) 8 << 20) - 1);
    void * q = qmm_malloc(d->qmm, need);
    if (!q) return false;
    if (p) d->retired.push_back(p);
    p = q;
    have = need;
    return true;
}
void free_retired() {
    for (int id = 0; id < g_ndev; ++id) {
        for (void * p : g_devs[id].retired) qmm_free(g_devs[id].qmm, p);
        g_devs[id].retired.clear();
    }
}

// MUL_MAT with row-split src0 (ggml_cuda_op_mul_mat, ggml-cuda.cu:1365-1673).  The device running the node is the root: it
// holds src1 and dst.  The root computes its rows straight into dst[:, lo:hi]; every other device copies src1 over the
// fabric, computes its rows into a staging slice and copies that into the root's dst (N runs of `rows` floats).  All of it
// is stream-ordered: devices wait for the root's "src1 ready" event, the root waits for each device's "slice landed".
// The exchange is a concat, so there is no reduction and the result does not depend on the number of devices.
// GGML_MI355X_RCCL=1: the same exchange through RCCL (qmm_comm_*: ncclBroadcast of src1, grouped ncclSend / ncclRecv of the slices)
// instead of peer copies + events.  One communicator over all devices of the process, made at the first split MUL_MAT; where RCCL
// cannot serve (a library that is not there, logical devices that share a GPU) the peer-copy path stays, with one log line.
qmm_comm * g_comm = nullptr;
int        g_comm_state = 0;      // 0 = not tried, 1 = in use, -1 = unavailable
qmm_comm * split_comm() {
    if (g_comm_state == 0) {
        const char * e = getenv("GGML_MI355X_RCCL");
        g_comm_state = -1;
        if (e && atoi(e) && g_ndev >= 2) {
            std::vector<qmm_ctx *> cs;
            for (int i = 0; i < g_ndev; ++i) cs.push_back(g_devs[i].qmm);
            if (qmm_comm_create(cs.data(), g_ndev, &g_comm) == 0) g_comm_state = 1;
            else GGML_LOG_WARN("MI355X row split: RCCL exchange unavailable (%s); using peer copies\n", qmm_last_error());
        }
    }
    return g_comm_state == 1 ? g_comm : nullptr;
}

qmm_tensor to_qt(const ggml_tensor * t, const mi355x_backend_ctx * ctx);
// One exchange per GROUP (round 3; until then only single matrices took RCCL, so with fusion on q/k/v and gate/up stayed on peer
// copies: VERDICT r2 item 6): src1 is broadcast once, every device computes its slice of every member into its staging block (member
// after member, N x rows each), ONE grouped send / recv per device brings the blocks to the root, which places the slices.
enum ggml_status compute_mul_mat_split_rccl(mi355x_backend_ctx * ctx, const ggml_tensor * const * members, int n_members, qmm_comm * comm) {
    const ggml_tensor * dst0 = members[0], * b = dst0->src[1];
    mi355x_device_ctx * root = ctx->dev;
    const int root_id = (int) (root - &g_devs[0]);
    const int64_t K = dst0->src[0]->ne[0], N = b->ne[1];
    std::vector<void *> xs(g_ndev), recv(g_ndev, nullptr);
    std::vector<const void *> send(g_ndev, nullptr);
    std::vector<size_t> bytes(g_ndev, 0);
    auto rows_of = [&](int m, int id) { auto * e = (const split_extra *) members[m]->src[0]->extra; return e->hi[id] - e->lo[id]; };
    // a single one-token matrix: its slice IS a run of dst, received in place; everything else lands in the root's staging block
    const bool in_place = N == 1 && n_members == 1;
    size_t stage_total = 0;
    for (int id = 0; id < g_ndev; ++id) {
        for (int m = 0; m < n_members; ++m) bytes[id] += (size_t) N * rows_of(m, id) * sizeof(float);
        if (id != root_id) stage_total += bytes[id];
    }
    if (!in_place && !grow(root, root->stage_d, root->stage_d_bytes, stage_total)) goto fail;
    {
        size_t off = 0;
        for (int id = 0; id < g_ndev; ++id) {
            mi355x_device_ctx * d = &g_devs[id];
            if (id == root_id) { xs[id] = b->data; bytes[id] = 0; continue; }
            if (!grow(d, d->stage_x, d->stage_x_bytes, (size_t) N * K * sizeof(float)) || !grow(d, d->stage_d, d->stage_d_bytes, bytes[id])) goto fail;
            xs[id] = d->stage_x;
            send[id] = d->stage_d;
            recv[id] = in_place ? (void *) ((float *) to_qt(dst0, ctx).data + ((const split_extra *) dst0->src[0]->extra)->lo[id]) : (void *) ((char *) root->stage_d + off);
            off += bytes[id];
        }
        if (qmm_comm_broadcast(comm, root_id, xs.data(), (size_t) N * K * sizeof(float), nullptr)) goto fail;
        for (int id = 0; id < g_ndev; ++id) {
            mi355x_device_ctx * d = &g_devs[id];
            size_t moff = 0;
            for (int m = 0; m < n_members; ++m) {
                const ggml_tensor * a = members[m]->src[0];
                auto * e = (const split_extra *) a->extra;
                const int64_t rows = e->hi[id] - e->lo[id];
                if (rows == 0) continue;
                if (id == root_id) {
                    if (qmm_mul_mat(d->qmm, a->type, e->data[id], ggml_row_size(a->type, K), K, rows, (const float *) b->data, N, K,
                                    (float *) to_qt(members[m], ctx).data + e->lo[id], members[m]->nb[1] / sizeof(float), qmm_stream(d->qmm))) goto fail;
                } else if (qmm_mul_mat(d->qmm, a->type, e->data[id], ggml_row_size(a->type, K), K, rows, (const float *) d->stage_x, N, K,
                                       (float *) d->stage_d + moff, rows, qmm_stream(d->qmm))) goto fail;
                moff += (size_t) N * rows;
            }
        }
        if (qmm_comm_gather(comm, root_id, send.data(), recv.data(), bytes.data(), nullptr)) goto fail;
        if (!in_place) {                                       // place the slices: per member, N runs of `rows` floats each
            off = 0;
            for (int id = 0; id < g_ndev; ++id) {
                if (id == root_id) continue;
                size_t moff = 0;
                for (int m = 0; m < n_members; ++m) {
                    auto * e = (const split_extra *) members[m]->src[0]->extra;
                    const int64_t rows = e->hi[id] - e->lo[id];
                    if (rows == 0) continue;
                    if (qmm_memcpy2d_d2d(root->qmm, (float *) to_qt(members[m], ctx).data + e->lo[id], members[m]->nb[1], (char *) root->stage_d + off + moff * sizeof(float),
                                         rows * sizeof(float), rows * sizeof(float), N, qmm_stream(root->qmm))) goto fail;
                    moff += (size_t) N * rows;
                }
                off += bytes[id];
            }
        }
    }
    return GGML_STATUS_SUCCESS;
fail:
    GGML_LOG_ERROR("MI355X MUL_MAT(%s) row split over RCCL: %s\n", dst0->name, qmm_last_error());
    return GGML_STATUS_FAILED;
}

// Round 2: the MUL_MATs of a group (same src1: wq / wk / wv, ffn_gate / ffn_up) go out together: every other device gets src1 ONCE,
// computes its slice of each matrix, sends the slices back, and there is one event round trip per device and group instead of one per
// matrix (`members`: found by compute_mul_mat_split_group with the unsplit path's hoisting rules).
void free_retired();
qmm_tensor to_qt(const ggml_tensor * t, const mi355x_backend_ctx * ctx);
enum ggml_status compute_mul_mat_split(mi355x_backend_ctx * ctx, const ggml_tensor * const * members, int n_members) {
    const ggml_tensor * dst = members[0];
    const ggml_tensor * b = dst->src[1];
    mi355x_device_ctx * root = ctx->dev;
    void * rst = qmm_stream(root->qmm);
    const int64_t K = dst->src[0]->ne[0], N = b->ne[1], ldx = b->nb[1] / sizeof(float);
    if (ldx == K)                                             // (RCCL moves whole buffers: a strided src1 keeps the 2-D peer copies)
        if (qmm_comm * comm = split_comm()) return compute_mul_mat_split_rccl(ctx, members, n_members, comm);
    if (!root->ev_ready) root->ev_ready = qmm_event_create(root->qmm);
    if (!root->ev_ready || qmm_event_record(root->qmm, root->ev_ready, rst)) goto fail;
    for (int id = 0; id < g_ndev; ++id) {
        mi355x_device_ctx * d = &g_devs[id];
        size_t slice_floats = 0;
        for (int m = 0; m < n_members; ++m) {
            auto * e = (const split_extra *) members[m]->src[0]->extra;
            slice_floats += (size_t) N * (e->hi[id] - e->lo[id]);
        }
        if (slice_floats == 0) continue;
        if (d == root) {
            for (int m = 0; m < n_members; ++m) {
                const ggml_tensor * a = members[m]->src[0];
                auto * e = (const split_extra *) a->extra;
                const int64_t rows = e->hi[id] - e->lo[id];
                if (rows && qmm_mul_mat(root->qmm, a->type, e->data[id], ggml_row_size(a->type, K), K, rows, (const float *) b->data, N, ldx,
                                        (float *) to_qt(members[m], ctx).data + e->lo[id], members[m]->nb[1] / sizeof(float), rst)) goto fail;
            }
            continue;
        }
        void * st = qmm_stream(d->qmm);
        if (!d->ev_done) d->ev_done = qmm_event_create(d->qmm);
        if (!d->ev_done || !grow(d, d->stage_x, d->stage_x_bytes, (size_t) N * K * sizeof(float)) ||
            !grow(d, d->stage_d, d->stage_d_bytes, slice_floats * sizeof(float))) goto fail;
        if (qmm_stream_wait_event(d->qmm, st, root->ev_ready) ||
            qmm_memcpy2d_d2d(d->qmm, d->stage_x, K * sizeof(float), b->data, b->nb[1], K * sizeof(float), N, st)) goto fail;
        {
            size_t off = 0;
            for (int m = 0; m < n_members; ++m) {
                const ggml_tensor * a = members[m]->src[0];
                auto * e = (const split_extra *) a->extra;
                const int64_t rows = e->hi[id] - e->lo[id];
                if (rows == 0) continue;
                float * part = (float *) d->stage_d + off;
                if (qmm_mul_mat(d->qmm, a->type, e->data[id], ggml_row_size(a->type, K), K, rows, (const float *) d->stage_x, N, K, part, rows, st) ||
                    qmm_memcpy2d_d2d(d->qmm, (float *) to_qt(members[m], ctx).data + e->lo[id], members[m]->nb[1], part, rows * sizeof(float),
                                     rows * sizeof(float), N, st)) goto fail;
                off += (size_t) N * rows;
            }
        }
        if (qmm_event_record(d->qmm, d->ev_done, st) || qmm_stream_wait_event(root->qmm, rst, d->ev_done)) goto fail;
    }
    return GGML_STATUS_SUCCESS;
fail:
    GGML_LOG_ERROR("MI355X MUL_MAT(%s) row split: %s\n", dst->name, qmm_last_error());
    return GGML_STATUS_FAILED;
}

bool dbg();
void * hoist_elsewhere(mi355x_backend_ctx * ctx, const ggml_tensor * d);
bool is_noop(const ggml_tensor * node);
bool can_hoist(const ggml_tensor * t, const std::vector<const ggml_tensor *> & skipped);
bool supports_mul_mat(const struct ggml_tensor * op);
// nodes[0] is a MUL_MAT on row-split weights: later MUL_MATs on the same src1 with split weights join it where they may run early
enum ggml_status compute_mul_mat_split_group(mi355x_backend_ctx * ctx, ggml_tensor * const * nodes, int n_nodes, char * done) {
    const ggml_tensor * members[4] = { nodes[0], nullptr, nullptr, nullptr };
    int n = 1;
    const ggml_tensor * b = nodes[0]->src[1];
    std::vector<const ggml_tensor *> & skipped = ctx->skipped;
    skipped.clear();
    for (int i = 1; i < n_nodes && i <= 12 && n < 4 && !GGML_MI355X_FUSE_OFF(); ++i) {
        const ggml_tensor * d = nodes[i];
        if (done[i] || is_noop(d)) continue;
        if (d->op == GGML_OP_MUL_MAT && d->src[1] == b && is_split(d->src[0]) && supports_mul_mat(d) && d->src[0]->ne[0] == nodes[0]->src[0]->ne[0] &&
            (can_hoist(d, skipped) || hoist_elsewhere(ctx, d))) {          // (its block still in use here: into the scratch, readers redirected)
            members[n++] = d;
            done[i] = 1;
            continue;
        }
        skipped.push_back(d);
    }
    if (dbg()) fprintf(stderr, "split group of %d at %s (N=%lld)\n", n, nodes[0]->name, (long long) b->ne[1]);
    return compute_mul_mat_split(ctx, members, n);
}

int glue_op(const ggml_tensor * node);
qmm_tensor to_qt(const ggml_tensor * t, const mi355x_backend_ctx * ctx);
enum ggml_status compute_glue(mi355x_backend_ctx * ctx, const ggml_tensor * node, int op, const ggml_tensor * s0, const ggml_tensor * s1,
                              const ggml_tensor * s2);

// byte range a tensor occupies (views: the viewed bytes)
bool ranges_overlap(const ggml_tensor * x, const ggml_tensor * y) {
    if (!x->data || !y->data) return false;
    const char * x0 = (const char *) x->data, * y0 = (const char *) y->data;
    return x0 < y0 + ggml_nbytes(y) && y0 < x0 + ggml_nbytes(x);
}
bool dbg();
bool bytes_overlap(const void * a, size_t an, const void * b, size_t bn) {
    const char * a0 = (const char *) a, * b0 = (const char *) b;
    return a && b && a0 < b0 + bn && b0 < a0 + an;
}
// A fused launch writes the buffer of a LATER node (`late`) at the position of an earlier one.  ggml-alloc may have given `late` a
// block that was freed once the earlier nodes' operands were dead, i.e. exactly the memory the launch still reads: other workgroups
// would overwrite it while it is being staged.  Legal only when `late` is disjoint from every operand (an operand at the very same
// address with the same row layout is fine where the launch reads an element before the same thread writes it: `inplace_ok`).
// The attention launches (qmm_attn_decode*, qmm_attn_prefill) write ct = the merged heads [Dv * H, N]; ggml-alloc likes to give ct
// the block of the dead Q.  That is in place and safe when the two coincide head for head: a workgroup owns one (head, token), reads
// its whole q row before anything else and writes the same bytes last; nobody else touches them.  `qv` is a view of Q as [D, H, N]
// (reshaped) or [D, N, H] (permuted); D must equal Dv.
bool attn_q_coincides(const ggml_tensor * ct, const ggml_tensor * qv) {
    if (!qv || qv->data != ct->data || qv->type != GGML_TYPE_F32 || qv->nb[0] != 4 || qv->ne[3] != 1) return false;
    const int64_t D = qv->ne[0];
    int hd = 1, td = 2;                                                             // head / token dimension of the view
    if (qv->nb[1] != (size_t) D * 4) { hd = 2; td = 1; }
    return qv->nb[hd] == (size_t) D * 4 && qv->nb[td] == ct->nb[1] && D * qv->ne[hd] == ct->ne[0] && qv->ne[td] == ct->ne[1];
}
bool early_write_ok(const ggml_tensor * late, std::initializer_list<const ggml_tensor *> operands, const ggml_tensor * inplace_ok = nullptr,
                    bool attn_q = false) {
    for (const ggml_tensor * o : operands) {
        if (!o || !ranges_overlap(late, o)) continue;
        if (o == inplace_ok && o->data == late->data && o->nb[1] == late->nb[1] && ggml_are_same_shape(o, late)) continue;
        if (attn_q && o == inplace_ok && attn_q_coincides(late, o)) continue;
        if (dbg()) fprintf(stderr, "fusion declined: %s would be written early over %s (%p ne %lld,%lld nb1 %zu | %p ne %lld,%lld,%lld nb %zu,%zu,%zu)\n", late->name, o->name,
                           late->data, (long long) late->ne[0], (long long) late->ne[1], late->nb[1], o->data, (long long) o->ne[0], (long long) o->ne[1],
                           (long long) o->ne[2], o->nb[0], o->nb[1], o->nb[2]);
        return false;
    }
    return true;
}
// may `t` run before the nodes in `skipped` although the graph lists it after them?  Its operands are ready (the caller
// checked), so the question is memory: ggml-alloc reuses freed blocks, so t's result must not land on anything a skipped
// node still reads or writes.
bool can_hoist(const ggml_tensor * t, const std::vector<const ggml_tensor *> & skipped) {
    for (const ggml_tensor * s : skipped) {
        if (ranges_overlap(t, s)) return false;
        for (int j = 0; j < GGML_MAX_SRC && s->src[j]; ++j)
            if (ranges_overlap(t, s->src[j])) return false;
    }
    return true;
}
bool is_noop(const ggml_tensor * node) {
    return ggml_is_empty(node) || node->op == GGML_OP_NONE || node->op == GGML_OP_RESHAPE || node->op == GGML_OP_VIEW ||
           node->op == GGML_OP_PERMUTE || node->op == GGML_OP_TRANSPOSE;
}

bool dbg() { static const bool on = getenv("GGML_MI355X_DEBUG") != nullptr; return on; }
// A MUL_MAT that cannot be hoisted in place (ggml-alloc gave it a block that is still live at the earlier point: in
// llama.cpp's layer Kcur reuses the block of the pre-RoPE Qcur) is computed into the context's scratch instead; every reader
// then gets the scratch pointer (to_qt).  Possible when all readers are glue ops of this graph and the result is not a graph
// output.  Returns NULL when it is not.
void * hoist_elsewhere(mi355x_backend_ctx * ctx, const ggml_tensor * d) {
    if (d->flags & GGML_TENSOR_FLAG_OUTPUT) return nullptr;
    auto it = std::lower_bound(ctx->readers.begin(), ctx->readers.end(), d,
                               [](const mi355x_backend_ctx::reader_info & x, const ggml_tensor * y) { return x.t < y; });
    if (it == ctx->readers.end() || it->t != d || !it->glue_only || it->uses == 0) return nullptr;
    const size_t bytes = (ggml_nbytes(d) + 255) & ~(size_t) 255;
    if (ctx->hoist_used + bytes > ctx->hoist_bytes) {
        if (ctx->hoist_used || !ctx->redirects.empty()) {                              // live results in the old block: not this time
            if (dbg()) fprintf(stderr, "hoist refused for %s: scratch exhausted (%zu of %zu bytes in use)\n", d->name, ctx->hoist_used, ctx->hoist_bytes);
            return nullptr;
        }
        if (!grow(ctx->dev, ctx->hoist_buf, ctx->hoist_bytes, std::max<size_t>((size_t) 64 << 20, 64 * bytes))) return nullptr;
    }
    char * p = (char *) ctx->hoist_buf + ctx->hoist_used;
    ctx->hoist_used += bytes;
    // the main loop drops redirects in order of their last reader
    auto pos = ctx->redirects.begin();
    while (pos != ctx->redirects.end() && pos->last_reader <= it->last_reader) ++pos;
    ctx->redirects.insert(pos, { d, p, it->last_reader });
    if (dbg()) fprintf(stderr, "redirect %s (%p) -> scratch %p, readers %d, last reader node %d\n", d->name, d->data, (void *) p, it->uses, it->last_reader);
    return p;
}

constexpr int LOOKAHEAD = 12;      // nodes scanned for MUL_MATs on the same src1 (q .. rope .. k .. rope .. v; gate, silu, up)

// nodes[0] is the MUL_MAT to run; done[] (parallel to nodes) marks later nodes this call has executed as part of its group
enum ggml_status compute_mul_mat(mi355x_backend_ctx * ctx, ggml_tensor * const * nodes, int n_nodes, char * done) {
    const ggml_tensor * dst = nodes[0];
    const ggml_tensor * a = dst->src[0], * b = dst->src[1];
    qmm_ctx * q = ctx->dev->qmm;
    void * st = qmm_stream(q);
    if (is_split(a)) return compute_mul_mat_split_group(ctx, nodes, n_nodes, done);
    const int64_t K = a->ne[0], N = b->ne[1];
    const bool flat = a->ne[2] == 1 && a->ne[3] == 1 && b->ne[2] == 1 && b->ne[3] == 1;
    if (flat) {
        // Group the MUL_MAT nodes that read the same src1 (wq/wk/wv, ffn gate/up): batch <= 8: one launch per weight type (or
        // one mixed-type launch); larger batches: the 16-bit activation operand is prepared once per group.  In llama.cpp's
        // graph order they are not neighbours (q, rope(q), k, rope(k), v; gate, silu, up), so the scan looks past other
        // nodes and hoists a later MUL_MAT when that is safe (can_hoist).
        qmm_weight ws[4];
        int n = 0, member[4] = { 0, 0, 0, 0 };                                 // node index (relative) of every matrix of the group
        ws[n++] = qmm_weight{ a->data, (int64_t) a->nb[1], a->ne[1], (float *) dst->data, (int64_t) (dst->nb[1] / sizeof(float)), weight_type(ctx, a) };
        std::vector<const ggml_tensor *> & skipped = ctx->skipped;
        skipped.clear();
        for (int i = 1; i < n_nodes && i <= LOOKAHEAD && n < 4 && !GGML_MI355X_FUSE_OFF(); ++i) {
            const ggml_tensor * d = nodes[i];
            if (done[i] || is_noop(d)) continue;
            const ggml_tensor * w = d->src[0];
            if (d->op == GGML_OP_MUL_MAT && d->src[1] == b && !glue_op(d) && supports_mul_mat(d) && is_ours(w) && !is_split(w) && w->ne[2] == 1 &&
                w->ne[3] == 1 && w->ne[0] == K) {
                float * out = (float *) d->data;
                if (!can_hoist(d, skipped)) out = (float *) hoist_elsewhere(ctx, d);   // its block is still in use here: compute into scratch
                if (out) {
                    member[n] = i;
                    ws[n++] = qmm_weight{ w->data, (int64_t) w->nb[1], w->ne[1], out, (int64_t) (d->nb[1] / sizeof(float)), weight_type(ctx, w) };
                    done[i] = 1;
                    continue;
                }
            }
            skipped.push_back(d);
        }
        if (dbg()) fprintf(stderr, "group of %d at %s (N=%lld) src1 %s %p -> %p\n", n, dst->name, (long long) N, b->name, b->data, to_qt(b, ctx).data);
        const float * x = (const float *) to_qt(b, ctx).data;                  // (a merged-heads CONT may live in the scratch: attention sites)
        int64_t ldx = b->nb[1] / sizeof(float);
        qmm_mv_extra ex{};
        bool use_ex = false;
        // (a) src1 is a held-back RMS_NORM * w (graph_compute): when the group holds every reader of it the kernels form the normed
        //     row themselves; otherwise it is materialized now, as the graph says
        if (ctx->pending_norm.mul == b) {
            const auto pn = ctx->pending_norm;
            ctx->pending_norm = {};
            // the kernels read the un-normed row while they write their results: a result that ggml-alloc placed in the block of
            // that row (dead once the norm has run, in the graph's order) would be overwritten under the staging of other workgroups
            bool norm_in_kernel = n == pn.readers;
            for (int i = 0; norm_in_kernel && i < n; ++i)
                norm_in_kernel = !bytes_overlap(ws[i].dst, (size_t) ((N - 1) * ws[i].ldd + ws[i].M) * sizeof(float), to_qt(pn.rn->src[0], ctx).data, ggml_nbytes(pn.rn->src[0]));
            if (pn.add) {
                // a prompt batch (site_add_rms_norm held ADD -> RMS_NORM -> MUL back): the group's activation prep adds, norms and stores the
                // sum.  The products are written by later launches of the same stream, so they may sit on the dead operands; not on the sum
                const ggml_tensor * x0 = pn.add->src[0], * x1 = pn.add->src[1];
                norm_in_kernel = n == pn.readers && qmm_mul_mat_group_norm_supported(q, ws, n, K, N);
                for (int i = 0; norm_in_kernel && i < n; ++i)
                    norm_in_kernel = !bytes_overlap(ws[i].dst, (size_t) ((N - 1) * ws[i].ldd + ws[i].M) * sizeof(float), to_qt(pn.add, ctx).data, ggml_nbytes(pn.add));
                if (norm_in_kernel) {
                    x = (const float *) to_qt(x0, ctx).data;
                    ldx = x0->nb[1] / sizeof(float);
                    ex.norm_w = (const float *) pn.w->data;
                    memcpy(&ex.norm_eps, pn.rn->op_params, sizeof(float));
                    ex.norm_add = (const float *) to_qt(x1, ctx).data;
                    ex.norm_add_ld = x1->nb[1] / sizeof(float);
                    ex.norm_sum = (float *) to_qt(pn.add, ctx).data;
                    ex.norm_sum_ld = pn.add->nb[1] / sizeof(float);
                    use_ex = true;
                    if (dbg()) fprintf(stderr, "fused: add + norm (%s) into the prep of %s\n", pn.add->name, dst->name);
                } else {                                                              // the graph's three nodes after all, as the site would have run them
                    const qmm_tensor a = to_qt(x0, ctx), bb = to_qt(x1, ctx), w = to_qt(pn.w, ctx), sum = to_qt(pn.add, ctx), d = to_qt(pn.mul, ctx);
                    float eps;
                    memcpy(&eps, pn.rn->op_params, sizeof(float));
                    if (dbg()) fprintf(stderr, "fusion declined: add + norm (%s) into the prep of %s (group of %d of %d readers)\n", pn.add->name, dst->name, n, pn.readers);
                    if (qmm_op_add_rms_norm(q, &a, &bb, &w, &sum, &d, eps, st)) {
                        GGML_LOG_ERROR("MI355X ADD+RMS_NORM(%s): %s\n", pn.add->name, qmm_last_error());
                        return GGML_STATUS_FAILED;
                    }
                }
            } else if (norm_in_kernel) {
                const qmm_tensor qx = to_qt(pn.rn->src[0], ctx);
                x = (const float *) qx.data;
                ldx = pn.rn->src[0]->nb[1] / sizeof(float);
                ex.norm_w = (const float *) pn.w->data;
                memcpy(&ex.norm_eps, pn.rn->op_params, sizeof(float));
                use_ex = true;
            } else {
                ggml_tensor tmp = *pn.mul;
                memcpy(tmp.op_params, pn.rn->op_params, sizeof(tmp.op_params));
                const enum ggml_status s = compute_glue(ctx, &tmp, QMM_OP_RMS_NORM_MUL, pn.rn->src[0], pn.w, nullptr);
                if (s != GGML_STATUS_SUCCESS) return s;
            }
        }
        // (c) ffn_gate + ffn_up of a few-token batch whose only readers are silu(gate) and the MUL of the two (build_ffn's SwiGLU):
        //     the kernel pairs the rows and writes the product; neither projection is stored
        if (n == 2 && N <= QMM_MATVEC_MAX_N && ws[0].type == ws[1].type && ws[0].M == ws[1].M && !GGML_MI355X_FUSE_OFF()) {
            auto uses = [&](const ggml_tensor * t) {
                auto it = std::lower_bound(ctx->readers.begin(), ctx->readers.end(), t,
                                           [](const mi355x_backend_ctx::reader_info & r, const ggml_tensor * y) { return r.t < y; });
                return it != ctx->readers.end() && it->t == t ? it->uses : -1;
            };
            const ggml_tensor * t0 = nodes[0], * t1 = nodes[member[1]];
            int js = -1, jm = -1;
            for (int j = 1; j < n_nodes && j <= 2 * LOOKAHEAD && jm < 0; ++j) {
                const ggml_tensor * t = nodes[j];
                if (done[j] || is_noop(t)) continue;
                if (js < 0) {
                    if (t->op == GGML_OP_UNARY && ggml_get_unary_op(t) == GGML_UNARY_OP_SILU && (t->src[0] == t0 || t->src[0] == t1)) js = j;
                    else break;
                } else {
                    const ggml_tensor * other = nodes[js]->src[0] == t0 ? t1 : t0;
                    if (t->op == GGML_OP_MUL && ((t->src[0] == nodes[js] && t->src[1] == other) || (t->src[1] == nodes[js] && t->src[0] == other))) jm = j;
                    else break;
                }
            }
            // the product lands in the MUL's buffer while the launch still stages x: that buffer must be disjoint from x and from
            // everything the nodes in between (run later) still read or write
            bool early_ok = jm >= 0 && !bytes_overlap(nodes[jm]->data, ggml_nbytes(nodes[jm]), x, (size_t) ((N - 1) * ldx + K) * sizeof(float));
            for (int j = 1; early_ok && j < jm; ++j) {
                if (done[j] || is_noop(nodes[j]) || j == js) continue;
                early_ok = !ranges_overlap(nodes[jm], nodes[j]);
                for (int k = 0; early_ok && k < GGML_MAX_SRC && nodes[j]->src[k]; ++k) early_ok = !ranges_overlap(nodes[jm], nodes[j]->src[k]);
            }
            if (jm >= 0 && !early_ok && dbg()) fprintf(stderr, "fusion declined: SwiGLU into %s\n", nodes[jm]->name);
            if (jm >= 0 && early_ok && uses(t0) == 1 && uses(t1) == 1 && uses(nodes[js]) == 1 && ggml_are_same_shape(nodes[jm], t0) && nodes[jm]->nb[0] == 4 &&
                !(nodes[jm]->flags & GGML_TENSOR_FLAG_OUTPUT) && !(t0->flags & GGML_TENSOR_FLAG_OUTPUT) && !(t1->flags & GGML_TENSOR_FLAG_OUTPUT) &&
                (size_t) N * K * 5 / 4 + (ex.norm_w ? (size_t) N * K * 4 : 0) + 4096 <= 150 * 1024) {
                ex.swiglu = nodes[js]->src[0] == t0 ? 1 : 2;
                ws[0].dst = (float *) nodes[jm]->data;
                ws[0].ldd = (int64_t) (nodes[jm]->nb[1] / sizeof(float));
                done[js] = done[jm] = 1;
                use_ex = true;
            }
        }
        // (b) a lone MUL_MAT whose only reader is the residual ADD right behind it (wo, ffn_down): dst = W x + residual, written
        //     where the ADD would have put it
        if (n == 1 && N <= QMM_MATVEC_MAX_N && !GGML_MI355X_FUSE_OFF() && !(dst->flags & GGML_TENSOR_FLAG_OUTPUT)) {
            int j = 1;
            while (j < n_nodes && (done[j] || is_noop(nodes[j]))) ++j;
            const ggml_tensor * add = j < n_nodes ? nodes[j] : nullptr;
            auto it = std::lower_bound(ctx->readers.begin(), ctx->readers.end(), dst,
                                       [](const mi355x_backend_ctx::reader_info & r, const ggml_tensor * t) { return r.t < t; });
            const bool single = it != ctx->readers.end() && it->t == dst && it->uses == 1;
            if (add && single && add->op == GGML_OP_ADD && (add->src[0] == dst || add->src[1] == dst) && add->src[0] != add->src[1]) {
                const ggml_tensor * r = add->src[0] == dst ? add->src[1] : add->src[0];
                // dst = W x + r is written where the ADD would put it, while x is still being staged by other workgroups: the ADD's buffer must
                // not be x's (ggml-alloc may hand the ADD the block of the dead src1), and r only where it is the very same rows (in place)
                const qmm_tensor qr = to_qt(r, ctx);
                const bool r_ok = !bytes_overlap(add->data, ggml_nbytes(add), qr.data, ggml_nbytes(r)) || (qr.data == add->data && r->nb[1] == add->nb[1]);
                const bool x_ok = !bytes_overlap(add->data, ggml_nbytes(add), x, (size_t) ((N - 1) * ldx + K) * sizeof(float));
                if ((!r_ok || !x_ok) && dbg()) fprintf(stderr, "fusion declined: residual into %s\n", add->name);
                if (r_ok && x_ok && ggml_are_same_shape(r, dst) && ggml_are_same_shape(add, dst) && r->type == GGML_TYPE_F32 && r->nb[0] == 4 && add->nb[0] == 4 &&
                    r->nb[1] == add->nb[1] && add->nb[1] % 4 == 0) {
                    ex.residual[0] = (const float *) to_qt(r, ctx).data;
                    ws[0].dst = (float *) add->data;
                    ws[0].ldd = (int64_t) (add->nb[1] / sizeof(float));
                    done[j] = 1;
                    use_ex = true;
                }
            }
        }
        if (use_ex ? qmm_mul_mat_group_ex(q, ws, n, K, x, N, ldx, &ex, st) : qmm_mul_mat_group(q, ws, n, K, x, N, ldx, st)) {
            GGML_LOG_ERROR("MI355X MUL_MAT(%s): %s\n", dst->name, qmm_last_error());
            return GGML_STATUS_FAILED;
        }
        return GGML_STATUS_SUCCESS;
    }
    // batched / broadcast form: one 2-D product per (i12, i13), src0 broadcast as ggml_compute_forward_mul_mat does
    // (ggml-cpu.c:6711-6716: i03 = i13 / r3, i02 = i12 / r2)
    const int64_t r2 = b->ne[2] / a->ne[2], r3 = b->ne[3] / a->ne[3];
    for (int64_t i13 = 0; i13 < b->ne[3]; ++i13)
        for (int64_t i12 = 0; i12 < b->ne[2]; ++i12) {
            const char * wp = (const char *) a->data + (i12 / r2) * a->nb[2] + (i13 / r3) * a->nb[3];
            const char * xp = (const char *) b->data + i12 * b->nb[2] + i13 * b->nb[3];
            char * dp = (char *) dst->data + i12 * dst->nb[2] + i13 * dst->nb[3];
            if (qmm_mul_mat(q, weight_type(ctx, a), wp, a->nb[1], K, a->ne[1], (const float *) xp, N, b->nb[1] / sizeof(float),
                            (float *) dp, dst->nb[1] / sizeof(float), st)) {
                GGML_LOG_ERROR("MI355X MUL_MAT(%s): %s\n", dst->name, qmm_last_error());
                return GGML_STATUS_FAILED;
            }
        }
    return GGML_STATUS_SUCCESS;
}

// ffn_gate_exps and ffn_up_exps are consecutive MUL_MAT_ID nodes on the same src1 and ids (llama.cpp build_moe_ffn): when the
// next node is such a twin, both go out in one call (one mat-vec launch, or one expert sort + activation prep)
bool moe_twin(const ggml_tensor * d0, const ggml_tensor * d1) {
    if (d1->op != GGML_OP_MUL_MAT_ID || !supports_mul_mat_id(d1)) return false;
    const ggml_tensor * a0 = d0->src[0], * a1 = d1->src[0];
    if (d0->src[1] != d1->src[1] || d0->src[2] != d1->src[2] || a0 == a1 || !is_ours(a1) || !is_ours(d1)) return false;
    if (a0->type != a1->type || !ggml_are_same_shape(a0, a1) || a0->nb[1] != a1->nb[1] || a0->nb[2] != a1->nb[2]) return false;
    return d0->nb[1] == d1->nb[1] && d0->nb[2] == d1->nb[2] && ggml_are_same_shape(d0, d1);
}

enum ggml_status compute_mul_mat_id(mi355x_backend_ctx * ctx, ggml_tensor * const * nodes, int n_nodes, char * done) {
    const ggml_tensor * dst = nodes[0];
    const ggml_tensor * as = dst->src[0], * b = dst->src[1], * ids = dst->src[2];
    qmm_ctx * q = ctx->dev->qmm;
    // the twin (ffn_up_exps after ffn_gate_exps) may sit behind the SILU of the first one
    int twin = 0;
    std::vector<const ggml_tensor *> & skipped = ctx->skipped;
    skipped.clear();
    for (int i = 1; i < n_nodes && i <= LOOKAHEAD && !twin && !GGML_MI355X_FUSE_OFF(); ++i) {
        if (done[i] || is_noop(nodes[i])) continue;
        if (moe_twin(dst, nodes[i]) && can_hoist(nodes[i], skipped)) twin = i;
        else skipped.push_back(nodes[i]);
        if (nodes[i]->op == GGML_OP_MUL_MAT_ID && !twin) break;
    }
    if (twin) {
        const ggml_tensor * dst1 = nodes[twin];
        const int t0 = weight_type(ctx, as), t1 = weight_type(ctx, dst1->src[0]);
        if (t0 != t1) { GGML_LOG_ERROR("MI355X MUL_MAT_ID pair: layouts differ\n"); return GGML_STATUS_FAILED; }
        // Token generation: the SwiGLU behind the pair (ggml_silu of one, ggml_mul with the other: ffn_moe_gate_par) in the same launch
        // (round 3).  gate, up and the silu are then never written: each must have exactly the one reader of the pattern, all of them
        // provably in this split (analyze_readers); the product is written HERE, two or three nodes early, while other workgroups still
        // stage src1 and read the ids, and across whatever else the graph lists in between.
        if (qmm_mul_mat_id_swiglu_supported(ids->ne[0], ids->ne[1]) && !getenv("GGML_MI355X_MOE_SWIGLU_OFF")) {
            auto uses = [&](const ggml_tensor * t) {
                const auto & rd = ctx->readers;
                auto it = std::lower_bound(rd.begin(), rd.end(), t, [](const mi355x_backend_ctx::reader_info & x, const ggml_tensor * y) { return x.t < y; });
                return it != rd.end() && it->t == t ? it->uses : 1 << 20;
            };
            int i_silu = 0, i_mul = 0;
            std::vector<const ggml_tensor *> between;
            for (int i = 1; i < n_nodes && i <= 2 * LOOKAHEAD && !i_mul; ++i) {
                const ggml_tensor * t = nodes[i];
                if (i == twin || done[i] || is_noop(t)) continue;
                if (!i_silu && t->op == GGML_OP_UNARY && ggml_get_unary_op(t) == GGML_UNARY_OP_SILU && (t->src[0] == dst || t->src[0] == dst1)) { i_silu = i; continue; }
                if (i_silu && t->op == GGML_OP_MUL) {
                    const ggml_tensor * sl = nodes[i_silu], * other = sl->src[0] == dst ? dst1 : dst;
                    if ((t->src[0] == sl && t->src[1] == other) || (t->src[1] == sl && t->src[0] == other)) { i_mul = i; continue; }
                }
                between.push_back(t);
            }
            if (i_silu && i_mul) {
                const ggml_tensor * sl = nodes[i_silu], * par = nodes[i_mul];
                const ggml_tensor * gate = sl->src[0], * up = gate == dst ? dst1 : dst;
                const bool flags_ok = !((gate->flags | up->flags | sl->flags) & GGML_TENSOR_FLAG_OUTPUT);
                if (flags_ok && uses(gate) == 1 && uses(up) == 1 && uses(sl) == 1 && is_ours(par) && par->type == GGML_TYPE_F32 && ggml_are_same_shape(par, dst) &&
                    par->nb[0] == 4 && par->nb[1] == dst->nb[1] && par->nb[2] == dst->nb[2] && can_hoist(par, between) && early_write_ok(par, { b, ids })) {
                    if (qmm_mul_mat_id_swiglu(q, t0, gate->src[0]->data, up->src[0]->data, as->nb[1], as->nb[2], as->ne[0], as->ne[1], as->ne[2],
                                              (const float *) b->data, b->ne[1], b->nb[1], b->nb[2],
                                              (const int32_t *) ids->data, ids->ne[0], ids->ne[1], ids->nb[1],
                                              (float *) par->data, par->nb[1], par->nb[2], qmm_stream(q))) {
                        GGML_LOG_ERROR("MI355X MUL_MAT_ID + SwiGLU(%s, %s): %s\n", dst->name, dst1->name, qmm_last_error());
                        return GGML_STATUS_FAILED;
                    }
                    if (dbg()) fprintf(stderr, "fused: expert pair + swiglu (%s)\n", par->name);
                    done[twin] = 1; done[i_silu] = 1; done[i_mul] = 1;
                    return GGML_STATUS_SUCCESS;
                }
            }
        }
        if (qmm_mul_mat_id_pair(q, t0, as->data, dst1->src[0]->data, as->nb[1], as->nb[2], as->ne[0], as->ne[1], as->ne[2],
                                (const float *) b->data, b->ne[1], b->nb[1], b->nb[2],
                                (const int32_t *) ids->data, ids->ne[0], ids->ne[1], ids->nb[1],
                                (float *) dst->data, (float *) dst1->data, dst->nb[1], dst->nb[2], qmm_stream(q))) {
            GGML_LOG_ERROR("MI355X MUL_MAT_ID(%s, %s): %s\n", dst->name, dst1->name, qmm_last_error());
            return GGML_STATUS_FAILED;
        }
        done[twin] = 1;
        return GGML_STATUS_SUCCESS;
    }
    if (qmm_mul_mat_id(q, weight_type(ctx, as), as->data, as->nb[1], as->nb[2], as->ne[0], as->ne[1], as->ne[2],
                       (const float *) b->data, b->ne[1], b->nb[1], b->nb[2],
                       (const int32_t *) ids->data, ids->ne[0], ids->ne[1], ids->nb[1],
                       (float *) dst->data, dst->nb[1], dst->nb[2], qmm_stream(q))) {
        GGML_LOG_ERROR("MI355X MUL_MAT_ID(%s): %s\n", dst->name, qmm_last_error());
        return GGML_STATUS_FAILED;
    }
    return GGML_STATUS_SUCCESS;
}


// ---- glue ops: everything between the quantized MUL_MATs of a layer, so that a layer is one split.
// The kernel library decides what it implements (qmm_op_supported); this side only translates ggml nodes.

qmm_tensor to_qt(const ggml_tensor * t, const mi355x_backend_ctx * ctx) {
    qmm_tensor q{};
    q.data = t->data;
    if (ctx && !ctx->redirects.empty()) {
        const ggml_tensor * root = t->view_src ? t->view_src : t;
        for (const auto & r : ctx->redirects)
            if (r.t == root) q.data = r.data + ((const char *) t->data - (const char *) root->data);
    }
    q.type = (int32_t) dev_type(t);
    for (int i = 0; i < 4; ++i) { q.ne[i] = t->ne[i]; q.nb[i] = (int64_t) t->nb[i]; }
    memcpy(q.op_params, t->op_params, sizeof(q.op_params));
    return q;
}

// ggml node -> library op; 0 when the node is not a glue op of this backend
int glue_op(const ggml_tensor * node) {
    switch (node->op) {
        case GGML_OP_ADD:      return QMM_OP_ADD;
        case GGML_OP_SUB:      return QMM_OP_SUB;
        case GGML_OP_MUL:      return QMM_OP_MUL;
        case GGML_OP_DIV:      return QMM_OP_DIV;
        case GGML_OP_SCALE:    return QMM_OP_SCALE;
        case GGML_OP_RMS_NORM: return QMM_OP_RMS_NORM;
        case GGML_OP_NORM:     return QMM_OP_NORM;
        case GGML_OP_ROPE:     return QMM_OP_ROPE;
        case GGML_OP_SOFT_MAX: return QMM_OP_SOFT_MAX;
        case GGML_OP_CPY: case GGML_OP_CONT: case GGML_OP_DUP: return QMM_OP_CPY;
        case GGML_OP_GET_ROWS: return QMM_OP_GET_ROWS;
        case GGML_OP_ARGSORT:  return QMM_OP_ARGSORT;
        case GGML_OP_SUM_ROWS: return QMM_OP_SUM_ROWS;
        case GGML_OP_MUL_MAT:  return node->src[0] && (node->src[0]->type == GGML_TYPE_F16 || node->src[0]->type == GGML_TYPE_F32) ? QMM_OP_MUL_MAT_F : 0;
        case GGML_OP_UNARY:
            switch (ggml_get_unary_op(node)) {
                case GGML_UNARY_OP_SILU:       return QMM_OP_SILU;
                case GGML_UNARY_OP_GELU:       return QMM_OP_GELU;
                case GGML_UNARY_OP_GELU_QUICK: return QMM_OP_GELU_QUICK;
                case GGML_UNARY_OP_RELU:       return QMM_OP_RELU;
                case GGML_UNARY_OP_TANH:       return QMM_OP_TANH;
                case GGML_UNARY_OP_SIGMOID:    return QMM_OP_SIGMOID;
                case GGML_UNARY_OP_NEG:        return QMM_OP_NEG;
                case GGML_UNARY_OP_EXP:        return QMM_OP_EXP;
                default:                       return 0;
            }
        default: return 0;
    }
}

// the operands as the library wants them: CPY's dst is src[1]'s layout (the node itself is a view of it)
bool supports_glue(const ggml_tensor * node) {
    const int op = glue_op(node);
    if (!op) return false;
    if (GGML_MI355X_GLUE_OFF()) return false;
    qmm_tensor s[3];
    const qmm_tensor * ps[3] = { nullptr, nullptr, nullptr };
    for (int i = 0; i < 3; ++i)
        if (node->src[i]) { s[i] = to_qt(node->src[i], nullptr); ps[i] = &s[i]; }
    const qmm_tensor d = to_qt(node, nullptr);
    if (op == QMM_OP_CPY) return qmm_op_supported(op, ps[0], nullptr, nullptr, &d) != 0;
    return qmm_op_supported(op, ps[0], ps[1], ps[2], &d) != 0;
}

enum ggml_status compute_glue(mi355x_backend_ctx * ctx, const ggml_tensor * node, int op, const ggml_tensor * s0, const ggml_tensor * s1,
                              const ggml_tensor * s2) {
    qmm_tensor s[3];
    const qmm_tensor * ps[3] = { nullptr, nullptr, nullptr };
    const ggml_tensor * srcs[3] = { s0, s1, s2 };
    for (int i = 0; i < 3; ++i)
        if (srcs[i]) { s[i] = to_qt(srcs[i], ctx); ps[i] = &s[i]; }
    const qmm_tensor d = to_qt(node, ctx);
    if (qmm_op_compute(ctx->dev->qmm, op, ps[0], ps[1], ps[2], &d, qmm_stream(ctx->dev->qmm))) {
        GGML_LOG_ERROR("MI355X %s(%s): %s\n", ggml_op_name(node->op), node->name, qmm_last_error());
        return GGML_STATUS_FAILED;
    }
    return GGML_STATUS_SUCCESS;
}

// Pairs the library runs as one launch: RMS_NORM -> MUL by a one-row weight (build_norm), SILU -> MUL (build_ffn's SwiGLU).
// Legal only when the first node's result has no other reader: `single_use` is computed per graph in graph_compute.
int fused_pair(const ggml_tensor * n0, const ggml_tensor * n1, const ggml_tensor ** other) {
    if (n1->op != GGML_OP_MUL || (n1->src[0] != n0 && n1->src[1] != n0) || n1->src[0] == n1->src[1]) return 0;
    *other = n1->src[0] == n0 ? n1->src[1] : n1->src[0];
    if (!ggml_are_same_shape(n0, n1) || (n0->flags & GGML_TENSOR_FLAG_OUTPUT)) return 0;
    qmm_tensor a = to_qt(n0->src[0], nullptr), b = to_qt(*other, nullptr), d = to_qt(n1, nullptr);
    memcpy(d.op_params, n0->op_params, sizeof(d.op_params));
    if (n0->op == GGML_OP_RMS_NORM && qmm_op_supported(QMM_OP_RMS_NORM_MUL, &a, &b, nullptr, &d)) return QMM_OP_RMS_NORM_MUL;
    if (n0->op == GGML_OP_UNARY && ggml_get_unary_op(n0) == GGML_UNARY_OP_SILU && qmm_op_supported(QMM_OP_SILU_MUL, &a, &b, nullptr, &d))
        return QMM_OP_SILU_MUL;
    return 0;
}

// ----------------------------------------------------------------------------------------------- backend (stream)

const char * backend_get_name(ggml_backend_t backend) { return ((mi355x_backend_ctx *) backend->context)->name.c_str(); }

void backend_free(ggml_backend_t backend) {
    auto * ctx = (mi355x_backend_ctx *) backend->context;
    if (ctx->ev_copy) qmm_event_destroy(ctx->dev->qmm, ctx->ev_copy);
    if (ctx->ev_t0) {
        fprintf(stderr, "MI355X timing %s: tg graphs %lld stream_ms %.3f | pp graphs %lld tokens %lld stream_ms %.3f min_ms %.3f\n", ctx->name.c_str(),
                (long long) ctx->graphs_tg, ctx->ms_tg, (long long) ctx->graphs_pp, (long long) ctx->tokens_pp, ctx->ms_pp, ctx->ms_pp_min);
        for (int i = 0; i < host_timer::N; ++i)
            if (host_timer::tg_calls[i]) fprintf(stderr, "MI355X timing %s: %s in front of a one-token graph: %.2f calls, %.1f us each, %.1f us per graph\n", ctx->name.c_str(), host_timer::name(i),
                                                 host_timer::tg_calls[i] / (double) std::max<int64_t>(ctx->graphs_tg, 1), host_timer::tg_us[i] / (double) host_timer::tg_calls[i],
                                                 host_timer::tg_us[i] / (double) std::max<int64_t>(ctx->graphs_tg, 1));
        if (ctx->graphs_tg > 1)
            fprintf(stderr, "MI355X timing %s: host us per tg graph: outside graph_compute %.1f | reader analysis %.1f | issue loop %.1f | waiting in synchronize %.1f\n", ctx->name.c_str(),
                    ctx->us_outside / (double) (ctx->graphs_tg - 1), ctx->us_analyze / (double) ctx->graphs_tg, ctx->us_issue / (double) ctx->graphs_tg, ctx->us_wait / (double) ctx->graphs_tg);
        for (int i = 0; i < host_timer::N && ctx->graphs_pp > 0; ++i)
            if (host_timer::pp_calls[i]) fprintf(stderr, "MI355X timing %s: %s in front of a prompt graph: %.2f calls, %.1f us each, %.1f us per graph\n", ctx->name.c_str(), host_timer::name(i),
                                                 host_timer::pp_calls[i] / (double) std::max<int64_t>(ctx->graphs_pp - 2, 1), host_timer::pp_us[i] / (double) host_timer::pp_calls[i],
                                                 host_timer::pp_us[i] / (double) std::max<int64_t>(ctx->graphs_pp - 2, 1));
        if (ctx->graphs_pp > 1)
            fprintf(stderr, "MI355X timing %s: host us per pp graph: outside graph_compute %.1f (between consecutive prompt graphs) | reader analysis %.1f | issue loop %.1f | waiting in synchronize %.1f\n",
                    ctx->name.c_str(), ctx->pp_outside / (double) std::max<int64_t>(ctx->pp_outside_n, 1), ctx->pp_analyze / (double) (ctx->graphs_pp - 1), ctx->pp_issue / (double) (ctx->graphs_pp - 1),
                    ctx->pp_wait / (double) (ctx->graphs_pp - 1));
        qmm_event_destroy(ctx->dev->qmm, ctx->ev_t0);
        qmm_event_destroy(ctx->dev->qmm, ctx->ev_t1);
    }
    if (ctx->hoist_buf) {
        qmm_synchronize(ctx->dev->qmm, qmm_stream(ctx->dev->qmm));
        qmm_free(ctx->dev->qmm, ctx->hoist_buf);
    }
    delete ctx;
    delete backend;
}

// ---- asynchronous transfers and events: everything a backend does is ordered on its context's stream, so
// these are the stream-ordered forms of set/get/cpy (ggml-backend-impl.h:93-95) and record/wait (:114-116).
// graph_compute itself still ends with a stream synchronize (that is where a bad expert id is reported).
bool on_device(const struct ggml_tensor * t, const mi355x_device_ctx * dev) {
    ggml_backend_buffer_t b = t->view_src ? t->view_src->buffer : t->buffer;
    return b && b->buft->iface.get_name == buft_get_name && b->buft->context == (void *) dev;
}
void backend_set_tensor_async(ggml_backend_t backend, struct ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    host_timer timer_(3);
    auto * ctx = (mi355x_backend_ctx *) backend->context;
    GGML_ASSERT(on_device(tensor, ctx->dev) && "set_tensor_async: tensor is not in this device's buffer type");
    ctx->dev->enq++;
    if (mi355x_buffer_ctx * bc = our_buffer_ctx(tensor)) planar_release(bc, (const char *) tensor->data + offset, size, true, qmm_stream(ctx->dev->qmm));
    if (qmm_memcpy_h2d_async(ctx->dev->qmm, (char *) tensor->data + offset, data, size, qmm_stream(ctx->dev->qmm)))
        GGML_ABORT("MI355X set_tensor_async: %s", qmm_last_error());
}
void backend_get_tensor_async(ggml_backend_t backend, const struct ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    host_timer timer_(4);
    auto * ctx = (mi355x_backend_ctx *) backend->context;
    GGML_ASSERT(on_device(tensor, ctx->dev) && "get_tensor_async: tensor is not in this device's buffer type");
    ctx->dev->enq++;
    if (mi355x_buffer_ctx * bc = our_buffer_ctx(tensor)) planar_release(bc, (const char *) tensor->data + offset, size, false, qmm_stream(ctx->dev->qmm));
    if (qmm_memcpy_d2h_async(ctx->dev->qmm, data, (const char *) tensor->data + offset, size, qmm_stream(ctx->dev->qmm)))
        GGML_ABORT("MI355X get_tensor_async: %s", qmm_last_error());
}
ggml_guid_t backend_guid();
bool backend_cpy_tensor_async(ggml_backend_t backend_src, ggml_backend_t backend_dst, const struct ggml_tensor * src, struct ggml_tensor * dst) {
    host_timer timer_(5);
    if (!ggml_guid_matches(backend_src->guid, backend_guid()) || !ggml_guid_matches(backend_dst->guid, backend_guid())) return false;
    auto * sctx = (mi355x_backend_ctx *) backend_src->context;
    auto * dctx = (mi355x_backend_ctx *) backend_dst->context;
    if (!on_device(src, sctx->dev) || !on_device(dst, dctx->dev) || !ggml_is_contiguous(src) || !ggml_is_contiguous(dst)) return false;
    if (mi355x_buffer_ctx * bc = our_buffer_ctx(src)) planar_release(bc, (const char *) src->data, ggml_nbytes(src), false, qmm_stream(sctx->dev->qmm));
    if (mi355x_buffer_ctx * bc = our_buffer_ctx(dst)) planar_release(bc, (const char *) dst->data, ggml_nbytes(src), true, qmm_stream(dctx->dev->qmm));
    qmm_ctx * dq = dctx->dev->qmm;
    void * dst_stream = qmm_stream(dq);
    dctx->dev->enq++;
    if (sctx->dev != dctx->dev) {            // the copy runs on the destination's stream, behind what the source has queued
        if (!sctx->ev_copy) sctx->ev_copy = qmm_event_create(sctx->dev->qmm);
        if (!sctx->ev_copy || qmm_event_record(sctx->dev->qmm, sctx->ev_copy, qmm_stream(sctx->dev->qmm)) ||
            qmm_stream_wait_event(dq, dst_stream, sctx->ev_copy))
            GGML_ABORT("MI355X cpy_tensor_async: %s", qmm_last_error());
    }
    if (qmm_memcpy_d2d(dq, dst->data, src->data, ggml_nbytes(src), dst_stream)) GGML_ABORT("MI355X cpy_tensor_async: %s", qmm_last_error());
    return true;
}
void backend_event_record(ggml_backend_t backend, ggml_backend_event_t event) {
    auto * ctx = (mi355x_backend_ctx *) backend->context;
    if (qmm_event_record(ctx->dev->qmm, (qmm_event *) event->context, qmm_stream(ctx->dev->qmm)))
        GGML_ABORT("MI355X event_record: %s", qmm_last_error());
}
void backend_event_wait(ggml_backend_t backend, ggml_backend_event_t event) {
    auto * ctx = (mi355x_backend_ctx *) backend->context;
    ctx->dev->enq++;
    if (qmm_stream_wait_event(ctx->dev->qmm, qmm_stream(ctx->dev->qmm), (qmm_event *) event->context))
        GGML_ABORT("MI355X event_wait: %s", qmm_last_error());
}

void backend_synchronize(ggml_backend_t backend) {
    host_timer timer_(6);
    auto * ctx = (mi355x_backend_ctx *) backend->context;
    const uint64_t queued = ctx->dev->enq.load();
    if (queued == ctx->dev->enq_synced.load()) return;       // nothing queued by this module since the last wait on this stream
    if (qmm_synchronize(ctx->dev->qmm, qmm_stream(ctx->dev->qmm)))
        GGML_LOG_ERROR("MI355X synchronize: %s\n", qmm_last_error());
    { std::lock_guard<std::mutex> lock(ctx->dev->ring_mu); ctx->dev->staged_pending = false; }
    ctx->dev->enq_synced.store(queued);
}

// the analysis step of a graph_compute call
void analyze_readers(mi355x_backend_ctx * ctx, const ggml_cgraph * cgraph) {
    // Reader analysis.  Candidates are the nodes a fusion wants to skip or move: RMS_NORM / SILU (fused into the MUL behind them),
    // kq / soft_max / kqv (the attention launch) and the quantized MUL_MATs (grouped with an earlier one on the same src1).
    // One pass over all operands records, per candidate, how many nodes of this graph read its memory (directly or through
    // views), the last of them, and whether all of them are glue ops (whose operand pointers this file can redirect).
    std::vector<mi355x_backend_ctx::reader_info> & rd = ctx->readers;
    ctx->redirects.clear();
    ctx->hoist_used = 0;
    ctx->pending_norm = {};
    // Token generation hands over the same graph again and again (same tensors at the same addresses: llama.cpp rebuilds it in the same
    // context memory, ggml-alloc places it the same way; only offsets INTO the KV cache move, and the cache is nobody's candidate): the
    // analysis below depends on nothing but what this signature covers, so an unchanged signature keeps the previous result
    // (58 us per token otherwise, in front of the first launch).
    uint64_t sig = 1469598103934665603ull ^ (uint64_t) cgraph->n_nodes ^ (GGML_MI355X_FUSE_OFF() ? 0x9e3779b97f4a7c15ull : 0);
    auto mix = [&sig](uint64_t v) { sig = (sig ^ v) * 1099511628211ull; };
    for (int i = 0; i < cgraph->n_nodes; ++i) {
        const ggml_tensor * n = cgraph->nodes[i];
        mix((uint64_t) (uintptr_t) n);  mix(n->view_src ? 0 : (uint64_t) (uintptr_t) n->data);      // (views into the KV cache move with every token; they own no memory)
        mix((uint64_t) n->op ^ ((uint64_t) n->flags << 32));  mix((uint64_t) (uintptr_t) n->view_src);
        mix((uint64_t) n->ne[0] ^ ((uint64_t) n->ne[1] << 20) ^ ((uint64_t) n->ne[2] << 40));  mix((uint64_t) n->nb[1] ^ ((uint64_t) n->nb[2] << 24));
        for (int j = 0; j < GGML_MAX_SRC && n->src[j]; ++j) mix((uint64_t) (uintptr_t) n->src[j] + (uint64_t) j);
    }
    if (sig == ctx->readers_sig && ctx->readers_sig_nodes == cgraph->n_nodes) return;
    ctx->readers_sig = sig;
    ctx->readers_sig_nodes = cgraph->n_nodes;
    rd.clear();
    if (!GGML_MI355X_FUSE_OFF()) {
        for (int i = 0; i + 1 < cgraph->n_nodes; ++i) {
            const ggml_tensor * n0 = cgraph->nodes[i];
            if (n0->op == GGML_OP_RMS_NORM || (n0->op == GGML_OP_UNARY && ggml_get_unary_op(n0) == GGML_UNARY_OP_SILU) ||
                n0->op == GGML_OP_SOFT_MAX || n0->op == GGML_OP_MUL_MAT || n0->op == GGML_OP_MUL_MAT_ID || n0->op == GGML_OP_ROPE || n0->op == GGML_OP_MUL ||
                n0->op == GGML_OP_GET_ROWS || n0->op == GGML_OP_SUM_ROWS || n0->op == GGML_OP_ADD || n0->op == GGML_OP_CONT || n0->op == GGML_OP_DIV)
                rd.push_back({ n0, 0, -1, true });
        }
        if (!rd.empty()) {
            auto less = [](const mi355x_backend_ctx::reader_info & x, const ggml_tensor * t) { return x.t < t; };
            std::sort(rd.begin(), rd.end(), [](const auto & x, const auto & y) { return x.t < y.t; });
            for (int i = 0; i < cgraph->n_nodes; ++i) {
                const ggml_tensor * n = cgraph->nodes[i];
                if (is_noop(n)) continue;                                          // a view reads nothing; its readers are found through view_src
                const bool glue = glue_op(n) != 0;
                // a quantized 2-D MUL_MAT of this device takes its src1 pointer through to_qt as well (compute_mul_mat)
                const bool mm_src1 = !glue && n->op == GGML_OP_MUL_MAT && n->src[0] && n->src[1] && !is_split(n->src[0]) && n->src[0]->ne[2] == 1 &&
                                     n->src[0]->ne[3] == 1 && n->src[1]->ne[2] == 1 && n->src[1]->ne[3] == 1;
                for (int j = 0; j < GGML_MAX_SRC && n->src[j]; ++j) {
                    const ggml_tensor * root = n->src[j]->view_src ? n->src[j]->view_src : n->src[j];
                    auto it = std::lower_bound(rd.begin(), rd.end(), root, less);
                    if (it != rd.end() && it->t == root) {
                        ++it->uses;
                        it->last_reader = i;
                        it->glue_only = it->glue_only && (glue || (mm_src1 && j == 1 && it->t->op == GGML_OP_CONT));   // (only the merged-heads CONT is sent to the scratch on this account)
                    }
                }
            }
            // Readers OUTSIDE this cgraph.  The scheduler hands a backend one split = a contiguous run of
            // the graph's nodes; a tensor of this split may also be read by a later split (an op this device refuses inside a layer, a
            // weight kept on the CPU with -ot: ggml-backend.cpp:1355-1448 copies such a tensor out of t->data after graph_compute).  The
            // counts above see this split only, so a candidate counts as fully known only where that can be PROVEN: some later node of
            // this cgraph owns memory that overlaps the candidate's.  ggml-alloc plans the whole graph at once and hands a block out
            // again only behind its tensor's last reader anywhere, so every reader then lies between the two nodes, inside this
            // split.  Anything else (nobody here reuses the block: typically the last tensors of a split) may have readers elsewhere: its
            // count is poisoned, no site matches it, it is computed into its own t->data at its own place.
            // One reverse sweep with the union of the later nodes' byte ranges (a compute arena: a handful of intervals).
            std::map<uintptr_t, uintptr_t> & later = ctx->later_ranges;        // start -> end, disjoint
            later.clear();
            for (int i = cgraph->n_nodes - 1; i >= 0; --i) {
                const ggml_tensor * n = cgraph->nodes[i];
                if (!n->data || ggml_is_empty(n)) continue;
                const uintptr_t a = (uintptr_t) n->data, b = a + ggml_nbytes(n);
                auto it = std::lower_bound(rd.begin(), rd.end(), n, less);
                if (it != rd.end() && it->t == n) {
                    auto nx = later.upper_bound(a);                            // first interval starting behind a
                    bool hit = nx != later.end() && nx->first < b;
                    if (!hit && nx != later.begin()) { auto pv = std::prev(nx); hit = pv->second > a; }
                    if (!hit) { it->uses = 1 << 20; it->glue_only = false; }
                }
                if (n->view_src) continue;                                     // views own nothing (in-place results are views of their operand)
                // insert [a, b), merging what it touches
                uintptr_t lo = a, hi = b;
                auto f = later.upper_bound(lo);
                if (f != later.begin() && std::prev(f)->second >= lo) { --f; lo = f->first; }
                while (f != later.end() && f->first <= hi) { hi = std::max(hi, f->second); f = later.erase(f); }
                later[lo] = hi;
            }
        }
    }
}

// ---- one graph_compute call (analysis / plan / issue apart) -----------------------------------------------------------
// analyze_readers() is the analysis; graph_pass holds what the launch sites share.  Every site_* looks at node i and returns 1 when it
// issued (or deferred) the node, possibly together with later nodes that it marks in done[]; 0 when its pattern is not there (the
// caller tries the next site, in the order below, and finally the node's own launch); -1 when a launch failed.
struct graph_pass {
    mi355x_backend_ctx * ctx;
    ggml_cgraph * cgraph;
    int n_nodes;
    std::vector<char> & done;
    std::vector<const ggml_tensor *> & deferred;
    const mi355x_backend_ctx::reader_info * info(const ggml_tensor * t) const {
        const auto & rd = ctx->readers;
        auto it = std::lower_bound(rd.begin(), rd.end(), t, [](const mi355x_backend_ctx::reader_info & x, const ggml_tensor * y) { return x.t < y; });
        return it != rd.end() && it->t == t ? &*it : nullptr;
    }
    bool single_use(const ggml_tensor * t) const {
        const auto * r = info(t);
        return r && r->uses == 1;
    }
    int site_deferred_silu_mul(int i, ggml_tensor * node, int gop);
    int site_hold_silu(int i, ggml_tensor * node, int gop);
    int site_add_rms_norm(int i, ggml_tensor * node, int gop);
    int site_attention(int i, ggml_tensor * node, int gop);
    int site_kqv_into_merged_heads(int i, ggml_tensor * node, int gop);
    int site_moe_combine(int i, ggml_tensor * node, int gop);
    int site_moe_router(int i, ggml_tensor * node, int gop);
    int site_rope_kv_attention(int i, ggml_tensor * node, int gop);
};

// a MUL whose operand is a SILU that was held back (site_hold_silu): silu(gate) * up in one pass, or folded into the staging of the MUL_MAT behind it
int graph_pass::site_deferred_silu_mul(int i, ggml_tensor * node, int gop) {
    (void) gop;
    enum ggml_status st;
    if (deferred[i]) {
        // MUL whose operand is a SILU that was held back: silu(gate) * up in one pass
        const ggml_tensor * silu = deferred[i];
        const ggml_tensor * gate = silu->src[0], * up = node->src[0] == silu ? node->src[1] : node->src[0];
        // prompt batch, and the product feeds exactly one quantized MUL_MAT right behind it (ffn_down): that MUL_MAT's
        // activation prep reads gate and up itself; no launch and no round trip for the product
        int j = i + 1;
        while (j < n_nodes && (done[j] || is_noop(cgraph->nodes[j]))) ++j;
        if (j < n_nodes && node->ne[1] > QMM_MATVEC_MAX_N && node->ne[2] == 1 && node->ne[3] == 1 && single_use(node) &&
            !(node->flags & GGML_TENSOR_FLAG_OUTPUT) && !getenv("GGML_MI355X_PREC")) {
            const ggml_tensor * mm = cgraph->nodes[j], * w = mm->src[0];
            const qmm_tensor qg = to_qt(gate, ctx), qu = to_qt(up, ctx);      // resolved now: a redirect may expire before node j
            if (mm->op == GGML_OP_MUL_MAT && !glue_op(mm) && mm->src[1] == node && supports_mul_mat(mm) && is_ours(w) && !is_split(w) &&
                w->ne[2] == 1 && w->ne[3] == 1 && gate->nb[0] == 4 && up->nb[0] == 4 && gate->nb[1] % 16 == 0 && up->nb[1] % 16 == 0 &&
                (uintptr_t) qg.data % 16 == 0 && (uintptr_t) qu.data % 16 == 0) {
                ctx->swiglu_in[j] = { (const float *) qg.data, (const float *) qu.data, (int64_t) (gate->nb[1] / 4), (int64_t) (up->nb[1] / 4) };
                return 1;
            }
        }
        st = compute_glue(ctx, node, QMM_OP_SILU_MUL, gate, up, nullptr);
        if (st != GGML_STATUS_SUCCESS) return -1;
        return 1;
    }
    return 0;
}

// SILU read by one MUL further down (build_ffn: gate, silu, up, mul): held back for that MUL
int graph_pass::site_hold_silu(int i, ggml_tensor * node, int gop) {
    (void) gop;
    const ggml_tensor * other = nullptr;
    if (node->op == GGML_OP_UNARY && ggml_get_unary_op(node) == GGML_UNARY_OP_SILU && single_use(node)) {
        // find the MUL that reads it (build_ffn: gate, silu, up, mul): hold the SILU back when nothing in between
        // writes over its input
        int j = i + 1;
        bool safe = true;
        for (; j < n_nodes && j <= i + LOOKAHEAD; ++j) {
            const ggml_tensor * t = cgraph->nodes[j];
            if (t->src[0] == node || t->src[1] == node) break;
            if (!done[j] && !is_noop(t) && ranges_overlap(t, node->src[0])) safe = false;
        }
        if (safe && j < n_nodes && j <= i + LOOKAHEAD && fused_pair(node, cgraph->nodes[j], &other) == QMM_OP_SILU_MUL) {
            deferred[j] = node;
            return 1;
        }
    }
    return 0;
}

// residual ADD -> RMS_NORM -> MUL(w): one launch with two results
int graph_pass::site_add_rms_norm(int i, ggml_tensor * node, int gop) {
    (void) gop;
    const ggml_tensor * other = nullptr;
    if (node->op == GGML_OP_ADD && i + 2 < n_nodes && !GGML_MI355X_FUSE_OFF()) {
        // residual add -> RMS_NORM -> MUL by the norm weight: one pass with two results
        ggml_tensor * rn = cgraph->nodes[i + 1], * mul = cgraph->nodes[i + 2];
        if (rn->op == GGML_OP_RMS_NORM && rn->src[0] == node && single_use(rn) && fused_pair(rn, mul, &other) == QMM_OP_RMS_NORM_MUL) {
            const qmm_tensor a = to_qt(node->src[0], ctx), b = to_qt(node->src[1], ctx), w = to_qt(other, ctx), sum = to_qt(node, ctx), d = to_qt(mul, ctx);
            // A prompt batch whose normed rows are read by ONE group of quantized MUL_MATs right behind the MUL (q / k / v, gate / up) and by
            // nothing else: no launch here; the group's activation prep adds, norms and stores the sum (compute_mul_mat; round 3)
            if (node->ne[1] > QMM_MATVEC_MAX_N && node->ne[2] == 1 && node->ne[3] == 1 && ggml_are_same_shape(node->src[0], node->src[1]) &&
                ggml_are_same_shape(node, node->src[0]) && !getenv("GGML_MI355X_PREP_NORM_OFF")) {
                const auto * ri = info(mul);
                int found = 0, first = -1;
                qmm_weight gw[4];
                for (int j = i + 3; j < n_nodes && j <= i + 3 + LOOKAHEAD && ri && found < 4; ++j) {
                    const ggml_tensor * t = cgraph->nodes[j];
                    if (done[j] || is_noop(t)) continue;
                    if (t->op == GGML_OP_MUL_MAT && !glue_op(t) && t->src[1] == mul && supports_mul_mat(t) && is_ours(t->src[0]) && !is_split(t->src[0]) &&
                        t->src[0]->ne[2] == 1 && t->src[0]->ne[3] == 1) {
                        if (first < 0) first = j;
                        gw[found++] = qmm_weight{ t->src[0]->data, (int64_t) t->src[0]->nb[1], t->src[0]->ne[1], nullptr, 0, weight_type(ctx, t->src[0]) };
                    } else if (first < 0) {
                        break;                                                         // something else reads or runs first: keep the graph's order
                    }
                }
                const ggml_tensor * x0 = node->src[0], * x1 = node->src[1];
                if (ri && found >= 1 && found == ri->uses && !(mul->flags & GGML_TENSOR_FLAG_OUTPUT) && !(node->flags & GGML_TENSOR_FLAG_OUTPUT) &&
                    x0->type == GGML_TYPE_F32 && x1->type == GGML_TYPE_F32 && x0->nb[0] == 4 && x1->nb[0] == 4 && node->nb[0] == 4 &&
                    x0->nb[1] % 16 == 0 && x1->nb[1] % 16 == 0 && node->nb[1] % 16 == 0 && (uintptr_t) a.data % 16 == 0 && (uintptr_t) b.data % 16 == 0 &&
                    (uintptr_t) sum.data % 16 == 0 && (uintptr_t) other->data % 16 == 0 &&
                    // the sum is stored while other rows of the operands are still read: in place over an operand only with the same rows
                    early_write_ok(node, { x0 }, x0) && early_write_ok(node, { x1 }, x1) && early_write_ok(node, { other }) &&
                    qmm_mul_mat_group_norm_supported(ctx->dev->qmm, gw, found, node->ne[0], node->ne[1])) {
                    ctx->pending_norm = { rn, mul, other, found, node };
                    done[i + 1] = done[i + 2] = 1;
                    if (dbg()) fprintf(stderr, "held back: add + norm (%s, %s) for the prep of %d MUL_MATs\n", node->name, mul->name, found);
                    return 1;
                }
            }
            // `mul`'s buffer is written two nodes early: it may be the block of an ADD operand that dies here (same rows: fine, a
            // workgroup holds its row in registers before it stores; anything else: keep the graph's order)
            const bool e_ok = early_write_ok(mul, { node->src[0], node->src[1], other, node }, nullptr) ||
                              (early_write_ok(mul, { node->src[1], other, node }, nullptr) && early_write_ok(mul, { node->src[0] }, node->src[0])) ||
                              (early_write_ok(mul, { node->src[0], other, node }, nullptr) && early_write_ok(mul, { node->src[1] }, node->src[1]));
            if (e_ok && qmm_op_add_rms_norm_supported(&a, &b, &w, &sum, &d)) {
                float eps;
                memcpy(&eps, rn->op_params, sizeof(float));
                if (qmm_op_add_rms_norm(ctx->dev->qmm, &a, &b, &w, &sum, &d, eps, qmm_stream(ctx->dev->qmm))) {
                    GGML_LOG_ERROR("MI355X ADD+RMS_NORM(%s): %s\n", node->name, qmm_last_error());
                    return -1;
                }
                done[i + 1] = done[i + 2] = 1;
                return 1;
            }
        }
    }
    return 0;
}

// kq -> soft_max -> kqv -> permute -> cont (build_attn_mha) as one launch
int graph_pass::site_attention(int i, ggml_tensor * node, int gop) {
    (void) gop;
    if (gop == QMM_OP_MUL_MAT_F && node->src[0]->type == GGML_TYPE_F16 && single_use(node) && !GGML_MI355X_FUSE_OFF()) {
        // kq -> soft_max -> kqv -> permute -> cont (build_attn_mha): one launch, for a few tokens (qmm_attn_decode) and for prompt
        // batches whose scores fit LDS (qmm_attn_prefill)
        int idx[4], k = 0;
        for (int j = i + 1; j < n_nodes && j <= i + 8 && k < 4; ++j) {
            const ggml_tensor * t = cgraph->nodes[j];
            if (t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_TRANSPOSE) continue;
            idx[k++] = j;
        }
        if (k == 4) {
            ggml_tensor * sm = cgraph->nodes[idx[0]], * kqv = cgraph->nodes[idx[1]], * pm = cgraph->nodes[idx[2]], * ct = cgraph->nodes[idx[3]];
            float scale, max_bias;
            memcpy(&scale, (const float *) sm->op_params + 0, sizeof(float));
            memcpy(&max_bias, (const float *) sm->op_params + 1, sizeof(float));
            if (sm->op == GGML_OP_SOFT_MAX && sm->src[0] == node && sm->src[1] && max_bias == 0.0f && single_use(sm) &&
                kqv->op == GGML_OP_MUL_MAT && kqv->src[1] == sm && kqv->src[0]->type == GGML_TYPE_F16 && single_use(kqv) &&
                pm->op == GGML_OP_PERMUTE && pm->src[0] == kqv && pm->ne[0] == kqv->ne[0] && pm->ne[1] == kqv->ne[2] &&
                pm->ne[2] == kqv->ne[1] && ct->op == GGML_OP_CONT && ct->src[0] == pm &&
                ((early_write_ok(ct, { node->src[0], kqv->src[0], sm->src[1] }) && early_write_ok(ct, { node->src[1] }, node->src[1], true)) ||
                 hoist_elsewhere(ctx, ct))) {      // ggml-alloc puts ct across the dead Q blocks in llama.cpp's layers: written to the scratch instead, wo reads it there
                const qmm_tensor q = to_qt(node->src[1], ctx), kk = to_qt(node->src[0], ctx), v = to_qt(kqv->src[0], ctx), m = to_qt(sm->src[1], ctx), d = to_qt(ct, ctx);
                const bool few = qmm_attn_decode_supported(&q, &kk, &v, &m, &d) != 0;
                if (few || qmm_attn_prefill_supported(&q, &kk, &v, &m, &d)) {
                    if (few ? qmm_attn_decode(ctx->dev->qmm, &q, &kk, &v, &m, &d, scale, qmm_stream(ctx->dev->qmm))
                            : qmm_attn_prefill(ctx->dev->qmm, &q, &kk, &v, &m, &d, scale, qmm_stream(ctx->dev->qmm))) {
                        GGML_LOG_ERROR("MI355X attention(%s): %s\n", node->name, qmm_last_error());
                        return -1;
                    }
                    if (dbg()) fprintf(stderr, "fused: attention (%s, %s)\n", few ? "few tokens" : "prompt", node->name);
                    for (int j = 0; j < 4; ++j) done[idx[j]] = 1;
                    return 1;
                }
            }
        }
    }
    return 0;
}

// kqv -> permute -> cont: the product written straight into the merged-heads layout
int graph_pass::site_kqv_into_merged_heads(int i, ggml_tensor * node, int gop) {
    (void) gop;
    if (gop == QMM_OP_MUL_MAT_F && node->src[0]->type == GGML_TYPE_F16 && single_use(node) && !GGML_MI355X_FUSE_OFF() && !done[i]) {
        // kqv -> permute(0, 2, 1, 3) -> cont (build_attn_mha's head merge) at any batch size: the product is written
        // straight into the cont's layout (dst strides of dims 1 and 2 swapped), the copy never runs
        int jp = i + 1;
        while (jp < n_nodes && (cgraph->nodes[jp]->op == GGML_OP_RESHAPE || cgraph->nodes[jp]->op == GGML_OP_VIEW)) ++jp;
        int jc = jp + 1;
        while (jc < n_nodes && (cgraph->nodes[jc]->op == GGML_OP_RESHAPE || cgraph->nodes[jc]->op == GGML_OP_VIEW)) ++jc;
        if (jc < n_nodes) {
            const ggml_tensor * pm = cgraph->nodes[jp], * ct = cgraph->nodes[jc];
            if (pm->op == GGML_OP_PERMUTE && pm->src[0] == node && ct->op == GGML_OP_CONT && ct->src[0] == pm && !done[jc] &&
                pm->ne[0] == node->ne[0] && pm->ne[1] == node->ne[2] && pm->ne[2] == node->ne[1] && node->ne[3] == 1 &&
                ct->type == GGML_TYPE_F32 && ggml_is_contiguous(ct) && ggml_nelements(ct) == ggml_nelements(node) &&
                early_write_ok(ct, { node->src[0], node->src[1] })) {
                qmm_tensor d = to_qt(node, ctx);
                d.data = to_qt(ct, ctx).data;                                 // element (d, n, h) of kqv = element (d, h, n) of the merged result (ct may live in the scratch)
                d.nb[1] = (int64_t) node->ne[0] * node->ne[2] * 4;
                d.nb[2] = (int64_t) node->ne[0] * 4;
                d.nb[3] = (int64_t) ggml_nbytes(ct);
                const qmm_tensor a = to_qt(node->src[0], ctx), b = to_qt(node->src[1], ctx);
                if (qmm_op_supported(QMM_OP_MUL_MAT_F, &a, &b, nullptr, &d)) {
                    if (qmm_op_compute(ctx->dev->qmm, QMM_OP_MUL_MAT_F, &a, &b, nullptr, &d, qmm_stream(ctx->dev->qmm))) {
                        GGML_LOG_ERROR("MI355X MUL_MAT(%s) into merged heads: %s\n", node->name, qmm_last_error());
                        return -1;
                    }
                    done[jc] = 1;
                    return 1;
                }
            }
        }
    }
    return 0;
}

// experts * weights and the sum over the used experts (build_moe_ffn's tail): one launch
int graph_pass::site_moe_combine(int i, ggml_tensor * node, int gop) {
    (void) gop;
    if (node->op == GGML_OP_MUL && node->src[1]->ne[0] == 1 && node->ne[1] >= 2 && node->ne[1] == node->src[1]->ne[1] && node->ne[3] == 1 &&
        !GGML_MI355X_FUSE_OFF()) {
        // experts * weights and the sum over the used experts through 2-D views (build_moe_ffn's tail): one launch
        const int U = (int) node->ne[1];
        const auto * rm = info(node);
        int idx[64], k = 0;
        for (int j = i + 1; j < n_nodes && j <= i + 4 * U + 4 && k < U - 1; ++j) {
            const ggml_tensor * t = cgraph->nodes[j];
            if (done[j] || is_noop(t)) continue;
            if (t->op != GGML_OP_ADD) break;
            idx[k++] = j;
        }
        const auto root = [](const ggml_tensor * t) { return t->view_src ? t->view_src : t; };
        const auto is_slice = [&](const ggml_tensor * v, int u) {           // view_2d(experts, E, N, nb[2], u * nb[1])
            return root(v) == node && v->ne[0] == node->ne[0] && v->ne[1] == node->ne[2] && v->ne[2] == 1 && v->nb[1] == node->nb[2] &&
                   (const char *) v->data == (const char *) node->data + (size_t) u * node->nb[1];
        };
        bool ok = k == U - 1 && rm && rm->uses == U && U <= 64 && !(node->flags & GGML_TENSOR_FLAG_OUTPUT);
        for (int a = 0; ok && a < U - 1; ++a) {
            const ggml_tensor * ad = cgraph->nodes[idx[a]];
            ok = is_slice(ad->src[1], a + 1) && (a == 0 ? is_slice(ad->src[0], 0) : ad->src[0] == cgraph->nodes[idx[a - 1]]) &&
                 (a == U - 2 || (single_use(ad) && !(ad->flags & GGML_TENSOR_FLAG_OUTPUT)));
        }
        if (ok && !getenv("GGML_MI355X_MOE_COMBINE_NORM_OFF")) {
            // Round 3: the residual add, the RMS norm and the norm weight that follow the block in the layer (ffn_moe_out + ffn_inp -> l_out,
            // then the next attn_norm / result_norm) in the same launch: the next three live nodes must be exactly ADD(last, r),
            // RMS_NORM(add), MUL(rms, w).  Never written then: the product, the partial sums, the block's output and the un-weighted
            // norm: each has exactly the readers of the pattern (closed readers: analyze_readers).  Written HERE, U + 2 nodes early,
            // while other workgroups (one per token) still read the expert rows, the weights and the residual: l_out and the normed row
            // must be clear of the experts and the weights (another layout) and may sit on the residual only as the very same rows.
            const ggml_tensor * last = cgraph->nodes[idx[U - 2]];
            int ja = idx[U - 2] + 1;
            while (ja < n_nodes && (done[ja] || is_noop(cgraph->nodes[ja]))) ++ja;
            const ggml_tensor * other = nullptr;
            if (ja + 2 < n_nodes && single_use(last) && !(last->flags & GGML_TENSOR_FLAG_OUTPUT)) {
                ggml_tensor * add = cgraph->nodes[ja], * rn = cgraph->nodes[ja + 1], * mul = cgraph->nodes[ja + 2];
                if (add->op == GGML_OP_ADD && (add->src[0] == last || add->src[1] == last) && add->src[0] != add->src[1] && rn->op == GGML_OP_RMS_NORM &&
                    rn->src[0] == add && single_use(rn) && !(rn->flags & GGML_TENSOR_FLAG_OUTPUT) && fused_pair(rn, mul, &other) == QMM_OP_RMS_NORM_MUL) {
                    const ggml_tensor * r = add->src[0] == last ? add->src[1] : add->src[0];
                    const auto clear_of = [&](const ggml_tensor * t) {
                        return early_write_ok(t, { node->src[0], node->src[1], other }) && (early_write_ok(t, { r }) || early_write_ok(t, { r }, r));
                    };
                    // one token = one workgroup, which holds all its inputs in registers before it stores (moe_combine_add_norm_kernel): the
                    // results may then lie on any input (ggml-alloc does put l_out on the dead router weights and the normed row on the residual)
                    const bool one = add->ne[1] == 1 && add->ne[2] == 1 && add->ne[3] == 1;
                    if ((one || (clear_of(add) && clear_of(mul))) && !ranges_overlap(add, mul)) {
                        const qmm_tensor x = to_qt(node->src[0], ctx), w = to_qt(node->src[1], ctx), qb = to_qt(r, ctx), nw = to_qt(other, ctx), qs = to_qt(add, ctx),
                                         qd = to_qt(mul, ctx);
                        if (qmm_moe_combine_add_rms_norm_supported(&x, &w, &qb, &nw, &qs, &qd)) {
                            float eps;
                            memcpy(&eps, rn->op_params, sizeof(float));
                            if (qmm_moe_combine_add_rms_norm(ctx->dev->qmm, &x, &w, &qb, &nw, &qs, &qd, eps, qmm_stream(ctx->dev->qmm))) {
                                GGML_LOG_ERROR("MI355X MoE combine + ADD + RMS_NORM(%s): %s\n", node->name, qmm_last_error());
                                return -1;
                            }
                            if (dbg()) fprintf(stderr, "fused: moe combine + add + rms_norm (%s)\n", add->name);
                            for (int a = 0; a < U - 1; ++a) done[idx[a]] = 1;
                            done[ja] = done[ja + 1] = done[ja + 2] = 1;
                            return 1;
                        }
                    }
                }
            }
        }
        if (ok) {
            const ggml_tensor * last = cgraph->nodes[idx[U - 2]];
            // the launch writes the LAST add's buffer at the MUL's position, while other workgroups still read the experts and the
            // weights: ggml-alloc may have put it on either (both are dead at the add's place in the graph).  A token's U expert
            // rows and its one output row are laid out differently, so not even the same address is in place
            if (!early_write_ok(last, { node->src[0], node->src[1] }) && !hoist_elsewhere(ctx, last)) return 0;
            const qmm_tensor x = to_qt(node->src[0], ctx), w = to_qt(node->src[1], ctx), o = to_qt(last, ctx);
            if (qmm_moe_combine_supported(&x, &w, &o)) {
                if (qmm_moe_combine(ctx->dev->qmm, &x, &w, &o, qmm_stream(ctx->dev->qmm))) {
                    GGML_LOG_ERROR("MI355X MoE combine(%s): %s\n", node->name, qmm_last_error());
                    return -1;
                }
                for (int a = 0; a < U - 1; ++a) done[idx[a]] = 1;
                return 1;
            }
        }
    }
    return 0;
}

// soft_max -> argsort -> get_rows -> sum_rows -> div behind the router logits: one launch
int graph_pass::site_moe_router(int i, ggml_tensor * node, int gop) {
    (void) gop;
    // entered one node earlier, at the logits' own MUL_MAT (F32 gate_inp, a few tokens), the launch computes the logits too
    // (qmm_moe_router_logits, round 3): the logits land in their own buffer at their own place in the graph, so nothing about them
    // changes for any other reader; the ids and the weights are written one node earlier than before, now also beside the reads of
    // the MUL_MAT's src1, which joins the operands they must stay clear of
    ggml_tensor * lgt = nullptr;
    // ... and entered two nodes before that, at the RMS norm whose product with ffn_norm's weights is the logits' src1 (the MoE branch of
    // the layer, src/llama-model.cpp:4301-4305), the launch forms the normed row too and stores it for the expert MUL_MAT_IDs
    // (qmm_moe_router_logits_norm): the separate norm launch (4.75 us of a Mixtral layer's ~84) is gone
    ggml_tensor * nrm = nullptr, * nmul = nullptr;
    const ggml_tensor * nwgt = nullptr;
    int i_nmul = -1, i_lgt = -1;
    if (node->op == GGML_OP_RMS_NORM && i + 2 < n_nodes && node->ne[1] <= 8 && node->ne[2] == 1 && node->ne[3] == 1 && single_use(node) && !GGML_MI355X_FUSE_OFF() &&
        !getenv("GGML_MI355X_ROUTER_NORM_OFF") && !getenv("GGML_MI355X_ROUTER_LOGITS_OFF")) {
        ggml_tensor * m = cgraph->nodes[i + 1];
        int j2 = i + 2;                                                         // (the 3-D reshape of the normed row for the expert MUL_MAT_IDs sits in between)
        while (j2 < n_nodes && (done[j2] || is_noop(cgraph->nodes[j2]))) ++j2;
        if (j2 >= n_nodes || done[i + 1] || fused_pair(node, m, &nwgt) != QMM_OP_RMS_NORM_MUL) return 0;
        ggml_tensor * t = cgraph->nodes[j2];
        if (t->op != GGML_OP_MUL_MAT || t->src[1] != m || t->src[0]->type != GGML_TYPE_F32 || (m->flags & GGML_TENSOR_FLAG_OUTPUT)) return 0;
        nrm = node;  nmul = m;  i_nmul = i + 1;  i_lgt = j2;
        i = j2;
        node = t;
    }
    if (node->op == GGML_OP_MUL_MAT && node->src[0]->type == GGML_TYPE_F32 && node->src[1]->type == GGML_TYPE_F32 && node->type == GGML_TYPE_F32 && node->ne[0] <= 64 &&
        node->ne[1] <= 8 && node->ne[2] == 1 && node->ne[3] == 1 && !GGML_MI355X_FUSE_OFF() && !getenv("GGML_MI355X_ROUTER_LOGITS_OFF")) {
        int j = i + 1;
        while (j < n_nodes && (done[j] || is_noop(cgraph->nodes[j]))) ++j;
        if (j >= n_nodes || cgraph->nodes[j]->op != GGML_OP_SOFT_MAX || cgraph->nodes[j]->src[0] != node) return 0;
        lgt = node;
        i = j;
        node = cgraph->nodes[j];
    } else if (nrm) {
        if (dbg()) fprintf(stderr, "fusion declined: moe router with its norm (%s): not the logits' MUL_MAT\n", nmul->name);
        return 0;
    }
    if (node->op == GGML_OP_SOFT_MAX && !node->src[1] && node->ne[0] <= 64 && node->ne[2] == 1 && node->ne[3] == 1 && !GGML_MI355X_FUSE_OFF()) {
        // the MoE router behind its logits (build_moe_ffn): soft_max -> argsort (top_k view) -> get_rows -> sum_rows -> div
        float scale, max_bias;
        memcpy(&scale, (const float *) node->op_params + 0, sizeof(float));
        memcpy(&max_bias, (const float *) node->op_params + 1, sizeof(float));
        // llama.cpp's graph order puts get_rows / sum_rows / div (the weights, needed only by the final mul) BEHIND the expert
        // MUL_MAT_IDs: the argsort follows the soft_max directly, the other three are looked for further down and run here,
        // early (their inputs exist; where the div's buffer is still in use at this point the weights go to the scratch)
        const auto root = [](const ggml_tensor * t) { return t->view_src ? t->view_src : t; };
        int idx[4], k = 0;
        std::vector<const ggml_tensor *> & skipped = ctx->skipped;
        skipped.clear();
        for (int j = i + 1; j < n_nodes && j <= i + 64 && k < 4; ++j) {
            const ggml_tensor * t = cgraph->nodes[j];
            if (done[j] || is_noop(t)) continue;
            const bool want = (k == 0 && t->op == GGML_OP_ARGSORT && t->src[0] == node) ||
                              (k == 1 && t->op == GGML_OP_GET_ROWS && root(t->src[0]) == node && root(t->src[1]) == cgraph->nodes[idx[0]]) ||
                              (k == 2 && t->op == GGML_OP_SUM_ROWS && root(t->src[0]) == cgraph->nodes[idx[1]]) ||
                              (k == 3 && t->op == GGML_OP_DIV && root(t->src[0]) == cgraph->nodes[idx[1]] && t->src[1] == cgraph->nodes[idx[2]]);
            if (want) idx[k++] = j;
            else if (k == 0) break;                                             // the argsort must come first
            else skipped.push_back(t);
        }
        if (k == 4 && scale == 1.0f && max_bias == 0.0f) {
            ggml_tensor * as = cgraph->nodes[idx[0]], * gr = cgraph->nodes[idx[1]], * sr = cgraph->nodes[idx[2]], * dv = cgraph->nodes[idx[3]];
            const auto * ri = info(node), * rg = info(gr), * rs = info(sr);
            if (as->op == GGML_OP_ARGSORT && as->src[0] == node && as->op_params[0] == GGML_SORT_ORDER_DESC &&
                gr->op == GGML_OP_GET_ROWS && root(gr->src[0]) == node && gr->src[0]->ne[0] == 1 && root(gr->src[1]) == as &&
                gr->src[1]->data == as->data && gr->src[1]->nb[1] == as->nb[1] && gr->src[1]->ne[1] == as->ne[1] &&
                sr->op == GGML_OP_SUM_ROWS && root(sr->src[0]) == gr && dv->op == GGML_OP_DIV && root(dv->src[0]) == gr && dv->src[1] == sr &&
                ri && ri->uses == 2 && rg && rg->uses == 2 && rs && rs->uses == 1 && ggml_is_contiguous(dv) && ggml_is_contiguous(gr) &&
                !(node->flags & GGML_TENSOR_FLAG_OUTPUT) && !(gr->flags & GGML_TENSOR_FLAG_OUTPUT) && !(sr->flags & GGML_TENSOR_FLAG_OUTPUT)) {
                const int64_t n_used = gr->src[1]->ne[0];
                // the ids are written while other waves (one per token) still read logits rows: in place only at the very same address
                // with the same rows (a wave reads its row first).  The weights are written now, not at the div's place in the graph:
                // their block must be free here AND clear of the logits: at the div's place the logits are dead, so a non-inplace
                // dv may sit on them with rows of 4 * n_used bytes against 4 * n_expert; otherwise into the scratch
                if (!early_write_ok(as, { node->src[0], lgt ? lgt->src[1] : nullptr, lgt ? lgt->src[0] : nullptr }, node->src[0])) return 0;
                if (!(can_hoist(dv, skipped) && early_write_ok(dv, { node->src[0], as, lgt ? lgt->src[1] : nullptr, lgt ? lgt->src[0] : nullptr })) && !hoist_elsewhere(ctx, dv)) return 0;
                const qmm_tensor lg = to_qt(node->src[0], ctx), ids = to_qt(as, ctx), w = to_qt(dv, ctx);
                if (nrm) {
                    // the normed row is written where the MUL would put it, two nodes early and beside this launch's reads of the un-normed
                    // rows (in place over them is fine: a workgroup holds its row in registers before it stores) and of the weights; the
                    // router's outputs stay clear of all of them
                    const ggml_tensor * x0 = nrm->src[0];
                    if (!(early_write_ok(nmul, { nwgt, lgt->src[0] }) && early_write_ok(nmul, { x0 }, x0))) return 0;
                    if (!early_write_ok(lgt, { x0, nwgt, nmul, lgt->src[0] }) || !early_write_ok(as, { x0, nwgt, nmul }) ||
                        (to_qt(dv, ctx).data == dv->data && !early_write_ok(dv, { x0, nwgt, nmul }))) return 0;
                    const qmm_tensor gi = to_qt(lgt->src[0], ctx), xin = to_qt(x0, ctx), nw = to_qt(nwgt, ctx), ny = to_qt(nmul, ctx);
                    float eps;
                    memcpy(&eps, nrm->op_params, sizeof(float));
                    if (!is_ours(lgt->src[0]) || is_split(lgt->src[0]) || !qmm_moe_router_logits_norm_supported(&gi, &xin, &nw, &ny, &lg, &ids, &w, n_used)) {
                        if (dbg()) fprintf(stderr, "fusion declined: moe router with its norm (%s): operands not supported\n", nmul->name);
                        return 0;
                    }
                    if (qmm_moe_router_logits_norm(ctx->dev->qmm, &gi, &xin, &nw, eps, &ny, &lg, &ids, &w, n_used, 1, qmm_stream(ctx->dev->qmm))) {
                        GGML_LOG_ERROR("MI355X MoE router with norm and logits(%s): %s\n", node->name, qmm_last_error());
                        return -1;
                    }
                    if (dbg()) fprintf(stderr, "fused: moe router (%s) with its logits (%s) and their norm (%s)\n", node->name, lgt->name, nmul->name);
                    done[i_nmul] = done[i_lgt] = done[i] = 1;                   // mul, logits, soft_max; the caller entered with the norm
                    for (int j = 0; j < 4; ++j) done[idx[j]] = 1;
                    return 1;
                }
                if (lgt) {
                    const qmm_tensor gi = to_qt(lgt->src[0], ctx), xin = to_qt(lgt->src[1], ctx);
                    if (!is_ours(lgt->src[0]) || is_split(lgt->src[0]) || !qmm_moe_router_logits_supported(&gi, &xin, &lg, &ids, &w, n_used)) return 0;
                    if (qmm_moe_router_logits(ctx->dev->qmm, &gi, &xin, &lg, &ids, &w, n_used, 1, qmm_stream(ctx->dev->qmm))) {
                        GGML_LOG_ERROR("MI355X MoE router with logits(%s): %s\n", node->name, qmm_last_error());
                        return -1;
                    }
                    if (dbg()) fprintf(stderr, "fused: moe router (%s) with its logits (%s)\n", node->name, lgt->name);
                    done[i] = 1;                                                // the soft_max; the caller marks the MUL_MAT it entered with
                    for (int j = 0; j < 4; ++j) done[idx[j]] = 1;
                    return 1;
                }
                if (qmm_moe_router_supported(&lg, &ids, &w, n_used)) {
                    if (qmm_moe_router(ctx->dev->qmm, &lg, &ids, &w, n_used, 1, qmm_stream(ctx->dev->qmm))) {
                        GGML_LOG_ERROR("MI355X MoE router(%s): %s\n", node->name, qmm_last_error());
                        return -1;
                    }
                    if (dbg()) fprintf(stderr, "fused: moe router (%s)\n", node->name);
                    for (int j = 0; j < 4; ++j) done[idx[j]] = 1;
                    return 1;
                }
            }
        }
    }
    return 0;
}

// rope(q) with rope(k) -> K cache, v -> V cache and, for a few tokens, the attention: one launch
int graph_pass::site_rope_kv_attention(int i, ggml_tensor * node, int gop) {
    (void) gop;
    if (node->op == GGML_OP_ROPE && !GGML_MI355X_FUSE_OFF()) {
        // rope(q) with, from further down the graph, rope(k) -> K cache and v -> V cache (build_attn's two ggml_cpy): one launch
        // (any batch size: at 512 tokens 29 us of four launches become one, pp512 31.2k -> 32.0k).
        // Their inputs must exist already (k and v were hoisted into the q/k/v group); the cache is not compute-buffer
        // memory, so storing early cannot collide with anything in between.
        auto ready = [&](const ggml_tensor * t) {                            // was t's root produced before this point?
            const ggml_tensor * root = t->view_src ? t->view_src : t;
            for (int j = i + 1; j < n_nodes && j <= i + 2 * LOOKAHEAD; ++j)
                if (cgraph->nodes[j] == root) return done[j] != 0;
            return true;
        };
        int jk = -1, jck = -1, jcv = -1;
        for (int j = i + 1; j < n_nodes && j <= i + 2 * LOOKAHEAD; ++j) {
            const ggml_tensor * t = cgraph->nodes[j];
            if (done[j] || is_noop(t)) continue;
            if (jk < 0 && t->op == GGML_OP_ROPE && t->src[1] == node->src[1] && t->src[2] == node->src[2] && t->ne[0] == node->ne[0] &&
                !memcmp(t->op_params, node->op_params, sizeof(t->op_params)) && t->type == GGML_TYPE_F32 && ggml_is_contiguous(t) &&
                single_use(t) && ready(t->src[0])) {
                jk = j;
            } else if (jk >= 0 && jck < 0 && t->op == GGML_OP_CPY && t->src[0] == cgraph->nodes[jk] && t->type == GGML_TYPE_F16 && ggml_is_contiguous(t)) {
                jck = j;
            } else if (jcv < 0 && t->op == GGML_OP_CPY && t->type == GGML_TYPE_F16 && t->src[0]->type == GGML_TYPE_F32 &&
                       (t->src[0]->view_src ? t->src[0]->view_src : t->src[0])->op == GGML_OP_MUL_MAT && ready(t->src[0]) &&
                       (jk < 0 || t->src[0] != cgraph->nodes[jk])) {
                jcv = j;
            } else if (t->op == GGML_OP_MUL_MAT || t->op == GGML_OP_SOFT_MAX) {
                break;                                                         // attention starts: nothing to find beyond
            }
        }
        if (jk >= 0 && jck < 0) jk = -1;                                       // rope(k) without its store stays where it is
        if (jk >= 0 || jcv >= 0) {
            const qmm_tensor q = to_qt(node->src[0], ctx), pos = to_qt(node->src[1], ctx), qd = to_qt(node, ctx);
            qmm_tensor ff{}, k{}, kd{}, v{}, vd{};
            if (node->src[2]) ff = to_qt(node->src[2], ctx);
            if (jk >= 0) {
                const ggml_tensor * rk = cgraph->nodes[jk];
                k = to_qt(rk->src[0], ctx);
                kd = to_qt(rk, ctx);                                           // shape of rope(k), bytes of the cache view
                kd.data = cgraph->nodes[jck]->data;
                kd.type = GGML_TYPE_F16;
                kd.nb[0] = 2;
                for (int a = 1; a < 4; ++a) kd.nb[a] = kd.nb[a - 1] * kd.ne[a - 1];
            }
            if (jcv >= 0) {
                v = to_qt(cgraph->nodes[jcv]->src[0], ctx);
                vd = to_qt(cgraph->nodes[jcv], ctx);
            }
            const qmm_tensor * pff = node->src[2] ? &ff : nullptr, * pk = jk >= 0 ? &k : nullptr, * pkd = jk >= 0 ? &kd : nullptr,
                             * pv = jcv >= 0 ? &v : nullptr, * pvd = jcv >= 0 ? &vd : nullptr;
            // ... and when the attention chain follows (few tokens): rope, KV store and attention in one launch
            // (GGML_MI355X_ATTN_ROPE=0: two launches; tg128 395 -> 403 tok/s)
            if (jk >= 0 && jcv >= 0 && node->ne[2] <= 8 && single_use(node) && GGML_MI355X_ATTN_ROPE()) {
                int idx[5], kq_n = 0;
                for (int j = std::max(jck, jcv) + 1; j < n_nodes && j <= i + 4 * LOOKAHEAD && kq_n < 5; ++j) {
                    const ggml_tensor * t = cgraph->nodes[j];
                    if (done[j] || t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_TRANSPOSE) continue;
                    if (kq_n == 0 && t->op == GGML_OP_PERMUTE) continue;                   // q's permute in front of kq
                    idx[kq_n++] = j;
                }
                if (kq_n == 5) {
                    ggml_tensor * kqn = cgraph->nodes[idx[0]], * sm = cgraph->nodes[idx[1]], * kqv = cgraph->nodes[idx[2]], * pm = cgraph->nodes[idx[3]],
                                * ct = cgraph->nodes[idx[4]];
                    float scale, max_bias;
                    memcpy(&scale, (const float *) sm->op_params + 0, sizeof(float));
                    memcpy(&max_bias, (const float *) sm->op_params + 1, sizeof(float));
                    if (kqn->op == GGML_OP_MUL_MAT && kqn->src[0]->type == GGML_TYPE_F16 && kqn->src[1]->op == GGML_OP_PERMUTE &&
                        kqn->src[1]->src[0] == node && single_use(kqn) && sm->op == GGML_OP_SOFT_MAX && sm->src[0] == kqn && sm->src[1] &&
                        max_bias == 0.0f && single_use(sm) && kqv->op == GGML_OP_MUL_MAT && kqv->src[1] == sm &&
                        kqv->src[0]->type == GGML_TYPE_F16 && single_use(kqv) && pm->op == GGML_OP_PERMUTE && pm->src[0] == kqv &&
                        pm->ne[0] == kqv->ne[0] && pm->ne[1] == kqv->ne[2] && pm->ne[2] == kqv->ne[1] && ct->op == GGML_OP_CONT && ct->src[0] == pm &&
                        ((early_write_ok(ct, { node->src[1], node->src[2], cgraph->nodes[jk]->src[0], cgraph->nodes[jcv]->src[0],
                                               kqn->src[0], kqv->src[0], sm->src[1] }) &&
                          early_write_ok(ct, { node->src[0] }, node->src[0], true) && early_write_ok(ct, { node }, node, true)) ||
                         hoist_elsewhere(ctx, ct))) {
                        const qmm_tensor kc = to_qt(kqn->src[0], ctx), vc = to_qt(kqv->src[0], ctx), m = to_qt(sm->src[1], ctx), d = to_qt(ct, ctx);
                        const int64_t off = (const char *) kd.data - (const char *) kc.data;
                        const int64_t j0 = kc.nb[1] > 0 && off >= 0 && off % kc.nb[1] == 0 ? off / kc.nb[1] : -1;
                        const bool v_ok = (const char *) vd.data - (const char *) vc.data == j0 * 2;
                        if (j0 >= 0 && v_ok && qmm_attn_decode_rope_supported(&q, &pos, pff, &qd, &k, &kd, &v, &vd, &kc, &vc, &m, &d, j0)) {
                            if (qmm_attn_decode_rope(ctx->dev->qmm, &q, &pos, pff, &qd, &k, &kd, &v, &vd, &kc, &vc, &m, &d, scale, j0,
                                                     qmm_stream(ctx->dev->qmm))) {
                                GGML_LOG_ERROR("MI355X rope + KV store + attention(%s): %s\n", node->name, qmm_last_error());
                                return -1;
                            }
                            if (dbg()) fprintf(stderr, "fused: rope + kv store + attention (%s)\n", node->name);
                            done[jk] = done[jck] = done[jcv] = 1;
                            for (int j = 0; j < 5; ++j) done[idx[j]] = 1;
                            return 1;
                        }
                    }
                }
            }
            if (qmm_rope_kv_store_supported(&q, &pos, pff, &qd, pk, pkd, pv, pvd)) {
                if (qmm_rope_kv_store(ctx->dev->qmm, &q, &pos, pff, &qd, pk, pkd, pv, pvd, qmm_stream(ctx->dev->qmm))) {
                    GGML_LOG_ERROR("MI355X rope + KV store(%s): %s\n", node->name, qmm_last_error());
                    return -1;
                }
                if (jk >= 0) done[jk] = done[jck] = 1;
                if (jcv >= 0) done[jcv] = 1;
                return 1;
            }
        }
    }
    return 0;
}

static double wall_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static enum ggml_status graph_compute_once(ggml_backend_t backend, struct ggml_cgraph * cgraph, bool * f16_overflow);

// a prompt whose f16-mode prefill met a weight block beyond the f16 range (valid GGUF bits: |d * sc * q| >= 65504) used to fail
// llama_decode although token generation on the same model works (the mat-vec path dots integers).  The library reports it at the
// graph's synchronize; the graph is then issued once more in QMM_PREC_BF16, which has f32's range (NMSE <= 5e-4 against the CPU, the
// reference's own bar), and the device stays in that mode: a model with such a block needs it for every prompt.  Re-issuing is safe:
// a graph's leaves and inputs are never reused by ggml-alloc, every other tensor is rewritten by its node, KV-cache stores rewrite the
// same rows with the same values.
enum ggml_status backend_graph_compute(ggml_backend_t backend, struct ggml_cgraph * cgraph) {
    bool overflow = false;
    enum ggml_status st = graph_compute_once(backend, cgraph, &overflow);
    if (st != GGML_STATUS_SUCCESS && overflow) {
        auto * ctx = (mi355x_backend_ctx *) backend->context;
        GGML_LOG_WARN("MI355X: a weight block of this model exceeds the f16 range; prompt batches on %s run in bf16 from here on (GGML_MI355X_PREC=bf16 avoids the first attempt)\n",
                      ctx->name.c_str());
        if (qmm_set_precision(ctx->dev->qmm, QMM_PREC_BF16)) return st;
        ctx->dev->prefill_bf16 = true;
        ctx->readers_sig = 0;                                   // (nothing cached depends on the mode; start the attempt from a clean analysis all the same)
        st = graph_compute_once(backend, cgraph, &overflow);
    }
    return st;
}

static enum ggml_status graph_compute_once(ggml_backend_t backend, struct ggml_cgraph * cgraph, bool * f16_overflow) {
    auto * ctx = (mi355x_backend_ctx *) backend->context;
    *f16_overflow = false;
    const bool timing = GGML_MI355X_TIMING();
    const uint64_t queued = ++ctx->dev->enq;
    const double t_enter = timing ? wall_us() : 0;
    if (timing) {
        bool one = true;
        for (int i = 0; i < cgraph->n_nodes && one; ++i)
            if (cgraph->nodes[i]->op == GGML_OP_MUL_MAT && cgraph->nodes[i]->ne[2] == 1 && cgraph->nodes[i]->ne[1] > 1) one = false;
        host_timer::flush(one);
    }
    analyze_readers(ctx, cgraph);
    const double t_analyzed = timing ? wall_us() : 0;
    const int n_nodes = cgraph->n_nodes;
    if (timing) {
        if (!ctx->ev_t0) { ctx->ev_t0 = qmm_event_create_timing(ctx->dev->qmm); ctx->ev_t1 = qmm_event_create_timing(ctx->dev->qmm); }
        if (ctx->ev_t0 && ctx->ev_t1) qmm_event_record(ctx->dev->qmm, ctx->ev_t0, qmm_stream(ctx->dev->qmm));
    }
    const bool chain = GGML_MI355X_CHAIN() && !GGML_MI355X_FUSE_OFF();
    if (chain && qmm_chain_begin(ctx->dev->qmm)) {
        GGML_LOG_ERROR("MI355X graph_compute: %s\n", qmm_last_error());
        return GGML_STATUS_FAILED;
    }
    struct chain_guard {            // every return path below stops the recording (and launches what was recorded)
        qmm_ctx * q; bool on;
        ~chain_guard() { if (on) qmm_chain_end(q); }
    } guard{ ctx->dev->qmm, chain };
    std::vector<char> & done = ctx->done;
    done.assign(n_nodes, 0);
    std::vector<const ggml_tensor *> & deferred = ctx->deferred;
    deferred.assign(n_nodes, nullptr);
    ctx->swiglu_in.assign(n_nodes, {});
    graph_pass P{ ctx, cgraph, n_nodes, done, deferred };
    for (int i = 0; i < n_nodes; ++i) {
        struct ggml_tensor * node = cgraph->nodes[i];
        if (done[i] || is_noop(node)) continue;                                      // ggml-hexagon.cpp:5561-5566
        while (!ctx->redirects.empty() && ctx->redirects.front().last_reader < i) ctx->redirects.erase(ctx->redirects.begin());
        if (ctx->redirects.empty()) ctx->hoist_used = 0;       // nothing lives in the scratch any more: the next layer starts at its base again
        enum ggml_status st;
        const int gop = glue_op(node);
        if (gop) {
            const ggml_tensor * other = nullptr;
            int fop = 0;
            int r = 0;
            if (deferred[i]) r = P.site_deferred_silu_mul(i, node, gop);
            if (r == 0) r = P.site_hold_silu(i, node, gop);
            if (r == 0) r = P.site_add_rms_norm(i, node, gop);
            if (r == 0) r = P.site_attention(i, node, gop);
            if (r == 0) r = P.site_kqv_into_merged_heads(i, node, gop);
            if (r == 0) r = P.site_moe_combine(i, node, gop);
            if (r == 0) r = P.site_moe_router(i, node, gop);
            if (r == 0) r = P.site_rope_kv_attention(i, node, gop);
            if (r < 0) return GGML_STATUS_FAILED;
            if (r > 0) continue;
            if (i + 1 < n_nodes && node->op == GGML_OP_RMS_NORM && P.single_use(node)) fop = fused_pair(node, cgraph->nodes[i + 1], &other);
            if (fop == QMM_OP_RMS_NORM_MUL && node->ne[1] <= QMM_MATVEC_MAX_N && node->ne[2] == 1 && node->ne[3] == 1 && !GGML_MI355X_FUSE_OFF()) {
                // every reader of the normed row a quantized MUL_MAT of one group (q/k/v, gate/up, output)?  Then no launch here:
                // compute_mul_mat hands the norm to the mat-vec kernels, or materializes it if the group turns out smaller
                const ggml_tensor * mul = cgraph->nodes[i + 1];
                const auto * ri = P.info(mul);
                const ggml_tensor * x = node->src[0];
                int found = 0, first = -1;
                for (int j = i + 2; j < n_nodes && j <= i + 2 + LOOKAHEAD && ri; ++j) {
                    const ggml_tensor * t = cgraph->nodes[j];
                    if (done[j] || is_noop(t)) continue;
                    if (t->op == GGML_OP_MUL_MAT && !glue_op(t) && t->src[1] == mul && supports_mul_mat(t) && is_ours(t->src[0]) && !is_split(t->src[0]) &&
                        t->src[0]->ne[2] == 1 && t->src[0]->ne[3] == 1) {
                        if (first < 0) first = j;
                        ++found;
                    } else if (first < 0) {
                        break;                                                         // something else reads or runs first: keep the graph's order
                    }
                }
                const int64_t K = node->ne[0], N = node->ne[1];
                if (ri && found == ri->uses && found >= 1 && found <= 4 && !(mul->flags & GGML_TENSOR_FLAG_OUTPUT) && x->nb[0] == 4 && x->nb[1] % 16 == 0 &&
                    (uintptr_t) to_qt(x, ctx).data % 16 == 0 && (uintptr_t) other->data % 16 == 0 && K % 256 == 0 &&
                    (size_t) N * K * 4 + (size_t) N * K * 11 / 8 + 4096 <= 150 * 1024) {
                    ctx->pending_norm = { node, mul, other, found };
                    done[i + 1] = 1;
                    continue;
                }
            }
            if (fop && !(early_write_ok(cgraph->nodes[i + 1], { other }) && early_write_ok(cgraph->nodes[i + 1], { node->src[0] }, node->src[0]))) fop = 0;
            if (fop) {
                ggml_tensor * out = cgraph->nodes[i + 1];
                ggml_tensor tmp = *out;                                              // dst of the pair, carrying the first node's op_params (eps)
                memcpy(tmp.op_params, node->op_params, sizeof(tmp.op_params));
                st = compute_glue(ctx, &tmp, fop, node->src[0], other, nullptr);
                done[i + 1] = 1;
            } else if (gop == QMM_OP_CPY) {
                st = compute_glue(ctx, node, gop, node->src[0], nullptr, nullptr);
            } else {
                st = compute_glue(ctx, node, gop, node->src[0], node->src[1], node->src[2]);
            }
        } else if (node->op == GGML_OP_MUL_MAT && ctx->swiglu_in[i].gate) {
            const ggml_tensor * w = node->src[0];
            const auto & sg = ctx->swiglu_in[i];
            st = GGML_STATUS_SUCCESS;
            if (qmm_mul_mat_swiglu_in(ctx->dev->qmm, weight_type(ctx, w), w->data, w->nb[1], w->ne[0], w->ne[1], sg.gate, sg.ld_gate, sg.up, sg.ld_up,
                                      node->src[1]->ne[1], (float *) node->data, node->nb[1] / sizeof(float), qmm_stream(ctx->dev->qmm))) {
                GGML_LOG_ERROR("MI355X MUL_MAT(%s) with SwiGLU input: %s\n", node->name, qmm_last_error());
                st = GGML_STATUS_FAILED;
            }
        } else if (node->op == GGML_OP_MUL_MAT) {
            st = compute_mul_mat(ctx, cgraph->nodes + i, n_nodes - i, done.data() + i);                      // may hoist later MUL_MATs
        } else if (node->op == GGML_OP_MUL_MAT_ID) {
            st = compute_mul_mat_id(ctx, cgraph->nodes + i, n_nodes - i, done.data() + i);
        } else {
            GGML_LOG_ERROR("MI355X: op %s (%s) is outside the offloaded surface\n", ggml_op_name(node->op), node->name);
            st = GGML_STATUS_FAILED;
        }
        if (st != GGML_STATUS_SUCCESS) return st;
    }
    if (chain) {
        guard.on = false;
        if (qmm_chain_end(ctx->dev->qmm)) {
            GGML_LOG_ERROR("MI355X graph_compute: %s\n", qmm_last_error());
            return GGML_STATUS_FAILED;
        }
    }
    if (timing && ctx->ev_t0 && ctx->ev_t1) qmm_event_record(ctx->dev->qmm, ctx->ev_t1, qmm_stream(ctx->dev->qmm));
    const double t_issued = timing ? wall_us() : 0;
    // the scheduler reads results right after graph_compute/synchronize; a bad expert id surfaces here
    if (const int rc = qmm_synchronize(ctx->dev->qmm, qmm_stream(ctx->dev->qmm))) {
        if (rc == QMM_EUNSUPPORTED && strstr(qmm_last_error(), "non-finite") && !ctx->dev->prefill_bf16) {
            *f16_overflow = true;                               // the caller re-issues the graph in bf16
            return GGML_STATUS_FAILED;
        }
        GGML_LOG_ERROR("MI355X graph_compute: %s\n", qmm_last_error());
        return GGML_STATUS_FAILED;
    }
    free_retired();                 // (the root has waited for every device's slice: nothing queued reads an outgrown staging block)
    { std::lock_guard<std::mutex> lock(ctx->dev->ring_mu); ctx->dev->staged_pending = false; }
    if (ctx->dev->enq.load() == queued) ctx->dev->enq_synced.store(queued);
    if (timing && ctx->ev_t0 && ctx->ev_t1) {
        float ms = 0.0f;
        int64_t n_tok = 1;                                   // tokens of the ubatch = ne[1] of the widest 2-D activation in the graph
        for (int i = 0; i < n_nodes; ++i)
            if (cgraph->nodes[i]->op == GGML_OP_MUL_MAT && cgraph->nodes[i]->ne[2] == 1) n_tok = std::max<int64_t>(n_tok, cgraph->nodes[i]->ne[1]);
        if (!qmm_event_elapsed_ms(ctx->dev->qmm, ctx->ev_t0, ctx->ev_t1, &ms)) {
            if (n_tok == 1) {
                ctx->ms_tg += ms; ctx->graphs_tg++;
                const double t_done = wall_us();
                if (ctx->t_exit > 0) ctx->us_outside += t_enter - ctx->t_exit;
                ctx->us_analyze += t_analyzed - t_enter;  ctx->us_issue += t_issued - t_analyzed;  ctx->us_wait += t_done - t_issued;
                ctx->t_exit = t_done;
                ctx->t_exit_pp = 0;
            } else {
                ctx->t_exit = 0; ctx->ms_pp += ms; ctx->graphs_pp++; ctx->tokens_pp += n_tok; if (ctx->ms_pp_min == 0 || ms < ctx->ms_pp_min) ctx->ms_pp_min = ms;
                const double t_done = wall_us();
                if (ctx->graphs_pp > 1) {                                // (the first prompt graph also re-lays weights at their first use: not a sample)
                    if (ctx->t_exit_pp > 0 && ctx->graphs_pp > 2) { ctx->pp_outside += t_enter - ctx->t_exit_pp; ctx->pp_outside_n++; }
                    ctx->pp_analyze += t_analyzed - t_enter;  ctx->pp_issue += t_issued - t_analyzed;  ctx->pp_wait += t_done - t_issued;
                }
                ctx->t_exit_pp = t_done;
            }
        }
    }
    return GGML_STATUS_SUCCESS;
}

const ggml_backend_i backend_iface = {
    /* .get_name           = */ backend_get_name,
    /* .free               = */ backend_free,
    /* .set_tensor_async   = */ backend_set_tensor_async,
    /* .get_tensor_async   = */ backend_get_tensor_async,
    /* .cpy_tensor_async   = */ backend_cpy_tensor_async,
    /* .synchronize        = */ backend_synchronize,
    /* .graph_plan_create  = */ nullptr,
    /* .graph_plan_free    = */ nullptr,
    /* .graph_plan_update  = */ nullptr,
    /* .graph_plan_compute = */ nullptr,
    /* .graph_compute      = */ backend_graph_compute,
    /* .event_record       = */ backend_event_record,
    /* .event_wait         = */ backend_event_wait,
};

ggml_guid_t backend_guid() {
    static ggml_guid guid = { 0x4d, 0x49, 0x33, 0x35, 0x35, 0x58, 0x2d, 0x67, 0x66, 0x78, 0x39, 0x35, 0x30, 0x2d, 0x71, 0x6d };
    return &guid;
}

// ----------------------------------------------------------------------------------------------- device

const char * dev_get_name(ggml_backend_dev_t dev) { return ((mi355x_device_ctx *) dev->context)->name.c_str(); }
const char * dev_get_description(ggml_backend_dev_t dev) { return ((mi355x_device_ctx *) dev->context)->description.c_str(); }
void dev_get_memory(ggml_backend_dev_t dev, size_t * free, size_t * total) {
    qmm_device_info(((mi355x_device_ctx *) dev->context)->qmm, nullptr, 0, free, total, nullptr);
}
enum ggml_backend_dev_type dev_get_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_GPU; }
void dev_get_props(ggml_backend_dev_t dev, struct ggml_backend_dev_props * props) {
    props->name = dev_get_name(dev);
    props->description = dev_get_description(dev);
    props->type = GGML_BACKEND_DEVICE_TYPE_GPU;
    dev_get_memory(dev, &props->memory_free, &props->memory_total);
    props->caps = { /* async */ true, /* host_buffer */ true, /* buffer_from_host_ptr */ false, /* events */ true };
}
ggml_backend_t dev_init_backend(ggml_backend_dev_t dev, const char *) {
    auto * d = (mi355x_device_ctx *) dev->context;
    return new ggml_backend{ backend_guid(), backend_iface, dev, new mi355x_backend_ctx{ d, d->name } };
}
ggml_backend_buffer_type_t dev_get_buffer_type(ggml_backend_dev_t dev) { return &((mi355x_device_ctx *) dev->context)->buft; }

ggml_backend_buffer_type_t dev_get_host_buffer_type(ggml_backend_dev_t) { return host_buffer_type(); }

ggml_backend_event_t dev_event_new(ggml_backend_dev_t dev) {
    qmm_event * e = qmm_event_create(((mi355x_device_ctx *) dev->context)->qmm);
    if (!e) {
        GGML_LOG_ERROR("MI355X event_new: %s\n", qmm_last_error());
        return nullptr;
    }
    return new ggml_backend_event{ dev, e };
}
void dev_event_free(ggml_backend_dev_t dev, ggml_backend_event_t event) {
    qmm_event_destroy(((mi355x_device_ctx *) dev->context)->qmm, (qmm_event *) event->context);
    delete event;
}
void dev_event_synchronize(ggml_backend_dev_t dev, ggml_backend_event_t event) {
    if (qmm_event_synchronize(((mi355x_device_ctx *) dev->context)->qmm, (qmm_event *) event->context))
        GGML_ABORT("MI355X event_synchronize: %s", qmm_last_error());
}

bool dev_supports_op(ggml_backend_dev_t, const struct ggml_tensor * op) {
    // a row-split tensor has no address of its own (split_buffer_get_base): only the quantized MUL_MAT knows how to read one, as
    // src0.  Answering "no" here is also what keeps llama.cpp from placing norm weights in the split buffer type
    // (weight_buft_supported, src/llama-model.cpp:123-242; ggml-cuda.cu:2972-2982 has the same gate).
    for (int i = 0; i < GGML_MAX_SRC; ++i) {
        const ggml_tensor * s = op->src[i];
        if (s && s->buffer && buft_is_split(s->buffer->buft) && !(op->op == GGML_OP_MUL_MAT && i == 0 && type_supported(s->type))) return false;
    }
    switch (op->op) {
        case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE:
            return true;
        case GGML_OP_MUL_MAT:    return glue_op(op) ? supports_glue(op) : supports_mul_mat(op);
        case GGML_OP_MUL_MAT_ID: return supports_mul_mat_id(op);
        default: return supports_glue(op);
    }
}
bool dev_supports_buft(ggml_backend_dev_t dev, ggml_backend_buffer_type_t buft) {
    if (buft_is_split(buft)) return true;                             // any device can be the root of a split MUL_MAT
    return buft->iface.get_name == buft_get_name && buft->context == dev->context;
}

const ggml_backend_device_i device_iface = {
    /* .get_name             = */ dev_get_name,
    /* .get_description      = */ dev_get_description,
    /* .get_memory           = */ dev_get_memory,
    /* .get_type             = */ dev_get_type,
    /* .get_props            = */ dev_get_props,
    /* .init_backend         = */ dev_init_backend,
    /* .get_buffer_type      = */ dev_get_buffer_type,
    /* .get_host_buffer_type = */ dev_get_host_buffer_type,
    /* .buffer_from_host_ptr = */ nullptr,
    /* .supports_op          = */ dev_supports_op,
    /* .supports_buft        = */ dev_supports_buft,
    /* .offload_op           = */ nullptr,
    /* .event_new            = */ dev_event_new,
    /* .event_free           = */ dev_event_free,
    /* .event_synchronize    = */ dev_event_synchronize,
};

// ----------------------------------------------------------------------------------------------- reg

const char * reg_get_name(ggml_backend_reg_t) { return GGML_MI355X_BACKEND_NAME; }
size_t reg_get_device_count(ggml_backend_reg_t) { return (size_t) g_ndev; }
ggml_backend_dev_t reg_get_device(ggml_backend_reg_t, size_t index) { return index < (size_t) g_ndev ? &g_devices[index] : nullptr; }
// tensor_split: per-device proportions as llama.cpp passes them (src/llama-model.cpp:316-346); all zero or NULL = equal
// shares.  Cached per (main_device, fractions) like ggml_backend_cuda_split_buffer_type (ggml-cuda.cu:1010-1052).
ggml_backend_buffer_type_t split_buffer_type(int main_device, const float * tensor_split) {
    static std::mutex mutex;
    std::lock_guard<std::mutex> lock(mutex);
    static std::map<std::pair<int, std::array<float, GGML_MI355X_MAX_DEVICES>>, ggml_backend_buffer_type> bufts;
    if (main_device < 0 || main_device >= g_ndev) return nullptr;
    std::array<float, GGML_MI355X_MAX_DEVICES> cum = {};
    float sum = 0.0f;
    for (int i = 0; i < g_ndev; ++i) sum += tensor_split ? tensor_split[i] : 0.0f;
    float acc = 0.0f;
    for (int i = 0; i < g_ndev; ++i) {
        cum[i] = sum > 0.0f ? acc / sum : (float) i / g_ndev;
        acc += tensor_split ? tensor_split[i] : 0.0f;
    }
    auto key = std::make_pair(main_device, cum);
    auto it = bufts.find(key);
    if (it != bufts.end()) return &it->second;
    auto * ctx = new split_buft_ctx{ main_device, cum, std::string(GGML_MI355X_BACKEND_NAME) + std::to_string(main_device) + "_Split" };
    return &bufts.emplace(key, ggml_backend_buffer_type{ split_buft_iface, &g_devices[main_device], ctx }).first->second;
}

void * reg_get_proc_address(ggml_backend_reg_t, const char * name) {
    if (strcmp(name, "ggml_backend_split_buffer_type") == 0) return (void *) split_buffer_type;    // llama.cpp -sm row
    // void (*)(int on): multi-node launches on (1) / one launch per node (0) / back to GGML_MI355X_FUSE (-1); lets one process
    // compare both schedules of the same graph (tests/cpp/test_graph_fuzz.cpp)
    if (strcmp(name, "ggml_backend_mi355x_set_fuse") == 0) return (void *) set_fuse;
    return nullptr;
}

const ggml_backend_reg_i reg_iface = { reg_get_name, reg_get_device_count, reg_get_device, reg_get_proc_address };

} // namespace

extern "C" {

ggml_backend_reg_t ggml_backend_mi355x_reg(void) {
    static ggml_backend_reg reg = { GGML_BACKEND_API_VERSION, reg_iface, nullptr };
    static std::once_flag once;                                     // the reference guards its reg init too (ggml-hexagon.cpp:5953-5955)
    std::call_once(once, [] {
        const int n_phys = qmm_device_count();
        // GGML_MI355X_VIRTUAL_DEVICES=n registers n logical devices over the physical ones (round robin): lets the row split
        // be exercised on a one-GPU box (tests/test_gpu_split_buffer.py); each logical device has its own context and stream
        const char * vd = getenv("GGML_MI355X_VIRTUAL_DEVICES");
        const int n = vd && atoi(vd) > 0 && n_phys > 0 ? atoi(vd) : n_phys;
        for (int i = 0; i < n && g_ndev < GGML_MI355X_MAX_DEVICES; ++i) {
            qmm_ctx * q = qmm_create(i % n_phys);
            if (!q) {
                GGML_LOG_WARN("MI355X: skipping HIP device %d: %s\n", i, qmm_last_error());
                continue;
            }
            mi355x_device_ctx & d = g_devs[g_ndev];
            d.ordinal = i % n_phys;
            d.qmm = q;
            d.name = std::string(GGML_MI355X_BACKEND_NAME) + std::to_string(g_ndev);
            char nm[128] = { 0 };
            int cus = 0;
            qmm_device_info(q, nm, sizeof(nm), nullptr, nullptr, &cus);
            d.description = std::string(nm) + " (gfx950, " + std::to_string(cus) + " CUs)";
            d.buft_name = d.name;
            g_devices[g_ndev] = ggml_backend_device{ device_iface, &reg, &d };
            d.buft = ggml_backend_buffer_type{ buft_iface, &g_devices[g_ndev], &d };
            ++g_ndev;
        }
        GGML_LOG_INFO("MI355X backend: %d device(s); quantized MUL_MAT / MUL_MAT_ID (Q4_0 Q8_0 Q4_K Q5_K Q6_K)%s\n", g_ndev,
                      GGML_MI355X_GLUE_OFF() ? "" : " and the glue ops of a transformer layer");
    });
    return &reg;
}

int ggml_backend_mi355x_get_device_count(void) {
    ggml_backend_mi355x_reg();
    return g_ndev;
}

const char * ggml_backend_mi355x_get_devname(size_t dev_num) {
    ggml_backend_mi355x_reg();
    return dev_num < (size_t) g_ndev ? g_devs[dev_num].name.c_str() : "unknown";
}

ggml_backend_buffer_type_t ggml_backend_mi355x_buffer_type(size_t dev_num) {
    ggml_backend_mi355x_reg();
    return dev_num < (size_t) g_ndev ? &g_devs[dev_num].buft : nullptr;
}

ggml_backend_t ggml_backend_mi355x_init(size_t dev_num) {
    ggml_backend_mi355x_reg();
    if (dev_num >= (size_t) g_ndev) {
        GGML_LOG_ERROR("%s: invalid device %zu (have %d)\n", __func__, dev_num, g_ndev);
        return nullptr;
    }
    return dev_init_backend(&g_devices[dev_num], nullptr);
}

bool ggml_backend_is_mi355x(ggml_backend_t backend) { return backend != nullptr && ggml_guid_matches(backend->guid, backend_guid()); }

static int ggml_backend_mi355x_score(void) { return qmm_device_count() > 0 ? 100 : 0; }

} // extern "C"

GGML_BACKEND_DL_IMPL(ggml_backend_mi355x_reg)
GGML_BACKEND_DL_SCORE_IMPL(ggml_backend_mi355x_score)
