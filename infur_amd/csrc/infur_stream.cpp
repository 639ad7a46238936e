// infur_stream.cpp -- the streaming ring of the C ABI (include/infur_hip.h: infur_stream_*, infur_batch_advance, infur_host_*).
//
// Reference behaviour mirrored here: the `Proc` thread's bounded channel of frames in flight (infur/src/main.rs:27-99,105) and the
// decoder that fills a caller-owned, reused frame buffer in place (ff-video/src/decoder.rs:156-165).  A stream is a ring of `depth`
// pinned + device slots; H2D, the fused frame path and D2H of neighbouring frames overlap on three HIP streams.  Split out of
// infur_capi.cpp in round 5; the frame path itself (infur_frame_advance_dev) stays there.
#include "../../include/infur_hip.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "infur_ctx.h"

using namespace infur;

#define fail ctx_fail
#define HIPCHK(c, expr)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return ctx_fail((c), INFUR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                            __FILE__, __LINE__);                                                \
    } while (0)
#define RETIF(expr)                 \
    do {                            \
        int32_t rc__ = (expr);      \
        if (rc__ != INFUR_OK) return rc__; \
    } while (0)
static inline void enter(const infur_ctx* c) { ctx_enter(c); }


struct infur_stream {
    struct Slot {
        uint8_t* h_in = nullptr;   // pinned
        uint8_t* h_out = nullptr;  // pinned: [rgba | scaled bgr]
        void* d_in = nullptr;
        void* d_out = nullptr;  // [rgba | scaled bgr]
        size_t in_cap = 0, out_cap = 0;
        uint32_t small_in = 0, small_out = 0;  // consecutive requests below a quarter of the capacity (slot_reserve's hysteresis)
        hipEvent_t ev_h2d = nullptr, ev_comp = nullptr, ev_done = nullptr;
        uint64_t id = 0;
        uint32_t ow = 0, oh = 0;
        int32_t status = INFUR_OK;
        bool busy = false;
        // zero-copy egress: the mask goes by DMA straight into a pinned buffer of the caller (infur_batch_advance with buffers from
        // infur_host_alloc); nullptr: into h_out
        uint8_t* direct_out = nullptr;
    };
    infur_ctx* ctx = nullptr;  // owner: holds the copy streams' device, receives the error messages
    // compute lanes: frame i runs on lanes[i % n] (lanes[0] == ctx).  Frames are independent, so a second context of
    // the same device (infur_stream_add_lane) lets the kernels of consecutive frames overlap
    std::vector<infur_ctx*> lanes;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    std::vector<Slot> slots;
    uint64_t head = 0, tail = 0;  // tail = next to collect, head = next to submit
    // zero-copy ingest / egress (infur_stream_acquire / _commit, _collect_view / _release): the slot at `head` is lent to the producer
    // (acq_*: what it was sized for), the slot at `tail` is lent to the consumer (viewing)
    bool acquired = false, viewing = false;
    uint32_t acq_w = 0, acq_h = 0, acq_ow = 0, acq_oh = 0;
    uint32_t acq_factor_bits = 0;
};

namespace {
// buffers grow on demand and are given back when requests have needed less than a quarter of them for a while (a ring that lives
// as long as its context -- infur_batch_advance's -- would otherwise keep the largest frame it ever saw).  "For a while" = 8
// consecutive small requests of that slot: a batch that ALTERNATES large and small frames (4K and 480p) would otherwise free and
// reallocate pinned + device memory on every frame -- each a device-wide synchronisation, and new pointers that no cached
// graph of the fused frame path can match.
constexpr uint32_t kSlotShrinkAfter = 8;
int32_t slot_reserve(infur_ctx* c, infur_stream::Slot& sl, size_t in_bytes, size_t out_bytes) {
    sl.small_in = sl.in_cap / 4 > in_bytes ? sl.small_in + 1 : 0;
    sl.small_out = sl.out_cap / 4 > out_bytes ? sl.small_out + 1 : 0;
    if (sl.in_cap < in_bytes || sl.small_in >= kSlotShrinkAfter) {
        if (sl.h_in) HIPCHK(c, hipHostFree(sl.h_in));
        if (sl.d_in) HIPCHK(c, hipFree(sl.d_in));
        sl.h_in = nullptr; sl.d_in = nullptr; sl.in_cap = 0;
        HIPCHK(c, hipHostMalloc((void**)&sl.h_in, in_bytes, hipHostMallocDefault));
        HIPCHK(c, hipMalloc(&sl.d_in, in_bytes));
        sl.in_cap = in_bytes;
        sl.small_in = 0;
    }
    if (sl.out_cap < out_bytes || sl.small_out >= kSlotShrinkAfter) {
        if (sl.h_out) HIPCHK(c, hipHostFree(sl.h_out));
        if (sl.d_out) HIPCHK(c, hipFree(sl.d_out));
        sl.h_out = nullptr; sl.d_out = nullptr; sl.out_cap = 0;
        HIPCHK(c, hipHostMalloc((void**)&sl.h_out, out_bytes, hipHostMallocDefault));
        HIPCHK(c, hipMalloc(&sl.d_out, out_bytes));
        sl.out_cap = out_bytes;
        sl.small_out = 0;
    }
    return INFUR_OK;
}
}  // namespace

void infur::stream_orphan(infur_stream* st) {
    infur_ctx* c = st->ctx;
    if (!c) return;
    enter(c);
    for (infur_ctx* l : st->lanes)
        if (l->stream) (void)hipStreamSynchronize(l->stream);
    if (st->s_h2d) (void)hipStreamSynchronize(st->s_h2d);
    if (st->s_d2h) (void)hipStreamSynchronize(st->s_d2h);
    for (auto& sl : st->slots) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        for (hipEvent_t e : {sl.ev_h2d, sl.ev_comp, sl.ev_done})
            if (e) (void)hipEventDestroy(e);
    }
    st->slots.clear();
    if (st->s_h2d) (void)hipStreamDestroy(st->s_h2d);
    if (st->s_d2h) (void)hipStreamDestroy(st->s_d2h);
    st->s_h2d = st->s_d2h = nullptr;
    st->head = st->tail = 0;
    st->acquired = st->viewing = false;
    for (infur_ctx* l : st->lanes)  // the stream is registered with every lane's context: any of them may go first
        for (size_t i = 0; i < l->streams.size(); i++)
            if (l->streams[i] == st) {
                l->streams.erase(l->streams.begin() + (long)i);
                break;
            }
    st->lanes.clear();
    st->ctx = nullptr;
}

extern "C" {

int32_t infur_stream_create(infur_ctx* c, uint32_t depth, infur_stream** out) {
    try {
        enter(c);
        if (!c || !out || depth == 0 || depth > 64) return INFUR_E_INVALID_ARG;
        *out = nullptr;
        infur_stream* st = new infur_stream();
        st->ctx = c;
        st->lanes.push_back(c);
        c->streams.push_back(st);
        st->slots.resize(depth);
        bool ok = hipStreamCreateWithFlags(&st->s_h2d, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&st->s_d2h, hipStreamNonBlocking) == hipSuccess;
        for (auto& sl : st->slots)
            ok = ok && hipEventCreateWithFlags(&sl.ev_h2d, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&sl.ev_comp, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            infur_stream_destroy(st);
            return fail(c, INFUR_E_HIP, "could not create the streaming ring");
        }
        *out = st;
        return INFUR_OK;
    } catch (const std::bad_alloc&) {
        return fail(c, INFUR_E_CAPACITY, "out of host memory");
    } catch (const std::exception& e) {
        return fail(c, INFUR_E_INVALID_ARG, "internal error: %s", e.what());
    }
}

void infur_stream_destroy(infur_stream* st) {
    if (!st) return;
    stream_orphan(st);  // no-op when the context went first
    delete st;
}

uint32_t infur_stream_pending(const infur_stream* st) { return st ? (uint32_t)(st->head - st->tail) : 0; }

int32_t infur_stream_add_lane(infur_stream* st, infur_ctx* other) {
    if (!st || !st->ctx || !other) return INFUR_E_INVALID_ARG;
    infur_ctx* c = st->ctx;
    if (other->device != c->device) return fail(c, INFUR_E_INVALID_ARG, "a lane must be a context of the stream's device (%d), got device %d", c->device, other->device);
    for (infur_ctx* l : st->lanes)
        if (l == other) return fail(c, INFUR_E_INVALID_ARG, "that context already is a lane of this stream");
    if (st->head != st->tail) return fail(c, INFUR_E_INVALID_ARG, "add lanes while no frame is pending");
    // odd and even frames must run the SAME arithmetic: the option set infur_group_weights_broadcast checks (the fusion
    // switches are bit-identical forms and may differ; F(4x4) and F(6x6) logits differ by ~1e-6, enough to flip a tie)
    if (other->opt.compute_dtype != c->opt.compute_dtype || other->opt.winograd_tile != c->opt.winograd_tile ||
        other->opt.winograd_min_cin != c->opt.winograd_min_cin || other->opt.compute_aux != c->opt.compute_aux)
        return fail(c, INFUR_E_INVALID_ARG, "a lane must share the stream's compute_dtype / winograd_tile / winograd_min_cin / compute_aux: its frames would otherwise differ");
    if (!other->loaded) return fail(c, INFUR_E_MODEL_NOT_LOADED, "the lane's context has no model loaded (replicate it first: infur_group_weights_broadcast)");
    if (c->loaded && other->quant != c->quant)
        return fail(c, INFUR_E_INVALID_ARG, "a lane must hold the stream's model: one of the two contexts has a quantised model, the other a float one");
    st->lanes.push_back(other);
    other->streams.push_back(st);
    return INFUR_OK;
}

namespace {
inline uint32_t f32_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

// pinned host memory (hipHostMalloc / hipHostRegister): DMA can read and write it directly
bool host_is_pinned(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // (an ordinary malloc'ed pointer is "invalid value" to the runtime: not an error of ours)
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// checks + slot reservation shared by submit and acquire: the slot at `head`, sized for a w x h frame scaled by `factor`
int32_t stream_prepare(infur_stream* st, uint32_t w, uint32_t h, float factor, infur_stream::Slot** slot, uint32_t* ow_out, uint32_t* oh_out) {
    infur_ctx* c = st->ctx;
    int32_t rc = infur_scale_validate(factor);
    if (rc) return fail(c, rc, "%s", infur_status_string(rc));
    uint32_t ow = 0, oh = 0;
    rc = infur_scale_out_dims(w, h, factor, &ow, &oh);
    if (rc) return fail(c, rc, "%s", infur_status_string(rc));
    infur_ctx* lane = st->lanes[st->head % st->lanes.size()];
    if (!lane->loaded) return fail(c, INFUR_E_MODEL_NOT_LOADED, "no model loaded");
    // (a model may have been (re)loaded on one context after the lane was added: both kinds of arithmetic in one stream would
    //  alternate frame by frame)
    if (lane->quant != st->lanes[0]->quant || lane->depth != st->lanes[0]->depth)
        return fail(c, INFUR_E_INVALID_ARG, "the stream's lanes hold different models (quantised / float, or different depths): replicate one model to all of them");
    const size_t depth = st->slots.size();
    if (st->head - st->tail >= depth)
        return fail(c, INFUR_E_CAPACITY, "all %zu slots are in flight: collect before submitting more", depth);
    infur_stream::Slot& sl = st->slots[st->head % depth];
    const size_t in_bytes = (size_t)w * h * 3, rgba_bytes = (size_t)ow * oh * 4, sc_bytes = (size_t)ow * oh * 3;
    if (in_bytes == 0 || rgba_bytes == 0) return fail(c, INFUR_E_SHAPE, "couldn't transform image: %ux%u", w, h);
    RETIF(slot_reserve(c, sl, in_bytes, rgba_bytes + sc_bytes));
    *slot = &sl;
    *ow_out = ow;
    *oh_out = oh;
    return INFUR_OK;
}

// enqueues H2D -> scale / model / decode -> D2H for the slot at `head`.  src: pinned host memory holding the frame (the slot's own
// h_in, or a pinned buffer of the caller); direct_out: pinned destination of the mask instead of the slot's h_out (or nullptr)
int32_t stream_enqueue(infur_stream* st, infur_stream::Slot& sl, const uint8_t* src, uint32_t w, uint32_t h, float factor, uint32_t mode,
                       uint64_t frame_id, uint32_t ow, uint32_t oh, uint8_t* direct_out) {
    infur_ctx* c = st->ctx;
    infur_ctx* lane = st->lanes[st->head % st->lanes.size()];
    const size_t in_bytes = (size_t)w * h * 3, rgba_bytes = (size_t)ow * oh * 4, sc_bytes = (size_t)ow * oh * 3;
    sl.id = frame_id;
    sl.ow = ow;
    sl.oh = oh;
    sl.direct_out = direct_out;
    // From here on work that reads / writes this slot's buffers is in flight.  The slot is handed out again by the next
    // submit (head does not advance on failure) and slot_reserve may free its buffers, so every failing return below
    // first waits for whatever was enqueued (quiesce).
    auto quiesce = [&]() {
        (void)hipStreamSynchronize(st->s_h2d);
        (void)hipStreamSynchronize(lane->stream);
        (void)hipStreamSynchronize(st->s_d2h);
    };
#define SUBMIT_CHK(expr)                                                                                                     \
    do {                                                                                                                     \
        hipError_t e__ = (expr);                                                                                             \
        if (e__ != hipSuccess) {                                                                                             \
            quiesce();                                                                                                       \
            return fail(c, INFUR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__);        \
        }                                                                                                                    \
    } while (0)
    SUBMIT_CHK(hipMemcpyAsync(sl.d_in, src, in_bytes, hipMemcpyHostToDevice, st->s_h2d));
    SUBMIT_CHK(hipEventRecord(sl.ev_h2d, st->s_h2d));
    SUBMIT_CHK(hipStreamWaitEvent(lane->stream, sl.ev_h2d, 0));
    uint8_t* d_rgba = (uint8_t*)sl.d_out;
    uint8_t* d_sc = d_rgba + rgba_bytes;
    uint32_t a = 0, b = 0;
    sl.status = infur_frame_advance_dev(lane, sl.d_in, w, h, factor, mode, d_rgba, rgba_bytes, d_sc, &a, &b);
    if (sl.status != INFUR_OK) {
        const std::string msg = lane->err;  // (quiesce must not lose the message)
        quiesce();
        c->err = msg;
        return sl.status;
    }
    SUBMIT_CHK(hipEventRecord(sl.ev_comp, lane->stream));
    SUBMIT_CHK(hipStreamWaitEvent(st->s_d2h, sl.ev_comp, 0));
    if (direct_out)  // (the mask alone: a caller-owned destination has no room for the scaled frame)
        SUBMIT_CHK(hipMemcpyAsync(direct_out, sl.d_out, rgba_bytes, hipMemcpyDeviceToHost, st->s_d2h));
    else
        SUBMIT_CHK(hipMemcpyAsync(sl.h_out, sl.d_out, rgba_bytes + sc_bytes, hipMemcpyDeviceToHost, st->s_d2h));
    SUBMIT_CHK(hipEventRecord(sl.ev_done, st->s_d2h));
#undef SUBMIT_CHK
    sl.busy = true;
    st->head++;
    st->acquired = false;
    return INFUR_OK;
}

// submit with optional zero-copy: pinned_src -- `bgr` is pinned and stays untouched until the frame is collected (the batch calls:
// they return only when everything is done); direct_out -- pinned destination for the mask
int32_t stream_submit_impl(infur_stream* st, const uint8_t* bgr, uint32_t w, uint32_t h, float factor, uint32_t mode, uint64_t frame_id,
                           bool pinned_src, uint8_t* direct_out) {
    if (!st || !st->ctx || !bgr) return INFUR_E_INVALID_ARG;  // (a stream whose context was destroyed is dead)
    enter(st->ctx);
    if (st->acquired) return fail(st->ctx, INFUR_E_INVALID_ARG, "a slot is acquired: commit it before submitting another frame");
    infur_stream::Slot* sl = nullptr;
    uint32_t ow = 0, oh = 0;
    RETIF(stream_prepare(st, w, h, factor, &sl, &ow, &oh));
    if (!pinned_src) memcpy(sl->h_in, bgr, (size_t)w * h * 3);  // the caller's buffer is free again when submit returns
    return stream_enqueue(st, *sl, pinned_src ? bgr : sl->h_in, w, h, factor, mode, frame_id, ow, oh, direct_out);
}
}  // namespace

int32_t infur_stream_submit(infur_stream* st, const uint8_t* bgr, uint32_t w, uint32_t h, float factor, uint32_t mode,
                            uint64_t frame_id) {
    try {
        return stream_submit_impl(st, bgr, w, h, factor, mode, frame_id, false, nullptr);
    } catch (const std::bad_alloc&) {
        return fail(st ? st->ctx : nullptr, INFUR_E_CAPACITY, "out of host memory");
    } catch (const std::exception& e) {
        return fail(st ? st->ctx : nullptr, INFUR_E_INVALID_ARG, "internal error: %s", e.what());
    }
}

// ---- zero-copy ingest: the producer fills the ring's own pinned slot (ff-video/src/decoder.rs:156-165 reads into a reused BgrImage) ----
int32_t infur_stream_acquire(infur_stream* st, uint32_t w, uint32_t h, float factor, uint8_t** bgr_slot) {
    try {
        if (!st || !st->ctx || !bgr_slot) return INFUR_E_INVALID_ARG;
        enter(st->ctx);
        *bgr_slot = nullptr;
        // an earlier acquire is void from here on: if the re-size below fails (slot_reserve frees before it allocates) the slot has
        // no buffers, and a commit against the old dimensions must not find `acquired` still set
        st->acquired = false;
        infur_stream::Slot* sl = nullptr;
        uint32_t ow = 0, oh = 0;
        RETIF(stream_prepare(st, w, h, factor, &sl, &ow, &oh));  // (acquiring again re-sizes the same slot: nothing is in flight on it)
        st->acquired = true;
        st->acq_w = w;
        st->acq_h = h;
        st->acq_ow = ow;
        st->acq_oh = oh;
        st->acq_factor_bits = f32_bits(factor);
        *bgr_slot = sl->h_in;
        return INFUR_OK;
    } catch (const std::bad_alloc&) {
        return fail(st ? st->ctx : nullptr, INFUR_E_CAPACITY, "out of host memory");
    } catch (const std::exception& e) {
        return fail(st ? st->ctx : nullptr, INFUR_E_INVALID_ARG, "internal error: %s", e.what());
    }
}

int32_t infur_stream_commit(infur_stream* st, uint32_t w, uint32_t h, float factor, uint32_t mode, uint64_t frame_id) {
    try {
        if (!st || !st->ctx) return INFUR_E_INVALID_ARG;
        enter(st->ctx);
        infur_ctx* c = st->ctx;
        if (!st->acquired) return fail(c, INFUR_E_INVALID_ARG, "no slot is acquired");
        if (w != st->acq_w || h != st->acq_h || f32_bits(factor) != st->acq_factor_bits)
            return fail(c, INFUR_E_INVALID_ARG, "commit of a %ux%u frame (factor %g) into a slot acquired for %ux%u", w, h, (double)factor, st->acq_w, st->acq_h);
        infur_ctx* lane = st->lanes[st->head % st->lanes.size()];
        if (!lane->loaded) return fail(c, INFUR_E_MODEL_NOT_LOADED, "no model loaded");  // (unloaded between acquire and commit)
        // (... or another model loaded on one lane between acquire and commit: the same rule stream_prepare enforces)
        if (lane->quant != st->lanes[0]->quant || lane->depth != st->lanes[0]->depth)
            return fail(c, INFUR_E_INVALID_ARG, "the stream's lanes hold different models (quantised / float, or different depths): replicate one model to all of them");
        infur_stream::Slot& sl = st->slots[st->head % st->slots.size()];
        if (!sl.h_in || !sl.d_in) return fail(c, INFUR_E_INVALID_ARG, "the acquired slot has no buffers");
        return stream_enqueue(st, sl, sl.h_in, w, h, factor, mode, frame_id, st->acq_ow, st->acq_oh, nullptr);
    } catch (const std::bad_alloc&) {
        return fail(st ? st->ctx : nullptr, INFUR_E_CAPACITY, "out of host memory");
    } catch (const std::exception& e) {
        return fail(st ? st->ctx : nullptr, INFUR_E_INVALID_ARG, "internal error: %s", e.what());
    }
}

// the producer found no frame to put into the slot it acquired (end of input, read error): give it back uncommitted
int32_t infur_stream_abandon(infur_stream* st) {
    if (!st || !st->ctx) return INFUR_E_INVALID_ARG;
    enter(st->ctx);
    st->acquired = false;  // (idempotent: abandoning with nothing acquired is not an error -- EOF paths call it unconditionally)
    return INFUR_OK;
}

int32_t infur_stream_next_dims(const infur_stream* st, uint64_t* frame_id, uint32_t* ow, uint32_t* oh) {
    if (!st || !st->ctx || st->head == st->tail) return INFUR_E_INVALID_ARG;
    const infur_stream::Slot& sl = st->slots[st->tail % st->slots.size()];
    if (frame_id) *frame_id = sl.id;
    if (ow) *ow = sl.ow;
    if (oh) *oh = sl.oh;
    return INFUR_OK;
}

int32_t infur_stream_collect(infur_stream* st, uint8_t* rgba, size_t cap, uint8_t* scaled, uint64_t* frame_id,
                             uint32_t* ow, uint32_t* oh) {
    if (!st || !st->ctx) return INFUR_E_INVALID_ARG;
    enter(st->ctx);
    infur_ctx* c = st->ctx;
    if (st->head == st->tail) return fail(c, INFUR_E_INVALID_ARG, "no frame is pending");
    infur_stream::Slot& sl = st->slots[st->tail % st->slots.size()];
    const size_t rgba_bytes = (size_t)sl.ow * sl.oh * 4, sc_bytes = (size_t)sl.ow * sl.oh * 3;
    if (rgba && cap < rgba_bytes) return fail(c, INFUR_E_CAPACITY, "mask needs %zu bytes, buffer has %zu", rgba_bytes, cap);
    if (scaled && sl.direct_out) return fail(c, INFUR_E_INVALID_ARG, "this frame's mask went straight to a caller-owned buffer: the scaled frame was not kept");
    HIPCHK(c, hipEventSynchronize(sl.ev_done));
    if (rgba && rgba != sl.direct_out) memcpy(rgba, sl.direct_out ? sl.direct_out : sl.h_out, rgba_bytes);
    if (scaled) memcpy(scaled, sl.h_out + rgba_bytes, sc_bytes);
    if (frame_id) *frame_id = sl.id;
    if (ow) *ow = sl.ow;
    if (oh) *oh = sl.oh;
    sl.busy = false;
    sl.direct_out = nullptr;
    st->viewing = false;
    st->tail++;
    return INFUR_OK;
}

// ---- zero-copy egress: the oldest finished frame's mask (and scaled frame) in place, in the ring's pinned slot ----
int32_t infur_stream_collect_view(infur_stream* st, const uint8_t** rgba, const uint8_t** scaled, uint64_t* frame_id, uint32_t* ow, uint32_t* oh) {
    if (!st || !st->ctx) return INFUR_E_INVALID_ARG;
    enter(st->ctx);
    infur_ctx* c = st->ctx;
    if (st->head == st->tail) return fail(c, INFUR_E_INVALID_ARG, "no frame is pending");
    infur_stream::Slot& sl = st->slots[st->tail % st->slots.size()];
    HIPCHK(c, hipEventSynchronize(sl.ev_done));
    const size_t rgba_bytes = (size_t)sl.ow * sl.oh * 4;
    if (rgba) *rgba = sl.direct_out ? sl.direct_out : sl.h_out;
    if (scaled) *scaled = sl.direct_out ? nullptr : sl.h_out + rgba_bytes;
    if (frame_id) *frame_id = sl.id;
    if (ow) *ow = sl.ow;
    if (oh) *oh = sl.oh;
    st->viewing = true;  // the slot stays the consumer's until infur_stream_release (or a copying collect of the same frame)
    return INFUR_OK;
}

int32_t infur_stream_release(infur_stream* st) {
    if (!st || !st->ctx) return INFUR_E_INVALID_ARG;
    infur_ctx* c = st->ctx;
    if (!st->viewing || st->head == st->tail) return fail(c, INFUR_E_INVALID_ARG, "no frame is being viewed");
    infur_stream::Slot& sl = st->slots[st->tail % st->slots.size()];
    sl.busy = false;
    sl.direct_out = nullptr;
    st->viewing = false;
    st->tail++;
    return INFUR_OK;
}

// ---- pinned host memory for the caller's own frame / mask buffers: the batch calls move such buffers by DMA, without the
//      pageable -> pinned staging copy (and back) they otherwise make ----
int32_t infur_host_alloc(size_t bytes, void** p) {
    if (!p || bytes == 0) return INFUR_E_INVALID_ARG;
    *p = nullptr;
    // portable: every device of the process may DMA it (a group's workers each move their own slice of one batch)
    return hipHostMalloc(p, bytes, hipHostMallocPortable) == hipSuccess ? INFUR_OK : INFUR_E_CAPACITY;
}

int32_t infur_host_free(void* p) {
    if (!p) return INFUR_OK;
    return hipHostFree(p) == hipSuccess ? INFUR_OK : INFUR_E_INVALID_ARG;
}

uint32_t infur_host_is_pinned(const void* p) { return host_is_pinned(p) ? 1u : 0u; }

// ---- frame batch ----
int32_t infur_batch_advance(infur_ctx* c, const uint8_t* const* frames, const uint32_t* ws, const uint32_t* hs, uint32_t n,
                            float factor, uint32_t mode, uint8_t* const* rgba, const size_t* caps, uint32_t* ows,
                            uint32_t* ohs) {
    try {
        enter(c);
        if (!c || (n && (!frames || !ws || !hs || !rgba || !caps))) return INFUR_E_INVALID_ARG;
        if (n == 0) return INFUR_OK;
        // The depth-3 ring lives as long as the context (6 pinned + device buffer pairs, 2 streams, 9 events: building it
        // per call is a visible fixed cost when a batch is 8 frames per GPU -- BASELINE configs[3] at N = 8).  It is created
        // on the first batch, shrinks with the frames (slot_reserve) and goes with infur_ctx_destroy.
        if (!c->batch_ring) RETIF(infur_stream_create(c, 3, &c->batch_ring));
        infur_stream* st = c->batch_ring;
        int32_t rc = INFUR_OK;
        uint32_t done = 0;
        auto collect_one = [&]() -> int32_t {
            uint64_t id = 0;
            uint32_t ow = 0, oh = 0;
            int32_t r = infur_stream_next_dims(st, &id, &ow, &oh);
            if (r != INFUR_OK) return r;
            r = infur_stream_collect(st, rgba[id], caps[id], nullptr, &id, &ow, &oh);
            if (r == INFUR_OK) {
                if (ows) ows[id] = ow;
                if (ohs) ohs[id] = oh;
                done++;
            }
            return r;
        };
        for (uint32_t i = 0; i < n && rc == INFUR_OK; i++) {
            if (infur_stream_pending(st) >= 3) rc = collect_one();
            if (rc == INFUR_OK) {
                // caller-owned PINNED buffers (infur_host_alloc) are moved by DMA directly -- this call returns only when every frame
                // is done, so they are not touched behind the caller's back; pageable ones go through the ring's pinned slots
                const bool pin_in = host_is_pinned(frames[i]);
                uint32_t eow = 0, eoh = 0;
                const bool dims_ok = infur_scale_out_dims(ws[i], hs[i], factor, &eow, &eoh) == INFUR_OK;
                uint8_t* direct = (dims_ok && caps[i] >= (size_t)eow * eoh * 4 && host_is_pinned(rgba[i])) ? rgba[i] : nullptr;
                rc = stream_submit_impl(st, frames[i], ws[i], hs[i], factor, mode, i, pin_in, direct);
            }
        }
        while (rc == INFUR_OK && infur_stream_pending(st) > 0) rc = collect_one();
        if (rc != INFUR_OK) {  // frames may still be in flight into the caller's view of the ring: drop it, the next call builds a new one
            const std::string keep = c->err;  // destroy() synchronises and must not lose the message
            infur_stream_destroy(st);
            c->batch_ring = nullptr;
            c->err = keep;
        }
        return rc;
    } catch (const std::bad_alloc&) {
        return fail(c, INFUR_E_CAPACITY, "out of host memory");
    } catch (const std::exception& e) {
        return fail(c, INFUR_E_INVALID_ARG, "internal error: %s", e.what());
    }
}

}  // extern "C"
