// infur_multi.cpp -- several GPUs from one host process: groups of contexts, RCCL weight broadcast over xGMI,
// frame-batch sharding (include/infur_hip.h, "several GPUs from ONE host process").
//
// Why it exists: the reference keeps every processor on ONE thread of ONE process (infur/src/main.rs:38-40,
// 110-112), so the Rust host that replaces `Model` cannot be a torch.distributed job.  A group gives that host
// N GPUs behind plain C calls:
//   * one persistent worker thread per context (a context is single-threaded, like `&mut self`);
//   * frames are independent (infur/src/app.rs:107-153), so a batch is split into contiguous slices, one per
//     context, with NO data-path collective;
//   * the only exchange is the replication of the weights at load: ONE ncclBroadcast of the already repacked
//     weight arena (the receivers neither re-read the file nor repack).  xGMI is point-to-point -- every GPU has
//     its own link to the root -- so one broadcast of 141 MB is link-bound at about 1 ms.
#include <dlfcn.h>
#include <pthread.h>
#include <rccl/rccl.h>  // types and prototypes only: the library itself is resolved at run time (struct Rccl below)
#include <sched.h>

#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>

#include "infur_ctx.h"

using namespace infur;

namespace {

// ---- RCCL, resolved lazily ----
// libinfur_hip.so does not link librccl: the single-GPU Processor path (Scale / Model / ColorCode, one context) must load
// on a host without RCCL, or with a copy whose ABI differs from the one PyTorch bundles under the same soname.  The
// first group that needs a communicator (>= 2 devices, or INFUR_FORCE_RCCL=1) dlopens it -- "librccl.so.1" binds to a copy
// the process already mapped (torch's), else the system one -- and an unavailable library is an INFUR_E_RCCL of that
// call, never a load failure of the whole library.  INFUR_RCCL_LIB overrides the name (tests point it at nothing).
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;  // why it is unavailable
    bool ok() const { return handle != nullptr; }
};

const Rccl& rccl() {
    static std::mutex mu;
    static Rccl r;
    static std::string tried;
    std::lock_guard<std::mutex> lk(mu);
    const char* over = getenv("INFUR_RCCL_LIB");
    const std::string want = over && over[0] ? over : "";
    if (r.handle || (!r.why.empty() && tried == want)) return r;  // (a failed attempt is retried only under a different name)
    tried = want;
    r = Rccl();
    const char* names[] = {want.empty() ? "librccl.so.1" : want.c_str(), want.empty() ? "librccl.so" : nullptr};
    void* h = nullptr;
    for (const char* n : names) {
        if (!n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        const char* de = dlerror();  // (dlerror clears its message: read it once)
        r.why = de ? de : (std::string("dlopen(") + n + ") failed");
    }
    if (!h) {
        if (r.why.empty()) r.why = "librccl.so.1 not found";
        return r;
    }
    auto sym = [&](const char* n) -> void* {
        void* p = dlsym(h, n);
        if (!p && r.why.empty()) r.why = std::string("RCCL library lacks ") + n;
        return p;
    };
    r.why.clear();
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!r.why.empty()) {
        dlclose(h);
        return r;
    }
    r.handle = h;
    return r;
}

// test hook (tests/test_gpu_multi.py): INFUR_RCCL_INJECT_FAIL=broadcast makes the next collective report
// ncclInternalError without touching the communicator -- the error path (arenas released, every context keeps its old
// model, INFUR_E_RCCL with the message) cannot be reached otherwise on a healthy box.  Read on every call.
bool inject_fail(const char* what) {
    const char* e = getenv("INFUR_RCCL_INJECT_FAIL");
    return e && strcmp(e, what) == 0;
}

// ---- worker placement ----
// A group worker moves its slice of the batch through pageable -> pinned memcpys (infur_stream_submit / collect): 1.45
// GB/s per 100 frames/s of 1080p, eight of them on a two-socket host.  Each worker is therefore pinned to the CPUs of the
// NUMA node its GPU hangs off (PCI bus id -> /sys/bus/pci/devices/<id>/numa_node -> /sys/devices/system/node/nodeN/
// cpulist), so that its staging copies and its pinned buffers (first touch) stay on that socket.  INFUR_NO_NUMA_PIN=1
// leaves the threads where the scheduler puts them; a host without the sysfs files (containers) does the same silently.
int numa_node_of_device(int device) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) return -1;
    for (char* p = bus; *p; p++)
        if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');  // sysfs spells the id in lower case
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

// "0-15,128-143" -> cpu set; false when the list is missing or empty
bool cpus_of_node(int node, cpu_set_t* set) {
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char buf[4096] = {0};
    const bool got = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!got) return false;
    CPU_ZERO(set);
    int n = 0;
    for (const char* p = buf; *p;) {
        char* end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            if (end == p + 1) break;
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0) {
                CPU_SET((int)c, set);
                n++;
            }
        while (*p == ',' || *p == ' ' || *p == '\n') p++;
    }
    return n > 0;
}

// one host thread bound to one context: runs the jobs posted to it, one at a time
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int32_t()> job;
    bool has_job = false, done = true, quit = false;
    int32_t rc = INFUR_OK;
    int numa = -1;  // node the thread is pinned to, -1 = not pinned

    void start(int device) {
        const char* no = getenv("INFUR_NO_NUMA_PIN");
        int node = -1;
        cpu_set_t set;
        if (!(no && no[0] && no[0] != '0')) {
            node = numa_node_of_device(device);
            if (node >= 0 && !cpus_of_node(node, &set)) node = -1;
        }
        th = std::thread([this, device] {
            (void)hipSetDevice(device);
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [this] { return has_job || quit; });
                if (quit) return;
                std::function<int32_t()> j;
                j.swap(job);
                has_job = false;
                lk.unlock();
                int32_t r;
                try {
                    r = j();
                } catch (...) {  // nothing may unwind into the C caller
                    r = INFUR_E_INVALID_ARG;
                }
                lk.lock();
                rc = r;
                done = true;
                cv.notify_all();
            }
        });
        if (node >= 0 && pthread_setaffinity_np(th.native_handle(), sizeof set, &set) == 0) numa = node;
    }
    void post(std::function<int32_t()> j) {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(j);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    int32_t wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return done; });
        return rc;
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

}  // namespace

struct infur_group {
    std::vector<infur_ctx*> ctxs;
    std::vector<Worker*> workers;
    // RCCL: one rank per distinct device, rank r drives device devs[r]
    std::vector<int> devs;
    std::vector<ncclComm_t> comms;
    std::string err;
};

namespace {

int32_t gfail(infur_group* g, int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) {
        g->err = buf;
        if (!g->ctxs.empty()) g->ctxs[0]->err = buf;  // the one-shot forms report through ctxs[0]
    }
    return code;
}

#define GHIPCHK(g, expr)                                                                               \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return gfail((g), INFUR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

bool force_rccl() {
    const char* e = getenv("INFUR_FORCE_RCCL");
    return e && e[0] && e[0] != '0';
}

int rank_of_device(const infur_group* g, int dev) {
    for (size_t r = 0; r < g->devs.size(); r++)
        if (g->devs[r] == dev) return (int)r;
    return -1;
}

// [lo, hi) of n items owned by slice r of world (sizes differ by at most one) -- the same rule as
// infur_amd/dist.py shard_range
void shard_range(uint32_t n, uint32_t r, uint32_t world, uint32_t* lo, uint32_t* hi) {
    const uint32_t q = n / world, rem = n % world;
    *lo = r * q + (r < rem ? r : rem);
    *hi = *lo + q + (r < rem ? 1 : 0);
}

// the parts of the model state that live on the host: copied from the root, device pointers re-based
void adopt_model(infur_ctx* dst, const infur_ctx* root, void* d_weights) {
    ctx_model_free(dst);
    const uint8_t* rb = (const uint8_t*)root->d_weights;
    uint8_t* nb = (uint8_t*)d_weights;
    auto rebase = [&](const void* p) -> void* { return p ? nb + ((const uint8_t*)p - rb) : nullptr; };
    dst->convs = root->convs;
    for (ConvLayer& L : dst->convs) L.map_device_pointers(rebase);  // every pointer field, incl. the pixel-pair copies
    dst->d_weights = d_weights;
    dst->weight_bytes = root->weight_bytes;
    dst->depth = root->depth;
    dst->num_classes = root->num_classes;
    dst->has_aux = root->has_aux;
    dst->input_u8 = root->input_u8;
    dst->quant = root->quant;
    dst->qadds = root->qadds;
    dst->d_qlut = (uint8_t*)rebase(root->d_qlut);
    dst->d_qstem_w = (float*)rebase(root->d_qstem_w);
    dst->d_qstem_lut = (float*)rebase(root->d_qstem_lut);
    dst->d_qstem_bias = (int32_t*)rebase(root->d_qstem_bias);
    dst->q_resize_u8 = root->q_resize_u8;
    for (int k = 0; k < 2; k++) {
        dst->q_head_zp[k] = root->q_head_zp[k];
        dst->q_head_scale[k] = root->q_head_scale[k];
    }
    dst->info = root->info;
    dst->info.n_outputs = 1 + ((root->has_aux && dst->opt.compute_aux) ? 1 : 0);
    if (dst->info.n_outputs < 2) dst->info.output_names[1][0] = 0;
    // (a root created with compute_aux = 0 reports one output and no second name; a context that does evaluate the head names it)
    else if (!dst->info.output_names[1][0]) snprintf(dst->info.output_names[1], sizeof dst->info.output_names[1], "aux");
    dst->loaded = true;
}

}  // namespace

extern "C" {

int32_t infur_group_create(infur_ctx* const* ctxs, uint32_t n_ctx, infur_group** out) {
    if (!out) return INFUR_E_INVALID_ARG;
    *out = nullptr;
    if (!ctxs || n_ctx == 0 || n_ctx > 1024) return INFUR_E_INVALID_ARG;
    for (uint32_t i = 0; i < n_ctx; i++) {
        if (!ctxs[i]) return INFUR_E_INVALID_ARG;
        for (uint32_t j = 0; j < i; j++)
            if (ctxs[j] == ctxs[i]) return ctx_fail(ctxs[0], INFUR_E_INVALID_ARG, "context %u appears twice in the group", i);
    }
    infur_group* g = new (std::nothrow) infur_group();
    if (!g) return INFUR_E_INVALID_ARG;
    try {
        g->ctxs.assign(ctxs, ctxs + n_ctx);
        for (infur_ctx* c : g->ctxs)
            if (rank_of_device(g, c->device) < 0) g->devs.push_back(c->device);
        if (g->devs.size() >= 2 || force_rccl()) {
            const Rccl& R = rccl();
            if (!R.ok()) {
                const int32_t rc = ctx_fail(ctxs[0], INFUR_E_RCCL, "a group over %zu devices needs RCCL, which is not available: %s",
                                            g->devs.size(), R.why.c_str());
                delete g;
                return rc;
            }
            g->comms.resize(g->devs.size());
            int prev = -1;
            (void)hipGetDevice(&prev);
            const ncclResult_t r = inject_fail("init") ? ncclInternalError : R.CommInitAll(g->comms.data(), (int)g->devs.size(), g->devs.data());
            if (prev >= 0) (void)hipSetDevice(prev);  // (ncclCommInitAll visits every device)
            if (r != ncclSuccess) {
                const int32_t rc = ctx_fail(ctxs[0], INFUR_E_RCCL, "ncclCommInitAll over %zu devices failed: %s", g->devs.size(),
                                            R.GetErrorString(r));
                g->comms.clear();
                delete g;
                return rc;
            }
        }
        for (infur_ctx* c : g->ctxs) {
            Worker* w = new Worker();
            g->workers.push_back(w);
            w->start(c->device);
        }
    } catch (...) {
        infur_group_destroy(g);
        return ctx_fail(ctxs[0], INFUR_E_INVALID_ARG, "could not create the group (out of host memory or threads)");
    }
    *out = g;
    return INFUR_OK;
}

void infur_group_destroy(infur_group* g) {
    if (!g) return;
    for (Worker* w : g->workers) {
        w->stop();
        delete w;
    }
    if (!g->comms.empty()) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        for (size_t r = 0; r < g->comms.size(); r++) {
            (void)hipSetDevice(g->devs[r]);
            (void)rccl().CommDestroy(g->comms[r]);
        }
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    delete g;
}

const char* infur_group_last_error(const infur_group* g) { return g ? g->err.c_str() : "null group"; }
uint32_t infur_group_size(const infur_group* g) { return g ? (uint32_t)g->ctxs.size() : 0; }
uint32_t infur_group_uses_rccl(const infur_group* g) { return g && !g->comms.empty() ? 1u : 0u; }
int32_t infur_group_worker_numa_node(const infur_group* g, uint32_t i) { return g && i < g->workers.size() ? g->workers[i]->numa : -1; }

static int32_t group_weights_broadcast_body(infur_group* g, uint32_t root);

// The body visits every context's device (hipMalloc and launches follow the calling thread's current device): the caller's
// own current device -- a torch or HIP host has one -- is put back whatever happens inside.
static int32_t group_weights_broadcast_impl(infur_group* g, uint32_t root) {
    int prev = -1;
    (void)hipGetDevice(&prev);
    const int32_t rc = group_weights_broadcast_body(g, root);
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}

static int32_t group_weights_broadcast_body(infur_group* g, uint32_t root) {
    if (!g || root >= g->ctxs.size()) return INFUR_E_INVALID_ARG;
    infur_ctx* rc = g->ctxs[root];
    if (!rc->loaded || !rc->d_weights) return gfail(g, INFUR_E_MODEL_NOT_LOADED, "root context %u has no model to broadcast", root);
    const size_t bytes = rc->weight_bytes;
    for (infur_ctx* c : g->ctxs)
        if (c->opt.compute_dtype != rc->opt.compute_dtype || c->opt.winograd_tile != rc->opt.winograd_tile ||
            c->opt.winograd_min_cin != rc->opt.winograd_min_cin)
            return gfail(g, INFUR_E_INVALID_ARG, "contexts of a group must share compute_dtype / winograd options: the weight arena layout depends on them");

    // a receive arena per non-root context; a context's old model is replaced only once the copy has landed
    const size_t n = g->ctxs.size();
    std::vector<void*> arena(n, nullptr);
    auto release = [&]() {
        for (size_t i = 0; i < n; i++)
            if (arena[i] && i != root) {
                ctx_enter(g->ctxs[i]);
                (void)hipFree(arena[i]);
            }
    };
    arena[root] = rc->d_weights;
    for (size_t i = 0; i < n; i++) {
        if (i == root) continue;
        ctx_enter(g->ctxs[i]);
        const hipError_t e = hipMalloc(&arena[i], bytes);
        if (e != hipSuccess) {
            arena[i] = nullptr;
            release();
            return gfail(g, INFUR_E_HIP, "hipMalloc of %zu weight bytes on device %d failed: %s", bytes, g->ctxs[i]->device, hipGetErrorString(e));
        }
    }
    // leader of a device = the context whose arena takes part in the collective: the root on its own device,
    // elsewhere the first context on that device
    std::vector<int> leader(g->devs.size(), -1);
    leader[rank_of_device(g, rc->device)] = (int)root;
    for (size_t i = 0; i < n; i++) {
        const int r = rank_of_device(g, g->ctxs[i]->device);
        if (leader[r] < 0) leader[r] = (int)i;
    }
    ctx_enter(rc);
    hipError_t he = hipStreamSynchronize(rc->stream);  // the root's load has finished
    int32_t status = he == hipSuccess ? INFUR_OK : gfail(g, INFUR_E_HIP, "root stream: %s", hipGetErrorString(he));

    const int root_rank = rank_of_device(g, rc->device);
    if (status == INFUR_OK && !g->comms.empty() && g->devs.size() >= 2) {
        // ---- the collective: one broadcast over all devices ----
        // RCCL's rules for ONE thread driving SEVERAL devices (single-process multi-device): the per-rank calls of one
        // collective must sit inside one ncclGroupStart / ncclGroupEnd -- outside a group the first rank's call would block
        // waiting for peers this same thread has not called yet; every call names its own rank's communicator, runs with
        // that rank's device current (ctx_enter) and is enqueued on a stream OF that device (the leader context's); the send
        // buffer argument matters on the root only, where send == recv == the loaded arena (in place).  ncclGroupEnd
        // launches them all; completion is per stream, waited for below.
        const Rccl& R = rccl();
        ncclResult_t r = inject_fail("broadcast") ? ncclInternalError : R.GroupStart();
        const bool started = r == ncclSuccess;
        for (size_t k = 0; k < g->devs.size() && r == ncclSuccess; k++) {
            infur_ctx* c = g->ctxs[leader[k]];
            ctx_enter(c);
            r = R.Broadcast(rc->d_weights, arena[leader[k]], bytes, ncclUint8, root_rank, g->comms[k], c->stream);
        }
        if (started) {
            const ncclResult_t r2 = R.GroupEnd();
            if (r == ncclSuccess) r = r2;
        }
        if (r != ncclSuccess) status = gfail(g, INFUR_E_RCCL, "ncclBroadcast of %zu weight bytes failed: %s", bytes, R.GetErrorString(r));
        for (size_t k = 0; k < g->devs.size() && status == INFUR_OK; k++) {
            infur_ctx* c = g->ctxs[leader[k]];
            ctx_enter(c);
            he = hipStreamSynchronize(c->stream);
            if (he != hipSuccess) status = gfail(g, INFUR_E_HIP, "broadcast on device %d: %s", c->device, hipGetErrorString(he));
        }
    }
    // ---- contexts that share a device with a served one: device-to-device copy (through the one-rank
    //      communicator when INFUR_FORCE_RCCL asks for the RCCL code path on a single GPU) ----
    for (size_t i = 0; i < n && status == INFUR_OK; i++) {
        const int k = rank_of_device(g, g->ctxs[i]->device);
        if ((int)i == leader[k]) continue;
        infur_ctx* c = g->ctxs[i];
        ctx_enter(c);
        if (!g->comms.empty() && g->devs.size() == 1) {
            const Rccl& R = rccl();
            const ncclResult_t r = inject_fail("broadcast") ? ncclInternalError
                                                            : R.Broadcast(arena[leader[k]], arena[i], bytes, ncclUint8, 0, g->comms[0], c->stream);
            if (r != ncclSuccess) status = gfail(g, INFUR_E_RCCL, "ncclBroadcast (one rank) failed: %s", R.GetErrorString(r));
        } else {
            he = hipMemcpyAsync(arena[i], arena[leader[k]], bytes, hipMemcpyDeviceToDevice, c->stream);
            if (he != hipSuccess) status = gfail(g, INFUR_E_HIP, "device-to-device weight copy: %s", hipGetErrorString(he));
        }
        if (status == INFUR_OK) {
            he = hipStreamSynchronize(c->stream);
            if (he != hipSuccess) status = gfail(g, INFUR_E_HIP, "weight copy on device %d: %s", c->device, hipGetErrorString(he));
        }
    }
    if (status != INFUR_OK) {
        release();
        return status;
    }
    for (size_t i = 0; i < n; i++) {
        if (i == root) continue;
        ctx_enter(g->ctxs[i]);
        (void)hipStreamSynchronize(g->ctxs[i]->stream);  // nothing of the old model is in flight
        adopt_model(g->ctxs[i], rc, arena[i]);
    }
    return INFUR_OK;
}

static int32_t group_batch_advance_impl(infur_group* g, const uint8_t* const* frames, const uint32_t* ws, const uint32_t* hs,
                                       uint32_t n, float factor, uint32_t mode, uint8_t* const* rgba, const size_t* caps,
                                       uint32_t* ows, uint32_t* ohs) {
    if (!g || (n && (!frames || !ws || !hs || !rgba || !caps))) return INFUR_E_INVALID_ARG;
    if (n == 0) return INFUR_OK;
    const uint32_t world = (uint32_t)g->ctxs.size();
    std::vector<char> busy(world, 0);
    for (uint32_t r = 0; r < world; r++) {
        uint32_t lo, hi;
        shard_range(n, r, world, &lo, &hi);
        if (lo == hi) continue;
        infur_ctx* c = g->ctxs[r];
        busy[r] = 1;
        g->workers[r]->post([=]() {
            return infur_batch_advance(c, frames + lo, ws + lo, hs + lo, hi - lo, factor, mode, rgba + lo, caps + lo,
                                       ows ? ows + lo : nullptr, ohs ? ohs + lo : nullptr);
        });
    }
    int32_t status = INFUR_OK;
    for (uint32_t r = 0; r < world; r++) {
        if (!busy[r]) continue;
        const int32_t rc = g->workers[r]->wait();  // always wait for every slice: the caller's buffers are in use
        if (rc != INFUR_OK && status == INFUR_OK) {
            uint32_t lo, hi;
            shard_range(n, r, world, &lo, &hi);
            status = gfail(g, rc, "context %u (device %d, frames %u..%u): %s", r, g->ctxs[r]->device, lo, hi - 1,
                           std::string(infur_last_error(g->ctxs[r])).c_str());
        }
    }
    return status;
}

// nothing may unwind through the C boundary (std::vector / std::function allocate)
int32_t infur_group_weights_broadcast(infur_group* g, uint32_t root) {
    try {
        return group_weights_broadcast_impl(g, root);
    } catch (...) {
        return gfail(g, INFUR_E_INVALID_ARG, "infur_group_weights_broadcast: out of host memory");
    }
}

int32_t infur_group_batch_advance(infur_group* g, const uint8_t* const* frames, const uint32_t* ws, const uint32_t* hs,
                                  uint32_t n, float factor, uint32_t mode, uint8_t* const* rgba, const size_t* caps,
                                  uint32_t* ows, uint32_t* ohs) {
    try {
        return group_batch_advance_impl(g, frames, ws, hs, n, factor, mode, rgba, caps, ows, ohs);
    } catch (...) {
        // a slice may already be running on a worker: wait for all of them before the caller's buffers go away
        if (g)
            for (Worker* w : g->workers) (void)w->wait();
        return gfail(g, INFUR_E_INVALID_ARG, "infur_group_batch_advance: out of host memory");
    }
}

int32_t infur_weights_broadcast(infur_ctx* const* ctxs, uint32_t n_ctx) {
    infur_group* g = nullptr;
    int32_t rc = infur_group_create(ctxs, n_ctx, &g);
    if (rc != INFUR_OK) return rc;
    rc = infur_group_weights_broadcast(g, 0);
    infur_group_destroy(g);
    return rc;
}

int32_t infur_batch_advance_multi(infur_ctx* const* ctxs, uint32_t n_ctx, const uint8_t* const* frames, const uint32_t* ws,
                                  const uint32_t* hs, uint32_t n, float factor, uint32_t mode, uint8_t* const* rgba,
                                  const size_t* caps, uint32_t* ows, uint32_t* ohs) {
    infur_group* g = nullptr;
    int32_t rc = infur_group_create(ctxs, n_ctx, &g);
    if (rc != INFUR_OK) return rc;
    rc = infur_group_batch_advance(g, frames, ws, hs, n, factor, mode, rgba, caps, ows, ohs);
    infur_group_destroy(g);
    return rc;
}

}  // extern "C"
