// infur_rt.h -- internal interface between the host runtime's translation units (round 5: infur_capi.cpp was 2,900 lines):
//   infur_capi.cpp          context, arena, the float model (load, conv dispatch, forward), the C entry points of the stages
//   infur_quant_model.cpp   quantised models (INFURQ01): load + forward
//   infur_tuner.cpp         tile-configuration tuner (pick_cfg) and its database (infur_tune_import / _export)
//   infur_stream.cpp        streaming ring, frame batch, pinned host buffers
//   infur_multi.cpp         groups of contexts, RCCL
// Everything here lives in namespace infur and is NOT part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "infur_ctx.h"
#include "kernels.h"

namespace infur {

// records the message on the context and returns `code`
int32_t fail(infur_ctx* c, int32_t code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));

#define HIPCHK(c, expr)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return ::infur::fail((c), INFUR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                                 __FILE__, __LINE__);                                           \
    } while (0)

#define RETIF(expr)                 \
    do {                            \
        int32_t rc__ = (expr);      \
        if (rc__ != INFUR_OK) return rc__; \
    } while (0)

// ---- arena ----
int32_t ensure(infur_ctx* c, Buf& b, size_t bytes);
int32_t pool_acquire(infur_ctx* c, size_t bytes, int* slot);
void pool_release(infur_ctx* c, Tensor& t);
void pool_release_all(infur_ctx* c);
void pool_free(infur_ctx* c);
void pool_trim(infur_ctx* c);
int32_t talloc(infur_ctx* c, int h, int w, int ch, int es, Tensor* t);
constexpr uint32_t kPoolTrimAfter = 4;

// ---- arithmetic mode of a context ----
inline bool ctx_f16(const infur_ctx* c) { return c->opt.compute_dtype == INFUR_DTYPE_F16; }
// GEMM arithmetic of launch_conv_igemm: 0 f32 MFMA, 1 f16, 2 f32 tensors split into f16 pairs
// (INFUR_DTYPE_F32_SPLIT_FP8 is the split mode everywhere except inside the GEMM: conv_mode() = 3 selects its MFMA sequence,
//  its weight rows and its own tuning entries)
inline bool ctx_fp8x(const infur_ctx* c) { return c->opt.compute_dtype == INFUR_DTYPE_F32_SPLIT_FP8; }
inline int ctx_mode(const infur_ctx* c) { return ctx_fp8x(c) ? (int)INFUR_DTYPE_F32_SPLIT : (int)c->opt.compute_dtype; }
inline int conv_mode(const infur_ctx* c) { return ctx_fp8x(c) ? 3 : ctx_mode(c); }
// INFUR_DTYPE_F16_HL (= conv mode 5): three-byte tensors (f16 hi + e5m2 lo planes), conv_hl.hip
inline bool ctx_hl(const infur_ctx* c) { return c->opt.compute_dtype == INFUR_DTYPE_F16_HL; }
inline int act_es(const infur_ctx* c) { return ctx_f16(c) ? 2 : (ctx_hl(c) ? 3 : 4); }
inline const float* stem_lut(const infur_ctx* c) { return c->input_u8 ? c->d_u8_lut : c->d_pre_lut; }
inline int conv_out(int n, int k, int s, int p, int d) { return (n + 2 * p - d * (k - 1) - 1) / s + 1; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- roctx ranges + per-kernel HIP events (infur_capi.cpp) ----
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
const Roctx* roctx();
struct RoctxRange {
    const Roctx* rx;
    explicit RoctxRange(const char* name) : rx(roctx()) {
        if (rx) rx->push(name);
    }
    ~RoctxRange() {
        if (rx) rx->pop();
    }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};
struct ProfScope {
    infur_ctx* c;
    bool on;
    const Roctx* rx;
    ProfRec r;
    ProfScope(infur_ctx* c_, const std::string& name, const char* kernel, double flops, double bytes, double algo_flops = -1.0);
    ~ProfScope();
};
void prof_reset(infur_ctx* c);

// ---- lookup tables (host side, exact reference operation order) ----
void build_pre_lut(float* lut);
void build_color_lut(uint32_t* lut);

// ---- the model ----
std::vector<ConvLayer> build_graph(int depth, int ncls, bool aux);
void model_free(infur_ctx* c);
UpQuant head_quant(const infur_ctx* c, int k);
int32_t stem16_image(infur_ctx* c, const float* wt, float w_scale, int split, const void** img);
// measured tile configuration of the conv kernel for one problem shape (infur_tuner.cpp)
int32_t pick_cfg(infur_ctx* c, const ConvArgs& a, int mode, int out_f32, int* cfg);
struct EventPair {  // two timing events, released on every return path
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t create() {
        hipError_t e = hipEventCreate(&e0);
        return e != hipSuccess ? e : hipEventCreate(&e1);
    }
    ~EventPair() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};
// quantised models (infur_quant_model.cpp)
int32_t model_load_q_dev(infur_ctx* c, const void* d_blob, size_t len);
int32_t forward_q(infur_ctx* c, const uint8_t* d_bgr, int w, int h);

}  // namespace infur
