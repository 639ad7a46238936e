// infur_ctx.h -- the context object behind the C ABI, shared by the host runtime files (infur_capi.cpp: single
// context entry points; infur_multi.cpp: groups of contexts, RCCL).  Internal: not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <cstdint>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/infur_hip.h"

struct infur_stream;

namespace infur {

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
    bool used = false;
    uint64_t last_use = 0;  // forward() number that last acquired it (pool trimming)
};

struct Tensor {  // NHWC activation (f32 or f16) living in the pool
    void* p = nullptr;
    int h = 0, w = 0, c = 0;
    int slot = -1;
    int es = 4;  // element size: 4 = f32, 2 = f16, 1 = u8 (quantised), 3 = three-byte format: f16 hi plane at p, e5m2 lo plane at lo
    void* lo = nullptr;
    size_t elems() const { return (size_t)h * w * c; }
    // (es == 3: the lo plane starts at the next 256-byte boundary behind the hi plane -- kernels.h: hl_lo_offset)
    size_t bytes() const { return es == 3 ? ((elems() * 2 + 255) & ~(size_t)255) + ((elems() + 255) & ~(size_t)255) : elems() * (size_t)es; }
};

struct ConvLayer {
    std::string name;
    int cout = 0, cin = 0, k = 0, stride = 1, pad = 0, dil = 1;
    bool relu = false;
    char role = 0;        // s stem, 1 2 3 block convs, d downsample, h head3x3, c classifier
    void* d_w = nullptr;   // repacked weights (context dtype; the stem's stay f32)
    float* d_b = nullptr;  // bias, always f32
    float* d_u = nullptr;  // Winograd-domain weights U[(mt+2)^2][cout][cin] (f32 stride-1 3x3 convs only)
    // INFUR_DTYPE_F32_SPLIT: d_w / d_u hold f16 (hi, lo) pairs of w * w_scale / u * u_scale (powers of two)
    float w_scale = 1.0f, u_scale = 1.0f;
    float* d_uacc = nullptr;  // split modes: per Winograd plane, 1 / (activation scale * that plane's weight scale)
    // INFUR_DTYPE_F16_HL: d_w / d_u hold the f16 hi planes of w * w_scale / u * (plane scale), d_wl / d_ul the e5m2 lo planes
    void* d_wl = nullptr;
    void* d_ul = nullptr;
    void* d_wcatl = nullptr;  // lo plane of d_wcat
    // conv3 of a stage's first block: its weights and the downsample branch's side by side ([cout][cin + ds.cin],
    // context dtype), the two biases summed -- the two-source GEMM of run_conv_dual
    void* d_wcat = nullptr;
    float* d_bcat = nullptr;
    float wcat_scale = 1.0f;
    // f16 mode, conv3 of a block whose successor's conv1 can run in the same launch (conv1x1_b2b.hip): the weight matrix in
    // the step-interleaved row order that kernel streams
    void* d_w3i = nullptr;
    // quantised models (INFURQ01): d_w = s8 OHWI with cin_p input channels and cout_p rows (zero padded to the K step of the i8
    // GEMM); d_qbias = operator bias + (128 - x_zp) * sum_k w (the kernel feeds x - 128); d_qmult = x_scale * w_scale[o] / y_scale
    int cin_p = 0, cout_p = 0;
    float x_scale = 0.f, y_scale = 0.f;
    int x_zp = 0, y_zp = 0;
    int32_t* d_qbias = nullptr;
    float* d_qmult = nullptr;
    // layer1 of a quantised model, PIXEL-PAIR view (forward_q): two horizontally adjacent pixels of a 64-channel tensor are one
    // 128-byte row of a (H, W/2, 128) image -- no channel padding -- and the layer's weights are re-arranged to act on pairs:
    // cin2 = 2 cin, cout2 = 2 cout, block structure [[W, 0], [0, W]] for the 1x1 convs, the three pair-columns of a 3x3
    int cin2 = 0, cout2 = 0;
    void* d_w2 = nullptr;
    int32_t* d_qbias2 = nullptr;
    float* d_qmult2 = nullptr;

    // EVERY device pointer of the layer goes through `f` (void* -> void*): the one place a replica of the model re-bases its
    // pointers into its own weight arena (infur_multi.cpp: adopt_model).  A new pointer field belongs in this list.
    template <class F>
    void map_device_pointers(F&& f) {
        auto ap = [&](auto*& p) { p = static_cast<std::remove_reference_t<decltype(p)>>(f((void*)p)); };
        ap(d_w); ap(d_b); ap(d_u); ap(d_uacc); ap(d_wl); ap(d_ul); ap(d_wcatl); ap(d_wcat); ap(d_bcat); ap(d_w3i);
        ap(d_qbias); ap(d_qmult); ap(d_w2); ap(d_qbias2); ap(d_qmult2);
    }
};

struct QAddParams {  // com.microsoft QLinearAdd of one bottleneck: C = A (conv3) + B (identity / downsample)
    float a_scale, b_scale, c_scale;
    int a_zp, b_zp, c_zp;
};

struct ProfRec {
    std::string name;
    const char* kernel;
    double flops, bytes, algo_flops;
    hipEvent_t e0, e1;
};

}  // namespace infur

struct infur_ctx {
    infur_options opt{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int pool_slot = -1;  // >= 0: `stream` is entry pool_slot of the device's stream pool (infur_capi.cpp), possibly shared
    std::string err;

    // lookup tables
    float* d_pre_lut = nullptr;     // [3][256] f32, RGB order: Float-input models (predict_onnx.rs:126-137)
    float* d_u8_lut = nullptr;      // [3][256] f32, lut[c][v] = v: Uint8-input models get the bytes themselves (:116-122)
    uint32_t* d_color_lut = nullptr;  // [20][256] premultiplied RGBA

    // model
    bool loaded = false;
    infur_model_info info{};
    int depth = 0, num_classes = 0;
    bool has_aux = false;
    bool input_u8 = false;  // the model declares a Uint8 image input: raw BGR bytes, no normalisation
    bool quant = false;     // a quantised (QOperator) model: u8 activations, s8 weights, the i8 MFMA -- whatever compute_dtype says
    std::vector<infur::QAddParams> qadds;
    // the fused quantised stem (launch_stem_pool_q): s8 weights as f32 [147][64], q - x_zp as f32 [3][256], the operator's own bias
    float* d_qstem_w = nullptr;
    float* d_qstem_lut = nullptr;
    int32_t* d_qstem_bias = nullptr;
    // quantised model whose file resizes the u8 logits before DequantizeLinear: out_low / aux_low hold the CODES, the post kernels
    // dequantise after the interpolation (kernels.h: UpQuant); [0] = out, [1] = aux
    bool q_resize_u8 = false;
    float q_head_zp[2] = {0.f, 0.f}, q_head_scale[2] = {1.f, 1.f};
    uint8_t* d_qlut = nullptr;  // [3][256] u8: byte value -> QuantizeLinear of the normalised value (RGB order); in d_weights
    std::vector<infur::ConvLayer> convs;
    void* d_weights = nullptr;  // single allocation holding every repacked tensor
    size_t weight_bytes = 0;

    // activation pool + results of the last forward
    std::vector<infur::Buf> pool;
    infur::Tensor out_low, aux_low;  // NHWC [lh][lw][K]
    // ---- hipGraph replay of the fused frame path (infur_ctx_set_graph_replay; infur_capi.cpp: frame_advance_dev) ----
    struct FrameGraph {
        const void* d_bgr = nullptr;
        void* d_rgba = nullptr;
        void* d_scaled = nullptr;
        uint32_t w = 0, h = 0, mode = 0, factor_bits = 0;
        hipGraphExec_t exec = nullptr;
        uint32_t ow = 0, oh = 0;
        infur::Tensor out_low, aux_low;  // what infur_model_read_lowres sees after a replay
        uint64_t stamp = 0;
    };
    bool graph_replay = false;
    std::vector<FrameGraph> graphs;
    uint64_t mem_gen = 0;      // bumped by every device allocation / release and model change: the cached graphs hold raw pointers
    uint64_t graphs_gen = 0;   // mem_gen the cached graphs were captured under
    uint32_t graph_streak = 0; // consecutive eager frames of one shape during which mem_gen did not move
    uint64_t streak_gen = 0, graph_clock = 0;
    uint32_t streak_w = 0, streak_h = 0, streak_mode = 0, streak_factor = 0;
    uint64_t graph_replays = 0, graph_captures = 0;
    int last_h = 0, last_w = 0;
    uint64_t frame_no = 0;   // forward() calls so far
    uint32_t same_size = 0;  // consecutive forwards at last_h x last_w
    // live infur_stream objects of this context: infur_ctx_destroy releases their device resources and
    // orphans them, so destroying context and streams in either order is safe
    std::vector<infur_stream*> streams;
    infur_stream* batch_ring = nullptr;  // infur_batch_advance's depth-3 ring, created on first use, owned by the context
    std::vector<infur::Tensor> kept;  // keep_activations: output of every conv

    // staging for the host-pointer entry points
    infur::Buf st_in, st_scaled, st_rgba, st_f32a, st_f32b;

    // profiling
    std::vector<infur::ProfRec> prof;
    std::vector<hipEvent_t> ev_free;

    // measured tile configuration per conv shape (see pick_cfg)
    std::map<std::array<int, 13>, int> tuned;
    bool tune_warm = false;

    // INFUR_DTYPE_F32_SPLIT range monitor: [0] max |activation| fed to a GEMM, [1] max |Winograd-domain input|
    // of the last forward (bit patterns of non-negative floats, atomicMax targets); infur_split_range
    unsigned* d_range = nullptr;
    // the f16-rate / quantised stem's weight image (launch_stem16_pack) and what it was built from: rebuilt when any of it changes
    void* d_stem16 = nullptr;
    const void* stem16_wt = nullptr;
    float stem16_scale = 0.f;
    int stem16_split = -1;
};


namespace infur {
// records the message on the context and returns `code`
int32_t ctx_fail(infur_ctx* c, int32_t code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
// every entry point runs on its context's device (hipMalloc / launches follow the calling thread's current device)
void ctx_enter(const infur_ctx* c);
// releases the context's weights and marks it unloaded
void ctx_model_free(infur_ctx* c);
// infur_stream.cpp: releases a streaming ring's resources and detaches it from its context(s); the handle stays allocated
void stream_orphan(infur_stream* st);
}  // namespace infur
