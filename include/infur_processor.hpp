// infur_processor.hpp -- header-only C++ mirror of the reference's `Processor` plugin surface
// over the C ABI of infur_hip.h.  (The reference is Rust; no Rust toolchain exists in the build
// image, so the compiled-language host layer is C++.  The Rust adapter is in INTEGRATION.md.)
//
//   trait Processor          infur/src/processing.rs:23-60
//   Scale                    infur/src/processing.rs:179-282
//   Model<f32>               infur/src/predict_onnx.rs:146-345
//   ColorCode                infur/src/decode_predict.rs:38-84
//
// Rust `Result<_, E>` becomes a status code (`infur::Status`, 0 = Ok) carried by small error
// structs; `&mut Option<T>` outputs become `std::optional<T>&`; buffers are reused across calls
// exactly where the reference reuses them.
#pragma once
#include <cstdint>
#include <optional>
#include <string>
#include <vector>

#include "infur_hip.h"

namespace infur {

using Status = int32_t;

/// processing.rs:9-18 -- `Frame { id, img: BgrImage }`; packed B,G,R u8, row-major, no padding
struct BgrImage {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> data;  // height * width * 3
    BgrImage() = default;
    BgrImage(uint32_t w, uint32_t h) : width(w), height(h), data((size_t)w * h * 3, 0) {}
};
struct Frame {
    uint64_t id = 0;
    BgrImage img;
    bool operator==(const Frame& o) const { return id == o.id; }  // processing.rs:14-18
};

/// epaint ColorImage: premultiplied r,g,b,a bytes
struct ColorImage {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba;  // height * width * 4
};

/// [K, H, W] f32 planar tensor (ndarray Array3 / ArrayD of the reference)
struct Tensor3 {
    uint32_t k = 0, h = 0, w = 0;
    std::vector<float> data;
};

/// RAII owner of an infur_ctx (one GPU, one stream; not thread-safe)
class Context {
public:
    /// compute_dtype: INFUR_DTYPE_F32 (exact f32 MFMA), INFUR_DTYPE_F32_SPLIT_FP8 (see infur_hip.h), INFUR_DTYPE_F32_SPLIT (f32 tensors, f16 matrix cores with
    /// hi+lo operand pairs: f32-grade logits at ~1.9x the rate) or INFUR_DTYPE_F16
    explicit Context(int device = 0, bool compute_aux = true, uint32_t compute_dtype = INFUR_DTYPE_F32) {
        infur_options o;
        infur_options_default(&o);
        o.device = device;
        o.compute_aux = compute_aux ? 1 : 0;
        o.compute_dtype = compute_dtype;
        status_ = infur_ctx_create(&o, &ctx_);
    }
    ~Context() { infur_ctx_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    bool ok() const { return status_ == INFUR_OK; }
    Status status() const { return status_; }
    infur_ctx* get() const { return ctx_; }
    std::string last_error() const { return ctx_ ? infur_last_error(ctx_) : infur_status_string(status_); }

private:
    infur_ctx* ctx_ = nullptr;
    Status status_ = INFUR_OK;
};

/// processing.rs:179-282.  Command = f32, Input = Output = Option<Frame>.
class Scale {
public:
    explicit Scale(Context& c, uint32_t mode = INFUR_SCALE_NEAREST) : c_(c), mode_(mode) {}

    /// ValidScale::try_from + dirty tracking (processing.rs:220-226).  On error state is untouched.
    Status control(float factor) {
        Status s = infur_scale_validate(factor);
        if (s != INFUR_OK) return s;  // ValidScaleError
        dirty_ = factor != factor_;
        factor_ = factor;
        return INFUR_OK;
    }
    bool is_dirty() const { return dirty_; }

    /// processing.rs:232-281
    Status advance(const std::optional<Frame>& input, std::optional<Frame>& out) {
        dirty_ = false;
        if (!input) return INFUR_OK;
        if (factor_ == 1.0f) {  // clone
            out = *input;
            return INFUR_OK;
        }
        uint32_t ow = 0, oh = 0;
        Status s = infur_scale_out_dims(input->img.width, input->img.height, factor_, &ow, &oh);
        if (s != INFUR_OK) return s;  // ZeroSizeIn / ZeroSizeOut
        if (!out) out = Frame{input->id, BgrImage(ow, oh)};
        if (out->img.width != ow || out->img.height != oh) out->img = BgrImage(ow, oh);  // only on size change
        out->id = input->id;
        return infur_scale(c_.get(), input->img.data.data(), input->img.width, input->img.height, factor_, mode_,
                           out->img.data.data(), out->img.data.size(), &ow, &oh);
    }

private:
    Context& c_;
    uint32_t mode_;
    float factor_ = 1.0f;  // Default: ValidScale(1.0), dirty = true (processing.rs:185-193)
    bool dirty_ = true;
};

/// predict_onnx.rs:56-62
struct ModelInfo {
    std::vector<std::string> input_names;
    std::string input0_dtype;
    std::vector<std::string> output_names;
    bool quantised = false;        ///< a QOperator / QDQ int8 model: runs on the i8 MFMA whatever the context's dtype (ABI 4)
    bool resize_u8_heads = false;  ///< ... whose file resizes the u8 logits before DequantizeLinear
};

/// predict_onnx.rs:146-345.  Command = ModelCmd::Load(path), Input = BgrImage, Output = Vec<ArrayD<f32>>.
class Model {
public:
    explicit Model(Context& c) : c_(c) {}

    /// ModelCmd::Load(path); empty path unloads (predict_onnx.rs:288-312)
    Status control_load(const std::string& path) { return infur_model_load(c_.get(), path.c_str()); }
    Status control_load_blob(const void* blob, size_t len) { return infur_model_load_blob(c_.get(), blob, len); }
    bool is_dirty() const { return false; }  // predict_onnx.rs:336-338

    std::optional<ModelInfo> get_info() const {
        infur_model_info mi;
        if (infur_model_info_get(c_.get(), &mi) != INFUR_OK) return std::nullopt;
        ModelInfo r;
        r.input_names = {mi.input_name};
        r.input0_dtype = mi.input0_dtype;
        for (uint32_t i = 0; i < mi.n_outputs; i++) r.output_names.push_back(mi.output_names[i]);
        r.quantised = mi.quantised != 0;
        r.resize_u8_heads = mi.resize_u8_heads != 0;
        return r;
    }

    /// predict_onnx.rs:317-334: with no model `out` is left untouched and Ok is returned.  `out` receives as many
    /// tensors as the model has outputs (2, or 1 without the aux head / with compute_aux = false).
    Status advance(const BgrImage& img, std::vector<Tensor3>& out) {
        infur_model_info mi;
        if (infur_model_info_get(c_.get(), &mi) != INFUR_OK) return INFUR_OK;
        std::vector<Tensor3> t(mi.n_outputs);
        for (auto& x : t) {
            x.k = mi.num_classes;
            x.h = img.height;
            x.w = img.width;
            x.data.resize((size_t)x.k * x.h * x.w);
        }
        uint32_t n = 0;
        Status s = infur_model_advance(c_.get(), img.data.data(), img.width, img.height, t[0].data.data(),
                                       t.size() > 1 ? t[1].data.data() : nullptr, &n);
        if (s != INFUR_OK) return s;
        out.clear();
        for (auto& x : t) out.push_back(std::move(x));
        return INFUR_OK;
    }

private:
    Context& c_;
};

/// decode_predict.rs:38-84.  Input = Array3<f32> [K,H,W], Output = Option<ColorImage>.
class ColorCode {
public:
    explicit ColorCode(Context& c) : c_(c) {}
    Status control() { return INFUR_OK; }
    bool is_dirty() const { return false; }
    Status advance(const Tensor3& inp, std::optional<ColorImage>& out) {
        if (!out || out->width != inp.w || out->height != inp.h) {  // decode_predict.rs:58-65
            ColorImage img;
            img.width = inp.w;
            img.height = inp.h;
            img.rgba.assign((size_t)inp.w * inp.h * 4, 0);
            out = std::move(img);
        }
        return infur_colorcode(c_.get(), inp.data.data(), inp.k, inp.h, inp.w, out->rgba.data());
    }

private:
    Context& c_;
};

/// infur_group: n contexts (one per GPU) of one process -- RCCL weight broadcast + frame-batch sharding
/// (BASELINE configs[3]).  The contexts must outlive the group.
class Group {
public:
    explicit Group(const std::vector<Context*>& ctxs) {
        std::vector<infur_ctx*> raw;
        for (Context* c : ctxs) raw.push_back(c->get());
        status_ = infur_group_create(raw.data(), (uint32_t)raw.size(), &g_);
    }
    ~Group() { infur_group_destroy(g_); }
    Group(const Group&) = delete;
    Group& operator=(const Group&) = delete;
    bool ok() const { return status_ == INFUR_OK; }
    infur_group* get() const { return g_; }
    std::string last_error() const { return infur_group_last_error(g_); }
    Status weights_broadcast(uint32_t root = 0) { return infur_group_weights_broadcast(g_, root); }
    /// masks[i] is resized to the mask of frames[i] (scale `factor` applied first, as app.rs:107-153 does per frame)
    Status batch_advance(const std::vector<BgrImage>& frames, float factor, std::vector<ColorImage>& masks,
                         uint32_t scale_mode = INFUR_SCALE_NEAREST) {
        const size_t n = frames.size();
        masks.resize(n);
        std::vector<const uint8_t*> in(n);
        std::vector<uint8_t*> out(n);
        std::vector<uint32_t> ws(n), hs(n);
        std::vector<size_t> caps(n);
        for (size_t i = 0; i < n; i++) {
            uint32_t ow = 0, oh = 0;
            Status s = infur_scale_out_dims(frames[i].width, frames[i].height, factor, &ow, &oh);
            if (s != INFUR_OK) return s;
            masks[i].width = ow;
            masks[i].height = oh;
            masks[i].rgba.resize((size_t)ow * oh * 4);
            in[i] = frames[i].data.data();
            out[i] = masks[i].rgba.data();
            ws[i] = frames[i].width;
            hs[i] = frames[i].height;
            caps[i] = masks[i].rgba.size();
        }
        return infur_group_batch_advance(g_, in.data(), ws.data(), hs.data(), (uint32_t)n, factor, scale_mode, out.data(),
                                         caps.data(), nullptr, nullptr);
    }

private:
    infur_group* g_ = nullptr;
    Status status_ = INFUR_OK;
};

}  // namespace infur
