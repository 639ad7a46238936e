// infur_pipeline.hpp -- header-only C++ counterpart of the reference's processing graph:
//
//   VideoPlayer      infur/src/processing.rs:62-139   (over pluggable frame sources instead of an ffmpeg child)
//   ProcessingApp    infur/src/app.rs:51-158          vid -> scale -> model -> decode(out[0]) + BGR -> RGBA display copy
//   StreamPath       infur/src/main.rs:27-99,105      bounded queue of frames in flight (infur_stream_*)
//
// Same commands, same dirty / frame-id semantics, same error relaying (an error is returned to the caller and the next
// advance carries on, main.rs:69-71,94-96).  Rust `Result<_, E>` is an `infur::Status` (0 = Ok) plus, for the video
// source, a `VideoStatus`; `Option<T>` is `std::optional<T>`.  tools/infur_pipeline.cpp is the headless front end built
// on this header; tests/cpp/pipeline_test.cpp re-expresses app.rs:175-253 on it.
#pragma once
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>

#include "infur_processor.hpp"

namespace infur {

/// ff-video/src/error.rs: outcome of reading one frame
enum class VideoStatus { Ok, FinishedNormally, ExactReadError, NoSource };

/// What `FFMpegDecoder` is to the reference's VideoPlayer: dimensions + read_frame + close
class FrameSource {
public:
    virtual ~FrameSource() = default;
    virtual uint32_t width() const = 0;
    virtual uint32_t height() const = 0;
    /// fills `img` (already width x height), returns the 1-based frame id through `id` (decoder.rs:163-164)
    virtual VideoStatus read_frame(BgrImage& img, uint64_t& id) = 0;
    /// the same into caller-owned memory of width * height * 3 bytes -- a pinned slot of the streaming ring (StreamPath::run_zero_copy).
    /// Default: through a temporary image; RawVideoSource reads the pipe straight into `dst`.
    virtual VideoStatus read_into(uint8_t* dst, uint64_t& id) {
        BgrImage tmp = empty_image();
        const VideoStatus v = read_frame(tmp, id);
        if (v == VideoStatus::Ok) std::memcpy(dst, tmp.data.data(), tmp.data.size());
        return v;
    }
    virtual void close() {}
    BgrImage empty_image() const { return BgrImage(width(), height()); }
};

/// Packed bgr24 frames of known WxH from a stdio stream: exactly what `ffmpeg -an -f image2pipe -fflags nobuffer
/// -pix_fmt bgr24 -c:v rawvideo pipe:1` writes and the reference reads with read_exact(W*H*3)
/// (ff-video/src/decoder.rs:53-64,156-165)
class RawVideoSource : public FrameSource {
public:
    RawVideoSource(std::FILE* f, uint32_t w, uint32_t h, bool close_file = false) : f_(f), w_(w), h_(h), close_(close_file) {}
    ~RawVideoSource() override { close(); }
    uint32_t width() const override { return w_; }
    uint32_t height() const override { return h_; }
    VideoStatus read_frame(BgrImage& img, uint64_t& id) override { return read_into(img.data.data(), id); }
    VideoStatus read_into(uint8_t* dst, uint64_t& id) override {
        if (!f_) return VideoStatus::NoSource;
        const size_t n = (size_t)w_ * h_ * 3;
        size_t got = 0;
        while (got < n) {  // read_exact
            const size_t k = std::fread(dst + got, 1, n - got, f_);
            if (k == 0) break;
            got += k;
        }
        if (got == 0) return VideoStatus::FinishedNormally;
        if (got < n) return VideoStatus::ExactReadError;
        id = ++counter_;
        return VideoStatus::Ok;
    }
    void close() override {
        if (f_ && close_) std::fclose(f_);
        f_ = nullptr;
    }

private:
    std::FILE* f_;
    uint32_t w_, h_;
    bool close_;
    uint64_t counter_ = 0;
};

/// Deterministic frames for tests and benchmarks (no ffmpeg / lavfi testsrc in the build image): a counter-based
/// byte pattern, `n_frames` of them (0 = endless)
class SyntheticSource : public FrameSource {
public:
    SyntheticSource(uint32_t w, uint32_t h, uint64_t n_frames = 0, uint32_t seed = 0) : w_(w), h_(h), n_(n_frames), seed_(seed) {}
    uint32_t width() const override { return w_; }
    uint32_t height() const override { return h_; }
    VideoStatus read_frame(BgrImage& img, uint64_t& id) override {
        if (n_ && counter_ >= n_) return VideoStatus::FinishedNormally;
        uint64_t s = 0x1F0A2026ull ^ (uint64_t)(seed_ + counter_) * 0x9E3779B97F4A7C15ull;
        uint8_t* p = img.data.data();
        for (uint32_t y = 0; y < h_; y++)
            for (uint32_t x = 0; x < w_; x++) {
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                const uint32_t r = (uint32_t)(s >> 40);
                const uint32_t gx = w_ > 1 ? 255u * x / (w_ - 1) : 0u, gy = h_ > 1 ? 255u * y / (h_ - 1) : 0u;
                *p++ = (uint8_t)(((r & 255u) + gx) >> 1);
                *p++ = (uint8_t)((((r >> 8) & 255u) + gy) >> 1);
                *p++ = (uint8_t)((((r >> 16) & 255u) + ((gx + gy) >> 1)) >> 1);
            }
        id = ++counter_;
        return VideoStatus::Ok;
    }

private:
    uint32_t w_, h_;
    uint64_t n_;
    uint32_t seed_;
    uint64_t counter_ = 0;
};

/// processing.rs:62-71
struct VideoCmd {
    enum Kind { Play, Pause, Stop } kind = Stop;
    std::shared_ptr<FrameSource> source;  // Play
    bool paused = false;                  // Pause
    static VideoCmd play(std::shared_ptr<FrameSource> s) { return VideoCmd{Play, std::move(s), false}; }
    static VideoCmd pause(bool p) { return VideoCmd{Pause, nullptr, p}; }
    static VideoCmd stop() { return VideoCmd{Stop, nullptr, false}; }
};

/// processing.rs:73-139.  Input = (), Output = Option<Frame>.
class VideoPlayer {
public:
    /// false = FFVideoError (Play without a source)
    bool control(const VideoCmd& cmd) {
        switch (cmd.kind) {
            case VideoCmd::Play:
                close_video();
                if (!cmd.source) return false;
                vid_ = cmd.source;
                break;
            case VideoCmd::Pause: paused_ = cmd.paused; break;
            case VideoCmd::Stop: close_video(); break;
        }
        return true;
    }
    bool is_dirty() const { return !paused_ && vid_ != nullptr; }  // processing.rs:110-112
    VideoStatus advance(std::optional<Frame>& out) {
        if (paused_ || !vid_) return VideoStatus::Ok;
        if (!out) out = Frame{0, vid_->empty_image()};
        if (out->img.width != vid_->width() || out->img.height != vid_->height()) out->img = vid_->empty_image();  // :121-131
        uint64_t id = 0;
        const VideoStatus s = vid_->read_frame(out->img, id);
        if (s == VideoStatus::FinishedNormally) close_video();  // :133-135
        if (s == VideoStatus::Ok) out->id = id;
        return s;
    }

private:
    void close_video() {
        if (vid_) vid_->close();
        vid_.reset();
    }
    std::shared_ptr<FrameSource> vid_;
    bool paused_ = false;
};

/// app.rs:64-69: frame id, the scaled frame as RGBA for display, the optional mask
struct GUIFrame {
    uint64_t id = 0;
    ColorImage buffer;                         // r,g,b,255
    std::optional<ColorImage> decoded_buffer;  // premultiplied mask; nullopt without a model
};

/// What one `advance` reports besides the frame: the first error met, by stage (main.rs relays it and carries on)
struct AppResult {
    VideoStatus video = VideoStatus::Ok;
    Status status = INFUR_OK;  // Scale / Model / ColorCode / display conversion
    bool ok() const { return video == VideoStatus::Ok && status == INFUR_OK; }
};

/// app.rs:51-158.  `fused` takes the device-resident route for model + decode (infur_frame_advance: no full-resolution
/// logits cross the boundary; app.rs:116 decodes out[0] only), otherwise Scale / Model / ColorCode are chained exactly
/// as app.rs:112-123 does.  Both give the same masks (tests).
class ProcessingApp {
public:
    explicit ProcessingApp(Context& c, bool fused = true, uint32_t scale_mode = INFUR_SCALE_NEAREST)
        : c_(c), fused_(fused), scale_mode_(scale_mode), scale_(c, scale_mode), model_(c), decoder_(c) {}

    bool control_video(const VideoCmd& cmd) { return vid_.control(cmd); }
    Status control_scale(float factor) { return scale_.control(factor); }
    Status control_model_load(const std::string& path) { return model_.control_load(path); }
    Status control_model_load_blob(const void* blob, size_t len) { return model_.control_load_blob(blob, len); }
    void control_exit() { to_exit_ = true; }
    bool to_exit() const { return to_exit_; }
    std::optional<ModelInfo> info() const { return model_.get_info(); }
    bool is_dirty() const { return vid_.is_dirty() || scale_.is_dirty(); }  // app.rs:155-157

    /// app.rs:107-153; `out` is nullopt until a first frame exists
    AppResult advance(std::optional<GUIFrame>& out) {
        AppResult r;
        out.reset();
        r.video = vid_.advance(frame_);
        if (r.video != VideoStatus::Ok) return r;  // `?` at app.rs:108
        if (is_dirty()) {                          // only Scale is gated (app.rs:109-111)
            r.status = scale_.advance(frame_, scaled_);
            if (r.status != INFUR_OK) return r;
        }
        if (!scaled_) return r;
        const BgrImage& sf = scaled_->img;
        if (fused_) {
            if (!decoded_ || decoded_->width != sf.width || decoded_->height != sf.height) {
                ColorImage m;
                m.width = sf.width;
                m.height = sf.height;
                m.rgba.assign((size_t)sf.width * sf.height * 4, 0);
                decoded_ = std::move(m);
            }
            uint32_t ow = 0, oh = 0;
            r.status = infur_frame_advance(c_.get(), sf.data.data(), sf.width, sf.height, 1.0f, scale_mode_, decoded_->rgba.data(),
                                           decoded_->rgba.size(), nullptr, &ow, &oh);
            if (r.status == INFUR_E_MODEL_NOT_LOADED) {  // no model: the mask is cleared (app.rs:127-129), not an error
                decoded_.reset();
                r.status = INFUR_OK;
            }
            if (r.status != INFUR_OK) return r;
        } else {
            tensors_.clear();
            r.status = model_.advance(sf, tensors_);  // runs on every generate(), dirty or not (app.rs:113-114)
            if (r.status != INFUR_OK) return r;
            if (!tensors_.empty()) {
                r.status = decoder_.advance(tensors_[0], decoded_);  // only out[0] (app.rs:116)
                if (r.status != INFUR_OK) return r;
            } else {
                decoded_.reset();
            }
        }
        GUIFrame g;
        g.id = scaled_->id;
        g.buffer.width = sf.width;
        g.buffer.height = sf.height;
        g.buffer.rgba.resize((size_t)sf.width * sf.height * 4);
        r.status = infur_bgr_to_rgba(c_.get(), sf.data.data(), sf.width, sf.height, g.buffer.rgba.data());  // app.rs:132-144
        if (r.status != INFUR_OK) return r;
        g.decoded_buffer = decoded_;
        out = std::move(g);
        return r;
    }
    AppResult generate(std::optional<GUIFrame>& out) { return advance(out); }

private:
    Context& c_;
    bool fused_;
    uint32_t scale_mode_;
    VideoPlayer vid_;
    Scale scale_;
    Model model_;
    ColorCode decoder_;
    std::optional<Frame> frame_, scaled_;
    std::optional<ColorImage> decoded_;
    std::vector<Tensor3> tensors_;
    bool to_exit_ = false;
};

/// Ring of `depth` frames in flight (the reference's sync_channel(2), main.rs:105) over infur_stream_*: uploads,
/// kernels and downloads of neighbouring frames overlap; masks come back strictly in submission order.
class StreamPath {
public:
    explicit StreamPath(Context& c, uint32_t depth = 2, uint32_t scale_mode = INFUR_SCALE_NEAREST) : c_(c), depth_(depth), mode_(scale_mode) {
        status_ = infur_stream_create(c.get(), depth, &st_);
    }
    ~StreamPath() { infur_stream_destroy(st_); }
    StreamPath(const StreamPath&) = delete;
    StreamPath& operator=(const StreamPath&) = delete;
    bool ok() const { return status_ == INFUR_OK; }
    uint32_t depth() const { return depth_; }
    /// a second context of the same device (with the model loaded) takes every other frame
    Status add_lane(Context& other) { return infur_stream_add_lane(st_, other.get()); }
    uint32_t pending() const { return infur_stream_pending(st_); }
    Status submit(const BgrImage& img, float factor, uint64_t frame_id) {
        return infur_stream_submit(st_, img.data.data(), img.width, img.height, factor, mode_, frame_id);
    }
    /// the oldest pending frame; `mask` is resized only when the dimensions change
    Status collect(uint64_t& frame_id, ColorImage& mask, BgrImage* scaled = nullptr) {
        uint32_t ow = 0, oh = 0;
        Status s = infur_stream_next_dims(st_, &frame_id, &ow, &oh);
        if (s != INFUR_OK) return s;
        if (mask.width != ow || mask.height != oh) {
            mask.width = ow;
            mask.height = oh;
            mask.rgba.assign((size_t)ow * oh * 4, 0);
        }
        if (scaled && (scaled->width != ow || scaled->height != oh)) *scaled = BgrImage(ow, oh);
        return infur_stream_collect(st_, mask.rgba.data(), mask.rgba.size(), scaled ? scaled->data.data() : nullptr, &frame_id, &ow, &oh);
    }
    // ---- zero-copy ingest / egress (ABI 5): the ring's pinned slots lent to the caller ----
    /// a mask (and the scaled frame) in place in the ring's pinned output slot: valid until release()
    struct View {
        const uint8_t* rgba = nullptr;
        const uint8_t* scaled_bgr = nullptr;
        uint32_t width = 0, height = 0;
        uint64_t frame_id = 0;
    };
    Status acquire(uint32_t w, uint32_t h, float factor, uint8_t*& slot) { return infur_stream_acquire(st_, w, h, factor, &slot); }
    Status commit(uint32_t w, uint32_t h, float factor, uint64_t frame_id) { return infur_stream_commit(st_, w, h, factor, mode_, frame_id); }
    Status abandon() { return infur_stream_abandon(st_); }
    Status collect_view(View& v) { return infur_stream_collect_view(st_, &v.rgba, &v.scaled_bgr, &v.frame_id, &v.width, &v.height); }
    Status release() { return infur_stream_release(st_); }
    /// as run(), without the two pageable <-> pinned copies per frame: the source reads each frame straight into the next pinned slot
    /// (what the reference's decoder does with its reused BgrImage, ff-video/src/decoder.rs:156-165) and `sink(view)` sees the mask in
    /// place in the pinned output slot
    Status run_zero_copy(FrameSource& src, float factor, const std::function<void(const View&)>& sink, uint64_t* n_frames = nullptr) {
        uint64_t n = 0, id = 0;
        Status err = INFUR_OK;
        auto drain_one = [&]() -> Status {
            View v;
            Status s = collect_view(v);
            if (s != INFUR_OK) return s;
            sink(v);
            n++;
            return release();
        };
        for (;;) {
            if (pending() >= depth_ && (err = drain_one()) != INFUR_OK) break;
            uint8_t* slot = nullptr;
            if ((err = acquire(src.width(), src.height(), factor, slot)) != INFUR_OK) break;
            const VideoStatus v = src.read_into(slot, id);
            if (v != VideoStatus::Ok) {
                if (v != VideoStatus::FinishedNormally) err = INFUR_E_IO;
                (void)abandon();  // the acquired slot goes back uncommitted: copying submits work again on this stream
                break;
            }
            if ((err = commit(src.width(), src.height(), factor, id)) != INFUR_OK) break;
        }
        while (pending()) {
            const Status s = drain_one();
            if (s != INFUR_OK) {
                if (err == INFUR_OK) err = s;
                break;
            }
        }
        if (n_frames) *n_frames = n;
        return err;
    }

    /// pump a source through the ring: `sink(id, mask)` is called once per frame, in order.  Returns the first error
    /// (video errors other than FinishedNormally map to INFUR_E_IO); frames already in flight are drained first.
    Status run(FrameSource& src, float factor, const std::function<void(uint64_t, const ColorImage&)>& sink, uint64_t* n_frames = nullptr) {
        BgrImage img = src.empty_image();
        ColorImage mask;
        uint64_t n = 0, id = 0;
        Status err = INFUR_OK;
        for (;;) {
            const VideoStatus v = src.read_frame(img, id);
            if (v != VideoStatus::Ok) {
                if (v != VideoStatus::FinishedNormally) err = INFUR_E_IO;
                break;
            }
            if (pending() >= depth_) {
                uint64_t fid = 0;
                if ((err = collect(fid, mask)) != INFUR_OK) break;
                sink(fid, mask);
                n++;
            }
            if ((err = submit(img, factor, id)) != INFUR_OK) break;  // submit() copies the frame into a pinned slot
        }
        while (pending()) {
            uint64_t fid = 0;
            const Status s = collect(fid, mask);
            if (s != INFUR_OK) {
                if (err == INFUR_OK) err = s;
                break;
            }
            sink(fid, mask);
            n++;
        }
        if (n_frames) *n_frames = n;
        return err;
    }

private:
    Context& c_;
    uint32_t depth_, mode_;
    infur_stream* st_ = nullptr;
    Status status_ = INFUR_OK;
};

}  // namespace infur
