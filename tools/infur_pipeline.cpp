// infur_pipeline -- headless native front end of the hot path: packed bgr24 frames in, premultiplied
// RGBA masks out, through libinfur_hip.so.  It is what the `Proc` thread of the reference does (infur/src/main.rs:27-99)
// without the GUI: commands are given once on the command line, frames come from a pipe.
//
//   ffmpeg -i in.mp4 -an -f image2pipe -fflags nobuffer -pix_fmt bgr24 -c:v rawvideo pipe:1 |
//     infur_pipeline --width 1280 --height 720 --scale 0.5 --model fcn-resnet50-12.onnx > masks.rgba
//
// stdin carries what the reference's decoder reads from its ffmpeg child (ff-video/src/decoder.rs:53-64,156-165):
// W*H*3 bytes per frame.  stdout receives one mask (ow*oh*4 bytes) per frame, in order.  `--depth` frames are in
// flight (the reference's bounded channel of 2, main.rs:105); `--lanes 2` adds a second context of the same GPU that
// takes every other frame.  `--app` drives ProcessingApp::generate() instead (one frame at a time, app.rs:107-153).
// Build:  g++ -std=c++17 -O2 -Iinclude tools/infur_pipeline.cpp -Linfur_amd -linfur_hip -o infur_pipeline
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>

#include "infur_pipeline.hpp"

namespace {

struct Args {
    uint32_t width = 0, height = 0, depth = 2, lanes = 1, dtype = INFUR_DTYPE_F32, mode = INFUR_SCALE_NEAREST;
    int device = 0;
    float scale = 1.0f;
    uint64_t synthetic = 0;
    bool app = false, quiet = false, copy = false;
    std::string model, input = "-", output = "-";
};

int usage(const char* msg) {
    std::fprintf(stderr,
                 "%s\nusage: infur_pipeline --width W --height H --model PATH [--scale F] [--bilinear] [--dtype f32|f32s|f32x|f16|f16hl]\n"
                 "       [--depth N] [--lanes N] [--device D] [--input FILE|-] [--output FILE|-|none] [--synthetic N] [--app] [--copy] [--quiet]\n",
                 msg);
    return 2;
}

}  // namespace

int main(int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; i++) {
        const std::string k = argv[i];
        auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (k == "--width") a.width = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (k == "--height") a.height = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (k == "--scale") a.scale = std::strtof(val(), nullptr);
        else if (k == "--bilinear") a.mode = INFUR_SCALE_BILINEAR;
        else if (k == "--model") a.model = val();
        else if (k == "--depth") a.depth = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (k == "--lanes") a.lanes = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (k == "--device") a.device = std::atoi(val());
        else if (k == "--input") a.input = val();
        else if (k == "--output") a.output = val();
        else if (k == "--synthetic") a.synthetic = std::strtoull(val(), nullptr, 10);
        else if (k == "--app") a.app = true;
        else if (k == "--copy") a.copy = true;
        else if (k == "--quiet") a.quiet = true;
        else if (k == "--dtype") {
            const std::string d = val();
            if (d == "f32") a.dtype = INFUR_DTYPE_F32;
            else if (d == "f32s") a.dtype = INFUR_DTYPE_F32_SPLIT;
            else if (d == "f32x") a.dtype = INFUR_DTYPE_F32_SPLIT_FP8;
            else if (d == "f16") a.dtype = INFUR_DTYPE_F16;
            else if (d == "f16hl") a.dtype = INFUR_DTYPE_F16_HL;
            else return usage("unknown --dtype");
        } else return usage(("unknown argument " + k).c_str());
    }
    if (!a.width || !a.height) return usage("--width and --height are required");
    if (a.model.empty()) return usage("--model is required (INFURW01 blob or float fcn-resnet50/101 .onnx)");
    if (a.depth == 0 || a.lanes == 0 || a.lanes > 4) return usage("--depth >= 1, 1 <= --lanes <= 4");

    // Invalid commands are reported the way the reference relays them to its GUI (main.rs:69-71) and end the run here
    if (infur_scale_validate(a.scale) != INFUR_OK) {
        std::fprintf(stderr, "infur_pipeline: %s\n", infur_status_string(INFUR_E_INVALID_SCALE));
        return 1;
    }
    infur::Context ctx(a.device, true, a.dtype);
    if (!ctx.ok()) {
        std::fprintf(stderr, "infur_pipeline: %s\n", ctx.last_error().c_str());
        return 1;
    }

    std::FILE* fin = a.input == "-" ? stdin : std::fopen(a.input.c_str(), "rb");
    if (!fin && !a.synthetic) {
        std::fprintf(stderr, "infur_pipeline: cannot open %s\n", a.input.c_str());
        return 1;
    }
    std::FILE* fout = a.output == "none" ? nullptr : a.output == "-" ? stdout : std::fopen(a.output.c_str(), "wb");
    if (!fout && a.output != "none") {
        std::fprintf(stderr, "infur_pipeline: cannot open %s\n", a.output.c_str());
        return 1;
    }
    std::shared_ptr<infur::FrameSource> src;
    if (a.synthetic) src = std::make_shared<infur::SyntheticSource>(a.width, a.height, a.synthetic);
    else src = std::make_shared<infur::RawVideoSource>(fin, a.width, a.height, a.input != "-");

    uint64_t n = 0;
    int rc = 0;
    const auto t_setup = std::chrono::steady_clock::now();
    auto t0 = t_setup;  // the frame loop is timed from the end of the setup (model load, lanes)
    if (a.app) {
        infur::ProcessingApp app(ctx, true, a.mode);
        if (app.control_model_load(a.model) != INFUR_OK) {
            std::fprintf(stderr, "infur_pipeline: %s\n", ctx.last_error().c_str());
            return 1;
        }
        app.control_scale(a.scale);
        app.control_video(infur::VideoCmd::play(src));
        std::optional<infur::GUIFrame> g;
        t0 = std::chrono::steady_clock::now();
        for (;;) {
            const infur::AppResult r = app.generate(g);
            if (r.video == infur::VideoStatus::FinishedNormally) break;
            if (!r.ok()) {
                std::fprintf(stderr, "infur_pipeline: frame %llu: %s\n", (unsigned long long)(n + 1),
                             r.status != INFUR_OK ? ctx.last_error().c_str() : "short read on the frame stream");
                rc = 1;
                break;
            }
            if (g && g->decoded_buffer && fout) std::fwrite(g->decoded_buffer->rgba.data(), 1, g->decoded_buffer->rgba.size(), fout);
            n++;
        }
    } else {
        infur::Model model(ctx);
        if (model.control_load(a.model) != INFUR_OK) {
            std::fprintf(stderr, "infur_pipeline: %s\n", ctx.last_error().c_str());
            return 1;
        }
        infur::StreamPath sp(ctx, a.depth, a.mode);
        if (!sp.ok()) {
            std::fprintf(stderr, "infur_pipeline: %s\n", ctx.last_error().c_str());
            return 1;
        }
        // extra lanes: further contexts of the same device, weights replicated device-to-device
        std::vector<std::unique_ptr<infur::Context>> lanes;
        std::vector<infur::Context*> all{&ctx};
        for (uint32_t l = 1; l < a.lanes; l++) {
            lanes.push_back(std::make_unique<infur::Context>(a.device, true, a.dtype));
            if (!lanes.back()->ok()) {
                std::fprintf(stderr, "infur_pipeline: %s\n", lanes.back()->last_error().c_str());
                return 1;
            }
            all.push_back(lanes.back().get());
        }
        if (a.lanes > 1) {
            infur::Group g(all);
            if (!g.ok() || g.weights_broadcast(0) != INFUR_OK) {
                std::fprintf(stderr, "infur_pipeline: %s\n", g.ok() ? g.last_error().c_str() : "group creation failed");
                return 1;
            }
            for (auto& l : lanes)
                if (sp.add_lane(*l) != INFUR_OK) {
                    std::fprintf(stderr, "infur_pipeline: %s\n", ctx.last_error().c_str());
                    return 1;
                }
        }
        // first-frame work (activation arena, tile configurations of this frame size) is done before the clock starts
        uint32_t ow = 0, oh = 0;
        if (infur_scale_out_dims(a.width, a.height, a.scale, &ow, &oh) == INFUR_OK)
            for (infur::Context* c : all)
                if (infur_model_warmup(c->get(), ow, oh) != INFUR_OK) {
                    std::fprintf(stderr, "infur_pipeline: %s\n", c->last_error().c_str());
                    return 1;
                }
        t0 = std::chrono::steady_clock::now();
        // default: the pipe is read straight into the ring's pinned slots and the masks are written out of them (--copy: the copying
        // submit / collect calls, one pageable <-> pinned memcpy per frame and direction)
        const infur::Status s =
            a.copy ? sp.run(*src, a.scale, [&](uint64_t, const infur::ColorImage& m) { if (fout) std::fwrite(m.rgba.data(), 1, m.rgba.size(), fout); }, &n)
                   : sp.run_zero_copy(*src, a.scale, [&](const infur::StreamPath::View& v) { if (fout) std::fwrite(v.rgba, 1, (size_t)v.width * v.height * 4, fout); }, &n);
        if (s != INFUR_OK) {
            std::fprintf(stderr, "infur_pipeline: after %llu frames: %s\n", (unsigned long long)n,
                         s == INFUR_E_IO ? "short read on the frame stream" : ctx.last_error().c_str());
            rc = 1;
        }
    }
    if (fout) std::fflush(fout);
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!a.quiet)
        std::fprintf(stderr, "infur_pipeline: %llu frames in %.2f s (%.1f frames/s) after %.2f s of setup\n", (unsigned long long)n, el,
                     n / (el > 1e-9 ? el : 1e-9), std::chrono::duration<double>(t0 - t_setup).count());
    if (fout && fout != stdout) std::fclose(fout);
    return rc;
}
