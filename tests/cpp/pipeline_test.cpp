// Exercises include/infur_pipeline.hpp (C++ counterpart of VideoPlayer / ProcessingApp / the bounded frame queue).
//   pipeline_test cpu          -- host-only logic: frame sources, VideoPlayer state machine (processing.rs:62-139)
//   pipeline_test gpu BLOB     -- the reference's app tests (infur/src/app.rs:175-253) re-expressed on synthetic clips of
//                                 the same dimensions, error relaying (main.rs:69-71,94-96), fused == unfused,
//                                 streamed == one-at-a-time
#include <cstring>

#include "infur_pipeline.hpp"

#define CHECK(x)                                                        \
    do {                                                                \
        if (!(x)) {                                                     \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #x);    \
            return 1;                                                   \
        }                                                               \
    } while (0)

using infur::VideoCmd;
using infur::VideoStatus;

static std::shared_ptr<infur::FrameSource> short_large_input() { return std::make_shared<infur::SyntheticSource>(1280, 720, 150); }
static std::shared_ptr<infur::FrameSource> long_small_input() { return std::make_shared<infur::SyntheticSource>(640, 480, 400); }

static int cpu_tests() {
    // synthetic frames are deterministic per (seed, index) and differ between indices
    {
        infur::SyntheticSource a(32, 24, 2), b(32, 24, 2);
        infur::BgrImage ia(32, 24), ib(32, 24), ic(32, 24);
        uint64_t id = 0;
        CHECK(a.read_frame(ia, id) == VideoStatus::Ok && id == 1);
        CHECK(b.read_frame(ib, id) == VideoStatus::Ok && ia.data == ib.data);
        CHECK(a.read_frame(ic, id) == VideoStatus::Ok && id == 2 && ic.data != ia.data);
        CHECK(a.read_frame(ic, id) == VideoStatus::FinishedNormally);
    }
    // raw bgr24 stream: read_exact semantics (decoder.rs:156-165)
    {
        std::FILE* f = std::tmpfile();
        CHECK(f);
        std::vector<uint8_t> bytes(2 * 5 * 3 * 3 + 7);  // three 5x2 frames and a truncated fourth
        for (size_t i = 0; i < bytes.size(); i++) bytes[i] = (uint8_t)(i * 13 + 1);
        CHECK(std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size());
        std::rewind(f);
        infur::RawVideoSource src(f, 5, 2, true);
        infur::BgrImage img = src.empty_image();
        uint64_t id = 0;
        for (uint64_t k = 1; k <= 3; k++) {
            CHECK(src.read_frame(img, id) == VideoStatus::Ok && id == k);
            CHECK(!std::memcmp(img.data.data(), bytes.data() + (k - 1) * 30, 30));
        }
        CHECK(src.read_frame(img, id) == VideoStatus::ExactReadError);
        CHECK(src.read_frame(img, id) == VideoStatus::FinishedNormally);
    }
    // VideoPlayer: dirty only while playing and not paused; frame buffer re-created on size change only; closes at the end
    {
        infur::VideoPlayer v;
        std::optional<infur::Frame> fr;
        CHECK(!v.is_dirty() && v.advance(fr) == VideoStatus::Ok && !fr);
        CHECK(!v.control(VideoCmd::play(nullptr)));
        CHECK(v.control(VideoCmd::play(std::make_shared<infur::SyntheticSource>(16, 8, 2))) && v.is_dirty());
        CHECK(v.advance(fr) == VideoStatus::Ok && fr && fr->id == 1 && fr->img.width == 16);
        const uint8_t* buf = fr->img.data.data();
        CHECK(v.control(VideoCmd::pause(true)) && !v.is_dirty());
        CHECK(v.advance(fr) == VideoStatus::Ok && fr->id == 1);
        CHECK(v.control(VideoCmd::pause(false)) && v.is_dirty());
        CHECK(v.advance(fr) == VideoStatus::Ok && fr->id == 2 && fr->img.data.data() == buf);
        CHECK(v.advance(fr) == VideoStatus::FinishedNormally && !v.is_dirty() && fr->id == 2);
        CHECK(v.control(VideoCmd::play(std::make_shared<infur::SyntheticSource>(8, 8, 1))));
        CHECK(v.advance(fr) == VideoStatus::Ok && fr->img.width == 8 && fr->id == 1);
        CHECK(v.control(VideoCmd::stop()) && !v.is_dirty());
    }
    std::printf("cpu ok\n");
    return 0;
}

static int gpu_tests(const char* blob_path) {
    infur::Context c(0);
    CHECK(c.ok());
    std::optional<infur::GUIFrame> f1, f2, f3;
    {  // void (app.rs:175-180)
        infur::ProcessingApp app(c);
        CHECK(app.generate(f1).ok() && !f1);
        CHECK(app.generate(f1).ok() && !f1);
    }
    {  // scale (app.rs:182-189)
        infur::ProcessingApp app(c);
        app.control_video(VideoCmd::play(short_large_input()));
        CHECK(app.control_scale(0.5f) == INFUR_OK);
        CHECK(app.generate(f2).ok() && f2 && f2->buffer.width == 1280 / 2 && f2->buffer.height == 720 / 2 && !f2->decoded_buffer);
    }
    {  // switch_scale (app.rs:191-201)
        infur::ProcessingApp app(c);
        app.control_video(VideoCmd::play(long_small_input()));
        CHECK(app.generate(f1).ok() && f1->buffer.width == 640 && f1->buffer.height == 480);
        app.control_scale(0.5f);
        CHECK(app.generate(f2).ok() && f2->buffer.width == 320 && f2->buffer.height == 240);
    }
    {  // switch_video_then_scale (app.rs:203-218)
        infur::ProcessingApp app(c);
        app.control_video(VideoCmd::play(long_small_input()));
        CHECK(app.generate(f1).ok() && f1->buffer.width == 640 && f1->buffer.height == 480);
        app.control_video(VideoCmd::play(short_large_input()));
        CHECK(app.generate(f2).ok() && f2->buffer.width == 1280 && f2->buffer.height == 720);
        app.control_scale(2.0f);
        CHECK(app.generate(f3).ok() && f3->buffer.width == 2560 && f3->buffer.height == 1440);
    }
    {  // scaled_frame_after_stopped_video (app.rs:220-236)
        infur::ProcessingApp app(c);
        app.control_video(VideoCmd::play(short_large_input()));
        CHECK(app.generate(f1).ok() && f1->buffer.width == 1280);
        app.control_video(VideoCmd::stop());
        CHECK(app.generate(f2).ok() && f1->id == f2->id && !app.is_dirty());
        app.control_scale(0.5f);
        CHECK(app.is_dirty());
        CHECK(app.generate(f3).ok() && f2->id == f3->id && f3->buffer.width == 640 && f3->buffer.height == 360);
    }
    {  // pause_video (app.rs:238-252)
        infur::ProcessingApp app(c);
        app.control_video(VideoCmd::play(long_small_input()));
        CHECK(app.generate(f1).ok());
        app.control_video(VideoCmd::pause(true));
        CHECK(!app.is_dirty());
        CHECK(app.generate(f2).ok() && f1->id == f2->id && !app.is_dirty());
        app.control_video(VideoCmd::pause(false));
        CHECK(app.is_dirty());
        CHECK(app.generate(f3).ok() && f2->id != f3->id);
    }
    {  // errors are relayed, processing continues (main.rs:69-71,94-96)
        infur::ProcessingApp app(c);
        CHECK(app.control_scale(-1.0f) == INFUR_E_INVALID_SCALE);
        app.control_video(VideoCmd::play(std::make_shared<infur::SyntheticSource>(32, 24, 1)));
        CHECK(app.generate(f1).ok() && f1->id == 1);
        const infur::AppResult r = app.generate(f2);
        CHECK(r.video == VideoStatus::FinishedNormally && !f2);
        CHECK(app.generate(f3).ok() && f3 && f3->id == 1 && !app.is_dirty());  // carries on with the last frame
        // display copy: r,g,b,255 of the scaled frame (app.rs:132-144)
        infur::SyntheticSource again(32, 24, 1);
        infur::BgrImage img(32, 24);
        uint64_t id = 0;
        again.read_frame(img, id);
        for (size_t p = 0; p < (size_t)32 * 24; p++) {
            const uint8_t* q = &f3->buffer.rgba[4 * p];
            CHECK(q[0] == img.data[3 * p + 2] && q[1] == img.data[3 * p + 1] && q[2] == img.data[3 * p] && q[3] == 255);
        }
    }
    // with a model: fused route == Scale / Model / ColorCode chained; the streamed masks are the same bytes in the same order
    {
        std::vector<std::vector<uint8_t>> masks[2];
        for (int fused = 0; fused < 2; fused++) {
            infur::ProcessingApp app(c, fused != 0);
            CHECK(app.control_model_load("/nonexistent/file") == INFUR_E_IO && !app.info());
            CHECK(app.control_model_load(blob_path) == INFUR_OK);
            auto info = app.info();
            CHECK(info && info->output_names.size() == 2 && info->output_names[0] == "out");
            app.control_video(VideoCmd::play(std::make_shared<infur::SyntheticSource>(160, 96, 3)));
            app.control_scale(0.5f);
            for (int i = 0; i < 3; i++) {
                CHECK(app.generate(f1).ok() && f1 && f1->id == (uint64_t)(i + 1));
                CHECK(f1->decoded_buffer && f1->decoded_buffer->width == 80 && f1->decoded_buffer->height == 48);
                masks[fused].push_back(f1->decoded_buffer->rgba);
            }
            CHECK(app.generate(f1).video == VideoStatus::FinishedNormally);
            CHECK(app.control_model_load("") == INFUR_OK && !app.info());  // unload: the mask is cleared (app.rs:127-129)
            CHECK(app.generate(f2).ok() && f2 && !f2->decoded_buffer);
        }
        CHECK(masks[0] == masks[1]);
        CHECK(masks[0][0] != masks[0][1]);
        infur::Model m(c);
        CHECK(m.control_load(blob_path) == INFUR_OK);
        for (uint32_t lanes = 1; lanes <= 2; lanes++) {
            infur::Context c2(0);
            infur::StreamPath sp(c, 2);
            CHECK(sp.ok() && c2.ok());
            if (lanes == 2) {
                infur::Group g({&c, &c2});
                CHECK(g.ok() && g.weights_broadcast(0) == INFUR_OK);
                CHECK(sp.add_lane(c2) == INFUR_OK);
            }
            infur::SyntheticSource src(160, 96, 3);
            std::vector<std::vector<uint8_t>> got;
            std::vector<uint64_t> ids;
            uint64_t n = 0;
            CHECK(sp.run(src, 0.5f, [&](uint64_t id, const infur::ColorImage& mk) { ids.push_back(id); got.push_back(mk.rgba); }, &n) == INFUR_OK);
            CHECK(n == 3 && ids == (std::vector<uint64_t>{1, 2, 3}) && got == masks[1]);
        }
    }
    std::printf("gpu ok\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 2 && !std::strcmp(argv[1], "gpu")) return gpu_tests(argc >= 3 ? argv[2] : "");
    return cpu_tests();
}
