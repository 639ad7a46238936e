// Exercises include/infur_processor.hpp (the C++ mirror of the reference's Processor trait).
//   host_test cpu   -- host-only logic, runs without a GPU (context creation must fail loudly)
//   host_test gpu   -- the reference's own unit tests re-expressed over the HIP path:
//                      processing.rs:289-303, decode_predict.rs:100-116, predict_onnx.rs:371-381
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "infur_processor.hpp"

#define CHECK(x)                                                        \
    do {                                                                \
        if (!(x)) {                                                     \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #x);    \
            return 1;                                                   \
        }                                                               \
    } while (0)

static int cpu_tests() {
    CHECK(infur_abi_version() == INFUR_ABI_VERSION);
    CHECK(infur_scale_validate(0.0f) == INFUR_E_INVALID_SCALE);
    CHECK(infur_scale_validate(-1.0f) == INFUR_E_INVALID_SCALE);
    CHECK(infur_scale_validate(0.5f) == INFUR_OK);
    uint32_t ow = 0, oh = 0;
    CHECK(infur_scale_out_dims(0, 10, 0.99f, &ow, &oh) == INFUR_E_ZERO_SIZE_IN);
    CHECK(infur_scale_out_dims(10, 10, 0.00000001f, &ow, &oh) == INFUR_E_ZERO_SIZE_OUT);
    CHECK(infur_scale_out_dims(1280, 720, 0.5f, &ow, &oh) == INFUR_OK && ow == 640 && oh == 360);
    CHECK(infur_scale_out_dims(640, 480, 0.5f, &ow, &oh) == INFUR_OK && ow == 320 && oh == 240);
    CHECK(infur_scale_out_dims(1280, 720, 2.0f, &ow, &oh) == INFUR_OK && ow == 2560 && oh == 1440);
    if (infur_device_count() == 0) {
        infur::Context c(0);
        CHECK(!c.ok() && c.status() == INFUR_E_HIP);  // no CPU fallback
    }
    std::printf("cpu ok\n");
    return 0;
}

static int gpu_tests(const char* blob_path) {
    infur::Context c(0);
    CHECK(c.ok());
    // scale_from_size0 / scale_to_size0
    {
        infur::Scale s(c);
        CHECK(s.is_dirty());
        CHECK(s.control(0.99f) == INFUR_OK);
        std::optional<infur::Frame> out;
        CHECK(s.advance(infur::Frame{0, infur::BgrImage(0, 10)}, out) == INFUR_E_ZERO_SIZE_IN);
        CHECK(s.control(0.00000001f) == INFUR_OK);
        CHECK(s.advance(infur::Frame{0, infur::BgrImage(10, 10)}, out) == INFUR_E_ZERO_SIZE_OUT);
        CHECK(s.control(-1.0f) == INFUR_E_INVALID_SCALE);
        CHECK(s.control(0.5f) == INFUR_OK && s.is_dirty());
        infur::Frame f{7, infur::BgrImage(640, 480)};
        for (size_t i = 0; i < f.img.data.size(); i++) f.img.data[i] = (uint8_t)(i * 31);
        CHECK(s.advance(f, out) == INFUR_OK && !s.is_dirty());
        CHECK(out && out->id == 7 && out->img.width == 320 && out->img.height == 240);
        const uint8_t* before = out->img.data.data();
        f.id = 8;
        CHECK(s.advance(f, out) == INFUR_OK && out->id == 8 && out->img.data.data() == before);  // buffer reused
    }
    // decode_0to1
    {
        infur::ColorCode cc(c);
        infur::Tensor3 hm;
        hm.k = 22; hm.h = 24; hm.w = 32;
        const size_t n = (size_t)22 * 24 * 32;
        hm.data.resize(n);
        for (size_t i = 0; i < n; i++) hm.data[i] = (float)((double)i / (double)(n - 1));
        std::optional<infur::ColorImage> img;
        CHECK(cc.advance(hm, img) == INFUR_OK);
        CHECK(img && img->width == 32 && img->height == 24);
        int conf = 0;
        for (size_t p = 0; p < (size_t)24 * 32; p++) {
            const int a = img->rgba[4 * p + 3];
            CHECK(conf <= a);
            conf = a;
        }
        CHECK(conf == 255);
        // last pixel: class 21 -> palette[1] = (75, 25, 230), alpha 255 -> unmodified colour
        const uint8_t* l = &img->rgba[4 * ((size_t)24 * 32 - 1)];
        CHECK(l[0] == 75 && l[1] == 25 && l[2] == 230 && l[3] == 255);
    }
    // infer_seg_model: black 320x240 -> two tensors [21,240,320]; no model -> out untouched
    {
        infur::Model m(c);
        std::vector<infur::Tensor3> out;
        CHECK(!m.get_info());
        CHECK(m.advance(infur::BgrImage(320, 240), out) == INFUR_OK && out.empty());
        CHECK(m.control_load("/nonexistent/file") == INFUR_E_IO);
        CHECK(m.control_load(blob_path) == INFUR_OK);
        auto info = m.get_info();
        CHECK(info && info->input_names[0] == "input" && info->output_names.size() == 2 &&
              info->output_names[0] == "out" && info->output_names[1] == "aux");
        CHECK(m.advance(infur::BgrImage(320, 240), out) == INFUR_OK);
        CHECK(out.size() == 2);
        for (auto& t : out) {
            CHECK(t.k == 21 && t.h == 240 && t.w == 320);
            for (float v : t.data) CHECK(std::isfinite(v));
        }
        CHECK(m.control_load("") == INFUR_OK && !m.get_info());  // Load("") unloads
    }
    // a context that does not evaluate the aux head has ONE output (the Vec the reference would return has length 1)
    {
        infur::Context c1(0, /*compute_aux=*/false);
        CHECK(c1.ok());
        infur::Model m(c1);
        CHECK(m.control_load(blob_path) == INFUR_OK);
        auto info = m.get_info();
        CHECK(info && info->output_names.size() == 1 && info->output_names[0] == "out");
        std::vector<infur::Tensor3> out;
        CHECK(m.advance(infur::BgrImage(96, 64), out) == INFUR_OK && out.size() == 1 && out[0].k == 21 && out[0].h == 64);
        std::vector<float> aux((size_t)21 * 64 * 96);
        infur::BgrImage img(96, 64);
        uint32_t n = 9;
        CHECK(infur_model_advance(c1.get(), img.data.data(), 96, 64, nullptr, aux.data(), &n) == INFUR_E_INVALID_ARG);
        // a failed reload keeps the loaded model (Model::control leaves the session on error, predict_onnx.rs:288-309)
        CHECK(m.control_load("/nonexistent/file") == INFUR_E_IO && m.get_info());
        const char junk[64] = "INFURW01 but not really";
        CHECK(m.control_load_blob(junk, sizeof junk) == INFUR_E_MODEL_FORMAT && m.get_info());
        CHECK(m.advance(img, out) == INFUR_OK && out.size() == 1);
    }
    // several contexts from one process: weights replicated by infur_group_weights_broadcast, frames sharded by
    // infur_group_batch_advance; masks identical to a single context's, in frame order.  On a 1-GPU box both
    // contexts sit on device 0 (device-to-device copy; INFUR_FORCE_RCCL=1 routes it through a one-rank communicator)
    {
        const int ndev = infur_device_count();
        infur::Context a(0), b(ndev > 1 ? 1 : 0);
        CHECK(a.ok() && b.ok());
        infur::Model ma(a), mb(b);
        CHECK(ma.control_load(blob_path) == INFUR_OK && !mb.get_info());
        infur::Group solo({&a});
        CHECK(solo.ok() && infur_group_size(solo.get()) == 1);
        CHECK(solo.weights_broadcast(0) == INFUR_OK);  // n_ctx = 1: nothing to do
        std::vector<infur::BgrImage> frames;
        for (int i = 0; i < 5; i++) {
            infur::BgrImage f(160, 96 + 16 * (i % 2));  // ragged batch: two frame sizes
            for (size_t j = 0; j < f.data.size(); j++) f.data[j] = (uint8_t)((j * 7 + i * 53) ^ (j >> 9));
            frames.push_back(std::move(f));
        }
        std::vector<infur::ColorImage> ref, got;
        CHECK(solo.batch_advance(frames, 0.5f, ref) == INFUR_OK);
        infur::Group pair({&a, &b});
        CHECK(pair.ok() && infur_group_size(pair.get()) == 2);
        CHECK(pair.weights_broadcast(1) == INFUR_E_MODEL_NOT_LOADED);  // context 1 has nothing to send yet
        CHECK(pair.weights_broadcast(0) == INFUR_OK);
        auto ib = mb.get_info();
        CHECK(ib && ib->output_names.size() == 2);
        CHECK(pair.batch_advance(frames, 0.5f, got) == INFUR_OK);
        CHECK(got.size() == ref.size());
        for (size_t i = 0; i < ref.size(); i++) {
            CHECK(got[i].width == 80 && got[i].height == ref[i].height && got[i].rgba == ref[i].rgba);
        }
        // the one-shot forms named in SURVEY 8b
        infur_ctx* raw[2] = {a.get(), b.get()};
        CHECK(infur_weights_broadcast(raw, 2) == INFUR_OK);
        std::vector<infur::ColorImage> again;
        CHECK(pair.batch_advance(frames, 0.5f, again) == INFUR_OK && again[4].rgba == ref[4].rgba);
        // fewer frames than contexts: the empty slice is skipped
        std::vector<infur::BgrImage> one(frames.begin(), frames.begin() + 1);
        CHECK(pair.batch_advance(one, 0.5f, again) == INFUR_OK && again.size() == 1 && again[0].rgba == ref[0].rgba);
        std::printf("group ok (rccl %u)\n", infur_group_uses_rccl(pair.get()));
    }
    std::printf("gpu ok\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 2 && !std::strcmp(argv[1], "gpu")) return gpu_tests(argc >= 3 ? argv[2] : "");
    return cpu_tests();
}
