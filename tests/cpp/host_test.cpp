// Exercises include/infur_processor.hpp (the C++ mirror of the reference's Processor trait).
//   host_test cpu   -- host-only logic, runs without a GPU (context creation must fail loudly)
//   host_test gpu   -- the reference's own unit tests re-expressed over the HIP path:
//                      processing.rs:289-303, decode_predict.rs:100-116, predict_onnx.rs:371-381
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

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
    std::printf("gpu ok\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 2 && !std::strcmp(argv[1], "gpu")) return gpu_tests(argc >= 3 ? argv[2] : "");
    return cpu_tests();
}
