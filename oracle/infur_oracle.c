/*
 * infur_oracle.c -- plain-C CPU restatement of the InFur per-frame segmentation path.
 * TEST INFRASTRUCTURE ONLY (see infur_oracle.h for the parity status of each piece).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off matters: the elementwise pieces must round exactly like the
 * reference's scalar Rust f32 code (no fused multiply-add).
 *
 * All citations are path:line relative to /root/reference/.
 */
#include "infur_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------- */
/* Scale  (infur/src/processing.rs:142-282)                                   */
/* ------------------------------------------------------------------------- */

/* ValidScale::try_from, processing.rs:158-168: only `value <= 0.0` is rejected
 * (NaN passes, exactly like the Rust comparison). */
int oracle_scale_validate(float factor) {
    if (factor <= 0.0f) return ORACLE_E_INVALID_SCALE;
    return ORACLE_OK;
}

/* Rust `f32 as u32`: saturating, NaN -> 0, truncation toward zero. */
static uint32_t f32_as_u32(float v) {
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 4294967296.0f) return 4294967295u;
    return (uint32_t)v;
}

/* processing.rs:246-256: ZeroSizeIn is checked first (width, then height), then the
 * output dims `(w as f32 * factor) as u32` and ZeroSizeOut.  The unit-scale shortcut
 * (processing.rs:238-242) returns a clone before any size check. */
int oracle_scale_out_dims(uint32_t w, uint32_t h, float factor, uint32_t* ow, uint32_t* oh) {
    if (factor == 1.0f) {
        *ow = w;
        *oh = h;
        return ORACLE_OK;
    }
    if (w == 0 || h == 0) return ORACLE_E_ZERO_SIZE_IN;
    uint32_t nw = f32_as_u32((float)w * factor);
    uint32_t nh = f32_as_u32((float)h * factor);
    if (nw == 0 || nh == 0) return ORACLE_E_ZERO_SIZE_OUT;
    *ow = nw;
    *oh = nh;
    return ORACLE_OK;
}

/* fast_image_resize 1.x ResizeAlg::Nearest (third-party crate, `fast_image_resize = "1"`
 * in infur/Cargo.toml; call site processing.rs:189,278).  Restated from the crate's
 * published algorithm, UNVERIFIED here:
 *   scale = src_len / dst_len (f64); src = trunc(0.5*scale + scale*dst), clamped to src_len-1. */
static void nearest_table(uint32_t src, uint32_t dst, uint32_t* tab) {
    double scale = (double)src / (double)dst;
    double start = scale * 0.5;
    for (uint32_t i = 0; i < dst; i++) {
        double p = start + scale * (double)i;
        uint32_t s = (uint32_t)p;
        if (s > src - 1) s = src - 1;
        tab[i] = s;
    }
}

/* Bilinear mode is the north_star's extension (the reference only has a todo for it,
 * processing.rs:224).  Definition used by both oracle and HIP path: half-pixel centres,
 * src = (dst+0.5)*(src_len/dst_len) - 0.5 in f32, clamp to [0, src_len-1], 2x2 taps,
 * f32 lerp x then y, round half up, no antialias. */
static void bilinear_table(uint32_t src, uint32_t dst, uint32_t* i0, uint32_t* i1, float* frac) {
    float scale = (float)src / (float)dst;
    for (uint32_t i = 0; i < dst; i++) {
        float p = ((float)i + 0.5f) * scale - 0.5f;
        if (p < 0.0f) p = 0.0f;
        float lim = (float)(src - 1);
        if (p > lim) p = lim;
        uint32_t a = (uint32_t)p;
        uint32_t b = a + 1 < src ? a + 1 : src - 1;
        i0[i] = a;
        i1[i] = b;
        frac[i] = p - (float)a;
    }
}

int oracle_scale(const uint8_t* bgr, uint32_t w, uint32_t h, float factor, int mode, uint8_t* out,
                 uint32_t* ow, uint32_t* oh) {
    int rc = oracle_scale_validate(factor);
    if (rc) return rc;
    uint32_t nw, nh;
    rc = oracle_scale_out_dims(w, h, factor, &nw, &nh);
    if (rc) return rc;
    *ow = nw;
    *oh = nh;
    if (factor == 1.0f) { /* processing.rs:238-242 */
        memcpy(out, bgr, (size_t)w * h * 3);
        return ORACLE_OK;
    }
    if (mode == 0) {
        uint32_t* xt = (uint32_t*)malloc(sizeof(uint32_t) * nw);
        uint32_t* yt = (uint32_t*)malloc(sizeof(uint32_t) * nh);
        nearest_table(w, nw, xt);
        nearest_table(h, nh, yt);
        for (uint32_t y = 0; y < nh; y++) {
            const uint8_t* srow = bgr + (size_t)yt[y] * w * 3;
            uint8_t* drow = out + (size_t)y * nw * 3;
            for (uint32_t x = 0; x < nw; x++) {
                const uint8_t* s = srow + (size_t)xt[x] * 3;
                drow[3 * x + 0] = s[0];
                drow[3 * x + 1] = s[1];
                drow[3 * x + 2] = s[2];
            }
        }
        free(xt);
        free(yt);
    } else {
        uint32_t *x0 = malloc(4 * nw), *x1 = malloc(4 * nw), *y0 = malloc(4 * nh),
                 *y1 = malloc(4 * nh);
        float *fx = malloc(4 * nw), *fy = malloc(4 * nh);
        bilinear_table(w, nw, x0, x1, fx);
        bilinear_table(h, nh, y0, y1, fy);
        for (uint32_t y = 0; y < nh; y++) {
            const uint8_t* r0 = bgr + (size_t)y0[y] * w * 3;
            const uint8_t* r1 = bgr + (size_t)y1[y] * w * 3;
            for (uint32_t x = 0; x < nw; x++) {
                for (int c = 0; c < 3; c++) {
                    float p00 = r0[3 * x0[x] + c], p01 = r0[3 * x1[x] + c];
                    float p10 = r1[3 * x0[x] + c], p11 = r1[3 * x1[x] + c];
                    float top = p00 + (p01 - p00) * fx[x];
                    float bot = p10 + (p11 - p10) * fx[x];
                    float v = top + (bot - top) * fy[y];
                    float r = floorf(v + 0.5f);
                    if (r < 0.0f) r = 0.0f;
                    if (r > 255.0f) r = 255.0f;
                    out[((size_t)y * nw + x) * 3 + c] = (uint8_t)r;
                }
            }
        }
        free(x0); free(x1); free(y0); free(y1); free(fx); free(fy);
    }
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------- */
/* pre-proc  (infur/src/predict_onnx.rs:97-140)                               */
/* ------------------------------------------------------------------------- */

/* ColorNorm::new_torchvision_rgb, predict_onnx.rs:175-180 */
static const float TV_MEAN[3] = {0.485f, 0.456f, 0.406f};
static const float TV_STD[3] = {0.229f, 0.224f, 0.225f};

/* predict_onnx.rs:128: `f32::from(v) * 1f32 / 255f32` (a true division), then
 * :131-136: `lane -= mean; lane *= 1.0/std` (reciprocal computed once, :132). */
void oracle_preproc_lut(float* lut) {
    for (int c = 0; c < 3; c++) {
        volatile float std1 = 1.0f / TV_STD[c];
        for (int v = 0; v < 256; v++) {
            volatile float x = ((float)v * 1.0f) / 255.0f;
            volatile float d = x - TV_MEAN[c];
            lut[c * 256 + v] = d * std1;
        }
    }
}

/* predict_onnx.rs:103-112: invert_axis(3) turns B,G,R into R,G,B; permuted_axes
 * [0,3,1,2] makes it channel-planar.  Output channel 0 = R = input byte 2. */
void oracle_pack_normalize(const uint8_t* bgr, uint32_t w, uint32_t h, float* chw) {
    float lut[768];
    oracle_preproc_lut(lut);
    size_t hw = (size_t)w * h;
    for (int c = 0; c < 3; c++) {
        const float* l = lut + c * 256;
        float* dst = chw + c * hw;
        for (size_t i = 0; i < hw; i++) dst[i] = l[bgr[3 * i + (2 - c)]];
    }
}

/* predict_onnx.rs:114-122, the ColorRange::Uint8 arm (a model that declares a Uint8 image input, :255): the session
 * gets the frame's bytes themselves -- color_seq stays BGR (:296-301), no scaling, no normalisation.  What such a
 * model's first convolution then sees, as the channel-planar f32 tensor this oracle's forward takes:
 * plane c = input byte c (0 = B), value = the byte. */
void oracle_pack_u8(const uint8_t* bgr, uint32_t w, uint32_t h, float* chw) {
    size_t hw = (size_t)w * h;
    for (int c = 0; c < 3; c++) {
        float* dst = chw + c * hw;
        for (size_t i = 0; i < hw; i++) dst[i] = (float)bgr[3 * i + c];
    }
}

/* ------------------------------------------------------------------------- */
/* ColorCode  (infur/src/decode_predict.rs:9-79)                              */
/* ------------------------------------------------------------------------- */

/* COLORS_PALETTE, decode_predict.rs:9-30 (data table; used AS RGB at :34-35) */
static const uint8_t PALETTE[20][3] = {
    {75, 180, 60},   {75, 25, 230},   {25, 225, 255},  {200, 130, 0},   {48, 130, 245},
    {240, 240, 70},  {230, 50, 240},  {60, 245, 210},  {180, 30, 145},  {190, 190, 250},
    {128, 128, 0},   {255, 190, 230}, {40, 110, 170},  {200, 250, 255}, {0, 0, 128},
    {195, 255, 170}, {0, 128, 128},   {180, 215, 255}, {128, 0, 0},     {128, 128, 128},
};

void oracle_palette(uint8_t* rgb) { memcpy(rgb, PALETTE, 60); }

/* epaint 0.19 colour helpers (third-party crate `eframe = "0.19"`, infur/Cargo.toml:18;
 * call site decode_predict.rs:35).  Restated from the crate's published source,
 * UNVERIFIED here (the reference tests only compare this function with itself). */
static float linear_f32_from_gamma_u8(uint8_t s) {
    if (s <= 10) return (float)s / 3294.6f;
    return powf(((float)s + 14.025f) / 269.025f, 2.4f);
}
static uint8_t fast_round_u8(float r) {
    float f = floorf(r + 0.5f);
    if (!(f == f) || f <= 0.0f) return 0;
    if (f >= 255.0f) return 255;
    return (uint8_t)f;
}
static uint8_t gamma_u8_from_linear_f32(float l) {
    if (l <= 0.0f) return 0;
    if (l <= 0.0031308f) return fast_round_u8(3294.6f * l);
    if (l <= 1.0f) return fast_round_u8(269.025f * powf(l, 1.0f / 2.4f) - 14.025f);
    return 255;
}

void oracle_color32_from_rgba_unmultiplied(uint8_t r, uint8_t g, uint8_t b, uint8_t a,
                                           uint8_t out[4]) {
    if (a == 255) {
        out[0] = r; out[1] = g; out[2] = b; out[3] = 255;
    } else if (a == 0) {
        out[0] = out[1] = out[2] = out[3] = 0;
    } else {
        float al = (float)a / 255.0f;
        out[0] = gamma_u8_from_linear_f32(linear_f32_from_gamma_u8(r) * al);
        out[1] = gamma_u8_from_linear_f32(linear_f32_from_gamma_u8(g) * al);
        out[2] = gamma_u8_from_linear_f32(linear_f32_from_gamma_u8(b) * al);
        out[3] = a;
    }
}

/* Rust `f32 as u8`: saturating, NaN -> 0, truncation toward zero. */
static uint8_t f32_as_u8(float v) {
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

/* color_code, decode_predict.rs:32-36 */
void oracle_color_code(size_t klass, float alpha, uint8_t out[4]) {
    const uint8_t* c = PALETTE[klass % 20];
    oracle_color32_from_rgba_unmultiplied(c[0], c[1], c[2], f32_as_u8(alpha * 255.0f), out);
}

/* ColorCode::advance inner loop, decode_predict.rs:67-78: per pixel in raster order,
 * k_max=0, c_max=0.0, strict `>` over the K planes (stride H*W). */
void oracle_argmax(const float* khw, uint32_t k, uint32_t h, uint32_t w, uint8_t* klass,
                   uint8_t* alpha) {
    size_t hw = (size_t)h * w;
    for (size_t p = 0; p < hw; p++) {
        size_t k_max = 0;
        float c_max = 0.0f;
        for (uint32_t i = 0; i < k; i++) {
            float c = khw[(size_t)i * hw + p];
            if (c > c_max) {
                k_max = i;
                c_max = c;
            }
        }
        klass[p] = (uint8_t)k_max;
        alpha[p] = f32_as_u8(c_max * 255.0f);
    }
}

void oracle_colorcode(const float* khw, uint32_t k, uint32_t h, uint32_t w, uint8_t* rgba) {
    size_t hw = (size_t)h * w;
    for (size_t p = 0; p < hw; p++) {
        size_t k_max = 0;
        float c_max = 0.0f;
        for (uint32_t i = 0; i < k; i++) {
            float c = khw[(size_t)i * hw + p];
            if (c > c_max) {
                k_max = i;
                c_max = c;
            }
        }
        oracle_color_code(k_max, c_max, rgba + 4 * p);
    }
}

/* ------------------------------------------------------------------------- */
/* FCN-ResNet forward (replaces session.run at predict_onnx.rs:138)           */
/*                                                                           */
/* The arithmetic lives in ONNX Runtime + the fcn-resnet50-12 model file     */
/* (infur-test-gen/build.rs:88-93), neither of which exists here.  This is a  */
/* restatement of torchvision's fcn_resnet50 / fcn_resnet101 graph            */
/* (replace_stride_with_dilation=[False,True,True], BN folded into conv)      */
/* in NCHW planar layout, the layout of the reference's tensors               */
/* (predict_onnx.rs:109-112,378-380).                                         */
/* ------------------------------------------------------------------------- */

#define BLOB_MAGIC "INFURW01"
#define BLOB_HDR 32
#define BLOB_ENTRY 80

typedef struct {
    char name[40];
    uint32_t cout, cin, kh, kw;
    const float* w; /* OIHW */
    const float* b; /* [cout] */
} oconv;

struct oracle_model {
    uint32_t depth, num_classes, has_aux, n_convs;
    uint32_t input_u8; /* blob header offset 24: 1 = the model declares a Uint8 image input */
    oconv* convs;
    void* blob_copy;
};

static uint32_t rd_u32(const uint8_t* p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
static uint64_t rd_u64(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}

static int layer_blocks(uint32_t depth, int out[4]) {
    if (depth == 50) { out[0] = 3; out[1] = 4; out[2] = 6; out[3] = 3; return 0; }
    if (depth == 101) { out[0] = 3; out[1] = 4; out[2] = 23; out[3] = 3; return 0; }
    return -1;
}

int oracle_model_load(const void* blob, size_t len, oracle_model** out) {
    const uint8_t* p = (const uint8_t*)blob;
    if (len < BLOB_HDR || memcmp(p, BLOB_MAGIC, 8) != 0) return ORACLE_E_MODEL_FORMAT;
    oracle_model* m = (oracle_model*)calloc(1, sizeof(*m));
    m->depth = rd_u32(p + 8);
    m->num_classes = rd_u32(p + 12);
    m->has_aux = rd_u32(p + 16);
    m->n_convs = rd_u32(p + 20);
    m->input_u8 = rd_u32(p + 24);
    int lb[4];
    if (layer_blocks(m->depth, lb) != 0 || (size_t)BLOB_HDR + (size_t)m->n_convs * BLOB_ENTRY > len) {
        free(m);
        return ORACLE_E_MODEL_FORMAT;
    }
    uint32_t expect = 1 + 3 * (lb[0] + lb[1] + lb[2] + lb[3]) + 4 + 2 + (m->has_aux ? 2 : 0);
    if (m->n_convs != expect) {
        free(m);
        return ORACLE_E_MODEL_FORMAT;
    }
    m->blob_copy = malloc(len);
    memcpy(m->blob_copy, blob, len);
    p = (const uint8_t*)m->blob_copy;
    m->convs = (oconv*)calloc(m->n_convs, sizeof(oconv));
    for (uint32_t i = 0; i < m->n_convs; i++) {
        const uint8_t* e = p + BLOB_HDR + (size_t)i * BLOB_ENTRY;
        oconv* c = &m->convs[i];
        memcpy(c->name, e, 40);
        c->name[39] = 0;
        c->cout = rd_u32(e + 40);
        c->cin = rd_u32(e + 44);
        c->kh = rd_u32(e + 48);
        c->kw = rd_u32(e + 52);
        uint64_t wo = rd_u64(e + 56), bo = rd_u64(e + 64);
        size_t wn = (size_t)c->cout * c->cin * c->kh * c->kw * 4;
        if (wo + wn > len || bo + (size_t)c->cout * 4 > len) {
            oracle_model_free(m);
            return ORACLE_E_MODEL_FORMAT;
        }
        c->w = (const float*)(p + wo);
        c->b = (const float*)(p + bo);
    }
    *out = m;
    return ORACLE_OK;
}

void oracle_model_free(oracle_model* m) {
    if (!m) return;
    free(m->convs);
    free(m->blob_copy);
    free(m);
}

int oracle_model_num_classes(const oracle_model* m) { return (int)m->num_classes; }

static uint32_t conv_out(uint32_t in, uint32_t k, uint32_t s, uint32_t p, uint32_t d) {
    return (in + 2 * p - d * (k - 1) - 1) / s + 1;
}

void oracle_model_lowres_dims(uint32_t h, uint32_t w, uint32_t* lh, uint32_t* lw) {
    uint32_t a = conv_out(h, 7, 2, 3, 1), b = conv_out(w, 7, 2, 3, 1); /* stem */
    a = conv_out(a, 3, 2, 1, 1); b = conv_out(b, 3, 2, 1, 1);          /* maxpool */
    a = conv_out(a, 3, 2, 1, 1); b = conv_out(b, 3, 2, 1, 1);          /* layer2 stride */
    *lh = a;
    *lw = b;
}

/* direct convolution, NCHW, f32 accumulate in (ci, ky, kx) order, bias added last,
 * optional residual add then ReLU. */
static void conv2d(const float* in, uint32_t cin, uint32_t h, uint32_t w, const oconv* c,
                   uint32_t stride, uint32_t pad, uint32_t dil, const float* residual, int relu,
                   float* out, uint32_t oh, uint32_t ow) {
    const uint32_t kh = c->kh, kw = c->kw;
    const size_t ohw = (size_t)oh * ow;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t co = 0; co < (int64_t)c->cout; co++) {
        float* o = out + (size_t)co * ohw;
        for (size_t i = 0; i < ohw; i++) o[i] = 0.0f;
        for (uint32_t ci = 0; ci < cin; ci++) {
            const float* ip = in + (size_t)ci * h * w;
            const float* wp = c->w + (((size_t)co * cin + ci) * kh) * kw;
            for (uint32_t ky = 0; ky < kh; ky++) {
                for (uint32_t kx = 0; kx < kw; kx++) {
                    const float wv = wp[ky * kw + kx];
                    /* valid ox range: 0 <= ox*stride - pad + kx*dil < w */
                    int64_t off_x = (int64_t)kx * dil - pad;
                    int64_t ox0 = off_x < 0 ? (-off_x + stride - 1) / stride : 0;
                    int64_t num = (int64_t)w - 1 - off_x;
                    int64_t ox1 = num < 0 ? -1 : num / stride; /* inclusive; floor, not C's truncation */
                    if (ox1 >= (int64_t)ow) ox1 = ow - 1;
                    for (uint32_t oy = 0; oy < oh; oy++) {
                        int64_t iy = (int64_t)oy * stride - pad + (int64_t)ky * dil;
                        if (iy < 0 || iy >= (int64_t)h) continue;
                        const float* irow = ip + (size_t)iy * w + off_x;
                        float* orow = o + (size_t)oy * ow;
                        if (stride == 1) {
                            for (int64_t ox = ox0; ox <= ox1; ox++) orow[ox] += wv * irow[ox];
                        } else {
                            for (int64_t ox = ox0; ox <= ox1; ox++)
                                orow[ox] += wv * irow[ox * stride];
                        }
                    }
                }
            }
        }
        const float bv = c->b[co];
        const float* r = residual ? residual + (size_t)co * ohw : NULL;
        for (size_t i = 0; i < ohw; i++) {
            float v = o[i] + bv;
            if (r) v = v + r[i];
            if (relu && !(v > 0.0f)) v = 0.0f;
            o[i] = v;
        }
    }
}

/* MaxPool 3x3 stride 2 pad 1 (padding never wins: -inf) */
static void maxpool3x3s2(const float* in, uint32_t c, uint32_t h, uint32_t w, float* out,
                         uint32_t oh, uint32_t ow) {
#pragma omp parallel for
    for (int64_t ch = 0; ch < (int64_t)c; ch++) {
        const float* ip = in + (size_t)ch * h * w;
        float* op = out + (size_t)ch * oh * ow;
        for (uint32_t oy = 0; oy < oh; oy++)
            for (uint32_t ox = 0; ox < ow; ox++) {
                float m = -INFINITY;
                for (int ky = 0; ky < 3; ky++) {
                    int64_t iy = (int64_t)oy * 2 - 1 + ky;
                    if (iy < 0 || iy >= (int64_t)h) continue;
                    for (int kx = 0; kx < 3; kx++) {
                        int64_t ix = (int64_t)ox * 2 - 1 + kx;
                        if (ix < 0 || ix >= (int64_t)w) continue;
                        float v = ip[(size_t)iy * w + ix];
                        if (v > m) m = v;
                    }
                }
                op[(size_t)oy * ow + ox] = m;
            }
    }
}

/* ONNX Resize(mode=linear, coordinate_transformation_mode=pytorch_half_pixel) as ONNX
 * Runtime's CPU UpsampleBilinear evaluates it -- what torchvision's
 * F.interpolate(bilinear, align_corners=False) exports to.  Restated from ORT's published
 * algorithm, UNVERIFIED here:
 *   scale = out_len / in_len (f32);  src = out_len > 1 ? (dst + 0.5)/scale - 0.5 : 0;
 *   src clamped to [0, in_len-1]; i1 = trunc(src), i2 = min(i1+1, in_len-1);
 *   d1 = |src - i1|, d2 = |src - i2|, both 0.5 when i1 == i2;
 *   out = dx2*dy2*X11 + dx1*dy2*X21 + dx2*dy1*X12 + dx1*dy1*X22   (left to right, no FMA)
 * with X11=in[y1][x1], X21=in[y1][x2], X12=in[y2][x1], X22=in[y2][x2]. */
typedef struct {
    uint32_t i1, i2;
    float d1, d2;
} lerp_tab;

static void upsample_table(uint32_t in_len, uint32_t out_len, lerp_tab* t) {
    float scale = (float)out_len / (float)in_len;
    for (uint32_t i = 0; i < out_len; i++) {
        float src = out_len > 1 ? ((float)i + 0.5f) / scale - 0.5f : 0.0f;
        if (src < 0.0f) src = 0.0f;
        float lim = (float)(in_len - 1);
        if (src > lim) src = lim;
        uint32_t a = (uint32_t)src;
        if (a > in_len - 1) a = in_len - 1;
        uint32_t b = a + 1 < in_len ? a + 1 : in_len - 1;
        t[i].i1 = a;
        t[i].i2 = b;
        if (a == b) {
            t[i].d1 = 0.5f;
            t[i].d2 = 0.5f;
        } else {
            t[i].d1 = fabsf(src - (float)a);
            t[i].d2 = fabsf(src - (float)b);
        }
    }
}

void oracle_upsample_bilinear(const float* in, uint32_t k, uint32_t ih, uint32_t iw, float* out,
                              uint32_t oh, uint32_t ow) {
    lerp_tab* ty = (lerp_tab*)malloc(sizeof(lerp_tab) * oh);
    lerp_tab* tx = (lerp_tab*)malloc(sizeof(lerp_tab) * ow);
    upsample_table(ih, oh, ty);
    upsample_table(iw, ow, tx);
#pragma omp parallel for
    for (int64_t c = 0; c < (int64_t)k; c++) {
        const float* ip = in + (size_t)c * ih * iw;
        float* op = out + (size_t)c * oh * ow;
        for (uint32_t y = 0; y < oh; y++) {
            const float* r1 = ip + (size_t)ty[y].i1 * iw;
            const float* r2 = ip + (size_t)ty[y].i2 * iw;
            const float dy1 = ty[y].d1, dy2 = ty[y].d2;
            for (uint32_t x = 0; x < ow; x++) {
                const float dx1 = tx[x].d1, dx2 = tx[x].d2;
                const float X11 = r1[tx[x].i1], X21 = r1[tx[x].i2];
                const float X12 = r2[tx[x].i1], X22 = r2[tx[x].i2];
                float v = dx2 * dy2 * X11;
                v = v + dx1 * dy2 * X21;
                v = v + dx2 * dy1 * X12;
                v = v + dx1 * dy1 * X22;
                op[(size_t)y * ow + x] = v;
            }
        }
    }
    free(ty);
    free(tx);
}

typedef struct {
    float* p;
    uint32_t c, h, w;
} otensor;

static otensor talloc(uint32_t c, uint32_t h, uint32_t w) {
    otensor t;
    t.c = c; t.h = h; t.w = w;
    t.p = (float*)malloc(sizeof(float) * (size_t)c * h * w);
    return t;
}

/* conv helper that allocates the output */
static otensor conv_layer(const otensor* in, const oconv* c, uint32_t stride, uint32_t pad,
                          uint32_t dil, const otensor* residual, int relu) {
    uint32_t oh = conv_out(in->h, c->kh, stride, pad, dil);
    uint32_t ow = conv_out(in->w, c->kw, stride, pad, dil);
    otensor o = talloc(c->cout, oh, ow);
    conv2d(in->p, in->c, in->h, in->w, c, stride, pad, dil, residual ? residual->p : NULL, relu,
           o.p, oh, ow);
    return o;
}

int oracle_model_forward(const oracle_model* m, const float* chw, uint32_t h, uint32_t w,
                         float* out, float* aux, float* out_low, float* aux_low) {
    if (h == 0 || w == 0) return ORACLE_E_SHAPE;
    int lb[4] = {0, 0, 0, 0};
    layer_blocks(m->depth, lb);
    uint32_t ci = 0; /* conv cursor, graph order == blob order */
    otensor x;
    x.p = (float*)chw; x.c = 3; x.h = h; x.w = w;

    /* stem: conv 7x7/2 pad 3 + ReLU, maxpool 3x3/2 pad 1 */
    otensor s = conv_layer(&x, &m->convs[ci++], 2, 3, 1, NULL, 1);
    otensor cur = talloc(s.c, conv_out(s.h, 3, 2, 1, 1), conv_out(s.w, 3, 2, 1, 1));
    maxpool3x3s2(s.p, s.c, s.h, s.w, cur.p, cur.h, cur.w);
    free(s.p);

    /* torchvision ResNet._make_layer with replace_stride_with_dilation=[F,T,T] */
    uint32_t dilation = 1;
    otensor layer3_out;
    layer3_out.p = NULL;
    for (int L = 0; L < 4; L++) {
        uint32_t stride = L == 0 ? 1 : 2;
        int dilate = L >= 2;
        uint32_t prev_dil = dilation;
        if (dilate) {
            dilation *= stride;
            stride = 1;
        }
        for (int B = 0; B < lb[L]; B++) {
            uint32_t bs = B == 0 ? stride : 1;
            uint32_t bd = B == 0 ? prev_dil : dilation;
            const oconv* c1 = &m->convs[ci++];
            const oconv* c2 = &m->convs[ci++];
            const oconv* c3 = &m->convs[ci++];
            const oconv* ds = B == 0 ? &m->convs[ci++] : NULL;
            /* Bottleneck v1.5: stride on the 3x3 */
            otensor t1 = conv_layer(&cur, c1, 1, 0, 1, NULL, 1);
            otensor t2 = conv_layer(&t1, c2, bs, bd, bd, NULL, 1);
            free(t1.p);
            otensor idt = cur;
            if (ds) idt = conv_layer(&cur, ds, bs, 0, 1, NULL, 0);
            otensor t3 = conv_layer(&t2, c3, 1, 0, 1, &idt, 1);
            free(t2.p);
            if (ds) free(idt.p);
            free(cur.p);
            cur = t3;
        }
        if (L == 2 && m->has_aux) {
            layer3_out = talloc(cur.c, cur.h, cur.w);
            memcpy(layer3_out.p, cur.p, sizeof(float) * (size_t)cur.c * cur.h * cur.w);
        }
    }

    /* FCNHead: 3x3 pad 1 + ReLU (dropout = identity at inference), 1x1 + bias */
    otensor h1 = conv_layer(&cur, &m->convs[ci++], 1, 1, 1, NULL, 1);
    otensor lo = conv_layer(&h1, &m->convs[ci++], 1, 0, 1, NULL, 0);
    free(h1.p);
    free(cur.p);
    if (out_low) memcpy(out_low, lo.p, sizeof(float) * (size_t)lo.c * lo.h * lo.w);
    if (out) oracle_upsample_bilinear(lo.p, lo.c, lo.h, lo.w, out, h, w);
    free(lo.p);

    if (m->has_aux) {
        otensor a1 = conv_layer(&layer3_out, &m->convs[ci++], 1, 1, 1, NULL, 1);
        otensor al = conv_layer(&a1, &m->convs[ci++], 1, 0, 1, NULL, 0);
        free(a1.p);
        free(layer3_out.p);
        if (aux_low) memcpy(aux_low, al.p, sizeof(float) * (size_t)al.c * al.h * al.w);
        if (aux) oracle_upsample_bilinear(al.p, al.c, al.h, al.w, aux, h, w);
        free(al.p);
    }
    return ORACLE_OK;
}

/* infur/src/app.rs:107-153: scale -> model -> decode out[0] only (:116) */
int oracle_frame_advance(const oracle_model* m, const uint8_t* bgr, uint32_t w, uint32_t h,
                         float factor, int scale_mode, uint8_t* rgba, uint32_t* ow, uint32_t* oh) {
    int rc = oracle_scale_validate(factor);
    if (rc) return rc;
    uint32_t nw, nh;
    rc = oracle_scale_out_dims(w, h, factor, &nw, &nh);
    if (rc) return rc;
    uint8_t* scaled = (uint8_t*)malloc((size_t)nw * nh * 3 + 1);
    rc = oracle_scale(bgr, w, h, factor, scale_mode, scaled, &nw, &nh);
    if (rc) {
        free(scaled);
        return rc;
    }
    float* chw = (float*)malloc(sizeof(float) * 3 * (size_t)nw * nh);
    if (m->input_u8)
        oracle_pack_u8(scaled, nw, nh, chw);
    else
        oracle_pack_normalize(scaled, nw, nh, chw);
    float* logits = (float*)malloc(sizeof(float) * (size_t)m->num_classes * nw * nh);
    rc = oracle_model_forward(m, chw, nh, nw, logits, NULL, NULL, NULL);
    if (!rc) oracle_colorcode(logits, m->num_classes, nh, nw, rgba);
    *ow = nw;
    *oh = nh;
    free(scaled);
    free(chw);
    free(logits);
    return rc;
}
