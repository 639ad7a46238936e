/*
 * infur_oracle.h -- CPU restatement of the InFur per-frame segmentation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (infur_amd/, include/)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * PARITY STATUS (see DESIGN.md "Oracle"):
 *   - ColorCode argmax/alpha/palette: pinned by the reference's own KATs
 *     (infur/src/decode_predict.rs:94-116).
 *   - Scale dims + error cases: pinned by infur/src/app.rs:187,199,216 and
 *     infur/src/processing.rs:289-303.
 *   - pre-proc arithmetic: restated from infur/src/predict_onnx.rs:126-137
 *     (plain f32 ops, reproducible bit for bit).
 *   - PARITY UNPINNED: the FCN-ResNet forward (lives in ONNX Runtime, a
 *     third-party C++ library absent from /root/reference), the nearest sampling
 *     rule of fast_image_resize 1.x, the epaint 0.19 premultiply bytes and the
 *     ONNX Resize(linear) coordinate rule are restated from those projects'
 *     published algorithms; no reference test or golden vector pins them and
 *     none of those dependencies can be built or run in this image.
 *     Corroboration that IS possible here (it does not lift the "unpinned" status against ONNX Runtime): the network
 *     agrees to 4e-6 with torchvision's module graph evaluated by torch's own nn modules and exported by PyTorch's
 *     ONNX exporter -- what the zoo's model file is made from (tests/tv_fcn.py, tests/test_onnx_exporter_cpu.py).
 */
#ifndef INFUR_ORACLE_H
#define INFUR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes shared with include/infur_hip.h (same numeric values) */
#define ORACLE_OK 0
#define ORACLE_E_INVALID_SCALE 1 /* ValidScale: factor <= 0, processing.rs:161-163 */
#define ORACLE_E_ZERO_SIZE_IN 2  /* ScaleProcError::ZeroSizeIn, processing.rs:203-204 */
#define ORACLE_E_ZERO_SIZE_OUT 3 /* ScaleProcError::ZeroSizeOut, processing.rs:205-206 */
#define ORACLE_E_SHAPE 4
#define ORACLE_E_MODEL_FORMAT 6

/* ---- Scale (infur/src/processing.rs:142-282) ---- */
int oracle_scale_validate(float factor);
int oracle_scale_out_dims(uint32_t w, uint32_t h, float factor, uint32_t* ow, uint32_t* oh);
/* mode 0 = nearest (reference, processing.rs:189), 1 = bilinear (north_star extension) */
int oracle_scale(const uint8_t* bgr, uint32_t w, uint32_t h, float factor, int mode,
                 uint8_t* out, uint32_t* ow, uint32_t* oh);

/* ---- pre-proc (infur/src/predict_onnx.rs:97-140,167-188) ---- */
/* lut[c*256+v], c in RGB order: ((v*1)/255 - mean_c) * (1/std_c) */
void oracle_preproc_lut(float* lut /*768*/);
/* packed BGR u8 HWC -> planar RGB f32 CHW, torchvision normalisation */
void oracle_pack_normalize(const uint8_t* bgr, uint32_t w, uint32_t h, float* chw);
/* predict_onnx.rs:114-122: what a Uint8-input model is fed -- BGR kept, bytes as they are (as planar f32) */
void oracle_pack_u8(const uint8_t* bgr, uint32_t w, uint32_t h, float* chw);

/* ---- ColorCode (infur/src/decode_predict.rs:9-79) ---- */
void oracle_palette(uint8_t* rgb /*60*/);
/* epaint 0.19 Color32::from_rgba_unmultiplied(r,g,b,a) -> premultiplied [r',g',b',a] */
void oracle_color32_from_rgba_unmultiplied(uint8_t r, uint8_t g, uint8_t b, uint8_t a,
                                           uint8_t out[4]);
/* color_code(klass, alpha), decode_predict.rs:32-36 */
void oracle_color_code(size_t klass, float alpha, uint8_t out[4]);
/* ColorCode::advance: [K,H,W] f32 planar -> H*W premultiplied RGBA8 */
void oracle_colorcode(const float* khw, uint32_t k, uint32_t h, uint32_t w, uint8_t* rgba);
/* same argmax, but reports class index (u8) and alpha byte per pixel (for tie analysis) */
void oracle_argmax(const float* khw, uint32_t k, uint32_t h, uint32_t w, uint8_t* klass,
                   uint8_t* alpha);

/* ---- FCN-ResNet forward (replaces session.run, predict_onnx.rs:138) ---- */
/* weight blob: see infur_amd/weights.py / DESIGN.md "Weight blob".  Returns 0 or error. */
typedef struct oracle_model oracle_model;
int oracle_model_load(const void* blob, size_t len, oracle_model** out);
void oracle_model_free(oracle_model* m);
int oracle_model_num_classes(const oracle_model* m);
/* low-res (output-stride 8) logits dims for an h x w input */
void oracle_model_lowres_dims(uint32_t h, uint32_t w, uint32_t* lh, uint32_t* lw);
/* chw: [3,h,w] f32.  out/aux: [K,h,w] f32 (may be NULL).  out_low/aux_low: [K,lh,lw] (may be NULL) */
int oracle_model_forward(const oracle_model* m, const float* chw, uint32_t h, uint32_t w,
                         float* out, float* aux, float* out_low, float* aux_low);
/* bilinear up-sample [K,ih,iw] -> [K,oh,ow], ONNX Resize(linear, pytorch_half_pixel) */
void oracle_upsample_bilinear(const float* in, uint32_t k, uint32_t ih, uint32_t iw, float* out,
                              uint32_t oh, uint32_t ow);

/* whole path: BGR frame -> (scale) -> pack -> forward -> colorcode.  rgba: oh*ow*4 */
int oracle_frame_advance(const oracle_model* m, const uint8_t* bgr, uint32_t w, uint32_t h,
                         float factor, int scale_mode, uint8_t* rgba, uint32_t* ow, uint32_t* oh);

void oracle_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
