// infur_quant_model.cpp -- quantised models (INFURQ01: the QOperator int8 form of FCN-ResNet the reference's own tests load,
// infur/src/predict_onnx.rs:357-381, infur-test-gen/build.rs:88-93): load (weights repacked for the i8 MFMA, zero points folded into
// the biases, the pixel-pair view of layer1) and forward (QLinearConv / QLinearAdd arithmetic in the kernels' epilogues, bit-exact
// against oracle/infur_qoracle.py).  Split out of infur_capi.cpp in round 5.
#include "../../include/infur_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "blob_dir.h"
#include "infur_rt.h"

using namespace infur;

namespace infur {

// ---- quantised models (INFURQ01) ----
// d_blob resident on the device.  Header and directory are checked by blob_dir.h (host-only); weights are repacked to
// OHWI with the channel axes padded to the K step of the i8 GEMM (128 bytes: the 64-channel tensors of the stem and layer1
// travel as 128 channels, the upper half zero), the operator bias is folded with the (128 - x_zp) * sum w term of the kernel's
// signed operands, and the requantisation multipliers are computed as ONNX Runtime computes them (f32: x_s * w_s[o] / y_s).
inline int q_cpad(int c, bool padded) { return padded && c < 128 ? 128 : c; }
// bytes of a layer's repacked weights: s8 OHWI (padded); the stem: one dword (r, g, b, 0) per tap and channel
// layer1's convs (64-channel tensors on one side or both) also get the pixel-pair arrangement of their weights
inline bool q_pair_layer(const ConvLayer& L) {
    return L.name.compare(0, 16, "backbone.layer1.") == 0 && L.stride == 1 && L.dil == 1 && (L.k == 1 || (L.k == 3 && L.pad == 1)) &&
           (L.cin == 64 || L.cout == 64) && (L.cin % 64) == 0 && (L.cout % 64) == 0;
}
inline size_t q_wbytes(const ConvLayer& L) {
    return L.role == 's' ? (size_t)L.cout * L.k * L.k * 4 : (size_t)L.cout_p * L.k * L.k * L.cin_p + 64;
}

int32_t model_load_q_dev(infur_ctx* c, const void* d_blob, size_t len) {
    if (len < kBlobHdr) return fail(c, INFUR_E_MODEL_FORMAT, "weight blob too short (%zu bytes)", len);
    uint8_t hdr[kBlobHdr];
    HIPCHK(c, hipMemcpyAsync(hdr, d_blob, kBlobHdr, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    BlobHeader bh;
    uint32_t n_adds = 0;
    std::vector<ConvSpec> spec;
    std::string perr;
    if (!qblob_parse_header(hdr, len, &bh, &n_adds, &spec, &perr)) return fail(c, INFUR_E_MODEL_FORMAT, "%s", perr.c_str());
    const uint32_t n = bh.n_convs;
    std::vector<uint8_t> table((size_t)n * kQEntry + (size_t)n_adds * kQAdd);
    HIPCHK(c, hipMemcpyAsync(table.data(), (const uint8_t*)d_blob + kBlobHdr, table.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<QBlobConv> qc;
    std::vector<QBlobAdd> qa;
    if (!qblob_parse_directory(table.data(), len, spec, n_adds, &qc, &qa, &perr)) return fail(c, INFUR_E_MODEL_FORMAT, "%s", perr.c_str());
    std::vector<ConvLayer> g = build_graph(bh.depth, bh.num_classes, bh.aux);
    // a block's QLinearAdd takes conv3's output as A: the kernel requantises with conv3's own (y_scale, y_zp) and adds in place
    {
        uint32_t blk = 0;
        for (uint32_t i = 0; i < n; i++)
            if (g[i].role == '3') {
                if (blk >= n_adds || qa[blk].a_zp != qc[i].y_zp || qa[blk].a_scale != qc[i].y_scale)
                    return fail(c, INFUR_E_MODEL_FORMAT, "residual sum %u does not take '%s' as its first input (scale / zero point differ)", blk, g[i].name.c_str());
                blk++;
            }
    }
    size_t total = 1024;  // the quantisation table of the image
    constexpr size_t kQStemW = 147 * 64 * 4, kQStemLut = 768 * 4, kQStemBias = 64 * 4;  // operands of the fused stem (launch_stem_pool_q)
    total += align_up(kQStemW, 256) + align_up(kQStemLut, 256) + align_up(kQStemBias, 256);
    for (uint32_t i = 0; i < n; i++) {
        ConvLayer& L = g[i];
        const bool logits = L.role == 'c';
        L.cin_p = L.role == 's' ? L.cin : q_cpad(L.cin, true);
        L.cout_p = L.role == 's' ? L.cout : q_cpad(L.cout, !logits);
        total += align_up(q_wbytes(L), 256) + 2 * align_up((size_t)L.cout_p * 4, 256);
        if (q_pair_layer(L)) {  // the pixel-pair form of layer1 (forward_q), beside the padded one (odd widths, kept activations)
            L.cin2 = 2 * L.cin;
            L.cout2 = 2 * L.cout;
            total += align_up((size_t)L.cout2 * L.k * L.k * L.cin2 + 64, 256) + 2 * align_up((size_t)L.cout2 * 4, 256);
        }
    }
    struct DevMem {
        void* p = nullptr;
        ~DevMem() { if (p) (void)hipFree(p); }
    } arena, tmp;
    HIPCHK(c, hipMalloc(&arena.p, total));
    HIPCHK(c, hipMemsetAsync(arena.p, 0, total, c->stream));
    size_t max_c = 0;
    for (const ConvLayer& L : g) max_c = std::max(max_c, (size_t)L.cout_p);
    HIPCHK(c, hipMalloc(&tmp.p, max_c * 4));  // row sums of the layer being repacked
    uint8_t* const base = (uint8_t*)arena.p;
    size_t off = 0;
    // image quantisation table: QuantizeLinear of the reference's normalised value of every byte (predict_onnx.rs:126-137)
    std::vector<uint8_t> ql(768);
    {
        std::vector<float> pre(768);
        build_pre_lut(pre.data());
        const volatile float xs = qc[0].x_scale;
        for (int i = 0; i < 768; i++) {
            volatile float t = pre[i] / xs;  // (one f32 division, then round half to even)
            float r = std::nearbyintf(t) + (float)qc[0].x_zp;
            r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
            ql[i] = (uint8_t)r;
        }
        HIPCHK(c, hipMemcpyAsync(base + off, ql.data(), 768, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    uint8_t* const d_qlut = base + off;
    off += 1024;
    float* const d_qstem_w = (float*)(base + off);
    off += align_up(kQStemW, 256);
    float* const d_qstem_lut = (float*)(base + off);
    off += align_up(kQStemLut, 256);
    int32_t* const d_qstem_bias = (int32_t*)(base + off);
    off += align_up(kQStemBias, 256);
    std::vector<int32_t> h_sum, h_bias;
    std::vector<float> h_ws, h_mult;
    for (uint32_t i = 0; i < n; i++) {
        ConvLayer& L = g[i];
        L.x_scale = qc[i].x_scale; L.x_zp = qc[i].x_zp; L.y_scale = qc[i].y_scale; L.y_zp = qc[i].y_zp;
        L.d_w = base + off;
        off += align_up(q_wbytes(L), 256);
        L.d_qbias = (int32_t*)(base + off);
        off += align_up((size_t)L.cout_p * 4, 256);
        L.d_qmult = (float*)(base + off);
        off += align_up((size_t)L.cout_p * 4, 256);
        const int8_t* src_w = (const int8_t*)d_blob + qc[i].w_off;
        h_sum.assign(L.cout_p, 0);
        h_bias.assign(L.cout, 0);
        h_ws.assign(L.cout, 0.f);
        if (L.role == 's') {
            // stem: one dword (r, g, b, 0) per tap and channel, built on the host (9.4 KB)
            std::vector<int8_t> w((size_t)L.cout * 3 * 49);
            HIPCHK(c, hipMemcpyAsync(w.data(), src_w, w.size(), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            std::vector<int32_t> wq((size_t)L.cout * 49);
            for (int o = 0; o < L.cout; o++)
                for (int t = 0; t < 49; t++) {
                    uint32_t d = 0;
                    for (int ch = 0; ch < 3; ch++) {
                        const int8_t v = w[((size_t)o * 3 + ch) * 49 + t];
                        d |= (uint32_t)(uint8_t)v << (8 * ch);
                        h_sum[o] += v;
                    }
                    wq[(size_t)o * 49 + t] = (int32_t)d;
                }
            HIPCHK(c, hipMemcpyAsync(L.d_w, wq.data(), wq.size() * 4, hipMemcpyHostToDevice, c->stream));
            // the fused form's operands: weights as f32 [k][o] with k = (ky * 7 + kx) * 3 + channel, the table as q - x_zp
            if (L.cout != 64) return fail(c, INFUR_E_MODEL_FORMAT, "stem has %d output channels, expected 64", L.cout);
            std::vector<float> wf(147 * 64), lf(768);
            for (int o = 0; o < 64; o++)
                for (int ch = 0; ch < 3; ch++)
                    for (int t = 0; t < 49; t++) wf[(size_t)(t * 3 + ch) * 64 + o] = (float)w[((size_t)o * 3 + ch) * 49 + t];
            for (int i = 0; i < 768; i++) lf[i] = (float)((int)ql[i] - qc[0].x_zp);
            HIPCHK(c, hipMemcpyAsync(d_qstem_w, wf.data(), kQStemW, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(d_qstem_lut, lf.data(), kQStemLut, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(d_qstem_bias, (const uint8_t*)d_blob + qc[i].b_off, kQStemBias, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        } else {
            HIPCHK(c, launch_repack_q(src_w, (int8_t*)L.d_w, (int32_t*)tmp.p, L.cout, L.cin, L.k, L.k, L.cout_p, L.cin_p, c->stream));
            HIPCHK(c, hipMemcpyAsync(h_sum.data(), tmp.p, (size_t)L.cout_p * 4, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipMemcpyAsync(h_bias.data(), (const uint8_t*)d_blob + qc[i].b_off, (size_t)L.cout * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(h_ws.data(), (const uint8_t*)d_blob + qc[i].ws_off, (size_t)L.cout * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        std::vector<int32_t> qb(L.cout_p, 0);
        h_mult.assign(L.cout_p, 0.f);
        for (int o = 0; o < L.cout; o++) {
            if (!qscale_ok(h_ws[o])) return fail(c, INFUR_E_MODEL_FORMAT, "conv '%s': weight scale of channel %d is not positive and finite", L.name.c_str(), o);
            const int64_t b = (int64_t)h_bias[o] + (int64_t)(128 - L.x_zp) * (int64_t)h_sum[o];
            if (b > INT32_MAX || b < INT32_MIN) return fail(c, INFUR_E_MODEL_FORMAT, "conv '%s': bias of channel %d overflows int32", L.name.c_str(), o);
            qb[o] = (int32_t)b;
            volatile float xw = L.x_scale * h_ws[o];  // f32 product, then f32 division: ONNX Runtime's output scale
            h_mult[o] = xw / L.y_scale;
        }
        HIPCHK(c, hipMemcpyAsync(L.d_qbias, qb.data(), (size_t)L.cout_p * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(L.d_qmult, h_mult.data(), (size_t)L.cout_p * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (L.cin2) {
            // Pixel-pair weights, OHWI over pairs: output row p * cout + o (p = which pixel of the output pair), input column
            // q * cin + i.  1x1: W where p == q.  3x3 (pad 1, in pair units too): pair-column kx' in {0, 1, 2} holds input pixel
            // x_in = 2 (X + kx' - 1) + q for output pixel x_out = 2 X + p, i.e. the tap kx = 2 (kx' - 1) + q - p + 1 where that is
            // a tap of the 3x3, zero elsewhere.  A structural zero multiplies whatever the other pixel holds by 0; the row sums,
            // hence the folded bias, and the multipliers are the channel's own, once per pixel of the pair.
            const int taps = L.k * L.k;
            std::vector<int8_t> w((size_t)L.cout * L.cin * taps), w2((size_t)L.cout2 * taps * L.cin2, 0);
            HIPCHK(c, hipMemcpyAsync(w.data(), src_w, w.size(), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            for (int pp = 0; pp < 2; pp++)
                for (int o = 0; o < L.cout; o++)
                    for (int ky = 0; ky < L.k; ky++)
                        for (int kxp = 0; kxp < L.k; kxp++)
                            for (int qq = 0; qq < 2; qq++) {
                                const int kx = L.k == 1 ? (pp == qq ? 0 : -1) : 2 * (kxp - 1) + qq - pp + 1;
                                if (kx < 0 || kx >= L.k) continue;
                                int8_t* dst = &w2[(((size_t)(pp * L.cout + o) * L.k + ky) * L.k + kxp) * L.cin2 + (size_t)qq * L.cin];
                                for (int i = 0; i < L.cin; i++) dst[i] = w[((size_t)o * L.cin + i) * taps + ky * L.k + kx];
                            }
            std::vector<int32_t> qb2(L.cout2);
            std::vector<float> qm2(L.cout2);
            for (int pp = 0; pp < 2; pp++)
                for (int o = 0; o < L.cout; o++) {
                    qb2[pp * L.cout + o] = qb[o];
                    qm2[pp * L.cout + o] = h_mult[o];
                }
            L.d_w2 = base + off;
            off += align_up(w2.size() + 64, 256);
            L.d_qbias2 = (int32_t*)(base + off);
            off += align_up((size_t)L.cout2 * 4, 256);
            L.d_qmult2 = (float*)(base + off);
            off += align_up((size_t)L.cout2 * 4, 256);
            HIPCHK(c, hipMemcpyAsync(L.d_w2, w2.data(), w2.size(), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(L.d_qbias2, qb2.data(), (size_t)L.cout2 * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(L.d_qmult2, qm2.data(), (size_t)L.cout2 * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    }
    model_free(c);  // the old model goes only now
    c->d_weights = arena.p;
    arena.p = nullptr;
    c->convs.swap(g);
    c->qadds.clear();
    for (const QBlobAdd& a : qa) c->qadds.push_back(QAddParams{a.a_scale, a.b_scale, a.c_scale, a.a_zp, a.b_zp, a.c_zp});
    c->d_qlut = d_qlut;
    c->d_qstem_w = d_qstem_w;
    c->d_qstem_lut = d_qstem_lut;
    c->d_qstem_bias = d_qstem_bias;
    c->q_resize_u8 = bh.resize_u8;
    for (const ConvLayer& L : c->convs)
        if (L.role == 'c') {
            const int k = L.name.rfind("aux_", 0) == 0 ? 1 : 0;
            c->q_head_zp[k] = (float)L.y_zp;
            c->q_head_scale[k] = L.y_scale;
        }
    c->quant = true;
    c->depth = bh.depth;
    c->num_classes = bh.num_classes;
    c->has_aux = bh.aux;
    c->input_u8 = false;
    c->weight_bytes = total;
    c->loaded = true;
    infur_model_info& mi = c->info;
    memset(&mi, 0, sizeof mi);
    snprintf(mi.input_name, sizeof mi.input_name, "input");
    snprintf(mi.input0_dtype, sizeof mi.input0_dtype, "Float");  // the int8 zoo model keeps float I/O (QuantizeLinear is its first node)
    snprintf(mi.output_names[0], 32, "out");
    mi.n_outputs = 1 + ((bh.aux && c->opt.compute_aux) ? 1 : 0);
    if (mi.n_outputs == 2) snprintf(mi.output_names[1], 32, "aux");
    mi.num_classes = (uint32_t)bh.num_classes;
    mi.depth = (uint32_t)bh.depth;
    mi.n_convs = n;
    mi.weight_bytes = total;
    mi.quantised = 1;
    mi.resize_u8_heads = bh.resize_u8 ? 1 : 0;
    return INFUR_OK;
}

// one QLinearConv (+ the block's QLinearAdd when `res` is given; f32 output = + DequantizeLinear) on the i8 MFMA
// (pair: `in` / `res` / `out` are pixel-pair views -- (H, W/2, 2C) -- and the layer's pair weights are used: forward_q)
int32_t run_qconv(infur_ctx* c, const ConvLayer& L, const Tensor& in, const Tensor* res, const QAddParams* add, Tensor* out, bool pair = false) {
    const int oh = conv_out(in.h, L.k, L.stride, L.pad, L.dil), ow = conv_out(in.w, L.k, L.stride, L.pad, L.dil);
    const int out_f32 = L.role == 'c' ? 1 : 0;
    if (pair && !L.d_w2) return fail(c, INFUR_E_SHAPE, "'%s' has no pixel-pair weights", L.name.c_str());
    const int cin_k = pair ? L.cin2 : L.cin_p, cout_k = pair ? L.cout2 : L.cout_p;
    if (in.c != cin_k || in.es != 1) return fail(c, INFUR_E_SHAPE, "'%s' expects %d u8 channels, got %d", L.name.c_str(), cin_k, in.c);
    RETIF(talloc(c, oh, ow, cout_k, out_f32 ? 4 : 1, out));
    ConvArgs a;
    a.in = in.p; a.wt = pair ? L.d_w2 : L.d_w; a.bias = nullptr; a.res = res ? res->p : nullptr; a.out = out->p;
    a.H = in.h; a.W = in.w; a.Cin = in.c; a.OH = oh; a.OW = ow; a.Cout = cout_k;
    a.KH = L.k; a.KW = L.k; a.stride = L.stride; a.pad = L.pad; a.dil = L.dil; a.relu = 0;
    a.q_mult = pair ? L.d_qmult2 : L.d_qmult; a.q_bias = pair ? L.d_qbias2 : L.d_qbias; a.q_yzp = L.y_zp; a.q_dq = L.y_scale;
    if (out_f32 && c->q_resize_u8) {  // the file resizes the codes: leave them (as floats) for the post kernels to interpolate
        a.q_dq = 1.0f;
        a.q_dq_off = (float)L.y_zp;
    }
    if (res) {
        if (!add || res->c != cout_k || res->h != oh || res->w != ow) return fail(c, INFUR_E_SHAPE, "residual of '%s' has the wrong shape", L.name.c_str());
        volatile float ra = add->a_scale / add->c_scale, rb = add->b_scale / add->c_scale;  // f32 divisions, as MLAS' QLinearAdd
        a.q_ra = ra; a.q_rb = rb; a.q_bzp = add->b_zp; a.q_czp = add->c_zp;
    }
    const double flops = 2.0 * oh * ow * (double)L.cout * L.cin * L.k * L.k;
    const double bytes = (double)in.bytes() + (double)out->bytes() + (res ? (double)res->bytes() : 0.0) + (double)cout_k * cin_k * L.k * L.k;
    int cfg = -1;
    RETIF(pick_cfg(c, a, 4, out_f32, &cfg));
    {
        ProfScope ps(c, L.name, conv_igemm_config_name(cfg, 4), flops, bytes);
        HIPCHK(c, launch_conv_igemm(a, 4, out_f32, cfg, c->stream));
    }
    if (c->opt.keep_activations) c->kept.push_back(*out);
    return INFUR_OK;
}

// the forward of a quantised model: u8 NHWC activations end to end, dequantised f32 logits in c->out_low / c->aux_low
int32_t forward_q(infur_ctx* c, const uint8_t* d_bgr, int w, int h) {
    size_t ci = 0;
    const ConvLayer& stem = c->convs[ci++];
    const int sh = conv_out(h, 7, 2, 3, 1), sw = conv_out(w, 7, 2, 3, 1);
    const int ph = conv_out(sh, 3, 2, 1, 1), pw = conv_out(sw, 3, 2, 1, 1);
    Tensor s, x;
    // layer1 on PIXEL PAIRS (an even pooled width; not with kept activations, whose per-layer read-back is the padded layout): its
    // 64-channel tensors are stored compact, two neighbouring pixels = one 128-byte GEMM row of a (H, W/2) image, and the convs use
    // the pair arrangement of their weights (model_load_q_dev) -- no channel padding in HBM, half the rows (hence half the MFMA
    // work) for conv2 and layer1.0.conv1; the 256-channel tensors are unchanged: (H, W/2, 512) IS (H, W, 256).  Same integer sums,
    // same epilogue per channel: bit-identical to the padded form (INFUR_Q_NOPAIR=1 keeps that one: tests/test_gpu_quant.py).
    static const bool no_pair_env = getenv("INFUR_Q_NOPAIR") && atoi(getenv("INFUR_Q_NOPAIR")) != 0;
    bool pair = !no_pair_env && !c->opt.keep_activations && !c->opt.no_fuse_stem_pool && (pw % 2) == 0;
    for (const ConvLayer& L : c->convs)
        if (L.name.compare(0, 16, "backbone.layer1.") == 0 && !L.d_w2) pair = false;
    if (!c->opt.keep_activations && !c->opt.no_fuse_stem_pool) {
        // QuantizeLinear + QLinearConv + MaxPool in one launch, exact on the f16 MFMA; the 64-channel stem tensor is never written
        RETIF(talloc(c, ph, pw, pair ? 64 : 128, 1, &x));
        const void* wimg = nullptr;
        RETIF(stem16_image(c, c->d_qstem_w, 1.0f, 0, &wimg));
        ProfScope ps(c, "backbone.conv1+maxpool", "stem_pool_q", 2.0 * sh * sw * 64 * 147, (double)h * w * 3 + (double)x.bytes());
        HIPCHK(c, launch_stem_pool_q(d_bgr, h, w, wimg, c->d_qstem_lut, c->d_qstem_bias, stem.d_qmult, stem.y_zp, (uint8_t*)x.p, pair ? 64 : 128,
                                     sh, sw, ph, pw, c->stream));
    } else {
    {
        RETIF(talloc(c, sh, sw, 64, 1, &s));
        ProfScope ps(c, stem.name, "stem_q", 2.0 * sh * sw * 64 * 147, (double)h * w * 3 + (double)s.bytes());
        HIPCHK(c, launch_stem_q(d_bgr, h, w, c->d_qlut, stem.x_zp, (const int32_t*)stem.d_w, stem.d_qbias, stem.d_qmult, stem.y_zp, (uint8_t*)s.p, sh, sw, c->stream));
    }
    if (c->opt.keep_activations) c->kept.push_back(s);
    {
        RETIF(talloc(c, ph, pw, 128, 1, &x));
        ProfScope ps(c, "backbone.maxpool", "maxpool_q", 0, (double)s.bytes() + (double)x.bytes());
        HIPCHK(c, launch_maxpool_q((const uint8_t*)s.p, sh, sw, 64, (uint8_t*)x.p, ph, pw, 128, c->stream));
    }
    pool_release(c, s);
    }
    Tensor l3;
    size_t blk = 0;
    while (c->convs[ci].role == '1') {
        const ConvLayer& c1 = c->convs[ci];
        const ConvLayer& c2 = c->convs[ci + 1];
        const ConvLayer& c3 = c->convs[ci + 2];
        const bool has_ds = c->convs[ci + 3].role == 'd';
        if (blk >= c->qadds.size()) return fail(c, INFUR_E_SHAPE, "quantised model has fewer residual sums than blocks");
        Tensor t1, t2, idt, y;
        const bool pv = pair && c1.name.compare(0, 16, "backbone.layer1.") == 0;
        Tensor xv = x;  // the block's input as the convs see it
        if (pv) {
            xv.w = x.w / 2;
            xv.c = x.c * 2;
        }
        RETIF(run_qconv(c, c1, xv, nullptr, nullptr, &t1, pv));
        RETIF(run_qconv(c, c2, t1, nullptr, nullptr, &t2, pv));
        pool_release(c, t1);
        if (has_ds) RETIF(run_qconv(c, c->convs[ci + 3], xv, nullptr, nullptr, &idt, pv));
        RETIF(run_qconv(c, c3, t2, has_ds ? &idt : &xv, &c->qadds[blk], &y, pv));
        if (pv) {  // (H, W/2, 512) is (H, W, 256)
            y.w *= 2;
            y.c /= 2;
        }
        if (has_ds && c->opt.keep_activations) std::swap(c->kept[c->kept.size() - 1], c->kept[c->kept.size() - 2]);  // blob order: conv3, downsample
        pool_release(c, t2);
        if (has_ds) pool_release(c, idt);
        blk++;
        ci += has_ds ? 4 : 3;
        const bool end_l3 = c1.name.compare(0, 16, "backbone.layer3.") == 0 && c->convs[ci].name.compare(0, 16, "backbone.layer4.") == 0;
        if (!(l3.p && x.p == l3.p)) pool_release(c, x);
        x = y;
        if (end_l3 && c->has_aux && c->opt.compute_aux) l3 = y;
    }
    {
        Tensor h1;
        RETIF(run_qconv(c, c->convs[ci], x, nullptr, nullptr, &h1));
        pool_release(c, x);
        RETIF(run_qconv(c, c->convs[ci + 1], h1, nullptr, nullptr, &c->out_low));
        pool_release(c, h1);
        ci += 2;
    }
    if (c->has_aux && c->opt.compute_aux) {
        Tensor a1;
        RETIF(run_qconv(c, c->convs[ci], l3, nullptr, nullptr, &a1));
        pool_release(c, l3);
        RETIF(run_qconv(c, c->convs[ci + 1], a1, nullptr, nullptr, &c->aux_low));
        pool_release(c, a1);
    }
    c->last_h = h;
    c->last_w = w;
    return INFUR_OK;
}

}  // namespace infur
