// infur_tuner.cpp -- which tile configuration of the conv kernels runs a layer shape: measured on first use (pick_cfg), kept per
// context, pre-loadable from / exportable to text (infur_tune_import / _export; infur_amd/conv_tune_gfx950.txt holds the decisions
// for the BASELINE shapes).  Every configuration of a mode gives bit-identical results (tests/test_gpu_conv_configs.py,
// test_gpu_hl.py), so the choice is a speed knob only.  Split out of infur_capi.cpp in round 5.
#include "../../include/infur_hip.h"

#include <hip/hip_runtime.h>

#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "infur_rt.h"

using namespace infur;

namespace infur {

// ---- tile configuration of the conv kernel for one problem shape ----
// The first time a shape is seen (a new frame size), every candidate configuration is launched
// for real on the actual operands and timed with HIP events; the fastest is remembered for the
// context's lifetime.  All configurations give bit-identical outputs, so the trial launches are
// simply redundant evaluations of the layer.

int32_t pick_cfg(infur_ctx* c, const ConvArgs& a, int mode, int out_f32, int* cfg) {
    *cfg = conv_igemm_default_config(a);
    if (mode == 5 && !conv_igemm_config_valid(a, *cfg, mode, out_f32)) *cfg = 0;  // (128 x 128: valid for every mode-5 shape)
    // test hook: INFUR_CONV_CFG=<k> forces configuration k wherever it is a candidate
    static const int forced = getenv("INFUR_CONV_CFG") ? atoi(getenv("INFUR_CONV_CFG")) : -1;
    if (forced >= 0) {
        if (conv_igemm_config_valid(a, forced, mode, out_f32)) *cfg = forced;
        return INFUR_OK;
    }
    if (c->opt.no_autotune) return INFUR_OK;
    const std::array<int, 13> key = {a.H, a.W, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.stride, a.dil, a.batch,
                                     a.res ? 1 : (a.in2 ? 2 : 0), mode, out_f32};
    auto it = c->tuned.find(key);
    if (it != c->tuned.end() && conv_igemm_config_valid(a, it->second, mode, out_f32)) {
        *cfg = it->second;
        return INFUR_OK;
    }
    EventPair ev;
    HIPCHK(c, ev.create());
    if (!c->tune_warm) {  // bring clocks and caches to their steady state before the first measurement
        for (int r = 0; r < 12; r++) HIPCHK(c, launch_conv_igemm(a, mode, out_f32, *cfg, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->tune_warm = true;
    }
    float best = 1e30f;
    std::vector<std::pair<int, float>> timed;
    // (configuration 20 -- the BN = 256 form of conv3x3_halo.hip -- is not a tuning candidate: timed in isolation, with its operands
    //  warm in the Infinity Cache, it beats the tiled `dmai` form on the long-K head convs by 2-4 %; inside a frame, where its
    //  one-patch-image chunk boundaries meet HBM latency, it is 5-12 % slower (classifier.0 at 1080p 535 against 477 us).  It stays
    //  selectable -- INFUR_CONV_CFG=20, INFUR_TUNE_HALO256=1 -- and bit-identical: tests/test_gpu_halo.py.)
    static const bool tune_halo256 = getenv("INFUR_TUNE_HALO256") != nullptr;
    for (int k = 0; k < conv_igemm_num_configs(); k++) {
        if (!conv_igemm_config_valid(a, k, mode, out_f32)) continue;
        if (k == 20 && !tune_halo256) continue;
        // a candidate that cannot launch on this shape after all (invalid value) is skipped, not fatal: the layer still
        // has the other configurations; anything else (a fault, a lost device) is an error of the frame
        const hipError_t le = launch_conv_igemm(a, mode, out_f32, k, c->stream);  // warm-up (attributes, caches)
        if (le == hipErrorInvalidValue) continue;
        HIPCHK(c, le);
        float fastest = 1e30f;
        for (int r = 0; r < 4; r++) {  // minimum of 4 single-launch timings
            HIPCHK(c, hipEventRecord(ev.e0, c->stream));
            HIPCHK(c, launch_conv_igemm(a, mode, out_f32, k, c->stream));
            HIPCHK(c, hipEventRecord(ev.e1, c->stream));
            HIPCHK(c, hipEventSynchronize(ev.e1));
            float ms = 0;
            HIPCHK(c, hipEventElapsedTime(&ms, ev.e0, ev.e1));
            if (ms < fastest) fastest = ms;
        }
        timed.emplace_back(k, fastest);
        if (fastest < best) {
            best = fastest;
            *cfg = k;
        }
    }
    // Tie-break towards the larger tile: among the configurations within 2 % of the fastest, the one with the largest
    // BM x BN re-reads its operands least (A once per N tile, B once per M tile) -- the same speed for less L2 / Infinity
    // Cache / HBM traffic, which is also what leaves room for a second frame in flight
    for (const auto& kt : timed)
        if (kt.second <= best * 1.02f && conv_igemm_config_tile_area(kt.first) > conv_igemm_config_tile_area(*cfg)) *cfg = kt.first;
    c->tuned[key] = *cfg;
    c->mem_gen++;  // (a new decision: frames captured as graphs before it are stale)
    return INFUR_OK;
}

}  // namespace infur

static inline void enter(const infur_ctx* c) { ctx_enter(c); }

extern "C" {

// ---- tuning database ----
int32_t infur_tune_export(infur_ctx* c, char* buf, size_t cap, size_t* len) {
    try {
        enter(c);
        if (!c || !len) return INFUR_E_INVALID_ARG;
        std::string out;
        char line[256];
        for (const auto& kv : c->tuned) {
            int n = 0;
            for (int v : kv.first) n += snprintf(line + n, sizeof line - n, "%d ", v);
            snprintf(line + n, sizeof line - n, "%d\n", kv.second);
            out += line;
        }
        *len = out.size();
        if (!buf) return INFUR_OK;
        if (cap < out.size()) return fail(c, INFUR_E_CAPACITY, "tuning text needs %zu bytes", out.size());
        memcpy(buf, out.data(), out.size());
        return INFUR_OK;
    } catch (const std::bad_alloc&) {
        return fail(c, INFUR_E_CAPACITY, "out of host memory");
    } catch (const std::exception& e) {
        return fail(c, INFUR_E_INVALID_ARG, "internal error: %s", e.what());
    }
}

int32_t infur_tune_import(infur_ctx* c, const char* text, size_t len) {
    try {
        enter(c);
        if (!c || (!text && len)) return INFUR_E_INVALID_ARG;
        std::string t(text ? text : "", len);
        size_t pos = 0;
        while (pos < t.size()) {
            size_t eol = t.find('\n', pos);
            if (eol == std::string::npos) eol = t.size();
            const std::string ln = t.substr(pos, eol - pos);
            pos = eol + 1;
            if (ln.empty() || ln[0] == '#') continue;
            std::array<int, 13> key;
            int cfg = -1, off = 0, n = 0;
            bool ok = true;
            for (int i = 0; i < 13 && ok; i++) {
                ok = sscanf(ln.c_str() + off, "%d%n", &key[i], &n) == 1;
                off += n;
            }
            ok = ok && sscanf(ln.c_str() + off, "%d", &cfg) == 1;
            if (!ok || cfg < 0 || cfg >= conv_igemm_num_configs()) return fail(c, INFUR_E_INVALID_ARG, "bad tuning line: %s", ln.c_str());
            c->tuned[key] = cfg;
            c->mem_gen++;
        }
        return INFUR_OK;
    } catch (const std::bad_alloc&) {
        return fail(c, INFUR_E_CAPACITY, "out of host memory");
    } catch (const std::exception& e) {
        return fail(c, INFUR_E_INVALID_ARG, "internal error: %s", e.what());
    }
}

}  // extern "C"
