// conv_igemm.hip -- the configuration table of the implicit-GEMM convolution (conv_igemm_kernel.h) and the dispatch to the
// per-mode translation units.
#include <mutex>
#include <string>

#include "kernels.h"

namespace infur {

// one launcher per arithmetic mode, each in its own translation unit (conv_igemm_<mode>.hip)
hipError_t conv_igemm_launch_f32(const ConvArgs& a, int cfg, hipStream_t s);
hipError_t conv_igemm_launch_f16(const ConvArgs& a, int out_f32, int cfg, hipStream_t s);
hipError_t conv_igemm_launch_split(const ConvArgs& a, int fp8_cross, int cfg, hipStream_t s);
hipError_t conv_igemm_launch_i8(const ConvArgs& a, int out_f32, int cfg, hipStream_t s);
#ifdef KTRACE
hipError_t ktrace_read_f32(unsigned long long* out);
hipError_t ktrace_read_f16(unsigned long long* out);
hipError_t ktrace_read_split(unsigned long long* out);
hipError_t ktrace_read_i8(unsigned long long* out);
hipError_t ktrace_read(unsigned long long* out) {  // the buffer of the mode that traced something
    hipError_t (*const rd[4])(unsigned long long*) = {ktrace_read_f32, ktrace_read_f16, ktrace_read_split, ktrace_read_i8};
    unsigned long long tmp[8 * 8 * 4];
    bool any = false;
    for (auto f : rd) {
        const hipError_t e = f(tmp);
        if (e != hipSuccess) return e;
        bool nz = false;
        for (auto v : tmp) nz |= v != 0;
        if (nz || !any) {
            for (int i = 0; i < 8 * 8 * 4; i++) out[i] = tmp[i];
            any |= nz;
        }
    }
    return hipSuccess;
}
#endif

// ---- tile configurations ----
// All configurations accumulate k in the same order for every output element, so the choice only
// changes speed, never a single bit of the result.
struct CfgInfo {
    int bm, bn;
    const char* name[3];  // per mode: f32, f16, f32 split into f16 pairs
};
static const CfgInfo kCfgs[] = {
    {128, 128, {"conv_igemm_f32<128,128>", "conv_igemm_f16<128,128>", "conv_igemm_f32s<128,128>"}},
    {64, 128, {"conv_igemm_f32<64,128>", "conv_igemm_f16<64,128>", "conv_igemm_f32s<64,128>"}},
    {128, 64, {"conv_igemm_f32<128,64>", "conv_igemm_f16<128,64>", "conv_igemm_f32s<128,64>"}},
    {64, 64, {"conv_igemm_f32<64,64>", "conv_igemm_f16<64,64>", "conv_igemm_f32s<64,64>"}},
    {256, 32, {"conv_igemm_f32<256,32>", "conv_igemm_f16<256,32>", "conv_igemm_f32s<256,32>"}},
    {128, 256, {"conv_igemm_f32<128,256>", "conv_igemm_f16<128,256>", "conv_igemm_f32s<128,256>"}},
    {256, 128, {"conv_igemm_f32<256,128>", "conv_igemm_f16<256,128>", "conv_igemm_f32s<256,128>"}},
    {128, 128, {"conv_igemm_f32<128,128,1buf>", "conv_igemm_f16<128,128,1buf>", "conv_igemm_f32s<128,128,1buf>"}},
    {128, 64, {"conv_igemm_f32<128,64,1buf>", "conv_igemm_f16<128,64,1buf>", "conv_igemm_f32s<128,64,1buf>"}},
    {64, 128, {"conv_igemm_f32<64,128,1buf>", "conv_igemm_f16<64,128,1buf>", "conv_igemm_f32s<64,128,1buf>"}},
    {64, 64, {"conv_igemm_f32<64,64,1buf>", "conv_igemm_f16<64,64,1buf>", "conv_igemm_f32s<64,64,1buf>"}},
    {256, 256, {"conv_igemm_f32<256,256,1frag>", "conv_igemm_f16<256,256,1frag>", "conv_igemm_f32s<256,256,1frag>"}},
    {256, 128, {"conv_igemm_f32<256,128,1frag>", "conv_igemm_f16<256,128,1frag>", "conv_igemm_f32s<256,128,1frag>"}},
    // LDS-DMA staging (f16 operands only, conv_igemm_config_valid_mode; the kernel is written in bytes and the DMA forms were
    // built and measured for f32 too: 134-138 TFLOP/s against 139-144 for the tuned register forms at 1080p -- the 64-cycle
    // f32 MFMAs hide the register staging anyway and 256-row tiles are too coarse for M = 32400)
    {256, 256, {"conv_igemm_f32<256,256,dma>", "conv_igemm_f16<256,256,dma>", "conv_igemm_f32s<256,256,dma>"}},
    {256, 128, {"conv_igemm_f32<256,128,dma>", "conv_igemm_f16<256,128,dma>", "conv_igemm_f32s<256,128,dma>"}},
    // short-K 1x1 convs, f16: activation tile in registers, all N tiles walked by one workgroup (conv1x1_areg.hip)
    {256, 128, {"conv1x1_f32<256,areg>", "conv1x1_f16<256,areg>", "conv1x1_f32s<256,areg>"}},
    // LDS-DMA staging with the DMA instructions issued between the slices (NBUF == 5: the MFMA-bound layers)
    {256, 256, {"conv_igemm_f32<256,256,dmai>", "conv_igemm_f16<256,256,dmai>", "conv_igemm_f32s<256,256,dmai>"}},
    {256, 128, {"conv_igemm_f32<256,128,dmai>", "conv_igemm_f16<256,128,dmai>", "conv_igemm_f32s<256,128,dmai>"}},
    // configuration 15 of the quantised mode with the N tiles of an M tile shared out over several workgroups (conv1x1_q8.hip):
    // M = 32400 alone gives 254 workgroups for 256 CUs
    {256, 128, {"conv1x1_f32<256,areg,nsplit>", "conv1x1_f16<256,areg,nsplit>", "conv1x1_f32s<256,areg,nsplit>"}},
    // round 4: stride-1 3x3 convs in the f16 mode with the input patch of a 16 x 16 output tile resident in LDS for all nine taps
    // (conv3x3_halo.hip): a fifth of the activation ingest of the tiled forms -- for M = 32400, where one tile per CU is bound by
    // the L2 -> LDS path, not by the MFMA
    {256, 128, {"conv3x3_f32<16x16,128,halo>", "conv3x3_f16<16x16,128,halo>", "conv3x3_f32s<16x16,128,halo>"}},
    {256, 256, {"conv3x3_f32<16x16,256,halo>", "conv3x3_f16<16x16,256,halo>", "conv3x3_f32s<16x16,256,halo>"}},
    // the same with ONE wave per SIMD and a 128 x 128 wave tile (two thirds of the fragment reads per MFMA), software-pipelined across
    // the weight steps (conv3x3_halo.hip, "the 4-wave form")
    {256, 256, {"conv3x3_f32<16x16,256,halo4>", "conv3x3_f16<16x16,256,halo4>", "conv3x3_f32s<16x16,256,halo4>"}},
    // (late round 3: a 128x32 tile of two waves for the logit convs -- M = 32400 gives only 127 workgroups of 256 rows for 256 CUs --
    // is bit-identical and 2 us faster per launch (f32 28.8 -> 26.6 / 20.8 -> 18.6 us, i8 10.6 -> 9.5): the launch is latency-bound,
    // not short of workgroups; not worth a configuration.)
    // (a RING OF THREE LDS images with a counted vmcnt -- two K steps of DMA in flight across the barrier -- was measured on
    // 256x128, 128x256 and 128x128 tiles for the HBM-bound 1x1 convs: better than the two-image form of the same tile
    // (layer3 conv1 at 4K: 0.123 -> 0.100 ms) but never better than 256x256 with two images (0.087) or the register form
    // (conv3: 0.170 vs 0.202); 256x256 x 3 images does not fit the 160 KB.  The tuner picked none of them: not shipped.)
    // (a two-group PING-PONG schedule on top of the DMA form -- waves 0-3 load while waves 4-7 compute, 4 barriers per K
    // step, raised priority on the MFMA clusters -- was built, is bit-identical, and is NOT faster: 1143 vs 1183 TFLOP/s on
    // the 4K classifier.0.  What did help is WHERE the DMA instructions are issued inside the step: the `dmai` form above,
    // +6-8 % on the MFMA-bound layers.  Beyond that the gap to the peak is mostly the package clock: 1.99 GHz on real data
    // against 2.38 GHz on all-zero operands for the SAME binary -- scripts/zero_data_probe.py, profiles/r02_dvfs_probe.md.)
    // (round 3, both measured on the 4K FCN-ResNet101 and dropped: (1) FOUR waves of 128 x 128 on the `dmai` tile -- a third less
    // fragment-read traffic per FLOP, the whole 512-register file per wave -- is 20 % SLOWER on every MFMA-bound layer (layer3 conv2
    // 150 -> 184 us, classifier.0 1973 -> 2295): with one wave per SIMD nothing runs while that wave sits at a DMA issue or a
    // barrier; (2) walking K with the TAPS INSIDE each 128-byte channel chunk -- the nine shifted windows of a chunk back to
    // back, so that L2 serves their overlap instead of the 9x re-fetch the counters show -- changes nothing (154.7 vs 155.6 us):
    // those re-reads come out of the Infinity Cache and the loop is not waiting for them.)
    // (round 3, quantised mode: a first port of the A-resident walk of conv1x1_areg.hip to the i8 MFMA -- 8 waves, output through
    // per-wave LDS slices, 134 KB of LDS = one workgroup per CU -- was slower than the tiled forms on every 1x1 (layer3 conv3 at
    // 1080p 37 vs 31 us): nothing covered its per-N-tile residual loads and requantisation.  conv1x1_q8.hip is the second form --
    // 16 consecutive channels per lane straight from the accumulators, no LDS staging, bias / multiplier tables in LDS -- and is
    // configuration 15 of mode 4: it wins where M is large (4K: conv3 and downsample convs 5-12 % faster than the tiled forms;
    // 1080p: layer1 only -- with M = 32400 there is one 32-pixel wave per SIMD and the N-split tiles have more to overlap).)
    // (128x128 and 128x256 DMA tiles were measured too: slower than the register-staged forms on every layer of the 4K
    // FCN-ResNet101, including the HBM-bound 1x1 convs they were meant for -- 0.199 / 0.208 ms vs 0.170 on layer3 conv3)
};
constexpr int kNumCfgs = (int)(sizeof(kCfgs) / sizeof(kCfgs[0]));

int conv_igemm_num_configs() { return kNumCfgs; }

int conv_igemm_config_tile_area(int cfg) { return cfg < 0 || cfg >= kNumCfgs ? 0 : kCfgs[cfg].bm * kCfgs[cfg].bn; }

const char* conv_igemm_config_name(int cfg, int mode) {
    if (mode == 5) {  // the three-byte mode's own kernel (conv_hl.hip) on the tile shapes of these four entries
        switch (cfg) {
            case 11: return "conv_hl<256,256>";
            case 0: return "conv_hl<128,128>";
            case 6: return "conv_hl<256,128>";
            case 5: return "conv_hl<128,256>";
            case 12: return "conv_hl<256,128,4w>";
            case 14: return "conv_hl<128,256,4w>";
            case 13: return "conv_hl<256,256,wn2>";
            case 16: return "conv_hl<128,256,4w,wn2>";
            case 17: return "conv_hl<256,128,4w,wn2>";
            case 15: return "conv_hl<128,areg>";
            default: return "conv_hl<?>";
        }
    }
    if (cfg < 0 || cfg >= kNumCfgs || mode < 0 || mode > 4) return "conv_igemm<?>";
    if (mode >= 3) {  // "conv_igemm_f32s<...>" -> "conv_igemm_f32x<...>" / "conv_igemm_i8<...>"
        static std::string names[2][64];
        static std::once_flag once;
        std::call_once(once, [] {
            for (int k = 0; k < kNumCfgs && k < 64; k++) {
                names[0][k] = names[1][k] = kCfgs[k].name[2];
                const size_t p = names[0][k].find("f32s");
                if (p != std::string::npos) {
                    names[0][k][p + 3] = 'x';
                    names[1][k].replace(p, 4, "i8");
                }
            }
        });
        return names[mode - 3][cfg].c_str();
    }
    return kCfgs[cfg].name[mode];
}

// the heuristic used when no measurement is available
int conv_igemm_default_config(const ConvArgs& a) {
    if (a.Cout >= 128) return 0;
    if (a.Cout > 32) return 2;
    return 4;
}

// a configuration is a candidate when its N tile is not mostly padding (and, for the LDS-DMA forms, in the f16 mode)
bool conv_igemm_config_valid(const ConvArgs& a, int cfg, int mode, int out_f32) {
    if (cfg < 0 || cfg >= kNumCfgs) return false;
    if (mode == 5) return conv_hl_config_valid(a, cfg, out_f32);
    if (mode == 3) mode = 2;  // the fp8 cross-term form stages like the split mode
    if (cfg == 19 || cfg == 20) return conv3x3_halo_valid(a, mode, out_f32, cfg == 19 ? 128 : 256);
    if (cfg == 21) return conv3x3_halo4_valid(a, mode, out_f32);
    if (cfg >= 13 && mode != 1 && mode != 4) return false;  // LDS-DMA staging: byte operands that need no conversion (f16, i8)
    if (cfg == 18) return mode == 4 && conv1x1_q8_valid(a, mode, out_f32) && conv1x1_q8_nsplit(a) > 1;
    if (cfg == 15) return mode == 4 ? conv1x1_q8_valid(a, mode, out_f32) : conv1x1_areg_valid(a, mode, out_f32);  // (never the f32 logits)
    const int bn = kCfgs[cfg].bn;
    if (a.Cout <= 32) return bn == 32;
    if (bn == 32) return false;
    return bn <= a.Cout || bn == 64;  // Cout = 64 -> BN 64 only; Cout >= 128 -> 64 and 128 (and 256 when Cout >= 256)
}

hipError_t launch_conv_igemm(const ConvArgs& a, int mode, int out_f32, int cfg, hipStream_t s) {
    if (mode == 5) return launch_conv_hl(a, out_f32, cfg, s);
    if (mode == 0) return conv_igemm_launch_f32(a, cfg, s);
    if (mode == 2) return conv_igemm_launch_split(a, 0, cfg, s);
    if (mode == 3) return conv_igemm_launch_split(a, 1, cfg, s);  // f32 tensors, f16 MFMA + fp8 MX MFMA for the cross terms
    if (mode == 4) {  // quantised: u8 activations x s8 weights on the i8 MFMA, requantised in the epilogue
        if (!a.q_mult || !a.q_bias) return hipErrorInvalidValue;
        return conv_igemm_launch_i8(a, out_f32, cfg, s);
    }
    return conv_igemm_launch_f16(a, out_f32, cfg, s);
}

}  // namespace infur
