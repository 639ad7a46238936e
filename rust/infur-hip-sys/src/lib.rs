//! Raw bindings of `include/infur_hip.h`.  UNTESTED: never compiled (no Rust toolchain here).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

#[repr(C)]
pub struct infur_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct infur_stream {
    _private: [u8; 0],
}
#[repr(C)]
pub struct infur_group {
    _private: [u8; 0],
}

#[repr(C)]
pub struct infur_options {
    pub struct_size: u32,
    pub device: i32,
    pub compute_dtype: u32,
    pub compute_aux: u32,
    pub profile: u32,
    pub keep_activations: u32,
    pub winograd_min_cin: u32,
    pub winograd_tile: u32,
    pub no_autotune: u32,
    pub no_fuse_downsample: u32,
    pub no_fuse_stem_pool: u32,
    pub no_fuse_b2b: u32,
    pub stream: *mut c_void,
}

#[repr(C)]
pub struct infur_model_info {
    pub input_name: [c_char; 32],
    pub input0_dtype: [c_char; 16],
    pub output_names: [[c_char; 32]; 2],
    pub n_outputs: u32,
    pub num_classes: u32,
    pub depth: u32,
    pub n_convs: u32,
    pub weight_bytes: u64,
    pub quantised: u32,
    pub resize_u8_heads: u32,
}

/// one profiled kernel launch of the last advance (`infur_profile_get`)
#[repr(C)]
pub struct infur_kernel_record {
    pub name: [c_char; 48],
    pub kernel: [c_char; 32],
    pub ms: f32,
    pub flops: f64,
    pub bytes: f64,
    pub algo_flops: f64,
}

pub const INFUR_OK: i32 = 0;
pub const INFUR_E_INVALID_SCALE: i32 = 1;
pub const INFUR_E_ZERO_SIZE_IN: i32 = 2;
pub const INFUR_E_ZERO_SIZE_OUT: i32 = 3;
pub const INFUR_E_SHAPE: i32 = 4;
pub const INFUR_E_MODEL_NOT_LOADED: i32 = 5;
pub const INFUR_E_MODEL_FORMAT: i32 = 6;
pub const INFUR_E_HIP: i32 = 7;
pub const INFUR_E_RCCL: i32 = 8;
pub const INFUR_E_INVALID_ARG: i32 = 9;
pub const INFUR_E_IO: i32 = 10;
pub const INFUR_E_CAPACITY: i32 = 11;
pub const INFUR_ABI_VERSION: u32 = 6;
pub const INFUR_SCALE_NEAREST: u32 = 0;
pub const INFUR_SCALE_BILINEAR: u32 = 1;
pub const INFUR_DTYPE_F32: u32 = 0;
pub const INFUR_DTYPE_F16: u32 = 1;
/// f32 tensors, conv GEMMs on the f16 matrix cores with hi+lo operand pairs (f32-grade logits, ~1.9x the f32 MFMA rate)
pub const INFUR_DTYPE_F32_SPLIT: u32 = 2;
/// split mode with the two cross terms on the fp8 (e4m3) MX MFMA: logits ~1.5e-4 from f32, ~8 % faster than F32_SPLIT
pub const INFUR_DTYPE_F32_SPLIT_FP8: u32 = 3;
/// three-byte tensors (f16 hi + e5m2 lo planes written by the producer, staged by LDS-DMA), two MFMA units per product:
/// logits ~1.5e-4 from f32 on heavy-tailed weights, ~2.5x the f32 MFMA rate (round 5; 4 is not an option value)
pub const INFUR_DTYPE_F16_HL: u32 = 5;

extern "C" {
    pub fn infur_abi_version() -> u32;
    pub fn infur_status_string(status: i32) -> *const c_char;
    pub fn infur_device_count() -> i32;
    pub fn infur_options_default(o: *mut infur_options);
    pub fn infur_ctx_create(o: *const infur_options, out: *mut *mut infur_ctx) -> i32;
    pub fn infur_ctx_destroy(c: *mut infur_ctx);
    pub fn infur_last_error(c: *const infur_ctx) -> *const c_char;
    pub fn infur_ctx_synchronize(c: *mut infur_ctx) -> i32;

    pub fn infur_scale_validate(factor: f32) -> i32;
    pub fn infur_scale_out_dims(w: u32, h: u32, factor: f32, ow: *mut u32, oh: *mut u32) -> i32;
    pub fn infur_scale(c: *mut infur_ctx, bgr: *const u8, w: u32, h: u32, factor: f32, mode: u32,
                       out: *mut u8, out_capacity: usize, ow: *mut u32, oh: *mut u32) -> i32;

    pub fn infur_model_load(c: *mut infur_ctx, path: *const c_char) -> i32;
    pub fn infur_model_load_blob(c: *mut infur_ctx, blob: *const c_void, len: usize) -> i32;
    pub fn infur_model_unload(c: *mut infur_ctx) -> i32;
    pub fn infur_model_info_get(c: *const infur_ctx, info: *mut infur_model_info) -> i32;
    pub fn infur_model_info_get_sized(c: *const infur_ctx, info: *mut c_void, info_size: usize) -> i32;
    pub fn infur_model_advance(c: *mut infur_ctx, bgr: *const u8, w: u32, h: u32, out: *mut f32,
                               aux: *mut f32, n_outputs: *mut u32) -> i32;

    pub fn infur_colorcode(c: *mut infur_ctx, khw: *const f32, k: u32, h: u32, w: u32, rgba: *mut u8) -> i32;
    pub fn infur_bgr_to_rgba(c: *mut infur_ctx, bgr: *const u8, w: u32, h: u32, rgba: *mut u8) -> i32;
    pub fn infur_frame_advance(c: *mut infur_ctx, bgr: *const u8, w: u32, h: u32, factor: f32, mode: u32,
                               rgba: *mut u8, rgba_capacity: usize, scaled_bgr: *mut u8,
                               ow: *mut u32, oh: *mut u32) -> i32;

    pub fn infur_stream_create(c: *mut infur_ctx, depth: u32, out: *mut *mut infur_stream) -> i32;
    pub fn infur_stream_destroy(s: *mut infur_stream);
    pub fn infur_stream_add_lane(s: *mut infur_stream, other: *mut infur_ctx) -> i32;
    pub fn infur_stream_submit(s: *mut infur_stream, bgr: *const u8, w: u32, h: u32, factor: f32,
                               mode: u32, frame_id: u64) -> i32;
    pub fn infur_stream_pending(s: *const infur_stream) -> u32;
    pub fn infur_stream_next_dims(s: *const infur_stream, frame_id: *mut u64, ow: *mut u32, oh: *mut u32) -> i32;
    pub fn infur_stream_collect(s: *mut infur_stream, rgba: *mut u8, cap: usize, scaled_bgr: *mut u8,
                                frame_id: *mut u64, ow: *mut u32, oh: *mut u32) -> i32;
    // zero-copy ingest / egress (ABI 5): the ring's pinned slots lent to the caller
    pub fn infur_stream_acquire(s: *mut infur_stream, w: u32, h: u32, factor: f32, bgr_slot: *mut *mut u8) -> i32;
    pub fn infur_stream_commit(s: *mut infur_stream, w: u32, h: u32, factor: f32, mode: u32, frame_id: u64) -> i32;
    pub fn infur_stream_collect_view(s: *mut infur_stream, rgba: *mut *const u8, scaled_bgr: *mut *const u8,
                                     frame_id: *mut u64, ow: *mut u32, oh: *mut u32) -> i32;
    pub fn infur_stream_release(s: *mut infur_stream) -> i32;
    pub fn infur_stream_abandon(s: *mut infur_stream) -> i32;
    // pinned host memory for caller-owned frame / mask buffers (moved by DMA by the batch calls)
    pub fn infur_host_alloc(bytes: usize, p: *mut *mut c_void) -> i32;
    pub fn infur_host_free(p: *mut c_void) -> i32;
    pub fn infur_host_is_pinned(p: *const c_void) -> u32;
    /// n frames through one context's depth-3 ring, masks in frame order (BASELINE configs[3] on one GPU)
    pub fn infur_batch_advance(c: *mut infur_ctx, frames: *const *const u8, ws: *const u32, hs: *const u32, n: u32,
                               factor: f32, mode: u32, rgba: *const *mut u8, caps: *const usize,
                               ows: *mut u32, ohs: *mut u32) -> i32;
    pub fn infur_model_warmup(c: *mut infur_ctx, w: u32, h: u32) -> i32;
    pub fn infur_ctx_set_graph_replay(c: *mut infur_ctx, enable: u32) -> i32;
    pub fn infur_ctx_graph_stats(c: *const infur_ctx, captures: *mut u64, replays: *mut u64, cached: *mut u32) -> i32;

    // several GPUs from one process: RCCL weight broadcast + frame-batch sharding
    pub fn infur_group_create(ctxs: *const *mut infur_ctx, n_ctx: u32, out: *mut *mut infur_group) -> i32;
    pub fn infur_group_destroy(g: *mut infur_group);
    pub fn infur_group_last_error(g: *const infur_group) -> *const c_char;
    pub fn infur_group_size(g: *const infur_group) -> u32;
    pub fn infur_group_uses_rccl(g: *const infur_group) -> u32;
    pub fn infur_group_worker_numa_node(g: *const infur_group, i: u32) -> i32;
    pub fn infur_group_weights_broadcast(g: *mut infur_group, root: u32) -> i32;
    pub fn infur_group_batch_advance(g: *mut infur_group, frames: *const *const u8, ws: *const u32, hs: *const u32,
                                     n: u32, factor: f32, mode: u32, rgba: *const *mut u8, caps: *const usize,
                                     ows: *mut u32, ohs: *mut u32) -> i32;
    pub fn infur_weights_broadcast(ctxs: *const *mut infur_ctx, n_ctx: u32) -> i32;
    pub fn infur_batch_advance_multi(ctxs: *const *mut infur_ctx, n_ctx: u32, frames: *const *const u8, ws: *const u32,
                                     hs: *const u32, n: u32, factor: f32, mode: u32, rgba: *const *mut u8,
                                     caps: *const usize, ows: *mut u32, ohs: *mut u32) -> i32;

    /// INFUR_DTYPE_F32_SPLIT: largest |activation| fed to a GEMM and largest |Winograd-domain input| of the last
    /// forward, and whether either left the exact range of the f16 pairs
    pub fn infur_split_range(c: *mut infur_ctx, act_amax: *mut f32, wino_amax: *mut f32, saturated: *mut u32) -> i32;
    // ---- the rest of the header (round 5): device-resident entry points, stage kernels on device buffers, the ONNX converter,
    //      tuning database, per-kernel profile, device memory helpers -- for hosts that keep frames in HBM (a decoder writing into
    //      device memory, a display reading from it) or want the library's measurements; none is needed by `impl Processor` ----
    pub fn infur_ctx_stream(c: *mut infur_ctx) -> *mut c_void;
    pub fn infur_scale_dev(c: *mut infur_ctx, d_bgr: *const c_void, w: u32, h: u32, factor: f32, mode: u32, d_out: *mut c_void,
                           out_capacity: usize, ow: *mut u32, oh: *mut u32) -> i32;
    pub fn infur_onnx_to_blob(onnx: *const c_void, len: usize, blob: *mut *mut c_void, blob_len: *mut usize, err: *mut c_char,
                              errcap: usize) -> i32;
    pub fn infur_buffer_free(p: *mut c_void);
    pub fn infur_model_load_blob_dev(c: *mut infur_ctx, d_blob: *const c_void, len: usize) -> i32;
    pub fn infur_model_advance_dev(c: *mut infur_ctx, d_bgr: *const c_void, w: u32, h: u32, d_out: *mut c_void, d_aux: *mut c_void,
                                   n_outputs: *mut u32) -> i32;
    pub fn infur_model_lowres_dims(h: u32, w: u32, lh: *mut u32, lw: *mut u32) -> i32;
    pub fn infur_model_read_lowres(c: *mut infur_ctx, out_low: *mut f32, aux_low: *mut f32, lh: *mut u32, lw: *mut u32) -> i32;
    pub fn infur_debug_read_activation(c: *mut infur_ctx, index: u32, host_chw: *mut f32, cap_floats: usize, ch: *mut u32,
                                       h: *mut u32, w: *mut u32) -> i32;
    pub fn infur_pack_normalize(c: *mut infur_ctx, bgr: *const u8, w: u32, h: u32, chw: *mut f32) -> i32;
    pub fn infur_pack_normalize_dev(c: *mut infur_ctx, d_bgr: *const c_void, w: u32, h: u32, d_chw: *mut c_void) -> i32;
    pub fn infur_colorcode_dev(c: *mut infur_ctx, d_khw: *const c_void, k: u32, h: u32, w: u32, d_rgba: *mut c_void) -> i32;
    pub fn infur_bgr_to_rgba_dev(c: *mut infur_ctx, d_bgr: *const c_void, w: u32, h: u32, d_rgba: *mut c_void) -> i32;
    pub fn infur_frame_advance_dev(c: *mut infur_ctx, d_bgr: *const c_void, w: u32, h: u32, factor: f32, scale_mode: u32,
                                   d_rgba: *mut c_void, rgba_capacity: usize, d_scaled_bgr: *mut c_void, ow: *mut u32,
                                   oh: *mut u32) -> i32;
    pub fn infur_tune_export(c: *mut infur_ctx, buf: *mut c_char, cap: usize, len: *mut usize) -> i32;
    pub fn infur_tune_import(c: *mut infur_ctx, text: *const c_char, len: usize) -> i32;
    pub fn infur_profile_enable(c: *mut infur_ctx, on: u32) -> i32;
    pub fn infur_profile_count(c: *mut infur_ctx, n: *mut u32) -> i32;
    pub fn infur_profile_get(c: *mut infur_ctx, i: u32, rec: *mut infur_kernel_record) -> i32;
    pub fn infur_dev_alloc(c: *mut infur_ctx, bytes: usize, d_ptr: *mut *mut c_void) -> i32;
    pub fn infur_dev_free(c: *mut infur_ctx, d_ptr: *mut c_void) -> i32;
    pub fn infur_memcpy_h2d(c: *mut infur_ctx, d_dst: *mut c_void, src: *const c_void, bytes: usize) -> i32;
    pub fn infur_memcpy_d2h(c: *mut infur_ctx, dst: *mut c_void, d_src: *const c_void, bytes: usize) -> i32;
}
