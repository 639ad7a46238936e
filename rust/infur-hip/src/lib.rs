//! `impl Processor` for the HIP-backed Scale / Model / ColorCode.
//!
//! UNTESTED: never compiled (no Rust toolchain in the build environment).  Written against the
//! reference's trait (`infur/src/processing.rs:23-60`); inside the `infur` crate replace
//! `use crate::processing::Processor` accordingly and swap the three type names in
//! `ProcessingApp` (`infur/src/app.rs:51-62`).
use std::{ffi::{CStr, CString}, rc::Rc};

use eframe::epaint::{Color32, ColorImage};
use image_ext::BgrImage;
use infur_hip_sys as sys;
use ndarray::{Array3, ArrayD, IxDyn};
use thiserror::Error;

/// The reference's plugin contract, restated so this crate stands alone.
pub trait Processor {
    type Command;
    type ControlError;
    type Input;
    type Output;
    type ProcessResult;
    fn control(&mut self, cmd: Self::Command) -> Result<&mut Self, Self::ControlError>;
    fn advance(&mut self, inp: &Self::Input, out: &mut Self::Output) -> Self::ProcessResult;
    fn is_dirty(&self) -> bool;
    /// Provided method of the reference's trait (`infur/src/processing.rs:53-59`): run the node on default input and
    /// output -- how the `Proc` loop drives the final node of the graph (`main.rs:85`).
    fn generate(&mut self) -> Self::ProcessResult
    where
        Self::Input: Default,
        Self::Output: Default,
    {
        let inp = <Self::Input as Default>::default();
        let mut out = <Self::Output as Default>::default();
        self.advance(&inp, &mut out)
    }
}

pub struct Frame {
    pub id: u64,
    pub img: BgrImage,
}

/// One GPU + one HIP stream.  `!Send`: keep it on the `Proc` thread like the ORT session
/// (`infur/src/main.rs:38-40`).
pub struct Ctx(*mut sys::infur_ctx);

impl Ctx {
    pub fn new(device: i32) -> Result<Rc<Self>, HipError> {
        let mut o = std::mem::MaybeUninit::<sys::infur_options>::uninit();
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe {
            sys::infur_options_default(o.as_mut_ptr());
            let mut o = o.assume_init();
            o.device = device;
            sys::infur_ctx_create(&o, &mut ctx)
        };
        if rc != sys::INFUR_OK { return Err(HipError::status(rc)); }
        Ok(Rc::new(Ctx(ctx)))
    }
    fn err(&self, rc: i32) -> HipError {
        let msg = unsafe { CStr::from_ptr(sys::infur_last_error(self.0)) }.to_string_lossy().into_owned();
        HipError { code: rc, msg }
    }
    /// the decisions of the tile-configuration tuner so far, as text (`infur_tune_export`): feed it to `tuning_import` of a later
    /// process and its first frame finds every conv shape's configuration already measured
    pub fn tuning_text(&self) -> Result<String, HipError> {
        let mut len = 0usize;
        let rc = unsafe { sys::infur_tune_export(self.0, std::ptr::null_mut(), 0, &mut len) };
        if rc != sys::INFUR_OK { return Err(self.err(rc)); }
        let mut buf = vec![0u8; len + 1];
        let rc = unsafe { sys::infur_tune_export(self.0, buf.as_mut_ptr() as *mut std::os::raw::c_char, buf.len(), &mut len) };
        if rc != sys::INFUR_OK { return Err(self.err(rc)); }
        buf.truncate(len);
        Ok(String::from_utf8_lossy(&buf).into_owned())
    }
    pub fn tuning_import(&self, text: &str) -> Result<(), HipError> {
        let rc = unsafe { sys::infur_tune_import(self.0, text.as_ptr() as *const std::os::raw::c_char, text.len()) };
        if rc != sys::INFUR_OK { return Err(self.err(rc)); }
        Ok(())
    }
    /// per-kernel HIP-event records of the last advance (after `set_profile(true)`): (layer, kernel family, milliseconds, FLOPs, bytes)
    pub fn set_profile(&self, on: bool) -> Result<(), HipError> {
        let rc = unsafe { sys::infur_profile_enable(self.0, on as u32) };
        if rc != sys::INFUR_OK { return Err(self.err(rc)); }
        Ok(())
    }
    pub fn profile(&self) -> Result<Vec<(String, String, f32, f64, f64)>, HipError> {
        let mut n = 0u32;
        let rc = unsafe { sys::infur_profile_count(self.0, &mut n) };
        if rc != sys::INFUR_OK { return Err(self.err(rc)); }
        let mut out = Vec::with_capacity(n as usize);
        for i in 0..n {
            let mut rec = std::mem::MaybeUninit::<sys::infur_kernel_record>::zeroed();
            let rc = unsafe { sys::infur_profile_get(self.0, i, rec.as_mut_ptr()) };
            if rc != sys::INFUR_OK { return Err(self.err(rc)); }
            let rec = unsafe { rec.assume_init() };
            let name = unsafe { CStr::from_ptr(rec.name.as_ptr()) }.to_string_lossy().into_owned();
            let kernel = unsafe { CStr::from_ptr(rec.kernel.as_ptr()) }.to_string_lossy().into_owned();
            out.push((name, kernel, rec.ms, rec.flops, rec.bytes));
        }
        Ok(out)
    }
}
impl Drop for Ctx {
    fn drop(&mut self) { unsafe { sys::infur_ctx_destroy(self.0) } }
}

#[derive(Error, Debug)]
#[error("{msg} (status {code})")]
pub struct HipError { pub code: i32, pub msg: String }
impl HipError {
    fn status(rc: i32) -> Self {
        let msg = unsafe { CStr::from_ptr(sys::infur_status_string(rc)) }.to_string_lossy().into_owned();
        HipError { code: rc, msg }
    }
    /// the context's own message for `rc` (infur_last_error), falling back to the status string
    fn from_ctx(ctx: &Ctx, rc: i32) -> Self {
        let p = unsafe { sys::infur_last_error(ctx.0) };
        if p.is_null() { return Self::status(rc); }
        let msg = unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned();
        if msg.is_empty() { Self::status(rc) } else { HipError { code: rc, msg } }
    }
}

// ---------------------------------------------------------------- Scale (processing.rs:142-282)
#[derive(PartialEq, Debug, Clone, Copy)]
pub struct ValidScale(f32);
#[derive(Error, Debug)]
#[error("Cannot scale by negative number")]
pub struct ValidScaleError;
impl TryFrom<f32> for ValidScale {
    type Error = ValidScaleError;
    fn try_from(v: f32) -> Result<Self, Self::Error> {
        if unsafe { sys::infur_scale_validate(v) } != sys::INFUR_OK { Err(ValidScaleError) } else { Ok(Self(v)) }
    }
}

#[derive(Error, Debug)]
pub enum ScaleProcError {
    #[error("scaling from 0-sized input")]
    ZeroSizeIn,
    #[error("scaling to 0-sized output")]
    ZeroSizeOut,
    #[error(transparent)]
    Hip(#[from] HipError),
}

pub struct HipScale { ctx: Rc<Ctx>, factor: ValidScale, dirty: bool, mode: u32 }
impl HipScale {
    pub fn new(ctx: Rc<Ctx>) -> Self { Self { ctx, factor: ValidScale(1.0), dirty: true, mode: sys::INFUR_SCALE_NEAREST } }
}
impl Processor for HipScale {
    type Command = f32;
    type ControlError = ValidScaleError;
    type Input = Option<Frame>;
    type Output = Option<Frame>;
    type ProcessResult = Result<(), ScaleProcError>;

    fn control(&mut self, cmd: f32) -> Result<&mut Self, ValidScaleError> {
        let factor: ValidScale = cmd.try_into()?;
        self.dirty = factor != self.factor;
        self.factor = factor;
        Ok(self)
    }
    fn is_dirty(&self) -> bool { self.dirty }
    fn advance(&mut self, input: &Option<Frame>, out: &mut Option<Frame>) -> Result<(), ScaleProcError> {
        self.dirty = false;
        let input = match input { Some(i) => i, None => return Ok(()) };
        if self.factor.0 == 1.0 {
            *out = Some(Frame { id: input.id, img: input.img.clone() });
            return Ok(());
        }
        let (w, h) = (input.img.width(), input.img.height());
        let (mut ow, mut oh) = (0u32, 0u32);
        match unsafe { sys::infur_scale_out_dims(w, h, self.factor.0, &mut ow, &mut oh) } {
            sys::INFUR_E_ZERO_SIZE_IN => return Err(ScaleProcError::ZeroSizeIn),
            sys::INFUR_E_ZERO_SIZE_OUT => return Err(ScaleProcError::ZeroSizeOut),
            _ => {}
        }
        let frame = out.get_or_insert_with(|| Frame { id: input.id, img: BgrImage::new(ow, oh) });
        if frame.img.width() != ow || frame.img.height() != oh { frame.img = BgrImage::new(ow, oh); }
        frame.id = input.id;
        let cap = frame.img.as_raw().len();
        let rc = unsafe {
            sys::infur_scale(self.ctx.0, input.img.as_raw().as_ptr(), w, h, self.factor.0, self.mode,
                             frame.img.as_mut().as_mut_ptr(), cap, &mut ow, &mut oh)
        };
        if rc == sys::INFUR_OK { Ok(()) } else { Err(self.ctx.err(rc).into()) }
    }
}

// ---------------------------------------------------------------- Model (predict_onnx.rs:146-345)
#[derive(Clone, Debug)]
pub enum ModelCmd { Load(String) }
#[derive(Debug, Clone)]
pub struct ModelInfo { pub input_names: Vec<String>, pub input0_dtype: String, pub output_names: Vec<String>, pub quantised: bool, pub resize_u8_heads: bool }

pub struct HipModel { ctx: Rc<Ctx> }
impl HipModel {
    pub fn new(ctx: Rc<Ctx>) -> Self { Self { ctx } }
    fn raw_info(&self) -> Option<sys::infur_model_info> {
        let mut mi = std::mem::MaybeUninit::<sys::infur_model_info>::uninit();
        if unsafe { sys::infur_model_info_get(self.ctx.0, mi.as_mut_ptr()) } != sys::INFUR_OK { return None; }
        Some(unsafe { mi.assume_init() })
    }
    pub fn get_info(&self) -> Option<ModelInfo> {
        let mi = self.raw_info()?;
        let s = |p: *const std::os::raw::c_char| unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned();
        Some(ModelInfo {
            input_names: vec![s(mi.input_name.as_ptr())],
            input0_dtype: s(mi.input0_dtype.as_ptr()),
            output_names: (0..mi.n_outputs as usize).map(|i| s(mi.output_names[i].as_ptr())).collect(),
            quantised: mi.quantised != 0,
            resize_u8_heads: mi.resize_u8_heads != 0,
        })
    }
}
/// Error processing model -- the reference's `ModelProcError` (`predict_onnx.rs:32-39`) with the ORT arm replaced
/// by the HIP runtime's (same variant names, same messages, so `AppProcError` keeps its `#[from]`, app.rs:17-36)
#[derive(Error, Debug)]
pub enum ModelProcError {
    #[error("couldn't transform image")]
    ShapeError(#[from] ndarray::ShapeError),
    #[error("scaling to 0-sized output")]
    RuntimeError(#[from] HipError),
}

/// Error loading model -- the reference's `ModelCmdError` (`predict_onnx.rs:41-48`): the runtime's own load error,
/// or the file's input 0 not being an image the path can run (`infer_img_pre_proc`, `predict_onnx.rs:223-265`)
#[derive(Error, Debug)]
pub enum ModelCmdError {
    #[error(transparent)]
    Hip(HipError),
    #[error(transparent)]
    RuntimeError(#[from] ModelInputFormatError),
}

#[derive(Error, Debug)]
pub enum ModelInputFormatError {
    #[error("couldn't infer image input")]
    Infer(String),
}

impl Processor for HipModel {
    type Command = ModelCmd;
    type ControlError = ModelCmdError;
    type Input = BgrImage;
    type Output = Vec<ArrayD<f32>>;
    type ProcessResult = Result<(), ModelProcError>;

    fn control(&mut self, cmd: ModelCmd) -> Result<&mut Self, ModelCmdError> {
        let ModelCmd::Load(path) = cmd; // "" unloads inside the library (predict_onnx.rs:310-312)
        let c = CString::new(path).map_err(|_| ModelCmdError::Hip(HipError { code: sys::INFUR_E_INVALID_ARG, msg: "path contains NUL".into() }))?;
        match unsafe { sys::infur_model_load(self.ctx.0, c.as_ptr()) } {
            sys::INFUR_OK => Ok(self),
            // the library reports input-layout problems with the reference's own messages (onnx_reader.cpp)
            sys::INFUR_E_MODEL_FORMAT => Err(ModelInputFormatError::Infer(self.ctx.err(sys::INFUR_E_MODEL_FORMAT).msg).into()),
            rc => Err(ModelCmdError::Hip(self.ctx.err(rc))),
        }
    }
    fn is_dirty(&self) -> bool { false }
    fn advance(&mut self, img: &BgrImage, out: &mut Vec<ArrayD<f32>>) -> Result<(), ModelProcError> {
        let mi = match self.raw_info() { Some(mi) => mi, None => return Ok(()) }; // no session: out untouched
        let (k, h, w) = (mi.num_classes as usize, img.height() as usize, img.width() as usize);
        // as many tensors as the model has outputs: [out, aux], or [out] without the aux head
        let mut bufs: Vec<Vec<f32>> = (0..mi.n_outputs).map(|_| vec![0f32; k * h * w]).collect();
        let aux = if bufs.len() > 1 { bufs[1].as_mut_ptr() } else { std::ptr::null_mut() };
        let mut n = 0u32;
        let rc = unsafe {
            sys::infur_model_advance(self.ctx.0, img.as_raw().as_ptr(), w as u32, h as u32,
                                     bufs[0].as_mut_ptr(), aux, &mut n)
        };
        if rc != sys::INFUR_OK { return Err(self.ctx.err(rc).into()); }
        out.clear();
        for b in bufs { out.push(ArrayD::from_shape_vec(IxDyn(&[k, h, w]), b)?); }
        Ok(())
    }
}

// ---------------------------------------------------------------- several GPUs (main.rs:38-40: one Proc thread)
/// `infur_group`: the contexts of one process driven together -- RCCL weight broadcast over xGMI and frame-batch
/// sharding (BASELINE configs[3]).  The `Ctx`s are kept alive by the `Rc`s held here.
pub struct HipGroup { raw: *mut sys::infur_group, ctxs: Vec<Rc<Ctx>> }
impl HipGroup {
    pub fn new(ctxs: Vec<Rc<Ctx>>) -> Result<Self, HipError> {
        let ptrs: Vec<*mut sys::infur_ctx> = ctxs.iter().map(|c| c.0).collect();
        let mut raw = std::ptr::null_mut();
        match unsafe { sys::infur_group_create(ptrs.as_ptr(), ptrs.len() as u32, &mut raw) } {
            sys::INFUR_OK => Ok(Self { raw, ctxs }),
            rc => Err(ctxs[0].err(rc)),
        }
    }
    fn err(&self, rc: i32) -> HipError {
        let msg = unsafe { CStr::from_ptr(sys::infur_group_last_error(self.raw)) }.to_string_lossy().into_owned();
        HipError { code: rc, msg }
    }
    /// Replicate the model loaded in `ctxs[root]` to every other context (one ncclBroadcast of the repacked arena).
    pub fn weights_broadcast(&mut self, root: u32) -> Result<(), HipError> {
        match unsafe { sys::infur_group_weights_broadcast(self.raw, root) } { sys::INFUR_OK => Ok(()), rc => Err(self.err(rc)) }
    }
    /// scale -> model -> decode for a batch of independent frames, contiguous slices per context, masks in frame order.
    pub fn batch_advance(&mut self, frames: &[BgrImage], factor: f32, masks: &mut Vec<ColorImage>) -> Result<(), HipError> {
        masks.clear();
        for f in frames {
            let (mut ow, mut oh) = (0u32, 0u32);
            let rc = unsafe { sys::infur_scale_out_dims(f.width(), f.height(), factor, &mut ow, &mut oh) };
            if rc != sys::INFUR_OK { return Err(HipError::status(rc)); }
            masks.push(ColorImage::new([ow as usize, oh as usize], Color32::BLACK));
        }
        let ins: Vec<*const u8> = frames.iter().map(|f| f.as_raw().as_ptr()).collect();
        let outs: Vec<*mut u8> = masks.iter_mut().map(|m| m.pixels.as_mut_ptr() as *mut u8).collect();
        let ws: Vec<u32> = frames.iter().map(|f| f.width()).collect();
        let hs: Vec<u32> = frames.iter().map(|f| f.height()).collect();
        let caps: Vec<usize> = masks.iter().map(|m| m.pixels.len() * 4).collect();
        let rc = unsafe {
            sys::infur_group_batch_advance(self.raw, ins.as_ptr(), ws.as_ptr(), hs.as_ptr(), frames.len() as u32, factor,
                                           sys::INFUR_SCALE_NEAREST, outs.as_ptr(), caps.as_ptr(), std::ptr::null_mut(), std::ptr::null_mut())
        };
        if rc == sys::INFUR_OK { Ok(()) } else { Err(self.err(rc)) }
    }
    pub fn len(&self) -> usize { self.ctxs.len() }
}
impl Drop for HipGroup {
    fn drop(&mut self) { unsafe { sys::infur_group_destroy(self.raw) } }
}

// ---------------------------------------------------------------- ColorCode (decode_predict.rs:38-84)
pub struct HipColorCode { ctx: Rc<Ctx> }
impl HipColorCode { pub fn new(ctx: Rc<Ctx>) -> Self { Self { ctx } } }
impl Processor for HipColorCode {
    type Command = ();
    type ControlError = ();
    type Input = Array3<f32>;
    type Output = Option<ColorImage>;
    type ProcessResult = ();

    fn control(&mut self, _cmd: ()) -> Result<&mut Self, ()> { Ok(self) }
    fn is_dirty(&self) -> bool { false }
    fn advance(&mut self, inp: &Array3<f32>, out: &mut Option<ColorImage>) {
        let s = inp.shape();
        let (k, h, w) = (s[0], s[1], s[2]);
        let img = out.get_or_insert_with(|| ColorImage::new([w, h], Color32::BLACK));
        if img.width() != w || img.height() != h { *img = ColorImage::new([w, h], Color32::BLACK); }
        let std = inp.as_standard_layout();
        // Color32 is #[repr(C)] [u8; 4] premultiplied r,g,b,a: exactly the bytes the kernel writes
        unsafe {
            sys::infur_colorcode(self.ctx.0, std.as_ptr(), k as u32, h as u32, w as u32,
                                 img.pixels.as_mut_ptr() as *mut u8);
        }
    }
}

// ---------------------------------------------------------------- streaming ring with zero-copy slots (main.rs:27-99,105; ABI 5)
/// The bounded queue of frames in flight (`sync_channel(2)`, main.rs:105) over `infur_stream_*`.  `next_slot` / `commit` let the
/// decoder fill the ring's own pinned buffer in place -- what `ff-video/src/decoder.rs:156-165` does with its reused `BgrImage` --
/// and `view` / `release` hand the finished mask out of the pinned output slot: no pageable <-> pinned copy on either side.
/// UNTESTED like the rest of the crate (no Rust toolchain in the build image).
pub struct HipStream { raw: *mut sys::infur_stream, ctx: Rc<Ctx>, depth: u32, mode: u32 }
/// A finished frame in place in the ring's pinned output slot; the slot returns to the ring when the view is dropped.
pub struct MaskView<'a> { stream: &'a mut HipStream, pub frame_id: u64, pub size: [usize; 2], pub rgba: &'a [u8], pub scaled_bgr: Option<&'a [u8]> }
impl<'a> Drop for MaskView<'a> {
    fn drop(&mut self) { unsafe { sys::infur_stream_release(self.stream.raw); } }
}
impl HipStream {
    pub fn new(ctx: Rc<Ctx>, depth: u32) -> Result<Self, HipError> {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { sys::infur_stream_create(ctx.0, depth, &mut raw) };
        if rc != sys::INFUR_OK { return Err(HipError::from_ctx(&ctx, rc)); }
        Ok(Self { raw, ctx, depth, mode: sys::INFUR_SCALE_NEAREST })
    }
    pub fn pending(&self) -> u32 { unsafe { sys::infur_stream_pending(self.raw) } }
    pub fn depth(&self) -> u32 { self.depth }
    /// the next slot's pinned input buffer, `w * h * 3` bytes of packed bgr24 for the decoder to `read_exact` into
    pub fn next_slot(&mut self, w: u32, h: u32, factor: f32) -> Result<&mut [u8], HipError> {
        let mut p: *mut u8 = std::ptr::null_mut();
        let rc = unsafe { sys::infur_stream_acquire(self.raw, w, h, factor, &mut p) };
        if rc != sys::INFUR_OK { return Err(HipError::from_ctx(&self.ctx, rc)); }
        Ok(unsafe { std::slice::from_raw_parts_mut(p, (w as usize) * (h as usize) * 3) })
    }
    pub fn commit(&mut self, w: u32, h: u32, factor: f32, frame_id: u64) -> Result<(), HipError> {
        let rc = unsafe { sys::infur_stream_commit(self.raw, w, h, factor, self.mode, frame_id) };
        if rc != sys::INFUR_OK { return Err(HipError::from_ctx(&self.ctx, rc)); }
        Ok(())
    }
    /// the decoder found no frame for the slot `next_slot` lent out (end of input, read error): give it back uncommitted, so that
    /// copying `submit`s work again on this stream
    pub fn abandon(&mut self) { unsafe { sys::infur_stream_abandon(self.raw); } }
    /// waits for the oldest pending frame
    pub fn view(&mut self) -> Result<MaskView<'_>, HipError> {
        let (mut rgba, mut sc): (*const u8, *const u8) = (std::ptr::null(), std::ptr::null());
        let (mut id, mut ow, mut oh) = (0u64, 0u32, 0u32);
        let rc = unsafe { sys::infur_stream_collect_view(self.raw, &mut rgba, &mut sc, &mut id, &mut ow, &mut oh) };
        if rc != sys::INFUR_OK { return Err(HipError::from_ctx(&self.ctx, rc)); }
        let n = (ow as usize) * (oh as usize);
        // (a frame whose mask went straight to a caller-owned pinned buffer keeps no scaled copy: collect_view hands back NULL for it)
        let r = unsafe { std::slice::from_raw_parts(rgba, n * 4) };
        let s = if sc.is_null() { None } else { Some(unsafe { std::slice::from_raw_parts(sc, n * 3) }) };
        Ok(MaskView { stream: self, frame_id: id, size: [ow as usize, oh as usize], rgba: r, scaled_bgr: s })
    }
}
impl Drop for HipStream {
    fn drop(&mut self) { unsafe { sys::infur_stream_destroy(self.raw) } }
}
