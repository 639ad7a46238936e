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
pub struct ModelInfo { pub input_names: Vec<String>, pub input0_dtype: String, pub output_names: Vec<String> }

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
        })
    }
}
impl Processor for HipModel {
    type Command = ModelCmd;
    type ControlError = HipError;
    type Input = BgrImage;
    type Output = Vec<ArrayD<f32>>;
    type ProcessResult = Result<(), HipError>;

    fn control(&mut self, cmd: ModelCmd) -> Result<&mut Self, HipError> {
        let ModelCmd::Load(path) = cmd; // "" unloads inside the library (predict_onnx.rs:310-312)
        let c = CString::new(path).map_err(|_| HipError { code: 9, msg: "path contains NUL".into() })?;
        match unsafe { sys::infur_model_load(self.ctx.0, c.as_ptr()) } {
            sys::INFUR_OK => Ok(self),
            rc => Err(self.ctx.err(rc)),
        }
    }
    fn is_dirty(&self) -> bool { false }
    fn advance(&mut self, img: &BgrImage, out: &mut Vec<ArrayD<f32>>) -> Result<(), HipError> {
        let mi = match self.raw_info() { Some(mi) => mi, None => return Ok(()) }; // no session: out untouched
        let (k, h, w) = (mi.num_classes as usize, img.height() as usize, img.width() as usize);
        let (mut o, mut a) = (vec![0f32; k * h * w], vec![0f32; k * h * w]);
        let mut n = 0u32;
        let rc = unsafe {
            sys::infur_model_advance(self.ctx.0, img.as_raw().as_ptr(), w as u32, h as u32,
                                     o.as_mut_ptr(), a.as_mut_ptr(), &mut n)
        };
        if rc != sys::INFUR_OK { return Err(self.ctx.err(rc)); }
        out.clear();
        out.push(ArrayD::from_shape_vec(IxDyn(&[k, h, w]), o).expect("shape"));
        out.push(ArrayD::from_shape_vec(IxDyn(&[k, h, w]), a).expect("shape"));
        Ok(())
    }
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
