//! NOT COMPILED: the build image has no Rust toolchain (no `rustc`, no `cargo`).  This file is the binding a
//! maintainer of the `zen-jpeg` crate would add as `src/gpu.rs`; it has been checked by eye against
//! `include/zj.h` only.  Link with `-L <dir of libzj.so> -l zj` (a `build.rs` with
//! `println!("cargo:rustc-link-lib=dylib=zj")`).
//!
//! Field order, widths and padding follow the C declarations exactly (`#[repr(C)]`); every pointer is borrowed
//! for the duration of the call that takes it.

use core::ffi::{c_char, c_int, c_void};

pub const ZJ_OK: c_int = 0;
pub const ZJ_E_INVALID: c_int = -1;
pub const ZJ_E_FORMAT: c_int = -2;
pub const ZJ_E_UNSUPPORTED: c_int = -3;
pub const ZJ_E_DEVICE: c_int = -4;
pub const ZJ_E_NOMEM: c_int = -5;
pub const ZJ_E_IO: c_int = -6;
pub const ZJ_E_CHAIN: c_int = -7;

pub const ZJ_ST_BAD_CODE: u32 = 1;
pub const ZJ_ST_COEF_OVERFLOW: u32 = 2;
pub const ZJ_ST_TRUNCATED: u32 = 4;
pub const ZJ_ST_UNSUPPORTED: u32 = 8;

pub const ZJ_MEM_HOST: c_int = 0;
pub const ZJ_MEM_DEVICE: c_int = 1;

/// One DHT table as stored in the file: BITS and HUFFVAL.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct ZjHuff {
    pub counts: [u8; 16],
    pub symbols: [u8; 256],
}

/// The state `Decoder::decode` holds when it reaches the SOS arm.
#[repr(C)]
pub struct ZjImage {
    pub scan: *const u8,
    pub scan_len: usize,
    pub width: u16,
    pub height: u16,
    pub ncomp: u8,
    pub hsamp: [u8; 3],
    pub vsamp: [u8; 3],
    pub tq: [u8; 3],
    pub td: [u8; 3],
    pub ta: [u8; 3],
    pub restart_interval: u16,
    pub qt: [*const u8; 4],
    /// `[table id][0 = AC, 1 = DC]`, the indexing of `huffman_table` in `Decoder::decode`.
    pub ht: [[*const ZjHuff; 2]; 2],
}

#[repr(C)]
pub struct ZjOutput {
    pub rgb: *mut u8,
    pub rgb_stride: usize,
    pub qcoef: *mut i16,
    pub dqcoef: *mut i16,
    pub mem: c_int,
}

/// What a part of a stream that is split over several GPUs tells the other parts (all-gathered by the caller).
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct ZjRangeReport {
    pub start_delta: i32,
    pub start_k: u32,
    pub end_delta: i32,
    pub end_k: u32,
    pub mcus: u32,
    pub dc_sum: [i16; 3],
    pub reserved: u16,
    pub reserved2: u32,
}

pub enum ZjCtx {}
pub enum ZjBatch {}
pub enum ZjWorkspace {}

#[link(name = "zj")]
extern "C" {
    pub fn zj_device_count() -> c_int;
    pub fn zj_create(device: c_int, stream: *mut c_void) -> *mut ZjCtx;
    pub fn zj_destroy(ctx: *mut ZjCtx);
    pub fn zj_last_error(ctx: *mut ZjCtx) -> *const c_char;

    /// One-shot: plan, decode, finish, copy out.  What `Decoder::decode` calls.
    pub fn zj_decode_batch(ctx: *mut ZjCtx, imgs: *const ZjImage, n: usize, outs: *const ZjOutput, status: *mut u32) -> c_int;
    /// The same with device outputs and no device allocation (fixed scratch of 64 MiB per GPU).
    pub fn zj_decode_batch_noalloc(ctx: *mut ZjCtx, imgs: *const ZjImage, n: usize, rgb: *const *mut u8, stride: *const usize,
                                   status: *mut u32) -> c_int;

    // Batches that stay resident (a data loader decoding many images per call).
    pub fn zj_batch_create(ctx: *mut ZjCtx, imgs: *const ZjImage, n: usize, flags: u32) -> *mut ZjBatch;
    pub fn zj_batch_destroy(b: *mut ZjBatch);
    pub fn zj_batch_decode(b: *mut ZjBatch, rgb: *const *mut u8, stride: *const usize) -> c_int;
    // the two halves of zj_batch_decode; `stream` = NULL for the context's stream
    pub fn zj_batch_decode_entropy(b: *mut ZjBatch) -> c_int;
    pub fn zj_batch_decode_pixels(b: *mut ZjBatch, rgb: *const *mut u8, stride: *const usize, stream: *mut c_void) -> c_int;
    pub fn zj_batch_finish(b: *mut ZjBatch, status: *mut u32) -> c_int;
    pub fn zj_batch_counters(b: *mut ZjBatch, counters: *mut u32) -> c_int;
    pub fn zj_batch_export_coef(b: *mut ZjBatch, i: usize, qcoef: *mut i16, dqcoef: *mut i16, mem: c_int) -> c_int;

    // Sub-batches that share device memory.
    pub fn zj_workspace_estimate(imgs: *const ZjImage, n: usize, plan_bytes: *mut usize, scratch_bytes: *mut usize) -> c_int;
    pub fn zj_workspace_create(ctx: *mut ZjCtx, scratch_bytes: usize, plan_bytes: usize, nslots: u32) -> *mut ZjWorkspace;
    pub fn zj_workspace_destroy(ws: *mut ZjWorkspace);
    pub fn zj_batch_create_in(ws: *mut ZjWorkspace, slot: u32, imgs: *const ZjImage, n: usize, flags: u32) -> *mut ZjBatch;
    pub fn zj_batch_create_rows(ctx: *mut ZjCtx, imgs: *const ZjImage, n: usize, flags: u32, row0: *const u32, nrows: *const u32) -> *mut ZjBatch;

    // One image without restart markers over several GPUs, entropy front included: plan, front (fills this part's report),
    // all-gather of the reports by the caller, resolve (ZJ_E_CHAIN: the parts do not join up, use zj_batch_create_rows),
    // decode into the MCU rows [row0, row0 + nrows), zj_batch_finish.
    pub fn zj_batch_create_range(ctx: *mut ZjCtx, img: *const ZjImage, rank: u32, nranks: u32, lead_bytes: u32, flags: u32) -> *mut ZjBatch;
    pub fn zj_batch_range_front(b: *mut ZjBatch, report: *mut ZjRangeReport) -> c_int;
    pub fn zj_batch_range_resolve(b: *mut ZjBatch, all: *const ZjRangeReport, nranks: u32, mcu_first: *mut u32, mcu_count: *mut u32,
                                  row0: *mut u32, nrows: *mut u32) -> c_int;
    pub fn zj_batch_range_decode(b: *mut ZjBatch, rgb: *mut u8, stride: usize) -> c_int;

    // One restart-marked image cut into bands (one per GPU).
    pub fn zj_restart_offsets(img: *const ZjImage, offsets: *mut usize, cap: usize) -> usize;
    pub fn zj_image_band(img: *const ZjImage, row0: u32, nrows: u32, band: *mut ZjImage) -> c_int;
}

/// Owner of a `zj_ctx`; one per GPU and thread.
pub struct Gpu(*mut ZjCtx);

impl Gpu {
    pub fn new(device: i32) -> Option<Gpu> {
        let p = unsafe { zj_create(device, core::ptr::null_mut()) };
        if p.is_null() { None } else { Some(Gpu(p)) }
    }
    pub fn raw(&self) -> *mut ZjCtx { self.0 }
    pub fn last_error(&self) -> String {
        unsafe { std::ffi::CStr::from_ptr(zj_last_error(self.0)).to_string_lossy().into_owned() }
    }
}

impl Drop for Gpu {
    fn drop(&mut self) { unsafe { zj_destroy(self.0) } }
}
