//! NOT COMPILED: the build image has no Rust toolchain.  A sketch of `Decoder` (the crate's `src/decoder.rs`)
//! with the two hot loops handed to the GPU through `gpu.rs`: same public API -- `Decoder::new(file)`,
//! `decode() -> Result<(), DecodeError>`, `out.ppm` written to the working directory -- with the container
//! parse kept in Rust.  Written from the interface description in include/zj.h; it is the shape of the
//! change, not a tested patch.
//!
//! What changes against the crate's `decode()`:
//!   * DQT: the 64 bytes are kept as read (file order is zig-zag order) per table id 0..3.
//!   * DHT: BITS and HUFFVAL are kept as read per [table id][is_dc]; no code table is built on the host.
//!   * SOF0: sampling factors and quantisation selectors are kept instead of dropped.
//!   * DRI (FFDD) is read; every other segment with a length (all APPn, COM, ...) is skipped by its length.
//!   * SOS: the component selectors are read; the rest of the file is the entropy-coded segment up to the
//!     first marker that is neither FF00 nor RSTn; one `zj_decode_batch` call replaces the MCU loop and to_rgb.
//!   * Errors: `DecodeError` gains variants next to `Io` (additive); nothing panics on malformed input.

use std::fs::File;
use std::io::{BufReader, Read, Write};

use crate::gpu::*;

#[derive(Debug)]
pub enum DecodeError {
    Io(std::io::Error),
    /// The container is malformed or outside baseline (what the CPU path answers with a panic).
    Format(&'static str),
    /// The GPU path failed: return code of the C ABI and its message.
    Device(i32, String),
    /// The entropy-coded data is damaged: ZJ_ST_* bits of the image.
    Corrupt(u32),
}

impl From<std::io::Error> for DecodeError {
    fn from(e: std::io::Error) -> Self { Self::Io(e) }
}

pub struct Decoder {
    reader: BufReader<File>,
    w: u16,
    h: u16,
    /// The GPU context, created on the first `decode()` and kept for the decoder's life: creating one costs far more
    /// than a decode (stream, first-use initialisation of the runtime), so a caller that decodes file after file reuses
    /// a `Decoder` -- see `reset()` -- or hands one `Gpu` to many decoders with `with_gpu()`.
    gpu: Option<std::rc::Rc<Gpu>>,
}

fn be16(b: &[u8], at: usize) -> Result<usize, DecodeError> {
    if at + 2 > b.len() { return Err(DecodeError::Format("truncated segment")); }
    Ok(((b[at] as usize) << 8) | b[at + 1] as usize)
}

impl Decoder {
    pub fn new(file: File) -> Self {
        Decoder { reader: BufReader::new(file), w: 0, h: 0, gpu: None }
    }

    /// A decoder that shares an existing GPU context (one per GPU and thread).
    pub fn with_gpu(file: File, gpu: std::rc::Rc<Gpu>) -> Self {
        Decoder { reader: BufReader::new(file), w: 0, h: 0, gpu: Some(gpu) }
    }

    /// Point the decoder at another file, keeping its GPU context.
    pub fn reset(&mut self, file: File) {
        self.reader = BufReader::new(file);
        self.w = 0;
        self.h = 0;
    }

    fn gpu(&mut self) -> Result<std::rc::Rc<Gpu>, DecodeError> {
        if self.gpu.is_none() {
            let g = Gpu::new(0).ok_or_else(|| DecodeError::Device(ZJ_E_DEVICE, "no usable GPU".into()))?;
            self.gpu = Some(std::rc::Rc::new(g));
        }
        Ok(self.gpu.as_ref().unwrap().clone())
    }

    pub fn decode(&mut self) -> Result<(), DecodeError> {
        let mut file = Vec::new();
        self.reader.read_to_end(&mut file)?;
        let mut out_file = File::create("out.ppm")?;

        let mut qt = [[0u8; 64]; 4];
        let mut qt_set = [false; 4];
        let mut ht = [[ZjHuff { counts: [0; 16], symbols: [0; 256] }; 2]; 2]; // [id][is_dc]
        let mut ht_set = [[false; 2]; 2];
        let (mut ncomp, mut hs, mut vs, mut tq) = (0u8, [1u8; 3], [1u8; 3], [0u8; 3]);
        let (mut td, mut ta) = ([0u8; 3], [0u8; 3]);
        let mut restart_interval = 0u16;

        let mut pos = 0usize;
        loop {
            let marker = be16(&file, pos)?;
            pos += 2;
            if marker >> 8 != 0xff { return Err(DecodeError::Format("marker expected")); }
            let m = marker & 0xff;
            if m == 0xd8 { continue; }                                   // SOI
            if m == 0xd9 { return Err(DecodeError::Format("no scan")); } // EOI before SOS
            let len = be16(&file, pos)?;
            if len < 2 || pos + len > file.len() { return Err(DecodeError::Format("segment length")); }
            let seg = &file[pos + 2..pos + len];
            match m {
                0xdb => { // DQT, several tables per segment allowed
                    let mut p = 0;
                    while p < seg.len() {
                        let id = (seg[p] & 0x0f) as usize;
                        if seg[p] & 0xf0 != 0 || id > 3 || p + 65 > seg.len() { return Err(DecodeError::Format("DQT")); }
                        qt[id].copy_from_slice(&seg[p + 1..p + 65]);
                        qt_set[id] = true;
                        p += 65;
                    }
                }
                0xc4 => { // DHT, several tables per segment allowed
                    let mut p = 0;
                    while p < seg.len() {
                        let (id, is_dc) = ((seg[p] & 0x0f) as usize, (seg[p] >> 4) & 1 == 0);
                        if seg[p] & 0xe0 != 0 || id > 1 || p + 17 > seg.len() { return Err(DecodeError::Format("DHT")); }
                        let t = &mut ht[id][is_dc as usize];
                        t.counts.copy_from_slice(&seg[p + 1..p + 17]);
                        let n: usize = t.counts.iter().map(|&c| c as usize).sum();
                        if n == 0 || n > 256 || p + 17 + n > seg.len() { return Err(DecodeError::Format("DHT")); }
                        t.symbols[..n].copy_from_slice(&seg[p + 17..p + 17 + n]);
                        ht_set[id][is_dc as usize] = true;
                        p += 17 + n;
                    }
                }
                0xc0 => { // SOF0
                    if seg.len() < 6 || seg[0] != 8 { return Err(DecodeError::Format("SOF0")); }
                    self.h = be16(seg, 1)? as u16;
                    self.w = be16(seg, 3)? as u16;
                    ncomp = seg[5];
                    if (ncomp != 1 && ncomp != 3) || seg.len() < 6 + 3 * ncomp as usize { return Err(DecodeError::Format("components")); }
                    for c in 0..ncomp as usize {
                        hs[c] = seg[7 + 3 * c] >> 4;
                        vs[c] = seg[7 + 3 * c] & 15;
                        tq[c] = seg[8 + 3 * c];
                    }
                    if ncomp == 1 { hs[0] = 1; vs[0] = 1; }
                }
                0xc1..=0xcf if m != 0xc8 && m != 0xcc => return Err(DecodeError::Format("not baseline")),
                0xdd => restart_interval = be16(seg, 0)? as u16, // DRI
                0xda => { // SOS: selectors, then the entropy-coded segment
                    if seg.is_empty() || seg[0] != ncomp || seg.len() < 1 + 2 * ncomp as usize { return Err(DecodeError::Format("SOS")); }
                    for c in 0..ncomp as usize {
                        td[c] = seg[2 + 2 * c] >> 4;
                        ta[c] = seg[2 + 2 * c] & 15;
                        if td[c] > 1 || ta[c] > 1 || !ht_set[td[c] as usize][1] || !ht_set[ta[c] as usize][0] || !qt_set[tq[c] as usize] {
                            return Err(DecodeError::Format("table not defined"));
                        }
                    }
                    let scan0 = pos + len;
                    let mut end = scan0;
                    while end + 1 < file.len() {
                        if file[end] == 0xff && file[end + 1] != 0 && !(0xd0..=0xd7).contains(&file[end + 1]) && file[end + 1] != 0xff { break; }
                        end += 1;
                    }
                    if end + 1 >= file.len() { end = file.len(); }
                    let img = ZjImage {
                        scan: file[scan0..].as_ptr(), scan_len: end - scan0,
                        width: self.w, height: self.h, ncomp,
                        hsamp: hs, vsamp: vs, tq, td, ta, restart_interval,
                        qt: [0, 1, 2, 3].map(|i| if qt_set[i] { qt[i].as_ptr() } else { core::ptr::null() }),
                        ht: [0, 1].map(|id| [0, 1].map(|k| if ht_set[id][k] { &ht[id][k] as *const ZjHuff } else { core::ptr::null() })),
                    };
                    let mut buf = vec![0u8; 3 * self.w as usize * self.h as usize];
                    let out = ZjOutput { rgb: buf.as_mut_ptr(), rgb_stride: 0, qcoef: core::ptr::null_mut(), dqcoef: core::ptr::null_mut(), mem: ZJ_MEM_HOST };
                    let gpu = self.gpu()?; // created once, kept in the decoder
                    let mut status = 0u32;
                    let rc = unsafe { zj_decode_batch(gpu.raw(), &img, 1, &out, &mut status) };
                    if rc != ZJ_OK { return Err(DecodeError::Device(rc, gpu.last_error())); }
                    if status != 0 { return Err(DecodeError::Corrupt(status)); }
                    write!(out_file, "P6\n{} {}\n255\n", self.w, self.h)?;
                    out_file.write_all(&buf)?;
                    return Ok(());
                }
                _ => {} // APPn, COM and anything else with a length: skipped
            }
            pos += len;
        }
    }
}
