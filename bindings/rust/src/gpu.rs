//! `extern "C"` declarations, 1:1 with `include/soundsym_amd.h` (ABI version 2).
//!
//! What each call replaces in the crate (`src/sound.rs`): see the header's comments and `INTEGRATION.md`
//! section 3.  Nothing here allocates or frees Rust memory on the other side of the boundary; the library
//! copies caller buffers at create time and never unwinds into the caller.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_void};

#[repr(C)] pub struct SsymCtx { _p: [u8; 0] }
#[repr(C)] pub struct SsymDict { _p: [u8; 0] }
#[repr(C)] pub struct SsymQueries { _p: [u8; 0] }
#[repr(C)] pub struct SsymSamples { _p: [u8; 0] }
#[repr(C)] pub struct SsymComm { _p: [u8; 0] }
#[repr(C)] pub struct SsymLocalGroup { _p: [u8; 0] }

pub const SSYM_ABI_VERSION: i32 = 3;

pub const SSYM_OK: i32 = 0;
pub const SSYM_E_INVALID: i32 = -1;
pub const SSYM_E_EMPTY_DICT: i32 = -2;   // the panic at src/sound.rs:369
pub const SSYM_E_NO_DEVICE: i32 = -3;
pub const SSYM_E_HIP: i32 = -4;
pub const SSYM_E_NOMEM: i32 = -5;
pub const SSYM_E_UNSUPPORTED: i32 = -6;
pub const SSYM_E_TIMEOUT: i32 = -7;      // a rank of the sharded match did not arrive; the communicator is aborted
pub const SSYM_E_COMM: i32 = -8;         // the communicator is dead: destroy it

pub const SSYM_METRIC_REFCOS: i32 = 0;   // the crate's own cosine_sim / at_distance, bit for bit
pub const SSYM_METRIC_DTW: i32 = 1;
pub const SSYM_DTYPE_F64: i32 = 0;       // Sound::mfccs() is Vec<f64>
pub const SSYM_DTYPE_F32: i32 = 1;

pub const SSYM_OUT_DEVICE: u32 = 1;
pub const SSYM_DTW_FORCE_EXACT: u32 = 2;
pub const SSYM_DTW_PRUNE: u32 = 4;
pub const SSYM_MFCC_PAD_TAIL: u32 = 4;
pub const SSYM_TOPK_MAX: u32 = 64;
pub const SSYM_NO_MATCH: u32 = 0xffff_ffff;
pub const SSYM_COMM_ID_BYTES: usize = 128;

#[repr(C)]
pub struct SsymConfig {
    pub struct_size: u32,
    pub device: i32,
    pub metric: i32,
    pub dtype: i32,
    pub band: i32,         // dtw only, -1 = none
    pub dtw_squared: i32,
    pub stream: *mut c_void,
    pub dtw_prune: i32,
    pub reserved: i32,
}

#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct SsymTimings {
    pub pack_ms: f32,
    pub main_ms: f32,
    pub select_ms: f32,
    pub refine_ms: f32,
    pub reduce_ms: f32,
    pub total_ms: f32,
    pub n_pairs: u64,
    pub n_refined: u64,
    pub main_launches: i32,
    pub used_filter: i32,
    pub prune_ms: f32,
    pub pruned: i32,
    pub n_filter_cells: u64,
    pub collective_ms: f32,
    pub attempts: i32,
    pub exact_redone: i32,
    pub refcos_filter: i32,
}

extern "C" {
    pub fn ssym_abi_version() -> i32;
    pub fn ssym_ctx_create(cfg: *const SsymConfig, out: *mut *mut SsymCtx) -> i32;
    pub fn ssym_ctx_destroy(ctx: *mut SsymCtx) -> i32;
    pub fn ssym_last_error(ctx: *const SsymCtx) -> *const c_char;
    pub fn ssym_ctx_synchronize(ctx: *mut SsymCtx) -> i32;
    pub fn ssym_get_timings(ctx: *const SsymCtx, out: *mut SsymTimings) -> i32;

    pub fn ssym_dict_create(ctx: *mut SsymCtx, feats: *const c_void, frame_offsets: *const u64,
                            n_segments: u32, dim: u32, out: *mut *mut SsymDict) -> i32;
    pub fn ssym_dict_create_device(ctx: *mut SsymCtx, feats_dev: *const c_void, frame_offsets: *const u64,
                                   n_segments: u32, dim: u32, out: *mut *mut SsymDict) -> i32;
    pub fn ssym_dict_append(ctx: *mut SsymCtx, dict: *mut SsymDict, feats: *const c_void,
                            frame_offsets: *const u64, n_segments: u32) -> i32;
    pub fn ssym_dict_size(dict: *const SsymDict, out_n_segments: *mut u32) -> i32;
    pub fn ssym_dict_destroy(ctx: *mut SsymCtx, dict: *mut SsymDict) -> i32;

    pub fn ssym_queries_create(ctx: *mut SsymCtx, feats: *const c_void, frame_offsets: *const u64,
                               n_targets: u32, dim: u32, out: *mut *mut SsymQueries) -> i32;
    pub fn ssym_queries_create_device(ctx: *mut SsymCtx, feats_dev: *const c_void, frame_offsets: *const u64,
                                      n_targets: u32, dim: u32, out: *mut *mut SsymQueries) -> i32;
    pub fn ssym_queries_destroy(ctx: *mut SsymCtx, q: *mut SsymQueries) -> i32;

    pub fn ssym_match_queries(ctx: *mut SsymCtx, dict: *const SsymDict, q: *const SsymQueries, distance: *const f64,
                              index_base: u32, out_idx: *mut u32, out_cost: *mut f64, flags: u32) -> i32;
    pub fn ssym_match_topk(ctx: *mut SsymCtx, dict: *const SsymDict, q: *const SsymQueries, distance: *const f64,
                           k: u32, index_base: u32, out_idx: *mut u32, out_cost: *mut f64, flags: u32) -> i32;
    pub fn ssym_match_batch(ctx: *mut SsymCtx, dict: *const SsymDict, tgt_feats: *const c_void,
                            tgt_frame_offsets: *const u64, n_targets: u32, distance: *const f64,
                            out_idx: *mut u32, out_cost: *mut f64) -> i32;
    pub fn ssym_match_one(ctx: *mut SsymCtx, dict: *const SsymDict, feats: *const c_void, n_frames: u64,
                          distance: f64, out_idx: *mut u32, out_cost: *mut f64) -> i32;
    pub fn ssym_chain(ctx: *mut SsymCtx, dict: *mut SsymDict, start_feats: *const c_void, start_frames: u64,
                      distances: *const f64, n_steps: u32, out_idx: *mut u32, out_cost: *mut f64) -> i32;
    pub fn ssym_pair_matrix(ctx: *mut SsymCtx, dict: *const SsymDict, q: *const SsymQueries, exact: i32,
                            out_matrix: *mut f64) -> i32;

    // source-sharded runs, exchange done by the caller (device pointers): filter / all-reduce(MIN) / finish / merge
    pub fn ssym_match_begin(ctx: *mut SsymCtx, dict: *const SsymDict, q: *const SsymQueries, distance: *const f64,
                            index_base: u32, bounds_dev: *mut f64) -> i32;
    pub fn ssym_match_finish(ctx: *mut SsymCtx, bounds_dev: *const f64, out_idx: *mut u32, out_cost: *mut f64,
                             flags: u32) -> i32;
    pub fn ssym_match_candidates(ctx: *mut SsymCtx, dict: *const SsymDict, q: *const SsymQueries,
                                 cost_dev: *mut f64) -> i32;
    pub fn ssym_match_begin_pruned(ctx: *mut SsymCtx, dict: *const SsymDict, q: *const SsymQueries, index_base: u32,
                                   cost_dev: *const f64, bounds_dev: *mut f64) -> i32;
    pub fn ssym_merge_shards(ctx: *mut SsymCtx, n_shards: u32, n_targets: u32, costs_dev: *const f64,
                             idx_dev: *const u32, out_idx_dev: *mut u32, out_cost_dev: *mut f64) -> i32;
    pub fn ssym_merge_shards_at(ctx: *mut SsymCtx, n_shards: u32, n_targets: u32, costs_dev: *const f64,
                                idx_dev: *const u32, distance: *const f64, out_idx_dev: *mut u32,
                                out_cost_dev: *mut f64) -> i32;

    // source-sharded runs, exchange done by the library: RCCL on the context's stream, one host sync per step
    pub fn ssym_comm_unique_id(out_id: *mut c_void /* SSYM_COMM_ID_BYTES */) -> i32;
    pub fn ssym_comm_create(ctx: *mut SsymCtx, id: *const c_void, rank: i32, world: i32, out: *mut *mut SsymComm) -> i32;
    pub fn ssym_comm_destroy(ctx: *mut SsymCtx, comm: *mut SsymComm) -> i32;
    pub fn ssym_match_sharded(ctx: *mut SsymCtx, comm: *mut SsymComm, dict: *const SsymDict, q: *const SsymQueries,
                              distance: *const f64, index_base: u32, out_idx: *mut u32, out_cost: *mut f64,
                              flags: u32) -> i32;

    // the ranks of ONE process (a thread per rank) without RCCL: host barriers around device copies -- what the
    // one-GPU tests use to run more than one rank (RCCL refuses two ranks on one device)
    pub fn ssym_comm_available() -> i32;
    pub fn ssym_comm_set_timeout(comm: *mut SsymComm, milliseconds: i64) -> i32;
    pub fn ssym_comm_is_dead(comm: *const SsymComm) -> i32;
    // test and measurement hooks: refuse (SSYM_E_UNSUPPORTED) unless the process runs with SSYM_TEST_HOOKS=1
    pub fn ssym_comm_inject_fault(comm: *mut SsymComm, phase: i32, kind: i32) -> i32;
    pub fn ssym_comm_replay_bounds(comm: *mut SsymComm, bounds_dev: *const f64, n: u32) -> i32;
    pub fn ssym_local_group_create(world: i32, out: *mut *mut SsymLocalGroup) -> i32;
    pub fn ssym_local_group_destroy(group: *mut SsymLocalGroup) -> i32;
    pub fn ssym_comm_create_local(ctx: *mut SsymCtx, group: *mut SsymLocalGroup, rank: i32, out: *mut *mut SsymComm) -> i32;

    // reconstruction tail (src/sound.rs:456-465, 475-480, 139)
    pub fn ssym_samples_create(ctx: *mut SsymCtx, samples: *const f64, sample_offsets: *const u64, n_sounds: u32,
                               out: *mut *mut SsymSamples) -> i32;
    pub fn ssym_samples_destroy(ctx: *mut SsymCtx, s: *mut SsymSamples) -> i32;
    pub fn ssym_reconstruct(ctx: *mut SsymCtx, s: *const SsymSamples, idx: *const u32, out_offsets: *const u64,
                            n_targets: u32, out_samples: *mut f64, out_pcm32: *mut i32) -> i32;

    // feature front-end (own MFCC definition -- parity with vox_box unpinned)
    pub fn ssym_mfcc_num_frames(n_samples: u64, flags: u32, out_frames: *mut u64) -> i32;
    pub fn ssym_mfcc(ctx: *mut SsymCtx, samples: *const f64, n_samples: u64, sample_rate: f64, n_coeffs: u32,
                     f_lo: f64, f_hi: f64, flags: u32, out_mfccs: *mut f64, out_mean: *mut f64) -> i32;
}

/// `Err(message)` for any status but SSYM_OK; SSYM_E_EMPTY_DICT keeps the crate's behaviour (a panic, :369).
pub unsafe fn check(ctx: *const SsymCtx, rc: i32) -> Result<(), String> {
    if rc == SSYM_OK {
        return Ok(());
    }
    if rc == SSYM_E_EMPTY_DICT {
        panic!("index out of bounds: the len is 0 but the index is 0");   // what src/sound.rs:369 does today
    }
    let msg = ssym_last_error(ctx);
    let text = if msg.is_null() { String::new() } else { std::ffi::CStr::from_ptr(msg).to_string_lossy().into_owned() };
    Err(format!("soundsym_amd error {}: {}", rc, text))
}
