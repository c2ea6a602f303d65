//! A `SoundDictionary` whose entries are split over the GPUs of one node (source-axis sharding), one rank
//! per GPU -- here one OS thread per rank inside one process; one process per rank works the same way with the
//! 128-byte id handed over by any IPC the host has.
//!
//! Rank g owns `sounds[lo_g .. hi_g]` (contiguous, ordered: the first-minimum rule of `at_distance`,
//! src/sound.rs:361-367, survives the merge because lower ranks hold lower indices) and ALL targets.
//! `ssym_match_sharded` does the whole step -- filter, RCCL all-reduce(MIN) of the per-target bounds,
//! selection and exact re-scoring, RCCL all-gather of (cost, global index), merge -- on the rank's stream and
//! returns the same complete answer on every rank.
use crate::gpu::*;
use std::os::raw::c_void;
use std::ptr;

/// [lo, hi) of rank `rank`: the first `n % world` ranks get one entry more (soundsym_amd/sharding.py shard_range).
pub fn shard_range(n: usize, world: usize, rank: usize) -> (usize, usize) {
    let (q, r) = (n / world, n % world);
    let lo = rank * q + rank.min(r);
    (lo, lo + q + if rank < r { 1 } else { 0 })
}

pub struct Rank {
    pub ctx: *mut SsymCtx,
    pub comm: *mut SsymComm,
    pub dict: *mut SsymDict,
    pub lo: u32,
}

/// Rank 0 draws the id; every rank then calls `Rank::new` with the same bytes (ncclCommInitRank is collective:
/// all `world` calls must be in flight together, hence one thread per rank).
pub fn unique_id() -> Result<[u8; SSYM_COMM_ID_BYTES], String> {
    let mut id = [0u8; SSYM_COMM_ID_BYTES];
    unsafe { check(ptr::null(), ssym_comm_unique_id(id.as_mut_ptr() as *mut c_void))? };
    Ok(id)
}

impl Rank {
    /// `feats` / `frame_offsets`: this rank's shard only (frame-major `Sound::mfccs()` back to back).
    pub unsafe fn new(device: i32, metric: i32, dtype: i32, id: &[u8; SSYM_COMM_ID_BYTES], rank: i32, world: i32,
                      feats: *const c_void, frame_offsets: &[u64], dim: u32, lo: u32) -> Result<Rank, String> {
        let cfg = SsymConfig { struct_size: std::mem::size_of::<SsymConfig>() as u32, device, metric, dtype, band: -1,
                               dtw_squared: 0, stream: ptr::null_mut(), dtw_prune: 0, reserved: 0 };
        let mut ctx = ptr::null_mut();
        check(ptr::null(), ssym_ctx_create(&cfg, &mut ctx))?;
        let mut comm = ptr::null_mut();
        check(ctx, ssym_comm_create(ctx, id.as_ptr() as *const c_void, rank, world, &mut comm))?;
        let mut dict = ptr::null_mut();
        check(ctx, ssym_dict_create(ctx, feats, frame_offsets.as_ptr(), (frame_offsets.len() - 1) as u32, dim, &mut dict))?;
        Ok(Rank { ctx, comm, dict, lo })
    }

    /// One batch of targets (the loop of clone_from_dictionary, src/sound.rs:451-455, or of morph_to with
    /// `distance`): every rank passes the SAME targets and gets the SAME global indices back.
    /// Failure: a rank whose work inside ssym_match_sharded fails still takes part, and EVERY rank gets that rank's
    /// status as its `Err`.  A rank that returns BEFORE the call (ssym_queries_create failing on this rank only) never
    /// arrives: its peers return SSYM_E_TIMEOUT after the communicator's deadline (ssym_comm_set_timeout) and their
    /// communicator is aborted (SSYM_E_COMM from then on): drop the Rank and build a new one.
    pub unsafe fn match_all(&self, tgt_feats: *const c_void, tgt_offsets: &[u64], dim: u32,
                            distance: Option<&[f64]>) -> Result<Vec<u32>, String> {
        let n = (tgt_offsets.len() - 1) as u32;
        let mut q = ptr::null_mut();
        check(self.ctx, ssym_queries_create(self.ctx, tgt_feats, tgt_offsets.as_ptr(), n, dim, &mut q))?;
        let mut idx = vec![0u32; n as usize];
        let rc = ssym_match_sharded(self.ctx, self.comm, self.dict, q, distance.map_or(ptr::null(), |d| d.as_ptr()),
                                    self.lo, idx.as_mut_ptr(), ptr::null_mut(), 0);
        ssym_queries_destroy(self.ctx, q);
        check(self.ctx, rc)?;
        Ok(idx)      // dict.sounds[idx[t]].clone() + the length fit of src/sound.rs:456-465, unchanged
    }
}

impl Drop for Rank {
    fn drop(&mut self) {
        unsafe {
            ssym_dict_destroy(self.ctx, self.dict);
            ssym_comm_destroy(self.ctx, self.comm);
            ssym_ctx_destroy(self.ctx);
        }
    }
}
