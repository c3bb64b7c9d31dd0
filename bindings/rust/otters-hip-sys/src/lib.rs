//! Raw FFI declarations of `libotters_hip.so` — one item per item of `include/otters_hip.h` (ABI version 4).
//!
//! The reference crate (AtharvBhat/otters) has no FFI seam; this is the one a `hip` feature would bind.  It replaces
//! `VecStore::{new, add_vector, add_vectors, len}` (src/vec.rs:346-384), the body of `VecQueryPlan::collect`
//! (src/vec.rs:206-311) and the score + merge block of `MetaQueryPlan::collect` (src/meta.rs:671-709); see
//! `bindings/rust/patch/` for those bodies.
//!
//! Never compiled in the image this repository is built in (no rustc there).  `tests/test_rust_binding.py` parses this
//! file and holds every `#[repr(C)]` struct (field names, order, sizes, offsets), every constant and every `extern "C"`
//! prototype (name, arity, argument and return types) against the C header and against the layout a C11 compiler
//! reports for it (`tests/c/abi_layout`), so the two cannot drift apart unnoticed.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const OTT_ABI_VERSION: c_int = 4;
pub const OTT_COMM_ID_BYTES: usize = 128;

// ott_status
pub const OTT_OK: c_int = 0;
pub const OTT_ERR_INVALID: c_int = -1;
pub const OTT_ERR_HIP: c_int = -2;
pub const OTT_ERR_OOM: c_int = -3;
pub const OTT_ERR_UNSUPPORTED: c_int = -4;

// ott_metric (src/vec.rs:11-16)
pub const OTT_METRIC_COSINE: u32 = 0;
pub const OTT_METRIC_EUCLIDEAN: u32 = 1;
pub const OTT_METRIC_DOT: u32 = 2;
// ott_take (src/vec.rs:18-22)
pub const OTT_TAKE_MIN: u32 = 0;
pub const OTT_TAKE_MAX: u32 = 1;
// ott_cmp (src/vec.rs:24-31; NONE = no filter_criteria)
pub const OTT_CMP_NONE: u32 = 0;
pub const OTT_CMP_LT: u32 = 1;
pub const OTT_CMP_GT: u32 = 2;
pub const OTT_CMP_LTE: u32 = 3;
pub const OTT_CMP_GTE: u32 = 4;
pub const OTT_CMP_EQ: u32 = 5;
// ott_op (src/expr.rs:83-91)
pub const OTT_OP_EQ: u32 = 0;
pub const OTT_OP_NEQ: u32 = 1;
pub const OTT_OP_LT: u32 = 2;
pub const OTT_OP_LTE: u32 = 3;
pub const OTT_OP_GT: u32 = 4;
pub const OTT_OP_GTE: u32 = 5;
// ott_dtype (src/type_utils.rs:11-19; String columns stay host-side)
pub const OTT_DT_INT32: u32 = 0;
pub const OTT_DT_INT64: u32 = 1;
pub const OTT_DT_FLOAT32: u32 = 2;
pub const OTT_DT_FLOAT64: u32 = 3;
pub const OTT_DT_DATETIME: u32 = 5;
// ott_mode: MERGED = the reference's semantics (src/vec.rs:217-219), PER_QUERY = extension
pub const OTT_MODE_MERGED: u32 = 0;
pub const OTT_MODE_PER_QUERY: u32 = 1;
// ott_path
pub const OTT_PATH_AUTO: u32 = 0;
pub const OTT_PATH_EXACT: u32 = 1;
pub const OTT_PATH_MFMA: u32 = 2;
// ott_reduce: horizontal-sum order of wide::f32x8::reduce_add
pub const OTT_REDUCE_AVX: u32 = 0;
pub const OTT_REDUCE_SEQ4: u32 = 1;

/// `SearchResult` (src/vec.rs:34-38) + the query that scored it; `index` is the global row.
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct ott_hit {
    pub index: u64,
    pub score: f32,
    pub query: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ott_query_desc {
    pub queries: *const f32,
    pub nq: u32,
    pub metric: u32,
    pub take: u32,
    pub filter_cmp: u32,
    pub filter_thr: f32,
    pub mode: u32,
    pub k: u64,
    pub chunk_mask: *const u64,
    pub row_mask: *const u64,
    pub row_mask_bits: u64,
    pub use_device_row_mask: u32,
    pub path: u32,
}

/// `MetaQueryStats` (src/meta.rs:832-842) plus device-side facts.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct ott_stats {
    pub total_chunks: u64,
    pub pruned_chunks: u64,
    pub evaluated_chunks: u64,
    pub vectors_compared: u64,
    pub prune_ns: u64,
    pub score_ns: u64,
    pub merge_ns: u64,
    pub total_ns: u64,
    pub bytes_scanned: u64,
    pub path_used: u32,
    pub passes: u32,
    pub rescored: u64,
    pub retries: u32,
    pub refined: u32,
    pub err_ratio_max: f32,
    pub gate_failed: u32,
    pub bound_violations: u32,
    pub i8_refined: u32,
    pub exchange_ns: u64,
}

/// One leaf of a compiled CNF filter (`ColumnFilter::Numeric`, src/expr.rs:199-205).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ott_leaf {
    pub column: u32,
    pub op: u32,
    pub clause: u32,
    pub reserved: u32,
    pub lit_i64: i64,
    pub lit_f64: f64,
}

#[repr(C)]
pub struct ott_store {
    _opaque: [u8; 0],
}

#[repr(C)]
pub struct ott_comm {
    _opaque: [u8; 0],
}

pub type ott_allgather_fn = Option<unsafe extern "C" fn(user: *mut c_void, send: *const c_void, recv: *mut c_void, bytes: u64) -> c_int>;

extern "C" {
    pub fn ott_abi_version() -> c_int;
    pub fn ott_last_error() -> *const c_char;
    pub fn ott_device_count(out: *mut c_int) -> c_int;

    pub fn ott_store_create(dim: u32, device: c_int, out: *mut *mut ott_store) -> c_int;
    pub fn ott_store_create_multi(dim: u32, n_dev: u32, dev_ids: *const c_int, out: *mut *mut ott_store) -> c_int;
    pub fn ott_multi_plan(n_rows: u64, chunk_size: u64, n_dev: u32, out_first_rows: *mut u64) -> c_int;
    pub fn ott_store_shard_count(s: *const ott_store) -> c_int;
    pub fn ott_store_shard_info(s: *const ott_store, shard: u32, device: *mut c_int, first_row: *mut u64, n_rows: *mut u64) -> c_int;
    pub fn ott_store_transport(s: *const ott_store) -> *const c_char;
    pub fn ott_store_destroy(s: *mut ott_store) -> c_int;
    pub fn ott_store_reserve(s: *mut ott_store, n_rows: u64) -> c_int;
    pub fn ott_store_append(s: *mut ott_store, rows_host: *const f32, n_rows: u64) -> c_int;
    pub fn ott_store_append_device(s: *mut ott_store, rows_dev: *const c_void, n_rows: u64) -> c_int;
    pub fn ott_store_append_random(s: *mut ott_store, n_rows: u64, seed: u64) -> c_int;
    pub fn ott_store_append_clustered(s: *mut ott_store, n_rows: u64, seed: u64, n_clusters: u32, spread: f32, aniso: f32) -> c_int;
    pub fn ott_store_write_rows(s: *mut ott_store, first_row: u64, rows_host: *const f32, n_rows: u64) -> c_int;
    pub fn ott_store_len(s: *const ott_store) -> u64;
    pub fn ott_store_dim(s: *const ott_store) -> u32;
    pub fn ott_store_device(s: *const ott_store) -> c_int;
    pub fn ott_store_set_chunk_size(s: *mut ott_store, chunk_size: u64) -> c_int;
    pub fn ott_store_set_batch_image(s: *mut ott_store, enabled: c_int) -> c_int;
    pub fn ott_store_prepare_batch(s: *mut ott_store) -> c_int;
    pub fn ott_store_batch_ready(s: *const ott_store) -> c_int;
    pub fn ott_store_set_option(s: *mut ott_store, name: *const c_char, value: i64) -> c_int;
    pub fn ott_store_set_base_offset(s: *mut ott_store, base: u64) -> c_int;
    pub fn ott_store_set_reduce_order(s: *mut ott_store, reduce: u32) -> c_int;
    pub fn ott_store_read_rows(s: *const ott_store, first_row: u64, n_rows: u64, out_host: *mut f32) -> c_int;
    pub fn ott_store_read_inv_norms(s: *const ott_store, first_row: u64, n_rows: u64, out_host: *mut f32) -> c_int;
    pub fn ott_store_add_column(s: *mut ott_store, dtype: u32, values_host: *const c_void, nulls: *const u64, n: u64, out_column_id: *mut u32) -> c_int;
    pub fn ott_store_zone_stats(s: *mut ott_store, column: u32, chunk_size: u64, out_min: *mut c_void, out_max: *mut c_void, out_non_null: *mut u64) -> c_int;
    pub fn ott_store_eval_row_mask(s: *mut ott_store, leaves: *const ott_leaf, n_leaves: u32, n_clauses: u32, out_host: *mut u64) -> c_int;

    pub fn ott_query(s: *mut ott_store, d: *const ott_query_desc, out: *mut ott_hit, cap: u64, n_out: *mut u64, n_per_query: *mut u64, stats: *mut ott_stats) -> c_int;
    pub fn ott_query_device(s: *mut ott_store, d: *const ott_query_desc, out_dev: *mut c_void, cap: u64, n_out_dev: *mut c_void, stats: *mut ott_stats) -> c_int;
    pub fn ott_store_sync(s: *mut ott_store) -> c_int;
    pub fn ott_store_stream(s: *mut ott_store) -> *mut c_void;
    pub fn ott_merge_hits_device(s: *mut ott_store, lists_dev: *const c_void, n_lists: u64, list_len: u64, take: u32, k: u64, out_host: *mut ott_hit, n_out: *mut u64) -> c_int;
    pub fn ott_merge_hits_device_grouped(s: *mut ott_store, lists_dev: *const c_void, n_lists: u64, n_groups: u64, list_len: u64, take: u32, k: u64, out_host: *mut ott_hit, n_out: *mut u64, n_per_group: *mut u64) -> c_int;

    pub fn ott_comm_unique_id(id_out: *mut c_void) -> c_int;
    pub fn ott_comm_create(unique_id: *const c_void, rank: c_int, world: c_int, device: c_int, out: *mut *mut ott_comm) -> c_int;
    pub fn ott_comm_create_host(rank: c_int, world: c_int, r#fn: ott_allgather_fn, user: *mut c_void, out: *mut *mut ott_comm) -> c_int;
    pub fn ott_comm_destroy(c: *mut ott_comm) -> c_int;
    pub fn ott_comm_rank(c: *const ott_comm) -> c_int;
    pub fn ott_comm_world(c: *const ott_comm) -> c_int;
    pub fn ott_comm_transport(c: *const ott_comm) -> *const c_char;
    pub fn ott_comm_info(c: *const ott_comm, nranks: *mut c_int, version: *mut c_int) -> c_int;
    pub fn ott_comm_set_timeout_ms(c: *mut ott_comm, timeout_ms: i64) -> c_int;
    pub fn ott_comm_all_gather_host(c: *mut ott_comm, send_host: *const c_void, recv_host: *mut c_void, bytes: u64) -> c_int;
    pub fn ott_query_sharded(s: *mut ott_store, c: *mut ott_comm, d: *const ott_query_desc, out: *mut ott_hit, cap: u64, n_out: *mut u64, n_per_query: *mut u64, stats: *mut ott_stats) -> c_int;
}

/// The thread-local message behind the last non-zero status returned on this thread.
pub fn last_error() -> String {
    unsafe {
        let p = ott_last_error();
        if p.is_null() {
            String::new()
        } else {
            std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }
}

/// `Err(last_error())` for a non-zero status: the shape of the reference's `Result<_, String>` (src/vec.rs:206).
pub fn check(rc: c_int) -> Result<(), String> {
    if rc == OTT_OK {
        Ok(())
    } else {
        Err(last_error())
    }
}

// The layout the library was compiled with, restated so that rustc refuses to build a binding that drifted
// (the same numbers tests/c/abi_layout.c `_Static_assert`s on the C side).
const _: () = {
    assert!(std::mem::size_of::<ott_hit>() == 16);
    assert!(std::mem::size_of::<ott_query_desc>() == 72);
    assert!(std::mem::size_of::<ott_stats>() == 120);
    assert!(std::mem::size_of::<ott_leaf>() == 32);
};

#[cfg(test)]
mod tests {
    use super::*;

    #[test]
    fn library_and_binding_agree_on_the_abi_version() {
        assert_eq!(unsafe { ott_abi_version() }, OTT_ABI_VERSION);
    }
}
