//! `cfg(feature = "hip")` bodies for src/meta.rs: ONE device store for all chunks instead of one `VecStore` per
//! `MetaChunk` (src/meta.rs:203-281), numeric / datetime columns resident in HBM for the row predicate, and the
//! score + merge block of `MetaQueryPlan::collect` (src/meta.rs:671-709: rayon fan-out over `process_chunk`, concat,
//! sort, truncate) as one `ott_query` that reads only the chunks the zonemap kept.
//!
//! Unchanged and still on the host: `Expr::compile`, `build_chunk_mask_for_plan` (src/meta.rs:407-544, microseconds),
//! Bloom filters, string leaves, `MetaQueryStats` printing, materialisation (src/meta.rs:722-828).
use std::collections::HashMap;

use bitvec::prelude::BitVec;
use otters_hip_sys as sys;

use crate::expr::{CmpOp, ColumnFilter, CompiledFilter, NumericLiteral};
use crate::type_utils::DataType;
use crate::vec::{Cmp as VecCmp, Metric, TakeType};
use crate::vec_hip::{cmp_code, mask_words, metric_code, take_code, VecStore};

/// What `MetaStoreBuilder::build` keeps beside the zonemaps when the feature is on.
pub(crate) struct DeviceMeta {
    pub store: VecStore,                       // all rows, chunk c = rows [c * chunk_size, ..)
    pub column_ids: HashMap<String, u32>,      // numeric / datetime columns uploaded with ott_store_add_column
    pub n_chunks: usize,
}

/// src/meta.rs:203-281 on the device: one reserve, one append of the flattened vectors, the chunk size, the columns.
/// `columns`: (name, dtype, raw little-endian values, null-mask words (1 = NULL) or empty).
pub(crate) fn build_device_meta(
    dim: usize,
    chunk_size: usize,
    vectors: &[Vec<f32>],
    columns: &[(String, DataType, Vec<u8>, Vec<u64>)],
) -> Result<DeviceMeta, String> {
    let mut store = VecStore::new(dim); // the GPUs of OTTERS_HIP_DEVICES: several = one store sharded by contiguous chunk ranges
    let h = store.handle();
    // chunk size first, then the plan: on a multi-GPU store the reserve lays the shards out on chunk boundaries
    sys::check(unsafe { sys::ott_store_set_chunk_size(h, chunk_size as u64) })?;
    sys::check(unsafe { sys::ott_store_reserve(h, vectors.len() as u64) })?;
    // MetaQueryPlan::collect keeps, at exact score ties, what one TopKCollector PER CHUNK plus the stable concat-sort-truncate
    // keeps (src/meta_compute.rs:153-192, src/meta.rs:699-709): tie_order 2 — for ANY chunk size (src/meta.rs:86-89 accepts
    // every chunk_size >= 1): since round 5 the library counts a chunk's 8-row blocks from the chunk's own first row.
    store.set_tie_order(2)?;
    store.add_vectors(vectors.to_vec())?;
    let mut column_ids = HashMap::new();
    for (name, dtype, values, nulls) in columns {
        let code = match dtype {
            DataType::Int32 => sys::OTT_DT_INT32,
            DataType::Int64 => sys::OTT_DT_INT64,
            DataType::Float32 => sys::OTT_DT_FLOAT32,
            DataType::Float64 => sys::OTT_DT_FLOAT64,
            DataType::DateTime => sys::OTT_DT_DATETIME,
            _ => continue, // String columns stay on the host (src/meta_compute.rs:291-318)
        };
        let mut id = 0u32;
        let nulls_ptr = if nulls.is_empty() { std::ptr::null() } else { nulls.as_ptr() };
        sys::check(unsafe {
            sys::ott_store_add_column(h, code, values.as_ptr() as *const _, nulls_ptr, vectors.len() as u64, &mut id)
        })?;
        column_ids.insert(name.clone(), id);
    }
    let n_chunks = (vectors.len() + chunk_size - 1) / chunk_size;
    Ok(DeviceMeta { store, column_ids, n_chunks })
}

fn op_code(op: &CmpOp) -> u32 {
    match op {
        CmpOp::Eq => sys::OTT_OP_EQ,
        CmpOp::Neq => sys::OTT_OP_NEQ,
        CmpOp::Lt => sys::OTT_OP_LT,
        CmpOp::Lte => sys::OTT_OP_LTE,
        CmpOp::Gt => sys::OTT_OP_GT,
        CmpOp::Gte => sys::OTT_OP_GTE,
    }
}

/// CNF -> `ott_leaf`s when every leaf is numeric and its column is resident (literal coerced to the column's type exactly
/// as src/meta_compute.rs:249-283 does: integers for Int32 / Int64 / DateTime, f64 otherwise); `None` = some leaf has
/// to be evaluated on the host (strings), then the caller builds the row mask there and passes it in `desc.row_mask`.
fn device_leaves(compiled: &CompiledFilter, dev: &DeviceMeta) -> Option<(Vec<sys::ott_leaf>, u32)> {
    let mut leaves = Vec::new();
    for (ci, clause) in compiled.clauses.iter().enumerate() {
        for leaf in clause {
            match leaf {
                ColumnFilter::Numeric { column, cmp, rhs } => {
                    let id = *dev.column_ids.get(column)?;
                    let (lit_i64, lit_f64) = match rhs {
                        NumericLiteral::I64(v) => (*v, *v as f64),
                        NumericLiteral::F64(v) => (*v as i64, *v),
                    };
                    leaves.push(sys::ott_leaf { column: id, op: op_code(cmp), clause: ci as u32, reserved: 0, lit_i64, lit_f64 });
                }
                _ => return None,
            }
        }
    }
    Some((leaves, compiled.clauses.len() as u32))
}

/// Replaces src/meta.rs:671-709.  `chunk_mask` = `build_chunk_mask_for_plan(compiled)` (None without a meta filter);
/// `host_row_mask` builds the row predicate over ALL rows on the host when some leaf cannot go to the GPU.
/// Returns the aggregated `(global row, score)` list, already sorted and truncated, plus the library's stats.
pub(crate) fn score_and_merge_on_device(
    dev: &DeviceMeta,
    queries: &[Vec<f32>],
    metric: &Metric,
    take_type: &TakeType,
    k: usize,
    vec_filter: Option<&(f32, VecCmp)>,
    compiled: Option<&CompiledFilter>,
    chunk_mask: Option<&BitVec>,
    host_row_mask: impl FnOnce(&CompiledFilter) -> BitVec,
) -> Result<(Vec<(usize, f32)>, sys::ott_stats), String> {
    let h = dev.store.handle();
    let n = dev.store.len();
    let mut stats = sys::ott_stats::default();
    if n == 0 || k == 0 || queries.is_empty() {
        return Ok((Vec::new(), stats));
    }
    // row predicate: on the GPU over the resident columns, else a host-built BitVec over all rows
    let mut host_mask: Option<BitVec> = None;
    let mut use_device_row_mask = 0u32;
    if let Some(c) = compiled {
        match device_leaves(c, dev) {
            Some((leaves, n_clauses)) => {
                sys::check(unsafe {
                    sys::ott_store_eval_row_mask(h, leaves.as_ptr(), leaves.len() as u32, n_clauses, std::ptr::null_mut())
                })?;
                use_device_row_mask = 1;
            }
            None => host_mask = Some(host_row_mask(c)),
        }
    }
    let flat: Vec<f32> = queries.iter().flatten().copied().collect();
    let (filter_cmp, filter_thr) = match vec_filter {
        Some((thr, cmp)) => (cmp_code(cmp), *thr),
        None => (sys::OTT_CMP_NONE, 0.0),
    };
    let desc = sys::ott_query_desc {
        queries: flat.as_ptr(),
        nq: queries.len() as u32,
        metric: metric_code(metric),
        take: take_code(take_type),
        filter_cmp,
        filter_thr,
        mode: sys::OTT_MODE_MERGED, // aggregated over all queries, like `aggregated` (src/meta.rs:661)
        k: k as u64,
        chunk_mask: chunk_mask.map_or(std::ptr::null(), mask_words),
        row_mask: host_mask.as_ref().map_or(std::ptr::null(), mask_words),
        row_mask_bits: host_mask.as_ref().map_or(0, |m| m.len() as u64),
        use_device_row_mask,
        path: sys::OTT_PATH_AUTO,
    };
    let cap = k.min(n * queries.len()).max(1);
    let mut hits = vec![sys::ott_hit { index: 0, score: 0.0, query: 0 }; cap];
    let mut n_out = 0u64;
    sys::check(unsafe {
        sys::ott_query(h, &desc, hits.as_mut_ptr(), cap as u64, &mut n_out, std::ptr::null_mut(), &mut stats)
    })?;
    hits.truncate(n_out as usize);
    // stats.total_chunks / pruned_chunks / evaluated_chunks / vectors_compared fill MetaQueryStats (src/meta.rs:711-720)
    Ok((hits.into_iter().map(|h| (h.index as usize, h.score)).collect(), stats))
}
