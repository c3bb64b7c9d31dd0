//! `cfg(feature = "hip")` bodies for src/vec.rs: the vectors live in HBM behind an `ott_store`, and
//! `VecQueryPlan::collect` hands the validated plan to `ott_query` instead of running the 8-row SIMD loop and the
//! `TopKCollector` on the CPU (src/vec.rs:217-310, src/vec_compute.rs).
//!
//! How it is wired in (src/vec.rs keeps its enums, `SearchResult`, `QueryBatch`, the whole `VecQueryPlan` builder and
//! `validate()`):
//!
//! ```ignore
//! #[cfg(feature = "hip")]      pub use crate::vec_hip::VecStore;
//! #[cfg(not(feature = "hip"))] pub struct VecStore { vectors: Vec<f32>, dim: usize, inv_norms: Vec<f32>, n_vecs: usize }
//! ...
//! pub fn collect(self) -> Result<Vec<SearchResult>, String> {
//!     self.validate()?;
//!     #[cfg(feature = "hip")]
//!     return crate::vec_hip::collect_on_device(self.vector_store.unwrap(), self.query_vectors.as_ref().unwrap(),
//!         self.search_metric.as_ref().unwrap(), self.take_count, self.take_type.as_ref(), self.filter_criteria.as_ref(),
//!         self.row_mask.as_ref());
//!     ...
//! ```
use bitvec::prelude::BitVec;
use otters_hip_sys as sys;

use crate::vec::{Cmp, Metric, SearchResult, TakeType};

/// src/vec.rs:338-344 with the two `Vec<f32>` replaced by a device store (rows + inverse norms in HBM).
#[derive(Debug)]
pub struct VecStore {
    handle: *mut sys::ott_store,
    pub(crate) dim: usize,
    pub(crate) n_vecs: usize,
}

// `ott_query` is re-entrant on one store from several threads (include/otters_hip.h, "Threading"); everything that
// changes the store takes `&mut self` here, which is the exclusive access the library asks for.
unsafe impl Send for VecStore {}
unsafe impl Sync for VecStore {}

/// The GPUs a store lives on: the environment variable `OTTERS_HIP_DEVICES` ("0,1,2,3,4,5,6,7" — HIP device ordinals of THIS
/// process), else GPU 0.  More than one entry = ONE store over all of them (`ott_store_create_multi`): otters stays the
/// single-process library it is (the rayon fan-out of src/meta.rs:678-691 becomes a fan-out over GPUs inside `ott_query`), and
/// `store.query(q).take(10).collect()` reaches every GPU without a launcher, ranks or a rendezvous.
pub(crate) fn devices_from_env() -> Vec<i32> {
    match std::env::var("OTTERS_HIP_DEVICES") {
        Ok(v) => v.split(',').filter_map(|t| t.trim().parse::<i32>().ok()).collect(),
        Err(_) => Vec::new(),
    }
}

impl VecStore {
    /// src/vec.rs:348-355
    pub fn new(dim: usize) -> Self {
        Self::new_on(dim, &devices_from_env())
    }

    /// `devices`: empty = GPU 0; one entry = that GPU; several = one store sharded over them by contiguous chunk ranges.
    pub fn new_on(dim: usize, devices: &[i32]) -> Self {
        let mut handle = std::ptr::null_mut();
        let rc = match devices.len() {
            0 => unsafe { sys::ott_store_create(dim as u32, 0, &mut handle) },
            1 => unsafe { sys::ott_store_create(dim as u32, devices[0], &mut handle) },
            n => unsafe { sys::ott_store_create_multi(dim as u32, n as u32, devices.as_ptr(), &mut handle) },
        };
        assert!(rc == sys::OTT_OK, "ott_store_create: {}", sys::last_error());
        // Score bits for dim >= 8 depend on the order of wide::f32x8::reduce_add, which depends on how THIS crate is
        // compiled: with target_feature = "avx" (e.g. RUSTFLAGS="-C target-cpu=native") it is the AVX shuffle order, on a
        // default x86_64 build (SSE2 only) f32x8 is two f32x4 whose lanes are summed one after the other.  The backend has
        // both; pick the one the CPU path of the same build would have used, so results are bit-identical to it.
        let order = if cfg!(target_feature = "avx") { sys::OTT_REDUCE_AVX } else { sys::OTT_REDUCE_SEQ4 };
        let rc = unsafe { sys::ott_store_set_reduce_order(handle, order) };
        assert!(rc == sys::OTT_OK, "ott_store_set_reduce_order: {}", sys::last_error());
        // At exact score ties the CPU path keeps what its TopKCollector keeps (strict-improvement inserts in visit order,
        // src/vec_compute.rs:236-277).  The drop-in returns exactly that by default — the same set as the CPU path of the same
        // build — at the cost of ONE extra candidate per query when nothing is tied (measured: profiles/round4/tie_order_cost.md).
        let mut store = Self { handle, dim, n_vecs: 0 };
        store.set_tie_order(1).expect("tie_order");
        store
    }

    /// 0 = the library's canonical total order (score, row, query); 1 = the reference's single collector (the default
    /// here); 2 = one collector per chunk, then concat-sort-truncate (what `MetaStore` sets, src/meta.rs:678-709).
    pub(crate) fn set_tie_order(&mut self, order: i64) -> Result<(), String> {
        let name = std::ffi::CString::new("tie_order").unwrap();
        sys::check(unsafe { sys::ott_store_set_option(self.handle, name.as_ptr(), order) })
    }

    /// src/vec.rs:357-371: same length check and message; the inverse norm (sequential sum of squares, sqrt,
    /// reciprocal, 0 for a zero vector) is computed on the GPU in the same order of operations.
    pub fn add_vector(&mut self, vector: Vec<f32>) -> Result<(), String> {
        if vector.len() != self.dim {
            return Err(format!(
                "Input vector length {} does not match expected dimension {}",
                vector.len(),
                self.dim
            ));
        }
        sys::check(unsafe { sys::ott_store_append(self.handle, vector.as_ptr(), 1) })?;
        self.n_vecs += 1;
        Ok(())
    }

    /// src/vec.rs:373-376 (`try_for_each`: rows in front of a bad one stay added) as ONE upload per good prefix.
    pub fn add_vectors(&mut self, vectors: Vec<Vec<f32>>) -> Result<(), String> {
        let good = vectors.iter().take_while(|v| v.len() == self.dim).count();
        if good > 0 {
            let mut flat: Vec<f32> = Vec::with_capacity(good * self.dim);
            vectors[..good].iter().for_each(|v| flat.extend_from_slice(v));
            sys::check(unsafe { sys::ott_store_append(self.handle, flat.as_ptr(), good as u64) })?;
            self.n_vecs += good;
        }
        match vectors.get(good) {
            Some(bad) => Err(format!(
                "Input vector length {} does not match expected dimension {}",
                bad.len(),
                self.dim
            )),
            None => Ok(()),
        }
    }

    pub fn len(&self) -> usize {
        self.n_vecs
    }

    pub fn is_empty(&self) -> bool {
        self.n_vecs == 0
    }

    pub(crate) fn handle(&self) -> *mut sys::ott_store {
        self.handle
    }

    /// Opt out of the reference's tie outcome: the library's canonical total order (better score, lower row, lower query) —
    /// deterministic across chunk sizes and shard counts, one candidate cheaper.
    pub fn use_canonical_tie_order(&mut self) -> Result<(), String> {
        self.set_tie_order(0)
    }
}

impl Drop for VecStore {
    fn drop(&mut self) {
        unsafe { sys::ott_store_destroy(self.handle) };
    }
}

pub(crate) fn metric_code(m: &Metric) -> u32 {
    match m {
        Metric::Cosine => sys::OTT_METRIC_COSINE,
        Metric::Euclidean => sys::OTT_METRIC_EUCLIDEAN,
        Metric::DotProduct => sys::OTT_METRIC_DOT,
    }
}

pub(crate) fn take_code(t: &TakeType) -> u32 {
    match t {
        TakeType::Min => sys::OTT_TAKE_MIN,
        TakeType::Max => sys::OTT_TAKE_MAX,
    }
}

pub(crate) fn cmp_code(c: &Cmp) -> u32 {
    match c {
        Cmp::Lt => sys::OTT_CMP_LT,
        Cmp::Gt => sys::OTT_CMP_GT,
        Cmp::Lte => sys::OTT_CMP_LTE,
        Cmp::Gte => sys::OTT_CMP_GTE,
        Cmp::Eq => sys::OTT_CMP_EQ,
    }
}

/// `BitVec<usize, Lsb0>` words as the `u64` words of `ott_query_desc.row_mask` / `.chunk_mask` (64-bit targets only:
/// usize == u64 there, and the library is an amd64 / ROCm artefact anyway).
#[cfg(target_pointer_width = "64")]
pub(crate) fn mask_words(m: &BitVec) -> *const u64 {
    m.as_raw_slice().as_ptr() as *const u64
}

/// The body of `VecQueryPlan::collect` behind `validate()` (src/vec.rs:209-310): one `ott_query`.
/// MERGED mode = one top-k over all (query, row) pairs, which is what the single global collector computes
/// (src/vec.rs:217-219); the hits come back best first, like `into_sorted_vec()`.
pub(crate) fn collect_on_device(
    store: &VecStore,
    queries: &[Vec<f32>],
    metric: &Metric,
    take_count: Option<usize>,
    take_type: Option<&TakeType>,
    filter: Option<&(f32, Cmp)>,
    row_mask: Option<&BitVec>,
) -> Result<Vec<SearchResult>, String> {
    let take_count = take_count.unwrap_or(store.n_vecs); // src/vec.rs:213
    let take_type = take_type.unwrap_or(&TakeType::Max); // src/vec.rs:214
    if store.n_vecs == 0 || take_count == 0 {
        return Ok(Vec::new()); // src/vec.rs:222, 270; src/vec_compute.rs:174
    }
    let flat: Vec<f32> = queries.iter().flatten().copied().collect();
    let (filter_cmp, filter_thr) = match filter {
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
        mode: sys::OTT_MODE_MERGED,
        k: take_count as u64,
        chunk_mask: std::ptr::null(),
        row_mask: row_mask.map_or(std::ptr::null(), mask_words),
        row_mask_bits: row_mask.map_or(0, |m| m.len() as u64), // rows past the mask are kept (src/vec.rs:234)
        use_device_row_mask: 0,
        path: sys::OTT_PATH_AUTO,
    };
    let cap = take_count.min(store.n_vecs * queries.len()).max(1);
    let mut hits = vec![sys::ott_hit { index: 0, score: 0.0, query: 0 }; cap];
    let mut n_out = 0u64;
    sys::check(unsafe {
        sys::ott_query(
            store.handle(),
            &desc,
            hits.as_mut_ptr(),
            cap as u64,
            &mut n_out,
            std::ptr::null_mut(),
            std::ptr::null_mut(),
        )
    })?;
    hits.truncate(n_out as usize);
    Ok(hits
        .into_iter()
        .map(|h| SearchResult { index: h.index as usize, score: h.score })
        .collect())
}
