// otters.hpp — header-only C++17 host mirror of otters' `vec` module (src/vec.rs) over the
// libotters_hip.so C ABI (include/otters_hip.h).  Same method names, argument meaning and error
// strings as the reference; where the reference returns Err(String), this throws otters::Error.
//
//   otters::VecStore store(3);
//   store.add_vectors({{1,0,0},{0,1,0}});
//   auto hits = store.query({1,0,0}, otters::Metric::Cosine).filter(0.5f, otters::Cmp::Gt).take(5).collect();
#pragma once

#include <cstdint>
#include <cstdlib>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "otters_hip.h"

namespace otters {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

enum class Metric { Cosine = 0, Euclidean = 1, DotProduct = 2 };  // src/vec.rs:11-16
enum class TakeType { Min = 0, Max = 1 };                          // src/vec.rs:18-22
enum class Cmp { Lt = 1, Gt = 2, Lte = 3, Gte = 4, Eq = 5 };       // src/vec.rs:24-31

struct SearchResult {  // src/vec.rs:34-38
    std::size_t index;
    float score;
    bool operator==(const SearchResult& o) const { return index == o.index && score == o.score; }
};

inline void check(int rc) {
    if (rc != 0) throw Error(ott_last_error());
}

class VecStore;

class VecQueryPlan {  // src/vec.rs:55-312
  public:
    VecQueryPlan() = default;  // VecQueryPlan::new

    VecQueryPlan& with_vector_store(const VecStore* s) {
        if (!error_) store_ = s;
        return *this;
    }
    VecQueryPlan& with_query_vectors(std::vector<std::vector<float>> q) {
        if (!error_) queries_ = std::move(q);
        return *this;
    }
    VecQueryPlan& with_metric(Metric m) {
        if (!error_) metric_ = m;
        return *this;
    }
    // bit i = row i, true = keep; rows past the end of the mask are kept (src/vec.rs:234)
    VecQueryPlan& with_row_mask(std::vector<bool> mask) {
        if (!error_) row_mask_ = std::move(mask);
        return *this;
    }
    VecQueryPlan& filter(float score, Cmp cmp) {  // src/vec.rs:150-153
        if (!error_) filter_ = std::make_pair(score, cmp);
        return *this;
    }
    VecQueryPlan& take(std::size_t count) { return take_with_options(count, std::nullopt); }
    VecQueryPlan& take_min(std::size_t count) { return take_with_options(count, TakeType::Min); }
    VecQueryPlan& take_max(std::size_t count) { return take_with_options(count, TakeType::Max); }

    std::vector<SearchResult> collect() const;  // src/vec.rs:205-311

  private:
    VecQueryPlan& take_with_options(std::size_t count, std::optional<TakeType> tt) {  // src/vec.rs:103-116
        if (error_) return *this;
        take_count_ = count;
        if (tt) take_type_ = tt;
        else if (!take_type_ && metric_) take_type_ = (*metric_ == Metric::Euclidean) ? TakeType::Min : TakeType::Max;
        return *this;
    }
    void validate() const;

    std::optional<std::vector<std::vector<float>>> queries_;
    std::optional<Metric> metric_;
    std::optional<std::pair<float, Cmp>> filter_;
    std::optional<TakeType> take_type_;
    std::optional<std::size_t> take_count_;
    const VecStore* store_ = nullptr;
    std::optional<std::string> error_;
    std::optional<std::vector<bool>> row_mask_;
    friend class VecStore;
};

class VecStore {  // src/vec.rs:338-412
  public:
    // The device list: the constructor's, else the environment variable OTTERS_HIP_DEVICES ("0,1,2,3": one store over those GPUs
    // of this process, ott_store_create_multi), else the one GPU `device`.
    explicit VecStore(std::size_t dim, int device = 0) : dim_(dim), device_(device), devices_(devices_from_env()) {}
    VecStore(std::size_t dim, std::vector<int> devices) : dim_(dim), device_(devices.empty() ? 0 : devices[0]), devices_(std::move(devices)) {}
    static std::vector<int> devices_from_env() {
        std::vector<int> out;
        const char* e = std::getenv("OTTERS_HIP_DEVICES");
        if (!e) return out;
        const std::string s(e);
        std::size_t i = 0;
        while (i < s.size()) {
            std::size_t j = s.find(',', i);
            if (j == std::string::npos) j = s.size();
            if (j > i) out.push_back(std::atoi(s.substr(i, j - i).c_str()));
            i = j + 1;
        }
        return out;
    }
    VecStore(const VecStore&) = delete;
    VecStore& operator=(const VecStore&) = delete;
    ~VecStore() {
        if (h_) ott_store_destroy(h_);
    }

    void add_vector(const std::vector<float>& v) {  // src/vec.rs:357-371
        if (v.size() != dim_)
            throw Error("Input vector length " + std::to_string(v.size()) + " does not match expected dimension " + std::to_string(dim_));
        check(ott_store_append(handle(), v.data(), 1));
        n_ += 1;
    }
    void add_vectors(const std::vector<std::vector<float>>& vs) {  // src/vec.rs:373-376 (try_for_each)
        std::vector<float> flat;
        std::size_t good = 0;
        std::optional<std::string> err;
        for (const auto& v : vs) {
            if (v.size() != dim_) {
                err = "Input vector length " + std::to_string(v.size()) + " does not match expected dimension " + std::to_string(dim_);
                break;
            }
            flat.insert(flat.end(), v.begin(), v.end());
            good++;
        }
        if (good) {
            check(ott_store_append(handle(), flat.data(), good));
            n_ += good;
        }
        if (err) throw Error(*err);
    }
    std::size_t len() const { return n_; }
    bool is_empty() const { return n_ == 0; }
    std::size_t dim() const { return dim_; }
    // ott_store_set_option (include/otters_hip.h): "tie_order", "hi_fmt", diagnostics
    void set_option(const std::string& name, long long value) { check(ott_store_set_option(handle(), name.c_str(), value)); }
    // which of several EQUAL-scoring pairs survives the cut at take(k): false = the library's canonical total order (score, row,
    // query); true = what the reference's TopKCollector keeps (src/vec_compute.rs:236-277, one collector over the store)
    void use_reference_tie_order(bool on = true) {
        tie_order_ = on ? 1 : 0;
        set_option("tie_order", tie_order_);
    }
    // 0 canonical, 1 the reference's single collector (default), 2 one collector per chunk (MetaStore)
    void set_tie_order(int order) {
        tie_order_ = order;
        set_option("tie_order", order);
    }

    VecQueryPlan query(std::vector<float> q, Metric m) const { return query(std::vector<std::vector<float>>{std::move(q)}, m); }
    VecQueryPlan query(std::vector<std::vector<float>> qs, Metric m) const {  // src/vec.rs:386-411
        VecQueryPlan p;
        p.queries_ = std::move(qs);
        p.metric_ = m;
        p.store_ = this;
        return p;
    }
    ott_store* handle() const {
        if (!h_) {
            if (!devices_.empty()) check(ott_store_create_multi(static_cast<uint32_t>(dim_), static_cast<uint32_t>(devices_.size()), devices_.data(), &h_));
            else check(ott_store_create(static_cast<uint32_t>(dim_), device_, &h_));
            // the reference's outcome at exact score ties is the default of this mirror, as of the Rust patch (one collector over
            // the store, src/vec_compute.rs:236-277; MetaStore switches its store to the per-chunk form)
            check(ott_store_set_option(h_, "tie_order", tie_order_));
        }
        return h_;
    }
    // Pre-size the store (a multi-GPU store also plans the even split of n rows over its shards with it)
    void reserve(std::size_t n_rows) { check(ott_store_reserve(handle(), n_rows)); }
    int shard_count() const { return ott_store_shard_count(handle()); }

  private:
    std::size_t dim_;
    int device_;
    std::vector<int> devices_;  // several GPUs of this process: ONE store over all of them (ott_store_create_multi)
    int tie_order_ = 1;
    std::size_t n_ = 0;
    mutable ott_store* h_ = nullptr;
};

inline void VecQueryPlan::validate() const {  // src/vec.rs:170-203
    if (error_) throw Error(*error_);
    if (!queries_) throw Error("Query vectors or their norms are not set");
    if (!metric_) throw Error("Search metric is not set");
    if (!store_) throw Error("Vector store is not set");
    if (queries_->empty()) throw Error("No queries provided");
    for (const auto& q : *queries_)
        if (q.size() != store_->dim())
            throw Error("Query vector length " + std::to_string(q.size()) + " does not match expected dimension " +
                        std::to_string(store_->dim()));
}

inline std::vector<SearchResult> VecQueryPlan::collect() const {
    validate();
    const std::size_t n = store_->len(), nq = queries_->size();
    const std::size_t k = take_count_.value_or(n);                // src/vec.rs:213
    const TakeType tt = take_type_.value_or(TakeType::Max);       // src/vec.rs:214
    if (n == 0 || k == 0) return {};
    std::vector<float> flat;
    flat.reserve(nq * store_->dim());
    for (const auto& q : *queries_) flat.insert(flat.end(), q.begin(), q.end());
    std::vector<uint64_t> words;
    ott_query_desc d{};
    d.queries = flat.data();
    d.nq = static_cast<uint32_t>(nq);
    d.metric = static_cast<uint32_t>(*metric_);
    d.take = static_cast<uint32_t>(tt);
    d.filter_cmp = filter_ ? static_cast<uint32_t>(filter_->second) : OTT_CMP_NONE;
    d.filter_thr = filter_ ? filter_->first : 0.0f;
    d.mode = OTT_MODE_MERGED;
    d.k = k;
    if (row_mask_ && !row_mask_->empty()) {
        words.assign((row_mask_->size() + 63) / 64, 0);
        for (std::size_t i = 0; i < row_mask_->size(); i++)
            if ((*row_mask_)[i]) words[i >> 6] |= uint64_t(1) << (i & 63);
        d.row_mask = words.data();
        d.row_mask_bits = row_mask_->size();
    }
    const std::size_t cap = k < n * nq ? k : n * nq;
    std::vector<ott_hit> hits(cap ? cap : 1);
    uint64_t n_out = 0;
    check(ott_query(store_->handle(), &d, hits.data(), cap, &n_out, nullptr, nullptr));
    std::vector<SearchResult> out;
    out.reserve(n_out);
    for (uint64_t i = 0; i < n_out; i++) out.push_back({static_cast<std::size_t>(hits[i].index), hits[i].score});
    return out;
}

}  // namespace otters
