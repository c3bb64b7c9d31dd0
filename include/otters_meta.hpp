// otters_meta.hpp — header-only C++17 host mirror of otters' `col`, `expr` and `meta` modules
// (src/col.rs, src/expr.rs, src/meta.rs) over the libotters_hip.so C ABI.  Same names, argument
// meaning and error strings as the reference; Err(String) becomes otters::Error.
//
//   using namespace otters;
//   auto age   = Column("age", DataType::Int32).from({10, 20, 30, null});
//   auto grade = Column("grade", DataType::String).from({"A", "B", "A", "C"});
//   auto meta  = MetaStore::from_columns({age, grade}).with_vectors(vectors).with_chunk_size(2).build();
//   auto res   = meta.query({1,0,0}, Metric::Cosine)
//                    .meta_filter(col("age").gt(15) & col("grade").eq("A")).vec_filter(0.5f, Cmp::Gt).take(4).collect();
//
// Host side (as in a patched otters): builders, Expr::compile, zonemap prune, string predicates,
// materialisation.  GPU side: vectors, zone statistics, numeric/datetime row predicates, scoring,
// score filter, top-k, merge.
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <unordered_set>
#include <variant>

#include "otters.hpp"

namespace otters {

enum class DataType { Int32 = 0, Int64 = 1, Float32 = 2, Float64 = 3, String = 4, DateTime = 5 };  // src/type_utils.rs:11-19
inline const char* dtype_name(DataType d) {
    static const char* n[] = {"Int32", "Int64", "Float32", "Float64", "String", "DateTime"};
    return n[static_cast<int>(d)];
}

struct Null {};
inline constexpr Null null{};
// ColumnValue / Literal (src/col.rs:36-45, src/expr.rs:44-80): NULL, integer, float or string.  Explicit
// constructors so that `10`, `19.99`, "A" and `null` all convert without ambiguity.
struct Value {
    std::variant<Null, int64_t, double, std::string> v;
    Value() : v(Null{}) {}
    Value(Null) : v(Null{}) {}
    Value(int x) : v(static_cast<int64_t>(x)) {}
    Value(long x) : v(static_cast<int64_t>(x)) {}
    Value(long long x) : v(static_cast<int64_t>(x)) {}
    Value(float x) : v(static_cast<double>(x)) {}
    Value(double x) : v(x) {}
    Value(const char* x) : v(std::string(x)) {}
    Value(std::string x) : v(std::move(x)) {}
    bool is_null() const { return std::holds_alternative<Null>(v); }
    bool is_int() const { return std::holds_alternative<int64_t>(v); }
    bool is_float() const { return std::holds_alternative<double>(v); }
    bool is_str() const { return std::holds_alternative<std::string>(v); }
    int64_t as_int() const { return std::get<int64_t>(v); }
    double as_float() const { return std::get<double>(v); }
    const std::string& as_str() const { return std::get<std::string>(v); }
    bool operator==(const Value& o) const { return v.index() == o.v.index() && (is_null() || (is_int() ? as_int() == o.as_int() : is_float() ? as_float() == o.as_float() : as_str() == o.as_str())); }
};

// ---- datetime parsing: src/col.rs:506-527, src/expr.rs:267-283 ------------------------------------
inline bool days_from_civil_ok(int y, int m, int d) { return m >= 1 && m <= 12 && d >= 1 && d <= 31 && y >= 1 && y <= 9999; }
inline int64_t days_from_civil(int y, int m, int d) {  // proleptic Gregorian, days since 1970-01-01
    y -= m <= 2;
    const int64_t era = (y >= 0 ? y : y - 399) / 400;
    const unsigned yoe = static_cast<unsigned>(y - era * 400);
    const unsigned doy = (153u * static_cast<unsigned>(m + (m > 2 ? -3 : 9)) + 2u) / 5u + static_cast<unsigned>(d) - 1u;
    const unsigned doe = yoe * 365u + yoe / 4u - yoe / 100u + doy;
    return era * 146097 + static_cast<int64_t>(doe) - 719468;
}
inline bool valid_date(int y, int m, int d) {
    if (!days_from_civil_ok(y, m, d)) return false;
    static const int dm[] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    const bool leap = (y % 4 == 0 && y % 100 != 0) || y % 400 == 0;
    return d <= dm[m - 1] + (m == 2 && leap ? 1 : 0);
}
inline std::optional<int64_t> parse_datetime_millis(const std::string& s) {
    int y, mo, d, h, mi, sec, n = 0;
    // RFC 3339: YYYY-MM-DD[Tt ]HH:MM:SS[.frac](Z|+hh:mm)
    if (s.size() >= 20 && std::sscanf(s.c_str(), "%4d-%2d-%2d%*1[Tt ]%2d:%2d:%2d%n", &y, &mo, &d, &h, &mi, &sec, &n) == 6 && n == 19) {
        size_t p = 19;
        int64_t frac_ms = 0;
        if (p < s.size() && s[p] == '.') {
            p++;
            int digits = 0;
            int64_t nanos = 0;
            while (p < s.size() && s[p] >= '0' && s[p] <= '9') {
                if (digits < 9) nanos = nanos * 10 + (s[p] - '0'), digits++;
                p++;
            }
            if (digits == 0) return std::nullopt;
            while (digits++ < 9) nanos *= 10;
            frac_ms = nanos / 1000000;
        }
        int64_t off = 0;
        bool tz_ok = false;
        if (p + 1 == s.size() && (s[p] == 'Z' || s[p] == 'z')) tz_ok = true;
        else if (p + 6 == s.size() && (s[p] == '+' || s[p] == '-') && s[p + 3] == ':') {
            const int oh = (s[p + 1] - '0') * 10 + (s[p + 2] - '0'), om = (s[p + 4] - '0') * 10 + (s[p + 5] - '0');
            off = (oh * 3600 + om * 60) * (s[p] == '+' ? 1 : -1);
            tz_ok = true;
        }
        if (tz_ok && valid_date(y, mo, d) && h < 24 && mi < 60 && sec <= 60)
            return (days_from_civil(y, mo, d) * 86400 + h * 3600 + mi * 60 + sec - off) * 1000 + frac_ms;
        return std::nullopt;
    }
    n = 0;
    if (std::sscanf(s.c_str(), "%4d-%2d-%2d%n", &y, &mo, &d, &n) == 3 && static_cast<size_t>(n) == s.size() && valid_date(y, mo, d))
        return days_from_civil(y, mo, d) * 86400 * 1000;
    n = 0;
    if (std::sscanf(s.c_str(), "%4d-%2d-%2d %2d:%2d:%2d%n", &y, &mo, &d, &h, &mi, &sec, &n) == 6 && static_cast<size_t>(n) == s.size() &&
        valid_date(y, mo, d) && h < 24 && mi < 60 && sec <= 60)
        return (days_from_civil(y, mo, d) * 86400 + h * 3600 + mi * 60 + sec) * 1000;
    return std::nullopt;
}

// parse_datetime_fmt (src/col.rs:529-545): the whole string must match `fmt`; formats without a time of day mean midnight UTC
// (chrono's NaiveDateTime first, NaiveDate second).  strptime fills only what the format names.
inline std::optional<int64_t> parse_datetime_fmt_millis(const std::string& s, const std::string& fmt) {
    std::tm tm{};
    tm.tm_mday = 1;
    const char* end = strptime(s.c_str(), fmt.c_str(), &tm);
    if (!end || *end != '\0') return std::nullopt;
    const int y = tm.tm_year + 1900, mo = tm.tm_mon + 1, d = tm.tm_mday;
    if (!valid_date(y, mo, d) || tm.tm_hour > 23 || tm.tm_min > 59 || tm.tm_sec > 60) return std::nullopt;
    return (days_from_civil(y, mo, d) * 86400 + tm.tm_hour * 3600 + tm.tm_min * 60 + tm.tm_sec) * 1000;
}

// ---- Column: src/col.rs:21-28, 195-503 ---------------------------------------------------------------
class Column {
  public:
    Column(std::string name, DataType dtype) : name_(std::move(name)), dtype_(dtype) {}
    const std::string& name() const { return name_; }
    DataType dtype() const { return dtype_; }
    std::size_t len() const { return nulls_.size(); }
    bool is_empty() const { return nulls_.empty(); }
    DataType data_type() const { return dtype_; }  // ColumnValues::data_type, src/col.rs:74-83
    // strftime-style format for datetime strings pushed from now on (src/col.rs:352-355): "%m/%d/%Y", "%Y-%m-%d %H:%M", ..
    Column with_datetime_fmt(const std::string& fmt) && {
        fmt_ = fmt;
        return std::move(*this);
    }
    Column& with_datetime_fmt(const std::string& fmt) & {
        fmt_ = fmt;
        return *this;
    }
    // Column::head_n (src/col.rs:409-443): the text the reference prints, returned (print it with std::puts)
    std::string head_n(std::size_t n) const {
        std::string out = "Column: " + name_ + " (" + dtype_name(dtype_) + ")";
        char buf[96];
        const std::size_t limit = std::min(len(), n);
        for (std::size_t i = 0; i < limit; i++) {
            out += "\n  [" + std::to_string(i) + "]: ";
            if (nulls_[i]) { out += "NULL"; continue; }
            switch (dtype_) {
                case DataType::Int32: out += std::to_string(i32_[i]); break;
                case DataType::Int64: out += std::to_string(i64_[i]); break;
                case DataType::Float32: std::snprintf(buf, sizeof buf, "%.4f", static_cast<double>(f32_[i])); out += buf; break;
                case DataType::Float64: std::snprintf(buf, sizeof buf, "%.4f", f64_[i]); out += buf; break;
                case DataType::String: out += "\"" + str_[i] + "\""; break;
                case DataType::DateTime: {
                    const int64_t ms = i64_[i];
                    const std::time_t secs = static_cast<std::time_t>(ms >= 0 ? ms / 1000 : -((-ms + 999) / 1000));
                    std::tm tm{};
                    gmtime_r(&secs, &tm);
                    std::strftime(buf, sizeof buf, "%Y-%m-%d %H:%M:%S UTC", &tm);
                    out += std::string(buf) + " (" + std::to_string(ms) + ")";
                    break;
                }
            }
        }
        if (len() > n) out += "\n  ... (" + std::to_string(len() - n) + " more rows)";
        return out;
    }
    std::string head() const { return head_n(5); }  // src/col.rs:403-406

    // unified push (src/col.rs:357-390); Null{} = NULL with the reference's sentinel in the value slot
    void push(const Value& v) {
        const bool is_null = v.is_null();
        switch (dtype_) {
            case DataType::Int32:
                if (!is_null && !v.is_int()) mismatch();
                i32_.push_back(is_null ? std::numeric_limits<int32_t>::min() : static_cast<int32_t>(v.as_int()));
                break;
            case DataType::Int64:
                if (!is_null && !v.is_int()) mismatch();
                i64_.push_back(is_null ? std::numeric_limits<int64_t>::min() : v.as_int());
                break;
            case DataType::Float32:
                f32_.push_back(is_null ? std::nanf("") : static_cast<float>(as_double(v)));
                break;
            case DataType::Float64:
                f64_.push_back(is_null ? std::nan("") : as_double(v));
                break;
            case DataType::String:
                if (!is_null && !v.is_str()) mismatch();
                str_.push_back(is_null ? std::string() : v.as_str());
                break;
            case DataType::DateTime:
                if (is_null) i64_.push_back(std::numeric_limits<int64_t>::min());
                else if (v.is_int()) i64_.push_back(v.as_int());
                else if (v.is_str()) {
                    std::optional<int64_t> ms;
                    if (!fmt_.empty()) {  // parse_datetime_fmt, src/col.rs:529-545
                        ms = parse_datetime_fmt_millis(v.as_str(), fmt_);
                        if (!ms) throw Error("Parse error: Cannot parse '" + v.as_str() + "' with format '" + fmt_ + "'");
                    } else {
                        ms = parse_datetime_millis(v.as_str());
                        if (!ms) throw Error("Parse error: Cannot parse '" + v.as_str() +
                                             "' as datetime. Supported formats: ISO 8601, YYYY-MM-DD, YYYY-MM-DD HH:MM:SS");
                    }
                    i64_.push_back(*ms);
                } else mismatch();
                break;
        }
        nulls_.push_back(is_null);
    }
    Column from(const std::vector<Value>& values) && {  // Column::from, src/col.rs:392-401
        for (const auto& v : values) push(v);
        return std::move(*this);
    }
    Column from(const std::vector<Value>& values) & {
        for (const auto& v : values) push(v);
        return *this;
    }
    const std::vector<int32_t>& i32_values() const { return i32_; }
    const std::vector<int64_t>& i64_values() const { return i64_; }  // Int64 and DateTime (epoch millis)
    const std::vector<float>& f32_values() const { return f32_; }
    const std::vector<double>& f64_values() const { return f64_; }
    const std::vector<std::string>& string_values() const { return str_; }
    const std::vector<bool>& null_mask() const { return nulls_; }  // true = NULL
    Value get(std::size_t i) const {
        if (nulls_[i]) return Null{};
        switch (dtype_) {
            case DataType::Int32: return static_cast<int64_t>(i32_[i]);
            case DataType::Int64: case DataType::DateTime: return i64_[i];
            case DataType::Float32: return static_cast<double>(f32_[i]);
            case DataType::Float64: return f64_[i];
            default: return str_[i];
        }
    }
    Column take(const std::vector<std::size_t>& idx) const {  // materialisation, src/meta.rs:728-821
        Column out(name_, dtype_);
        for (auto i : idx) out.push(get(i));
        return out;
    }
    const void* raw() const {
        switch (dtype_) {
            case DataType::Int32: return i32_.data();
            case DataType::Int64: case DataType::DateTime: return i64_.data();
            case DataType::Float32: return f32_.data();
            case DataType::Float64: return f64_.data();
            default: return nullptr;
        }
    }

  private:
    [[noreturn]] void mismatch() const { throw Error(std::string("Type mismatch: expected ") + dtype_name(dtype_) + ", got incompatible type"); }
    double as_double(const Value& v) const {
        if (v.is_float()) return v.as_float();
        if (v.is_int()) return static_cast<double>(v.as_int());
        mismatch();
    }
    std::string name_;
    DataType dtype_;
    std::string fmt_;  // datetime_format, src/col.rs:27
    std::vector<int32_t> i32_;
    std::vector<int64_t> i64_;
    std::vector<float> f32_;
    std::vector<double> f64_;
    std::vector<std::string> str_;
    std::vector<bool> nulls_;
};

// ---- Expr DSL: src/expr.rs ------------------------------------------------------------------------------
enum class CmpOp { Eq = 0, Neq = 1, Lt = 2, Lte = 3, Gt = 4, Gte = 5 };  // src/expr.rs:83-91

struct NumericLiteral {  // src/expr.rs:192-196
    bool is_f64;
    int64_t i;
    double f;
    bool operator==(const NumericLiteral& o) const { return is_f64 == o.is_f64 && (is_f64 ? f == o.f : i == o.i); }
};
struct ColumnFilter {  // src/expr.rs:198-211
    bool numeric;
    std::string column;
    CmpOp cmp;
    NumericLiteral num{};
    std::string str;
};
struct CompiledFilter {  // src/expr.rs:222-226: AND of clauses, each an OR of filters
    std::vector<std::vector<ColumnFilter>> clauses;
};
using Schema = std::map<std::string, DataType>;

class Expr {
  public:
    enum class Kind { Column, Literal, Cmp, And, Or };
    Kind kind;
    std::string name;                   // Column
    Value literal;                      // Literal
    CmpOp op = CmpOp::Eq;               // Cmp
    std::shared_ptr<Expr> a, b;

    Expr eq(const Value& v) const { return cmp(v, CmpOp::Eq); }
    Expr neq(const Value& v) const { return cmp(v, CmpOp::Neq); }
    Expr lt(const Value& v) const { return cmp(v, CmpOp::Lt); }
    Expr lte(const Value& v) const { return cmp(v, CmpOp::Lte); }
    Expr gt(const Value& v) const { return cmp(v, CmpOp::Gt); }
    Expr gte(const Value& v) const { return cmp(v, CmpOp::Gte); }
    Expr and_(const Expr& o) const { return combine(Kind::And, o); }
    Expr or_(const Expr& o) const { return combine(Kind::Or, o); }
    friend Expr operator&(const Expr& x, const Expr& y) { return x.and_(y); }
    friend Expr operator|(const Expr& x, const Expr& y) { return x.or_(y); }

    CompiledFilter compile(const Schema& schema) const {  // src/expr.rs:285-298
        CompiledFilter out;
        for (auto& clause : lower(schema)) {  // normalize_plan: drop (c == v) OR (c != v), src/expr.rs:300-343
            bool taut = false;
            for (const auto& lf : clause) {
                if (lf.cmp != CmpOp::Eq) continue;
                for (const auto& x : clause)
                    if (x.numeric == lf.numeric && x.cmp == CmpOp::Neq && x.column == lf.column && (lf.numeric ? x.num == lf.num : x.str == lf.str)) taut = true;
            }
            if (!taut) out.clauses.push_back(std::move(clause));
        }
        return out;
    }

  private:
    Expr cmp(const Value& v, CmpOp o) const {
        Expr e;
        e.kind = Kind::Cmp;
        e.op = o;
        e.a = std::make_shared<Expr>(*this);
        Expr l;
        l.kind = Kind::Literal;
        l.literal = v;
        e.b = std::make_shared<Expr>(l);
        return e;
    }
    Expr combine(Kind k, const Expr& o) const {
        Expr e;
        e.kind = k;
        e.a = std::make_shared<Expr>(*this);
        e.b = std::make_shared<Expr>(o);
        return e;
    }
    std::vector<std::vector<ColumnFilter>> lower(const Schema& schema) const {  // src/expr.rs:355-372
        if (kind == Kind::And) {
            auto l = a->lower(schema), r = b->lower(schema);
            if (l.empty()) return r;
            l.insert(l.end(), r.begin(), r.end());  // and_concat_clauses
            return l;
        }
        if (kind == Kind::Or) {
            auto l = a->lower(schema), r = b->lower(schema);
            if (l.empty()) return r;
            if (r.empty()) return l;
            std::vector<std::vector<ColumnFilter>> out;  // or_distribute_clauses
            for (const auto& ca : l)
                for (const auto& cb : r) {
                    auto m = ca;
                    m.insert(m.end(), cb.begin(), cb.end());
                    out.push_back(std::move(m));
                }
            return out;
        }
        if (kind == Kind::Cmp) return {{leaf(schema)}};
        throw Error("Invalid expression (unexpected literal or column without comparator)");
    }
    ColumnFilter leaf(const Schema& schema) const {  // compile_cmp_leaf, src/expr.rs:385-466
        if (!(a->kind == Kind::Column && b->kind == Kind::Literal)) throw Error("Invalid expression shape for comparison (expect column vs literal)");
        const std::string& cname = a->name;
        auto it = schema.find(cname);
        if (it == schema.end()) throw Error("Unknown column '" + cname + "'");
        const DataType dt = it->second;
        const Value& lv = b->literal;
        auto mism = [&](const char* got) { return Error("Type mismatch for column '" + cname + "': expected " + dtype_name(dt) + ", got literal " + got); };
        ColumnFilter f;
        f.column = cname;
        f.cmp = op;
        if (dt == DataType::String) {
            if (op != CmpOp::Eq && op != CmpOp::Neq) throw Error("Unsupported comparator for string column '" + cname + "'");
            if (!lv.is_str()) throw mism("string");
            f.numeric = false;
            f.str = lv.as_str();
            return f;
        }
        f.numeric = true;
        if (dt == DataType::Int32 || dt == DataType::Int64) {
            if (lv.is_float()) throw mism("float");
            if (!lv.is_int()) throw mism("string");
            f.num = {false, lv.as_int(), 0.0};
        } else if (dt == DataType::DateTime) {
            if (!lv.is_str()) throw mism("datetime string");
            auto ms = parse_datetime_millis(lv.as_str());
            if (!ms) throw mism("datetime string");
            f.num = {false, *ms, 0.0};
        } else {
            if (lv.is_str() || lv.is_null()) throw mism("string");
            f.num = {true, 0, lv.is_float() ? lv.as_float() : static_cast<double>(lv.as_int())};
        }
        return f;
    }
};
inline Expr lit(const Value& v) {  // src/expr.rs:112-115
    Expr e;
    e.kind = Expr::Kind::Literal;
    e.literal = v;
    return e;
}
// Expr::Cmp { left, right, op } built by hand (the reference's enum is public: tests/expr_tests.rs:36-40 puts the literal on the left)
inline Expr cmp_expr(const Expr& left, const Expr& right, CmpOp op) {
    Expr e;
    e.kind = Expr::Kind::Cmp;
    e.op = op;
    e.a = std::make_shared<Expr>(left);
    e.b = std::make_shared<Expr>(right);
    return e;
}
inline Expr col(const std::string& name) {  // src/expr.rs:108-111
    Expr e;
    e.kind = Expr::Kind::Column;
    e.name = name;
    return e;
}

// ---- display: src/display.rs ----------------------------------------------------------------------------------------
// AsciiTable::render (display.rs:32-96): the title, when set, is the first line; no trailing newline
inline std::string ascii_table(const std::vector<std::string>& headers, const std::vector<std::vector<std::string>>& rows, const std::string* title = nullptr) {
    if (headers.empty()) return "";
    std::vector<std::size_t> w(headers.size());
    for (std::size_t i = 0; i < headers.size(); i++) w[i] = headers[i].size();
    for (const auto& r : rows)
        for (std::size_t i = 0; i < r.size() && i < w.size(); i++) w[i] = std::max(w[i], r[i].size());
    std::string sep = "+";
    for (auto x : w) sep += std::string(x + 2, '-') + "+";
    auto line = [&](const std::vector<std::string>& r) {
        std::string o = "|";
        for (std::size_t i = 0; i < w.size(); i++) {
            const std::string cell = i < r.size() ? r[i] : std::string();
            o += " " + cell + std::string(w[i] - cell.size() + 1, ' ') + "|";
        }
        return o;
    };
    std::string out = title ? *title + "\n" : std::string();
    out += sep + "\n" + line(headers) + "\n" + sep + "\n";
    for (const auto& r : rows) out += line(r) + "\n";
    return out + sep;
}
inline std::string format_millis_utc(int64_t ms) {  // chrono "%Y-%m-%d %H:%M:%S UTC" of DateTime::from_timestamp_millis
    const std::time_t secs = static_cast<std::time_t>(ms >= 0 ? ms / 1000 : -((-ms + 999) / 1000));
    std::tm tm{};
    gmtime_r(&secs, &tm);
    char buf[64];
    std::strftime(buf, sizeof buf, "%Y-%m-%d %H:%M:%S UTC", &tm);
    return buf;
}
inline std::string format_cell(const Column& c, std::size_t i) {  // display.rs:104-123
    if (c.null_mask()[i]) return "NULL";
    char buf[64];
    switch (c.dtype()) {
        case DataType::Int32: return std::to_string(c.i32_values()[i]);
        case DataType::Int64: return std::to_string(c.i64_values()[i]);
        case DataType::Float32: std::snprintf(buf, sizeof buf, "%.4f", static_cast<double>(c.f32_values()[i])); return buf;
        case DataType::Float64: std::snprintf(buf, sizeof buf, "%.4f", c.f64_values()[i]); return buf;
        case DataType::String: return c.string_values()[i];
        default: return format_millis_utc(c.i64_values()[i]);
    }
}
inline std::string fixed3(double v) {
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.3f", v);
    return buf;
}

// ---- stats: src/meta.rs:832-852 ------------------------------------------------------------------------------
struct MetaQueryStats {
    std::size_t total_chunks = 0, pruned_chunks = 0, evaluated_chunks = 0, vectors_compared = 0;
    double prune_ms = 0, score_ms = 0, merge_ms = 0, total_ms = 0;
    std::string format() const {  // format_query_stats, display.rs:221-249
        const std::string title = "Last Meta Query Stats";
        return ascii_table({"metric", "value"},
                           {{"total_chunks", std::to_string(total_chunks)}, {"pruned_chunks", std::to_string(pruned_chunks)},
                            {"evaluated_chunks", std::to_string(evaluated_chunks)}, {"vectors_compared", std::to_string(vectors_compared)},
                            {"prune_ms", fixed3(prune_ms)}, {"score_ms", fixed3(score_ms)}, {"merge_ms", fixed3(merge_ms)}, {"total_ms", fixed3(total_ms)}},
                           &title);
    }
};
struct MetaBuildStats {  // src/meta.rs:844-852
    std::size_t n_rows = 0, dim = 0, n_chunks = 0;
    double vectors_ingest_ms = 0, zonemap_build_ms = 0, build_total_ms = 0;
    std::string format() const {  // format_build_stats, display.rs:196-219
        const std::string title = "MetaStore Build Stats";
        return ascii_table({"metric", "value"},
                           {{"rows", std::to_string(n_rows)}, {"dimensions", std::to_string(dim)}, {"chunks", std::to_string(n_chunks)},
                            {"vector_ingest_ms", fixed3(vectors_ingest_ms)}, {"zonemap_build_ms", fixed3(zonemap_build_ms)},
                            {"build_total_ms", fixed3(build_total_ms)}},
                           &title);
    }
};
struct MetaQueryResults {  // src/meta.rs:23-40
    std::vector<std::string> columns;
    std::map<std::string, Column> data;
    std::vector<std::size_t> indices;
    std::vector<float> scores;
    std::size_t len() const { return indices.size(); }
    bool is_empty() const { return indices.empty(); }
    const Column* column(const std::string& n) const {
        auto it = data.find(n);
        return it == data.end() ? nullptr : &it->second;
    }
    std::string to_string() const {  // impl Display for MetaQueryResults, display.rs:164-188
        std::vector<std::string> headers{"index", "score"};
        headers.insert(headers.end(), columns.begin(), columns.end());
        std::vector<std::vector<std::string>> rows;
        char buf[64];
        for (std::size_t i = 0; i < len(); i++) {
            std::snprintf(buf, sizeof buf, "%.6f", static_cast<double>(scores[i]));
            std::vector<std::string> r{std::to_string(indices[i]), buf};
            for (const auto& c : columns) {
                const Column* col = column(c);
                r.push_back(col ? format_cell(*col, i) : std::string());
            }
            rows.push_back(std::move(r));
        }
        return ascii_table(headers, rows);
    }
};

class MetaStore;
class MetaQueryPlan;

class MetaStoreBuilder {  // src/meta.rs:62-306
  public:
    MetaStoreBuilder& with_vectors(std::vector<std::vector<float>> v) {
        vectors_ = std::move(v);
        has_vectors_ = true;
        return *this;
    }
    MetaStoreBuilder& with_chunk_size(std::size_t cs) {  // src/meta.rs:86-89
        chunk_size_ = cs < 1 ? 1 : cs;
        return *this;
    }
    MetaStore build();

  private:
    friend class MetaStore;
    Schema schema_;
    std::map<std::string, Column> columns_;
    std::vector<std::vector<float>> vectors_;
    bool has_vectors_ = false;
    std::size_t chunk_size_ = 1024;
    int device_ = 0;
};

class MetaStore {  // src/meta.rs:48-60, 308-577
  public:
    static MetaStoreBuilder from_columns(const std::vector<Column>& cols, int device = 0) {  // src/meta.rs:332-347
        MetaStoreBuilder b;
        b.device_ = device;
        for (const auto& c : cols) {
            b.schema_[c.name()] = c.dtype();
            b.columns_.emplace(c.name(), c);
        }
        return b;
    }
    std::size_t n_chunks() const { return n_chunks_; }
    std::size_t chunk_size() const { return chunk_size_; }
    const Schema& schema() const { return schema_; }
    const std::map<std::string, Column>& columns() const { return columns_; }
    const std::optional<MetaQueryStats>& last_query_stats() const { return last_stats_; }
    const std::optional<MetaBuildStats>& build_stats() const { return build_stats_; }  // src/meta.rs:399-403
    // MetaStore::head_n (src/meta.rs:371-374 -> metastore_head, display.rs:125-161): printed and returned
    std::string head_n(std::size_t n) const {
        std::vector<std::string> headers{"index"};
        for (const auto& kv : schema_) headers.push_back(kv.first);  // (std::map: sorted by name, as display.rs:128 sorts)
        std::vector<std::vector<std::string>> rows;
        for (std::size_t i = 0; i < std::min(n, n_rows_); i++) {
            std::vector<std::string> r{std::to_string(i)};
            for (const auto& kv : schema_) r.push_back(format_cell(columns_.at(kv.first), i));
            rows.push_back(std::move(r));
        }
        const std::string title = "MetaStore \xE2\x80\xA2 rows=" + std::to_string(n_rows_) + " \xE2\x80\xA2 chunks=" + std::to_string(n_chunks_) +
                                  " \xE2\x80\xA2 chunk_size=" + std::to_string(chunk_size_);
        const std::string out = ascii_table(headers, rows, &title);
        std::puts(out.c_str());
        return out;
    }
    std::string head() const { return head_n(5); }  // src/meta.rs:366-369
    void print_build_stats() const { std::puts(build_stats_ ? build_stats_->format().c_str() : "(no build stats)"); }      // src/meta.rs:546-552
    void print_last_query_stats() const { std::puts(last_stats_ ? last_stats_->format().c_str() : "(no query stats)"); }  // src/meta.rs:554-560
    void print_last_stats() const {  // src/meta.rs:562-566
        print_build_stats();
        print_last_query_stats();
    }
    // at exact score ties keep what the reference's MetaQueryPlan::collect keeps: one TopKCollector per surviving chunk
    // (src/meta_compute.rs:153-192), lists concatenated in chunk order, sorted, truncated (src/meta.rs:699-709) — the DEFAULT
    // of this mirror for every chunk size (build() sets it; 8-row blocks count from the chunk's first row).  false = the library's canonical total
    // order.  INTEGRATION.md 6a
    void use_reference_tie_order(bool on = true) {
        if (store_) store_->set_tie_order(on ? 2 : 0);
    }

    MetaQueryPlan query(std::vector<float> q, Metric m) const;
    MetaQueryPlan query_batch(std::vector<std::vector<float>> qs, Metric m) const;

    // zonemap prune: build_chunk_mask_for_plan, src/meta.rs:407-428
    std::vector<bool> build_chunk_mask_for_plan(const CompiledFilter& cf) const {
        std::vector<bool> acc(n_chunks_, true);
        for (const auto& clause : cf.clauses) {
            std::vector<bool> cm(n_chunks_, false);
            for (const auto& lf : clause)
                for (std::size_t c = 0; c < n_chunks_; c++)
                    if (lf.numeric ? numeric_chunk_sat(lf, c) : string_chunk_sat(lf, c)) cm[c] = true;
            for (std::size_t c = 0; c < n_chunks_; c++) acc[c] = acc[c] && cm[c];
        }
        return acc;
    }

  private:
    friend class MetaStoreBuilder;
    friend class MetaQueryPlan;
    struct Zone {  // PackedRanges, src/meta.rs:71-76 (kept as i64 / f64; narrowed per dtype when tested)
        std::vector<int64_t> imin, imax;
        std::vector<double> fmin, fmax;
        std::vector<uint64_t> non_null;
    };
    static int32_t wrap_i32(int64_t v) { return static_cast<int32_t>(static_cast<uint32_t>(static_cast<uint64_t>(v))); }  // `as i32`
    template <typename T>
    static bool range_sat(T mn, T mx, CmpOp op, T thr) {  // src/type_utils.rs:762-769
        switch (op) {
            case CmpOp::Eq: return mn <= thr && thr <= mx;
            case CmpOp::Lt: return mn < thr;
            case CmpOp::Lte: return mn <= thr;
            case CmpOp::Gt: return mx > thr;
            case CmpOp::Gte: return mx >= thr;
            default: return true;
        }
    }
    bool numeric_chunk_sat(const ColumnFilter& lf, std::size_t c) const {  // src/meta.rs:431-521
        auto zi = zones_.find(lf.column);
        auto di = schema_.find(lf.column);
        if (zi == zones_.end() || di == schema_.end()) return false;
        const Zone& z = zi->second;
        if (z.non_null[c] == 0) return false;
        switch (di->second) {
            case DataType::Float32:
                if (!lf.num.is_f64) return false;
                return range_sat<float>(static_cast<float>(z.fmin[c]), static_cast<float>(z.fmax[c]), lf.cmp, static_cast<float>(lf.num.f));
            case DataType::Float64:
                if (!lf.num.is_f64) return false;
                return range_sat<double>(z.fmin[c], z.fmax[c], lf.cmp, lf.num.f);
            case DataType::Int32:
                if (lf.num.is_f64) return false;
                return range_sat<int32_t>(wrap_i32(z.imin[c]), wrap_i32(z.imax[c]), lf.cmp, wrap_i32(lf.num.i));
            case DataType::Int64: case DataType::DateTime:
                if (lf.num.is_f64) return false;
                return range_sat<int64_t>(z.imin[c], z.imax[c], lf.cmp, lf.num.i);
            default: return false;
        }
    }
    bool string_chunk_sat(const ColumnFilter& lf, std::size_t c) const {  // src/meta.rs:523-544
        auto it = str_sets_.find(lf.column);
        if (it == str_sets_.end()) return true;  // conservatively keep when unknown
        if (str_nonnull_.at(lf.column)[c] == 0) return false;
        if (lf.cmp == CmpOp::Eq) return it->second[c].count(lf.str) != 0;  // exact set: a Bloom filter with no false positives
        return lf.cmp == CmpOp::Neq;
    }

    Schema schema_;
    std::map<std::string, Column> columns_;
    std::size_t chunk_size_ = 1024, n_rows_ = 0, dim_ = 0, n_chunks_ = 0;
    std::shared_ptr<VecStore> store_;
    std::map<std::string, Zone> zones_;
    std::map<std::string, std::vector<std::unordered_set<std::string>>> str_sets_;
    std::map<std::string, std::vector<uint64_t>> str_nonnull_;
    std::map<std::string, uint32_t> dev_cols_;
    // String columns: dictionary codes (value -> code; the coded Int32 column is resident in HBM like a numeric one, so `==` / `!=`
    // leaves run in the GPU's row-mask kernel too: src/meta_compute.rs:291-318 compares the strings row by row on the CPU)
    std::map<std::string, std::map<std::string, int32_t>> str_codes_;
    mutable std::optional<MetaQueryStats> last_stats_;
    std::optional<MetaBuildStats> build_stats_;
};

inline MetaStore MetaStoreBuilder::build() {  // src/meta.rs:151-305
    if (!has_vectors_) throw Error("vectors must be provided to build MetaStore");
    const std::size_t n = vectors_.size();
    for (const auto& [name, dt] : schema_) {
        (void)dt;
        auto it = columns_.find(name);
        if (it == columns_.end()) throw Error("missing column '" + name + "' in builder columns");
        if (it->second.len() != n)
            throw Error("column '" + name + "' length " + std::to_string(it->second.len()) + " does not match vectors length " + std::to_string(n));
    }
    const std::size_t dim = n ? vectors_[0].size() : 0;
    if (dim == 0 && n > 0) throw Error("vector dimension cannot be zero");
    for (std::size_t i = 0; i < n; i++)
        if (vectors_[i].size() != dim)
            throw Error("vector at index " + std::to_string(i) + " has dim " + std::to_string(vectors_[i].size()) + ", expected " + std::to_string(dim));
    MetaStore ms;
    ms.schema_ = schema_;
    ms.columns_ = columns_;
    ms.chunk_size_ = chunk_size_;
    ms.n_rows_ = n;
    ms.dim_ = dim;
    ms.n_chunks_ = (n + chunk_size_ - 1) / chunk_size_;
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
    const auto t_build = clk::now();
    ms.build_stats_ = MetaBuildStats{n, dim, ms.n_chunks_, 0, 0, 0};  // src/meta.rs:292-299
    if (!n) return ms;
    ms.store_ = std::make_shared<VecStore>(dim, device_);
    ms.store_->set_tie_order(2);  // MetaStore's tie outcome: per-chunk collectors (src/meta.rs:678-709), any chunk size (src/meta.rs:86-89)
    check(ott_store_set_chunk_size(ms.store_->handle(), chunk_size_));
    check(ott_store_reserve(ms.store_->handle(), n));
    ms.store_->add_vectors(vectors_);
    ms.build_stats_->vectors_ingest_ms = ms_since(t_build);
    const auto t_zone = clk::now();
    for (const auto& [name, dt] : schema_) {
        const Column& c = ms.columns_.at(name);
        if (dt == DataType::String) {
            auto& sets = ms.str_sets_[name];
            auto& nn = ms.str_nonnull_[name];
            sets.resize(ms.n_chunks_);
            nn.assign(ms.n_chunks_, 0);
            for (std::size_t i = 0; i < n; i++)
                if (!c.null_mask()[i]) {
                    sets[i / chunk_size_].insert(c.string_values()[i]);
                    nn[i / chunk_size_]++;
                }
            // the row predicate's copy: dictionary codes as a resident Int32 column (NULL rows: code 0 under the null bitmap)
            auto& dict = ms.str_codes_[name];
            for (std::size_t i = 0; i < n; i++)
                if (!c.null_mask()[i]) dict.emplace(c.string_values()[i], 0);
            int32_t next = 0;
            for (auto& kv : dict) kv.second = next++;
            std::vector<int32_t> codes(n, 0);
            std::vector<uint64_t> swords((n + 63) / 64, 0);
            bool s_null = false;
            for (std::size_t i = 0; i < n; i++) {
                if (c.null_mask()[i]) swords[i >> 6] |= uint64_t(1) << (i & 63), s_null = true;
                else codes[i] = dict.at(c.string_values()[i]);
            }
            uint32_t scid = 0;
            check(ott_store_add_column(ms.store_->handle(), OTT_DT_INT32, codes.data(), s_null ? swords.data() : nullptr, n, &scid));
            ms.dev_cols_[name] = scid;
            continue;
        }
        // numeric / datetime: values + null bitmap go to HBM once; zone statistics are computed there
        std::vector<uint64_t> words((n + 63) / 64, 0);
        bool any_null = false;
        for (std::size_t i = 0; i < n; i++)
            if (c.null_mask()[i]) words[i >> 6] |= uint64_t(1) << (i & 63), any_null = true;
        uint32_t cid = 0;
        check(ott_store_add_column(ms.store_->handle(), static_cast<uint32_t>(dt), c.raw(), any_null ? words.data() : nullptr, n, &cid));
        ms.dev_cols_[name] = cid;
        MetaStore::Zone z;
        z.non_null.assign(ms.n_chunks_, 0);
        const bool is_f = dt == DataType::Float32 || dt == DataType::Float64;
        if (is_f) {
            z.fmin.assign(ms.n_chunks_, 0);
            z.fmax.assign(ms.n_chunks_, 0);
            check(ott_store_zone_stats(ms.store_->handle(), cid, chunk_size_, z.fmin.data(), z.fmax.data(), z.non_null.data()));
        } else {
            z.imin.assign(ms.n_chunks_, 0);
            z.imax.assign(ms.n_chunks_, 0);
            check(ott_store_zone_stats(ms.store_->handle(), cid, chunk_size_, z.imin.data(), z.imax.data(), z.non_null.data()));
        }
        ms.zones_[name] = std::move(z);
    }
    ms.build_stats_->zonemap_build_ms = ms_since(t_zone);
    ms.build_stats_->build_total_ms = ms_since(t_build);
    return ms;
}

class MetaQueryPlan {  // src/meta.rs:579-830
  public:
    MetaQueryPlan(const MetaStore* s, std::vector<std::vector<float>> q, Metric m) : store_(s), queries_(std::move(q)), metric_(m) {}
    MetaQueryPlan& meta_filter(const Expr& e) {  // src/meta.rs:605-616: compile error deferred to collect
        try {
            filter_ = e.compile(store_->schema_);
            meta_error_.reset();
        } catch (const Error& err) {
            meta_error_ = std::string("meta_filter compile error: ") + err.what();
        }
        return *this;
    }
    MetaQueryPlan& vec_filter(float score, Cmp cmp) {
        vec_filter_ = std::make_pair(score, cmp);
        return *this;
    }
    MetaQueryPlan& take(std::size_t k) {  // src/meta.rs:623-630
        take_count_ = k;
        take_type_ = metric_ == Metric::Euclidean ? TakeType::Min : TakeType::Max;
        return *this;
    }
    MetaQueryResults collect() const {  // src/meta.rs:632-829
        if (meta_error_) throw Error(*meta_error_);
        const MetaStore& st = *store_;
        if (queries_.empty()) throw Error("No queries provided");
        const std::size_t k = take_count_.value_or(st.n_rows_);
        const TakeType tt = take_type_.value_or(metric_ == Metric::Euclidean ? TakeType::Min : TakeType::Max);
        MetaQueryStats stats;
        stats.total_chunks = st.n_chunks_;
        std::vector<ott_hit> hits;
        uint64_t n_out = 0;
        std::vector<bool> cmask;
        if (filter_) cmask = st.build_chunk_mask_for_plan(*filter_);
        const bool any_chunk = !filter_ || std::find(cmask.begin(), cmask.end(), true) != cmask.end();
        stats.evaluated_chunks = filter_ ? static_cast<std::size_t>(std::count(cmask.begin(), cmask.end(), true)) : st.n_chunks_;
        stats.pruned_chunks = st.n_chunks_ - stats.evaluated_chunks;
        if (st.store_ && st.n_rows_ && k && any_chunk) {
            for (const auto& q : queries_)
                if (q.size() != st.dim_)
                    throw Error("Query vector length " + std::to_string(q.size()) + " does not match expected dimension " + std::to_string(st.dim_));
            std::vector<float> flat;
            for (const auto& q : queries_) flat.insert(flat.end(), q.begin(), q.end());
            ott_query_desc d{};
            d.queries = flat.data();
            d.nq = static_cast<uint32_t>(queries_.size());
            d.metric = static_cast<uint32_t>(metric_);
            d.take = static_cast<uint32_t>(tt);
            d.filter_cmp = vec_filter_ ? static_cast<uint32_t>(vec_filter_->second) : OTT_CMP_NONE;
            d.filter_thr = vec_filter_ ? vec_filter_->first : 0.f;
            d.mode = OTT_MODE_MERGED;
            d.k = k;
            std::vector<uint64_t> cwords;
            if (filter_) {
                cwords.assign((st.n_chunks_ + 63) / 64, 0);
                for (std::size_t c = 0; c < st.n_chunks_; c++)
                    if (cmask[c]) cwords[c >> 6] |= uint64_t(1) << (c & 63);
                d.chunk_mask = cwords.data();
                // row predicates on the GPU (build_row_mask_for_chunk, src/meta_compute.rs:194-318): numeric, datetime and — over their
                // dictionary codes — string leaves alike
                std::vector<ott_leaf> leaves;
                for (std::size_t ci = 0; ci < filter_->clauses.size(); ci++)
                    for (const auto& lf : filter_->clauses[ci]) leaves.push_back(st_leaf(st, lf, static_cast<uint32_t>(ci)));
                check(ott_store_eval_row_mask(st.store_->handle(), leaves.data(), static_cast<uint32_t>(leaves.size()),
                                              static_cast<uint32_t>(filter_->clauses.size()), nullptr));
                d.use_device_row_mask = 1;
            }
            const std::size_t pool = st.n_rows_ * queries_.size();
            const std::size_t cap = k < pool ? k : pool;
            hits.resize(cap ? cap : 1);
            ott_stats gs{};
            check(ott_query(st.store_->handle(), &d, hits.data(), cap, &n_out, nullptr, &gs));
            stats.vectors_compared = gs.vectors_compared;
            stats.prune_ms = gs.prune_ns / 1e6;
            stats.score_ms = gs.score_ns / 1e6;
            stats.merge_ms = gs.merge_ns / 1e6;
            stats.total_ms = gs.total_ns / 1e6;
        }
        st.last_stats_ = stats;
        MetaQueryResults res;
        for (uint64_t i = 0; i < n_out; i++) {
            res.indices.push_back(static_cast<std::size_t>(hits[i].index));
            res.scores.push_back(hits[i].score);
        }
        for (const auto& [name, dt] : st.schema_) {  // sorted names (std::map), src/meta.rs:723-724
            (void)dt;
            res.columns.push_back(name);
            res.data.emplace(name, st.columns_.at(name).take(res.indices));
        }
        return res;
    }

  private:
    static int64_t sat_i64(double v) {  // Rust `f64 as i64`
        if (std::isnan(v)) return 0;
        if (v <= -9223372036854775808.0) return std::numeric_limits<int64_t>::min();
        if (v >= 9223372036854775807.0) return std::numeric_limits<int64_t>::max();
        return static_cast<int64_t>(v);
    }
    static ott_leaf st_leaf(const MetaStore& st, const ColumnFilter& lf, uint32_t clause) {  // coercions: src/meta_compute.rs:249-283
        ott_leaf l{};
        l.column = st.dev_cols_.at(lf.column);
        l.op = static_cast<uint32_t>(lf.cmp);
        l.clause = clause;
        if (!lf.numeric) {
            // src/meta_compute.rs:291-318: Eq / Neq on the non-null rows, every other operator matches nothing.  A literal absent
            // from the dictionary gets code -1 (no row has it); "matches nothing" is Eq -1
            const auto& dict = st.str_codes_.at(lf.column);
            const auto it = dict.find(lf.str);
            int64_t code = it == dict.end() ? -1 : it->second;
            if (lf.cmp != CmpOp::Eq && lf.cmp != CmpOp::Neq) {
                l.op = static_cast<uint32_t>(CmpOp::Eq);
                code = -1;
            }
            l.lit_i64 = code;
            return l;
        }
        const DataType dt = st.schema_.at(lf.column);
        if (dt == DataType::Float32) l.lit_f64 = static_cast<double>(static_cast<float>(lf.num.is_f64 ? lf.num.f : static_cast<double>(lf.num.i)));
        else if (dt == DataType::Float64) l.lit_f64 = lf.num.is_f64 ? lf.num.f : static_cast<double>(lf.num.i);
        else if (dt == DataType::Int32) {
            const int64_t v = lf.num.is_f64 ? sat_i64(lf.num.f) : lf.num.i;
            l.lit_i64 = lf.num.is_f64 ? std::clamp<int64_t>(v, std::numeric_limits<int32_t>::min(), std::numeric_limits<int32_t>::max())
                                      : static_cast<int64_t>(static_cast<int32_t>(static_cast<uint32_t>(static_cast<uint64_t>(v))));
        } else l.lit_i64 = lf.num.is_f64 ? sat_i64(lf.num.f) : lf.num.i;
        return l;
    }
    const MetaStore* store_;
    std::vector<std::vector<float>> queries_;
    Metric metric_;
    std::optional<CompiledFilter> filter_;
    std::optional<std::string> meta_error_;
    std::optional<std::pair<float, Cmp>> vec_filter_;
    std::optional<TakeType> take_type_;
    std::optional<std::size_t> take_count_;
};

inline MetaQueryPlan MetaStore::query(std::vector<float> q, Metric m) const { return MetaQueryPlan(this, {std::move(q)}, m); }
inline MetaQueryPlan MetaStore::query_batch(std::vector<std::vector<float>> qs, Metric m) const { return MetaQueryPlan(this, std::move(qs), m); }

}  // namespace otters
