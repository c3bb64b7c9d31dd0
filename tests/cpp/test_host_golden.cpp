// C++ host mirror against the reference's own host-planner suites: tests/golden/expr_cases.json and column_cases.json
// (tests/expr_tests.rs, tests/column_tests.rs as data; tests/golden/make_host_golden.py) interpreted against
// include/otters_meta.hpp.  Host-only: no query runs, no GPU is needed.  usage: test_host_golden expr_cases.json column_cases.json
#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>

#include "otters_meta.hpp"

using namespace otters;

// ---- a JSON reader just large enough for the fixtures ------------------------------------------------------------------------
struct J {
    enum T { Nul, Bool, Int, Dbl, Str, Arr, Obj } t = Nul;
    bool b = false;
    int64_t i = 0;
    double d = 0;
    std::string s;
    std::vector<J> a;
    std::vector<std::pair<std::string, J>> o;
    const J* get(const std::string& k) const {
        for (const auto& kv : o)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
    const J& at(const std::string& k) const {
        const J* p = get(k);
        if (!p) throw std::runtime_error("missing key " + k);
        return *p;
    }
    bool has(const std::string& k) const { return get(k) != nullptr; }
};
struct Parser {
    const std::string& s;
    size_t p = 0;
    explicit Parser(const std::string& str) : s(str) {}
    void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) p++; }
    J parse() {
        ws();
        J j;
        const char c = s.at(p);
        if (c == '{') {
            j.t = J::Obj;
            p++;
            ws();
            if (s[p] == '}') { p++; return j; }
            for (;;) {
                ws();
                J k = parse();
                ws();
                if (s.at(p++) != ':') throw std::runtime_error("json: ':' expected");
                j.o.emplace_back(k.s, parse());
                ws();
                if (s[p] == ',') { p++; continue; }
                if (s.at(p++) != '}') throw std::runtime_error("json: '}' expected");
                return j;
            }
        }
        if (c == '[') {
            j.t = J::Arr;
            p++;
            ws();
            if (s[p] == ']') { p++; return j; }
            for (;;) {
                j.a.push_back(parse());
                ws();
                if (s[p] == ',') { p++; continue; }
                if (s.at(p++) != ']') throw std::runtime_error("json: ']' expected");
                return j;
            }
        }
        if (c == '"') {
            j.t = J::Str;
            p++;
            while (s.at(p) != '"') {
                if (s[p] == '\\') {
                    p++;
                    const char e = s.at(p++);
                    if (e == 'n') j.s += '\n';
                    else if (e == 't') j.s += '\t';
                    else if (e == 'u') {  // the fixtures hold ASCII only
                        j.s += static_cast<char>(std::stoi(s.substr(p, 4), nullptr, 16));
                        p += 4;
                    } else j.s += e;
                } else j.s += s[p++];
            }
            p++;
            return j;
        }
        if (s.compare(p, 4, "null") == 0) { p += 4; return j; }
        if (s.compare(p, 4, "true") == 0) { p += 4; j.t = J::Bool; j.b = true; return j; }
        if (s.compare(p, 5, "false") == 0) { p += 5; j.t = J::Bool; return j; }
        size_t q = p;
        bool is_dbl = false;
        while (q < s.size() && (std::isdigit(static_cast<unsigned char>(s[q])) || s[q] == '-' || s[q] == '+' || s[q] == '.' || s[q] == 'e' || s[q] == 'E')) {
            if (s[q] == '.' || s[q] == 'e' || s[q] == 'E') is_dbl = true;
            q++;
        }
        const std::string num = s.substr(p, q - p);
        p = q;
        if (is_dbl) { j.t = J::Dbl; j.d = std::stod(num); }
        else { j.t = J::Int; j.i = std::stoll(num); j.d = static_cast<double>(j.i); }
        return j;
    }
};
static J load(const char* path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    return Parser(text).parse();
}

static int failures = 0;
static std::string where;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::printf("FAIL [%s] %s:%d: %s\n", where.c_str(), __FILE__, __LINE__, #cond); \
            failures++;                                                              \
        }                                                                            \
    } while (0)

static DataType dtype_of(const std::string& n) {
    for (int d = 0; d < 6; d++)
        if (n == dtype_name(static_cast<DataType>(d))) return static_cast<DataType>(d);
    throw std::runtime_error("dtype " + n);
}
static CmpOp op_of(const std::string& n) {
    static const char* names[] = {"eq", "neq", "lt", "lte", "gt", "gte"};
    for (int i = 0; i < 6; i++)
        if (n == names[i]) return static_cast<CmpOp>(i);
    throw std::runtime_error("op " + n);
}
static const char* op_name(CmpOp o) {
    static const char* names[] = {"eq", "neq", "lt", "lte", "gt", "gte"};
    return names[static_cast<int>(o)];
}
static Value literal(const J& j) {  // {"i": 25} | {"f": 80.5} | {"s": "x"} | null
    if (j.t == J::Nul) return null;
    const auto& kv = j.o.at(0);
    if (kv.first == "i") return Value(static_cast<long long>(kv.second.i));
    if (kv.first == "f") return Value(kv.second.d);
    return Value(kv.second.s);
}
static Expr build(const J& e) {
    if (e.has("cmp")) {
        const auto& c = e.at("cmp").a;
        const Expr column = col(c[0].s);
        const Value v = literal(c[2]);
        switch (op_of(c[1].s)) {
            case CmpOp::Eq: return column.eq(v);
            case CmpOp::Neq: return column.neq(v);
            case CmpOp::Lt: return column.lt(v);
            case CmpOp::Lte: return column.lte(v);
            case CmpOp::Gt: return column.gt(v);
            default: return column.gte(v);
        }
    }
    if (e.has("and")) return build(e.at("and").a[0]) & build(e.at("and").a[1]);
    if (e.has("or")) return build(e.at("or").a[0]) | build(e.at("or").a[1]);
    if (e.has("col")) return col(e.at("col").s);
    if (e.has("lit")) return lit(literal(e.at("lit")));
    const J& r = e.at("raw_cmp");
    return cmp_expr(build(r.at("left")), build(r.at("right")), op_of(r.at("op").s));
}
static bool leaf_is(const ColumnFilter& f, const J& w) {
    if (f.column != w.at("column").s || std::string(op_name(f.cmp)) != w.at("cmp").s) return false;
    const J& rhs = w.at("rhs");
    if (w.at("kind").s == "String") return !f.numeric && rhs.t == J::Str && f.str == rhs.s;
    if (!f.numeric) return false;
    const auto& kv = rhs.o.at(0);
    if (kv.first == "I64") return !f.num.is_f64 && f.num.i == kv.second.i;
    return f.num.is_f64 && f.num.f == kv.second.d;
}

static void run_expr(const J& root) {
    Schema schema;
    for (const auto& kv : root.at("schema").o) schema[kv.first] = dtype_of(kv.second.s);
    for (const J& c : root.at("cases").a) {
        where = c.at("name").s + " (" + c.at("ref").s + ")";
        const J& want = c.at("expect");
        std::string msg;
        CompiledFilter cf;
        bool threw = false;
        try { cf = build(c.at("expr")).compile(schema); } catch (const Error& e) { threw = true; msg = e.what(); }
        if (want.has("error")) {
            CHECK(threw && msg == want.at("display").s);
            continue;
        }
        CHECK(!threw);
        if (threw) continue;
        if (want.has("clauses")) {
            const auto& wc = want.at("clauses").a;
            CHECK(cf.clauses.size() == wc.size());
            for (size_t i = 0; i < wc.size() && i < cf.clauses.size(); i++) {
                CHECK(cf.clauses[i].size() == wc[i].a.size());
                for (size_t j = 0; j < wc[i].a.size() && j < cf.clauses[i].size(); j++) CHECK(leaf_is(cf.clauses[i][j], wc[i].a[j]));
            }
        }
        if (want.has("n_clauses")) CHECK(cf.clauses.size() == static_cast<size_t>(want.at("n_clauses").i));
        if (want.has("clause_sizes"))
            for (size_t i = 0; i < want.at("clause_sizes").a.size(); i++) CHECK(cf.clauses.at(i).size() == static_cast<size_t>(want.at("clause_sizes").a[i].i));
        if (want.has("clause_sizes_sorted")) {
            std::vector<size_t> got;
            for (const auto& cl : cf.clauses) got.push_back(cl.size());
            std::sort(got.begin(), got.end());
            CHECK(got.size() == want.at("clause_sizes_sorted").a.size());
            for (size_t i = 0; i < got.size() && i < want.at("clause_sizes_sorted").a.size(); i++) CHECK(got[i] == static_cast<size_t>(want.at("clause_sizes_sorted").a[i].i));
        }
        if (want.has("first_leaf_kinds"))
            for (size_t i = 0; i < want.at("first_leaf_kinds").a.size(); i++)
                CHECK(cf.clauses.at(i).at(0).numeric == (want.at("first_leaf_kinds").a[i].s == "Numeric"));
    }
}

static void check_column(const Column& c, const J& want) {
    for (const auto& kv : want.o) {
        const std::string& key = kv.first;
        const J& v = kv.second;
        if (key == "name") CHECK(c.name() == v.s);
        else if (key == "dtype") CHECK(c.dtype() == dtype_of(v.s));
        else if (key == "len") CHECK(c.len() == static_cast<size_t>(v.i));
        else if (key == "is_empty") CHECK(c.is_empty() == v.b);
        else if (key == "null_mask") {
            CHECK(c.null_mask().size() == v.a.size());
            for (size_t i = 0; i < v.a.size() && i < c.null_mask().size(); i++) CHECK(c.null_mask()[i] == v.a[i].b);
        } else if (key == "accessors") {
            // the reference's typed accessors answer None for another type (src/col.rs:446-485); the C++ mirror's answer an empty vector
            for (const auto& a : v.o) {
                size_t n = a.first == "i32" ? c.i32_values().size() : a.first == "f32" ? c.f32_values().size() : c.string_values().size();
                CHECK(n == (a.second.t == J::Nul ? 0u : static_cast<size_t>(a.second.i)));
            }
        } else if (key == "values_len") CHECK(c.len() == static_cast<size_t>(v.i));
        else if (key == "values_is_empty") CHECK(c.is_empty() == v.b);
        else if (key == "values_dtype") CHECK(c.data_type() == dtype_of(v.s));
        else if (key == "raw") {
            CHECK(c.len() == v.a.size());
            for (size_t i = 0; i < v.a.size() && i < c.len(); i++) {
                const J& w = v.a[i];
                switch (c.dtype()) {
                    case DataType::Int32: CHECK(c.i32_values()[i] == w.i); break;
                    case DataType::Int64: case DataType::DateTime: CHECK(c.i64_values()[i] == w.i); break;
                    case DataType::Float32: CHECK(w.t == J::Str ? std::isnan(c.f32_values()[i]) : c.f32_values()[i] == static_cast<float>(w.d)); break;
                    case DataType::Float64: CHECK(w.t == J::Str ? std::isnan(c.f64_values()[i]) : c.f64_values()[i] == w.d); break;
                    default: CHECK(c.string_values()[i] == w.s); break;
                }
            }
        } else if (key == "head_n") CHECK(c.head_n(static_cast<size_t>(v.a[0].i)) == v.a[1].s);
        else {
            std::printf("unknown expectation %s\n", key.c_str());
            failures++;
        }
    }
}

static void run_columns(const J& root) {
    for (const J& cs : root.a) {
        where = cs.at("name").s + " (" + cs.at("ref").s + ")";
        std::unique_ptr<Column> c;
        for (const J& st : cs.at("steps").a) {
            if (st.has("new")) {
                const J& n = st.at("new");
                c = std::make_unique<Column>(n.at("name").s, dtype_of(n.at("dtype").s));
                if (n.at("fmt").t == J::Str) c->with_datetime_fmt(n.at("fmt").s);
            } else if (st.has("expect")) check_column(*c, st.at("expect"));
            else {
                bool threw = false;
                std::string msg;
                try {
                    if (st.has("push")) c->push(literal(st.at("push")));
                    else if (st.has("from")) {
                        std::vector<Value> vals;
                        for (const J& v : st.at("from").a) vals.push_back(literal(v));
                        c->from(vals);
                    } else {
                        std::vector<Value> vals;
                        for (int64_t x = st.at("from_range").a[0].i; x < st.at("from_range").a[1].i; x++) vals.push_back(Value(static_cast<long long>(x)));
                        c->from(vals);
                    }
                } catch (const Error& e) { threw = true; msg = e.what(); }
                CHECK(threw == !st.at("ok").b);
                if (threw && st.has("error") && st.at("error").s == "ParseError") CHECK(msg.rfind("Parse error: ", 0) == 0);  // src/col.rs:86-97
            }
        }
    }
}

int main(int argc, char** argv) {
    if (argc != 3) {
        std::printf("usage: %s expr_cases.json column_cases.json\n", argv[0]);
        return 2;
    }
    try {
        const J e = load(argv[1]), c = load(argv[2]);
        run_expr(e);
        run_columns(c);
        std::printf("%zu expr cases, %zu column cases\n", e.at("cases").a.size(), c.a.size());
    } catch (const std::exception& ex) {
        std::printf("FAIL [%s]: exception %s\n", where.c_str(), ex.what());
        return 1;
    }
    std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
    return failures ? 1 : 0;
}
