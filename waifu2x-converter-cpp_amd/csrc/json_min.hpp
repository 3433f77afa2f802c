// json_min.hpp -- a minimal JSON DOM parser for the model files.
//
// The reference parses with picojson (vendored at /root/reference/include/picojson.h, NOT copied):
// numbers go through strtod to double (picojson.h:725-793).  This parser does the same --
// every number is strtod'ed to a double -- so the double -> float narrowing of weights
// (modelHandler.cpp:95-97) sees bit-identical inputs.
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace jsonmin {

struct Value {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    double num = 0.0;
    bool b = false;
    std::string str;
    std::vector<Value> arr;
    std::map<std::string, Value> obj;

    bool is_array() const { return type == Array; }
    bool is_object() const { return type == Object; }
    bool is_number() const { return type == Number; }
    const Value *find(const char *key) const
    {
        auto it = obj.find(key);
        return it == obj.end() ? nullptr : &it->second;
    }
};

class Parser {
public:
    Parser(const char *begin, const char *end) : p_(begin), end_(end) {}
    bool parse(Value &out, std::string &err)
    {
        if (!value(out)) { err = err_.empty() ? "syntax error" : err_; return false; }
        ws();
        if (p_ != end_) { err = "trailing characters after JSON value"; return false; }
        return true;
    }

private:
    const char *p_, *end_;
    std::string err_;
    int depth_ = 0;

    void ws() { while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) ++p_; }
    bool fail(const char *m) { if (err_.empty()) err_ = m; return false; }
    bool lit(const char *s)
    {
        size_t n = strlen(s);
        if ((size_t)(end_ - p_) < n || memcmp(p_, s, n) != 0) return fail("bad literal");
        p_ += n;
        return true;
    }
    bool string(std::string &s)
    {
        if (p_ >= end_ || *p_ != '"') return fail("expected string");
        ++p_;
        while (p_ < end_ && *p_ != '"') {
            if (*p_ == '\\') {
                if (++p_ >= end_) return fail("bad escape");
                switch (*p_) {
                case '"': s += '"'; break;
                case '\\': s += '\\'; break;
                case '/': s += '/'; break;
                case 'b': s += '\b'; break;
                case 'f': s += '\f'; break;
                case 'n': s += '\n'; break;
                case 'r': s += '\r'; break;
                case 't': s += '\t'; break;
                case 'u':
                    if (end_ - p_ < 5) return fail("bad \\u escape");
                    s += '?';   // keys in model files are ASCII; code points are not needed
                    p_ += 4;
                    break;
                default: return fail("bad escape");
                }
                ++p_;
            } else {
                s += *p_++;
            }
        }
        if (p_ >= end_) return fail("unterminated string");
        ++p_;
        return true;
    }
    bool value(Value &v)
    {
        if (++depth_ > 64) return fail("nesting too deep");
        ws();
        if (p_ >= end_) return fail("unexpected end of input");
        bool ok = true;
        const char c = *p_;
        if (c == '[') {
            v.type = Value::Array;
            ++p_;
            ws();
            if (p_ < end_ && *p_ == ']') { ++p_; }
            else {
                for (;;) {
                    v.arr.emplace_back();
                    if (!value(v.arr.back())) { ok = false; break; }
                    ws();
                    if (p_ < end_ && *p_ == ',') { ++p_; continue; }
                    if (p_ < end_ && *p_ == ']') { ++p_; break; }
                    ok = fail("expected ',' or ']'");
                    break;
                }
            }
        } else if (c == '{') {
            v.type = Value::Object;
            ++p_;
            ws();
            if (p_ < end_ && *p_ == '}') { ++p_; }
            else {
                for (;;) {
                    ws();
                    std::string key;
                    if (!string(key)) { ok = false; break; }
                    ws();
                    if (p_ >= end_ || *p_ != ':') { ok = fail("expected ':'"); break; }
                    ++p_;
                    if (!value(v.obj[key])) { ok = false; break; }
                    ws();
                    if (p_ < end_ && *p_ == ',') { ++p_; continue; }
                    if (p_ < end_ && *p_ == '}') { ++p_; break; }
                    ok = fail("expected ',' or '}'");
                    break;
                }
            }
        } else if (c == '"') {
            v.type = Value::String;
            ok = string(v.str);
        } else if (c == 't') { v.type = Value::Bool; v.b = true; ok = lit("true"); }
        else if (c == 'f') { v.type = Value::Bool; v.b = false; ok = lit("false"); }
        else if (c == 'n') { v.type = Value::Null; ok = lit("null"); }
        else if (c == '-' || (c >= '0' && c <= '9')) {
            // the buffer handed to the parser is NUL-terminated (see load), so strtod is safe
            char *e = nullptr;
            v.type = Value::Number;
            v.num = strtod(p_, &e);
            if (e == p_) ok = fail("bad number");
            p_ = e;
        } else {
            ok = fail("unexpected character");
        }
        --depth_;
        return ok;
    }
};

}  // namespace jsonmin
