// json_min.hpp -- just enough JSON to read the glTF fields the reference consumes
// (SURVEY.md Appendix B).  Replaces tinygltf's JSON layer (an un-vendored submodule of the
// reference, .gitmodules:15-17) for Scene::loadGLTFmodel (Core/Scene/Scene.cu:22-57).
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace drt {

struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JsonValue> arr;
    std::map<std::string, JsonValue> obj;

    bool has(const char *key) const { return kind == Object && obj.count(key) != 0; }
    const JsonValue &at(const char *key) const {
        static const JsonValue null_value;
        if (kind != Object) return null_value;
        auto it = obj.find(key);
        return it == obj.end() ? null_value : it->second;
    }
    const JsonValue &at(size_t i) const {
        static const JsonValue null_value;
        return (kind == Array && i < arr.size()) ? arr[i] : null_value;
    }
    size_t size() const { return kind == Array ? arr.size() : (kind == Object ? obj.size() : 0); }
    bool is_number() const { return kind == Number; }
    long long as_int(long long dflt) const { return kind == Number ? (long long)num : dflt; }
    double as_double(double dflt) const { return kind == Number ? num : dflt; }
    std::string as_string() const { return kind == String ? str : std::string(); }
};

class JsonParser {
public:
    JsonParser(const char *p, size_t n) : p_(p), end_(p + n) {}
    JsonValue parse() {
        JsonValue v = value(0);
        ws();
        if (p_ != end_) fail("trailing characters");
        return v;
    }

private:
    const char *p_, *end_;
    [[noreturn]] void fail(const char *what) const { throw std::runtime_error(std::string("JSON: ") + what); }
    void ws() { while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) ++p_; }
    bool lit(const char *s) {
        const char *q = p_;
        while (*s) { if (q >= end_ || *q != *s) return false; ++q; ++s; }
        p_ = q;
        return true;
    }
    static void utf8(std::string &out, unsigned cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    unsigned hex4() {
        if (end_ - p_ < 4) fail("short \\u escape");
        unsigned v = 0;
        for (int i = 0; i < 4; i++) {
            char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string string() {
        std::string out;
        ++p_;  // opening quote
        for (;;) {
            if (p_ >= end_) fail("unterminated string");
            char c = *p_++;
            if (c == '"') break;
            if (c != '\\') { out += c; continue; }
            if (p_ >= end_) fail("unterminated escape");
            char e = *p_++;
            switch (e) {
            case '"': out += '"'; break;   case '\\': out += '\\'; break; case '/': out += '/'; break;
            case 'b': out += '\b'; break;  case 'f': out += '\f'; break;  case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;  case 't': out += '\t'; break;
            case 'u': {
                unsigned cp = hex4();
                if (cp >= 0xD800 && cp < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                    p_ += 2;
                    unsigned lo = hex4();
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                }
                utf8(out, cp);
                break;
            }
            default: fail("bad escape");
            }
        }
        return out;
    }
    JsonValue value(int depth) {
        if (depth > 128) fail("nesting too deep");
        ws();
        if (p_ >= end_) fail("unexpected end");
        JsonValue v;
        char c = *p_;
        if (c == '{') {
            v.kind = JsonValue::Object;
            ++p_; ws();
            if (p_ < end_ && *p_ == '}') { ++p_; return v; }
            for (;;) {
                ws();
                if (p_ >= end_ || *p_ != '"') fail("object key expected");
                std::string key = string();
                ws();
                if (p_ >= end_ || *p_ != ':') fail("':' expected");
                ++p_;
                v.obj[key] = value(depth + 1);
                ws();
                if (p_ < end_ && *p_ == ',') { ++p_; continue; }
                if (p_ < end_ && *p_ == '}') { ++p_; break; }
                fail("',' or '}' expected");
            }
        } else if (c == '[') {
            v.kind = JsonValue::Array;
            ++p_; ws();
            if (p_ < end_ && *p_ == ']') { ++p_; return v; }
            for (;;) {
                v.arr.push_back(value(depth + 1));
                ws();
                if (p_ < end_ && *p_ == ',') { ++p_; continue; }
                if (p_ < end_ && *p_ == ']') { ++p_; break; }
                fail("',' or ']' expected");
            }
        } else if (c == '"') {
            v.kind = JsonValue::String;
            v.str = string();
        } else if (lit("true")) { v.kind = JsonValue::Bool; v.b = true; }
        else if (lit("false")) { v.kind = JsonValue::Bool; v.b = false; }
        else if (lit("null")) { v.kind = JsonValue::Null; }
        else {
            const char *s = p_;
            if (p_ < end_ && (*p_ == '-' || *p_ == '+')) ++p_;
            while (p_ < end_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '-' || *p_ == '+')) ++p_;
            if (p_ == s) fail("value expected");
            std::string tmp(s, p_);
            char *endp = nullptr;
            v.kind = JsonValue::Number;
            v.num = std::strtod(tmp.c_str(), &endp);   // correctly rounded decimal -> double, like tinygltf's json layer
            if (endp == tmp.c_str()) fail("bad number");
        }
        return v;
    }
};

}  // namespace drt
