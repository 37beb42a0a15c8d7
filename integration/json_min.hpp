// json_min.hpp — small recursive-descent JSON reader (RFC 8259: objects, arrays, strings with escapes,
// numbers, true/false/null), sufficient for .scene files.
//
// Numbers follow the reference's JSON library (lib/json.h, nlohmann/json 2.1.1, get_number at :11259-11310)
// because loadScene narrows them straight from its stored type (scene.cpp:53-57 `float x = i[0]`,
// :97-99 `int x = i["center"][0]`): a token without fraction and exponent is stored as an unsigned or
// signed 64-bit integer when it fits ("-0" is the integer 0, hence +0.0f), every other token as the
// double strtod gives; f32() / i32() are the static_casts from that stored type.  A token whose value
// is not finite (1e999) is rejected here (the reference stores null and then throws on conversion).
#pragma once
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt_hip {

class Json {
public:
    enum Kind { Null, Bool, Number, String, Array, Object };
    enum NumKind { Float, Integer, Unsigned };
    Kind kind = Null;
    NumKind num_kind = Float;
    bool b = false;
    double num = 0.0;             // the value as a double (exact for Float; rounded for 64-bit integers beyond 2^53)
    int64_t i64 = 0;              // stored value of an Integer token
    uint64_t u64 = 0;             // stored value of an Unsigned token
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;     // insertion order; lookups are linear (small objects)

    bool is_null() const { return kind == Null; }
    size_t size() const { return kind == Array ? arr.size() : (kind == Object ? obj.size() : 0); }
    const Json &at(size_t i) const {
        if (kind != Array || i >= arr.size()) throw std::runtime_error("json: array index out of range");
        return arr[i];
    }
    // member or a shared null value when absent
    const Json &get(const char *key) const {
        static const Json none;
        if (kind != Object) return none;
        for (const auto &kv : obj) if (kv.first == key) return kv.second;
        return none;
    }
    const Json &need(const char *key) const {
        const Json &v = get(key);
        if (v.is_null()) throw std::runtime_error(std::string("json: missing member '") + key + "'");
        return v;
    }
    double number() const {
        if (kind != Number) throw std::runtime_error("json: number expected");
        return num;
    }
    // `float x = value;` / `int x = value;` of the reference's library: static_cast from the stored type
    float f32() const {
        if (kind != Number) throw std::runtime_error("json: number expected");
        return num_kind == Integer ? static_cast<float>(i64) : (num_kind == Unsigned ? static_cast<float>(u64) : static_cast<float>(num));
    }
    int32_t i32() const {
        if (kind != Number) throw std::runtime_error("json: number expected");
        if (num_kind == Integer) return static_cast<int32_t>(i64);
        if (num_kind == Unsigned) return static_cast<int32_t>(u64);
        // a double outside the int range is undefined in the reference's cast; x86-64 gives INT_MIN, kept here
        if (!(num > -2147483649.0 && num < 2147483648.0)) return INT32_MIN;
        return static_cast<int32_t>(num);
    }
    const std::string &string() const {
        if (kind != String) throw std::runtime_error("json: string expected");
        return str;
    }

    static constexpr int MAX_DEPTH = 256;      // nesting limit: a hostile file cannot overflow the stack

    static Json parse(const std::string &text) {
        Parser p{text.c_str(), text.c_str() + text.size(), 0};
        Json v = p.value();
        p.ws();
        if (p.cur != p.end) p.fail("trailing characters");
        return v;
    }

private:
    struct Parser {
        const char *cur, *end;
        int depth;
        [[noreturn]] void fail(const char *what) const { throw std::runtime_error(std::string("json: ") + what); }
        void ws() { while (cur < end && (*cur == ' ' || *cur == '\t' || *cur == '\n' || *cur == '\r')) cur++; }
        bool eat(char c) { ws(); if (cur < end && *cur == c) { cur++; return true; } return false; }
        void expect(char c) { if (!eat(c)) fail("unexpected character"); }
        void literal(const char *w) {
            size_t n = strlen(w);
            if ((size_t)(end - cur) < n || strncmp(cur, w, n) != 0) fail("bad literal");
            cur += n;
        }
        static void utf8(std::string &out, unsigned cp) {
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        }
        unsigned hex4() {
            if (end - cur < 4) fail("bad \\u escape");
            unsigned v = 0;
            for (int i = 0; i < 4; i++) {
                char c = *cur++;
                v <<= 4;
                if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
                else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
                else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
                else fail("bad \\u escape");
            }
            return v;
        }
        std::string string_body() {
            std::string out;
            for (;;) {
                if (cur >= end) fail("unterminated string");
                char c = *cur++;
                if (c == '"') return out;
                if (c != '\\') { out += c; continue; }
                if (cur >= end) fail("unterminated escape");
                char e = *cur++;
                switch (e) {
                case '"': out += '"'; break; case '\\': out += '\\'; break; case '/': out += '/'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break; case 'n': out += '\n'; break;
                case 'r': out += '\r'; break; case 't': out += '\t'; break;
                case 'u': {
                    unsigned cp = hex4();
                    if (cp >= 0xD800 && cp < 0xDC00 && end - cur >= 6 && cur[0] == '\\' && cur[1] == 'u') {
                        cur += 2;
                        unsigned lo = hex4();
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    utf8(out, cp);
                    break;
                }
                default: fail("bad escape");
                }
            }
        }
        // number token: -? (0 | [1-9][0-9]*) (. [0-9]+)? ([eE] [+-]? [0-9]+)?
        void number_token(Json &v) {
            const char *start = cur;
            bool neg = false, integral = true;
            auto digit = [&]() { return cur < end && *cur >= '0' && *cur <= '9'; };
            if (cur < end && *cur == '-') { neg = true; cur++; }
            if (!digit()) fail("number expected");
            if (*cur == '0') cur++; else while (digit()) cur++;
            if (cur < end && *cur == '.') { integral = false; cur++; if (!digit()) fail("digit expected after '.'"); while (digit()) cur++; }
            if (cur < end && (*cur == 'e' || *cur == 'E')) {
                integral = false; cur++;
                if (cur < end && (*cur == '+' || *cur == '-')) cur++;
                if (!digit()) fail("digit expected in exponent");
                while (digit()) cur++;
            }
            std::string tok(start, cur);
            v.kind = Number; v.num_kind = Float;
            if (integral) {
                errno = 0;
                char *stop = nullptr;
                if (neg) {
                    long long x = strtoll(tok.c_str(), &stop, 10);
                    if (errno == 0 && *stop == 0) { v.num_kind = Integer; v.i64 = x; v.num = (double)x; return; }
                } else {
                    unsigned long long x = strtoull(tok.c_str(), &stop, 10);
                    if (errno == 0 && *stop == 0) { v.num_kind = Unsigned; v.u64 = x; v.num = (double)x; return; }
                }
            }
            v.num = strtod(tok.c_str(), nullptr);
            if (!std::isfinite(v.num)) fail("number out of range");
        }
        Json value() {
            ws();
            if (cur >= end) fail("unexpected end");
            if (++depth > MAX_DEPTH) fail("nesting too deep");
            Json v = value_at_depth();
            depth--;
            return v;
        }
        Json value_at_depth() {
            Json v;
            char c = *cur;
            if (c == '{') {
                cur++; v.kind = Object;
                if (eat('}')) return v;
                do {
                    ws();
                    if (cur >= end || *cur != '"') fail("member name expected");
                    cur++;
                    std::string k = string_body();
                    expect(':');
                    v.obj.emplace_back(std::move(k), value());
                } while (eat(','));
                expect('}');
            } else if (c == '[') {
                cur++; v.kind = Array;
                if (eat(']')) return v;
                do v.arr.push_back(value()); while (eat(','));
                expect(']');
            } else if (c == '"') {
                cur++; v.kind = String; v.str = string_body();
            } else if (c == 't') { literal("true"); v.kind = Bool; v.b = true; }
            else if (c == 'f') { literal("false"); v.kind = Bool; v.b = false; }
            else if (c == 'n') { literal("null"); v.kind = Null; }
            else number_token(v);
            return v;
        }
    };
};

}  // namespace rt_hip
