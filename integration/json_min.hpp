// json_min.hpp — small recursive-descent JSON reader (RFC 8259 subset sufficient for .scene files:
// objects, arrays, strings with escapes, numbers, true/false/null).  Numbers are kept as double,
// like the reference's JSON library does before they are narrowed (scene.cpp:24-134).
#pragma once
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
    Kind kind = Null;
    bool b = false;
    double num = 0.0;
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
    const std::string &string() const {
        if (kind != String) throw std::runtime_error("json: string expected");
        return str;
    }

    static Json parse(const std::string &text) {
        Parser p{text.c_str(), text.c_str() + text.size()};
        Json v = p.value();
        p.ws();
        if (p.cur != p.end) p.fail("trailing characters");
        return v;
    }

private:
    struct Parser {
        const char *cur, *end;
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
        Json value() {
            ws();
            if (cur >= end) fail("unexpected end");
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
            else {
                char *stop = nullptr;
                v.num = strtod(cur, &stop);
                if (stop == cur || stop > end) fail("number expected");
                cur = stop; v.kind = Number;
            }
            return v;
        }
    };
};

}  // namespace rt_hip
