// json_min.hpp -- a small JSON reader for the two JSON files on the path's input side: config.json
// (docs/configuration.md, parsed by the reference with nlohmann::json in src/Util/ProgramArgumentParser.cu)
// and *.vtk.series (src/Util/VTKTimeReader.cu:31-88).  Objects, arrays, strings (with escapes), numbers,
// true / false / null; numbers are kept as double and as their source text.
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace hrt_io {

struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;        // insertion order kept

    bool contains(const std::string &k) const { for (auto &kv : obj) if (kv.first == k) return true; return false; }
    const Json &at(const std::string &k) const {
        if (kind != Object) throw std::runtime_error("JSON: not an object while looking up \"" + k + "\"");
        for (auto &kv : obj) if (kv.first == k) return kv.second;
        throw std::runtime_error("JSON: key \"" + k + "\" is missing");
    }
    const Json &operator[](size_t i) const {
        if (kind != Array || i >= arr.size()) throw std::runtime_error("JSON: array index out of range");
        return arr[i];
    }
    size_t size() const { return kind == Array ? arr.size() : kind == Object ? obj.size() : 0; }
    double number() const { if (kind != Number) throw std::runtime_error("JSON: number expected"); return num; }
    float number_f() const { return (float)number(); }     // nlohmann's get<float>() narrows the parsed double the same way
    bool boolean() const { if (kind != Bool) throw std::runtime_error("JSON: boolean expected"); return b; }
    const std::string &string() const { if (kind != String) throw std::runtime_error("JSON: string expected"); return str; }
    std::vector<float> floats(size_t want = 0) const {
        if (kind != Array) throw std::runtime_error("JSON: array expected");
        std::vector<float> v;
        for (auto &e : arr) v.push_back(e.number_f());
        if (want && v.size() < want) throw std::runtime_error("JSON: array too short");
        return v;
    }
};

class JsonParser {
public:
    explicit JsonParser(const std::string &text) : s_(text) {}
    Json parse() {
        Json v = value();
        ws();
        if (p_ != s_.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string &s_;
    size_t p_ = 0;
    [[noreturn]] void fail(const char *what) const { throw std::runtime_error(std::string("JSON parse error at byte ") + std::to_string(p_) + ": " + what); }
    void ws() { while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\t' || s_[p_] == '\n' || s_[p_] == '\r')) ++p_; }
    bool eat(char c) { ws(); if (p_ < s_.size() && s_[p_] == c) { ++p_; return true; } return false; }
    int depth_ = 0;
    struct Nest { int &d; explicit Nest(int &x) : d(x) { ++d; } ~Nest() { --d; } };
    Json value() {
        ws();
        if (p_ >= s_.size()) fail("unexpected end");
        const char c = s_[p_];
        // containers recurse: a file of 100 000 '[' would otherwise run the stack out (the formats read here nest 3 deep)
        if (c == '{' || c == '[') { if (depth_ >= 64) fail("nesting deeper than 64"); Nest n(depth_); return c == '{' ? object() : array(); }
        if (c == '"') { Json j; j.kind = Json::String; j.str = string(); return j; }
        if (s_.compare(p_, 4, "true") == 0) { p_ += 4; Json j; j.kind = Json::Bool; j.b = true; return j; }
        if (s_.compare(p_, 5, "false") == 0) { p_ += 5; Json j; j.kind = Json::Bool; j.b = false; return j; }
        if (s_.compare(p_, 4, "null") == 0) { p_ += 4; return Json(); }
        return number();
    }
    Json number() {
        const char *b = s_.c_str() + p_;
        char *e = nullptr;
        const double d = std::strtod(b, &e);
        if (e == b) fail("value expected");
        Json j; j.kind = Json::Number; j.num = d; j.str.assign(b, (size_t)(e - b));
        p_ += (size_t)(e - b);
        return j;
    }
    std::string string() {
        std::string out;
        ++p_;                                            // opening quote
        while (true) {
            if (p_ >= s_.size()) fail("unterminated string");
            const char c = s_[p_++];
            if (c == '"') break;
            if (c != '\\') { out.push_back(c); continue; }
            if (p_ >= s_.size()) fail("unterminated escape");
            const char e = s_[p_++];
            switch (e) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    if (p_ + 4 > s_.size()) fail("short \\u escape");
                    const unsigned cp = (unsigned)std::strtoul(s_.substr(p_, 4).c_str(), nullptr, 16);
                    p_ += 4;
                    if (cp < 0x80) out.push_back((char)cp);
                    else if (cp < 0x800) { out.push_back((char)(0xc0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3f))); }
                    else { out.push_back((char)(0xe0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3f))); out.push_back((char)(0x80 | (cp & 0x3f))); }
                    break;
                }
                default: fail("bad escape");
            }
        }
        return out;
    }
    Json array() {
        Json j; j.kind = Json::Array;
        ++p_;
        if (eat(']')) return j;
        while (true) {
            j.arr.push_back(value());
            if (eat(',')) continue;
            if (eat(']')) break;
            fail("',' or ']' expected");
        }
        return j;
    }
    Json object() {
        Json j; j.kind = Json::Object;
        ++p_;
        if (eat('}')) return j;
        while (true) {
            ws();
            if (p_ >= s_.size() || s_[p_] != '"') fail("key expected");
            std::string k = string();
            if (!eat(':')) fail("':' expected");
            j.obj.emplace_back(std::move(k), value());
            if (eat(',')) continue;
            if (eat('}')) break;
            fail("',' or '}' expected");
        }
        return j;
    }
};

}  // namespace hrt_io
