// msq_json.hpp -- the small part of PacBio::JSON::Json (pbcopper's alias of nlohmann::json) that juliet's report
// needs: build a tree with operator[] / push_back, and dump(indent) it.
//
// The reference writes its result with `json.dump(2)` (src/tools/JulietWorkflow.cpp:165-170).  pbcopper is a
// submodule the reference does not vendor, so this restates nlohmann's published serializer: objects are
// std::map (keys in byte order), `dump(n)` puts every member / element on its own line indented by n per level,
// "key": value with one space, empty containers as {} / [], strings escaped (\" \\ \b \f \n \r \t, other control
// characters as \u00XX), integers in decimal, bools as true / false.  Floating-point numbers follow the 2.x
// line that pbcopper bundled in 2017: "%.15g" (digits10), with ".0" appended when the text has neither '.' nor
// 'e'.  A build against a later nlohmann prints the shortest round-trip form instead; the values are the same
// doubles.  Parity unpinned: the reference holds no JSON fixture.
#pragma once

#include <cstdint>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace PacBio {
namespace JSON {
class Json
{
public:
    enum class Type { Null, Object, Array, String, Int, UInt, Double, Bool };
    using object_t = std::map<std::string, Json>;
    using array_t = std::vector<Json>;

    Json() = default;
    Json(const char* s) : type_(Type::String), string_(s) {}
    Json(const std::string& s) : type_(Type::String), string_(s) {}
    Json(int v) : type_(Type::Int), int_(v) {}
    Json(long v) : type_(Type::Int), int_(v) {}
    Json(long long v) : type_(Type::Int), int_(v) {}
    Json(unsigned v) : type_(Type::UInt), uint_(v) {}
    Json(unsigned long v) : type_(Type::UInt), uint_(v) {}
    Json(unsigned long long v) : type_(Type::UInt), uint_(v) {}
    Json(double v) : type_(Type::Double), double_(v) {}
    Json(bool v) : type_(Type::Bool), bool_(v) {}
    template <typename T>
    Json(const std::vector<T>& v) : type_(Type::Array)
    {
        for (const auto& x : v) array_.emplace_back(Json(x));
    }
    Json(const std::vector<bool>& v) : type_(Type::Array)
    {
        for (const bool x : v) array_.emplace_back(Json(x));
    }

    Type type() const { return type_; }
    /// a null value becomes an object, as in nlohmann
    Json& operator[](const std::string& key)
    {
        if (type_ == Type::Null) type_ = Type::Object;
        return object_[key];
    }
    void push_back(const Json& v)
    {
        if (type_ == Type::Null) type_ = Type::Array;
        array_.push_back(v);
    }
    object_t::const_iterator find(const std::string& key) const { return object_.find(key); }
    object_t::const_iterator cend() const { return object_.cend(); }
    const object_t& object() const { return object_; }
    const array_t& array() const { return array_; }

    std::string dump(int indent = -1) const
    {
        std::string out;
        Write(out, indent, 0);
        return out;
    }

    // ---- reading (the target configuration: src/TargetConfig.cpp:120-265 parses one with Json::parse)
    /// RFC 8259 text -> tree; std::invalid_argument with the byte offset on malformed input.
    static Json parse(const std::string& text)
    {
        size_t at = 0;
        Json v = ParseValue(text, at, 0);
        SkipSpace(text, at);
        if (at != text.size()) Fail("trailing characters", at);
        return v;
    }
    /// nlohmann's empty(): null, or a container without elements; scalars are never empty
    bool empty() const { return type_ == Type::Null || (type_ == Type::Object && object_.empty()) || (type_ == Type::Array && array_.empty()); }
    bool is_string() const { return type_ == Type::String; }
    bool is_number() const { return type_ == Type::Int || type_ == Type::UInt || type_ == Type::Double; }
    /// member of an object; std::out_of_range when absent (nlohmann's const operator[] is undefined there)
    const Json& at(const std::string& key) const { return object_.at(key); }
    const std::string& AsString() const
    {
        if (type_ != Type::String) throw std::domain_error("type must be string");  // nlohmann: type_error.302
        return string_;
    }
    long long AsInt() const
    {
        if (type_ == Type::Int) return int_;
        if (type_ == Type::UInt) return static_cast<long long>(uint_);
        if (type_ == Type::Double) return static_cast<long long>(double_);
        throw std::domain_error("type must be number");
    }

private:
    [[noreturn]] static void Fail(const char* what, size_t at) { throw std::invalid_argument(std::string("JSON parse error: ") + what + " at byte " + std::to_string(at)); }
    static void SkipSpace(const std::string& t, size_t& at)
    {
        while (at < t.size() && (t[at] == ' ' || t[at] == '\t' || t[at] == '\n' || t[at] == '\r')) ++at;
    }
    static void AppendUtf8(std::string& out, unsigned cp)
    {
        if (cp < 0x80) {
            out += static_cast<char>(cp);
        } else if (cp < 0x800) {
            out += static_cast<char>(0xC0 | (cp >> 6));
            out += static_cast<char>(0x80 | (cp & 0x3F));
        } else if (cp < 0x10000) {
            out += static_cast<char>(0xE0 | (cp >> 12));
            out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
            out += static_cast<char>(0x80 | (cp & 0x3F));
        } else {
            out += static_cast<char>(0xF0 | (cp >> 18));
            out += static_cast<char>(0x80 | ((cp >> 12) & 0x3F));
            out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
            out += static_cast<char>(0x80 | (cp & 0x3F));
        }
    }
    static unsigned Hex4(const std::string& t, size_t& at)
    {
        if (at + 4 > t.size()) Fail("truncated \\u escape", at);
        unsigned v = 0;
        for (int i = 0; i < 4; ++i) {
            const char c = t[at++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0';
            else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
            else Fail("bad \\u escape", at - 1);
        }
        return v;
    }
    static std::string ParseString(const std::string& t, size_t& at)
    {
        std::string out;
        ++at;  // the opening quote
        for (;;) {
            if (at >= t.size()) Fail("unterminated string", at);
            const unsigned char c = static_cast<unsigned char>(t[at++]);
            if (c == '"') return out;
            if (c < 0x20) Fail("control character in string", at - 1);
            if (c != '\\') {
                out += static_cast<char>(c);
                continue;
            }
            if (at >= t.size()) Fail("unterminated escape", at);
            const char e = t[at++];
            switch (e) {
                case '"': out += '"'; break;
                case '\\': out += '\\'; break;
                case '/': out += '/'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'n': out += '\n'; break;
                case 'r': out += '\r'; break;
                case 't': out += '\t'; break;
                case 'u': {
                    unsigned cp = Hex4(t, at);
                    if (cp >= 0xD800 && cp < 0xDC00) {  // a surrogate pair
                        if (at + 2 > t.size() || t[at] != '\\' || t[at + 1] != 'u') Fail("lone surrogate", at);
                        at += 2;
                        const unsigned lo = Hex4(t, at);
                        if (lo < 0xDC00 || lo > 0xDFFF) Fail("bad low surrogate", at);
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    AppendUtf8(out, cp);
                    break;
                }
                default: Fail("bad escape", at - 1);
            }
        }
    }
    static Json ParseValue(const std::string& t, size_t& at, int depth)
    {
        if (depth > 200) Fail("nesting too deep", at);
        SkipSpace(t, at);
        if (at >= t.size()) Fail("unexpected end", at);
        const char c = t[at];
        if (c == '{') {
            Json v;
            v.type_ = Type::Object;
            ++at;
            SkipSpace(t, at);
            if (at < t.size() && t[at] == '}') {
                ++at;
                return v;
            }
            for (;;) {
                SkipSpace(t, at);
                if (at >= t.size() || t[at] != '"') Fail("expected a member name", at);
                std::string key = ParseString(t, at);
                SkipSpace(t, at);
                if (at >= t.size() || t[at] != ':') Fail("expected ':'", at);
                ++at;
                v.object_[key] = ParseValue(t, at, depth + 1);  // a repeated name: the last one wins, as in nlohmann
                SkipSpace(t, at);
                if (at < t.size() && t[at] == ',') {
                    ++at;
                    continue;
                }
                if (at < t.size() && t[at] == '}') {
                    ++at;
                    return v;
                }
                Fail("expected ',' or '}'", at);
            }
        }
        if (c == '[') {
            Json v;
            v.type_ = Type::Array;
            ++at;
            SkipSpace(t, at);
            if (at < t.size() && t[at] == ']') {
                ++at;
                return v;
            }
            for (;;) {
                v.array_.push_back(ParseValue(t, at, depth + 1));
                SkipSpace(t, at);
                if (at < t.size() && t[at] == ',') {
                    ++at;
                    continue;
                }
                if (at < t.size() && t[at] == ']') {
                    ++at;
                    return v;
                }
                Fail("expected ',' or ']'", at);
            }
        }
        if (c == '"') return Json(ParseString(t, at));
        if (t.compare(at, 4, "true") == 0) {
            at += 4;
            return Json(true);
        }
        if (t.compare(at, 5, "false") == 0) {
            at += 5;
            return Json(false);
        }
        if (t.compare(at, 4, "null") == 0) {
            at += 4;
            return Json();
        }
        // a number: -? int frac? exp?
        const size_t start = at;
        if (at < t.size() && t[at] == '-') ++at;
        if (at >= t.size() || t[at] < '0' || t[at] > '9') Fail("unexpected character", start);
        if (t[at] == '0') {
            ++at;
        } else {
            while (at < t.size() && t[at] >= '0' && t[at] <= '9') ++at;
        }
        bool integral = true;
        if (at < t.size() && t[at] == '.') {
            integral = false;
            ++at;
            if (at >= t.size() || t[at] < '0' || t[at] > '9') Fail("digit expected after '.'", at);
            while (at < t.size() && t[at] >= '0' && t[at] <= '9') ++at;
        }
        if (at < t.size() && (t[at] == 'e' || t[at] == 'E')) {
            integral = false;
            ++at;
            if (at < t.size() && (t[at] == '+' || t[at] == '-')) ++at;
            if (at >= t.size() || t[at] < '0' || t[at] > '9') Fail("digit expected in exponent", at);
            while (at < t.size() && t[at] >= '0' && t[at] <= '9') ++at;
        }
        const std::string num = t.substr(start, at - start);
        if (integral) {
            try {
                if (num[0] == '-') return Json(std::stoll(num));
                return Json(static_cast<unsigned long long>(std::stoull(num)));
            } catch (const std::out_of_range&) {
                // too large for 64 bits: a double, as in nlohmann
            }
        }
        return Json(std::stod(num));
    }

    static void Escape(std::string& out, const std::string& s)
    {
        out += '"';
        for (const unsigned char c : s) {
            switch (c) {
                case '"': out += "\\\""; break;
                case '\\': out += "\\\\"; break;
                case '\b': out += "\\b"; break;
                case '\f': out += "\\f"; break;
                case '\n': out += "\\n"; break;
                case '\r': out += "\\r"; break;
                case '\t': out += "\\t"; break;
                default:
                    if (c < 0x20) {
                        char buf[8];
                        std::snprintf(buf, sizeof buf, "\\u%04x", c);
                        out += buf;
                    } else {
                        out += static_cast<char>(c);
                    }
            }
        }
        out += '"';
    }
    void Write(std::string& out, int indent, int depth) const
    {
        const bool pretty = indent >= 0;
        auto newline = [&](int d) {
            if (pretty) {
                out += '\n';
                out.append(static_cast<size_t>(indent) * d, ' ');
            }
        };
        switch (type_) {
            case Type::Null: out += "null"; break;
            case Type::Bool: out += bool_ ? "true" : "false"; break;
            case Type::Int: out += std::to_string(int_); break;
            case Type::UInt: out += std::to_string(uint_); break;
            case Type::String: Escape(out, string_); break;
            case Type::Double: {
                if (double_ != double_ || double_ - double_ != 0) {  // nan / inf have no JSON form
                    out += "null";
                    break;
                }
                if (double_ == 0) {
                    out += (1 / double_ < 0) ? "-0.0" : "0.0";
                    break;
                }
                char buf[40];
                std::snprintf(buf, sizeof buf, "%.15g", double_);
                std::string s(buf);
                if (s.find_first_of(".e") == std::string::npos) s += ".0";
                out += s;
                break;
            }
            case Type::Object: {
                if (object_.empty()) {
                    out += "{}";
                    break;
                }
                out += '{';
                bool first = true;
                for (const auto& kv : object_) {
                    if (!first) out += ',';
                    first = false;
                    newline(depth + 1);
                    Escape(out, kv.first);
                    out += pretty ? ": " : ":";
                    kv.second.Write(out, indent, depth + 1);
                }
                newline(depth);
                out += '}';
                break;
            }
            case Type::Array: {
                if (array_.empty()) {
                    out += "[]";
                    break;
                }
                out += '[';
                bool first = true;
                for (const auto& v : array_) {
                    if (!first) out += ',';
                    first = false;
                    newline(depth + 1);
                    v.Write(out, indent, depth + 1);
                }
                newline(depth);
                out += ']';
                break;
            }
        }
    }

    Type type_ = Type::Null;
    object_t object_;
    array_t array_;
    std::string string_;
    long long int_ = 0;
    unsigned long long uint_ = 0;
    double double_ = 0;
    bool bool_ = false;
};
}  // namespace JSON
}  // namespace PacBio
