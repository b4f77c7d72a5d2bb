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

private:
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
