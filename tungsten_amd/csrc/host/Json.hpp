// Minimal JSON DOM for Tungsten scene files (host side, C++11).
//
// The reference reads scenes through rapidjson + JsonPtr (src/core/io/JsonPtr.cpp,
// JsonDocument.cpp).  We only need the read side of that contract: objects, arrays,
// numbers (parsed as double, narrowed to float exactly like JsonPtr::cast<float>),
// strings, booleans and null, with "scalar broadcasts to vector" (JsonPtr.hpp:52-65).
#ifndef TGAMD_JSON_HPP_
#define TGAMD_JSON_HPP_

#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace tungsten_amd {

struct JsonLoadException : std::runtime_error {
    explicit JsonLoadException(const std::string &what) : std::runtime_error(what) {}
};

class JsonValue
{
public:
    enum Type { Null, Bool, Number, String, Array, Object };

private:
    Type _type = Null;
    bool _b = false;
    double _num = 0.0;
    std::string _str;
    std::vector<JsonValue> _arr;
    std::vector<std::pair<std::string, JsonValue>> _obj; // keeps file order (needed: bsdfs/primitives are ordered)

    friend class JsonParser;

public:
    Type type() const { return _type; }
    bool isNull()   const { return _type == Null; }
    bool isBool()   const { return _type == Bool; }
    bool isNumber() const { return _type == Number; }
    bool isString() const { return _type == String; }
    bool isArray()  const { return _type == Array; }
    bool isObject() const { return _type == Object; }
    explicit operator bool() const { return _type != Null; }

    size_t size() const { return _type == Array ? _arr.size() : _obj.size(); }
    const JsonValue &operator[](size_t i) const { return _arr.at(i); }
    const JsonValue &operator[](int i) const { return _arr.at(size_t(i)); }

    // Missing members yield a shared Null value (mirrors JsonPtr's "falsy" pointer).
    const JsonValue &operator[](const char *key) const
    {
        static const JsonValue nullValue;
        if (_type != Object) return nullValue;
        for (const auto &kv : _obj)
            if (kv.first == key) return kv.second;
        return nullValue;
    }
    const std::vector<std::pair<std::string, JsonValue>> &members() const { return _obj; }

    double asDouble() const
    {
        if (_type != Number) throw JsonLoadException("JSON: expected a number");
        return _num;
    }
    float asFloat() const { return float(asDouble()); }
    int asInt() const { return int(asDouble()); }
    bool asBool() const
    {
        if (_type != Bool) throw JsonLoadException("JSON: expected a boolean");
        return _b;
    }
    const std::string &asString() const
    {
        if (_type != String) throw JsonLoadException("JSON: expected a string");
        return _str;
    }

    bool getField(const char *key, float &dst) const { const JsonValue &v = (*this)[key]; if (!v) return false; dst = v.asFloat(); return true; }
    bool getField(const char *key, int &dst)   const { const JsonValue &v = (*this)[key]; if (!v) return false; dst = v.asInt(); return true; }
    bool getField(const char *key, unsigned &dst) const { const JsonValue &v = (*this)[key]; if (!v) return false; dst = unsigned(v.asDouble()); return true; }
    bool getField(const char *key, bool &dst)  const { const JsonValue &v = (*this)[key]; if (!v) return false; dst = v.asBool(); return true; }
    bool getField(const char *key, std::string &dst) const { const JsonValue &v = (*this)[key]; if (!v) return false; dst = v.asString(); return true; }

    static JsonValue parse(const std::string &text);
    static JsonValue parseFile(const std::string &path);
};

class JsonParser
{
    const std::string &_s;
    size_t _p = 0;

    [[noreturn]] void fail(const std::string &msg) const
    {
        size_t line = 1;
        for (size_t i = 0; i < _p && i < _s.size(); ++i) if (_s[i] == '\n') line++;
        throw JsonLoadException("JSON parse error (line " + std::to_string(line) + "): " + msg);
    }
    void ws()
    {
        while (_p < _s.size()) {
            char c = _s[_p];
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r') { _p++; continue; }
            if (c == '/' && _p + 1 < _s.size() && _s[_p + 1] == '/') { // tolerate // comments
                while (_p < _s.size() && _s[_p] != '\n') _p++;
                continue;
            }
            break;
        }
    }
    std::string parseString()
    {
        std::string out;
        _p++; // opening quote
        while (true) {
            if (_p >= _s.size()) fail("unterminated string");
            char c = _s[_p++];
            if (c == '"') break;
            if (c == '\\') {
                if (_p >= _s.size()) fail("bad escape");
                char e = _s[_p++];
                switch (e) {
                case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                case 'u': {
                    if (_p + 4 > _s.size()) fail("bad \\u escape");
                    unsigned cp = unsigned(std::strtoul(_s.substr(_p, 4).c_str(), nullptr, 16));
                    _p += 4;
                    if (cp < 0x80) out += char(cp);
                    else if (cp < 0x800) { out += char(0xC0 | (cp >> 6)); out += char(0x80 | (cp & 0x3F)); }
                    else { out += char(0xE0 | (cp >> 12)); out += char(0x80 | ((cp >> 6) & 0x3F)); out += char(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: out += e;
                }
            } else {
                out += c;
            }
        }
        return out;
    }
    JsonValue parseValue()
    {
        ws();
        if (_p >= _s.size()) fail("unexpected end of input");
        JsonValue v;
        char c = _s[_p];
        if (c == '{') {
            v._type = JsonValue::Object;
            _p++; ws();
            if (_p < _s.size() && _s[_p] == '}') { _p++; return v; }
            while (true) {
                ws();
                if (_p >= _s.size() || _s[_p] != '"') fail("expected member name");
                std::string key = parseString();
                ws();
                if (_p >= _s.size() || _s[_p] != ':') fail("expected ':'");
                _p++;
                v._obj.emplace_back(std::move(key), parseValue());
                ws();
                if (_p < _s.size() && _s[_p] == ',') { _p++; continue; }
                if (_p < _s.size() && _s[_p] == '}') { _p++; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v._type = JsonValue::Array;
            _p++; ws();
            if (_p < _s.size() && _s[_p] == ']') { _p++; return v; }
            while (true) {
                v._arr.push_back(parseValue());
                ws();
                if (_p < _s.size() && _s[_p] == ',') { _p++; continue; }
                if (_p < _s.size() && _s[_p] == ']') { _p++; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v._type = JsonValue::String;
            v._str = parseString();
        } else if (_s.compare(_p, 4, "true") == 0)  { v._type = JsonValue::Bool; v._b = true;  _p += 4; }
        else if (_s.compare(_p, 5, "false") == 0)   { v._type = JsonValue::Bool; v._b = false; _p += 5; }
        else if (_s.compare(_p, 4, "null") == 0)    { _p += 4; }
        else {
            const char *begin = _s.c_str() + _p;
            char *end = nullptr;
            double d = std::strtod(begin, &end);
            if (end == begin) fail("unexpected character");
            _p += size_t(end - begin);
            v._type = JsonValue::Number;
            v._num = d;
        }
        return v;
    }

public:
    explicit JsonParser(const std::string &s) : _s(s) {}
    JsonValue parseDocument()
    {
        JsonValue v = parseValue();
        ws();
        if (_p != _s.size()) fail("trailing characters");
        return v;
    }
};

inline JsonValue JsonValue::parse(const std::string &text) { return JsonParser(text).parseDocument(); }

} // namespace tungsten_amd

#endif
