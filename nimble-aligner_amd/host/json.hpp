// json.hpp -- minimal JSON reader for library files (replaces serde_json at reference_library.rs:24-25).
// Strict RFC 8259 subset: objects, arrays, strings (with escapes), numbers, true/false/null.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace nimble {
namespace json {

struct Value {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0.0;
  bool is_int = false;  // number written without fraction / exponent and fitting i64
  int64_t i = 0;
  std::string str;
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj;

  // serde_json indexing semantics: a missing key / wrong kind yields Null
  const Value &operator[](const std::string &key) const {
    static const Value null_value;
    if (kind != Object) return null_value;
    for (const auto &kv : obj)
      if (kv.first == key) return kv.second;
    return null_value;
  }
  const Value &operator[](size_t idx) const {
    static const Value null_value;
    if (kind != Array || idx >= arr.size()) return null_value;
    return arr[idx];
  }
  bool as_f64(double &out) const {
    if (kind != Number) return false;
    out = num;
    return true;
  }
  bool as_i64(int64_t &out) const {
    if (kind != Number || !is_int) return false;
    out = i;
    return true;
  }
  bool as_bool(bool &out) const {
    if (kind != Bool) return false;
    out = b;
    return true;
  }
  const std::string *as_str() const { return kind == String ? &str : nullptr; }
  const std::vector<Value> *as_array() const { return kind == Array ? &arr : nullptr; }
};

class Parser {
 public:
  explicit Parser(const std::string &text) : s_(text) {}
  Value parse() {
    Value v = value();
    ws();
    if (p_ != s_.size()) throw std::runtime_error("trailing characters");
    return v;
  }

 private:
  const std::string &s_;
  size_t p_ = 0;
  void ws() {
    while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\t' || s_[p_] == '\n' || s_[p_] == '\r')) ++p_;
  }
  [[noreturn]] void bad(const char *what) { throw std::runtime_error(std::string("JSON: ") + what); }
  Value value() {
    ws();
    if (p_ >= s_.size()) bad("unexpected end");
    char c = s_[p_];
    if (c == '{') return object();
    if (c == '[') return array();
    if (c == '"') {
      Value v;
      v.kind = Value::String;
      v.str = string();
      return v;
    }
    if (c == 't' || c == 'f' || c == 'n') return literal();
    return number();
  }
  Value literal() {
    Value v;
    if (s_.compare(p_, 4, "true") == 0) { v.kind = Value::Bool; v.b = true; p_ += 4; }
    else if (s_.compare(p_, 5, "false") == 0) { v.kind = Value::Bool; v.b = false; p_ += 5; }
    else if (s_.compare(p_, 4, "null") == 0) { v.kind = Value::Null; p_ += 4; }
    else bad("bad literal");
    return v;
  }
  Value number() {
    size_t start = p_;
    bool is_int = true;
    if (p_ < s_.size() && s_[p_] == '-') ++p_;
    if (p_ >= s_.size() || !(s_[p_] >= '0' && s_[p_] <= '9')) bad("bad number");
    while (p_ < s_.size() && s_[p_] >= '0' && s_[p_] <= '9') ++p_;
    if (p_ < s_.size() && s_[p_] == '.') {
      is_int = false;
      ++p_;
      if (p_ >= s_.size() || !(s_[p_] >= '0' && s_[p_] <= '9')) bad("bad fraction");
      while (p_ < s_.size() && s_[p_] >= '0' && s_[p_] <= '9') ++p_;
    }
    if (p_ < s_.size() && (s_[p_] == 'e' || s_[p_] == 'E')) {
      is_int = false;
      ++p_;
      if (p_ < s_.size() && (s_[p_] == '+' || s_[p_] == '-')) ++p_;
      if (p_ >= s_.size() || !(s_[p_] >= '0' && s_[p_] <= '9')) bad("bad exponent");
      while (p_ < s_.size() && s_[p_] >= '0' && s_[p_] <= '9') ++p_;
    }
    Value v;
    v.kind = Value::Number;
    std::string tok = s_.substr(start, p_ - start);
    v.num = strtod(tok.c_str(), nullptr);
    if (is_int) {
      errno = 0;
      long long ll = strtoll(tok.c_str(), nullptr, 10);
      if (errno == 0) { v.is_int = true; v.i = ll; }
    }
    return v;
  }
  static void utf8(std::string &o, uint32_t cp) {
    if (cp < 0x80) o.push_back((char)cp);
    else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) {
      o.push_back((char)(0xE0 | (cp >> 12)));
      o.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      o.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
      o.push_back((char)(0xF0 | (cp >> 18)));
      o.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
      o.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      o.push_back((char)(0x80 | (cp & 0x3F)));
    }
  }
  uint32_t hex4() {
    if (p_ + 4 > s_.size()) bad("bad \\u escape");
    uint32_t v = 0;
    for (int k = 0; k < 4; ++k) {
      char c = s_[p_++];
      v <<= 4;
      if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
      else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
      else bad("bad \\u escape");
    }
    return v;
  }
  std::string string() {
    ++p_;  // opening quote
    std::string o;
    for (;;) {
      if (p_ >= s_.size()) bad("unterminated string");
      unsigned char c = (unsigned char)s_[p_++];
      if (c == '"') break;
      if (c < 0x20) bad("control character in string");
      if (c != '\\') { o.push_back((char)c); continue; }
      if (p_ >= s_.size()) bad("bad escape");
      char e = s_[p_++];
      switch (e) {
        case '"': o.push_back('"'); break;
        case '\\': o.push_back('\\'); break;
        case '/': o.push_back('/'); break;
        case 'b': o.push_back('\b'); break;
        case 'f': o.push_back('\f'); break;
        case 'n': o.push_back('\n'); break;
        case 'r': o.push_back('\r'); break;
        case 't': o.push_back('\t'); break;
        case 'u': {
          uint32_t cp = hex4();
          if (cp >= 0xD800 && cp < 0xDC00) {
            if (p_ + 2 > s_.size() || s_[p_] != '\\' || s_[p_ + 1] != 'u') bad("lone surrogate");
            p_ += 2;
            uint32_t lo = hex4();
            if (lo < 0xDC00 || lo > 0xDFFF) bad("bad surrogate pair");
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          utf8(o, cp);
          break;
        }
        default: bad("bad escape");
      }
    }
    return o;
  }
  Value array() {
    ++p_;
    Value v;
    v.kind = Value::Array;
    ws();
    if (p_ < s_.size() && s_[p_] == ']') { ++p_; return v; }
    for (;;) {
      v.arr.push_back(value());
      ws();
      if (p_ >= s_.size()) bad("unterminated array");
      if (s_[p_] == ',') { ++p_; continue; }
      if (s_[p_] == ']') { ++p_; break; }
      bad("expected , or ]");
    }
    return v;
  }
  Value object() {
    ++p_;
    Value v;
    v.kind = Value::Object;
    ws();
    if (p_ < s_.size() && s_[p_] == '}') { ++p_; return v; }
    for (;;) {
      ws();
      if (p_ >= s_.size() || s_[p_] != '"') bad("expected key");
      std::string k = string();
      ws();
      if (p_ >= s_.size() || s_[p_] != ':') bad("expected :");
      ++p_;
      Value val = value();
      bool replaced = false;
      for (auto &kv : v.obj)
        if (kv.first == k) { kv.second = val; replaced = true; break; }  // serde_json: last key wins
      if (!replaced) v.obj.emplace_back(k, std::move(val));
      ws();
      if (p_ >= s_.size()) bad("unterminated object");
      if (s_[p_] == ',') { ++p_; continue; }
      if (s_[p_] == '}') { ++p_; break; }
      bad("expected , or }");
    }
    return v;
  }
};

inline Value parse(const std::string &text) { return Parser(text).parse(); }

}  // namespace json
}  // namespace nimble
