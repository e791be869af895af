// toml_subset.h -- a small TOML reader for model packages (header-only, std-only).
//
// The reference parses a package's `.toml` with toml11 (reference src/common/model_config.h:13,
// processor_proxy.h:55-56), an external submodule.  A model package needs little of TOML: tables
// (`[model]`, `[voice.0]`, `[voice.0.portrait]`), dotted and quoted keys, strings (basic, literal and their
// multi-line forms, with the standard escapes), integers, floats, booleans, plus -- so that a hand-edited file does not
// become a syntax error -- arrays, inline tables and date-times (kept as opaque text).  What the reader reports is
// what the reference distinguishes (processor_proxy.h:76-93): the file cannot be opened, the text is not TOML
// (SyntaxError), or a value has the wrong type / is missing (TypeError, thrown by the typed getters).  Types are
// strict as in toml11: an integer is not a float.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace beatrice_amd::toml_subset {

struct FileError : std::runtime_error { using std::runtime_error::runtime_error; };
struct SyntaxError : std::runtime_error { using std::runtime_error::runtime_error; };
struct TypeError : std::runtime_error { using std::runtime_error::runtime_error; };

struct Value;
using Table = std::map<std::string, Value>;
struct Value {
  enum Kind { kString, kInteger, kFloat, kBoolean, kArray, kTable, kDateTime } kind = kTable;
  std::string s;          // kString (UTF-8 bytes), kDateTime (as written)
  std::int64_t i = 0;
  double f = 0.0;
  bool b = false;
  std::vector<Value> array;
  std::shared_ptr<Table> table = std::make_shared<Table>();
  bool defined_by_header = false, closed_inline = false;

  const Value& at(const std::string& key) const {
    if (kind != kTable) throw TypeError("not a table");
    const auto it = table->find(key);
    if (it == table->end()) throw TypeError("key not found: " + key);   // toml11: find() on a missing key throws (out_of_range -> type-ish)
    return it->second;
  }
  bool has(const std::string& key) const { return kind == kTable && table->count(key) != 0; }
  const std::string& as_string() const { if (kind != kString) throw TypeError("not a string"); return s; }
  double as_float() const { if (kind != kFloat) throw TypeError("not a float"); return f; }
  std::int64_t as_integer() const { if (kind != kInteger) throw TypeError("not an integer"); return i; }
  const Table& as_table() const { if (kind != kTable) throw TypeError("not a table"); return *table; }
};

class Parser {
 public:
  explicit Parser(std::string text) : t_(std::move(text)) {}
  Value Parse() {
    Value root;
    Value* cur = &root;
    for (;;) {
      SkipBlankAndComments();
      if (End()) break;
      if (Peek() == '[') {
        ++p_;
        const bool aot = Peek() == '[';
        if (aot) ++p_;
        SkipWs();
        const std::vector<std::string> path = KeyPath();
        SkipWs();
        Expect(']');
        if (aot) Expect(']');
        EndOfLine();
        cur = aot ? OpenArrayTable(root, path) : OpenTable(root, path);
        continue;
      }
      const std::vector<std::string> path = KeyPath();
      SkipWs();
      Expect('=');
      SkipWs();
      Value v = ParseValue();
      EndOfLine();
      Value* t = cur;
      for (size_t k = 0; k + 1 < path.size(); ++k) t = Descend(*t, path[k], /*header=*/false);
      if (t->table->count(path.back())) Fail("duplicate key " + path.back());
      (*t->table)[path.back()] = std::move(v);
    }
    return root;
  }

 private:
  std::string t_;
  size_t p_ = 0;

  [[noreturn]] void Fail(const std::string& why) const {
    size_t line = 1;
    for (size_t k = 0; k < p_ && k < t_.size(); ++k) if (t_[k] == '\n') ++line;
    throw SyntaxError("line " + std::to_string(line) + ": " + why);
  }
  bool End() const { return p_ >= t_.size(); }
  char Peek(size_t ahead = 0) const { return p_ + ahead < t_.size() ? t_[p_ + ahead] : '\0'; }
  void Expect(char c) { if (Peek() != c) Fail(std::string("expected '") + c + "'"); ++p_; }
  void SkipWs() { while (Peek() == ' ' || Peek() == '\t') ++p_; }
  void SkipComment() {
    if (Peek() != '#') return;
    while (!End() && Peek() != '\n') {
      const unsigned char c = (unsigned char)Peek();
      if ((c < 0x20 && c != '\t' && c != '\r') || c == 0x7f) Fail("control character in comment");
      ++p_;
    }
  }
  void SkipBlankAndComments() {
    for (;;) {
      SkipWs();
      SkipComment();
      if (Peek() == '\r' && Peek(1) == '\n') { p_ += 2; continue; }
      if (Peek() == '\n') { ++p_; continue; }
      break;
    }
  }
  void EndOfLine() {
    SkipWs();
    SkipComment();
    if (End()) return;
    if (Peek() == '\r' && Peek(1) == '\n') { p_ += 2; return; }
    if (Peek() == '\n') { ++p_; return; }
    Fail("unexpected text after value");
  }
  static bool BareKeyChar(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == '-'; }
  std::vector<std::string> KeyPath() {
    std::vector<std::string> path;
    for (;;) {
      SkipWs();
      if (Peek() == '"') path.push_back(BasicString());
      else if (Peek() == '\'') path.push_back(LiteralString());
      else {
        const size_t a = p_;
        while (BareKeyChar(Peek())) ++p_;
        if (p_ == a) Fail("key expected");
        path.push_back(t_.substr(a, p_ - a));
      }
      SkipWs();
      if (Peek() != '.') break;
      ++p_;
    }
    return path;
  }
  Value* Descend(Value& parent, const std::string& key, bool header) {
    auto it = parent.table->find(key);
    if (it == parent.table->end()) {
      Value t;
      t.kind = Value::kTable;
      it = parent.table->emplace(key, std::move(t)).first;
    }
    Value* v = &it->second;
    if (v->kind == Value::kArray && header && !v->array.empty() && v->array.back().kind == Value::kTable) v = &v->array.back();
    if (v->kind != Value::kTable || v->closed_inline) Fail("key " + key + " is not a table");
    return v;
  }
  Value* OpenTable(Value& root, const std::vector<std::string>& path) {
    Value* t = &root;
    for (const std::string& k : path) t = Descend(*t, k, true);
    if (t->defined_by_header) Fail("table defined twice");
    t->defined_by_header = true;
    return t;
  }
  Value* OpenArrayTable(Value& root, const std::vector<std::string>& path) {
    Value* t = &root;
    for (size_t k = 0; k + 1 < path.size(); ++k) t = Descend(*t, path[k], true);
    Value& arr = (*t->table)[path.back()];
    if (arr.kind == Value::kTable && arr.table->empty() && !arr.defined_by_header) arr.kind = Value::kArray;
    if (arr.kind != Value::kArray) Fail("not an array of tables");
    Value e;
    e.kind = Value::kTable;
    e.defined_by_header = true;
    arr.array.push_back(std::move(e));
    return &arr.array.back();
  }
  static void AppendUtf8(std::string& out, std::uint32_t cp) {
    if (cp < 0x80) out += (char)cp;
    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
  }
  void Escape(std::string& out) {  // after the backslash
    const char c = Peek();
    ++p_;
    switch (c) {
      case 'b': out += '\b'; return;
      case 't': out += '\t'; return;
      case 'n': out += '\n'; return;
      case 'f': out += '\f'; return;
      case 'r': out += '\r'; return;
      case 'e': out += '\x1b'; return;
      case '"': out += '"'; return;
      case '\\': out += '\\'; return;
      case 'u': case 'U': {
        const int n = c == 'u' ? 4 : 8;
        std::uint32_t cp = 0;
        for (int k = 0; k < n; ++k) {
          const char h = Peek();
          ++p_;
          int d;
          if (h >= '0' && h <= '9') d = h - '0'; else if (h >= 'a' && h <= 'f') d = h - 'a' + 10; else if (h >= 'A' && h <= 'F') d = h - 'A' + 10; else Fail("bad unicode escape");
          cp = cp * 16 + (std::uint32_t)d;
        }
        if (cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) Fail("escape is not a unicode scalar value");
        AppendUtf8(out, cp);
        return;
      }
      default: Fail("unknown escape");
    }
  }
  std::string BasicString() {
    Expect('"');
    if (Peek() == '"' && Peek(1) == '"') {  // multi-line
      p_ += 2;
      if (Peek() == '\r' && Peek(1) == '\n') p_ += 2; else if (Peek() == '\n') ++p_;
      std::string out;
      for (;;) {
        if (End()) Fail("unterminated string");
        if (Peek() == '"' && Peek(1) == '"' && Peek(2) == '"') {  // a run of 3..5 quotes: the last three close the string
          int n = 3;
          while (n < 5 && Peek(n) == '"') ++n;
          out.append((size_t)(n - 3), '"');
          p_ += n;
          return out;
        }
        if (Peek() == '\\') {
          ++p_;
          size_t q = p_;
          while (q < t_.size() && (t_[q] == ' ' || t_[q] == '\t')) ++q;
          if (q < t_.size() && (t_[q] == '\n' || (t_[q] == '\r' && q + 1 < t_.size() && t_[q + 1] == '\n'))) {  // line-ending backslash
            p_ = q;
            while (!End() && (Peek() == ' ' || Peek() == '\t' || Peek() == '\n' || Peek() == '\r')) ++p_;
            continue;
          }
          Escape(out);
          continue;
        }
        out += Peek();
        ++p_;
      }
    }
    std::string out;
    for (;;) {
      if (End() || Peek() == '\n') Fail("unterminated string");
      const char c = Peek();
      if (c == '"') { ++p_; return out; }
      if (c == '\\') { ++p_; Escape(out); continue; }
      if (((unsigned char)c < 0x20 && c != '\t') || c == 0x7f) Fail("control character in string");
      out += c;
      ++p_;
    }
  }
  std::string LiteralString() {
    Expect('\'');
    if (Peek() == '\'' && Peek(1) == '\'') {
      p_ += 2;
      if (Peek() == '\r' && Peek(1) == '\n') p_ += 2; else if (Peek() == '\n') ++p_;
      const size_t e = t_.find("'''", p_);
      if (e == std::string::npos) Fail("unterminated string");
      size_t n = 3;
      while (n < 5 && e + n < t_.size() && t_[e + n] == '\'') ++n;   // up to two quotes may precede the delimiter
      std::string out = t_.substr(p_, e + (n - 3) - p_);
      p_ = e + n;
      return out;
    }
    const size_t a = p_;
    while (!End() && Peek() != '\'' && Peek() != '\n') ++p_;
    if (Peek() != '\'') Fail("unterminated string");
    std::string out = t_.substr(a, p_ - a);
    ++p_;
    return out;
  }
  // Arrays and inline tables nest by recursion: a cap keeps a hostile file ("[[[[[[...") from overflowing the stack.
  static constexpr int kMaxNesting = 64;
  int depth_ = 0;
  struct DepthGuard {
    Parser& p;
    explicit DepthGuard(Parser& p_) : p(p_) { if (++p.depth_ > kMaxNesting) p.Fail("values nested too deeply"); }
    ~DepthGuard() { --p.depth_; }
  };
  Value ParseValue() {
    const DepthGuard guard(*this);
    Value v;
    const char c = Peek();
    if (c == '"') { v.kind = Value::kString; v.s = BasicString(); return v; }
    if (c == '\'') { v.kind = Value::kString; v.s = LiteralString(); return v; }
    if (c == '[') {
      ++p_;
      v.kind = Value::kArray;
      for (;;) {
        SkipBlankAndComments();
        if (Peek() == ']') { ++p_; return v; }
        v.array.push_back(ParseValue());
        SkipBlankAndComments();
        if (Peek() == ',') { ++p_; continue; }
        SkipBlankAndComments();
        Expect(']');
        return v;
      }
    }
    if (c == '{') {
      ++p_;
      v.kind = Value::kTable;
      SkipWs();
      if (Peek() == '}') { ++p_; v.closed_inline = true; return v; }
      for (;;) {
        SkipWs();
        const std::vector<std::string> path = KeyPath();
        SkipWs();
        Expect('=');
        SkipWs();
        Value e = ParseValue();
        Value* t = &v;
        for (size_t k = 0; k + 1 < path.size(); ++k) t = Descend(*t, path[k], false);
        if (t->table->count(path.back())) Fail("duplicate key " + path.back());
        (*t->table)[path.back()] = std::move(e);
        SkipWs();
        if (Peek() == ',') { ++p_; continue; }
        Expect('}');
        v.closed_inline = true;
        return v;
      }
    }
    // scalars: read the token up to a delimiter
    const size_t a = p_;
    while (!End() && Peek() != ',' && Peek() != ']' && Peek() != '}' && Peek() != '#' && Peek() != '\n' && Peek() != '\r') ++p_;
    size_t b = p_;
    while (b > a && (t_[b - 1] == ' ' || t_[b - 1] == '\t')) --b;
    const std::string tok = t_.substr(a, b - a);
    if (tok.empty()) Fail("value expected");
    if (tok == "true" || tok == "false") { v.kind = Value::kBoolean; v.b = tok == "true"; return v; }
    if (LooksLikeDateTime(tok)) { v.kind = Value::kDateTime; v.s = tok; return v; }
    if (ParseNumber(tok, &v)) return v;
    Fail("cannot read value '" + tok + "'");
  }
  static bool LooksLikeDateTime(const std::string& s) {
    auto digit = [&](size_t k) { return k < s.size() && s[k] >= '0' && s[k] <= '9'; };
    if (s.size() >= 10 && digit(0) && digit(1) && digit(2) && digit(3) && s[4] == '-' && digit(5) && digit(6) && s[7] == '-' && digit(8) && digit(9)) return true;
    if (s.size() >= 8 && digit(0) && digit(1) && s[2] == ':' && digit(3) && digit(4) && s[5] == ':' && digit(6) && digit(7)) return true;
    return false;
  }
  static bool ParseNumber(const std::string& tok, Value* v) {
    std::string s;
    for (size_t k = 0; k < tok.size(); ++k) {
      if (tok[k] == '_') {  // only between digits
        if (k == 0 || k + 1 >= tok.size() || !std::isxdigit((unsigned char)tok[k - 1]) || !std::isxdigit((unsigned char)tok[k + 1])) return false;
        continue;
      }
      s += tok[k];
    }
    const std::string body = (s[0] == '+' || s[0] == '-') ? s.substr(1) : s;
    if (body == "inf" || body == "nan") {
      v->kind = Value::kFloat;
      v->f = body == "inf" ? HUGE_VAL : std::nan("");
      if (s[0] == '-') v->f = -v->f;
      return true;
    }
    if (body.size() > 2 && body[0] == '0' && (body[1] == 'x' || body[1] == 'o' || body[1] == 'b')) {
      if (s[0] == '+' || s[0] == '-') return false;
      const int base = body[1] == 'x' ? 16 : (body[1] == 'o' ? 8 : 2);
      char* end = nullptr;
      v->i = (std::int64_t)std::strtoull(body.c_str() + 2, &end, base);
      if (*end != '\0') return false;
      v->kind = Value::kInteger;
      return true;
    }
    if (body.empty() || !(body[0] >= '0' && body[0] <= '9')) return false;
    const bool is_float = body.find_first_of(".eE") != std::string::npos;
    // no leading zeros in the integer part
    const size_t int_len = body.find_first_of(".eE");
    const std::string int_part = body.substr(0, int_len);
    if (int_part.size() > 1 && int_part[0] == '0') return false;
    for (char c : int_part) if (c < '0' || c > '9') return false;
    char* end = nullptr;
    if (is_float) {
      const size_t dot = body.find('.');
      if (dot != std::string::npos && (dot + 1 >= body.size() || !(body[dot + 1] >= '0' && body[dot + 1] <= '9'))) return false;
      v->f = std::strtod(s.c_str(), &end);
      if (*end != '\0') return false;
      v->kind = Value::kFloat;
      return true;
    }
    errno = 0;
    v->i = std::strtoll(s.c_str(), &end, 10);
    if (*end != '\0' || errno == ERANGE) return false;
    v->kind = Value::kInteger;
    return true;
  }
};

inline Value ParseText(const std::string& text) { return Parser(text).Parse(); }
inline Value ParseFile(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw FileError("cannot open " + path);
  std::ostringstream ss;
  ss << f.rdbuf();
  std::string text = ss.str();
  if (text.size() >= 3 && (unsigned char)text[0] == 0xEF && (unsigned char)text[1] == 0xBB && (unsigned char)text[2] == 0xBF) text.erase(0, 3);  // BOM
  return ParseText(text);
}

}  // namespace beatrice_amd::toml_subset
