// Token ids -> LaTeX string and the reference's whitespace clean-up (include/d2t_prep.h).  Host only.
//
// The reference does this with Python `re` substitutions in a fixed-point loop per formula
// (doc2tex/utils/data_utils.py:433-455).  Here every substitution is one linear scan over code points with the
// character classes of Python's re (unicode_tables.h), so a formula costs a few passes over a few hundred code points.
//
// The three loop patterns, with N = [\W_^\d] ("noletter": anything that is not a Unicode letter-like word character;
// whitespace IS in N), L = [a-zA-Z], s = \s:
//   1. (?!\\ )(N)s+?(N) -> \1\2     lazy s+? followed by N: since whitespace is in N this is always exactly ONE s
//   2. (?!\\ )(N)s+?(L) -> \1\2     a whole run of whitespace between N and a letter
//   3. (L)s+?(N)        -> \1\2     exactly one s
// re.sub scans left to right over non-overlapping matches: the character matched as group 2 is consumed and cannot start
// the next match, which is why the reference iterates to a fixed point -- and so does this code, pass for pass.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/d2t.h"
#include "../../include/d2t_prep.h"
#include "unicode_tables.h"

namespace {

typedef std::vector<uint32_t> U32;

bool in_ranges(const CpRange* r, int n, uint32_t cp) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    if (cp < r[mid].lo) hi = mid - 1;
    else if (cp > r[mid].hi) lo = mid + 1;
    else return true;
  }
  return false;
}

enum : uint8_t { C_LETTER = 1, C_NOLETTER = 2, C_SPACE = 4 };

struct AsciiClasses {  // built once, thread-safely (function-local static of a constructed object): d2t_post_* may be called
  uint8_t v[128];      // from several serving threads at once
  AsciiClasses() {
    for (uint32_t c = 0; c < 128; ++c) {
      const bool alpha = (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'), digit = c >= '0' && c <= '9';
      const bool space = c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f);
      v[c] = (alpha ? C_LETTER : 0) | ((!alpha && !digit && c != '_') || digit || c == '_' || c == '^' ? C_NOLETTER : 0) |
             (space ? C_SPACE : 0);
    }
  }
};

uint8_t classify(uint32_t cp) {
  static const AsciiClasses table;
  const uint8_t* ascii = table.v;
  if (cp < 128) return ascii[cp];
  const bool word = in_ranges(kAlnumRanges, kAlnumRanges_n, cp);  // \w (the underscore is ASCII)
  const bool dec = in_ranges(kDecimalRanges, kDecimalRanges_n, cp);
  return ((!word || dec) ? C_NOLETTER : 0) | (in_ranges(kSpaceRanges, kSpaceRanges_n, cp) ? C_SPACE : 0);
}

bool decode_utf8(const char* s, size_t n, U32& out) {
  out.clear();
  out.reserve(n);
  for (size_t i = 0; i < n;) {
    const unsigned char c = (unsigned char)s[i];
    uint32_t cp;
    int len;
    if (c < 0x80) cp = c, len = 1;
    else if ((c & 0xE0) == 0xC0) cp = c & 0x1F, len = 2;
    else if ((c & 0xF0) == 0xE0) cp = c & 0x0F, len = 3;
    else if ((c & 0xF8) == 0xF0) cp = c & 0x07, len = 4;
    else return false;
    if (i + len > n) return false;
    for (int k = 1; k < len; ++k) {
      const unsigned char d = (unsigned char)s[i + k];
      if ((d & 0xC0) != 0x80) return false;
      cp = (cp << 6) | (d & 0x3F);
    }
    out.push_back(cp);
    i += len;
  }
  return true;
}

void encode_utf8(const U32& in, std::string& out) {
  out.clear();
  for (uint32_t cp : in) {
    if (cp < 0x80) out.push_back((char)cp);
    else if (cp < 0x800) out.push_back((char)(0xC0 | (cp >> 6))), out.push_back((char)(0x80 | (cp & 0x3F)));
    else if (cp < 0x10000)
      out.push_back((char)(0xE0 | (cp >> 12))), out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))),
          out.push_back((char)(0x80 | (cp & 0x3F)));
    else
      out.push_back((char)(0xF0 | (cp >> 18))), out.push_back((char)(0x80 | ((cp >> 12) & 0x3F))),
          out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))), out.push_back((char)(0x80 | (cp & 0x3F)));
  }
}

bool starts_with(const U32& s, size_t i, const char* lit) {
  for (size_t k = 0; lit[k]; ++k)
    if (i + k >= s.size() || s[i + k] != (unsigned char)lit[k]) return false;
  return true;
}

// `\s?\*? {` at position p; returns the index just after '{' or 0
size_t match_brace_open(const U32& s, size_t p) {
  const size_t n = s.size();
  auto rest = [&](size_t q) -> size_t {  // \*? {   (greedy star first; without it s[q] would have to be the space)
    if (q + 2 < n && s[q] == '*' && s[q + 1] == ' ' && s[q + 2] == '{') return q + 3;
    if (q + 1 < n && s[q] == ' ' && s[q + 1] == '{') return q + 2;
    return 0;
  };
  if (p < s.size() && (classify(s[p]) & C_SPACE)) {  // greedy \s? first, then without it
    if (size_t e = rest(p + 1)) return e;
  }
  return rest(p);
}

// first '}' at or after p with no '\n' before it (`.*?}`); returns the index just after '}' or 0
size_t match_lazy_close(const U32& s, size_t p) {
  for (size_t i = p; i < s.size(); ++i) {
    if (s[i] == '}') return i + 1;
    if (s[i] == '\n') return 0;
  }
  return 0;
}

// (\\(operatorname|mathrm|...)\s?\*? {.*?}) -> the match without U+0020           data_utils.py:443-447
void strip_font_commands(const U32& s, U32& out) {
  static const char* names[] = {"operatorname", "mathrm", "mathbf", "mathsf", "mathit", "mathfrak", "mathnormal"};
  out.clear();
  out.reserve(s.size());
  for (size_t i = 0; i < s.size();) {
    size_t end = 0;
    if (s[i] == '\\') {
      for (const char* nm : names) {
        if (!starts_with(s, i + 1, nm)) continue;
        const size_t open = match_brace_open(s, i + 1 + strlen(nm));
        if (open) end = match_lazy_close(s, open);
        break;  // no name is a prefix of another
      }
    }
    if (end) {
      for (size_t k = i; k < end; ++k)
        if (s[k] != ' ') out.push_back(s[k]);
      i = end;
    } else {
      out.push_back(s[i++]);
    }
  }
}

// one re.sub of (lookahead?)(A)\s+?(B) -> \1\2 ; returns whether anything changed
bool sub_pass(const U32& s, U32& out, uint8_t classA, uint8_t classB, bool lookahead) {
  out.clear();
  out.reserve(s.size());
  bool changed = false;
  const size_t n = s.size();
  for (size_t i = 0; i < n;) {
    bool hit = false;
    if (i + 2 < n && (classify(s[i]) & classA) && (classify(s[i + 1]) & C_SPACE) &&
        !(lookahead && s[i] == '\\' && s[i + 1] == ' ')) {
      size_t j = i + 1;  // lazy \s+?: the shortest run of whitespace that is followed by a B
      while (j < n && (classify(s[j]) & C_SPACE)) {
        ++j;
        if (j < n && (classify(s[j]) & classB)) {
          hit = true;
          break;
        }
      }
      if (hit) {
        out.push_back(s[i]);
        out.push_back(s[j]);
        i = j + 1;
        changed = true;
      }
    }
    if (!hit) out.push_back(s[i++]);
  }
  return changed;
}

// `hspace {(.*?)}`: remove U+0020 inside the braces                                   recog_flow.py:92-103
void strip_space_args(const U32& s, const char* lit, U32& out) {
  out.clear();
  const size_t ln = strlen(lit);
  for (size_t i = 0; i < s.size();) {
    size_t end = 0;
    if (starts_with(s, i, lit)) end = match_lazy_close(s, i + ln);
    if (end) {
      for (size_t k = i; k < i + ln; ++k) out.push_back(s[k]);
      for (size_t k = i + ln; k < end - 1; ++k)
        if (s[k] != ' ') out.push_back(s[k]);
      // the reference resumes copying at m.end(1), i.e. the '}' itself is copied with the following text; the NEXT search
      // however starts after the '}' (finditer on the unmodified string)
      out.push_back('}');
      i = end;
    } else {
      out.push_back(s[i++]);
    }
  }
}

void whitespace_pass(U32& s, int mode) {
  if (mode == D2T_POST_NONE) return;
  U32 a, b;
  strip_font_commands(s, a);
  if (mode == D2T_POST_DEMO) {
    strip_space_args(a, "hspace {", b);
    strip_space_args(b, "vspace {", s);
    return;
  }
  for (;;) {  // data_utils.py:448-455
    bool ch = sub_pass(a, b, C_NOLETTER, C_NOLETTER, true);
    ch |= sub_pass(b, a, C_NOLETTER, C_LETTER, true);
    ch |= sub_pass(a, b, C_LETTER, C_NOLETTER, false);
    a.swap(b);
    if (!ch) break;
  }
  s.swap(a);
}

}  // namespace

struct d2t_vocab {
  std::vector<std::string> tok;
};

extern "C" {

int d2t_vocab_create(const char* const* tokens, int n_tokens, d2t_vocab** out) {
  if (!tokens || n_tokens <= 0 || !out) return D2T_EINVAL;
  d2t_vocab* v = new d2t_vocab();
  v->tok.reserve(n_tokens);
  for (int i = 0; i < n_tokens; ++i) {
    if (!tokens[i]) {
      delete v;
      return D2T_EINVAL;
    }
    v->tok.emplace_back(tokens[i]);
  }
  *out = v;
  return D2T_OK;
}

void d2t_vocab_destroy(d2t_vocab* v) { delete v; }

int d2t_post_strip_whitespace(const char* s, int mode, char* out, int64_t out_cap) {
  if (!s || !out || mode < D2T_POST_NONE || mode > D2T_POST_DEMO) return D2T_EINVAL;
  const size_t n = strlen(s);
  if ((int64_t)n + 1 > out_cap) return D2T_ENOMEM;
  U32 cps;
  if (!decode_utf8(s, n, cps)) return D2T_EINVAL;
  whitespace_pass(cps, mode);
  std::string r;
  encode_utf8(cps, r);
  memcpy(out, r.c_str(), r.size() + 1);
  return D2T_OK;
}

int d2t_post_decode(const d2t_vocab* v, const int64_t* ids, int rows, int cols, const char* sep, int cut_at_end, int mode,
                    char* out, int64_t out_cap, int64_t* out_offsets, int64_t* needed) {
  if (!v || !ids || rows < 0 || cols < 0 || !sep || !out_offsets || mode < D2T_POST_NONE || mode > D2T_POST_DEMO ||
      (!out && out_cap > 0))
    return D2T_EINVAL;
  const int64_t V = (int64_t)v->tok.size();
  std::string text, enc;
  U32 cps;
  int64_t pos = 0;
  bool fits = true;
  for (int r = 0; r < rows; ++r) {
    text.clear();
    for (int c = 0; c < cols; ++c) {
      int64_t id = ids[(size_t)r * cols + c];
      if (id < 0) id += V;  // Python list indexing
      if (id < 0 || id >= V) return D2T_EINVAL;
      if (c) text += sep;
      text += v->tok[id];
    }
    if (cut_at_end) {  // pred[: pred.find("[s]")]
      const size_t f = text.find("[s]");
      if (f != std::string::npos) {
        text.resize(f);
      } else if (!text.empty()) {  // find() == -1: drop the last CHARACTER (code point)
        size_t e = text.size() - 1;
        while (e > 0 && ((unsigned char)text[e] & 0xC0) == 0x80) --e;
        text.resize(e);
      }
    }
    if (mode != D2T_POST_NONE) {
      if (!decode_utf8(text.data(), text.size(), cps)) return D2T_EINVAL;
      whitespace_pass(cps, mode);
      encode_utf8(cps, enc);
    } else {
      enc = text;
    }
    out_offsets[r] = pos;
    if (pos + (int64_t)enc.size() + 1 <= out_cap) memcpy(out + pos, enc.c_str(), enc.size() + 1);
    else fits = false;
    pos += (int64_t)enc.size() + 1;
  }
  if (needed) *needed = pos;
  return fits ? D2T_OK : D2T_ENOMEM;
}

}  // extern "C"
