// text_util.h — the character classes and the number syntax the reference's text readers use (host code only).
//
//   CharacterSet.whitespaces            = Unicode category Zs + CHARACTER TABULATION         (trimming, RTTMParser.swift:31; ARPA / RTTM fields)
//   CharacterSet.newlines               = U+000A ... U+000D, U+0085, U+2028, U+2029           (components(separatedBy: .newlines), RTTMParser.swift:30)
//   CharacterSet.whitespacesAndNewlines = both                                                 (ARPALanguageModel.swift:131)
//   Character.isWhitespace              = the Unicode White_Space property = the same union   (RTTMParser.swift:36)
//   Float(String)                       = the WHOLE string is one number: decimal or hexadecimal, inf / infinity / nan, optional sign;
//                                         no surrounding whitespace, no digit separators
// Text is UTF-8; every function returns the length in bytes of the class member that starts at p (0 = none).
#pragma once
#include <cstdlib>
#include <string>

namespace fa_text {

inline int ws_len(const char *p, const char *end) {          // CharacterSet.whitespaces
    if (p >= end) return 0;
    const unsigned char c = static_cast<unsigned char>(p[0]);
    if (c == ' ' || c == '\t') return 1;
    if (c == 0xC2 && end - p >= 2 && static_cast<unsigned char>(p[1]) == 0xA0) return 2;                       // U+00A0
    if (end - p >= 3) {
        const unsigned char d = static_cast<unsigned char>(p[1]), e = static_cast<unsigned char>(p[2]);
        if (c == 0xE1 && d == 0x9A && e == 0x80) return 3;                                                     // U+1680
        if (c == 0xE2 && d == 0x80 && ((e >= 0x80 && e <= 0x8A) || e == 0xAF)) return 3;                       // U+2000 ... U+200A, U+202F
        if (c == 0xE2 && d == 0x81 && e == 0x9F) return 3;                                                     // U+205F
        if (c == 0xE3 && d == 0x80 && e == 0x80) return 3;                                                     // U+3000
    }
    return 0;
}

inline int nl_len(const char *p, const char *end) {          // CharacterSet.newlines
    if (p >= end) return 0;
    const unsigned char c = static_cast<unsigned char>(p[0]);
    if (c >= 0x0A && c <= 0x0D) return 1;
    if (c == 0xC2 && end - p >= 2 && static_cast<unsigned char>(p[1]) == 0x85) return 2;                       // U+0085
    if (c == 0xE2 && end - p >= 3 && static_cast<unsigned char>(p[1]) == 0x80 &&
        (static_cast<unsigned char>(p[2]) == 0xA8 || static_cast<unsigned char>(p[2]) == 0xA9)) return 3;      // U+2028, U+2029
    return 0;
}

inline int ws_or_nl_len(const char *p, const char *end) { const int w = ws_len(p, end); return w ? w : nl_len(p, end); }

// [a, b) without leading / trailing members of the class
template <class Len>
inline void trim(const char *&a, const char *&b, Len len) {
    for (int k; a < b && (k = len(a, b)) > 0;) a += k;
    for (;;) {
        // the last character starts 1 ... 3 bytes before b
        int cut = 0;
        for (int back = 1; back <= 3 && back <= b - a; ++back) { const int k = len(b - back, b); if (k == back) { cut = back; break; } }
        if (!cut) break;
        b -= cut;
    }
}

// Float(String)
inline bool parse_float(const std::string &s, float &out) {
    if (s.empty()) return false;
    for (const char ch : s) {                                   // strtof would skip leading white space and accepts nothing non-ASCII anyway
        const unsigned char c = static_cast<unsigned char>(ch);
        if (c <= ' ' || c >= 0x7F || c == '_') return false;
    }
    char *end = nullptr;
    const float v = strtof(s.c_str(), &end);
    if (end != s.c_str() + s.size()) return false;
    out = v;
    return true;
}

}  // namespace fa_text
