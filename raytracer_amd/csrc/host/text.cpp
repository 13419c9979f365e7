#include "text.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "scene.h"

namespace rth {

std::vector<std::string> lines(const std::string& s) {
    std::vector<std::string> out;
    size_t pos = 0;
    while (pos < s.size()) {
        size_t nl = s.find('\n', pos);
        size_t end = nl == std::string::npos ? s.size() : nl;
        size_t len = end - pos;
        if (nl != std::string::npos && len > 0 && s[end - 1] == '\r') --len;
        out.emplace_back(s, pos, len);
        pos = nl == std::string::npos ? s.size() : nl + 1;
    }
    return out;
}

std::vector<std::string> split(const std::string& s, const std::string& sep) {
    std::vector<std::string> out;
    size_t pos = 0;
    for (;;) {
        size_t hit = s.find(sep, pos);
        if (hit == std::string::npos) {
            out.emplace_back(s, pos);
            return out;
        }
        out.emplace_back(s, pos, hit - pos);
        pos = hit + sep.size();
    }
}

static bool is_space(char c) {
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f';
}

std::vector<std::string> split_whitespace(const std::string& s) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && is_space(s[i])) ++i;
        size_t start = i;
        while (i < s.size() && !is_space(s[i])) ++i;
        if (i > start) out.emplace_back(s, start, i - start);
    }
    return out;
}

std::string trim(const std::string& s) {
    size_t b = 0, e = s.size();
    while (b < e && is_space(s[b])) ++b;
    while (e > b && is_space(s[e - 1])) --e;
    return s.substr(b, e - b);
}

std::string read_text_file(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::string data;
    char buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, got);
    std::fclose(f);
    return data;
}

float parse_f32(const std::string& token) {
    if (token.empty() || is_space(token[0]) ||
        (token.size() > 1 && token[0] == '0' && (token[1] == 'x' || token[1] == 'X')))
        throw std::runtime_error("not a number: '" + token + "'");
    char* end = nullptr;
    const float v = std::strtof(token.c_str(), &end);  // correctly rounded, like Rust
    if (end != token.c_str() + token.size())
        throw std::runtime_error("not a number: '" + token + "'");
    return v;
}

uint32_t parse_u32(const std::string& token) {
    if (token.empty()) throw std::runtime_error("not an integer: ''");
    size_t i = token[0] == '+' ? 1 : 0;
    if (i == token.size()) throw std::runtime_error("not an integer: '" + token + "'");
    uint64_t v = 0;
    for (; i < token.size(); ++i) {
        if (token[i] < '0' || token[i] > '9')
            throw std::runtime_error("not an integer: '" + token + "'");
        v = v * 10 + static_cast<uint64_t>(token[i] - '0');
        if (v > 0xFFFFFFFFull) throw std::runtime_error("integer too large: '" + token + "'");
    }
    return static_cast<uint32_t>(v);
}

}  // namespace rth
