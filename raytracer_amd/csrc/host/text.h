// String helpers with the semantics of the Rust `str` methods the reference's
// parsers are written with (`lines`, `split`, `split_whitespace`, `trim`,
// `split_once`, `str::parse`), so that the parsers in this directory accept and
// reject the same inputs.
#pragma once
#include <string>
#include <vector>

namespace rth {

// `str::lines`: split on '\n', drop one trailing '\r' per line, and do not
// yield a final empty line after a terminating newline.
std::vector<std::string> lines(const std::string& s);

// `str::split(&str)`: every occurrence of `sep` separates; empty pieces kept.
std::vector<std::string> split(const std::string& s, const std::string& sep);

// `str::split_whitespace`: runs of white space separate; no empty pieces.
std::vector<std::string> split_whitespace(const std::string& s);

// `str::trim`.
std::string trim(const std::string& s);

inline bool starts_with(const std::string& s, const char* prefix) {
    return s.rfind(prefix, 0) == 0;
}

std::string read_text_file(const std::string& path);

}  // namespace rth
