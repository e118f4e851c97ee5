// rten_hip_safetensors.hpp -- Safetensors reader / writer for golden inputs and outputs: the file format
// `rten-cli --inputs x.safetensors --check-outputs y.safetensors` exchanges (rten-cli/src/main.rs:433-457, read by
// rten-serialize).  Layout: u64 little-endian header length N, N bytes of JSON
//   {"name": {"dtype": "F32", "shape": [..], "data_offsets": [begin, end]}, ..., "__metadata__": {...}}
// then the tensor bytes.  Only what the format needs is parsed: objects, arrays of integers, strings, integers.
#pragma once

#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace rten_hip {
namespace safetensors {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

struct Entry {
    std::string dtype; // "F32", "I32", "I64", "U8", "I8", "BOOL", ...
    std::vector<int64_t> shape;
    std::string data;  // raw little-endian bytes
    int64_t len() const { int64_t n = 1; for (int64_t d : shape) n *= d; return n; }
};

inline size_t dtype_size(const std::string &d) {
    if (d == "F64" || d == "I64" || d == "U64") return 8;
    if (d == "F32" || d == "I32" || d == "U32") return 4;
    if (d == "F16" || d == "BF16" || d == "I16" || d == "U16") return 2;
    if (d == "U8" || d == "I8" || d == "BOOL") return 1;
    throw Error("safetensors: unknown dtype " + d);
}

namespace detail {
struct Json {
    const char *p, *end;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    void expect(char c) { ws(); if (p >= end || *p != c) throw Error(std::string("safetensors: malformed header, expected '") + c + "'"); p++; }
    bool peek(char c) { ws(); return p < end && *p == c; }
    std::string str() {
        expect('"');
        std::string s;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) { p++; s.push_back(*p == 'n' ? '\n' : *p == 't' ? '\t' : *p); p++; }
            else s.push_back(*p++);
        }
        expect('"');
        return s;
    }
    int64_t integer() {
        ws();
        bool neg = false;
        if (p < end && *p == '-') { neg = true; p++; }
        if (p >= end || *p < '0' || *p > '9') throw Error("safetensors: malformed header, expected an integer");
        int64_t v = 0;
        while (p < end && *p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0');
        return neg ? -v : v;
    }
    void skip_value() { // strings, numbers, nested objects / arrays (used for __metadata__)
        ws();
        if (peek('"')) { str(); return; }
        if (peek('{') || peek('[')) {
            const char open = *p, close = open == '{' ? '}' : ']';
            p++;
            while (!peek(close)) {
                if (open == '{') { str(); expect(':'); }
                skip_value();
                if (peek(',')) p++;
            }
            p++;
            return;
        }
        while (p < end && *p != ',' && *p != '}' && *p != ']') p++;
    }
};
} // namespace detail

inline std::map<std::string, Entry> read(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error("safetensors: cannot open " + path);
    std::string buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (buf.size() < 8) throw Error("safetensors: file shorter than its length prefix");
    uint64_t n;
    std::memcpy(&n, buf.data(), 8);
    if (n > buf.size() - 8) throw Error("safetensors: header length exceeds the file");
    const char *data = buf.data() + 8 + n;
    const size_t data_len = buf.size() - 8 - (size_t)n;
    detail::Json j{buf.data() + 8, buf.data() + 8 + n};
    std::map<std::string, Entry> out;
    j.expect('{');
    while (!j.peek('}')) {
        const std::string name = j.str();
        j.expect(':');
        if (name == "__metadata__") { j.skip_value(); if (j.peek(',')) j.p++; continue; }
        Entry e;
        int64_t begin = -1, stop = -1;
        j.expect('{');
        while (!j.peek('}')) {
            const std::string key = j.str();
            j.expect(':');
            if (key == "dtype") e.dtype = j.str();
            else if (key == "shape") { j.expect('['); while (!j.peek(']')) { e.shape.push_back(j.integer()); if (j.peek(',')) j.p++; } j.expect(']'); }
            else if (key == "data_offsets") { j.expect('['); begin = j.integer(); j.expect(','); stop = j.integer(); j.expect(']'); }
            else j.skip_value();
            if (j.peek(',')) j.p++;
        }
        j.expect('}');
        if (j.peek(',')) j.p++;
        if (begin < 0 || stop < begin || (size_t)stop > data_len) throw Error("safetensors: tensor " + name + " has invalid data_offsets");
        if ((size_t)(stop - begin) != (size_t)e.len() * dtype_size(e.dtype)) throw Error("safetensors: tensor " + name + " size does not match dtype x shape");
        e.data.assign(data + begin, (size_t)(stop - begin));
        out.emplace(name, std::move(e));
    }
    return out;
}

inline void write(const std::string &path, const std::vector<std::pair<std::string, Entry>> &tensors) {
    std::string header = "{";
    size_t off = 0;
    for (size_t i = 0; i < tensors.size(); i++) {
        const Entry &e = tensors[i].second;
        if (e.data.size() != (size_t)e.len() * dtype_size(e.dtype)) throw Error("safetensors: tensor " + tensors[i].first + " size does not match dtype x shape");
        std::string shape;
        for (size_t d = 0; d < e.shape.size(); d++) shape += (d ? "," : "") + std::to_string(e.shape[d]);
        std::string name;
        for (char c : tensors[i].first) { if (c == '"' || c == '\\') name.push_back('\\'); name.push_back(c); }
        header += std::string(i ? "," : "") + "\"" + name + "\":{\"dtype\":\"" + e.dtype + "\",\"shape\":[" + shape + "],\"data_offsets\":[" + std::to_string(off) + "," +
                  std::to_string(off + e.data.size()) + "]}";
        off += e.data.size();
    }
    header += "}";
    while (header.size() % 8) header.push_back(' '); // the reference implementation aligns the data section
    const uint64_t n = header.size();
    std::ofstream f(path, std::ios::binary);
    if (!f) throw Error("safetensors: cannot create " + path);
    f.write((const char *)&n, 8);
    f.write(header.data(), (std::streamsize)header.size());
    for (auto &t : tensors) f.write(t.second.data.data(), (std::streamsize)t.second.data.size());
}

} // namespace safetensors
} // namespace rten_hip
