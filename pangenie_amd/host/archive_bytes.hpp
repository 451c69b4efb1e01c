// archive_bytes.hpp — byte-level reader / writer of cereal's binary archives (little endian, no framing), shared by
// cereal_io.cpp and graph_io.cpp.  Internal.
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace pangenie {
namespace archive_bytes {

constexpr uint32_t MSB = 0x80000000u;  // shared-pointer / polymorphic ids: "first occurrence, data follows"

struct Reader {
    const unsigned char* p;
    size_t n, o = 0;
    template <class T>
    T take() {
        if (sizeof(T) > n - o) throw std::runtime_error("archive: truncated");  // (o <= n always)
        T v;
        std::memcpy(&v, p + o, sizeof(T));
        o += sizeof(T);
        return v;
    }
    std::string str() {
        const uint64_t len = take<uint64_t>();
        if (len > n - o) throw std::runtime_error("archive: truncated string");  // (no wrap for lengths near 2^64)
        std::string s((const char*)p + o, (size_t)len);
        o += (size_t)len;
        return s;
    }
    /** an element count read from the archive: at most what the remaining bytes can hold at `min_bytes` each */
    uint64_t count(size_t min_bytes) {
        const uint64_t c = take<uint64_t>();
        if (c > (n - o) / (min_bytes ? min_bytes : 1)) throw std::runtime_error("archive: element count exceeds the data");
        return c;
    }
};

struct Writer {
    std::vector<unsigned char> out;
    template <class T>
    void put(T v) {
        const unsigned char* q = (const unsigned char*)&v;
        out.insert(out.end(), q, q + sizeof(T));
    }
    void str(const std::string& s) {
        put<uint64_t>(s.size());
        out.insert(out.end(), s.begin(), s.end());
    }
};


}  // namespace archive_bytes
}  // namespace pangenie
