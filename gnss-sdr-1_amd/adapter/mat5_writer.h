/*!
 * \file mat5_writer.h
 * \brief Minimal writer of MATLAB Level-5 MAT-files (uncompressed), enough for the variables the
 * reference's acquisition dump (pcps_acquisition.cc:462-562) and tracking dump (dll_pll_veml_tracking.cc:1253-1438) hold.
 *
 * The reference writes its dump through matio as a v7.3 (HDF5) file; neither matio nor HDF5 is a
 * dependency here, so the same variables (names, classes, dimensions, column-major data) go into a
 * Level-5 file, which matio's Mat_Open (used by the reference's acquisition_dump_reader.cc),
 * MATLAB/Octave `load` (src/utils/matlab/plot_acq_grid.m) and scipy.io.loadmat all read.
 */
#ifndef GNSSCORR_MAT5_WRITER_H_
#define GNSSCORR_MAT5_WRITER_H_

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace gnsscorr
{
class Mat5Writer
{
public:
    // array classes (mx*_CLASS) and data types (mi*) of the MAT-file format
    enum : uint32_t
    {
        mxDOUBLE = 6,
        mxSINGLE = 7,
        mxINT32 = 12,
        mxUINT32 = 13,
        mxUINT64 = 15
    };
    enum : uint32_t
    {
        miINT8 = 1,
        miINT32 = 5,
        miUINT32 = 6,
        miSINGLE = 7,
        miDOUBLE = 9,
        miUINT64 = 13,
        miMATRIX = 14
    };

    Mat5Writer() : d_fp(nullptr) {}
    ~Mat5Writer() { close(); }
    Mat5Writer(const Mat5Writer&) = delete;
    Mat5Writer& operator=(const Mat5Writer&) = delete;

    bool open(const std::string& filename)
    {
        close();
        d_fp = std::fopen(filename.c_str(), "wb");
        if (d_fp == nullptr) return false;
        char hdr[128];
        std::memset(hdr, ' ', 116);
        const char* text = "MATLAB 5.0 MAT-file, Platform: gnsscorr, dump";
        std::memcpy(hdr, text, std::strlen(text));
        std::memset(hdr + 116, 0, 8);  // subsystem data offset: none
        const uint16_t version = 0x0100;
        const uint16_t endian = 0x4D49;  // "MI": a reader that sees "IM" must swap
        std::memcpy(hdr + 124, &version, 2);
        std::memcpy(hdr + 126, &endian, 2);
        return std::fwrite(hdr, 1, 128, d_fp) == 128;
    }

    bool is_open() const { return d_fp != nullptr; }

    //! real numeric array of `rows x cols` elements, column-major
    bool write_array(const char* name, uint32_t mx_class, uint32_t mi_type, size_t elem_size, size_t rows, size_t cols, const void* data)
    {
        if (d_fp == nullptr) return false;
        const size_t name_len = std::strlen(name);
        const size_t data_bytes = rows * cols * elem_size;
        const size_t body = 16 /* flags */ + 16 /* dims */ + 8 + pad8(name_len) + 8 + pad8(data_bytes);
        if (body > 0xFFFFFFFFull) return false;  // Level-5 elements are limited to 4 GiB
        bool ok = tag(miMATRIX, static_cast<uint32_t>(body));
        // array flags
        const uint32_t flags[2] = {mx_class, 0U};
        ok = ok && tag(miUINT32, 8) && put(flags, 8);
        // dimensions
        const int32_t dims[2] = {static_cast<int32_t>(rows), static_cast<int32_t>(cols)};
        ok = ok && tag(miINT32, 8) && put(dims, 8);
        // name
        ok = ok && tag(miINT8, static_cast<uint32_t>(name_len)) && put(name, name_len) && padding(name_len);
        // real part
        ok = ok && tag(mi_type, static_cast<uint32_t>(data_bytes)) && put(data, data_bytes) && padding(data_bytes);
        return ok;
    }

    bool write_single_matrix(const char* name, size_t rows, size_t cols, const float* data) { return write_array(name, mxSINGLE, miSINGLE, 4, rows, cols, data); }
    bool write_scalar(const char* name, float v) { return write_array(name, mxSINGLE, miSINGLE, 4, 1, 1, &v); }
    bool write_scalar(const char* name, uint32_t v) { return write_array(name, mxUINT32, miUINT32, 4, 1, 1, &v); }
    bool write_scalar(const char* name, int32_t v) { return write_array(name, mxINT32, miINT32, 4, 1, 1, &v); }
    bool write_scalar(const char* name, uint64_t v) { return write_array(name, mxUINT64, miUINT64, 8, 1, 1, &v); }

    void close()
    {
        if (d_fp != nullptr) std::fclose(d_fp);
        d_fp = nullptr;
    }

private:
    static size_t pad8(size_t n) { return (n + 7) & ~static_cast<size_t>(7); }
    bool put(const void* p, size_t n) { return n == 0 || std::fwrite(p, 1, n, d_fp) == n; }
    bool tag(uint32_t type, uint32_t nbytes)
    {
        const uint32_t t[2] = {type, nbytes};
        return put(t, 8);
    }
    bool padding(size_t n)
    {
        static const char zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        return put(zeros, pad8(n) - n);
    }
    std::FILE* d_fp;
};
}  // namespace gnsscorr

#endif /* GNSSCORR_MAT5_WRITER_H_ */
