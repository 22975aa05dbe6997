// gc_codes.cpp -- host-side PRN replica generators of libgnsscorr.so (set-up path:
// the tables produced here are what gc_trk_batch_set_code / gc_acq_set_local_code upload).
//
// Same outputs as the reference generators
//   gps_l1_ca_code_gen_float / _complex_sampled   (src/algorithms/libs/gps_sdr_signal_processing.cc:119-196)
//   beidou_b1i_code_gen_float / _complex_sampled   (src/algorithms/libs/beidou_b1i_signal_processing.cc:115-191)
//   glonass_l1_ca_code_gen_complex / _complex_sampled (src/algorithms/libs/glonass_l1_signal_processing.cc:37-153; L2 C/A is the same code)
//   galileo_e1_code_gen_sinboc11_float / _complex_sampled (src/algorithms/libs/galileo_e1_signal_processing.cc:108-255)
//   resampler()                                    (src/algorithms/libs/gnss_signal_processing.cc:161-182)
// written from the signal ICDs (IS-GPS-200 G1/G2 registers and G2 delays, BDS-SIS-ICD-B1I
// G1/G2 registers and phase selectors, Galileo OS SIS ICD memory codes) as word-wide LFSRs.
// The sampling functions keep the reference's float32 index arithmetic (ts*(i+1))/tc and
// its (int32)(int64)(x+1) "ceil", because acquisition code phase depends on it.
#include "gc_internal.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

namespace
{
// ---- GPS L1 C/A (IS-GPS-200, 3.3.2.3): G1 = 1+x^3+x^10, G2 = 1+x^2+x^3+x^6+x^8+x^9+x^10 ----
const int kGpsG2Delay[51] = {5, 6, 7, 8, 17, 18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258, 469, 470, 471, 472,
    473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862,  // PRN 1..32
    145, 175, 52, 21, 237, 235, 886, 657, 634, 762, 355, 1012, 176, 603, 130, 359, 595, 68, 386};  // PRN 120..138

bool gps_ca_chips(int8_t* out, int prn, unsigned chip_shift)
{
    int idx = (prn >= 120 && prn <= 138) ? prn - 88 : prn - 1;
    if (idx < 0 || idx > 50) return false;
    uint8_t g1[1023], g2[1023];
    unsigned r1 = 0x3ff, r2 = 0x3ff;  // bit k = stage k+1; output = stage 10
    for (int i = 0; i < 1023; i++)
        {
            g1[i] = (r1 >> 9) & 1;
            g2[i] = (r2 >> 9) & 1;
            unsigned f1 = ((r1 >> 2) ^ (r1 >> 9)) & 1;
            unsigned f2 = ((r2 >> 1) ^ (r2 >> 2) ^ (r2 >> 5) ^ (r2 >> 7) ^ (r2 >> 8) ^ (r2 >> 9)) & 1;
            r1 = ((r1 << 1) | f1) & 0x3ff;
            r2 = ((r2 << 1) | f2) & 0x3ff;
        }
    // chip i = G1[i] xor G2[i - delay]; chip_shift advances the whole code
    const int delay = kGpsG2Delay[idx];
    for (int i = 0; i < 1023; i++)
        {
            int k = (int)((i + chip_shift) % 1023);
            int j = ((k - delay) % 1023 + 1023) % 1023;
            out[i] = (g1[k] ^ g2[j]) ? 1 : -1;
        }
    return true;
}

// ---- BeiDou B1I (BDS-SIS-ICD-B1I): G1 = 1+x+x^7+x^8+x^9+x^10+x^11, G2 = 1+x+x^2+x^3+x^4+x^5+x^8+x^9+x^11,
// both initialised to 01010101010; G2 output = xor of two phase-selector stages ----
const int kBdsSel1[37] = {1, 1, 1, 1, 1, 1, 1, 1, 2, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 6, 6, 6, 6, 8, 8, 8, 9, 9, 10};
const int kBdsSel2[37] = {3, 4, 5, 6, 8, 9, 10, 11, 7, 4, 5, 6, 8, 9, 10, 11, 5, 6, 8, 9, 10, 11, 6, 8, 9, 10, 11, 8, 9, 10, 11, 9, 10, 11, 10, 11, 11};

bool bds_b1i_chips(int8_t* out, int prn, unsigned chip_shift)
{
    // the reference accepts PRN 1..33 (beidou_b1i_signal_processing.cc:57-62)
    if (prn < 1 || prn > 33) return false;
    // stage s (1..11) is bit (s-1); initial state 0,1,0,1,... from stage 1
    unsigned r1 = 0, r2 = 0;
    for (int s = 1; s <= 11; s++)
        if (s % 2 == 0)
            {
                r1 |= 1u << (s - 1);
                r2 |= 1u << (s - 1);
            }
    uint8_t g1[2046], g2[2046];
    const int a = kBdsSel1[prn - 1], b = kBdsSel2[prn - 1];
    for (int i = 0; i < 2046; i++)
        {
            g1[i] = (r1 >> 10) & 1;
            g2[i] = ((r2 >> (a - 1)) ^ (r2 >> (b - 1))) & 1;
            unsigned f1 = ((r1 >> 0) ^ (r1 >> 6) ^ (r1 >> 7) ^ (r1 >> 8) ^ (r1 >> 9) ^ (r1 >> 10)) & 1;
            unsigned f2 = ((r2 >> 0) ^ (r2 >> 1) ^ (r2 >> 2) ^ (r2 >> 3) ^ (r2 >> 4) ^ (r2 >> 7) ^ (r2 >> 8) ^ (r2 >> 10)) & 1;
            r1 = ((r1 << 1) | f1) & 0x7ff;
            r2 = ((r2 << 1) | f2) & 0x7ff;
        }
    for (int i = 0; i < 2046; i++)
        {
            int k = (int)((i + chip_shift) % 2046);
            out[i] = (g1[k] ^ g2[k]) ? 1 : -1;
        }
    return true;
}

inline int32_t aux_ceil(float x) { return static_cast<int32_t>(static_cast<int64_t>(x + 1)); }

// ---- GLONASS L1 / L2 C/A (GLONASS ICD 5.1, 3.3.2.2): one 511-chip m-sequence for every satellite, generator
// 1 + x^5 + x^9, all stages set at start, output from the 7th stage (glonass_l1_signal_processing.cc:37-97) ----
void glonass_ca_chips(int8_t* out, unsigned chip_shift)
{
    uint8_t g[511];
    unsigned r = 0x1ff;  // bit k = stage 9-k: bit 0 leaves first
    for (int i = 0; i < 511; i++)
        {
            g[i] = (r >> 2) & 1;                      // 7th stage
            const unsigned fb = ((r >> 4) ^ r) & 1;   // stages 5 and 9
            r = (r >> 1) | (fb << 8);
        }
    for (int i = 0; i < 511; i++) out[i] = g[(i + chip_shift) % 511] ? 1 : -1;
}

// chips -> samples with the reference's float32 digitising rule (gps_sdr_signal_processing.cc:163-190)
int sample_chips(float* dest_complex, const int8_t* chips, int code_len, int code_freq, int fs)
{
    const int spc = static_cast<int>(static_cast<double>(fs) / static_cast<double>(code_freq / code_len));
    const float ts = 1.0 / static_cast<float>(fs);
    const float tc = 1.0 / static_cast<float>(code_freq);
    for (int i = 0; i < spc; i++)
        {
            float aux = (ts * (i + 1)) / tc;
            int k = aux_ceil(aux) - 1;
            int v = (i == spc - 1) ? chips[code_len - 1] : chips[k];
            dest_complex[2 * i] = static_cast<float>(v);
            dest_complex[2 * i + 1] = 0.0f;
        }
    return spc;
}

// nearest-neighbour resampler (gnss_signal_processing.cc:161-182)
void resample(const float* from, float* dest, float fs_in, float fs_out, unsigned length_in, unsigned length_out)
{
    const float t_in = 1 / fs_in;
    const float t_out = 1 / fs_out;
    for (unsigned i = 0; i + 1 < length_out; i++)
        {
            float aux = (t_out * (i + 1)) / t_in;
            dest[i] = from[aux_ceil(aux) - 1];
        }
    dest[length_out - 1] = from[length_in - 1];
}

// ---- Galileo E1 memory codes: data file next to the library ----
std::mutex g_gal_mtx;
std::vector<uint8_t> g_gal_bits;  // [2 (B,C)][50][512] bytes, MSB first, bit 1 -> chip -1

std::string default_galileo_path()
{
    const char* env = std::getenv("GNSSCORR_GALILEO_E1_CODES");
    if (env && *env) return env;
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&default_galileo_path), &info) && info.dli_fname)
        {
            std::string p(info.dli_fname);
            size_t s = p.find_last_of('/');
            return (s == std::string::npos ? std::string(".") : p.substr(0, s)) + "/data/galileo_e1_primary_codes.bin";
        }
    return "data/galileo_e1_primary_codes.bin";
}

bool load_galileo()
{
    std::lock_guard<std::mutex> lk(g_gal_mtx);
    if (!g_gal_bits.empty()) return true;
    std::string path = default_galileo_path();
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f)
        {
            gc_set_error("Galileo E1 memory codes not found at %s", path.c_str());
            return false;
        }
    std::vector<uint8_t> buf(2 * 50 * 512);
    size_t n = std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    if (n != buf.size())
        {
            gc_set_error("Galileo E1 memory code file %s is truncated", path.c_str());
            return false;
        }
    g_gal_bits.swap(buf);
    return true;
}

// signal "1B" / "1C" (last two characters decide, like the reference's rfind)
bool galileo_primary(int8_t* out, const char* signal, unsigned prn)
{
    if (prn < 1 || prn > 50 || !signal) return false;
    std::string s(signal);
    int comp;
    if (s.size() >= 2 && s.rfind("1B") != std::string::npos)
        comp = 0;
    else if (s.size() >= 2 && s.rfind("1C") != std::string::npos)
        comp = 1;
    else
        return false;
    if (!load_galileo()) return false;
    const uint8_t* p = g_gal_bits.data() + ((size_t)comp * 50 + (prn - 1)) * 512;
    for (int i = 0; i < 4092; i++) out[i] = ((p[i >> 3] >> (7 - (i & 7))) & 1) ? -1 : 1;
    return true;
}
}  // namespace

extern "C" {

gc_status gc_gps_l1_ca_code_gen_float(float* dest, int32_t prn, uint32_t chip_shift)
{
    GC_REQUIRE(dest, "gc_gps_l1_ca_code_gen_float: dest is NULL");
    int8_t c[1023];
    GC_REQUIRE(gps_ca_chips(c, prn, chip_shift), "gc_gps_l1_ca_code_gen_float: PRN %d not in 1..32 / 120..138", prn);
    for (int i = 0; i < 1023; i++) dest[i] = static_cast<float>(c[i]);
    return GC_OK;
}

gc_status gc_gps_l1_ca_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs, uint32_t chip_shift, int32_t* n_samples)
{
    GC_REQUIRE(dest && fs > 0, "gc_gps_l1_ca_code_gen_complex_sampled: bad argument");
    int8_t c[1023];
    GC_REQUIRE(gps_ca_chips(c, (int)prn, chip_shift), "gc_gps_l1_ca_code_gen_complex_sampled: PRN %u not supported", prn);
    int n = sample_chips(dest, c, 1023, 1023000, fs);
    if (n_samples) *n_samples = n;
    return GC_OK;
}

gc_status gc_glonass_l1_ca_code_gen_float(float* dest, uint32_t chip_shift)
{
    GC_REQUIRE(dest, "gc_glonass_l1_ca_code_gen_float: dest is NULL");
    int8_t c[511];
    glonass_ca_chips(c, chip_shift);
    for (int i = 0; i < 511; i++) dest[i] = static_cast<float>(c[i]);
    return GC_OK;
}

gc_status gc_glonass_l1_ca_code_gen_complex_sampled(float* dest, int32_t fs, uint32_t chip_shift, int32_t* n_samples)
{
    GC_REQUIRE(dest && fs > 0, "gc_glonass_l1_ca_code_gen_complex_sampled: bad argument");
    int8_t c[511];
    glonass_ca_chips(c, chip_shift);
    int n = sample_chips(dest, c, 511, 511000, fs);
    if (n_samples) *n_samples = n;
    return GC_OK;
}

gc_status gc_beidou_b1i_code_gen_float(float* dest, int32_t prn, uint32_t chip_shift)
{
    GC_REQUIRE(dest, "gc_beidou_b1i_code_gen_float: dest is NULL");
    int8_t c[2046];
    GC_REQUIRE(bds_b1i_chips(c, prn, chip_shift), "gc_beidou_b1i_code_gen_float: PRN %d not in 1..33", prn);
    for (int i = 0; i < 2046; i++) dest[i] = static_cast<float>(c[i]);
    return GC_OK;
}

gc_status gc_beidou_b1i_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs, uint32_t chip_shift, int32_t* n_samples)
{
    GC_REQUIRE(dest && fs > 0, "gc_beidou_b1i_code_gen_complex_sampled: bad argument");
    int8_t c[2046];
    GC_REQUIRE(bds_b1i_chips(c, (int)prn, chip_shift), "gc_beidou_b1i_code_gen_complex_sampled: PRN %u not supported", prn);
    int n = sample_chips(dest, c, 2046, 2046000, fs);
    if (n_samples) *n_samples = n;
    return GC_OK;
}

gc_status gc_galileo_e1_code_gen_sinboc11_float(float* dest, const char* signal, uint32_t prn)
{
    GC_REQUIRE(dest, "gc_galileo_e1_code_gen_sinboc11_float: dest is NULL");
    std::vector<int8_t> c(4092);
    if (!galileo_primary(c.data(), signal, prn))
        {
            if (!*gc_last_error()) gc_set_error("gc_galileo_e1_code_gen_sinboc11_float: bad signal/PRN");
            return GC_ERR_INVALID;
        }
    for (int i = 0; i < 4092; i++)
        {
            dest[2 * i] = static_cast<float>(c[i]);
            dest[2 * i + 1] = -dest[2 * i];
        }
    return GC_OK;
}

gc_status gc_galileo_e1_code_gen_complex_sampled(float* dest, const char* signal, int32_t cboc, uint32_t prn, int32_t fs,
    uint32_t chip_shift, int32_t* n_samples)
{
    GC_REQUIRE(dest && fs > 0, "gc_galileo_e1_code_gen_complex_sampled: bad argument");
    std::vector<int8_t> c(4092);
    gc_set_error("");
    if (!galileo_primary(c.data(), signal, prn))
        {
            if (!*gc_last_error()) gc_set_error("gc_galileo_e1_code_gen_complex_sampled: bad signal/PRN");
            return GC_ERR_INVALID;
        }
    const bool is_c = std::string(signal).rfind("1C") != std::string::npos;
    // galileo_e1_code_gen_float_sampled (galileo_e1_signal_processing.cc:154-229), no secondary code
    const int code_freq = 1023000;
    const unsigned CL = 4092;
    unsigned spc = static_cast<unsigned>(static_cast<double>(fs) / (static_cast<double>(code_freq) / static_cast<double>(CL)));
    const int samples_per_chip = cboc ? 12 : 2;
    const unsigned delay = ((static_cast<int>(CL) - chip_shift) % static_cast<int>(CL)) * spc / CL;
    unsigned code_len = samples_per_chip * CL;
    std::vector<float> sig(code_len);
    if (cboc)
        {
            const float alpha = std::sqrt(10.0 / 11.0);
            const float beta = std::sqrt(1.0 / 11.0);
            for (unsigned i = 0; i < CL; i++)
                for (unsigned j = 0; j < 12; j++)
                    {
                        const float s11 = static_cast<float>(j < 6 ? c[i] : -c[i]);
                        const float s61 = static_cast<float>((j % 2 == 0) ? c[i] : -c[i]);
                        sig[i * 12 + j] = is_c ? alpha * s11 - beta * s61 : alpha * s11 + beta * s61;
                    }
        }
    else
        {
            for (unsigned i = 0; i < CL; i++)
                {
                    sig[2 * i] = static_cast<float>(c[i]);
                    sig[2 * i + 1] = static_cast<float>(-c[i]);
                }
        }
    if (fs != samples_per_chip * code_freq)
        {
            std::vector<float> rs(spc);
            resample(sig.data(), rs.data(), static_cast<float>(samples_per_chip * code_freq), static_cast<float>(fs), code_len, spc);
            sig.swap(rs);
        }
    for (unsigned i = 0; i < spc; i++)
        {
            unsigned d = (i + delay) % spc;
            dest[2 * d] = sig[i];
            dest[2 * d + 1] = 0.0f;
        }
    if (n_samples) *n_samples = static_cast<int32_t>(spc);
    return GC_OK;
}

}  // extern "C"
