// gc_codes.cpp -- host-side PRN replica generators of libgnsscorr.so (set-up path:
// the tables produced here are what gc_trk_batch_set_code / gc_acq_set_local_code upload).
//
// Same outputs as the reference generators
//   gps_l1_ca_code_gen_float / _complex_sampled   (src/algorithms/libs/gps_sdr_signal_processing.cc:119-196)
//   beidou_b1i_code_gen_float / _complex_sampled   (src/algorithms/libs/beidou_b1i_signal_processing.cc:115-191)
//   glonass_l1_ca_code_gen_complex / _complex_sampled (src/algorithms/libs/glonass_l1_signal_processing.cc:37-153; L2 C/A is the same code)
//   galileo_e1_code_gen_sinboc11_float / _complex_sampled (src/algorithms/libs/galileo_e1_signal_processing.cc:108-255)
//   gps_l2c_m_code_gen_float / _complex_sampled    (src/algorithms/libs/gps_l2c_signal.cc:44-137)
//   gps_l5i / gps_l5q _code_gen_float / _complex_sampled (src/algorithms/libs/gps_l5_signal.cc:41-344)
//   beidou_b3i_code_gen_float / _complex_sampled   (src/algorithms/libs/beidou_b3i_signal_processing.cc:37-246)
//   galileo_e5_a_code_gen_complex_primary / _sampled (src/algorithms/libs/galileo_e5_signal_processing.cc:38-142)
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

// ---- per-PRN constants of the 10.23 / 0.5115 Mcps signals: data/prn_tables.bin, data/galileo_e5a_primary_codes.bin ----
struct PrnTables
{
    std::vector<int32_t> l2c_init, l5i_advance, l5q_advance, b3i_phase;
    int n_e5a_q_secondary = 0;
    std::string e5a_q_secondary;  // 50 x 100 characters
    std::string e5a_i_secondary;  // 20 characters
    std::vector<uint8_t> e5a_bits;  // [2 (I,Q)][50][1279] bytes, MSB first, bit 1 -> chip -1
    bool loaded = false;
};
std::mutex g_tab_mtx;
PrnTables g_tab;

std::string data_file(const char* name)
{
    const char* env = std::getenv("GNSSCORR_DATA_DIR");
    if (env && *env) return std::string(env) + "/" + name;
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&default_galileo_path), &info) && info.dli_fname)
        {
            std::string p(info.dli_fname);
            size_t s = p.find_last_of('/');
            return (s == std::string::npos ? std::string(".") : p.substr(0, s)) + "/data/" + name;
        }
    return std::string("data/") + name;
}

bool read_all(const std::string& path, std::vector<uint8_t>& buf)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? static_cast<size_t>(n) : 0);
    size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    return got == buf.size();
}

bool load_tables()
{
    std::lock_guard<std::mutex> lk(g_tab_mtx);
    if (g_tab.loaded) return true;
    std::vector<uint8_t> raw;
    const std::string path = data_file("prn_tables.bin");
    if (!read_all(path, raw) || raw.size() < 28 || std::memcmp(raw.data(), "GCPRNTB1", 8) != 0)
        {
            gc_set_error("PRN tables not found (or not a table file) at %s", path.c_str());
            return false;
        }
    int32_t n[5];
    std::memcpy(n, raw.data() + 8, sizeof n);
    const size_t need = 28 + 4 * (static_cast<size_t>(n[0]) + n[1] + n[2] + n[3]) + 50 * 100 + 20;
    if (n[0] != 115 || n[1] != 210 || n[2] != 210 || n[3] != 63 || n[4] < 0 || n[4] > 50 || raw.size() != need)
        {
            gc_set_error("PRN table file %s has an unexpected layout", path.c_str());
            return false;
        }
    const uint8_t* p = raw.data() + 28;
    auto take = [&](std::vector<int32_t>& v, int count) {
        v.resize(count);
        std::memcpy(v.data(), p, 4 * static_cast<size_t>(count));
        p += 4 * static_cast<size_t>(count);
    };
    PrnTables t;
    take(t.l2c_init, n[0]);
    take(t.l5i_advance, n[1]);
    take(t.l5q_advance, n[2]);
    take(t.b3i_phase, n[3]);
    t.n_e5a_q_secondary = n[4];
    t.e5a_q_secondary.assign(reinterpret_cast<const char*>(p), 50 * 100);
    t.e5a_i_secondary.assign(reinterpret_cast<const char*>(p) + 50 * 100, 20);
    const std::string path5 = data_file("galileo_e5a_primary_codes.bin");
    if (!read_all(path5, t.e5a_bits) || t.e5a_bits.size() != 2u * 50u * 1279u)
        {
            gc_set_error("Galileo E5a memory codes not found (or truncated) at %s", path5.c_str());
            return false;
        }
    t.loaded = true;
    g_tab = std::move(t);
    return true;
}

// ---- GPS L2 CM (IS-GPS-200, 3.2.1.4 / Fig. 3-12): 27-stage modular shift register, polynomial 1112225171 (octal),
// short-cycled to 10230 chips; the per-PRN initial states are Table 3-IIa ----
bool gps_l2cm_chips(int8_t* out, unsigned prn)
{
    if (prn < 1 || prn > 50 || !load_tables()) return false;
    uint32_t x = static_cast<uint32_t>(g_tab.l2c_init[prn - 1]);
    for (int n = 0; n < 10230; n++)
        {
            out[n] = (x & 1u) ? -1 : 1;  // 1 - 2 * bit
            x = (x >> 1) ^ ((x & 1u) ? 0445112474u : 0u);
        }
    return true;
}

// ---- GPS L5 (IS-GPS-705, 3.2.1.1): XA = 1+x^9+x^10+x^12+x^13 short-cycled at 8190, XB = 1+x+x^3+x^4+x^6+x^7+x^8+x^12+x^13,
// both from all ones; chip n = XA(n) xor XB(n + advance).  Like the reference, the advanced XB index wraps at the 10230-chip
// length of the generated XB run (gps_l5_signal.cc:137-147), not at XB's natural period ----
bool gps_l5_chips(int8_t* out, unsigned prn, bool q_component)
{
    if (prn < 1 || prn > 50 || !load_tables()) return false;
    static uint8_t xa[10230], xb[10230];
    static bool built = false;
    static std::mutex mtx;
    {
        std::lock_guard<std::mutex> lk(mtx);
        if (!built)
            {
                unsigned ra = 0x1fff, rb = 0x1fff;  // bit k = stage k+1; the output is stage 13
                for (int i = 0; i < 10230; i++)
                    {
                        xa[i] = (ra >> 12) & 1;
                        xb[i] = (rb >> 12) & 1;
                        if (ra == (0x1fffu & ~(1u << 11)))
                            ra = 0x1fff;  // decoded state 1111111111101: the 8190-chip short cycle
                        else
                            ra = ((ra << 1) | (((ra >> 12) ^ (ra >> 11) ^ (ra >> 9) ^ (ra >> 8)) & 1u)) & 0x1fff;
                        rb = ((rb << 1) | (((rb >> 12) ^ (rb >> 11) ^ (rb >> 7) ^ (rb >> 6) ^ (rb >> 5) ^ (rb >> 3) ^ (rb >> 2) ^ rb) & 1u)) & 0x1fff;
                    }
                built = true;
            }
    }
    const int adv = (q_component ? g_tab.l5q_advance : g_tab.l5i_advance)[prn - 1];
    for (int n = 0; n < 10230; n++) out[n] = (xa[n] ^ xb[(adv + n) % 10230]) ? -1 : 1;
    return true;
}

// ---- BeiDou B3I (BDS-SIS-ICD-B3I, 4.2): G1 = 1+x+x^3+x^4+x^13 short-cycled at 8190, G2 = 1+x+x^5+x^6+x^7+x^9+x^10+x^12+x^13,
// G1 from all ones, G2 from the per-PRN initial phase of the ICD table; chip = G1 xor G2 (1 -> +1 like the reference) ----
bool bds_b3i_chips(int8_t* out, int prn, unsigned chip_shift)
{
    if (prn < 1 || prn > 63 || !load_tables()) return false;
    // bit k = register element k of the reference's arrays (element 0 is the output, the feedback enters at element 12)
    unsigned g1 = 0x1fff;
    const unsigned row = static_cast<unsigned>(g_tab.b3i_phase[prn - 1]);
    unsigned g2 = 0;
    for (int k = 0; k < 13; k++)
        if ((row >> (12 - k)) & 1u) g2 |= 1u << k;  // the table row is loaded in reverse order
    std::vector<uint8_t> s1(10230), s2(10230);
    for (int i = 0; i < 10230; i++)
        {
            s1[i] = g1 & 1;
            s2[i] = g2 & 1;
            const unsigned f1 = (g1 ^ (g1 >> 9) ^ (g1 >> 10) ^ (g1 >> 12)) & 1u;
            const unsigned f2 = (g2 ^ (g2 >> 1) ^ (g2 >> 3) ^ (g2 >> 4) ^ (g2 >> 6) ^ (g2 >> 7) ^ (g2 >> 8) ^ (g2 >> 12)) & 1u;
            g1 = (g1 >> 1) | (f1 << 12);
            g2 = (g2 >> 1) | (f2 << 12);
            if (g1 == 0x1ffcu) g1 = 0x1fff;  // 0011111111111 (elements 0 and 1 clear): restart G1
        }
    for (int i = 0; i < 10230; i++)
        {
            const int k = static_cast<int>((i + chip_shift) % 10230u);
            out[i] = (s1[k] ^ s2[k]) ? 1 : -1;
        }
    return true;
}

// chips -> samples with the float32 rule of the L2C / L5 generators (gps_l2c_signal.cc:113-133, gps_l5_signal.cc:225-255):
// index = ceil((ts * ((float)i + 1)) / tc) - 1 with a TRUE ceil, unlike sample_chips
int sample_chips_ceil(float* dest_complex, const int8_t* chips, int code_len, double code_rate_hz, int fs)
{
    const int spc = static_cast<int>(static_cast<double>(fs) / (code_rate_hz / static_cast<double>(code_len)));
    const float ts = 1.0 / static_cast<float>(fs);
    const float tc = 1.0 / static_cast<float>(code_rate_hz);
    for (int i = 0; i < spc; i++)
        {
            const int k = static_cast<int>(std::ceil((ts * (static_cast<float>(i) + 1)) / tc) - 1);
            const int v = (i == spc - 1) ? chips[code_len - 1] : chips[k];
            dest_complex[2 * i] = static_cast<float>(v);
            dest_complex[2 * i + 1] = 0.0f;
        }
    return spc;
}

// Galileo E5a memory code of one component (0 = I, 1 = Q) as +-1
bool galileo_e5a_component(int8_t* out, unsigned prn, int comp)
{
    if (prn < 1 || prn > 50 || !load_tables()) return false;
    const uint8_t* p = g_tab.e5a_bits.data() + (static_cast<size_t>(comp) * 50 + (prn - 1)) * 1279;
    for (int i = 0; i < 10230; i++) out[i] = ((p[i >> 3] >> (7 - (i & 7))) & 1) ? -1 : 1;
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


gc_status gc_gps_l2c_m_code_gen_float(float* dest, uint32_t prn)
{
    GC_REQUIRE(dest, "gc_gps_l2c_m_code_gen_float: dest is NULL");
    std::vector<int8_t> c(10230);
    gc_set_error("");
    if (!gps_l2cm_chips(c.data(), prn))
        {
            if (!*gc_last_error()) gc_set_error("gc_gps_l2c_m_code_gen_float: PRN %u not in 1..50", prn);
            return GC_ERR_INVALID;
        }
    for (int i = 0; i < 10230; i++) dest[i] = static_cast<float>(c[i]);
    return GC_OK;
}

gc_status gc_gps_l2c_m_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs, int32_t* n_samples)
{
    GC_REQUIRE(dest && fs > 0, "gc_gps_l2c_m_code_gen_complex_sampled: bad argument");
    std::vector<int8_t> c(10230);
    gc_set_error("");
    if (!gps_l2cm_chips(c.data(), prn))
        {
            if (!*gc_last_error()) gc_set_error("gc_gps_l2c_m_code_gen_complex_sampled: PRN %u not in 1..50", prn);
            return GC_ERR_INVALID;
        }
    const int n = sample_chips_ceil(dest, c.data(), 10230, 0.5115e6, fs);
    if (n_samples) *n_samples = n;
    return GC_OK;
}

static gc_status l5_float(float* dest, uint32_t prn, bool q, const char* who)
{
    GC_REQUIRE(dest, "%s: dest is NULL", who);
    std::vector<int8_t> c(10230);
    gc_set_error("");
    if (!gps_l5_chips(c.data(), prn, q))
        {
            if (!*gc_last_error()) gc_set_error("%s: PRN %u not in 1..50", who, prn);
            return GC_ERR_INVALID;
        }
    for (int i = 0; i < 10230; i++) dest[i] = static_cast<float>(c[i]);
    return GC_OK;
}

static gc_status l5_sampled(float* dest, uint32_t prn, int32_t fs, int32_t* n_samples, bool q, const char* who)
{
    GC_REQUIRE(dest && fs > 0, "%s: bad argument", who);
    std::vector<int8_t> c(10230);
    gc_set_error("");
    if (!gps_l5_chips(c.data(), prn, q))
        {
            if (!*gc_last_error()) gc_set_error("%s: PRN %u not in 1..50", who, prn);
            return GC_ERR_INVALID;
        }
    const int n = sample_chips_ceil(dest, c.data(), 10230, 10.23e6, fs);
    if (n_samples) *n_samples = n;
    return GC_OK;
}

gc_status gc_gps_l5i_code_gen_float(float* dest, uint32_t prn) { return l5_float(dest, prn, false, "gc_gps_l5i_code_gen_float"); }
gc_status gc_gps_l5q_code_gen_float(float* dest, uint32_t prn) { return l5_float(dest, prn, true, "gc_gps_l5q_code_gen_float"); }
gc_status gc_gps_l5i_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs, int32_t* n_samples)
{
    return l5_sampled(dest, prn, fs, n_samples, false, "gc_gps_l5i_code_gen_complex_sampled");
}
gc_status gc_gps_l5q_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs, int32_t* n_samples)
{
    return l5_sampled(dest, prn, fs, n_samples, true, "gc_gps_l5q_code_gen_complex_sampled");
}

gc_status gc_beidou_b3i_code_gen_float(float* dest, int32_t prn, uint32_t chip_shift)
{
    GC_REQUIRE(dest, "gc_beidou_b3i_code_gen_float: dest is NULL");
    std::vector<int8_t> c(10230);
    gc_set_error("");
    if (!bds_b3i_chips(c.data(), prn, chip_shift))
        {
            if (!*gc_last_error()) gc_set_error("gc_beidou_b3i_code_gen_float: PRN %d not in 1..63", prn);
            return GC_ERR_INVALID;
        }
    for (int i = 0; i < 10230; i++) dest[i] = static_cast<float>(c[i]);
    return GC_OK;
}

gc_status gc_beidou_b3i_code_gen_complex_sampled(float* dest, uint32_t prn, int32_t fs, uint32_t chip_shift, int32_t* n_samples)
{
    GC_REQUIRE(dest && fs > 0, "gc_beidou_b3i_code_gen_complex_sampled: bad argument");
    std::vector<int8_t> c(10230);
    gc_set_error("");
    if (!bds_b3i_chips(c.data(), static_cast<int>(prn), chip_shift))
        {
            if (!*gc_last_error()) gc_set_error("gc_beidou_b3i_code_gen_complex_sampled: PRN %u not in 1..63", prn);
            return GC_ERR_INVALID;
        }
    const int n = sample_chips(dest, c.data(), 10230, 10230000, fs);
    if (n_samples) *n_samples = n;
    return GC_OK;
}

gc_status gc_galileo_e5_a_code_gen_complex_primary(float* dest, int32_t prn, const char* signal)
{
    GC_REQUIRE(dest && signal, "gc_galileo_e5_a_code_gen_complex_primary: NULL argument");
    GC_REQUIRE(signal[0] == '5' && (signal[1] == 'I' || signal[1] == 'Q' || signal[1] == 'X'),
        "gc_galileo_e5_a_code_gen_complex_primary: signal must be \"5I\", \"5Q\" or \"5X\"");
    std::vector<int8_t> ci(10230), cq(10230);
    gc_set_error("");
    if (prn < 1 || !galileo_e5a_component(ci.data(), static_cast<unsigned>(prn), 0) || !galileo_e5a_component(cq.data(), static_cast<unsigned>(prn), 1))
        {
            if (!*gc_last_error()) gc_set_error("gc_galileo_e5_a_code_gen_complex_primary: PRN %d not in 1..50", prn);
            return GC_ERR_INVALID;
        }
    for (int i = 0; i < 10230; i++)
        {
            dest[2 * i] = signal[1] == 'Q' ? 0.0f : static_cast<float>(ci[i]);
            dest[2 * i + 1] = signal[1] == 'I' ? 0.0f : static_cast<float>(cq[i]);
        }
    return GC_OK;
}

gc_status gc_galileo_e5_a_code_gen_complex_sampled(float* dest, const char* signal, uint32_t prn, int32_t fs, uint32_t chip_shift, int32_t* n_samples)
{
    GC_REQUIRE(dest && fs > 0, "gc_galileo_e5_a_code_gen_complex_sampled: bad argument");
    std::vector<float> code(2 * 10230);
    gc_status st = gc_galileo_e5_a_code_gen_complex_primary(code.data(), static_cast<int32_t>(prn), signal);
    if (st != GC_OK) return st;
    const unsigned CL = 10230;
    const int code_freq = 10230000;
    const unsigned spc = static_cast<unsigned>(static_cast<double>(fs) / (static_cast<double>(code_freq) / static_cast<double>(CL)));
    const unsigned delay = ((CL - chip_shift) % CL) * spc / CL;
    std::vector<float> re(CL), im(CL), rre, rim;
    for (unsigned i = 0; i < CL; i++)
        {
            re[i] = code[2 * i];
            im[i] = code[2 * i + 1];
        }
    if (fs != code_freq)
        {
            // resampler() picks whole complex chips: the same index for both parts
            rre.resize(spc);
            rim.resize(spc);
            resample(re.data(), rre.data(), static_cast<float>(code_freq), static_cast<float>(fs), CL, spc);
            resample(im.data(), rim.data(), static_cast<float>(code_freq), static_cast<float>(fs), CL, spc);
            re.swap(rre);
            im.swap(rim);
        }
    for (unsigned i = 0; i < spc; i++)
        {
            const unsigned d = (i + delay) % spc;
            dest[2 * d] = re[i];
            dest[2 * d + 1] = im[i];
        }
    if (n_samples) *n_samples = static_cast<int32_t>(spc);
    return GC_OK;
}

gc_status gc_secondary_code(const char* signal, uint32_t prn, char* dest, int32_t capacity, int32_t* length)
{
    GC_REQUIRE(signal && dest && capacity > 0, "gc_secondary_code: bad argument");
    std::string code;
    const std::string s(signal);
    if (s == "1C")
        code = "0011100000001010110110010";  // Galileo E1-C CS25_1
    else if (s == "B1" || s == "B3")
        code = "00000100110101001110";  // BeiDou NH20 (MEO / IGSO satellites)
    else if (s == "L5I")
        code = "0000110101";  // GPS L5 I5 NH10
    else if (s == "L5Q")
        code = "00000100110101001110";  // GPS L5 Q5 NH20
    else if (s == "5I" || s == "5Q")
        {
            gc_set_error("");
            if (!load_tables())
                {
                    if (!*gc_last_error()) gc_set_error("gc_secondary_code: tables unavailable");
                    return GC_ERR_INVALID;
                }
            if (s == "5I")
                code = g_tab.e5a_i_secondary;
            else
                {
                    GC_REQUIRE(prn >= 1 && static_cast<int>(prn) <= g_tab.n_e5a_q_secondary, "gc_secondary_code: no E5a-Q secondary code for PRN %u (1..%d)", prn,
                        g_tab.n_e5a_q_secondary);
                    code = g_tab.e5a_q_secondary.substr((prn - 1) * 100, 100);
                }
        }
    else
        return gc_fail(GC_ERR_INVALID, "gc_secondary_code: unknown signal \"%s\" (1C, B1, B3, L5I, L5Q, 5I, 5Q)", signal);
    GC_REQUIRE(static_cast<int>(code.size()) + 1 <= capacity, "gc_secondary_code: capacity %d too small for %d symbols", capacity, static_cast<int>(code.size()));
    std::memcpy(dest, code.c_str(), code.size() + 1);
    if (length) *length = static_cast<int32_t>(code.size());
    return GC_OK;
}


gc_status gc_loop_sync_for_signal(char system, const char* signal, uint32_t prn, int track_pilot, int extend_correlation_symbols, gc_loop_sync_conf* out)
{
    GC_REQUIRE(signal && out, "gc_loop_sync_for_signal: NULL argument");
    GC_REQUIRE(extend_correlation_symbols >= 1, "gc_loop_sync_for_signal: extend_correlation_symbols must be >= 1");
    const std::string sig(signal);
    gc_loop_sync_conf y;
    std::memset(&y, 0, sizeof y);
    y.extend_correlation_symbols = extend_correlation_symbols;
    y.bit_sync_min_time_s = 10.0f;
    std::string secondary;
    std::vector<int> preamble_bits;
    int preamble_symbols_per_bit = 0;
    if (system == 'G' && sig == "1C")
        {
            y.symbols_per_bit = 20;
            preamble_bits = {1, 0, 0, 0, 1, 0, 1, 1};  // GPS_PREAMBLE
            preamble_symbols_per_bit = 20;
        }
    else if (system == 'G' && sig == "2S")
        y.symbols_per_bit = 1;
    else if (system == 'G' && sig == "L5")
        {
            y.symbols_per_bit = 10;
            y.track_pilot = track_pilot ? 1 : 0;
            secondary = track_pilot ? "L5Q" : "L5I";
        }
    else if (system == 'E' && sig == "1B")
        {
            y.symbols_per_bit = 1;
            y.track_pilot = track_pilot ? 1 : 0;
            if (track_pilot) secondary = "1C";
        }
    else if (system == 'E' && sig == "5X")
        {
            y.symbols_per_bit = 20;
            y.track_pilot = track_pilot ? 1 : 0;
            if (track_pilot) secondary = "5Q";  // on the data component the secondary code is left to the telemetry decoder
        }
    else if (system == 'C' && (sig == "B1" || sig == "B3"))
        {
            if (prn > 0 && prn < 6)
                {
                    y.symbols_per_bit = 2;  // GEO satellites: D2, no NH code
                    preamble_bits = {1, 1, 1, 0, 0, 0, 1, 0, 0, 1, 0};
                    preamble_symbols_per_bit = 2;
                }
            else
                {
                    y.symbols_per_bit = 20;
                    secondary = sig;
                }
        }
    else
        return gc_fail(GC_ERR_INVALID, "gc_loop_sync_for_signal: unknown system / signal '%c' \"%s\"", system, signal);
    if (!secondary.empty())
        {
            int32_t len = 0;
            gc_status st = gc_secondary_code(secondary.c_str(), prn, y.secondary_code, (int32_t)sizeof y.secondary_code, &len);
            if (st != GC_OK) return st;
            y.secondary_code_length = len;
        }
    int n = 0;
    for (int bit : preamble_bits)
        for (int j = 0; j < preamble_symbols_per_bit; j++) y.preamble_symbols[n++] = bit ? 1 : -1;
    y.preamble_length_symbols = n;
    *out = y;
    return GC_OK;
}

}  // extern "C"
