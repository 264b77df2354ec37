// the label work that precedes the plan, in native code (host only, no GPU).
//   wagg_resolve_cells   exact-equality join of segment labels to grid labels  (aggregations.py:27, S1)
//   wagg_backup_fill     per-row backup weight fill                            (aggregations.py:73, S4)
//   wagg_factorize_*     sorted-unique region labels and per-row codes         (aggregations.py:78, S3)
//   wagg_relabel         the pix_cent_x == 180.125 -> -179.875 relabel         (aggregations.py:144)
#include <algorithm>
#include <cmath>
#include <numeric>
#include <string>
#include <unordered_map>

#include "wagg_common.h"

namespace wagg {

// exact float equality as a hash key: -0.0 and +0.0 compare equal, NaN never matches anything
static inline bool key_of(double v, uint64_t &k) {
    if (v != v) return false;
    if (v == 0.0) v = 0.0;
    std::memcpy(&k, &v, sizeof(k));
    return true;
}

static int build_index(const double *lab, int64_t n, std::unordered_map<uint64_t, int64_t> &m, const char *name) {
    m.reserve((size_t)n * 2);
    for (int64_t i = 0; i < n; ++i) {
        uint64_t k;
        if (!key_of(lab[i], k)) continue;                 // a NaN grid label can never be selected
        if (!m.emplace(k, i).second) {
            set_error("grid coordinate '%s' is not unique (value %.17g at %lld)", name, lab[i], (long long)i);
            return WAGG_EINVAL;
        }
    }
    return WAGG_OK;
}

}  // namespace wagg

extern "C" int wagg_resolve_cells(const double *lat, int64_t nlat, const double *lon, int64_t nlon,
                                  const double *seg_lat, const double *seg_lon, int64_t nseg,
                                  int lon_major, int32_t *cell_idx, int64_t *bad_row) {
    using namespace wagg;
    WAGG_REQUIRE(lat && lon && nlat > 0 && nlon > 0, "empty grid");
    WAGG_REQUIRE(nseg == 0 || (seg_lat && seg_lon && cell_idx), "NULL segment arrays");
    WAGG_REQUIRE(nlat * nlon < (int64_t)0x7fffffff, "grid too large for int32 cell indices");
    if (bad_row) *bad_row = -1;
    try {
        std::unordered_map<uint64_t, int64_t> mlat, mlon;
        int rc = build_index(lat, nlat, mlat, "lat");
        if (rc == WAGG_OK) rc = build_index(lon, nlon, mlon, "lon");
        if (rc != WAGG_OK) return rc;
        for (int64_t i = 0; i < nseg; ++i) {
            uint64_t ka, ko;
            const bool oka = key_of(seg_lat[i], ka), oko = key_of(seg_lon[i], ko);
            const auto ia = oka ? mlat.find(ka) : mlat.end();
            const auto io = oko ? mlon.find(ko) : mlon.end();
            if (ia == mlat.end() || io == mlon.end()) {      // S1: no tolerance, no nearest neighbour
                if (bad_row) *bad_row = i;
                set_error("segment row %lld: label (lat %.17g, lon %.17g) not found in the grid", (long long)i,
                          seg_lat[i], seg_lon[i]);
                return WAGG_EKEY;
            }
            cell_idx[i] = (int32_t)(lon_major ? io->second * nlat + ia->second : ia->second * nlon + io->second);
        }
    } catch (const std::bad_alloc &) {
        set_error("host allocation failed");
        return WAGG_ENOMEM;
    }
    return WAGG_OK;
}

extern "C" int wagg_backup_fill(const double *w, const double *backup, int64_t n, double *w_eff) {
    using namespace wagg;
    WAGG_REQUIRE(n == 0 || (w && backup && w_eff), "NULL argument");
    for (int64_t i = 0; i < n; ++i) w_eff[i] = (w[i] > 0.0) ? w[i] : backup[i];   // NaN > 0 is false
    return WAGG_OK;
}

extern "C" int wagg_relabel(double *values, int64_t n, double from, double to) {
    using namespace wagg;
    WAGG_REQUIRE(n == 0 || values, "NULL argument");
    for (int64_t i = 0; i < n; ++i) if (values[i] == from) values[i] = to;
    return WAGG_OK;
}

extern "C" int wagg_factorize_i64(const int64_t *labels, const uint8_t *isnull, int64_t n, int32_t *codes,
                                  int64_t *uniq, int64_t *n_uniq) {
    using namespace wagg;
    WAGG_REQUIRE(n == 0 || (labels && codes && uniq), "NULL argument");
    WAGG_REQUIRE(n_uniq != nullptr, "NULL argument");
    try {
        std::vector<int64_t> u;
        u.reserve((size_t)n);
        for (int64_t i = 0; i < n; ++i) if (!isnull || !isnull[i]) u.push_back(labels[i]);
        std::sort(u.begin(), u.end());
        u.erase(std::unique(u.begin(), u.end()), u.end());
        WAGG_REQUIRE(u.size() < (size_t)0x7fffffff, "too many distinct labels");
        for (int64_t i = 0; i < n; ++i)
            codes[i] = (isnull && isnull[i]) ? -1 : (int32_t)(std::lower_bound(u.begin(), u.end(), labels[i]) - u.begin());
        std::copy(u.begin(), u.end(), uniq);
        *n_uniq = (int64_t)u.size();
    } catch (const std::bad_alloc &) {
        set_error("host allocation failed");
        return WAGG_ENOMEM;
    }
    return WAGG_OK;
}

// fixed-width byte strings (numpy dtype 'S<width>', NUL padded); order = bytewise = code-point order
// of the UTF-8 text.  uniq_rows[j] = a row that carries the j-th label.
extern "C" int wagg_factorize_bytes(const char *buf, int64_t width, const uint8_t *isnull, int64_t n,
                                    int32_t *codes, int64_t *uniq_rows, int64_t *n_uniq) {
    using namespace wagg;
    WAGG_REQUIRE(n == 0 || (buf && codes && uniq_rows), "NULL argument");
    WAGG_REQUIRE(width > 0 && n_uniq != nullptr, "bad width / NULL argument");
    try {
        // pass 1: distinct labels through an open-addressing table (FNV-1a over the fixed width);
        // first[j] = first row that carries distinct label j, tmp code = j
        size_t cap = 64;
        while (cap < (size_t)n * 2) cap <<= 1;
        std::vector<int32_t> slot(cap, -1);
        std::vector<int64_t> first;
        const size_t w = (size_t)width;
        for (int64_t i = 0; i < n; ++i) {
            if (isnull && isnull[i]) { codes[i] = -1; continue; }
            const unsigned char *p = reinterpret_cast<const unsigned char *>(buf) + (size_t)i * w;
            uint64_t h = 1469598103934665603ull;
            for (size_t k = 0; k < w; ++k) { h ^= p[k]; h *= 1099511628211ull; }
            size_t at = (size_t)(h ^ (h >> 29)) & (cap - 1);
            for (;;) {
                const int32_t j = slot[at];
                if (j < 0) {
                    WAGG_REQUIRE(first.size() < (size_t)0x7fffffff, "too many distinct labels");
                    slot[at] = (int32_t)first.size();
                    codes[i] = (int32_t)first.size();
                    first.push_back(i);
                    break;
                }
                if (std::memcmp(buf + (size_t)first[(size_t)j] * w, p, w) == 0) { codes[i] = j; break; }
                at = (at + 1) & (cap - 1);
            }
        }
        // pass 2: sort the distinct labels bytewise, renumber
        std::vector<int32_t> order(first.size()), rank(first.size());
        for (size_t j = 0; j < order.size(); ++j) order[j] = (int32_t)j;
        std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            return std::memcmp(buf + (size_t)first[(size_t)a] * w, buf + (size_t)first[(size_t)b] * w, w) < 0; });
        for (size_t r = 0; r < order.size(); ++r) { rank[(size_t)order[r]] = (int32_t)r; uniq_rows[r] = first[(size_t)order[r]]; }
        for (int64_t i = 0; i < n; ++i) if (codes[i] >= 0) codes[i] = rank[(size_t)codes[i]];
        *n_uniq = (int64_t)first.size();
    } catch (const std::bad_alloc &) {
        set_error("host allocation failed");
        return WAGG_ENOMEM;
    }
    return WAGG_OK;
}
