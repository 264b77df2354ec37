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
        std::vector<int64_t> rows;
        rows.reserve((size_t)n);
        for (int64_t i = 0; i < n; ++i) if (!isnull || !isnull[i]) rows.push_back(i);
        auto less = [&](int64_t a, int64_t b) { return std::memcmp(buf + a * width, buf + b * width, (size_t)width) < 0; };
        auto same = [&](int64_t a, int64_t b) { return std::memcmp(buf + a * width, buf + b * width, (size_t)width) == 0; };
        std::sort(rows.begin(), rows.end(), less);
        int64_t nu = 0;
        for (size_t i = 0; i < rows.size(); ++i) {
            if (i == 0 || !same(rows[i - 1], rows[i])) uniq_rows[nu++] = rows[i];
            codes[rows[i]] = (int32_t)(nu - 1);
        }
        if (isnull) for (int64_t i = 0; i < n; ++i) if (isnull[i]) codes[i] = -1;
        *n_uniq = nu;
    } catch (const std::bad_alloc &) {
        set_error("host allocation failed");
        return WAGG_ENOMEM;
    }
    return WAGG_OK;
}
