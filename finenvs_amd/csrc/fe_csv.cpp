// fe_csv.cpp -- native CSV reader for the reference's bar files.
//
// Replaces read_data + force_market_hours of the reference
// (finenvs/environments/time_series_env.py:80-91: pandas.read_csv, a Datetime index,
// between_time("9:30", "15:59")) with one pass over an mmap of the file:
//   row = Date,Time,Open,High,Low,Close,Volume    (finenvs/data/README.md:7-9)
// Date is an opaque key ("2022-04-01" and "01/02/1998" both occur in the reference's
// fixtures); Time is HH:MM[:SS]; Volume is dropped (TSE:170).
//
// Numbers: std::from_chars (correctly rounded).  The reference parses with pandas' default
// converter ("high" = precise_xstrtod in pandas >= 1.2; 2.3.3 here), which is also correctly
// rounded for <= 15 significant digits and |decimal exponent| <= 22 -- every price file is in
// that regime; tests/test_csv_native.py pins the equality against pandas.
//
// Host-side code: all pointers are HOST pointers.  No HIP calls here.
#include <fcntl.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <charconv>
#include <functional>
#include <thread>
#include <unordered_map>
#include <vector>

#include "finenvs_amd.h"

extern "C" __attribute__((visibility("hidden"))) int fe_set_error(int code, const char *fmt, ...);  // fe_env.hip (library-internal)

namespace {

struct Mapped {
    const char *p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool ok = false;
    explicit Mapped(const char *path) {
        fd = open(path, O_RDONLY);
        if (fd < 0) return;
        struct stat st;
        if (fstat(fd, &st) != 0) return;
        n = (size_t)st.st_size;
        if (n == 0) {
            ok = true;
            return;
        }
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return;
        p = (const char *)m;
        ok = true;
    }
    ~Mapped() {
        if (p) munmap((void *)p, n);
        if (fd >= 0) close(fd);
    }
};

inline uint64_t fnv1a(const char *b, const char *e) {
    uint64_t h = 1469598103934665603ull;
    for (; b < e; ++b) h = (h ^ (unsigned char)*b) * 1099511628211ull;
    return h;
}

inline const char *trim_l(const char *b, const char *e) {
    while (b < e && (*b == ' ' || *b == '\t')) ++b;
    return b;
}
inline const char *trim_r(const char *b, const char *e) {
    while (e > b && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r')) --e;
    return e;
}

// HH:MM[:SS] -> second of day, -1 on malformed input
inline int64_t parse_time(const char *b, const char *e) {
    int64_t part[3] = {0, 0, 0};
    int k = 0, digits = 0;
    for (const char *p = b; p < e; ++p) {
        if (*p >= '0' && *p <= '9') {
            part[k] = part[k] * 10 + (*p - '0');
            ++digits;
        } else if (*p == ':' && k < 2 && digits > 0) {
            ++k;
            digits = 0;
        } else {
            return -1;
        }
    }
    if (k < 1 || digits == 0) return -1;
    return part[0] * 3600 + part[1] * 60 + part[2];
}

// Calendar dates in the spellings the reference's fixtures use (YYYY-MM-DD, MM/DD/YYYY; also with the
// other separator) -> yyyymmdd; -1 for anything else (the date then stays an opaque text key).
inline int64_t parse_date(const char *b, const char *e) {
    int64_t part[3] = {0, 0, 0};
    int len[3] = {0, 0, 0};
    int k = 0;
    char sep = 0;
    for (const char *p = b; p < e; ++p) {
        if (*p >= '0' && *p <= '9') {
            if (++len[k] > 4) return -1;
            part[k] = part[k] * 10 + (*p - '0');
        } else if ((*p == '-' || *p == '/') && k < 2 && len[k] > 0 && (sep == 0 || sep == *p)) {
            sep = *p;
            ++k;
        } else {
            return -1;
        }
    }
    if (k != 2 || len[2] == 0) return -1;
    int64_t y, mth, d;
    if (len[0] == 4 && len[1] <= 2 && len[2] <= 2) {  // year first
        y = part[0]; mth = part[1]; d = part[2];
    } else if (len[2] == 4 && len[0] <= 2 && len[1] <= 2) {  // month/day/year, the US spelling of IBM / OIH
        mth = part[0]; d = part[1]; y = part[2];
    } else {
        return -1;
    }
    if (mth < 1 || mth > 12 || d < 1 || d > 31) return -1;
    return y * 10000 + mth * 100 + d;
}

// 45-bit join key of a row's date: the calendar date where the text is one (so "2022-04-01" in one file
// joins "04/01/2022" in another), else bit 44 + 44 bits of the text's FNV-1a hash.  Together with the
// 17-bit second of day it fits an int64 without wrapping.
inline int64_t date_join_key(const char *b, const char *e) {
    const int64_t cal = parse_date(b, e);
    if (cal >= 0) return cal;
    return (int64_t)((fnv1a(b, e) >> 20) | (1ull << 44));
}

constexpr int64_t kOpen = (9 * 60 + 30) * 60;   // 09:30:00, first bar kept
constexpr int64_t kLast = (15 * 60 + 59) * 60;  // 15:59:00, last bar kept (between_time is inclusive)

// One thread's share of the file: whole lines [begin, end), parsed into its own vectors.  Parsing stops at the piece's
// first malformed line (error != 0; the rows before it are kept, so that the stitching pass can report the first error
// of the FILE with its absolute line number).
struct Piece {
    const char *begin = nullptr, *end = nullptr;
    std::vector<double> prices;   // 4 per kept row
    std::vector<uint64_t> key;    // date join key per kept row
    std::vector<int64_t> sec;     // second of day per kept row
    int64_t lines = 0;            // lines in the piece (all of them, or up to and including the malformed one)
    int error = 0;                // 1 = too few fields, 2 = bad time, 3 = bad number
    int error_arg = 0;            // field count / column
    int64_t error_line = 0;       // 1-based within the piece
};

void parse_piece(Piece &pc, bool market_hours_only) {
    const size_t guess = (size_t)(pc.end - pc.begin) / 40 + 16;  // a bar line is ~45-60 bytes
    pc.prices.reserve(guess * 4);
    pc.key.reserve(guess);
    pc.sec.reserve(guess);
    const char *prev_date = nullptr;
    size_t prev_len = 0;
    uint64_t prev_key = 0;
    const char *p = pc.begin, *end = pc.end;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        ++pc.lines;
        const char *lb = trim_l(p, le);
        const char *lt = trim_r(lb, le);
        p = nl ? nl + 1 : end;
        if (lb == lt) continue;  // blank line
        // split the first six fields
        const char *fb[7], *fe_[7];
        int nf = 0;
        const char *q = lb;
        while (nf < 7) {
            const char *c = (const char *)memchr(q, ',', (size_t)(lt - q));
            fb[nf] = trim_l(q, c ? c : lt);
            fe_[nf] = trim_r(fb[nf], c ? c : lt);
            ++nf;
            if (!c) break;
            q = c + 1;
        }
        auto stop = [&](int code, int arg) {
            pc.error = code; pc.error_arg = arg; pc.error_line = pc.lines;
        };
        if (nf < 6) { stop(1, nf); return; }
        const int64_t sec = parse_time(fb[1], fe_[1]);
        if (sec < 0) { stop(2, 0); return; }
        if (market_hours_only && (sec < kOpen || sec > kLast)) continue;
        double v[4];
        for (int k = 0; k < 4; ++k) {
            const char *b = fb[2 + k], *e = fe_[2 + k];
            if (b < e && *b == '+') ++b;
            auto r = std::from_chars(b, e, v[k]);
            if (r.ec != std::errc() || r.ptr != e) { stop(3, 3 + k); return; }
        }
        pc.prices.insert(pc.prices.end(), v, v + 4);
        // rows of one date are consecutive in every real file: compare the date text with the previous row's first
        const size_t dlen = (size_t)(fe_[0] - fb[0]);
        if (!(prev_date && dlen == prev_len && memcmp(prev_date, fb[0], dlen) == 0)) {
            prev_key = (uint64_t)date_join_key(fb[0], fe_[0]);
            prev_date = fb[0]; prev_len = dlen;
        }
        pc.key.push_back(prev_key);
        pc.sec.push_back(sec);
    }
}

// parse_piece for worker threads: an exception leaving a std::thread's function is std::terminate
void parse_piece_noexcept(Piece &pc, bool market_hours_only) noexcept {
    try {
        parse_piece(pc, market_hours_only);
    } catch (...) {  // std::bad_alloc from the vectors
        pc.error = 4;
    }
}

}  // namespace

extern "C" {

int64_t fe_csv_count_lines(const char *path) {
    if (!path) return fe_set_error(FE_ERR_ARG, "fe_csv_count_lines: null path");
    Mapped m(path);
    if (!m.ok) return fe_set_error(FE_ERR_ARG, "fe_csv_count_lines: cannot open %s", path);
    int64_t lines = 0;
    const char *p = m.p, *end = m.p + m.n;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        ++lines;
        if (!nl) break;
        p = nl + 1;
    }
    return lines;
}

static int64_t csv_read_impl(const char *path, int64_t capacity, int32_t market_hours_only, double *prices,
                             int64_t *day_id, int64_t *date_key, int64_t *second_of_day);

int64_t fe_csv_read(const char *path, int64_t capacity, int32_t market_hours_only, double *prices,
                    int64_t *day_id, int64_t *date_key, int64_t *second_of_day) {
    if (!path || !prices || !day_id || !second_of_day || capacity < 0)
        return fe_set_error(FE_ERR_ARG, "fe_csv_read: bad argument");
    try {  // nothing C++ leaves this extern "C" entry point
        return csv_read_impl(path, capacity, market_hours_only, prices, day_id, date_key, second_of_day);
    } catch (const std::exception &e) {
        return fe_set_error(FE_ERR_ARG, "fe_csv_read: %s: %s", path, e.what());
    } catch (...) {
        return fe_set_error(FE_ERR_ARG, "fe_csv_read: %s: unknown C++ exception", path);
    }
}

static int64_t csv_read_impl(const char *path, int64_t capacity, int32_t market_hours_only, double *prices,
                             int64_t *day_id, int64_t *date_key, int64_t *second_of_day) {
    Mapped m(path);
    if (!m.ok) return fe_set_error(FE_ERR_ARG, "fe_csv_read: cannot open %s", path);
    // Parsing is independent per line: the file is cut at line boundaries into one piece per thread (>= 4 MiB each, at
    // most 8), every piece is parsed into its own vectors, and a sequential pass stitches them together -- row order,
    // day ids in order of first appearance, the first error in FILE order -- exactly as one pass over the file would.
    unsigned want = std::thread::hardware_concurrency();
    if (want == 0) want = 1;
    if (want > 8) want = 8;
    size_t pieces = m.n / (4u << 20);
    if (pieces < 1) pieces = 1;
    if (pieces > want) pieces = want;
    std::vector<Piece> piece(pieces);
    {
        const char *cut = m.p, *end = m.p + m.n;
        for (size_t i = 0; i < pieces; ++i) {
            piece[i].begin = cut;
            const char *target = i + 1 == pieces ? end : m.p + m.n / pieces * (i + 1);
            if (target < cut) target = cut;
            if (i + 1 < pieces) {
                const char *nl = (const char *)memchr(target, '\n', (size_t)(end - target));
                cut = nl ? nl + 1 : end;
            } else {
                cut = end;
            }
            piece[i].end = cut;
        }
    }
    // No C++ exception may cross this extern "C" boundary: a worker that runs out of memory records error 4 in its piece
    // (parse_piece_noexcept); a thread that cannot be created (std::system_error) leaves its piece to the calling thread.
    try {
        std::vector<std::thread> workers;
        std::vector<char> started(pieces, 0);
        for (size_t i = 1; i < pieces; ++i) {
            try {
                workers.emplace_back(parse_piece_noexcept, std::ref(piece[i]), market_hours_only != 0);
                started[i] = 1;
            } catch (const std::exception &) {
                break;  // no more threads: the remaining pieces are parsed here
            }
        }
        parse_piece_noexcept(piece[0], market_hours_only != 0);
        for (size_t i = 1; i < pieces; ++i)
            if (!started[i]) parse_piece_noexcept(piece[i], market_hours_only != 0);
        for (auto &w : workers) w.join();
    } catch (const std::exception &e) {
        return fe_set_error(FE_ERR_ARG, "fe_csv_read: %s: %s", path, e.what());
    }
    for (size_t i = 0; i < pieces; ++i)
        if (piece[i].error == 4) return fe_set_error(FE_ERR_ARG, "fe_csv_read: %s: out of memory while parsing", path);
    std::unordered_map<uint64_t, int64_t> days;  // date key -> id in order of first appearance
    int64_t rows = 0, lines_before = 0;
    bool have_prev = false;
    uint64_t prev_key = 0;
    int64_t prev_id = 0;
    for (size_t i = 0; i < pieces; ++i) {
        const Piece &pc = piece[i];
        const int64_t n_ok = (int64_t)pc.sec.size();
        // (a piece's malformed line comes after all of its kept rows, so an overflowing row is the earlier event)
        if (rows + n_ok > capacity) return fe_set_error(FE_ERR_ARG, "fe_csv_read: capacity %lld too small", (long long)capacity);
        if (n_ok) {
            memcpy(prices + rows * 4, pc.prices.data(), sizeof(double) * 4 * (size_t)n_ok);
            memcpy(second_of_day + rows, pc.sec.data(), sizeof(int64_t) * (size_t)n_ok);
            for (int64_t r = 0; r < n_ok; ++r) {
                const uint64_t key = pc.key[r];
                int64_t id;
                if (have_prev && key == prev_key) {  // rows of one date are consecutive in every real file
                    id = prev_id;
                } else {
                    auto it = days.find(key);
                    if (it == days.end()) {
                        id = (int64_t)days.size();
                        days.emplace(key, id);
                    } else {
                        id = it->second;
                    }
                    have_prev = true; prev_key = key; prev_id = id;
                }
                day_id[rows + r] = id;
                if (date_key) date_key[rows + r] = (int64_t)key;
            }
            rows += n_ok;
        }
        if (pc.error != 0) {
            const long long line = (long long)(lines_before + pc.error_line);
            if (pc.error == 1) return fe_set_error(FE_ERR_ARG, "fe_csv_read: %s line %lld has %d fields, need >= 6", path, line, pc.error_arg);
            if (pc.error == 2) return fe_set_error(FE_ERR_ARG, "fe_csv_read: %s line %lld: bad time", path, line);
            return fe_set_error(FE_ERR_ARG, "fe_csv_read: %s line %lld: bad number in column %d", path, line, pc.error_arg);
        }
        lines_before += pc.lines;
    }
    return rows;
}

}  // extern "C"
