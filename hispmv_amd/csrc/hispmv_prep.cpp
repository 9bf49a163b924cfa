// hispmv_prep.cpp -- see hispmv_prep.h.  Host-only, OpenMP.
#include "hispmv_prep.h"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <stdexcept>
#ifdef _OPENMP
#include <omp.h>
#include <sched.h>

#include <mutex>
#endif

namespace hispmv {

// ---------------------------------------------------------------------------
// MatrixMarket
// ---------------------------------------------------------------------------
static std::string lower(std::string s) {
    for (auto& ch : s) ch = (char)std::tolower((unsigned char)ch);
    return s;
}

Coo read_mtx(const std::string& path, MtxFlavor flavor) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("Error: Unable to open file " + path);
    std::fseek(f, 0, SEEK_END);
    long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::string buf((size_t)std::max(0L, sz), '\0');
    if (sz > 0 && std::fread(&buf[0], 1, (size_t)sz, f) != (size_t)sz) { std::fclose(f); throw std::runtime_error("Error: short read on " + path); }
    std::fclose(f);

    const char* p = buf.c_str();
    const char* end = p + buf.size();
    auto next_line = [&](const char*& b, const char*& e) -> bool {
        if (p >= end) return false;
        b = p;
        const char* nl = (const char*)std::memchr(p, '\n', (size_t)(end - p));
        e = nl ? nl : end;
        p = nl ? nl + 1 : end;
        return true;
    };
    const char *lb, *le;
    if (!next_line(lb, le)) throw std::runtime_error("Error: Not a valid Matrix Market file.");
    std::string tok[5];
    {
        std::string line(lb, le);
        size_t pos = 0;
        for (int i = 0; i < 5; i++) {
            while (pos < line.size() && std::isspace((unsigned char)line[pos])) pos++;
            size_t st = pos;
            while (pos < line.size() && !std::isspace((unsigned char)line[pos])) pos++;
            tok[i] = line.substr(st, pos - st);
        }
    }
    if (tok[0] != "%%MatrixMarket" || lower(tok[1]) != "matrix") throw std::runtime_error("Error: Not a valid Matrix Market file.");
    const std::string fmt = lower(tok[2]), dt = lower(tok[3]), sym = lower(tok[4]);
    if (fmt != "coordinate") throw std::runtime_error("Error: Only sparse matrices in 'coordinate' format are supported.");
    if (dt != "real" && dt != "integer" && dt != "pattern") throw std::runtime_error("Error: Unsupported data type.");
    if (sym != "general" && sym != "symmetric" && sym != "skew-symmetric")
        throw std::runtime_error("Error: Unsupported symmetry type. Only 'general', 'symmetric', and 'skew-symmetric' are supported.");
    const bool pattern = dt == "pattern";
    const bool mirror_sym = sym == "symmetric";
    const bool mirror_skew = (sym == "skew-symmetric") && flavor == kFlavorCommon;   // cpu/ does not mirror skew

    long M = 0, N = 0, nz = 0;
    while (next_line(lb, le)) {
        if (lb < le && *lb == '%') continue;
        std::string line(lb, le);
        if (std::sscanf(line.c_str(), "%ld %ld %ld", &M, &N, &nz) == 3) break;
    }
    if (M <= 0 || N <= 0 || nz < 0 || M > INT32_MAX || N > INT32_MAX) throw std::runtime_error("Error: bad size line in " + path);

    Coo out;
    out.rows = (int32_t)M; out.cols = (int32_t)N;
    // Entry lines, parsed in parallel: the byte range is cut at line starts, every chunk runs the same loop (an entry =
    // "row col [value]", the rest of the line ignored; entries before the size line's count only; the first malformed
    // entry ends the matrix), and the chunks are concatenated in file order.
    struct Chunk { const char* b; const char* e; std::vector<int32_t> r, c; std::vector<float> v; long iters = 0; bool stopped = false; std::string err; };
    auto parse = [&](Chunk& k, long limit) {
        const char* q = k.b;
        k.r.clear(); k.c.clear(); k.v.clear(); k.iters = 0; k.stopped = false;
        while (k.iters < limit && q < k.e) {
            char* e1;
            while (q < k.e && std::isspace((unsigned char)*q)) q++;      // blank lines do not count as entries
            if (q >= k.e) break;
            long r = std::strtol(q, &e1, 10);
            if (e1 == q) { k.stopped = true; break; }
            q = e1;
            long c = std::strtol(q, &e1, 10);
            if (e1 == q) { k.stopped = true; break; }
            q = e1;
            float v = 1.0f;
            if (!pattern) { v = std::strtof(q, &e1); if (e1 == q) { k.stopped = true; break; } q = e1; }
            while (q < end && *q != '\n') q++;   // ignore the rest of the line
            k.iters++;
            uint32_t bits; std::memcpy(&bits, &v, 4);
            const bool drop = (flavor == kFlavorCommon) ? (v == 0.0f) : (bits == 0u);
            if (drop) continue;
            if (r < 1 || c < 1 || r > M || c > N) { k.err = "Error: entry out of range in " + path; k.stopped = true; break; }
            k.r.push_back((int32_t)(r - 1)); k.c.push_back((int32_t)(c - 1)); k.v.push_back(v);
            if (r != c) {
                if (mirror_sym) { k.r.push_back((int32_t)(c - 1)); k.c.push_back((int32_t)(r - 1)); k.v.push_back(v); }
                else if (mirror_skew) { k.r.push_back((int32_t)(c - 1)); k.c.push_back((int32_t)(r - 1)); k.v.push_back(-v); }
            }
        }
    };
    long chunk_bytes = 4 << 20;
    if (const char* env = std::getenv("HISPMV_MTX_CHUNK_BYTES")) chunk_bytes = std::max(16L, std::atol(env));
    std::vector<Chunk> chunks;
    for (const char* b = p; b < end;) {
        const char* e = (end - b > chunk_bytes) ? b + chunk_bytes : end;
        if (e < end) { const char* nl = (const char*)std::memchr(e, '\n', (size_t)(end - e)); e = nl ? nl + 1 : end; }
        chunks.push_back(Chunk{b, e, {}, {}, {}});
        b = e;
    }
#pragma omp parallel for num_threads(host_threads()) schedule(dynamic, 1)
    for (long i = 0; i < (long)chunks.size(); ++i) parse(chunks[(size_t)i], nz);
    long taken = 0;
    for (Chunk& k : chunks) {
        if (taken >= nz) break;
        if (taken + k.iters > nz) parse(k, nz - taken);          // the file holds more lines than its size line says
        if (!k.err.empty()) throw std::runtime_error(k.err);
        out.r.insert(out.r.end(), k.r.begin(), k.r.end());
        out.c.insert(out.c.end(), k.c.begin(), k.c.end());
        out.v.insert(out.v.end(), k.v.begin(), k.v.end());
        taken += k.iters;
        if (k.stopped) break;
    }
    return out;
}

// ---------------------------------------------------------------------------
// COO -> CSR
// ---------------------------------------------------------------------------
// Stable sort by column inside each row (rows already ascending are skipped): everything downstream (column windows
// of a slice, column tiles) takes a row's first / last column from its ends.
void sort_rows_by_column(Csr& m) {
    const int32_t rows = m.rows;
#pragma omp parallel num_threads(host_threads())
    {
        std::vector<std::pair<int32_t, float>> tmp;
#pragma omp for schedule(dynamic, 256)
        for (int32_t i = 0; i < rows; ++i) {
            const int64_t s = m.row_ptr[i], e = m.row_ptr[(size_t)i + 1];
            if (e - s < 2) continue;
            bool sorted = true;
            for (int64_t k = s + 1; k < e; ++k) if (m.col[k] < m.col[k - 1]) { sorted = false; break; }
            if (sorted) continue;
            tmp.resize((size_t)(e - s));
            for (int64_t k = s; k < e; ++k) tmp[k - s] = {m.col[k], m.val[k]};
            std::stable_sort(tmp.begin(), tmp.end(),
                             [](const std::pair<int32_t, float>& a, const std::pair<int32_t, float>& b) { return a.first < b.first; });
            for (int64_t k = s; k < e; ++k) { m.col[k] = tmp[k - s].first; m.val[k] = tmp[k - s].second; }
        }
    }
}

// First touch of a freshly allocated buffer by all threads (one write per page): a copy into it -- a device-to-host download,
// done by one runtime thread -- then finds the pages mapped instead of faulting them in one by one.
void prefault_parallel(void* p, size_t bytes) {
    char* c = (char*)p;
    const long long pages = (long long)((bytes + 4095) / 4096);
#pragma omp parallel for num_threads(host_threads()) schedule(static)
    for (long long i = 0; i < pages; ++i) c[(size_t)i * 4096] = 0;
}

// OpenMP threads of the host preprocessor: the CPUs this process may actually USE -- the cgroup quota (v2 cpu.max, v1
// cpu.cfs_quota_us / cpu.cfs_period_us) and the affinity mask -- not the CPUs it can see.  The GPU boxes show 256 logical CPUs
// behind a 16-CPU quota: with 256 threads the packer of soc-Pokec's shape spent 1.55 s where 16 threads take a third of that.
// HISPMV_HOST_THREADS overrides (bench.py --gpus N gives every rank its share: quota / LOCAL_WORLD_SIZE), else OMP_NUM_THREADS
// is honoured.  The count stays PRIVATE to the library: every parallel region carries num_threads(host_threads()); the
// process-global OpenMP setting of the embedding application (torch CPU ops, an MKL baseline sharing the runtime) is never
// touched (until round 3 the first hispmv_create called omp_set_num_threads).
int host_threads() {
    static std::once_flag once;
    static int chosen = 1;
    std::call_once(once, [] {
        auto env_int = [](const char* name) { const char* e = std::getenv(name); return e ? std::atoi(e) : 0; };
        int n = env_int("HISPMV_HOST_THREADS");
        if (n <= 0) n = env_int("OMP_NUM_THREADS");
        if (n <= 0) {
            n = omp_get_num_procs();
            cpu_set_t set;
            if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a >= 1 && a < n) n = a; }
            auto quota_of = [](const char* path_quota, const char* path_period) -> long long {
                long long q = -1, per = 0;
                if (FILE* f = std::fopen(path_quota, "r")) {
                    char tok[64] = {0};
                    if (path_period) { if (std::fscanf(f, "%63s", tok) == 1) q = std::atoll(tok); }
                    else if (std::fscanf(f, "%63s %lld", tok, &per) == 2 && std::strcmp(tok, "max") != 0) q = std::atoll(tok);
                    std::fclose(f);
                }
                if (path_period) { if (FILE* f = std::fopen(path_period, "r")) { if (std::fscanf(f, "%lld", &per) != 1) per = 0; std::fclose(f); } }
                return (q > 0 && per > 0) ? (q + per - 1) / per : -1;
            };
            long long quota = quota_of("/sys/fs/cgroup/cpu.max", nullptr);
            if (quota < 0) quota = quota_of("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
            if (quota >= 1 && quota < n) n = (int)quota;
        }
        chosen = std::max(1, n);
    });
    return chosen;
}

Csr coo_to_csr(int32_t rows, int32_t cols, int64_t nnz, const int32_t* r, const int32_t* c, const float* v) {
    if (rows < 0 || cols < 0 || nnz < 0) throw std::out_of_range("negative dimension");
    // the kernels address x, bias and y with 32-bit byte offsets (buffer descriptors)
    if (rows >= (1 << 30) || cols >= (1 << 30)) throw std::out_of_range("dimension >= 2^30 is not supported");
    Csr m;
    m.rows = rows; m.cols = cols;
    m.row_ptr.assign((size_t)rows + 1, 0);
    m.col.resize((size_t)nnz);
    m.val.resize((size_t)nnz);

    bool bad = false;
#pragma omp parallel for num_threads(host_threads()) schedule(static) reduction(|| : bad)
    for (int64_t i = 0; i < nnz; ++i)
        bad = bad || (r[i] < 0 || r[i] >= rows || c[i] < 0 || c[i] >= cols);
    if (bad) throw std::out_of_range("COO index outside matrix dimensions");

    // Stable counting sort by row, parallel over ROW RANGES: every thread streams the whole COO once and
    // takes the entries of its own rows (input order inside a row is kept; the random writes of a thread stay
    // inside its block of the output).  The reference does this step serially with a vector per (tile,row)
    // (spmv-helper.cpp:139-227), ~87 % of its preprocessing time.
    int nt = 1;
#ifdef _OPENMP
    nt = host_threads();
#endif
    nt = (int)std::min<int64_t>(nt, std::max<int64_t>(1, nnz / (1 << 20)));
    if (nt <= 1) {
        for (int64_t i = 0; i < nnz; ++i) m.row_ptr[(size_t)r[i] + 1]++;
        for (int32_t i = 0; i < rows; ++i) m.row_ptr[(size_t)i + 1] += m.row_ptr[i];
        std::vector<int64_t> cur(m.row_ptr.begin(), m.row_ptr.end() - 1);
        for (int64_t i = 0; i < nnz; ++i) {
            int64_t k = cur[r[i]]++;
            m.col[k] = c[i];
            m.val[k] = v[i];
        }
    } else {
#pragma omp parallel num_threads(nt)
        {
            const int t = omp_get_thread_num();
            const int32_t r0 = (int32_t)((int64_t)rows * t / nt), r1 = (int32_t)((int64_t)rows * (t + 1) / nt);
            for (int64_t i = 0; i < nnz; ++i) { const int32_t ri = r[i]; if (ri >= r0 && ri < r1) m.row_ptr[(size_t)ri + 1]++; }
        }
        for (int32_t i = 0; i < rows; ++i) m.row_ptr[(size_t)i + 1] += m.row_ptr[i];
        std::vector<int64_t> cur(m.row_ptr.begin(), m.row_ptr.end() - 1);
#pragma omp parallel num_threads(nt)
        {
            const int t = omp_get_thread_num();
            const int32_t r0 = (int32_t)((int64_t)rows * t / nt), r1 = (int32_t)((int64_t)rows * (t + 1) / nt);
            for (int64_t i = 0; i < nnz; ++i) {
                const int32_t ri = r[i];
                if (ri < r0 || ri >= r1) continue;
                const int64_t k = cur[ri]++;
                m.col[k] = c[i];
                m.val[k] = v[i];
            }
        }
    }
    sort_rows_by_column(m);
    return m;
}

// ---------------------------------------------------------------------------
// CSR -> slice stream
// ---------------------------------------------------------------------------
// Element offset of every row in the stream: every row owns >= 1 element (an empty row gets one zero-valued filler so
// that "one row end per row" holds and row ids need no list), and -- row-aligned slices -- when every row fits a slice
// and it costs <= 6 % of the stream, a row that would be cut by a slice boundary starts at the next slice instead and
// the row before it is extended to the boundary with zero-valued elements (its row end moves to the last of them): no
// row is cut => no carry chain, no fix-up launch.  Waste ~ half a row per slice: 2.4 % for PFlow_742 (50 per row), too
// much for rows of hundreds of elements.  O(rows), sequential: shared by the host and the device packer.
std::vector<int64_t> stream_row_offsets(int32_t R, const int64_t* row_ptr) {
    const int64_t S = kSliceElems;
    std::vector<int64_t> eoff((size_t)R + 1, 0);
    int64_t max_len = 1, plain = 0;
    for (int32_t i = 0; i < R; ++i) {
        const int64_t len = std::max<int64_t>(row_ptr[(size_t)i + 1] - row_ptr[i], 1);
        max_len = std::max(max_len, len);
        plain += len;
    }
    bool align = false;
    const char* env_align = std::getenv("HISPMV_ROW_ALIGN");          // "0" switches the alignment off (experiments)
    if (max_len <= S && max_len > 1 && !(env_align && env_align[0] == '0')) {
        int64_t pos = 0;
        for (int32_t i = 0; i < R; ++i) {
            const int64_t len = std::max<int64_t>(row_ptr[(size_t)i + 1] - row_ptr[i], 1);
            const int64_t room = S - pos % S;
            if (room < S && len > room) pos += room;
            pos += len;
        }
        align = (pos - plain) * 100 <= 6 * plain;
    }
    int64_t pos = 0;
    for (int32_t i = 0; i < R; ++i) {
        const int64_t len = std::max<int64_t>(row_ptr[(size_t)i + 1] - row_ptr[i], 1);
        const int64_t room = S - pos % S;
        if (align && room < S && len > room) pos += room;      // row i-1 is extended over [pos, pos + room)
        eoff[i] = pos;
        pos += len;
    }
    eoff[R] = pos;
    return eoff;
}

SliceStream build_stream(const Csr& m) {
    SliceStream st;
    st.rows = m.rows; st.cols = m.cols; st.nnz = m.nnz();
    const int32_t R = m.rows;
    const int64_t S = kSliceElems;
    const std::vector<int64_t> eoff = stream_row_offsets(R, m.row_ptr.data());
    st.n_elems = eoff[R];
    st.n_slices = (st.n_elems + S - 1) / S;
    st.words.assign((size_t)(st.n_slices * S), pack_elem(0.0f, 0, false));
    st.hdr.assign((size_t)st.n_slices, SliceHdr{0, 0, 0, 1});

#pragma omp parallel for num_threads(host_threads()) schedule(dynamic, 1024)
    for (int32_t i = 0; i < R; ++i) {
        const int64_t s = m.row_ptr[i], e = m.row_ptr[(size_t)i + 1];
        const int64_t span = eoff[(size_t)i + 1] - eoff[i];          // real elements (or one filler), then the extension
        uint64_t* w = st.words.data() + eoff[i];
        const int32_t fill_col = e > s ? m.col[e - 1] : 0;            // a column the slice's window holds anyway
        for (int64_t k = 0; k < span; ++k)
            w[k] = k < e - s ? pack_elem(m.val[s + k], m.col[s + k], k + 1 == span) : pack_elem(0.0f, fill_col, k + 1 == span);
    }

    std::vector<uint8_t> has_fix((size_t)st.n_slices, 0);
#pragma omp parallel for num_threads(host_threads()) schedule(dynamic, 64)
    for (int64_t sl = 0; sl < st.n_slices; ++sl) {
        const int64_t b = sl * S, e = std::min(b + S, st.n_elems);
        // rows that END before element b
        const int64_t rb = std::upper_bound(eoff.begin() + 1, eoff.end(), b) - (eoff.begin() + 1);
        const int64_t rb_next = std::upper_bound(eoff.begin() + 1, eoff.end(), b + S) - (eoff.begin() + 1);
        SliceHdr h{(int32_t)rb, 0, 0, 1};
        if (rb < R && eoff[rb] < b && rb_next > rb) {    // first row ending here began earlier
            h.chain_len = (int32_t)(sl - eoff[rb] / S);
            has_fix[sl] = 1;
        }
        // column window over the real elements; fillers/padding take the window's base
        int32_t cmin = INT32_MAX, cmax = -1;
        for (int64_t r = rb; r < R && eoff[r] < e; ++r) {
            const int64_t len = m.row_ptr[(size_t)r + 1] - m.row_ptr[r];
            if (len == 0) continue;
            const int64_t k0 = m.row_ptr[r] + std::max<int64_t>(0, b - eoff[r]);
            const int64_t k1 = m.row_ptr[r] + std::min<int64_t>(len, e - eoff[r]);
            if (k0 < k1) { cmin = std::min(cmin, m.col[k0]); cmax = std::max(cmax, m.col[k1 - 1]); }
        }
        if (cmax < 0) { cmin = 0; cmax = 0; }
        h.x_base = cmin; h.x_span = cmax - cmin + 1;
        if (cmin != 0) {
            for (int64_t r = rb; r < R && eoff[r] < e; ++r)
                if (m.row_ptr[(size_t)r + 1] == m.row_ptr[r])          // an empty row: its filler(s) inside this slice
                    for (int64_t k = std::max(eoff[r], b); k < std::min(eoff[(size_t)r + 1], e); ++k)
                        st.words[k] = pack_elem(0.0f, cmin, k + 1 == eoff[(size_t)r + 1]);
            for (int64_t k = e; k < b + S; ++k) st.words[k] = pack_elem(0.0f, cmin, false);
        }
        st.hdr[sl] = h;
    }
    for (int64_t sl = 0; sl < st.n_slices; ++sl)
        if (has_fix[sl]) {
            const SliceHdr& h = st.hdr[sl];
            st.fix.push_back(FixEntry{h.row_base, (int32_t)(sl - h.chain_len), h.chain_len, 0});
        }
    return st;
}

}  // namespace hispmv
