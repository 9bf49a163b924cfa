// tts_stats.cpp -- CPU-only statistics of the transposed tile stream packer for candidate geometries (not product code).
//   g++ -O2 -fopenmp -Ihispmv_amd/csrc tools/tts_stats.cpp hispmv_amd/csrc/build/tts.o hispmv_amd/csrc/build/prep.o -o /tmp/tts_stats
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include "hispmv_tts.h"
using namespace hispmv;

static Csr gen(int rows, int cols, long long nnz, bool powerlaw, unsigned seed) {
    Csr m; m.rows = rows; m.cols = cols;
    std::mt19937_64 g(seed);
    std::vector<double> w(rows);
    std::vector<int> perm(rows); std::iota(perm.begin(), perm.end(), 0); std::shuffle(perm.begin(), perm.end(), g);
    double sum = 0;
    for (int i = 0; i < rows; ++i) { w[i] = powerlaw ? 1.0 / std::pow(perm[i] + 100.0, 0.8) : 1.0; sum += w[i]; }
    m.row_ptr.assign(rows + 1, 0);
    for (int i = 0; i < rows; ++i) { std::poisson_distribution<int> d(w[i] * nnz / sum); m.row_ptr[i + 1] = m.row_ptr[i] + d(g); }
    const long long n = m.row_ptr[rows];
    m.col.resize(n); m.val.resize(n);
#pragma omp parallel for schedule(dynamic, 4096)
    for (int i = 0; i < rows; ++i) {
        std::mt19937_64 gg(seed * 7919ull + i);
        for (long long k = m.row_ptr[i]; k < m.row_ptr[i + 1]; ++k) { m.col[k] = (int)(gg() % (unsigned long long)cols); m.val[k] = 1.0f; }
        std::sort(m.col.begin() + m.row_ptr[i], m.col.begin() + m.row_ptr[i + 1]);
    }
    return m;
}
static Csr col_part(const Csr& m, int c0, int c1) {
    Csr t; t.rows = m.rows; t.cols = m.cols; t.row_ptr.assign(m.rows + 1, 0);
    for (int i = 0; i < m.rows; ++i) {
        const int* b = m.col.data() + m.row_ptr[i]; const int* e = m.col.data() + m.row_ptr[i + 1];
        const int* lo = std::lower_bound(b, e, c0); const int* hi = std::lower_bound(b, e, c1);
        t.row_ptr[i + 1] = t.row_ptr[i] + (hi - lo);
        t.col.insert(t.col.end(), lo, hi); t.val.insert(t.val.end(), m.val.data() + (lo - m.col.data()), m.val.data() + (hi - m.col.data()));
    }
    return t;
}
int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 1632803; const long long nnz = argc > 2 ? atoll(argv[2]) : 30622600; const bool pl = argc > 3 ? atoi(argv[3]) : 1;
    Csr m = gen(rows, rows, nnz, pl, 1);
    printf("matrix %d x %d nnz %lld\n", m.rows, m.cols, (long long)m.nnz());
    struct G { int parts, slots, rows, tiles; bool zf; };
    for (G g : {G{1, 28 * 1024, 8192, 256, false}, G{1, 28 * 1024, 8192, 256, true}, G{2, 23 * 1024, 16384, 128, true}, G{2, 26 * 1024, 13 * 1024, 128, true}, G{2, 25 * 1024, 14 * 1024, 128, true}}) {
        if (g.rows == 0) continue;
        long long slots = 0, fill = 0, pads = 0, nt = 0, nb = 0, mx = 0; double lines = 0; long long sl = 0;
        for (int p = 0; p < g.parts; ++p) {
            std::vector<int32_t> cuts = tts_column_cuts(m, g.parts);
            Csr part = g.parts == 1 ? m : csr_column_range(m, p == 0 ? 0 : cuts[p - 1], p + 1 == g.parts ? m.cols : cuts[p]);
            TtsGeometry geo; geo.max_slots = g.slots; geo.max_rows = g.rows; geo.tiles_wanted = g.tiles; geo.zero_fill = g.zf;
            TtsStream s = build_tts(part, 0, geo);
            slots += s.total_slots; fill += s.n_fillers; pads += s.n_pad_words; nt += s.tiles.size(); nb += s.blocks.size(); mx = std::max(mx, (long long)s.max_tile_slots);
            lines += s.lines_per_gather * s.col_base.size(); sl += s.col_base.size();
        }
        printf("parts %d slots %d rows %d: tiles %lld blocks %lld slices %lld slots %lld (fillers %lld = %.1f%%, pad words %lld) max tile %lld mean %lld lines/gather %.2f\n",
               g.parts, g.slots, g.rows, nt, nb, sl, slots, fill, 100.0 * fill / m.nnz(), pads, mx, slots / nt, lines / sl);
        fflush(stdout);
    }
}
